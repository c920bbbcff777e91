#!/usr/bin/env python
"""Per-kernel averages of the rocprofv3 --pmc passes written by tools/profile.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            for name in ("kstrongest_rows_kernel", "kstrong_image_kernel", "kstrongest_cols_kernel", "kstrong_cloud_kernel", "cacfar_rows_kernel", "cacfar_cols_kernel", "cacfar_cloud_kernel", "coral_kernel",
                         "surface_points_kernel", "surface_prep_kernel", "surface_sort_kernel", "surface_finish_kernel", "legacy_prepare_kernel", "legacy_cloud_kernel", "matcher_kernel", "assoc_kernel", "eval_kernel", "compensate_kernel"):
                if name in k:
                    k = name
                    break
            else:
                k = k.split("(")[0][:60]
            c = row.get("Counter_Name", "?")
            v = float(row.get("Counter_Value", 0) or 0)
            acc[k][c][0] += v
            acc[k][c][1] += 1
    print("== %s" % os.path.basename(d))
    for k in sorted(acc):
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print("%-62s %-22s avg/dispatch %.6g  (n=%d)" % (k, c, s / max(n, 1), n))

# ---- HBM traffic of the polar sweep per scan (MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE counts
# half the bytes of a wide coalesced stream -> x2; FETCH_SIZE/WRITE_SIZE are in KiB; separate passes)
import json
import re

def _avg(d, kernel, counter):
    f = glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True)
    tot, n = 0.0, 0
    for fn in f:
        for row in csv.DictReader(open(fn)):
            if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                tot += float(row["Counter_Value"]); n += 1
    return tot / n if n else None

fetch, write = _avg("pmc_fetch", "kstrongest_rows_kernel", "FETCH_SIZE"), _avg("pmc_write", "kstrongest_rows_kernel", "WRITE_SIZE")
if fetch and write:
    images = int(os.environ.get("CFEAR_PMC_IMAGES", "512"))      # sweeps per kstrongest_rows launch of the profiled command
    out = {"kernel": "kstrongest_rows", "images_per_launch": images,
           "configuration": "the headline pipeline (bench.py: fused key output consumed by surface_prep), kernel-filtered",
           "fetch_bytes_per_scan": fetch * 1024.0 * 2.0 / images, "write_bytes_per_scan": write * 1024.0 / images,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB, separate passes); FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B request on wide coalesced reads)"}
    cf = _avg("pmc_cacfar_fetch", "cacfar_rows_kernel", "FETCH_SIZE")
    if cf:
        out["cacfar_rows_fetch_bytes_per_scan"] = cf * 1024.0 * 2.0 / 512
    cc = _avg("pmc_cacfar_cols_fetch", "cacfar_cols_kernel", "FETCH_SIZE")
    if cc:
        out["cacfar_cols_fetch_bytes_per_scan"] = cc * 1024.0 * 2.0 / 512      # tools/cfar_events.py 512 --bins-major
    df = _avg("pmc_decode_fetch", "kstrong_image_kernel", "FETCH_SIZE")
    if df:
        out["kstrong_image_fetch_bytes_per_scan"] = df * 1024.0 * 2.0 / 512     # tools/decode_bench.py 512: [bins][azimuths] sweeps
    print("== traffic", json.dumps(out))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
