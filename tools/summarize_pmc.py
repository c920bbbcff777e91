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
            for name in ("kstrongest_rows_kernel", "kstrong_cloud_kernel", "cacfar_rows_kernel", "cacfar_cloud_kernel",
                         "surface_points_kernel", "register_kernel", "assoc_kernel", "eval_kernel", "compensate_kernel"):
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
