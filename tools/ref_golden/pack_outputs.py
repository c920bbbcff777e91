#!/usr/bin/env python3
"""Packs the raw arrays ref_golden wrote (OUT_DIR/manifest.txt) into tests/golden/ref_filters.npz / ref_registration.npz and copies
ref_simple_graph.sgh next to them (usage: pack_outputs.py OUT_DIR GOLDEN_DIR)."""
import os
import shutil
import sys

import numpy as np


def main(src, dst):
    arrays = {}
    for line in open(os.path.join(src, "manifest.txt")):
        name, dtype, *dims = line.split()
        arrays[name] = np.fromfile(os.path.join(src, name + ".bin"), dtype=dtype).reshape([int(d) for d in dims])
    filt = {k[2:]: v for k, v in arrays.items() if k.startswith("f_")}
    reg = {k[2:]: v for k, v in arrays.items() if k.startswith("r_")}
    np.savez_compressed(os.path.join(dst, "ref_filters.npz"), **filt)
    np.savez_compressed(os.path.join(dst, "ref_registration.npz"), **reg)
    if os.path.exists(os.path.join(src, "ref_simple_graph.sgh")):
        shutil.copy(os.path.join(src, "ref_simple_graph.sgh"), os.path.join(dst, "ref_simple_graph.sgh"))
    print("ref_filters:", sorted(filt), "\nref_registration:", sorted(reg))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
