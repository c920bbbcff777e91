#!/usr/bin/env python3
"""Writes the INPUTS of tests/golden/{filters,registration}.npz as raw arrays + manifest.txt for tools/ref_golden/ref_golden.cpp
(usage: export_inputs.py OUT_DIR).  Only numpy is needed."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "..", "tests", "golden")


def main(out):
    os.makedirs(out, exist_ok=True)
    filt, reg = np.load(os.path.join(GOLD, "filters.npz")), np.load(os.path.join(GOLD, "registration.npz"))
    items = {"img": filt["img"].astype(np.uint8), "poses": reg["poses"].astype(np.float64), "mot": reg["mot"].astype(np.float64)}
    for i in range(3):
        items["cloud%d" % i] = reg["cloud%d" % i].astype(np.float32)          # [n][4] x, y, z, intensity
    with open(os.path.join(out, "manifest.txt"), "w") as m:
        for name, a in items.items():
            a = np.ascontiguousarray(a)
            a.tofile(os.path.join(out, name + ".bin"))
            m.write("%s %s %s\n" % (name, a.dtype.name, " ".join(str(d) for d in a.shape)))
    print("wrote", sorted(items), "to", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ref_in")
