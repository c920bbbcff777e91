#!/usr/bin/env python3
"""Writes the INPUTS of tests/golden/{filters,registration}.npz as raw arrays + manifest.txt for tools/ref_golden/ref_golden.cpp
(usage: export_inputs.py OUT_DIR), plus the tie-rule queries (tie_q: scan 1's cell means of registration.npz moved by the start
pose) and a 50-frame synthetic sweep sequence (seq_img: synth.scene_v1(4242, 50), 67 MB raw) for the odometry trace.  Only numpy
is needed (tbv_slam_public_amd/synth.py is numpy-only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "..", "tests", "golden")


SEQ_SEED, SEQ_FRAMES = 4242, 50


def tie_queries(reg):
    """The source scan's cell means (registration.npz: the oracle's cells of cloud2, the free scan) seen from scan 0:
    T0^-1 * T2 * mean -- what Register's first association pass asks scan 0's tree (float64 [m][2]; the reference casts to
    float itself).  tests/test_golden.py::tie_queries_of builds the same array."""
    m = np.asarray(reg["cells2"]["mean"], np.float64)
    p0, p2 = reg["poses"][0], reg["poses"][2]
    c, s = np.cos(p2[2] - p0[2]), np.sin(p2[2] - p0[2])
    c0, s0 = np.cos(p0[2]), np.sin(p0[2])
    d = p2[:2] - p0[:2]
    t = np.array([c0 * d[0] + s0 * d[1], -s0 * d[0] + c0 * d[1]])
    return np.ascontiguousarray(m @ np.array([[c, -s], [s, c]]).T + t)


def main(out):
    os.makedirs(out, exist_ok=True)
    filt, reg = np.load(os.path.join(GOLD, "filters.npz")), np.load(os.path.join(GOLD, "registration.npz"))
    items = {"img": filt["img"].astype(np.uint8), "poses": reg["poses"].astype(np.float64), "mot": reg["mot"].astype(np.float64)}
    for i in range(3):
        items["cloud%d" % i] = reg["cloud%d" % i].astype(np.float32)          # [n][4] x, y, z, intensity
    items["tie_q"] = tie_queries(reg)
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from tbv_slam_public_amd import synth                                      # numpy only
    items["seq_img"] = synth.scene_v1(SEQ_SEED, SEQ_FRAMES)[0].astype(np.uint8)
    with open(os.path.join(out, "manifest.txt"), "w") as m:
        for name, a in items.items():
            a = np.ascontiguousarray(a)
            a.tofile(os.path.join(out, name + ".bin"))
            m.write("%s %s %s\n" % (name, a.dtype.name, " ".join(str(d) for d in a.shape)))
    print("wrote", sorted(items), "to", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ref_in")
