"""CPU: the host logic of examples/loop_closure_demo.py (graph nodes, local maps, Scan Context retrieval policy,
candidate guesses, odometry bounds, verification, per-query selection) with the CPU oracle doing the arithmetic.
One synthetic lap: loops may only be accepted where the sensor really is back, and their registered transforms
must sit on the ground truth."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

from tests.loop_backends import OracleBackend   # noqa: E402


def test_one_lap_closes_the_loop_with_the_oracle_backend():
    import loop_closure_demo as demo
    out = demo.run(OracleBackend(), 68)
    gt = out["gt"]
    acc = [(c, r) for c, r in zip(out["candidates"], out["results"]) if r["accepted"]]
    assert len(out["candidates"]) > 100 and len(acc) >= 12
    for c, r in acc:
        assert c["from"] >= 54 and c["to"] <= 14
        true = demo.xyt_compose(demo.xyt_inverse(gt[c["from"]]), gt[c["to"]])
        e = r["t_be"] - true
        e[2] = (e[2] + np.pi) % (2 * np.pi) - np.pi
        assert np.hypot(e[0], e[1]) < 0.15 and abs(e[2]) < 0.01, (c["from"], c["to"], e)
    # candidates proposed half-way round the lap are retrieved (the database is never empty) but all rejected
    mid = [r for c, r in zip(out["candidates"], out["results"]) if 20 <= c["from"] <= 45]
    assert len(mid) > 30 and not any(r["accepted"] for r in mid)
    # the first nodes' clouds were never motion-compensated (Tmot is unknown for them, odometrykeyframefuser.cpp:146-150):
    # revisits of nodes 0..2 register but fail the alignment check -- as they would in the reference
    early = [r for c, r in zip(out["candidates"], out["results"]) if c["from"] >= 56 and c["to"] <= 2]
    assert early and not any(r["accepted"] for r in early)
