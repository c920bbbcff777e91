"""CPU: the oracle's radar Scan Context (RadarScancontext.cpp:59-131, Scancontext.cpp:60-268) against a NumPy
restatement written from the paper-level definition: polar binning with ceil indices, np.roll column
shifts, cosine distance over non-empty columns."""
import numpy as np
import pytest


def _cloud(seed, n=3000):
    rng = np.random.default_rng(seed)
    c = np.zeros((n, 4), np.float32)
    r = rng.uniform(1, 100, n)
    a = rng.uniform(0, 2 * np.pi, n)
    c[:, 0], c[:, 1] = r * np.cos(a), r * np.sin(a)
    c[:, 3] = rng.integers(60, 256, n)
    c[0, :2] = (0, 5); c[1, :2] = (-7, 0); c[2, :2] = (0, -3); c[3, :2] = (80, 0); c[4, :2] = (0, 0)   # axes, rim, origin
    return c


def _numpy_desc(c, R=40, S=120, rmax=80.0, fn="sum", div=1000.0, no_point=0.0, dy=0.0):
    x = c[:, 0].astype(np.float32)
    y = (c[:, 1].astype(np.float64) + dy).astype(np.float32) if dy else c[:, 1].astype(np.float32)
    rng_ = np.sqrt(x * x + y * y).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        th = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360.0
    th = th.astype(np.float32)
    keep = ~(rng_.astype(np.float64) > rmax)
    ring = np.clip(np.ceil(rng_.astype(np.float64) / rmax * R).astype(int), 1, R) - 1
    sec = np.clip(np.ceil(th.astype(np.float64) / 360.0 * S).astype(int), 1, S) - 1
    desc = np.full((R, S), -1000.0)
    for k in np.nonzero(keep)[0]:
        v = float(c[k, 3])
        if desc[ring[k], sec[k]] == -1000.0:
            desc[ring[k], sec[k]] = v
        elif fn == "sum":
            desc[ring[k], sec[k]] += v
        else:
            desc[ring[k], sec[k]] = max(desc[ring[k], sec[k]], v)
    desc = desc / div
    desc[desc == -1000.0] = no_point
    return desc, ring, sec, keep


def _numpy_dist(a, b, ratio=0.1):
    S = a.shape[1]
    v1, v2 = a.mean(0), b.mean(0)
    norms = [np.linalg.norm(v1 - np.roll(v2, sh)) for sh in range(S)]
    sh0 = int(np.argmin(norms))
    rad = int(round(0.5 * ratio * S))
    space = sorted({sh0} | {(sh0 + i) % S for i in range(1, rad + 1)} | {(sh0 - i) % S for i in range(1, rad + 1)})
    best, arg = 1e7, 0
    for sh in space:
        bs = np.roll(b, sh, axis=1)
        n1, n2 = np.linalg.norm(a, axis=0), np.linalg.norm(bs, axis=0)
        ok = (n1 != 0) & (n2 != 0)
        sim = ((a * bs).sum(0)[ok] / (n1[ok] * n2[ok])).sum() / max(int(ok.sum()), 1)
        if 1 - sim < best:
            best, arg = 1 - sim, sh
    return best, arg


@pytest.mark.parametrize("fn,div,dy", [("sum", 1000.0, 0.0), ("max", 1.0, 0.0), ("sum", 1000.0, -4.0), ("sum", 1.0, 2.0)])
def test_descriptor_matches_numpy(fn, div, dy):
    from oracle import pyoracle as O
    c = _cloud(1)
    d = O.sc_descriptor(c, desc_function=fn, desc_divider=div, no_point=0.0, shift_y=dy)
    e, ring, sec, keep = _numpy_desc(c, fn=fn, div=div, dy=dy)
    # float atan (reference) vs double atan2 may disagree on a point that sits within an ulp of a sector edge
    bad = np.abs(d - e) > 1e-12
    assert bad.sum() <= 4
    if div == 1000.0:
        assert (d[e == -1.0] == -1.0).all()        # empty bins keep NO_POINT / divider ("division before the check")
    else:
        assert (d[e == 0.0] == 0.0).all()
    rk, sk = O.sc_keys(d)
    np.testing.assert_allclose(rk, d.mean(1), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(sk, d.mean(0), rtol=1e-12, atol=1e-15)


def test_axis_points_and_rim():
    from oracle import pyoracle as O
    c = _cloud(2)[:5].copy()
    c[:, 3] = [10, 20, 30, 40, 50]
    d = O.sc_descriptor(c, desc_function="sum", desc_divider=1.0, no_point=0.0)
    assert d[2, 29] == 20 + 0 or d[2, 29] == 10        # (0, 5): range 5 -> ring 3, angle 90 deg -> sector 30
    assert d[3, 59] == 20                              # (-7, 0): 180 deg -> sector 60
    assert d[1, 89] == 30                              # (0, -3): 270 deg -> sector 90
    assert d[39, 0] == 40                              # (80, 0): on the rim, angle 0 -> sector 1 (max(.,1))
    assert d.sum() == 10 + 20 + 30 + 40 + 50           # the origin lands in ring 1, sector 1


@pytest.mark.parametrize("shift", [0, 7, 61, 119])
def test_distance_recovers_rotation(shift):
    from oracle import pyoracle as O
    a = O.sc_descriptor(_cloud(3), desc_divider=1000.0)
    b = np.roll(a, -shift, axis=1)                      # the same place seen with a heading change
    d, sh = O.sc_distance(a, b)
    e, esh = _numpy_dist(a, b)
    assert sh == esh == shift
    np.testing.assert_allclose(d, e, rtol=1e-9, atol=1e-12)
    assert d < 1e-9
    other = O.sc_descriptor(_cloud(4), desc_divider=1000.0)
    d2, sh2 = O.sc_distance(a, other)
    e2, esh2 = _numpy_dist(a, other)
    assert sh2 == esh2
    np.testing.assert_allclose(d2, e2, rtol=1e-9)
    assert d2 > 0.2
