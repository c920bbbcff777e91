"""CPU: the oracle's polar filters against independent brute-force NumPy restatements.

These pin oracle/cfear_oracle.cpp (a literal emulation of the reference's insert-sort / hash-map
code) to the closed-form semantics of SURVEY.md Appendix A.1-A.3.
"""
import numpy as np
import pytest

from oracle import pyoracle as O
from tbv_slam_public_amd import synth


def brute_kstrongest(img, k, z_min):
    rows, cols = img.shape
    sr = -np.ones((rows, k), np.int32)
    si = np.zeros((rows, k), np.uint8)
    sc = np.zeros(rows, np.int32)
    zmin = np.uint8(z_min)
    for r in range(rows):
        idx = np.nonzero(img[r] >= zmin)[0]
        # k largest (intensity, range) pairs under lexicographic order, stored ascending
        order = np.lexsort((idx, img[r, idx]))
        keep = idx[order][-k:]
        sc[r] = keep.size
        sr[r, :keep.size] = keep
        si[r, :keep.size] = img[r, keep]
    return sr, si, sc


@pytest.mark.parametrize("k,z_min", [(1, 60), (12, 60), (40, 60), (40, 0), (64, 200), (100, 250)])
def test_kstrongest_uniform(k, z_min):
    img = synth.uniform_v1(3, rows=37, cols=500)[0]
    got = O.kstrongest(img, k, z_min)
    exp = brute_kstrongest(img, k, z_min)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


def test_kstrongest_ties_and_empty():
    img = np.zeros((6, 300), np.uint8)
    img[1, :] = 255                       # all tied: the k largest ranges win
    img[2, 5:9] = 70                      # fewer than k candidates
    img[3, ::2] = 90
    img[3, 1::2] = 91
    img[4, 299] = 60                      # exactly z_min is kept (intensity < z_min is skipped)
    img[5, 0] = 59
    sr, si, sc = O.kstrongest(img, 12, 60)
    np.testing.assert_array_equal(sc, [0, 12, 4, 12, 1, 0])
    np.testing.assert_array_equal(sr[1], np.arange(288, 300))
    np.testing.assert_array_equal(sr[2, :4], [5, 6, 7, 8])
    np.testing.assert_array_equal(sr[3], np.arange(277, 300, 2))
    e = brute_kstrongest(img, 12, 60)
    np.testing.assert_array_equal(sr, e[0])


def test_kstrongest_scene_full_size():
    imgs, _, _ = synth.scene_v1(5, 1)
    got = O.kstrongest(imgs[0], 40, 60)
    exp = brute_kstrongest(imgs[0], 40, 60)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


def brute_peaks(img, k, sel_range, sel_count):
    """Independent statement of AxialNonMaxSupress: score(r) is the 7-tap box sum of the
    raw (contiguous) image if r lies within +-3 of a kept bin m with 3 <= m < C-3, else 0."""
    rows, cols = img.shape
    flat = np.concatenate([np.zeros(16, np.int64), img.astype(np.int64).ravel(), np.zeros(16, np.int64)])
    out = np.zeros((rows, k), np.uint8)
    for r in range(rows):
        kept = sel_range[r, :sel_count[r]]
        valid = kept[(kept >= 3) & (kept < cols - 3)]
        covered = set()
        for m in valid:
            covered.update(range(m - 3, m + 4))

        def score(c):
            if c not in covered:
                return 0
            base = 16 + r * cols + c
            return int(flat[base - 3:base + 4].sum()) & 0xFFFF
        for j, m in enumerate(kept):
            p = score(m)
            ok = all(not (score(m - i) > p or p < score(m + i)) for i in (1, 2, 3))
            out[r, j] = ok
    return out


def test_peaks_vs_brute():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(23, 400), dtype=np.uint8)
    img[:, 395:] = 250          # kept bins at the far edge (outside [3, C-3))
    img[5, :4] = 255            # and at the near edge
    sr, si, sc = O.kstrongest(img, 20, 100)
    got = O.peaks(img, 20, sr, sc)
    exp = brute_peaks(img, 20, sr, sc)
    np.testing.assert_array_equal(got, exp)
    imgs, _, _ = synth.scene_v1(2, 1)
    sr, si, sc = O.kstrongest(imgs[0], 40, 60)
    got = O.peaks(imgs[0], 40, sr, sc)
    exp = brute_peaks(imgs[0], 40, sr, sc)
    np.testing.assert_array_equal(got, exp)
    assert 0 < got.sum() < sc.sum()


def test_cloud_geometry():
    imgs, _, _ = synth.scene_v1(2, 1)
    sr, si, sc = O.kstrongest(imgs[0], 12, 60)
    cloud = O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5)
    rr = np.float32(0.0438).astype(np.float64)
    min_bin = int(np.ceil(np.float64(np.float32(2.5)) / rr))
    assert min_bin == 58
    pts = []
    for r in range(400):
        th = (float(r + 1) / 400) * 2.0 * np.pi
        for j in range(sc[r]):
            b = int(sr[r, j])
            if b > min_bin:
                rho = rr / 2.0 + rr * b
                pts.append((np.float32(rho * np.cos(th)), np.float32(rho * np.sin(th)), 0.0, float(si[r, j])))
    np.testing.assert_array_equal(cloud, np.array(pts, np.float32))


def brute_cacfar(img, window, guard, pfa, range_res, z_min, min_distance, max_distance=400.0):
    rows, cols = img.shape
    rr = np.float64(np.float32(range_res))
    pfa = np.float64(np.float32(pfa))
    N = float(2 * window)
    scaling = N * (pfa ** (-1.0 / N) - 1.0)
    sq = img.astype(np.float64) ** 2
    det = []
    for r in range(rows):
        for b in range(cols):
            rng_ = rr * b
            I = float(img[r, b])
            if not (rng_ > np.float64(np.float32(min_distance)) and rng_ < max_distance and I > np.float64(np.float32(z_min))):
                continue
            t0, t1 = max(0, b - guard - window), b - guard
            f0, f1 = b + guard, min(cols, b + guard + window)
            if t1 <= t0 or f1 <= f0:
                continue      # 0/0 = NaN -> comparison false
            mean = (sq[r, t0:t1].sum() / (t1 - t0) + sq[r, f0:f1].sum() / (f1 - f0)) / 2.0
            if I * I > scaling * mean:
                det.append((r, b))
    return np.array(det, np.int32).reshape(-1, 2)


def test_cacfar_vs_brute():
    rng = np.random.default_rng(4)
    img = (10 + rng.exponential(12, size=(16, 700))).clip(0, 255).astype(np.uint8)
    img[:, 300:303] = 200
    img[3, 690:] = 255          # leading window runs off the row end
    cloud, rc = O.cacfar(img, 40, 10, 0.01, 0.175, 20, 2.5)
    exp = brute_cacfar(img, 40, 10, 0.01, 0.175, 20, 2.5)
    np.testing.assert_array_equal(rc, exp)
    assert rc.shape[0] > 10
    rr = np.float64(np.float32(0.175))
    th = (rc[:, 0].astype(np.float64) + 1) / 16 * 2.0 * np.pi
    np.testing.assert_array_equal(cloud[:, 0], (rr * rc[:, 1] * np.cos(th)).astype(np.float32))
    np.testing.assert_array_equal(cloud[:, 3], img[rc[:, 0], rc[:, 1]].astype(np.float32))


def test_legacy_filter_closed_form():
    """The legacy k_strongest_filter (radar_filters.cpp:25-78) restated literally in the oracle equals its closed form: with f
    the first bin >= z_min (intensity m0) and D the later bins with intensity > m0, the row keeps the k largest of D under
    (intensity, -range) -- ties at the cut keep the smaller ranges -- plus f iff |D| < k."""
    rng = np.random.default_rng(3)
    for trial in range(40):
        cols = int(rng.integers(30, 400))
        k = int(rng.integers(1, 16))
        row = rng.integers(0, 256, cols).astype(np.uint8) if trial % 2 else rng.integers(50, 80, cols).astype(np.uint8)
        z_min = float(rng.choice([0.0, 60.0, 60.5, 200.0]))
        got = O.kstrongest_legacy(row[None], k, z_min, 0.0438, 0.0)          # min_distance 0: only bin 0 is dropped
        cand = np.nonzero(row >= z_min)[0]
        exp = []
        if cand.size:
            f = cand[0]
            D = [i for i in cand[1:] if row[i] > row[f]]
            D.sort(key=lambda i: (-int(row[i]), i))
            exp = D[:k] + ([f] if len(D) < k else [])
        exp = [i for i in exp if i > 0]
        theta = np.float32(np.float32(1.0) / np.float32(1)) * 2 * np.pi      # rows = 1: theta = float(2 pi)
        theta = np.float32(theta)
        ex = np.array([[np.float32(0.0438 * i * np.cos(theta)), np.float32(0.0438 * i * np.sin(theta)), 0.0, row[i]] for i in exp],
                      np.float32).reshape(-1, 4)
        assert got.shape == ex.shape, (trial, got.shape, ex.shape)
        np.testing.assert_array_equal(got[:, 3], ex[:, 3])
        np.testing.assert_allclose(got[:, :2], ex[:, :2], rtol=1e-6, atol=1e-6)
