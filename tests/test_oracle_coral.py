"""CPU: the oracle's CorAl alignment quality (AlignmentQuality.cpp:8-230) against an independent NumPy /
SciPy restatement: neighbour sets from scipy.spatial.cKDTree on the float32 points (checked against the
float distance test), covariances by numpy.cov (ddof = 1), entropies 1/2 log(2 pi e det + 1e-8)."""
import numpy as np
import pytest


def _peaks(seed, frames, k=12):
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    imgs, gt, _ = synth.scene_v1(seed, max(frames) + 1)
    out = []
    for f in frames:
        sr, si, sc = O.kstrongest(imgs[f], k, 60)
        pk = O.peaks(imgs[f], k, sr, sc)
        out.append(O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5, mask=pk))
    return out, gt


def _tf(cloud, pose):
    c, s = np.cos(pose[2]), np.sin(pose[2])
    x, y = cloud[:, 0].astype(np.float64), cloud[:, 1].astype(np.float64)
    return np.stack([((c * x + -s * y) + 0.0) + pose[0], ((s * x + c * y) + 0.0) + pose[1]], 1).astype(np.float32)


def _compose(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([c * b[0] - s * b[1] + a[0], s * b[0] + c * b[1] + a[1], a[2] + b[2]])


def _numpy_coral(ref, src, ref_pose, src_pose, offset, radius):
    from scipy.spatial import cKDTree
    P_src = _tf(src, _compose(src_pose, offset))
    P_ref = _tf(ref, ref_pose)
    r2 = np.float32(radius * radius)

    def near(P, q):
        d = (q[0] - P[:, 0]) ** 2 + (q[1] - P[:, 1]) ** 2        # float32 arithmetic, as FLANN
        return np.nonzero(d < r2)[0]

    merged = len(P_src) + len(P_ref)
    joint, sep, valid = np.full(merged, 100.0), np.full(merged, 100.0), np.zeros(merged, bool)
    t_src, t_ref = cKDTree(P_src.astype(np.float64)), cKDTree(P_ref.astype(np.float64))
    for pass_, Q in enumerate((P_src, P_ref)):
        for k, q in enumerate(Q):
            idx = k if pass_ == 0 else len(P_src) + k
            i_s, i_r = near(P_src, q), near(P_ref, q)
            # the kd-tree (double distances) may only differ from the float test on the radius boundary
            ks = set(t_src.query_ball_point(q.astype(np.float64), radius * (1 - 1e-6)))
            assert ks <= set(i_s.tolist())
            if (len(i_r) if pass_ == 0 else len(i_s)) < 1:
                continue
            own = P_src[i_s] if pass_ == 0 else P_ref[i_r]
            both = np.vstack([P_src[i_s], P_ref[i_r]])
            if len(own) <= 2 or len(both) <= 2:
                continue
            ds = np.linalg.det(np.cov(own.astype(np.float64).T, ddof=1))
            dj = np.linalg.det(np.cov(both.astype(np.float64).T, ddof=1))
            with np.errstate(invalid="ignore", divide="ignore"):
                es = 0.5 * np.log(2 * np.pi * np.e * ds + 1e-8)
                ej = 0.5 * np.log(2 * np.pi * np.e * dj + 1e-8)
            if np.isnan(es) or np.isnan(ej):
                continue
            sep[idx], joint[idx], valid[idx] = es, ej, True
    n = valid.sum()
    q = np.array([joint[valid].sum() / max(n, 1), sep[valid].sum() / max(n, 1), n / merged])
    return n / merged >= 0.1, q, joint, sep, valid


def _rel(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    d = b[:2] - a[:2]
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], b[2] - a[2]])


@pytest.mark.parametrize("offset", [(0, 0, 0), (0.5, 0, 0.0087), (-2.0, 2.0, 0.26)])
def test_coral_quality_matches_numpy(offset):
    from oracle import pyoracle as O
    clouds, gt = _peaks(8, [0, 2])
    ref_pose, src_pose = np.zeros(3), _rel(gt[0], gt[2])
    ok, q, pp = O.coral_quality(clouds[0], clouds[1], ref_pose, src_pose, offset, 1.0)
    eok, eq, ej, es, ev = _numpy_coral(clouds[0], clouds[1], ref_pose, src_pose, np.array(offset, float), 1.0)
    np.testing.assert_array_equal(pp[:, 2].astype(bool), ev)
    # det = c00 c11 - c01^2 cancels for near-collinear neighbourhoods and log(2 pi e det + 1e-8) amplifies its
    # rounding (the reference's value depends on Eigen's summation order to the same degree): 1e-6 per point
    np.testing.assert_allclose(pp[ev, 0], ej[ev], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(pp[ev, 1], es[ev], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(q, eq, rtol=1e-8)
    assert ok == eok
    assert 0.0 < q[2] <= 1.0


def test_coral_separates_aligned_from_misaligned():
    """The joint entropy grows relative to the separate entropy when the scans are misaligned: the property
    TBV's classifier consumes (alignmentinterface.cpp:296-347)."""
    from oracle import pyoracle as O
    clouds, gt = _peaks(9, [0, 1])
    src_pose = _rel(gt[0], gt[1])
    _, qa, _ = O.coral_quality(clouds[0], clouds[1], np.zeros(3), src_pose, (0, 0, 0), 1.0)
    _, qm, _ = O.coral_quality(clouds[0], clouds[1], np.zeros(3), src_pose, (1.0, 1.0, 0.05), 1.0)
    assert (qm[0] - qm[1]) > (qa[0] - qa[1])


def test_coral_degenerate_inputs():
    from oracle import pyoracle as O
    a = np.array([[1, 1, 0, 90], [1.1, 1, 0, 80], [1, 1.1, 0, 70]], np.float32)
    b = a + np.array([50, 0, 0, 0], np.float32)              # no overlap at all
    ok, q, pp = O.coral_quality(a, b, np.zeros(3), np.zeros(3))
    assert not ok and q[2] == 0.0 and not pp[:, 2].any()
    ok, q, pp = O.coral_quality(a, a.copy(), np.zeros(3), np.zeros(3))    # identical clouds overlap fully
    assert ok and q[2] == 1.0
