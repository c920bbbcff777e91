"""GPU: batched GetCost and covariance by cost sampling (odometrykeyframefuser.cpp:261-380,
loopclosure.cpp:99-208) through the C-ABI vs the CPU oracle.

Sample costs are sums of ~10^3 robustified residuals in a different order than the CPU's: rtol 1e-10.
The covariance is 2 H^-1 of a least-squares fit whose design matrix has condition ~1e6 (yaw^2 column):
rtol 1e-4 on the 3x3 block (the fit amplifies the 1e-12 sample differences), identical success flags.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.test_gpu_register import _cells, _rel, _oracle_par   # noqa: E402


def _setup(seed, frames, cost, loss, opt, outer=8, inner=20):
    from tbv_slam_public_amd import api
    cells, gt = _cells(seed, frames, k=12)
    poses = np.array([_rel(gt[0], gt[f]) for f in frames])
    poses[-1] += [0.25, -0.15, 0.006]
    reg = api.n_scan_normal_reg(cost, loss, 0.1, opt)
    reg.SetParameters(outer, inner)
    scans = [api.MapPointNormal(cells=c) for c in cells]
    return reg, scans, cells, poses


@pytest.mark.parametrize("cost,loss,opt", [("P2P", "Huber", 4), ("P2L", "Huber", 0), ("P2D", "Cauchy", 2)])
def test_get_cost_batch_matches_oracle(cost, loss, opt):
    from oracle import pyoracle as O
    reg, scans, cells, poses = _setup(3, [0, 1, 2, 3], cost, loss, opt)
    rng = np.random.default_rng(1)
    jobs = []
    for _ in range(9):
        p = poses.copy()
        p[-1] += rng.normal(0, [0.3, 0.3, 0.01])
        jobs.append((scans, p))
    jobs.append((scans[:2], poses[:2].copy() + np.array([[0, 0, 0], [40.0, 40.0, 1.0]])))   # nothing associates
    for itr in (1, 4):
        reg.par.itr = itr
        out = reg.GetCostBatch(jobs)
        for (sc, p), r in zip(jobs, out):
            ok, c, res, score = O.get_cost(cells[:len(sc)], p, _oracle_par(reg))
            assert (r["status"] == 0) == ok
            if ok:
                assert r["num_residuals"] == len(res)
                np.testing.assert_allclose(r["final_cost"], c, rtol=1e-10)
                np.testing.assert_allclose(r["score"], score, rtol=1e-10)
                np.testing.assert_array_equal(r["pose"], p[-1])
            else:
                assert r["num_residuals"] <= 1
        # the single-call path agrees with the batch
        ok1, c1, _ = reg.GetCost(*jobs[0])
        np.testing.assert_allclose(c1, out[0]["final_cost"], rtol=1e-12)


@pytest.mark.parametrize("cost,loss,opt,frames,n", [("P2P", "Huber", 4, [0, 1, 2, 3, 4], 3),
                                                    ("P2L", "Huber", 0, [0, 2], 3),
                                                    ("P2L", "Huber", 0, [0, 1, 2], 2),
                                                    ("P2D", "Huber", 0, [0, 1, 2], 4)])
def test_covariance_by_sampling_matches_oracle(cost, loss, opt, frames, n):
    from oracle import pyoracle as O
    reg, scans, cells, poses = _setup(4, frames, cost, loss, opt)
    ok, pg, _ = reg.Register(scans, poses)
    assert ok
    res = reg.summary_
    sp = reg.sampling_params(samples_per_axis=n)
    got_ok, cov, smp = reg.approximateCovarianceBySampling(scans, pg, sampling=sp, want_samples=True)
    # oracle: same poses / summary in, so only the sampling + fit are compared
    opar = _oracle_par(reg)
    assert opar.first_itr == res.outer_iters
    exp_ok, exp_cov, exp_smp = O.cov_by_sampling(cells, pg, opar, res.final_cost, res.num_residuals,
                                                 sp.xy_range, sp.yaw_range, n, sp.covariance_scaler)
    np.testing.assert_array_equal(smp[:, :3], exp_smp[:, :3])              # the sample grid itself is exact
    np.testing.assert_allclose(smp[:, 3], exp_smp[:, 3], rtol=1e-10)
    assert got_ok == exp_ok
    if exp_ok:
        idx = np.ix_([0, 1, 5], [0, 1, 5])
        np.testing.assert_allclose(cov[idx], exp_cov[idx], rtol=1e-4, atol=1e-14)
        rest = cov.copy()
        rest[idx] = 0
        np.testing.assert_array_equal(rest, np.diag([0, 0, 1.0, 1.0, 1.0, 0]))   # Identity elsewhere (:368)
    else:
        np.testing.assert_array_equal(cov, np.diag([0.01, 0.01, 0, 0, 0, 1e-4]))


def test_covariance_batch_loop_closure_style():
    """loopclosure::Register + approximateCovarianceBySampling over a candidate batch: P2L, Huber 0.1,
    SetParameters(4, 10), xy +-0.2, yaw +-0.0022, scaler 4 (loopclosure.cpp:56-57, 108-112)."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    cells, gt = _cells(7, [0, 1, 2, 3, 4, 5], k=12)
    scans = [api.MapPointNormal(cells=c) for c in cells]
    reg = api.n_scan_normal_reg("P2L", "Huber", 0.1, 0)
    reg.SetParameters(4, 10)
    rng = np.random.default_rng(3)
    pairs = [(0, 1), (1, 2), (0, 2), (3, 4), (2, 5), (4, 5), (1, 3)]
    jobs = []
    for a, b in pairs:
        p = np.array([[0, 0, 0], _rel(gt[a], gt[b])]) + np.array([[0, 0, 0], rng.normal(0, [0.3, 0.3, 0.01])])
        jobs.append(([scans[a], scans[b]], p))
    out = reg.RegisterBatch(jobs)
    post = [(sc, np.vstack([p[:-1], r["pose"]])) for (sc, p), r in zip(jobs, out)]
    sp = reg.sampling_params(xy_range=0.4, yaw_range=0.0044, samples_per_axis=3, covariance_scaler=4.0)
    oks, covs = reg.approximateCovarianceBySamplingBatch(post, out, sp)
    n_valid = 0
    for (a, b), (sc, p), r, ok, cov in zip(pairs, post, out, oks, covs):
        opar = _oracle_par(reg)
        opar.first_itr = int(r["outer_iters"])
        exp_ok, exp_cov, _ = O.cov_by_sampling([cells[a], cells[b]], p, opar, float(r["final_cost"]),
                                               int(r["num_residuals"]), 0.4, 0.0044, 3, 4.0)
        assert bool(ok) == exp_ok
        if exp_ok:
            n_valid += 1
            idx = np.ix_([0, 1, 5], [0, 1, 5])
            np.testing.assert_allclose(cov[idx], exp_cov[idx], rtol=1e-4, atol=1e-14)
        # single-job entry point == batch entry point
        ok1, cov1 = reg.approximateCovarianceBySampling(sc, p, reg_result=r, sampling=sp)
        assert ok1 == bool(ok)
        np.testing.assert_array_equal(cov1, cov)
    assert n_valid >= 3


def test_odometry_with_cov_sampling_matches_oracle_fuser():
    """processFrame with estimate_cov_by_sampling (odometrykeyframefuser.cpp:203-208): poses unchanged,
    cov_current per frame equal to the CPU fuser's (Identity on the first frame, Register's diagonal when
    the fit is rejected, the sampled covariance otherwise)."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, synth
    seeds, n_frames = [0, 5], 7
    seqs = [synth.scene_v1(sd, n_frames)[0] for sd in seeds]
    par = api.odometry_params(estimate_cov_by_sampling=1, kstrong_k_strongest=12)
    od = api.OdometryKeyframeFuser(len(seeds), 400, 3360, par)
    fus = []
    for _ in seeds:
        reg = O.reg_params(cost=par.reg.cost, loss=par.reg.loss, loss_limit=0.1, weight_opt=par.reg.weight_opt,
                           regularization=0.0)
        fus.append(O.Fuser(reg, res=par.res, submap_scan_size=par.submap_scan_size, weight_intensity=True,
                           estimate_cov_by_sampling=True))
    n_sampled = 0
    for f in range(n_frames):
        info = od.process(np.stack([seq[f] for seq in seqs]))
        cov, flag = od.covariance()
        for b, seq in enumerate(seqs):
            sr, si, sc = O.kstrongest(seq[f], 12, 60)
            pose_o, info_o = fus[b].process(O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5))
            cov_o, flag_o = fus[b].last_cov()
            d = np.abs(info["pose"][b] - pose_o)
            assert d[:2].max() <= 1e-4 and d[2] <= 1e-5
            assert bool(flag[b]) == flag_o, (f, b)
            if f == 0:
                np.testing.assert_array_equal(cov[b], np.eye(6))
            if flag_o:
                n_sampled += 1
                idx = np.ix_([0, 1, 5], [0, 1, 5])
                np.testing.assert_allclose(cov[b][idx], cov_o[idx], rtol=1e-4, atol=1e-14)
            else:
                np.testing.assert_array_equal(cov[b], cov_o)
    assert n_sampled >= 4
