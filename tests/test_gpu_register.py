"""GPU: stage M (association + robust LM scan matcher) through the C-ABI vs the CPU oracle.

Tolerance (BASELINE.json north_star): registered SE(2) pose within 1e-4 m / 1e-5 rad of the CPU
path.  Integer outcomes (association pairs, iteration counts, residual counts) must be identical.
"""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POS_TOL, ROT_TOL = 1e-4, 1e-5


def _cells(seed, frames, k=40, radius=3.0, wi=True):
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    imgs, gt, _ = synth.scene_v1(seed, max(frames) + 1)
    out = []
    for f in frames:
        sr, si, sc = O.kstrongest(imgs[f], k, 60)
        cloud = O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5)
        out.append(O.surface_points(cloud, radius, 1.0, (0, 0), wi))
    return out, gt


def _rel(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    d = b[:2] - a[:2]
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], b[2] - a[2]])


def _oracle_par(reg):
    from oracle import pyoracle as O
    p = reg.par
    return O.reg_params(cost=p.cost, loss=p.loss, loss_limit=p.loss_limit, weight_opt=p.weight_opt,
                        max_outer=p.max_itr_association, max_inner=p.max_itr_solver, min_outer=p.min_itr,
                        radius=p.radius, cov_scale=p.cov_scale, regularization=p.regularization, first_itr=p.itr)


def _compare_register(reg, cells, poses):
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    scans = [api.MapPointNormal(cells=c) for c in cells]
    ok_g, pg, _ = reg.Register(scans, poses)
    res = reg.summary_
    ok_o, po, ro = O.register(cells, poses, _oracle_par(reg))
    assert ok_g == ok_o
    assert res.outer_iters == ro.outer_iters, (res.outer_iters, ro.outer_iters)
    assert res.num_residuals == ro.num_residuals
    assert res.lm_iters == ro.lm_iters
    assert np.abs(pg[-1, :2] - po[-1, :2]).max() <= POS_TOL
    assert abs(pg[-1, 2] - po[-1, 2]) <= ROT_TOL
    np.testing.assert_allclose(res.final_cost, ro.final_cost, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.score, ro.score, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(pg[:-1], np.asarray(poses)[:-1])      # fixed scans untouched
    return pg, po


@pytest.mark.parametrize("cost,loss,opt", [("P2L", "Huber", 0), ("P2P", "Huber", 4), ("P2D", "Huber", 0),
                                           ("P2L", "Cauchy", 4), ("P2P", "None", 1), ("P2L", "Tukey", 2),
                                           ("P2P", "SoftLOne", 3), ("P2L", "Combined", 0)])
def test_pair_registration_matches_oracle(cost, loss, opt):
    from tbv_slam_public_amd import api
    cells, gt = _cells(1, [0, 2])
    reg = api.n_scan_normal_reg(cost, loss, 0.1, opt)
    truth = _rel(gt[0], gt[2])
    for dx, dy, dth in [(0.0, 0.0, 0.0), (0.8, -0.5, 0.02), (-1.2, 0.7, -0.04)]:
        poses = np.array([[0, 0, 0], truth + [dx, dy, dth]])
        pg, po = _compare_register(reg, cells, poses)
    assert np.abs(po[-1] - truth)[:2].max() < 0.5      # it actually registers


def test_loop_closure_configuration_4x10():
    """loopclosure::Register (tbv_slam/src/tbv_slam/loopclosure.cpp:56-59): P2L, SetParameters(4,10)."""
    from tbv_slam_public_amd import api
    cells, gt = _cells(2, [0, 3])
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    rng = np.random.default_rng(0)
    truth = _rel(gt[0], gt[3])
    for _ in range(4):
        guess = truth + np.concatenate([rng.normal(0, 1.0, 2), rng.normal(0, np.deg2rad(3), 1)])
        _compare_register(reg, cells, np.array([[0, 0, 0], guess]))


def test_many_to_one_window():
    """5-scan problem: 4 fixed keyframes + the free current scan (CFEAR-3 odometry preset)."""
    from tbv_slam_public_amd import api
    frames = [0, 1, 2, 3, 4]
    cells, gt = _cells(3, frames)
    poses = np.array([_rel(gt[0], gt[f]) for f in frames])
    poses[-1] += [0.4, -0.2, 0.01]
    reg = api.n_scan_normal_reg("P2P", "Huber", 0.1, 4)
    _compare_register(reg, cells, poses)
    reg2 = api.n_scan_normal_reg("P2L", "Huber", 0.1, 4)
    _compare_register(reg2, cells[1:], poses[1:])


def _dense_cells(seed, frames):
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    imgs, gt, _ = synth.scene_dense(seed, max(frames) + 1)
    out = []
    for f in frames:
        sr, si, sc = O.kstrongest(imgs[f], 40, 60)
        out.append(O.surface_points(O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5), 3.0, 1.0, (0, 0), True))
    return out, gt


def test_large_registrations_on_every_form():
    """Registrations the regular form (4 wavefronts, 40 KB) is not good at are deferred to the forms behind it: half a CU (8
    wavefronts, 80 KB: keyframe tables staged group by group; with ~2 300 cells per scan the match table moves to global
    scratch) and a whole CU (16 wavefronts).  1 900 / 2 300 / ~2 800 cells per scan (cells of two dense scenes side by side,
    600 m apart).  Alone, such a job runs on the small-batch form at once (8 wavefronts, the CU's LDS); with the regular form
    forced (context options) the large forms must be the launches that do the work.  Every route gives the oracle's result."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, synth
    from tbv_slam_public_amd import _lib as L
    frames = [0, 1, 2, 3, 4]
    worlds = [synth.scene_dense(seed, 5) for seed in (9, 10, 11)]
    gt = worlds[0][1]
    poses = np.array([_rel(gt[0], gt[f]) for f in frames])
    poses[-1] += [0.25, -0.15, 0.006]
    reg = api.n_scan_normal_reg("P2P", "Huber", 0.1, 4)
    try:
        for radius, keep, lo, hi, use in ((2.5, 950, 1700, 2000, (0, 1)), (3.0, 100000, 2100, 8000, (0, 1)), (2.5, 1150, 2299, 2301, (1, 2))):
            cells = []
            for f in frames:
                parts = []
                for w, (imgs, _, _) in enumerate([worlds[u] for u in use]):
                    sr, si, sc = O.kstrongest(imgs[f], 40, 60)
                    c = O.surface_points(O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5), radius, 1.0, (0, 0), True).copy()
                    c["mean"][:, 0] += 600.0 * w
                    parts.append(c[:keep])
                cells.append(np.concatenate(parts))
            assert lo < min(len(c) for c in cells) and max(len(c) for c in cells) < hi, [len(c) for c in cells]
            maps = [api.MapPointNormal(cells=c) for c in cells]
            ok_o, po, ro = O.register(cells, poses, _oracle_par(reg))
            for waves, kb in ((0, 0), (4, 40)):
                reg.ctx.set_option(L.OPT_MATCHER_WAVES, waves); reg.ctx.set_option(L.OPT_MATCHER_LDS_KB, kb)
                reg.ctx.profile_enable(True); reg.ctx.profile_read(reset=True)
                out = reg.RegisterBatch([(maps, poses)])[0]
                prof = reg.ctx.profile_read(reset=True); reg.ctx.profile_enable(False)
                if waves:                                                     # the large forms did the work
                    large = prof["register_large"][0] + prof["register_large16"][0]
                    assert large > 5 * prof["register"][0], prof
                else:
                    assert not any(v[1] for k, v in prof.items() if k.startswith("register_large")), prof
                assert (out["status"] == 0) == ok_o
                assert (out["outer_iters"], out["lm_iters"], out["num_residuals"]) == (ro.outer_iters, ro.lm_iters, ro.num_residuals)
                assert np.abs(out["pose"][:2] - po[-1, :2]).max() <= POS_TOL and abs(out["pose"][2] - po[-1, 2]) <= ROT_TOL
                assert out["reserved"] == 1.0
    finally:
        reg.ctx.set_option(L.OPT_MATCHER_WAVES, 0); reg.ctx.set_option(L.OPT_MATCHER_LDS_KB, 0)


def test_large_scans_take_the_second_launch():
    """Scans of ~1 400 cells: four keyframes do not fit the 80 KB association, so the batch entry adds the second launch
    (one workgroup per CU with all of its LDS); a five-scan window still fits it, and the result is the oracle's either
    way.  Mixed with an ordinary job in the same batch: that one must come out exactly as it does alone."""
    from tbv_slam_public_amd import api
    frames = [0, 1, 2, 3, 4]
    cells, gt = _dense_cells(7, frames)
    assert min(len(c) for c in cells) > 1000
    poses = np.array([_rel(gt[0], gt[f]) for f in frames])
    poses[-1] += [0.3, -0.2, 0.008]
    reg = api.n_scan_normal_reg("P2P", "Huber", 0.1, 4)
    _compare_register(reg, cells, poses)                              # 4 x ~1400 targets: second launch
    _compare_register(reg, cells[2:], poses[2:])                      # 2 x ~1400: still the ordinary geometry
    small, gts = _cells(3, frames)
    sp = np.array([_rel(gts[0], gts[f]) for f in frames])
    sp[-1] += [0.4, -0.2, 0.01]
    alone = reg.Register([api.MapPointNormal(cells=c) for c in small], sp)[1]
    jobs = [([api.MapPointNormal(cells=c) for c in cells], poses), ([api.MapPointNormal(cells=c) for c in small], sp)]
    both = reg.RegisterBatch(jobs)
    np.testing.assert_array_equal(both[1]["pose"], alone[-1])
    assert both[0]["reserved"] == 1.0 and both[1]["reserved"] == 0.0     # which geometry served the registration
    from oracle import pyoracle as O
    _, po, ro = O.register(cells, poses, _oracle_par(reg))
    assert np.abs(both[0]["pose"][:2] - po[-1, :2]).max() <= POS_TOL and abs(both[0]["pose"][2] - po[-1, 2]) <= ROT_TOL
    assert both[0]["outer_iters"] == ro.outer_iters and both[0]["num_residuals"] == ro.num_residuals


def test_failure_too_few_residuals():
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, _lib as L
    cells, _ = _cells(1, [0, 1])
    reg = api.n_scan_normal_reg("P2L")
    poses = np.array([[0, 0, 0], [500.0, 500.0, 0.0]])            # no overlap -> no associations
    scans = [api.MapPointNormal(cells=c) for c in cells]
    ok, pg, _ = reg.Register(scans, poses)
    assert not ok and reg.summary_.status == L.ERR_TOO_FEW_RESIDUALS
    np.testing.assert_array_equal(pg, poses)                      # Tsrc untouched on failure
    ok_o, po, ro = O.register(cells, poses, _oracle_par(reg))
    assert not ok_o and ro.outer_iters == reg.summary_.outer_iters == 1
    empty = api.MapPointNormal(cells=cells[0][:0])
    ok, _, _ = reg.Register([empty, scans[1]], poses)
    assert not ok


def test_association_pairs_and_normal_equations():
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    cells, gt = _cells(4, [0, 1, 2])
    poses = np.array([_rel(gt[0], gt[f]) for f in [0, 1, 2]])
    poses[-1] += [0.3, 0.2, -0.01]
    scans = [api.MapPointNormal(cells=c) for c in cells]
    for cost, opt in [("P2L", 4), ("P2P", 0), ("P2D", 4)]:
        reg = api.n_scan_normal_reg(cost, "Huber", 0.1, opt)
        for itr in (1, 2):
            cc = api.CeresCost(reg, scans, poses, itr=itr)
            pairs_g, w_g = cc.blocks()
            pairs_o, w_o = O.associate(cells, poses, _oracle_par(reg), itr)
            np.testing.assert_array_equal(pairs_g, pairs_o)
            np.testing.assert_allclose(w_g, w_o, rtol=1e-12)
            x = poses[-1] + [0.05, -0.02, 0.003]
            H, g, cost_v = cc.normal_eq(x)
            Ho, go, co, nres = O.normal_eq(cells, poses, _oracle_par(reg), itr, x)
            np.testing.assert_allclose(H, Ho, rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(g, go, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(cost_v, co, rtol=1e-11)
            r, J = cc.evaluate(x)
            assert r.shape[0] == nres
            # raw residuals/Jacobian: finite-difference check of the Ceres-compatible evaluation
            eps = 1e-6
            for kk in range(3):
                xp = x.copy(); xp[kk] += eps
                xm = x.copy(); xm[kk] -= eps
                fd = (cc.evaluate(xp)[0] - cc.evaluate(xm)[0]) / (2 * eps)
                np.testing.assert_allclose(J[:, kk], fd, atol=1e-5)


def test_get_cost_matches_oracle_and_decreases_when_aligned():
    """CFEARQuality (coral_alignment_quality/src/alignment_checker/AlignmentQuality.cpp:330-354):
    P2L, Huber 0.3, GetCost; and the reference's only assertion on this path
    (scan_learning_interface_tests.cpp:39-48): the aligned pose scores better than a (1,1,0) offset."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    cells, gt = _cells(5, [0, 1])
    scans = [api.MapPointNormal(cells=c) for c in cells]
    reg = api.n_scan_normal_reg("P2L", "Huber", 0.3)
    truth = _rel(gt[0], gt[1])
    out = {}
    for name, off in [("aligned", (0, 0, 0)), ("shifted", (1.0, 1.0, 0.0))]:
        poses = np.array([[0, 0, 0], truth + off])
        ok, cost, res = reg.GetCost(scans, poses)
        ok_o, cost_o, res_o, score_o = O.get_cost(cells, poses, _oracle_par(reg))
        assert ok and ok_o and res.shape == res_o.shape
        np.testing.assert_allclose(cost, cost_o, rtol=1e-11)
        np.testing.assert_allclose(res, res_o, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(reg.getScore(), score_o, rtol=1e-11)
        out[name] = (cost / max(res.size, 1), res.size)
    assert out["aligned"][1] > out["shifted"][1]          # more overlap when aligned


def test_register_batch_matches_single_calls():
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    cells, gt = _cells(6, [0, 1, 2, 3])
    scans = [api.MapPointNormal(cells=c) for c in cells]
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    rng = np.random.default_rng(1)
    jobs, ojobs = [], []
    for a in range(4):
        for b in range(4):
            if a == b:
                continue
            guess = _rel(gt[a], gt[b]) + np.concatenate([rng.normal(0, 0.5, 2), rng.normal(0, 0.02, 1)])
            T = np.array([[0, 0, 0], guess])
            jobs.append(([scans[a], scans[b]], T))
            ojobs.append(([cells[a], cells[b]], T))
    out = reg.RegisterBatch(jobs)
    for r, (c, T) in zip(out, ojobs):
        ok_o, po, ro = O.register(c, T, _oracle_par(reg))
        assert (r["status"] == 0) == ok_o
        assert r["outer_iters"] == ro.outer_iters and r["lm_iters"] == ro.lm_iters
        assert np.abs(r["pose"][:2] - po[-1, :2]).max() <= POS_TOL and abs(r["pose"][2] - po[-1, 2]) <= ROT_TOL


def test_iteration_counts_over_many_registrations():
    """The matcher's trust-region bookkeeping uses Newton-refined reciprocals / rsqrt and a polynomial sincos (an ulp or two
    from the IEEE forms the oracle uses): over 120 perturbed registrations of the headline configuration (CFEAR-3: P2P,
    Huber 0.1, weights 4, up to 8 associations x 100 LM iterations) every integer outcome must still be the oracle's."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    reg = api.n_scan_normal_reg("P2P", "Huber", 0.1)
    reg.par.weight_opt = 4
    reg.SetParameters(8, 100)
    rng = np.random.default_rng(11)
    jobs, ojobs = [], []
    for seed in (21, 22, 23):
        cells, gt = _cells(seed, [0, 1, 2, 3, 4], k=12)
        scans = [api.MapPointNormal(cells=c) for c in cells]
        for rep in range(40):
            n = int(rng.integers(2, 6))                                  # 1 .. 4 keyframes + the new scan
            idx = sorted(rng.choice(5, size=n, replace=False).tolist())
            T = np.array([gt[i] for i in idx], dtype=np.float64)
            T[-1] += np.concatenate([rng.normal(0, 0.4, 2), rng.normal(0, 0.015, 1)])
            jobs.append(([scans[i] for i in idx], T))
            ojobs.append(([cells[i] for i in idx], T))
    out = reg.RegisterBatch(jobs)
    lm_total = 0
    for r, (c, T) in zip(out, ojobs):
        ok_o, po, ro = O.register(c, T, _oracle_par(reg))
        assert (r["status"] == 0) == ok_o
        assert (r["outer_iters"], r["lm_iters"], r["num_residuals"]) == (ro.outer_iters, ro.lm_iters, ro.num_residuals)
        assert np.abs(r["pose"][:2] - po[-1, :2]).max() <= POS_TOL and abs(r["pose"][2] - po[-1, 2]) <= ROT_TOL
        lm_total += ro.lm_iters
    assert lm_total > 120 * 5                                           # the solver really iterated


def test_sharded_candidates_single_rank_through_the_library():
    """dist.register_candidates_sharded with the real per-rank compute (no process group = 1 rank)."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    from tbv_slam_public_amd import dist as cdist
    cells, gt = _cells(7, [0, 2, 4])
    scans = [api.MapPointNormal(cells=c) for c in cells]
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    rng = np.random.default_rng(3)
    jobs, ojobs = [], []
    for _ in range(9):
        a, b = rng.choice(3, size=2, replace=False)
        T = np.array([[0, 0, 0], _rel(gt[2 * a], gt[2 * b]) + rng.normal(0, 0.3, 3) * [1, 1, 0.05]])
        jobs.append(([scans[a], scans[b]], T))
        ojobs.append(([cells[a], cells[b]], T))
    prepared = reg.PrepareBatch(jobs)
    out = cdist.register_candidates_sharded(jobs, lambda local: reg.RegisterBatch(prepared))
    out2 = cdist.register_candidates_sharded(jobs, cdist.default_register_fn(reg))
    np.testing.assert_array_equal(out["pose"], out2["pose"])
    for r, (c, T) in zip(out, ojobs):
        ok_o, po, ro = O.register(c, T, _oracle_par(reg))
        assert (r["status"] == 0) == ok_o and r["outer_iters"] == ro.outer_iters
        assert np.abs(r["pose"][:2] - po[-1, :2]).max() <= POS_TOL and abs(r["pose"][2] - po[-1, 2]) <= ROT_TOL


def test_sixteen_scans():
    """n_scans = 16 (the C-ABI maximum): 15 keyframes x ~350 cells, and 8 keyframes.  Both must reproduce the oracle."""
    from tbv_slam_public_amd import api
    frames = [0, 1, 2, 3, 4, 5]
    cells, gt = _cells(6, frames)
    rel = [_rel(gt[0], gt[f]) for f in frames]
    for n_key in (15, 8):
        cs, ps = [], []
        for j in range(n_key):
            f = j % 5
            cs.append(cells[f])
            ps.append(rel[f])
        cs.append(cells[5])
        ps.append(rel[5] + np.array([0.3, -0.2, 0.008]))
        reg = api.n_scan_normal_reg("P2L", "Huber", 0.1, 0)
        _compare_register(reg, cs, np.array(ps))
        ok, c, res, _ = __import__("oracle.pyoracle", fromlist=["x"]).get_cost(cs, np.array(ps), _oracle_par(reg))
        scans = [api.MapPointNormal(cells=c_) for c_ in cs]
        okg, cg, resg = reg.GetCost(scans, np.array(ps))
        assert okg == ok and len(resg) == len(res)
        np.testing.assert_allclose(cg, c, rtol=1e-10)


def test_full_batch_rigid_frame_equivariance():
    """Size-independent property at the batch size of BASELINE configs[3] (4096 candidate registrations in one launch):
    moving every input pose of a job by one rigid transform G moves the registered pose by G and changes nothing else
    (the matcher only sees T_tar^-1 T_src).  Not a comparison with the oracle: a self-consistency check that scales."""
    from tbv_slam_public_amd import api
    cells, gt = _cells(3, [0, 1, 2, 3, 4, 5])
    scans = [api.MapPointNormal(cells=c) for c in cells]
    rng = np.random.default_rng(8)
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)

    def compose(a, b):
        c, s = np.cos(a[2]), np.sin(a[2])
        return np.array([c * b[0] - s * b[1] + a[0], s * b[0] + c * b[1] + a[1], a[2] + b[2]])
    jobs, moved, Gs = [], [], []
    for _ in range(4096):
        i = int(rng.integers(0, 5))
        j = int(rng.integers(i + 1, 6))
        guess = _rel(gt[i], gt[j]) + rng.normal(0, [0.5, 0.5, 0.03])
        G = np.array([rng.uniform(-400, 400), rng.uniform(-400, 400), rng.uniform(-np.pi, np.pi)])
        P = np.array([[0.0, 0.0, 0.0], guess])
        jobs.append(([scans[i], scans[j]], P))
        moved.append(([scans[i], scans[j]], np.array([compose(G, P[0]), compose(G, P[1])])))
        Gs.append(G)
    a = reg.RegisterBatch(jobs)
    b = reg.RegisterBatch(moved)
    assert (a["status"] == 0).mean() > 0.99
    np.testing.assert_array_equal(a["status"], b["status"])
    same_path = (a["outer_iters"] == b["outer_iters"]) & (a["num_residuals"] == b["num_residuals"])
    assert same_path.mean() > 0.97                       # float rounding of a moved frame may flip a borderline match
    exp = np.array([compose(G, p) for G, p in zip(Gs, a["pose"])])
    ok = (a["status"] == 0) & same_path
    d = np.abs(b["pose"][ok] - exp[ok])
    # rounding of the moved frame (|G| up to 400 m) perturbs the LM path at 1e-9..1e-8 m; a flipped association that
    # happens to keep the counts shows up as a few 1e-6 m -- all far inside the 1e-4 m / 1e-5 rad bar
    assert d[:, :2].max() < POS_TOL and d[:, 2].max() < ROT_TOL
    assert np.percentile(d[:, :2].max(1), 99) < 1e-6 and np.percentile(d[:, 2], 99) < 1e-8
    np.testing.assert_allclose(b["final_cost"][ok], a["final_cost"][ok], rtol=1e-6, atol=1e-9)
    # the batch is deterministic: the same launch twice gives the same bits
    np.testing.assert_array_equal(reg.RegisterBatch(jobs)["pose"], a["pose"])


def test_sharded_c_abi_entry_world_one_and_mock_world_two():
    """cfear_register_batch_sharded: world 1 equals cfear_register_batch; with world 2 and a mock all_gather that supplies
    the other rank's block (computed by a plain batch call), both "ranks" return the full, ordered result list."""
    import ctypes as C
    from tbv_slam_public_amd import api, _lib as L, synth
    from oracle import pyoracle as O
    imgs, gt, _ = synth.scene_v1(17, 4)
    scans = []
    for f in range(4):
        r = api.filter_kstrongest(imgs[f], 40, 60, 0.0438, 2.5)
        scans.append(api.MapPointNormal(r["xyzi"][0, :int(r["n_points"][0])], 3.0, (0, 0), True))
    rng = np.random.default_rng(0)
    jobs = []
    for q in range(9):
        i, j = (q % 3), (q % 3) + 1
        jobs.append(([scans[i], scans[j]], np.array([[0, 0, 0.0], gt[j] - gt[i] + rng.normal(0, 0.2, 3) * [1, 1, 0.05]])))
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    arr, n, keep = reg.PrepareBatch(jobs)
    ref = reg.RegisterBatch((arr, n, keep))
    lib, ctx = api.default_context()._lib, api.default_context()
    lib.cfear_register_batch_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
    out = np.zeros(n, L.RESULT_DTYPE)
    ctx.check(lib.cfear_register_batch_sharded(ctx.h, arr, n, C.byref(reg.par), 0, 1, None, None, out.ctypes.data))
    np.testing.assert_array_equal(out, ref)
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    per = (n + 1) // 2
    for me in range(2):
        def gather(user, send, recv, nbytes, me=me):
            other = 1 - me
            lo, hi = other * per, min(n, other * per + per)
            blk = np.zeros(per, L.RESULT_DTYPE)
            blk[:hi - lo] = ref[lo:hi]
            peer = blk.tobytes() + struct.pack("<ii", 0, hi - lo)          # the block + its trailer {rank status, records}
            assert nbytes == len(peer)
            parts = [C.string_at(send, nbytes), peer] if me == 0 else [peer, C.string_at(send, nbytes)]
            C.memmove(recv, b"".join(parts), 2 * nbytes)
            return 0
        cb = CB(gather)
        out = np.zeros(n, L.RESULT_DTYPE)
        ctx.check(lib.cfear_register_batch_sharded(ctx.h, arr, n, C.byref(reg.par), me, 2, C.cast(cb, C.c_void_p), None, out.ctypes.data))
        np.testing.assert_array_equal(out, ref)
    # a rank whose own block fails (here: an invalid job) still enters the collective and reports its status there
    entered = []
    def gather_fail(user, send, recv, nbytes):
        entered.append(struct.unpack_from("<ii", C.string_at(send, nbytes), nbytes - 8))
        C.memmove(recv, C.string_at(send, nbytes) * 2, 2 * nbytes)
        return 0
    cbf = CB(gather_fail)
    bad = (L.RegJob * n)()
    C.memmove(bad, arr, C.sizeof(bad))
    bad[0].n_scans = 1                                                      # n_scans must be >= 2: this rank's block fails
    rc = lib.cfear_register_batch_sharded(ctx.h, bad, n, C.byref(reg.par), 0, 2, C.cast(cbf, C.c_void_p), None, out.ctypes.data)
    assert rc != 0 and len(entered) == 1 and entered[0][0] == rc


def test_small_and_large_batches_agree():
    """Batches of <= 64 jobs run on the 8-wavefront kernel, larger ones on the 4-wavefront kernel (different summation
    order): the same nine registrations alone and inside a batch of 108 agree in every integer outcome and to 1e-12 in the
    pose -- what a sharded run relies on when its per-rank block falls below 65 jobs."""
    from tbv_slam_public_amd import api, synth
    imgs, gt, _ = synth.scene_v1(17, 4)
    scans = []
    for f in range(4):
        r = api.filter_kstrongest(imgs[f], 40, 60, 0.0438, 2.5)
        scans.append(api.MapPointNormal(r["xyzi"][0, :int(r["n_points"][0])], 3.0, (0, 0), True))
    rng = np.random.default_rng(3)
    jobs = []
    for q in range(9):
        i, j = (q % 3), (q % 3) + 1
        jobs.append(([scans[i], scans[j]], np.array([[0, 0, 0.0], gt[j] - gt[i] + rng.normal(0, 0.2, 3) * [1, 1, 0.05]])))
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    small = reg.RegisterBatch(jobs)
    large = reg.RegisterBatch(jobs * 12)
    for q in range(9):
        for rep in range(12):
            b = large[rep * 9 + q]
            assert b["status"] == small[q]["status"] and b["outer_iters"] == small[q]["outer_iters"]
            assert b["lm_iters"] == small[q]["lm_iters"] and b["num_residuals"] == small[q]["num_residuals"]
            np.testing.assert_allclose(b["pose"], small[q]["pose"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(b["final_cost"], small[q]["final_cost"], rtol=1e-11)


def test_every_form_of_the_matcher_agrees():
    """One kernel, several forms (wavefronts per registration x LDS per workgroup, chosen by the batch): 144 registrations run
    as the library chooses (8 wavefronts, tables and correspondence arrays side by side), then forced through the regular
    form (4 wavefronts, 40 KB: tables aliased with the LM arrays, restaged per outer iteration), through 14 KB (keyframe tables
    staged in several groups, the dense arrays' tail in global memory -- paths that only unusually large registrations reach
    otherwise), the half-CU and whole-CU forms and the 2-wavefront form.  Same statements everywhere: every integer outcome
    identical, poses and costs to rounding (the forms add the fp64 sums in different orders).  The oracle judges a sample."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api
    from tbv_slam_public_amd import _lib as L
    rng = np.random.default_rng(5)
    for cost, loss, opt in (("P2P", "Huber", 4), ("P2L", "Cauchy", 0), ("P2D", "Huber", 0)):
        reg = api.n_scan_normal_reg(cost, loss, 0.1)
        reg.par.weight_opt = opt
        reg.SetParameters(8, 100)
        jobs, ojobs = [], []
        for seed in (31, 32):
            cells, gt = _cells(seed, [0, 1, 2, 3, 4], k=20)
            scans = [api.MapPointNormal(cells=c) for c in cells]
            for rep in range(36):
                n = int(rng.integers(2, 6))
                idx = sorted(rng.choice(5, size=n, replace=False).tolist())
                T = np.array([gt[i] for i in idx], dtype=np.float64)
                T[-1] += np.concatenate([rng.normal(0, 0.4, 2), rng.normal(0, 0.015, 1)])
                jobs.append(([scans[i] for i in idx], T))
                ojobs.append(([cells[i] for i in idx], T))
        base = reg.RegisterBatch(jobs)
        runs = {}
        try:
            for waves, kb in ((4, 40), (4, 14), (8, 80), (16, 160), (2, 20)):
                reg.ctx.set_option(L.OPT_MATCHER_WAVES, waves); reg.ctx.set_option(L.OPT_MATCHER_LDS_KB, kb)
                runs["%d x %d KB" % (waves, kb)] = reg.RegisterBatch(jobs)
            # the 8-wavefront form has two builds: a batch of at most one workgroup per CU (the runs above) takes the one
            # compiled for the whole register file; 288 workgroups take the 128-VGPR build.  Same statements, same order
            # of additions: the records are identical.
            reg.ctx.set_option(L.OPT_MATCHER_WAVES, 8); reg.ctx.set_option(L.OPT_MATCHER_LDS_KB, 80)
            many = reg.RegisterBatch(jobs * 4)
            for a, b in zip(runs["8 x 80 KB"], many[:len(jobs)]):
                for f in ("status", "outer_iters", "lm_iters", "num_residuals", "final_cost", "score"):
                    assert a[f] == b[f], (cost, f, a[f], b[f])
                assert np.array_equal(a["pose"], b["pose"]), (cost, a["pose"], b["pose"])
        finally:
            reg.ctx.set_option(L.OPT_MATCHER_WAVES, 0); reg.ctx.set_option(L.OPT_MATCHER_LDS_KB, 0)
        for name, out in runs.items():
            for a, b in zip(base, out):
                key = lambda r: (int(r["status"]), int(r["outer_iters"]), int(r["lm_iters"]), int(r["num_residuals"]))
                assert key(a) == key(b), (cost, name, key(a), key(b))
                np.testing.assert_allclose(b["pose"], a["pose"], rtol=0, atol=1e-11, err_msg=name)
                np.testing.assert_allclose(b["final_cost"], a["final_cost"], rtol=1e-10, err_msg=name)
        for r, (c, T) in list(zip(runs["4 x 14 KB"], ojobs))[::6]:        # a sample against the oracle itself
            ok_o, po, ro = O.register(c, T, _oracle_par(reg))
            assert (r["status"] == 0) == ok_o
            assert (r["outer_iters"], r["lm_iters"], r["num_residuals"]) == (ro.outer_iters, ro.lm_iters, ro.num_residuals)
            assert np.abs(r["pose"][:2] - po[-1, :2]).max() <= POS_TOL and abs(r["pose"][2] - po[-1, 2]) <= ROT_TOL


def test_candidates_from_a_scan_table_equal_the_job_batch():
    """cfear_register_candidates: the scans' views live in a device table, a candidate is two indices and two poses (56
    bytes), the job records are written on the device.  Same kernels behind it: every record of a 600-candidate batch (the
    compact geometry) and of a 40-candidate batch (the 8-wavefront kernel) equals cfear_register_batch's, byte for byte;
    results left on the device read back the same; bad indices are an argument error."""
    import torch
    from tbv_slam_public_amd import _lib as L
    from tbv_slam_public_amd import api
    cells, gt = _cells(6, [0, 1, 2, 3, 4])
    scans = [api.MapPointNormal(cells=c) for c in cells]
    table = api.ScanTable(scans)
    assert len(table) == 5
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    rng = np.random.default_rng(2)
    for n in (600, 40):
        tgt, src = rng.integers(0, 5, n), rng.integers(0, 5, n)
        src = np.where(src == tgt, (src + 1) % 5, src)
        tp = rng.normal(0, 0.3, (n, 3)) * [1, 1, 0.02]
        guess = np.stack([gt[b] - gt[a] for a, b in zip(tgt, src)]) + tp + rng.normal(0, 0.3, (n, 3)) * [1, 1, 0.05]
        jobs = [([scans[a], scans[b]], np.array([tp[i], guess[i]])) for i, (a, b) in enumerate(zip(tgt, src))]
        ref = reg.RegisterBatch(jobs)
        cands = api.ScanTable.candidates(tgt, src, guess, tp)
        got = reg.RegisterCandidates(table, cands)
        assert got.tobytes() == ref.tobytes()
        buf = torch.zeros(n * L.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        assert reg.RegisterCandidates(table, cands, device_ptr=buf.data_ptr()) == n
        reg.ctx.synchronize()
        assert buf.cpu().numpy().tobytes() == ref.tobytes()
        assert (ref["status"] == 0).mean() > 0.8
    bad = api.ScanTable.candidates([0], [7], [[0, 0, 0]])
    with pytest.raises(L.CfearError) as e:
        reg.RegisterCandidates(table, bad)
    assert e.value.status == L.ERR_INVALID_ARGUMENT
    table.close()


def test_a_scan_table_keeps_its_scans_alive():
    """cfear_scan_table_create takes a reference on every scan: destroying the caller's handles -- and creating new scans, which
    would recycle the freed slabs -- leaves the table's candidates registering against the SAME cells (round 4's table pointed
    into recycled memory)."""
    from tbv_slam_public_amd import api
    cells, gt = _cells(6, [0, 1, 2, 3])
    scans = [api.MapPointNormal(cells=c) for c in cells]
    table = api.ScanTable(scans)
    table._scans = []                                            # (the Python wrapper's own keep-alive list out of the way)
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    rng = np.random.default_rng(4)
    n = 48
    tgt, src = rng.integers(0, 4, n), rng.integers(0, 4, n)
    src = np.where(src == tgt, (src + 1) % 4, src)
    guess = np.stack([gt[b] - gt[a] for a, b in zip(tgt, src)]) + rng.normal(0, 0.2, (n, 3)) * [1, 1, 0.03]
    cands = api.ScanTable.candidates(tgt, src, guess)
    before = reg.RegisterCandidates(table, cands)
    for sc in scans:
        sc.close()
    del scans
    other, _ = _cells(9, [0, 1, 2, 3])
    fresh = [api.MapPointNormal(cells=c) for c in other]         # same capacities: these would have taken the freed slabs
    after = reg.RegisterCandidates(table, cands)
    assert after.tobytes() == before.tobytes() and (before["status"] == 0).mean() > 0.8
    table.close()
    del fresh


def _rccl():
    """librccl through ctypes: a ONE-rank communicator (ncclGetUniqueId + ncclCommInitRank), as a C++ host would own it."""
    import ctypes as C
    for name in ("librccl.so.1", "librccl.so"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("librccl.so not found")


def test_sharded_c_abi_entry_over_a_real_one_rank_rccl_communicator():
    """The north-star collective really runs: cfear_register_batch_sharded(..., cfear_rccl_allgather, &comm) with an
    ncclComm_t of one rank (the single-GPU box) -- kernel output written straight into the send buffer, ncclAllGather on
    the context's stream, one read-back -- equals cfear_register_batch; so does the host-buffer callback route
    (cfear_gather_records) and the sharded verification entry."""
    import ctypes as C
    from tbv_slam_public_amd import api, _lib as L, synth
    rccl = _rccl()
    imgs, gt, _ = synth.scene_v1(21, 4)
    scans = []
    for f in range(4):
        r = api.filter_kstrongest(imgs[f], 40, 60, 0.0438, 2.5)
        scans.append(api.MapPointNormal(r["xyzi"][0, :int(r["n_points"][0])], 3.0, (0, 0), True))
    rng = np.random.default_rng(1)
    jobs = []
    for q in range(37):
        i, j = (q % 3), (q % 3) + 1
        jobs.append(([scans[i], scans[j]], np.array([[0, 0, 0.0], gt[j] - gt[i] + rng.normal(0, 0.2, 3) * [1, 1, 0.05]])))
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    arr, n, keep = reg.PrepareBatch(jobs)
    ref = reg.RegisterBatch((arr, n, keep))
    ctx = api.default_context()
    lib = ctx._lib

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        class RcclComm(C.Structure):
            _fields_ = [("ctx", C.c_void_p), ("nccl_comm", C.c_void_p), ("world", C.c_int32), ("pad", C.c_int32)]
        cc = RcclComm(ctx.h, comm, 1, 0)
        lib.cfear_register_batch_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]
        fn = C.cast(lib.cfear_rccl_allgather, C.c_void_p)
        out = np.zeros(n, L.RESULT_DTYPE)
        ctx.check(lib.cfear_register_batch_sharded(ctx.h, arr, n, C.byref(reg.par), 0, 1, fn, C.byref(cc), out.ctypes.data))
        np.testing.assert_array_equal(out, ref)
        # the generic (host-buffer) route through the same communicator
        lib.cfear_gather_records.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        out2 = np.zeros(n, L.RESULT_DTYPE)
        ctx.check(lib.cfear_gather_records(ref.ctypes.data, n, L.RESULT_DTYPE.itemsize, 1, 0, fn, C.byref(cc), out2.ctypes.data))
        np.testing.assert_array_equal(out2, ref)
        # results may live on the device: cfear_register_batch with a device pointer leaves them there
        import torch
        d_out = torch.zeros(n * L.RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        assert reg.RegisterBatchInto((arr, n, keep), d_out.data_ptr()) == n
        ctx.synchronize()
        np.testing.assert_array_equal(d_out.cpu().numpy().view(L.RESULT_DTYPE), ref)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_python_sharded_path_runs_the_collective_at_world_one():
    """dist.register_candidates_sharded inside a ONE-rank nccl process group: records written into the send tensor on the
    GPU, all_gather_into_tensor, one copy back -- equal to the plain batch."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from tbv_slam_public_amd import api, synth
    from tbv_slam_public_amd import dist as cdist
    imgs, gt, _ = synth.scene_v1(22, 3)
    scans = []
    for f in range(3):
        r = api.filter_kstrongest(imgs[f], 40, 60, 0.0438, 2.5)
        scans.append(api.MapPointNormal(r["xyzi"][0, :int(r["n_points"][0])], 3.0, (0, 0), True))
    rng = np.random.default_rng(2)
    jobs = [([scans[q % 2], scans[q % 2 + 1]], np.array([[0, 0, 0.0], gt[q % 2 + 1] - gt[q % 2] + rng.normal(0, 0.2, 3) * [1, 1, 0.05]]))
            for q in range(11)]
    reg = api.n_scan_normal_reg("P2L")
    reg.SetParameters(4, 10)
    ref = reg.RegisterBatch(jobs)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        out = cdist.register_candidates_sharded(jobs, cdist.default_register_fn(reg))
    finally:
        dist.destroy_process_group()
    np.testing.assert_array_equal(out, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("with_comm", [False, True])
def test_candidate_pipe_equals_register_candidates(graph, with_comm):
    """cfear_candidate_pipe (the sharded loop-closure step kept in flight: upload -> expand -> matcher on the context's stream,
    ncclAllGather + read-back on the exchange stream) returns byte for byte what cfear_register_candidates returns -- direct
    launches and the captured hipGraph, with and without a one-rank ncclComm_t, steps collected in and out of order, batches
    whose contents and sizes change between steps (a replayed graph must read the NEW candidates; another size re-captures)."""
    from tbv_slam_public_amd import api, _lib as L
    cells, gt = _cells(6, [0, 1, 2, 3, 4])
    ctx = api.Context(0)                                            # a stream of its own (the legacy stream cannot be captured)
    scans = [api.MapPointNormal(cells=c, ctx=ctx) for c in cells]
    table = api.ScanTable(scans, ctx=ctx)
    reg = api.n_scan_normal_reg("P2L", ctx=ctx)
    reg.SetParameters(4, 10)
    comm = None
    if with_comm:
        _rccl()                                                     # (skips without librccl)
        comm = api.RcclComm(reg.ctx, 1, 0)
    rng = np.random.default_rng(8)

    def batch(n):
        tgt, src = rng.integers(0, 5, n), rng.integers(0, 5, n)
        src = np.where(src == tgt, (src + 1) % 5, src)
        guess = np.stack([gt[b] - gt[a] for a, b in zip(tgt, src)]) + rng.normal(0, 0.3, (n, 3)) * [1, 1, 0.05]
        return api.ScanTable.candidates(tgt, src, guess)
    pipe = api.CandidatePipe(reg, table, 700, comm, 0, 1, depth=3, graph=graph, timing=True)
    batches = [batch(n) for n in (600, 600, 600, 600, 40, 600, 1, 700)]
    refs = [reg.RegisterCandidates(table, b) for b in batches]
    # in order, one at a time
    for b, r in zip(batches, refs):
        assert pipe.collect(pipe.submit(b)).tobytes() == r.tobytes()
    # three in flight, collected out of order
    for k in range(0, 6, 3):
        tk = [pipe.submit(b) for b in batches[k:k + 3]]
        for q in (1, 0, 2):
            assert pipe.collect(tk[q]).tobytes() == refs[k + q].tobytes()
    st = pipe.stats()
    assert st["steps"] == 14 and st["exchange_ms"] > 0.0
    assert st["graph_slots"] == (3 if graph else 0)
    # a fourth step without a collect is refused, the pipe stays usable
    tk = [pipe.submit(batches[0]) for _ in range(3)]
    with pytest.raises(L.CfearError) as e:
        pipe.submit(batches[0])
    assert e.value.status == L.ERR_INVALID_ARGUMENT
    for t in tk:
        assert pipe.collect(t).tobytes() == refs[0].tobytes()
    # a candidate outside the table fails THAT step (the rank still enters the exchange); the next step is fine
    bad = batches[4].copy()
    bad["source"][3] = 99
    t = pipe.submit(bad)
    with pytest.raises(L.CfearError) as e:
        pipe.collect(t)
    assert e.value.status == L.ERR_INVALID_ARGUMENT
    assert pipe.collect(pipe.submit(batches[4])).tobytes() == refs[4].tobytes()
    # an empty batch
    assert pipe.collect(pipe.submit(batches[0][:0])).shape[0] == 0
    pipe.close()
    if comm is not None:
        comm.close()
    table.close()
