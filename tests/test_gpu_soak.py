"""GPU: a randomized sweep of the whole path against the CPU oracle on seeds no other test uses.

Every other GPU test pins a handful of seeds per feature; this one draws scenes, filter parameters, cost / loss / weight
combinations, window sizes and start poses at random and pushes them through filter -> surface points -> registration, the
registrations as ONE batch per parameter set so that the batch-size dependent forms of the matcher all see them (and once
more through the regular 4-wavefront form, which only batches of thousands reach by themselves).  The oracle judges every
record.  Further down: the batched odometry on random presets, CorAl, covariance by cost sampling, Scan Context and the CA-CFAR
pipeline, the same way.  CFEAR_SOAK=<n> sets the number of scenes, CFEAR_SOAK_SEED=<k> draws another sweep (default 6: a few seconds; profiles/r05/soak.txt holds a run
with 2000: 23 892 registrations, 666 pipeline configurations x 4 streams x 7 frames, 5 994 CorAl jobs, ...)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POS_TOL, ROT_TOL = 1e-4, 1e-5
N_SCENES = int(os.environ.get("CFEAR_SOAK", "6"))
SEED = int(os.environ.get("CFEAR_SOAK_SEED", "0"))            # another sweep: other scenes, other draws


def _rel(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    d = b[:2] - a[:2]
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], b[2] - a[2]])


def _cmp_cells(got, exp):
    assert got.shape[0] == exp.shape[0]
    np.testing.assert_array_equal(got["nsamples"], exp["nsamples"])
    np.testing.assert_allclose(got["mean"], exp["mean"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got["cov"], exp["cov"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(got["scale"], exp["scale"], rtol=1e-8)


def test_random_scenes_through_the_whole_path():
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, synth
    from tbv_slam_public_amd import _lib as L
    rng = np.random.default_rng(20260929 + SEED)
    combos = [("P2L", "Huber", 0), ("P2P", "Huber", 4), ("P2D", "Huber", 0), ("P2L", "Cauchy", 4), ("P2P", "None", 1),
              ("P2L", "Tukey", 2), ("P2P", "SoftLOne", 3), ("P2L", "Combined", 0)]
    jobs_by_combo = {c: [] for c in combos}
    n_filter = n_cells = 0
    for q in range(N_SCENES):
        seed = 500000 + 10000 * SEED + q
        kind = int(rng.integers(0, 3))
        nf = 5
        if kind == 0:
            imgs, gt, _ = synth.scene_v1(seed, nf)
        elif kind == 1:
            imgs, gt, _ = synth.scene_dense(seed, nf)
        else:
            sc = synth.Scene(seed, circle_frames=64)
            imgs = synth.render_frames_torch(sc, list(range(nf)), "cpu").numpy()
            gt = np.stack([sc.pose_at(f, nf) for f in range(nf)])
        k = int(rng.choice([8, 12, 20, 40]))
        z_min = int(rng.choice([55, 60, 70, 90]))
        radius = float(rng.choice([2.5, 3.0, 3.5]))
        wi = bool(rng.integers(0, 2))
        # filter: the GPU's lists and clouds are the oracle's, bit for bit
        r = api.filter_kstrongest(np.ascontiguousarray(imgs[:nf]), k, z_min, 0.0438, 2.5, want_peaks=False)
        cells_o, maps = [], []
        for f in range(nf):
            sr, si, scn = O.kstrongest(imgs[f], k, z_min)
            cloud = O.kstrongest_cloud(sr, si, scn, 0.0438, 2.5)
            np.testing.assert_array_equal(r["sel_count"][f], scn)
            np.testing.assert_array_equal(r["sel_range"][f], sr)
            assert r["n_points"][f] == cloud.shape[0]
            np.testing.assert_array_equal(r["xyzi"][f, :cloud.shape[0]], cloud)
            n_filter += 1
            exp = O.surface_points(cloud, radius, 1.0, (0, 0), wi)
            m = api.MapPointNormal(cloud, radius, (0.0, 0.0), wi)
            _cmp_cells(m.GetCells(), exp)
            n_cells += exp.shape[0]
            # the registration below runs both sides on the ORACLE's cells: a 1e-9 difference in a mean must not decide a tie
            cells_o.append(exp)
            maps.append(api.MapPointNormal(cells=exp))
        if min(len(c) for c in cells_o) < 12:
            continue
        for rep in range(6):
            combo = combos[int(rng.integers(0, len(combos)))]
            n = int(rng.integers(2, nf + 1))
            idx = sorted(rng.choice(nf, size=n, replace=False).tolist())
            T = np.array([_rel(gt[idx[0]], gt[i]) for i in idx], dtype=np.float64)
            T[-1] += np.concatenate([rng.normal(0, 0.5, 2), rng.normal(0, 0.02, 1)])
            jobs_by_combo[combo].append(([maps[i] for i in idx], [cells_o[i] for i in idx], T))
    n_reg = 0
    for (cost, loss, opt), jobs in jobs_by_combo.items():
        if not jobs:
            continue
        reg = api.n_scan_normal_reg(cost, loss, 0.1, opt)
        opar = O.reg_params(cost=reg.par.cost, loss=reg.par.loss, loss_limit=reg.par.loss_limit, weight_opt=reg.par.weight_opt,
                            max_outer=reg.par.max_itr_association, max_inner=reg.par.max_itr_solver, min_outer=reg.par.min_itr,
                            radius=reg.par.radius, cov_scale=reg.par.cov_scale, regularization=reg.par.regularization,
                            first_itr=reg.par.itr)
        expect = [O.register(c, T, opar) for _, c, T in jobs]
        # GetCost at the start poses (n_scan_normal.cpp:186-211), one launch
        costs = reg.GetCostBatch([(m, T) for m, _, T in jobs])
        for rec, (_, c, T) in zip(costs, jobs):
            ok_c, cost_o, res_o, score_o = O.get_cost(c, T, opar)
            assert (rec["status"] == 0) == ok_c, (cost, loss, opt, [len(x) for x in c])
            if ok_c:
                assert rec["num_residuals"] == len(res_o)
                np.testing.assert_allclose(rec["final_cost"], cost_o, rtol=1e-9, atol=1e-12)
            else:                                             # too few residuals (:200-203): the reference leaves its outputs untouched
                assert rec["num_residuals"] <= 1
        try:
            for waves, kb in ((0, 0), (4, 40)):
                reg.ctx.set_option(L.OPT_MATCHER_WAVES, waves); reg.ctx.set_option(L.OPT_MATCHER_LDS_KB, kb)
                out = reg.RegisterBatch([(m, T) for m, _, T in jobs])
                for rec, (ok_o, po, ro), (_, c, T) in zip(out, expect, jobs):
                    tag = (cost, loss, opt, waves, [len(x) for x in c], T[-1].tolist())
                    assert (rec["status"] == 0) == ok_o, tag
                    assert (rec["outer_iters"], rec["lm_iters"], rec["num_residuals"]) == (ro.outer_iters, ro.lm_iters, ro.num_residuals), tag
                    if ok_o:
                        assert np.abs(rec["pose"][:2] - po[-1, :2]).max() <= POS_TOL and abs(rec["pose"][2] - po[-1, 2]) <= ROT_TOL, tag
                        np.testing.assert_allclose(rec["final_cost"], ro.final_cost, rtol=1e-9, atol=1e-12)
                    n_reg += 1
        finally:
            reg.ctx.set_option(L.OPT_MATCHER_WAVES, 0); reg.ctx.set_option(L.OPT_MATCHER_LDS_KB, 0)
    print("soak: %d scenes, %d sweeps filtered, %d cells, %d registrations compared" % (N_SCENES, n_filter, n_cells, n_reg))
    assert n_reg >= 2 * N_SCENES


def test_random_pipeline_configurations_keep_per_frame_parity():
    """The batched odometry (polar sweeps in, poses out) on random presets: cost, loss, weights, window size, voxel size, k and
    the input route drawn at random -- and the sweep width: 3360 (the BASELINE metric) or Oxford's native 3768, whose rows do not
    start on the 16-byte grid --, four fresh streams each, every frame of every stream against the oracle's fuser."""
    from test_gpu_odometry import _run
    rng = np.random.default_rng(7 + SEED)
    n_cfg = max(2, N_SCENES // 3)
    for q in range(n_cfg):
        par = dict(reg_cost=int(rng.choice([0, 1])), reg_loss=int(rng.choice([1, 2])), reg_weight_opt=int(rng.choice([0, 4])),
                   submap_scan_size=int(rng.choice([1, 3, 4, 5])), res=float(rng.choice([3.0, 3.5])),
                   kstrong_k_strongest=int(rng.choice([12, 40])), weight_intensity=int(rng.integers(0, 2)))
        seeds = [600000 + 10000 * SEED + 4 * q + j for j in range(4)]
        cols = 3768 if (q % 2 == 1 or rng.integers(0, 3) == 0) else 3360      # (every second configuration at least)
        od = _run(seeds, 7, bool(rng.integers(0, 2)), par=par, scene=dict(cols=cols))
        od.close()


def test_random_coral_batches():
    """CorAl alignment quality (AlignmentQuality.cpp:8-230) of random scan pairs under random offsets, radii and weighting, as
    batches: {joint, sep, overlap}, the validity flag and the valid-point count against the oracle."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, synth
    rng = np.random.default_rng(11 + SEED)
    total = 0
    for q in range(max(2, N_SCENES // 3)):
        imgs, gt, _ = synth.scene_v1(700000 + 10000 * SEED + q, 3)
        k = int(rng.choice([12, 40]))
        clouds = []
        for f in range(3):
            sr, si, scn = O.kstrongest(imgs[f], k, 60)
            clouds.append(O.kstrongest_cloud(sr, si, scn, 0.0438, 2.5, mask=O.peaks(imgs[f], k, sr, scn)))
        radius = float(rng.choice([0.6, 1.0, 1.5]))
        weight = bool(rng.integers(0, 2))
        jobs = []
        for a, b in [(0, 1), (0, 2), (1, 2)]:
            for _ in range(3):
                scale = rng.choice([0.05, 0.5, 2.0])
                jobs.append((clouds[a], gt[a], clouds[b], gt[b], rng.normal(0, [scale, scale, 0.03 * scale])))
        out, _ = api.coral_quality_batch(jobs, radius, weight)
        for (rc, rp, sc, sp, off), r in zip(jobs, out):
            ok, eq, pp = O.coral_quality(rc, sc, rp, sp, off, radius, weight)
            np.testing.assert_allclose([r["joint"], r["sep"], r["overlap"]], eq, rtol=1e-8, atol=1e-12)
            assert bool(r["valid"]) == ok and r["count_valid"] == int((pp[:, 2] > 0).sum())
            total += 1
    print("soak: %d CorAl jobs compared" % total)


def test_random_covariances_by_cost_sampling():
    """approximateCovarianceBySampling (odometrykeyframefuser.cpp:261-380) behind random registrations: the sample grid exact,
    the sample costs to 1e-10, the fitted covariance and its success flag as the oracle's."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, synth
    from test_gpu_register import _oracle_par
    rng = np.random.default_rng(13 + SEED)
    done = 0
    for q in range(max(2, N_SCENES // 3)):
        imgs, gt, _ = synth.scene_v1(800000 + 10000 * SEED + q, 4)
        cells = []
        for f in range(4):
            sr, si, scn = O.kstrongest(imgs[f], 12, 60)
            cells.append(O.surface_points(O.kstrongest_cloud(sr, si, scn, 0.0438, 2.5), 3.0, 1.0, (0, 0), True))
        cost, loss, opt = [("P2P", "Huber", 4), ("P2L", "Huber", 0), ("P2D", "Huber", 0), ("P2L", "Cauchy", 4)][int(rng.integers(0, 4))]
        n_scans = int(rng.integers(2, 5))
        idx = sorted(rng.choice(4, size=n_scans, replace=False).tolist())
        poses = np.array([_rel(gt[idx[0]], gt[i]) for i in idx])
        poses[-1] += rng.normal(0, [0.3, 0.3, 0.01])
        reg = api.n_scan_normal_reg(cost, loss, 0.1, opt)
        scans = [api.MapPointNormal(cells=cells[i]) for i in idx]
        ok, pg, _ = reg.Register(scans, poses)
        if not ok:
            continue
        res = reg.summary_
        n = int(rng.choice([2, 3, 4]))
        sp = reg.sampling_params(samples_per_axis=n)
        got_ok, cov, smp = reg.approximateCovarianceBySampling(scans, pg, sampling=sp, want_samples=True)
        exp_ok, exp_cov, exp_smp = O.cov_by_sampling([cells[i] for i in idx], pg, _oracle_par(reg), res.final_cost, res.num_residuals,
                                                     sp.xy_range, sp.yaw_range, n, sp.covariance_scaler)
        np.testing.assert_array_equal(smp[:, :3], exp_smp[:, :3])
        np.testing.assert_allclose(smp[:, 3], exp_smp[:, 3], rtol=1e-10)
        assert got_ok == exp_ok
        if exp_ok:
            sel = np.ix_([0, 1, 5], [0, 1, 5])
            np.testing.assert_allclose(cov[sel], exp_cov[sel], rtol=1e-4, atol=1e-14)
        done += 1
    print("soak: %d sampled covariances compared" % done)
    assert done >= 1


def test_random_scan_context_descriptors_and_distances():
    """Radar Scan Context (RadarScancontext.cpp) on random peak clouds: descriptors (a few edge bins may differ, see
    test_gpu_scancontext.py), and the column-shift distance of random pairs, bit for bit."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, synth
    rng = np.random.default_rng(17 + SEED)
    n_pairs = 0
    for q in range(max(2, N_SCENES // 3)):
        imgs, _, _ = synth.scene_v1(900000 + 10000 * SEED + q, 4)
        k = int(rng.choice([12, 40]))
        clouds = []
        for f in range(4):
            sr, si, scn = O.kstrongest(imgs[f], k, 60)
            clouds.append(O.kstrongest_cloud(sr, si, scn, 0.0438, 2.5, mask=O.peaks(imgs[f], k, sr, scn)))
        fn, div = [("sum", 1000.0), ("max", 1.0), ("sum", 1.0)][int(rng.integers(0, 3))]
        par = api.sc_params(desc_function=fn, desc_divider=div)
        desc = api.sc_descriptors(clouds, par)[0][:, 0]
        for i, c in enumerate(clouds):
            assert (desc[i] != O.sc_descriptor(c, 40, 120, 80.0, fn, div, 0.0, 0.0)).sum() <= 4
        cand = np.concatenate([desc, np.stack([np.roll(desc[j], -int(rng.integers(1, 120)), axis=1) for j in range(2)])])
        pairs = [(int(rng.integers(0, 4)), int(rng.integers(0, cand.shape[0]))) for _ in range(12)]
        dist, shift = api.sc_distance_batch(desc, cand, pairs)
        for (a, b), d, s in zip(pairs, dist, shift):
            ed, es = O.sc_distance(desc[a], cand[b])
            assert s == es and d == ed, (q, a, b, d, ed, s, es)
            n_pairs += 1
    print("soak: %d Scan Context distances compared" % n_pairs)


def test_random_cacfar_pipelines():
    """The CA-CFAR odometry (cfar.cpp:35-83 in front of the same fuser) with random window / guard / false-alarm settings on
    random 0.175 m worlds, [range bins][azimuths] input (the fused decode) against the pre-rotated route, and the pre-rotated
    route's point counts, cells and poses against the oracle."""
    from oracle import pyoracle as O
    from tbv_slam_public_amd import api, synth
    rng = np.random.default_rng(19 + SEED)
    for q in range(max(1, N_SCENES // 6)):
        imgs = synth.scene_v1(950000 + 10000 * SEED + q, 4, range_res=0.175, ccw=True, n_walls=int(rng.choice([40, 120])), noise_scale=float(rng.choice([4.0, 7.0])))[0]
        win, guard = int(rng.choice([24, 40])), int(rng.choice([4, 10]))
        pfa, zmin = float(rng.choice([0.01, 0.003])), float(rng.choice([20.0, 35.0]))
        kw = dict(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=zmin, cacfar_nb_guard_cells=guard, cacfar_window_size=win,
                  cacfar_false_alarm_rate=pfa, radar_ccw=1, kstrong_range_res=0.175)
        reps = 16                                                    # (the fused decode starts at 16 streams)
        plain = api.OdometryKeyframeFuser(reps, 400, 3360, api.odometry_params(**kw))
        rot = api.OdometryKeyframeFuser(reps, 3360, 400, api.odometry_params(rotate_ccw=1, **kw))
        par = plain.par if hasattr(plain, "par") else api.odometry_params(**kw)
        fz = O.Fuser(O.reg_params(cost=par.reg.cost, loss=par.reg.loss, loss_limit=0.1, weight_opt=par.reg.weight_opt, regularization=0.0),
                     res=par.res, submap_scan_size=par.submap_scan_size, weight_intensity=bool(par.weight_intensity), radar_ccw=True)
        for f in range(4):
            batch = np.ascontiguousarray(np.broadcast_to(imgs[f], (reps, 400, 3360)))
            a = plain.process(batch)
            c = rot.process(np.ascontiguousarray(np.rot90(batch, -1, axes=(1, 2))))
            for name in a.dtype.names:
                np.testing.assert_array_equal(a[name], c[name], err_msg="rotated " + name)
            cloud, _ = O.cacfar(imgs[f], win, guard, pfa, 0.175, zmin, 2.5)
            pose, info = fz.process(cloud)
            assert (a["n_points"] == cloud.shape[0]).all() and (a["n_cells"] == info[0]).all(), (q, f)
            d = np.abs(a["pose"] - pose)
            assert d[:, :2].max() <= POS_TOL and d[:, 2].max() <= ROT_TOL, (q, f, d.max(axis=0))
        plain.close(); rot.close()
