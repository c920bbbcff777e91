#!/usr/bin/env python
"""bench.py -- CFEAR scan registrations/s on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], "Oxford 10k sequence, per-frame odometry registration streamed on
1xMI355X"): `--streams` independent synthetic Oxford-like sequences (400 x 3360 uint8 polar sweeps,
CFEAR-3 preset: k=40, z_min=60, r=3, P2P/Huber 0.1/weights 4, 4-keyframe window) each advance one
frame per step.  One step = one pass of the whole hot path over the batch: k-strongest filter ->
motion compensation -> oriented surface points -> many-to-one registration -> keyframe policy, polar
image in, SE(2) pose out.  All polar images are resident in HBM before the timed region; every
stream has its own physical copy of its frames.  value = streams * steps / time, aggregated over
ranks (weak scaling: every rank runs its own `--streams` sequences, no collective on the data path).

Also reports: roofline of the only kernel that moves compulsory HBM bytes (kstrongest_rows, the
polar sweep) from hipEvent timings taken inside the timed region, the per-kernel time breakdown,
the CPU oracle timed on the host (1 thread) on a bounded sample of the same frames, and the pose
error of the GPU path against that CPU path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS, COLS, K_STRONGEST = 400, 3360, 40
N_SEEDS = 16                     # distinct synthetic sequences; streams replicate them round-robin
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _usable_cpus():
    """Host threads this process may really run at once: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())               # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def loopclosure_main(args):
    """BASELINE configs[3]: `--candidates` loop-closure candidate registrations (P2L, Huber 0.1, Uniform,
    SetParameters(4,10) -- loopclosure.cpp:56-57) between cached surface-point sets, block-sharded over the
    ranks, results all_gathered (RCCL) in candidate order."""
    import torch
    import torch.distributed as dist
    from tbv_slam_public_amd import api, synth
    from tbv_slam_public_amd import dist as cdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    ctx = api.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    n_frames, n_cand = 40, args.candidates
    sc = synth.Scene(3)
    gt = np.stack([sc.pose_at(f, n_frames) for f in range(n_frames)])
    scans = []
    for f in range(n_frames):                      # every rank featurises the scans it may reference
        r = api.filter_kstrongest(sc.render(f, n_frames), 40, 60, 0.0438, 2.5, ctx=ctx)
        scans.append(api.MapPointNormal(r["xyzi"][0, :int(r["n_points"][0])], 3.0, (0, 0), True, ctx=ctx))
    rng = np.random.Generator(np.random.PCG64(11))

    def rel(a, b):
        c, s = np.cos(a[2]), np.sin(a[2])
        d = b[:2] - a[:2]
        return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], b[2] - a[2]])
    jobs = []
    for _ in range(n_cand):                        # pairs 2..6 frames (5-15 m) apart, guess error N(0, 1 m), N(0, 3 deg)
        i = int(rng.integers(0, n_frames - 7))
        j = i + int(rng.integers(2, 7))
        guess = rel(gt[i], gt[j]) + np.concatenate([rng.normal(0, 1.0, 2), rng.normal(0, np.deg2rad(3.0), 1)])
        jobs.append(([scans[i], scans[j]], np.array([[0.0, 0.0, 0.0], guess])))
    reg = api.n_scan_normal_reg("P2L", ctx=ctx)
    reg.SetParameters(4, 10)
    lo, hi, _per = cdist.shard_range(n_cand, world, rank)
    prepared = reg.PrepareBatch(jobs[lo:hi])
    fn = lambda _local: reg.RegisterBatch(prepared)
    for _ in range(max(args.warmup, 1)):
        out = cdist.register_candidates_sharded(jobs, fn)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = cdist.register_candidates_sharded(jobs, fn)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "loop-closure candidate registrations/sec (cached features, P2L 4x10)",
            "value": n_cand * args.steps / elapsed, "unit": "registrations/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 1), "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32 search / f64 solve", "data": "synthetic (scene_v1)",
            "config": {"workload": "configs[3]: %d loop-closure candidates sharded over %d rank(s), all_gather of "
                                   "72-byte result records" % (n_cand, world), "candidates": n_cand},
            "ok_fraction": float((out["status"] == 0).mean()), "mean_outer_iters": float(out["outer_iters"].mean()),
            "reference_cpu_ms_per_candidate": "8.3-9.7 (evaluation/data/oxford_all_tbv_model_8/job_*/time_statistics.txt)"}))
    if world > 1:
        dist.destroy_process_group()


def verify_main(args):
    """Full loop-candidate verification of `--candidates` candidates per step (RegisterLoopCandidate +
    VerifyLoopCandidate + ApplyConstratins, tbv_slam/src/tbv_slam/loopclosure.cpp:320-384, 261-274): registration,
    CorAl and CFEAR alignment quality, both classifiers; 3 candidates per query node; block-sharded over the ranks,
    one all_gather of 480-byte records."""
    import torch
    import torch.distributed as dist
    from tbv_slam_public_amd import api, synth
    from tbv_slam_public_amd import dist as cdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    ctx = api.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    n_frames, n_cand = 40, args.candidates
    sc = synth.Scene(3)
    gt = np.stack([sc.pose_at(f, n_frames) for f in range(n_frames)])
    scans, peaks = [], []
    for f in range(n_frames):
        r = api.filter_kstrongest(sc.render(f, n_frames), 40, 60, 0.0438, 2.5, want_peaks=True, ctx=ctx)
        scans.append(api.MapPointNormal(r["xyzi"][0, :int(r["n_points"][0])], 3.0, (0, 0), True, ctx=ctx))
        peaks.append(torch.from_numpy(np.ascontiguousarray(r["xyzi_peaks"][0, :int(r["n_peaks"][0])])).cuda())   # device-resident
    rng = np.random.Generator(np.random.PCG64(11))

    def rel(a, b):
        c, s = np.cos(a[2]), np.sin(a[2])
        d = b[:2] - a[:2]
        return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], b[2] - a[2]])
    cands = []
    for q in range(n_cand):
        i = int(rng.integers(0, n_frames - 7))
        j = i + int(rng.integers(2, 7))
        guess = rel(gt[j], gt[i]) + np.concatenate([rng.normal(0, 1.0, 2), rng.normal(0, np.deg2rad(3.0), 1)])
        cands.append(dict(from_scan=scans[j], to_scan=scans[i], from_peaks=peaks[j], to_peaks=peaks[i], from_pose=gt[j],
                          t_be_guess=guess, sc_sim=float(rng.uniform(0.05, 0.5)), odom_bounds=float(rng.uniform(0, 0.3)),
                          group=q // 3))
    par = api.verify_params(ctx)
    lo, hi, _per = cdist.shard_range(n_cand, world, rank)
    prepared = api.prepare_verify_batch(cands[lo:hi])
    fn = lambda _local: api.verify_loop_candidates(prepared, par, ctx)
    for _ in range(max(args.warmup, 1)):
        out = cdist.verify_candidates_sharded(cands, fn, par.model_threshold, bool(par.all_candidates), fn_selects=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = cdist.verify_candidates_sharded(cands, fn, par.model_threshold, bool(par.all_candidates), fn_selects=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "loop-closure candidate verifications/sec (register P2L 4x10 + CorAl + CFEAR quality + classifiers)",
            "value": n_cand * args.steps / elapsed, "unit": "candidates/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 1), "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32 search / f64 solve + entropy", "data": "synthetic (scene_v1)",
            "config": {"workload": "%d loop-closure candidates (3 per query) verified end to end, sharded over %d rank(s), "
                                   "all_gather of 480-byte records" % (n_cand, world), "candidates": n_cand},
            "reg_ok_fraction": float(out["reg_ok"].mean()), "accepted_fraction": float(out["accepted"].mean()),
            "reference_cpu_ms_per_candidate": "8.3-9.7 Register + 20-22 VerifyByAlignment "
                                              "(evaluation/data/oxford_all_tbv_model_8/job_*/time_statistics.txt)"}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=2048, help="independent sequences per GPU")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of CPU baseline (0 = auto, ~10-30 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["odometry", "loopclosure", "verify"], default="odometry",
                    help="odometry = BASELINE configs[1] (the headline metric); loopclosure = configs[3]: "
                         "a batch of candidate registrations from cached features, sharded over the ranks "
                         "with one RCCL all_gather of the result records per step")
    ap.add_argument("--candidates", type=int, default=4096)
    ap.add_argument("--bins-major", action="store_true",
                    help="feed the images as [range bins][azimuths] (non-Oxford drivers): adds the GPU rotation of "
                         "radarDriver::Callback (radar_driver.cpp:74-90) to every step; not the BASELINE layout")
    ap.add_argument("--peaks-pass", action="store_true",
                    help="after the timed region, time a second pass that also builds the peaks cloud of every sweep and "
                         "report it as with_peaks_cloud (off by default: it would mix its launches into a kernel trace "
                         "of this command)")
    ap.add_argument("--keep-nodes", action="store_true",
                    help="also keep every frame's compensated peaks cloud (pose-graph nodes, cfear_odometry_get_*); "
                         "off in the BASELINE configuration")
    ap.add_argument("--cov-sampling", action="store_true",
                    help="odometry: also estimate every frame's covariance by cost sampling (27 GetCost per "
                         "registration, odometrykeyframefuser.cpp:203-208; off in the reference's presets)")
    args = ap.parse_args()
    if args.workload == "loopclosure":
        return loopclosure_main(args)
    if args.workload == "verify":
        return verify_main(args)

    import torch
    import torch.distributed as dist
    from tbv_slam_public_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B, K, W = args.streams, args.steps, max(args.warmup, 1)   # frame 0 of a stream is not a registration
    F = W + K + 1                 # one extra frame: the last timed step prefetches (filters) it
    # ---- synthetic input, resident in HBM: [F][B][ROWS][COLS] uint8 -------------------------------
    t0 = time.time()
    base = []
    for sd in range(min(N_SEEDS, B)):
        sc = synth.Scene(1000 * rank + sd)
        base.append(np.stack([sc.render(f, F) for f in range(F)]))
    base = np.stack(base)                                            # [S][F][R][C]
    dbase = torch.from_numpy(base).to(dev)
    frames = torch.empty((F, B, ROWS, COLS), dtype=torch.uint8, device=dev)
    for b in range(B):
        frames[:, b] = dbase[b % dbase.shape[0]]                     # a physical copy per stream
    del dbase
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    stream = torch.cuda.current_stream().cuda_stream
    ctx = api.Context(local_rank, stream=stream)
    if args.bins_major:
        frames = torch.rot90(frames, -1, dims=(2, 3)).contiguous()   # [F][B][COLS][ROWS]: what such a driver publishes
        torch.cuda.synchronize()                                     # the library enqueues on its own stream
    od = api.OdometryKeyframeFuser(B, *frames.shape[2:], api.odometry_params(estimate_cov_by_sampling=int(args.cov_sampling),
                                                                             rotate_ccw=int(args.bins_major),
                                                                             keep_nodes=int(args.keep_nodes)), ctx=ctx)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for f in range(W):
        info = od.process(frames[f], frames[f + 1])      # warm-up; also primes the filter prefetch of frame W
    barrier()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    poses = np.zeros((K, B, 3))
    n_points = np.zeros((K, B), np.int64)
    n_cells = np.zeros((K, B), np.int64)
    status_bad = 0
    t1 = time.perf_counter()
    for s in range(K):
        # The filter of the next frame is enqueued behind this frame's kernels (it needs no state of this
        # frame).  Frame W's sweep ran during warm-up and frame W+K's sweep runs inside the timed region,
        # so exactly K filter sweeps (and K of every other kernel) are timed.
        info = od.process(frames[W + s], frames[W + s + 1])
        poses[s] = info["pose"]
        n_points[s] = info["n_points"]
        n_cells[s] = info["n_cells"]
        status_bad += int((info["reg_status"] != 0).sum())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t2 = time.perf_counter()
    elapsed = t2 - t1
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    # Optional second pass (--peaks-pass), not part of `value`, with the peaks cloud switched on (AxialNonMaxSupress + second cloud + its
    # compensation): the matcher does not use it, but TBV's driver produces it for every sweep (radar_driver.cpp:59-62)
    # and loop closure consumes it, so the rate of that fuller per-frame job is reported next to the headline.
    with_peaks = None
    if args.peaks_pass and not args.keep_nodes and not args.cov_sampling:
        od.close()
        od2 = api.OdometryKeyframeFuser(B, *frames.shape[2:], api.odometry_params(rotate_ccw=int(args.bins_major), keep_nodes=1),
                                        ctx=ctx)
        for f in range(W):
            od2.process(frames[f], frames[f + 1])
        barrier()
        tp1 = time.perf_counter()
        for s_ in range(K):
            od2.process(frames[W + s_], frames[W + s_ + 1])
        barrier()
        tp = time.perf_counter() - tp1
        if world > 1:
            t = torch.tensor([tp], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tp = float(t.item())
        with_peaks = {"value": B * K * world / tp, "unit": "registrations/s", "ms_per_step": tp / K * 1e3,
                      "note": "same job plus the peaks cloud of every sweep (keep_nodes = 1); not the headline"}
        od2.close()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        bad = torch.tensor([status_bad], dtype=torch.int64, device=dev)
        dist.all_reduce(bad)
        status_bad = int(bad.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_units = B * K * world
    value = total_units / elapsed
    # ---- roofline of the polar sweep (kstrongest_rows) ------------------------------------------------
    ms, launches = prof.get("kstrongest_rows", (0.0, 0))
    avg_ms = ms / max(launches, 1)
    nf = float(n_points.mean())
    # algorithmic bytes per scan filtered = R*C + 16*N_f + 4*R, N_f = points kept
    bytes_per_scan = ROWS * COLS + 16.0 * nf + 4 * ROWS
    achieved = (bytes_per_scan * B) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    breakdown = {k: {"ms_per_step": v[0] / max(K, 1), "launches": int(v[1])} for k, v in prof.items()}
    # HBM traffic of that kernel from the committed PMC passes (rocprofv3 cannot run inside this process):
    # per-scan FETCH_SIZE x2 + WRITE_SIZE measured by tools/profile.sh, scaled to this launch's batch
    traffic = None
    tj = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    if os.path.exists(tj):
        t = json.load(open(tj))
        traffic = (t["fetch_bytes_per_scan"] + t["write_bytes_per_scan"]) * B

    # ---- CPU baseline: the oracle (port), 1 thread, bounded sample of the same frames -----------------
    cpu = None
    cpu_mt = None
    pose_err = None
    if not args.no_cpu_baseline and world == 1:        # rank 0 at N=1 only
        from oracle import pyoracle as O
        n_seq = base.shape[0]
        passes = 5 if not args.cpu_frames else 1          # ~15 s of single-thread CPU work at the defaults
        budget_frames = args.cpu_frames or passes * n_seq * F
        reg = O.reg_params(cost="P2P", loss="Huber", loss_limit=0.1, weight_opt=4, regularization=0.0)
        done, compared, tc0 = 0, 0, time.perf_counter()
        err_xy, err_th = 0.0, 0.0
        for ps in range(passes):
            for sd in range(n_seq):
                fz = O.Fuser(reg, res=3.0, submap_scan_size=4, weight_intensity=True)
                for f in range(F):
                    if done >= budget_frames:
                        break
                    sr, si, scn = O.kstrongest(base[sd, f], K_STRONGEST, 60)
                    cloud = O.kstrongest_cloud(sr, si, scn, 0.0438, 2.5)
                    pose, oi = fz.process(cloud)
                    done += 1
                    if ps == 0 and W <= f < W + K:
                        d = np.abs(poses[f - W, sd] - pose)
                        err_xy, err_th = max(err_xy, d[:2].max()), max(err_th, d[2])
                        compared += 1
        tc = time.perf_counter() - tc0
        cpu = {"value": done / tc, "unit": "registrations/s", "cores": 1, "kind": "port",
               "sample": "%d frames = %d pass(es) over %d sequences x %d frames of this run's input, full path "
                         "filter->pose, oracle/liboracle.so g++ -O3, 1 thread, %.1f s; host has %d cores"
                         % (done, passes, n_seq, F, tc, os.cpu_count())}
        pose_err = {"max_abs_xy_m": err_xy, "max_abs_theta_rad": err_th, "frames_compared": compared}
        # the same oracle on the host's cores, one sequence per thread at a time like the reference's NR_WORKERS
        # processes (one native call per sequence, GIL released; the oracle has no shared state)
        from concurrent.futures import ThreadPoolExecutor
        T = _usable_cpus()

        def one_sequence(sd):
            fz = O.Fuser(reg, res=3.0, submap_scan_size=4, weight_intensity=True)
            fz.run_sequence(base[sd % n_seq], K_STRONGEST, 60, 0.0438, 2.5)   # one native call per sequence
            return F
        n_tasks = 48 * T                                  # ~10 s at ~8 ms per frame
        tm0 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            done_mt = sum(ex.map(one_sequence, range(n_tasks)))
        tm = time.perf_counter() - tm0
        cpu_mt = {"value": done_mt / tm, "unit": "registrations/s", "cores": T, "kind": "port",
                  "sample": "%d sequences x %d frames on %d threads (cgroup quota / affinity of this host: %d of %d CPUs), "
                            "%.1f s" % (n_tasks, F, T, T, os.cpu_count() or 0, tm)}

    out = {
        "metric": "radar scan registrations/sec (400x3360 polar)",
        "value": value,
        "unit": "registrations/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8 filter / f32 search / f64 solve",
        "data": "synthetic (scene_v1 walls+scatterers, %d distinct sequences replicated to %d streams per GPU, "
                "each stream its own HBM copy)" % (base.shape[0], B),
        "config": {"workload": "configs[1]: per-frame CFEAR-3 odometry registration (k=40, z_min=60, r=3, P2P, "
                               "4-keyframe window%s), polar image -> pose, %d streams in flight per GPU"
                               % (", + covariance by cost sampling" if args.cov_sampling else "", B),
                   "streams_per_gpu": B, "rows": ROWS, "cols": COLS, "parallelism": "replicas x%d" % world},
        "roofline": {"bound": "hbm", "kernel": "kstrongest_rows", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/r01/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per scan x batch)",
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_scan * B,
                     "mean_points_per_scan": nf},
        "with_peaks_cloud": with_peaks,
        "cpu_baseline": cpu,
        "cpu_baseline_all_threads": cpu_mt,
        "pose_error_vs_cpu": pose_err,
        "kernel_breakdown": breakdown,
        "mean_cells_per_scan": float(n_cells.mean()),
        "failed_registrations": status_bad,
        "input_generation_s": t_gen,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
