#!/usr/bin/env python
"""bench.py -- CFEAR scan registrations/s on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], "Oxford 10k sequence, per-frame odometry registration streamed on
1xMI355X"): `--streams` independent synthetic Oxford-like sequences (400 x 3360 uint8 polar sweeps,
CFEAR-3 preset: k=40, z_min=60, r=3, P2P/Huber 0.1/weights 4, 4-keyframe window).  Every stream drives a
closed circle of `--ring` frames, so a ring of frames is an endless sequence; the rings of `--sequences`
distinct worlds are resident in HBM and every stream is a distinct (world, start frame) pair: no two streams
ever read the same image in one step, and the ring (>> the 256 MiB Infinity Cache) is re-read from HBM every
step.  One STEP = `--frames-per-step` consecutive frames of every stream through the whole hot path:
k-strongest filter -> motion compensation -> oriented surface points -> many-to-one registration -> keyframe
policy, polar image in, SE(2) pose out (so that the K timed steps last seconds, not milliseconds).
value = streams * frames_per_step * steps / time, aggregated over ranks (weak scaling: every rank runs its
own streams, no collective on the data path).

`python bench.py --gpus N` starts the N ranks itself (one process per GPU under torch.distributed.run, RCCL);
under an external launcher (WORLD_SIZE set) it is one of the ranks.

Also reported in the same JSON line (1 GPU): the roofline of the only kernel that moves compulsory HBM bytes
(kstrongest_rows, hipEvent timings from inside the timed region), the same for dense-row scenes
(roofline_dense + the full-path rate on them), the per-kernel breakdown, one stream alone (single_stream: the
literal configs[1] latency case), host-resident input (host_input: pinned images, H2D inside the timed region),
the loop-closure candidate workload (loopclosure), and the CPU oracle timed on the host beside it.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS, COLS, K_STRONGEST = 400, 3360, 40
IMG = ROWS * COLS
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


# ---------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with N > 1 re-executes itself as N ranks
# ---------------------------------------------------------------------------------------------------------
def launch_command(argv, n, port):
    """The command `python bench.py --gpus n ...` turns itself into (one rank per GPU, RCCL over xGMI)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def maybe_self_launch(args, argv):
    if "WORLD_SIZE" in os.environ or args.gpus <= 1:
        return
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but %d GPU(s) visible: refusing to run (a run on fewer GPUs "
                             "would report the wrong n_gpus)\n" % (args.gpus, have))
            sys.exit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = launch_command(argv, args.gpus, _free_port())
    os.execvpe(cmd[0], cmd, env)


def emit(D, out):
    """The ONE JSON line, as the LAST thing on stdout: the process group is torn down first and C stdio is flushed (RCCL
    prints its version banner through buffered C stdio, which would otherwise land behind the line at exit)."""
    rank = D.rank
    D.close()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)


class Dist:
    """torch.distributed plumbing of one rank: RCCL ("nccl") on GPUs, gloo for the launcher dry run."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = args.dry_run
        if self.world != max(args.gpus, 1):
            raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, self.world))
        import torch
        self.torch = torch
        if not self.dry:
            assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
            if torch.cuda.device_count() <= self.local_rank:
                raise SystemExit("bench.py: rank %d has no GPU %d" % (self.rank, self.local_rank))
            torch.cuda.set_device(self.local_rank)              # rank i <-> GPU i
        self.dev = torch.device("cpu") if self.dry else torch.device("cuda", self.local_rank)
        self.dist = None
        # the sharded workloads bring the process group up at ONE rank too, so that the driver's N = 1 run executes the
        # same collective (a 1-rank RCCL all_gather) as N = 8
        if self.world > 1 or not self.dry:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:
                import socket
                with socket.socket() as sk:                       # a free port: nothing else joins this group
                    sk.bind(("127.0.0.1", 0))
                    os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
                os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            try:
                if self.dry:
                    dist.init_process_group("gloo")
                else:
                    dist.init_process_group("nccl", device_id=self.dev)
                self.dist = dist
            except Exception as e:                                # a lone rank can do without the group; N ranks cannot
                if self.world > 1:
                    raise
                print("bench.py: no 1-rank process group (%s): collectives are skipped" % e, file=sys.stderr)

    def barrier(self):
        if not self.dry:
            self.torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if not self.dist:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, x):
        """[value of every rank] through one all_gather: also the proof of how many ranks RCCL really connects."""
        if not self.dist:
            return [float(x)]
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.dev)
        out = self.torch.empty(self.world, dtype=self.torch.float64, device=self.dev)
        self.dist.all_gather_into_tensor(out, t)
        return [float(v) for v in out.cpu()]

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------
# input: rings of sweeps along closed circles, rendered on the GPU
# ---------------------------------------------------------------------------------------------------------
def make_rings(n_seq, ring, seed0, dev, dense=False):
    """uint8 [n_seq][ring][ROWS][COLS] on the GPU + the Scene objects."""
    import torch
    from tbv_slam_public_amd import synth
    rings = torch.empty((n_seq, ring, ROWS, COLS), dtype=torch.uint8, device=dev)
    kw = dict(synth.DENSE_KW) if dense else {}
    for s in range(n_seq):
        sc = synth.Scene(seed0 + s, circle_frames=ring, cols=COLS, **kw)
        rings[s] = synth.render_frames_torch(sc, list(range(ring)), dev)
    torch.cuda.synchronize()
    return rings


class StreamSet:
    """B streams over a ring buffer: stream b = (sequence b % S, start frame (b // S) * F / P)."""

    def __init__(self, B, S, F):
        P = (B + S - 1) // S
        assert P <= F, "more streams per sequence than ring frames: streams would share images"
        b = np.arange(B)
        self.seq = (b % S).astype(np.int64)
        self.start = ((b // S) * F // P).astype(np.int64)
        self.F = F

    def offsets(self, t):
        return (self.seq * self.F + (self.start + t) % self.F) * IMG


def run_odometry(od, rings, ss, n_frames, t0=0, record=None):
    """Advances every stream by n_frames frames (frame t0 .. t0 + n_frames - 1 of its ring walk); the filter of frame
    t + 1 is prefetched behind frame t.  record: (array [n_frames, n, 3], n) receives the poses of streams 0 .. n-1."""
    stats = {"points": 0.0, "cells": 0.0, "bad": 0, "frames": 0}
    for t in range(t0, t0 + n_frames):
        info = od.process_offsets(rings, ss.offsets(t), ss.offsets(t + 1))
        stats["points"] += float(info["n_points"].mean())
        stats["cells"] += float(info["n_cells"].mean())
        stats["bad"] += int((info["reg_status"] != 0).sum())
        stats["frames"] += 1
        if record is not None:
            record[0][t - t0] = info["pose"][:record[1]]
    return stats


def roofline_of(prof, n_scans_per_launch, mean_points):
    """algorithmic bytes per scan filtered = R*C + 16*N_f + 4*R, N_f = points kept -- `achieved` / `frac`
    follow that figure.  In the batched odometry the sweep does not write 16-byte points but 4-byte keys (intensity << 24 |
    bin) and two counters per row, so what it really moves is R*C + 4*N_f + 8*R: reported beside it as `fused_*`."""
    kernel = "kstrongest_rows" if "kstrongest_rows" in prof or "kstrong_image" not in prof else "kstrong_image"   # ([bins][azimuths] sources: the fused decode + sweep)
    ms, launches = prof.get(kernel, (0.0, 0))
    avg_ms = ms / max(launches, 1)
    bytes_per_scan = IMG + 16.0 * mean_points + 4 * ROWS
    fused_bytes_per_scan = IMG + 4.0 * mean_points + 8 * ROWS
    per_s = n_scans_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    achieved = bytes_per_scan * per_s
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": avg_ms, "launches": int(launches),
            "algorithmic_bytes_per_launch": bytes_per_scan * n_scans_per_launch, "mean_points_per_scan": mean_points,
            "fused_bytes_per_launch": fused_bytes_per_scan * n_scans_per_launch,
            "fused_achieved": fused_bytes_per_scan * per_s, "fused_frac": fused_bytes_per_scan * per_s / HBM_PEAK_GBS,
            "polar_read_only_GBs": IMG * per_s, "polar_read_only_frac": IMG * per_s / HBM_PEAK_GBS,
            "scans_per_launch": n_scans_per_launch}


def pingpong(t, n):
    """0, 1, .., n-1, n-2, .., 1, 0, 1, ..: a finite run of frames walked forth and back."""
    if n < 2:
        return 0
    q = t % (2 * n - 2)
    return q if q < n else 2 * n - 2 - q


def whole_path_of(mean_points, mean_cells, keyframes, registrations_per_s_per_gpu):
    """SURVEY.md 8(d), "scan registration from raw polar": R*C + 32*N_f + 96*N_s + 48*sum_j N_t,j algorithmic bytes per
    registration (image read; points written then read; cells written then read; the keyframes' cells read once) against
    the HBM spec -- the honest figure for the whole path: only the polar sweep is HBM-bound, the surface-point and matcher
    kernels work on KB-scale sets in LDS / L2 and are issue- and latency-bound."""
    b = IMG + 32.0 * mean_points + 96.0 * mean_cells + 48.0 * keyframes * mean_cells
    gbs = b * registrations_per_s_per_gpu / 1e9
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_registration": b}


# ---------------------------------------------------------------------------------------------------------
# neighbouring workloads (functions so that the odometry run can report them in its own line)
# ---------------------------------------------------------------------------------------------------------
class LoopClosureWorld:
    """What a loop-closure thread holds (BASELINE configs[3]): the graph nodes' cached surface points as a scan table on this
    rank's GPU, and candidate batches drawn among them -- pairs 2..6 frames (5-15 m) apart, guess error N(0, 1 m), N(0, 3 deg);
    P2L, Huber 0.1, Uniform, SetParameters(4, 10) (loopclosure.cpp:56-57).  Every rank featurises the scans itself (duplicates
    across ranks tolerated, no feature exchange: SURVEY 8e) and draws the SAME candidate list (seeded)."""

    def __init__(self, D, graph=None):
        import torch
        from tbv_slam_public_amd import api, synth
        self.D = D
        # a stream of the library's own (the pipe's compute stream; its exchange and preparation streams are the pipe's): the
        # loop-closure thread shares no stream with torch -- the Python mirror of the step (dist.py) then waits for the
        # kernels before its collective, which it checks for itself (Context.shares_torch_stream)
        self.ctx = ctx = api.Context(D.local_rank)
        if graph:
            # a precomputed simple_graph.sgh (tools/make_graph.py, or the reference's SaveGraph): the cached surface points
            # of every node feed the matcher directly, as loopclosure::Register does (types.h:119-122)
            nodes = [nd for nd in api.LoadSimpleGraph(graph) if nd["cells"] is not None and len(nd["cells"]) > 0]
            assert len(nodes) >= 8, "the graph needs at least 8 nodes with surface points"
            self.gt = np.stack([nd["T_xyt"] for nd in nodes])
            self.scans = [api.MapPointNormal(cells=nd["cells"], ctx=ctx) for nd in nodes]
            self.lap = len(nodes)
            self.data = "simple_graph %s (%d nodes)" % (os.path.basename(graph), len(nodes))
        else:
            # 1024 DISTINCT scans: 16 synthetic worlds x a closed lap of 64 sweeps each (rendered and filtered on the GPU);
            # a 10 k-frame Oxford sequence holds ~3 000 keyframes, so a candidate batch must not revisit 40 scans
            n_worlds, lap = 16, 64
            gt_list, self.scans = [], []
            for wd in range(n_worlds):
                sc = synth.Scene(3000 + wd, circle_frames=lap)
                imgs = synth.render_frames_torch(sc, list(range(lap)), D.dev)
                torch.cuda.synchronize()                   # (the library's stream is not torch's)
                r = api.filter_kstrongest(imgs, 40, 60, 0.0438, 2.5, ctx=ctx)
                ctx.synchronize()
                xyzi, npts = r["xyzi"].cpu().numpy(), r["n_points"].cpu().numpy()
                for f in range(lap):
                    gt_list.append(sc.pose_at(f, lap))
                    self.scans.append(api.MapPointNormal(xyzi[f, :int(npts[f])], 3.0, (0, 0), True, ctx=ctx))
                del imgs, r
            self.gt = np.stack(gt_list)
            self.lap = lap
            self.data = "synthetic (scene_v1)"
        self.n_frames = len(self.scans)
        self.table = api.ScanTable(self.scans, ctx=ctx)
        self.reg = api.n_scan_normal_reg("P2L", ctx=ctx)
        self.reg.SetParameters(4, 10)
        self.comm = None
        if D.dist is not None:                                       # the host's own communicator (as a C++ node would hold one)
            def exchange(raw):
                t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(D.dev)
                D.dist.broadcast(t, 0)
                return bytes(t.cpu().numpy())
            try:
                self.comm = api.RcclComm(ctx, D.world, D.rank, exchange)
            except Exception as e:
                if D.world > 1:
                    raise
                print("bench.py: no 1-rank ncclComm_t (%s): the step runs without the collective" % e, file=sys.stderr)

    def candidates(self, n, seed=11):
        from tbv_slam_public_amd import api
        rng = np.random.Generator(np.random.PCG64(seed))
        base = rng.integers(0, self.n_frames // self.lap, n) * self.lap
        i = base + rng.integers(0, self.lap - 7, n)
        j = i + rng.integers(2, 7, n)
        a, b = self.gt[i], self.gt[j]
        c, s_ = np.cos(a[:, 2]), np.sin(a[:, 2])
        d = b[:, :2] - a[:, :2]
        rel = np.stack([c * d[:, 0] + s_ * d[:, 1], -s_ * d[:, 0] + c * d[:, 1], b[:, 2] - a[:, 2]], axis=1)
        guess = rel + np.concatenate([rng.normal(0, 1.0, (n, 2)), rng.normal(0, np.deg2rad(3.0), (n, 1))], axis=1)
        return api.ScanTable.candidates(i, j, guess)

    def close(self):
        if self.comm is not None:
            self.comm.close()


def pipe_measure(D, W, cands_all, steps, warmup, graph=False, depth=3):
    """One sharded candidate batch through the C-ABI pipe (cfear_candidate_pipe: expand on the preparation stream, matcher on the
    context's stream, ncclAllGather of the 72-byte records + read-back on the exchange stream), every rank handing in the full
    list: the step's latency (submit, collect, repeat), the same steps pipelined (step k + 1 submitted before step k is
    collected), the matcher's own time (hipEvents, a pass of its own) and -- from a second pipe with events around the exchange,
    which cost a few microseconds per step -- the exchange stream's time per step.  graph: replay the matcher launches from a
    captured hipGraph (measured slower than direct launches on ROCm 7.0: 0.145 against 0.138 ms per pipelined 512-block)."""
    from tbv_slam_public_amd import api, _lib as L
    n = int(cands_all.shape[0])
    ctx = W.ctx
    pipe = api.CandidatePipe(W.reg, W.table, n, W.comm, D.rank, D.world, depth=depth, graph=graph, timing=False)
    out = np.empty(n, L.RESULT_DTYPE)
    for _ in range(max(warmup, 1) + 2 * depth + 10):                   # (a communicator's first collectives set its channels up)
        pipe.collect(pipe.submit(cands_all), out)
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.collect(pipe.submit(cands_all), out)
    D.barrier()
    lat_ms = D.max_over_ranks(time.perf_counter() - t0) / steps * 1e3
    def pipelined(k):
        tks = [pipe.submit(cands_all) for _ in range(depth - 1)]
        for _ in range(k - (depth - 1)):
            tks.append(pipe.submit(cands_all))
            pipe.collect(tks.pop(0), out)
        while tks:
            pipe.collect(tks.pop(0), out)
    # The first overlapped steps of a process pay for the runtime's lazily made hardware queues (tens of milliseconds, once):
    # blocks of `steps` pipelined steps are repeated until two in a row agree within 5 % (at most six); the last one counts.
    # (every rank sees the same maxima, so every rank runs the same number of blocks)
    pl_ms, pl_blocks = None, 0
    for _ in range(6):
        D.barrier()
        t0 = time.perf_counter()
        pipelined(steps)
        D.barrier()
        cur = D.max_over_ranks(time.perf_counter() - t0) / steps * 1e3
        pl_blocks += 1
        settled = pl_ms is not None and abs(cur - pl_ms) <= 0.05 * pl_ms
        pl_ms = cur
        if settled:
            break
    graph_slots = pipe.stats()["graph_slots"]
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    for _ in range(steps):
        pipe.collect(pipe.submit(cands_all), out)
    D.barrier()
    prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
    kernel_ms = D.max_over_ranks(sum(v[0] for v in prof.values()) / max(steps, 1))
    pipe.close()
    tpipe = api.CandidatePipe(W.reg, W.table, n, W.comm, D.rank, D.world, depth=depth, graph=False, timing=True)
    for _ in range(4):
        tpipe.collect(tpipe.submit(cands_all), out)
    st0 = tpipe.stats()
    for _ in range(steps):
        tpipe.collect(tpipe.submit(cands_all), out)
    st = tpipe.stats()
    gather_ms = D.max_over_ranks((st["exchange_ms"] - st0["exchange_ms"]) / max(st["steps"] - st0["steps"], 1))
    tpipe.close()
    return {"candidates": n, "candidates_per_rank": (n + D.world - 1) // D.world, "step_ms": lat_ms, "kernel_ms": kernel_ms,
            "gather_ms": gather_ms, "value": n / (lat_ms * 1e-3), "pipelined_step_ms": pl_ms,
            "pipelined_value": n / (pl_ms * 1e-3), "pipelined_over_kernel": pl_ms / kernel_ms if kernel_ms > 0 else None,
            "pipelined_blocks_timed": pl_blocks, "graph_slots": graph_slots,
            "depth": depth, "ok_fraction": float((out["status"] == 0).mean()), "mean_outer_iters": float(out["outer_iters"].mean())}


def _on_side_stream(D, fn):
    """fn() under a torch stream of its own: torch reports handle 0 for its default stream, and a context given 0 makes a PRIVATE
    stream -- on a real (non-default) stream the context, torch's collectives and the read-back share ONE HIP stream
    (dist.py checks the handles itself: Context.shares_torch_stream)."""
    import torch
    side = torch.cuda.Stream(device=D.dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = fn()
    torch.cuda.current_stream().wait_stream(side)
    return out


def loopclosure_run(D, n_cand, steps, warmup, graph=None, world_obj=None):
    return _on_side_stream(D, lambda: _loopclosure_run(D, n_cand, steps, warmup, graph, world_obj))


def _loopclosure_run(D, n_cand, steps, warmup, graph=None, world_obj=None):
    """BASELINE configs[3]: loop-closure candidate registrations between cached surface-point sets, block-sharded over the
    ranks, results all_gathered (RCCL) in candidate order.  The step goes through the C-ABI pipe (what a C++ host calls);
    the Python mirror of the same step (dist.py over torch.distributed) is timed beside it."""
    from tbv_slam_public_amd import api
    from tbv_slam_public_amd import dist as cdist
    W = world_obj or LoopClosureWorld(D, graph)
    cands = W.candidates(n_cand)
    full = pipe_measure(D, W, cands, steps, warmup)
    # what a step costs whatever the batch: ONE candidate per rank through the same code (upload, launches, the collective,
    # read-back) -- a strong-scaling curve is read as  step(N) ~ fixed_cost_ms + kernel_ms(1) / N; the lone registration's own
    # latency (a chain of dependent phases, ~0.07 ms whatever the batch) is not overhead
    tiny = pipe_measure(D, W, cands[:D.world], steps, warmup)
    # What a rank of an 8-GPU run does per step, measured HERE (no 8-GPU node needed to read a future SCALE record against it):
    # the block of n / 8 candidates rank 0 would own, through the same code (the 8-rank all_gather moves 8 x 36 KiB over xGMI
    # instead of 36 KiB inside one GPU, a few microseconds more)
    proj = None
    if D.world == 1 and n_cand >= 64:
        per8 = (n_cand + 7) // 8
        blk = pipe_measure(D, W, cands[:per8], steps, warmup)
        proj = {"ranks": 8, "candidates_per_rank": per8, "step_ms_per_rank_block": blk["step_ms"],
                "kernel_ms_per_rank_block": blk["kernel_ms"], "gather_ms_per_rank_block": blk["gather_ms"],
                "pipelined_step_ms_per_rank_block": blk["pipelined_step_ms"],
                "pipelined_over_kernel": blk["pipelined_over_kernel"],
                "step_ms_one_rank_all_candidates": full["step_ms"], "projected_speedup_at_8": full["step_ms"] / blk["step_ms"],
                "projected_efficiency_at_8": full["step_ms"] / blk["step_ms"] / 8.0,
                "projected_pipelined_efficiency_at_8": full["pipelined_step_ms"] / blk["pipelined_step_ms"] / 8.0,
                "note": "measured at N = 1: the step of one rank's block (n / 8 candidates) through the same path incl. a one-rank "
                        "ncclAllGather; an 8-rank step costs this plus the wider gather (8 x %d KiB over xGMI)" % (per8 * 72 // 1024)}
    # the Python mirror (tests, this bench's verify workload): dist.register_candidates_sharded over torch.distributed
    lo, hi, _per = cdist.shard_range(n_cand, D.world, D.rank)
    mine = cands[lo:hi]
    fn = lambda _local: W.reg.RegisterCandidates(W.table, mine)
    fn.into = lambda _local, ptr: W.reg.RegisterCandidates(W.table, mine, device_ptr=ptr)   # records stay on the GPU until the gather
    fn.ctx = W.ctx
    jobs = [None] * n_cand                                             # (the mirror takes the list for its length only)
    for _ in range(3):
        out = cdist.register_candidates_sharded(jobs, fn)
    D.barrier()
    t1 = time.perf_counter()
    for _ in range(steps):
        out = cdist.register_candidates_sharded(jobs, fn)
    D.barrier()
    py_ms = D.max_over_ranks(time.perf_counter() - t1) / steps * 1e3
    if world_obj is None:
        W.close()
    return {"fixed_cost_ms": tiny["step_ms"], "fixed_cost_kernel_ms": tiny["kernel_ms"],
            "fixed_overhead_ms": max(tiny["step_ms"] - tiny["kernel_ms"], 0.0),
            "kernel_ms": full["kernel_ms"], "gather_ms": full["gather_ms"], "pipelined_ms_per_step": full["pipelined_step_ms"],
            "pipelined_value": full["pipelined_value"], "graph_slots": full["graph_slots"],
            "projected_strong_scaling": proj,
            "collective": "ncclAllGather on the pipe's exchange stream (cfear_candidate_pipe)" if W.comm is not None else "none (no communicator)",
            "python_mirror_ms_per_step": py_ms, "python_mirror_shares_torch_stream": bool(W.ctx.shares_torch_stream()),
            "distinct_scans": W.n_frames, "metric": "loop-closure candidate registrations/sec (cached features, P2L 4x10)",
            "value": full["value"], "unit": "registrations/s", "n_gpus": D.world, "steps": steps,
            "warmup": max(warmup, 1), "ms_per_step": full["step_ms"], "ms_per_4096": full["step_ms"] * 4096.0 / n_cand,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 search / f64 solve",
            "data": W.data,
            "config": {"workload": "configs[3]: %d loop-closure candidates sharded over %d rank(s), all_gather of "
                                   "72-byte result records" % (n_cand, D.world), "candidates": n_cand},
            "ok_fraction": full["ok_fraction"], "mean_outer_iters": full["mean_outer_iters"],
            "python_mirror_ok_fraction": float((out["status"] == 0).mean()),
            "reference_cpu_ms_per_candidate": "8.3-9.7 (evaluation/data/oxford_all_tbv_model_8/job_*/time_statistics.txt)"}


class VerifyWorld:
    """40 nodes of one synthetic lap with surface points and peak clouds on the GPU; candidates 3 per query node."""

    def __init__(self, D):
        import torch
        from tbv_slam_public_amd import api, synth
        self.ctx = ctx = api.Context(D.local_rank, stream=torch.cuda.current_stream().cuda_stream)
        self.n_frames = n_frames = 40
        sc = synth.Scene(3)
        self.gt = np.stack([sc.pose_at(f, n_frames) for f in range(n_frames)])
        self.scans, self.peaks = [], []
        for f in range(n_frames):
            r = api.filter_kstrongest(sc.render(f, n_frames), 40, 60, 0.0438, 2.5, want_peaks=True, ctx=ctx)
            self.scans.append(api.MapPointNormal(r["xyzi"][0, :int(r["n_points"][0])], 3.0, (0, 0), True, ctx=ctx))
            self.peaks.append(torch.from_numpy(np.ascontiguousarray(r["xyzi_peaks"][0, :int(r["n_peaks"][0])])).cuda())   # device-resident
        self.par = api.verify_params(ctx)

    def candidates(self, n, lo=0, hi=None):
        """candidates lo .. hi-1 of the seeded list of n (every rank draws the same list and keeps its block)"""
        rng = np.random.Generator(np.random.PCG64(11))
        hi = n if hi is None else hi
        i = rng.integers(0, self.n_frames - 7, n)
        j = i + rng.integers(2, 7, n)
        noise = np.concatenate([rng.normal(0, 1.0, (n, 2)), rng.normal(0, np.deg2rad(3.0), (n, 1))], axis=1)
        sim, ob = rng.uniform(0.05, 0.5, n), rng.uniform(0, 0.3, n)
        cands = []
        for q in range(lo, hi):
            a, b = self.gt[j[q]], self.gt[i[q]]
            c, s_ = np.cos(a[2]), np.sin(a[2])
            d = b[:2] - a[:2]
            guess = np.array([c * d[0] + s_ * d[1], -s_ * d[0] + c * d[1], b[2] - a[2]]) + noise[q]
            cands.append(dict(from_scan=self.scans[j[q]], to_scan=self.scans[i[q]], from_peaks=self.peaks[j[q]], to_peaks=self.peaks[i[q]],
                              from_pose=self.gt[j[q]], t_be_guess=guess, sc_sim=float(sim[q]), odom_bounds=float(ob[q]), group=q // 3))
        return cands


def verify_measure(D, V, n_cand, steps, warmup):
    """n_cand candidates verified end to end per step, block-sharded over the ranks, one all_gather of 480-byte records
    (dist.verify_candidates_sharded over torch.distributed), ApplyConstratins over the gathered list."""
    from tbv_slam_public_amd import api
    from tbv_slam_public_amd import dist as cdist
    lo, hi, _per = cdist.shard_range(n_cand, D.world, D.rank)
    prepared = api.prepare_verify_batch(V.candidates(n_cand, lo, hi))
    groups = [{"group": q // 3} for q in range(n_cand)]                # (the gather step reads the list's length and groups)
    par = V.par
    fn = lambda _local: api.verify_loop_candidates(prepared, par, V.ctx)
    fn.into = lambda _local, ptr: api.verify_loop_candidates(prepared, par, V.ctx, device_ptr=ptr)   # records stay on the GPU until the gather
    gids = np.arange(n_cand, dtype=np.int32) // 3
    fn.select = lambda out: api.verify_apply_constraints(out, gids, par)
    for _ in range(max(warmup, 1)):
        out = cdist.verify_candidates_sharded(groups, fn, par.model_threshold, bool(par.all_candidates), fn_selects=True)
    D.barrier()
    t1 = time.perf_counter()
    for _ in range(steps):
        out = cdist.verify_candidates_sharded(groups, fn, par.model_threshold, bool(par.all_candidates), fn_selects=True)
    D.barrier()
    ms = D.max_over_ranks(time.perf_counter() - t1) / steps * 1e3
    return {"candidates": n_cand, "candidates_per_rank": (n_cand + D.world - 1) // D.world, "step_ms": ms, "value": n_cand / (ms * 1e-3),
            "reg_ok_fraction": float(out["reg_ok"].mean()), "accepted_fraction": float(out["accepted"].mean())}


def verify_run(D, n_cand, steps, warmup):
    """Full loop-candidate verification per step (RegisterLoopCandidate + VerifyLoopCandidate + ApplyConstratins,
    tbv_slam/src/tbv_slam/loopclosure.cpp:320-384, 261-274): registration, CorAl and CFEAR alignment quality, both
    classifiers; 3 candidates per query node; block-sharded over the ranks, one all_gather of 480-byte records."""
    def go():
        m = verify_measure(D, VerifyWorld(D), n_cand, steps, warmup)
        return {"metric": "loop-closure candidate verifications/sec (register P2L 4x10 + CorAl + CFEAR quality + classifiers)",
                "value": m["value"], "unit": "candidates/s", "n_gpus": D.world, "steps": steps,
                "warmup": max(warmup, 1), "ms_per_step": m["step_ms"], "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32 search / f64 solve + entropy", "data": "synthetic (scene_v1)",
                "config": {"workload": "%d loop-closure candidates (3 per query) verified end to end, sharded over %d rank(s), "
                                       "all_gather of 480-byte records" % (n_cand, D.world), "candidates": n_cand},
                "reg_ok_fraction": m["reg_ok_fraction"], "accepted_fraction": m["accepted_fraction"],
                "reference_cpu_ms_per_candidate": "8.3-9.7 Register + 20-22 VerifyByAlignment "
                                                  "(evaluation/data/oxford_all_tbv_model_8/job_*/time_statistics.txt)"}
    return _on_side_stream(D, go)


def sharded_section(D, n_cand, steps, warmup, world_obj=None):
    """The workloads north_star SHARDS (loopclosure.cpp:658-721: independent candidates), run by every rank of `bench.py
    --gpus N` behind the odometry replicas, whatever N: strong = configs[3] as written, n_cand candidates block-partitioned
    over the N ranks, one all_gather of 72-byte (verification: 480-byte) records; weak = n_cand candidates PER rank (what a
    chip needs to be full: 512 pairs take 0.13 ms, 4096 take 0.32).  `ranks` is read off the gathered data."""
    def go():
        W = world_obj or LoopClosureWorld(D)
        strong = pipe_measure(D, W, W.candidates(n_cand), steps, warmup)
        weak = strong if D.world == 1 else pipe_measure(D, W, W.candidates(n_cand * D.world), steps, warmup)
        V = VerifyWorld(D)
        vs = verify_measure(D, V, n_cand, max(steps // 2, 2), 2)
        vw = vs if D.world == 1 else verify_measure(D, V, n_cand * D.world, max(steps // 2, 2), 2)
        ranks = len(D.gather(D.rank))
        if world_obj is None:
            W.close()
        return {"loopclosure_sharded": {"ranks": ranks, "strong": strong, "weak": weak,
                                        "path": "cfear_candidate_pipe (C-ABI): ncclAllGather of 72-byte records on the pipe's exchange "
                                                "stream" if W.comm is not None else "cfear_candidate_pipe, no communicator",
                                        "unit": "registrations/s (value = candidates of ALL ranks / step; pipelined_value: two steps in flight)"},
                "verify_sharded": {"ranks": ranks, "strong": vs, "weak": vw,
                                   "path": "dist.verify_candidates_sharded: torch.distributed all_gather of 480-byte records, "
                                           "ApplyConstratins after the gather", "unit": "candidates/s"}}
    return _on_side_stream(D, go)


# ---------------------------------------------------------------------------------------------------------
def dry_run(D, args):
    """Launcher dry run (no GPU): the ranks rendezvous over gloo, rank 0 prints the line's launch-related keys.  The sharded
    step runs too -- partition, padded blocks, ONE all_gather, unpadding (dist.py over gloo) -- with a stand-in for the
    per-rank compute (the product has no CPU path): record i carries its own candidate index and the rank that produced it."""
    from tbv_slam_public_amd import _lib as L
    from tbv_slam_public_amd import dist as cdist
    ranks = D.gather(D.rank)

    def stand_in(n_total):
        lo, hi, _per = cdist.shard_range(n_total, D.world, D.rank)

        def fn(local):
            out = np.zeros(len(local), L.RESULT_DTYPE)
            out["pose"][:, 0] = np.arange(lo, hi)
            out["pose"][:, 1] = D.rank
            return out
        t0 = time.perf_counter()
        out = cdist.register_candidates_sharded([None] * n_total, fn) if D.dist else fn([None] * n_total)
        ms = (time.perf_counter() - t0) * 1e3
        assert (out["pose"][:, 0] == np.arange(n_total)).all(), "gathered records are not in candidate order"
        return {"candidates": n_total, "candidates_per_rank": (n_total + D.world - 1) // D.world, "step_ms": ms,
                "value": n_total / (ms * 1e-3), "ranks_seen": int(np.unique(out["pose"][:, 1]).shape[0])}
    strong, weak = stand_in(args.candidates), stand_in(args.candidates * D.world)
    if D.rank == 0:
        print(json.dumps({"metric": "launcher dry run", "n_gpus": D.world, "rccl_ranks": len(ranks), "ranks": ranks,
                          "per_rank_value": D.gather(0.0), "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
                          "loopclosure_sharded": {"ranks": strong["ranks_seen"], "strong": strong, "weak": weak,
                                                  "path": "dist.register_candidates_sharded over gloo, stand-in compute (dry run)"}}))
    else:
        D.gather(0.0)
    D.close()


def main(argv=None):
    global COLS, IMG
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=32,
                    help="consecutive frames every stream advances per step (a step = one pass of the hot path over "
                         "streams x frames_per_step sweeps)")
    ap.add_argument("--streams", type=int, default=4096,
                    help="independent sequences per GPU, one frame each per frame batch (2048 fills the chip once; 4096 halves the "
                         "share of the uneven last registrations of a batch: +6 %% registrations/s at twice the batch latency)")
    ap.add_argument("--sequences", type=int, default=256, help="distinct synthetic worlds per GPU (streams = worlds x start frames)")
    ap.add_argument("--ring", type=int, default=64, help="frames of the closed circle every world is rendered along")
    ap.add_argument("--cols", type=int, default=3360,
                    help="range bins per azimuth (BASELINE metric: 3360 = MulRan's native width; Oxford's native sweeps are 3768 wide, "
                         "radar_filters.cpp:49-52 -- rows then start off the 16-byte grid); every R*C figure of the line follows it")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of CPU baseline (0 = auto, ~10-30 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the passes after the timed region (dense scenes, single stream, host input, loop closure): "
                         "a kernel trace of the command then holds the headline launches only")
    ap.add_argument("--dense", action="store_true", help="headline pass on the dense-row scenes (scene_dense) instead")
    ap.add_argument("--workload", choices=["odometry", "loopclosure", "verify"], default="odometry",
                    help="odometry = BASELINE configs[1] (the headline metric); loopclosure = configs[3]: "
                         "a batch of candidate registrations from cached features, sharded over the ranks "
                         "with one RCCL all_gather of the result records per step")
    ap.add_argument("--candidates", type=int, default=4096)
    ap.add_argument("--no-sharded", action="store_true",
                    help="odometry: skip the sharded loop-closure / verification section behind the replicas (a kernel trace of the "
                         "command then holds the odometry launches only)")
    ap.add_argument("--graph", default=None, help="loopclosure: take the nodes' cached surface points from this simple_graph.sgh "
                                                  "(tools/make_graph.py writes one) instead of featurising synthetic sweeps")
    ap.add_argument("--bins-major", action="store_true",
                    help="feed the images as [range bins][azimuths] (non-Oxford drivers): the decode of radarDriver::Callback "
                         "(radar_driver.cpp:74-90) becomes part of every step -- fused into the filter stage (kstrong_image), or "
                         "with --ctx-option FUSED_DECODE=0 the rotation kernel + the row sweep; not the BASELINE layout")
    ap.add_argument("--keep-nodes", action="store_true",
                    help="also keep every frame's compensated peaks cloud (pose-graph nodes, cfear_odometry_get_*); "
                         "off in the BASELINE configuration")
    ap.add_argument("--cov-sampling", action="store_true",
                    help="odometry: also estimate every frame's covariance by cost sampling (27 GetCost per "
                         "registration, odometrykeyframefuser.cpp:203-208; off in the reference's presets)")
    ap.add_argument("--no-profile", action="store_true", help="experiment: no per-kernel hipEvents inside the timed region")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B runs: a context option of include/cfear_hip.h (enum cfear_option without the CFEAR_OPT_ prefix, e.g. "
                         "FUSED_DECODE=0, MATCHER_LDS_KB=52) applied to every context this run creates")
    ap.add_argument("--dry-run", action="store_true", help="launcher test without GPUs: ranks rendezvous over gloo and exit")
    args = ap.parse_args(argv)
    COLS, IMG = int(args.cols), ROWS * int(args.cols)
    maybe_self_launch(args, argv)
    D = Dist(args)
    if args.dry_run:
        return dry_run(D, args)
    if args.ctx_option:
        from tbv_slam_public_amd import api as _api, _lib as _L
        for kv in args.ctx_option:
            name, val = kv.split("=", 1)
            _api.Context.default_options[getattr(_L, "OPT_" + name.upper())] = int(val)
    if args.workload in ("loopclosure", "verify"):
        out = (loopclosure_run(D, args.candidates, args.steps, args.warmup, args.graph) if args.workload == "loopclosure"
               else verify_run(D, args.candidates, args.steps, args.warmup))
        if args.workload == "loopclosure" and not args.graph:
            out.update(sharded_section(D, args.candidates, args.steps, args.warmup))
        vals = D.gather(out["value"])
        if D.rank == 0:
            out["rccl_ranks"] = len(vals)
        return emit(D, out)

    import torch
    from tbv_slam_public_amd import api
    dev = D.dev

    def note(msg):                                              # progress on stderr (stdout carries the JSON line only)
        if D.rank == 0:
            sys.stderr.write("bench.py: %s\n" % msg)
            sys.stderr.flush()
    B, K, W, FPS = args.streams, args.steps, max(args.warmup, 1), max(args.frames_per_step, 1)
    S, F = min(args.sequences, B), args.ring
    # ---- synthetic input, resident in HBM ------------------------------------------------------------
    t0 = time.time()
    rings = make_rings(S, F, 100000 * D.rank + (50000 if args.dense else 0), dev, dense=args.dense)
    ss = StreamSet(B, S, F)
    t_gen = time.time() - t0
    stream = torch.cuda.current_stream().cuda_stream
    ctx = api.Context(D.local_rank, stream=stream)

    def new_fuser(n_streams, **kw):
        return api.OdometryKeyframeFuser(n_streams, ROWS, COLS, api.odometry_params(**kw), ctx=ctx)

    n_cmp = min(16, S)                     # streams 0 .. n_cmp-1 start at frame 0 of sequences 0 .. n_cmp-1: compared with the CPU
    if args.bins_major or args.keep_nodes:
        # these variants go through the strided-batch entry: one [B][rows][cols] batch per ring frame
        od = new_fuser(B, estimate_cov_by_sampling=int(args.cov_sampling), rotate_ccw=int(args.bins_major),
                       keep_nodes=int(args.keep_nodes)) if not args.bins_major else \
            api.OdometryKeyframeFuser(B, COLS, ROWS, api.odometry_params(estimate_cov_by_sampling=int(args.cov_sampling),
                                                                         rotate_ccw=1, keep_nodes=int(args.keep_nodes)), ctx=ctx)

        def batch_of(t):
            idx = torch.from_numpy(ss.seq * F + (ss.start + t) % F).to(dev)
            x = rings.view(S * F, ROWS, COLS).index_select(0, idx)
            return torch.rot90(x, -1, dims=(1, 2)).contiguous() if args.bins_major else x
        G = min(F, 16)                     # gathered batches (each is B images: 2.75 GB), walked forth and back
        cache = [batch_of(t) for t in range(G)]
        torch.cuda.synchronize()           # the library's stream is not torch's: the batches must exist before the first frame
        if G < F:
            sys.stderr.write("bench.py: --bins-major/--keep-nodes walk %d gathered frames forth and back (the motion reverses "
                             "at both ends)\n" % G)

        def advance(n, t_first, record=None):
            st = {"points": 0.0, "cells": 0.0, "bad": 0, "frames": 0}
            for t in range(t_first, t_first + n):
                info = od.process(cache[pingpong(t, G)], cache[pingpong(t + 1, G)])
                st["points"] += float(info["n_points"].mean()); st["cells"] += float(info["n_cells"].mean())
                st["bad"] += int((info["reg_status"] != 0).sum()); st["frames"] += 1
            return st
    else:
        od = new_fuser(B, estimate_cov_by_sampling=int(args.cov_sampling))

        def advance(n, t_first, record=None):
            return run_odometry(od, rings, ss, n, t_first, record)

    poses = np.zeros(((W + K) * FPS, n_cmp, 3))
    for w in range(W):
        advance(FPS, w * FPS, (poses[w * FPS:], n_cmp))
    D.barrier()
    # events only around the polar sweep inside the timed region (the roofline's kernel; bracketing every kernel of the frame
    # costs 1.7 % of the rate); the per-kernel breakdown comes from an extra untimed pass below
    ctx.profile_enable(0 if args.no_profile else 2)
    ctx.profile_read(reset=True)
    t1 = time.perf_counter()
    tot = {"points": 0.0, "cells": 0.0, "bad": 0, "frames": 0}
    for s in range(K):
        # The filter of the next frame is enqueued behind each frame's kernels (it needs no state of that frame): the
        # sweep of the first timed frame ran during warm-up and the sweep of the frame after the last one runs inside
        # the timed region, so exactly K * frames_per_step filter sweeps (and as many of every other kernel) are timed.
        st = advance(FPS, (W + s) * FPS, (poses[(W + s) * FPS:], n_cmp))
        for k_ in tot:
            tot[k_] += st[k_]
    D.barrier()
    elapsed_local = time.perf_counter() - t1
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    prof_all, n_all = {}, 0
    if not args.no_profile:                          # untimed: the same frames again with every kernel bracketed
        ctx.profile_enable(1)
        st_all = advance(FPS, (W + K) * FPS)
        D.barrier()
        prof_all, n_all = ctx.profile_read(reset=True), st_all["frames"]
        ctx.profile_enable(False)
    elapsed = D.max_over_ranks(elapsed_local)
    per_rank = D.gather(B * FPS * K / elapsed_local)
    bad_total = sum(D.gather(tot["bad"]))
    # ---- the workloads that SHARD, on every rank, whatever N (the odometry above is replicas only) -------------------
    sharded, lc_world = {}, None
    if not (args.no_sharded or args.no_extras) and D.dist is not None:
        note("sharded: loop closure + verification over %d rank(s)" % D.world)
        try:
            lc_world = _on_side_stream(D, lambda: LoopClosureWorld(D))
            sharded = sharded_section(D, args.candidates, 20, 3, lc_world)
            if D.world > 1:
                lc_world.close()
        except Exception as e:       # (a failure every rank shares -- no librccl, no communicator -- must not cost the headline line)
            sys.stderr.write("bench.py: rank %d: sharded section failed: %r\n" % (D.rank, e))
            sharded, lc_world = {"loopclosure_sharded": {"error": repr(e)}, "verify_sharded": {"error": repr(e)}}, None
    if D.rank != 0:
        return D.close()

    total_units = B * FPS * K * D.world
    value = total_units / elapsed
    nf = tot["points"] / max(tot["frames"], 1)
    roof = roofline_of(prof, B, nf)
    n_launch = max(tot["frames"], 1)
    breakdown = {k: {"ms_per_frame_batch": v[0] / max(n_all, 1), "launches": int(v[1])} for k, v in prof_all.items()}
    # HBM traffic of that kernel from the committed PMC passes (rocprofv3 cannot run inside this process):
    # per-scan FETCH_SIZE x2 + WRITE_SIZE measured by tools/profile.sh, scaled to this launch's batch
    roof["traffic_measured_in_run"] = False       # rocprofv3 cannot attach to this process: the PMC passes are separate runs
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tj = os.path.join(ROOT, "profiles", tag, "pmc_traffic.json")
        if os.path.exists(tj):
            t = json.load(open(tj))
            roof["traffic"] = (t["fetch_bytes_per_scan"] + t["write_bytes_per_scan"]) * B
            roof["traffic_source"] = ("profiles/%s/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of %s, "
                                      "per scan x this launch's batch)" % (tag, t.get("configuration", "the standalone filter call")))
            break
    else:
        roof["traffic"] = None

    out = {
        "metric": "radar scan registrations/sec (400x%d polar)" % COLS,
        "value": value,
        "unit": "registrations/s",
        "n_gpus": D.world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8 filter / f32 search / f64 solve",
        "data": "synthetic (%s walls+scatterers rendered on the GPU: %d distinct worlds per GPU, each a ring of %d sweeps along a "
                "closed circle; %d streams per GPU = distinct (world, start frame) pairs; %.1f GB resident, re-read from HBM every "
                "frame)" % ("scene_dense" if args.dense else "scene_v1", S, F, B, S * F * IMG / 1e9),
        "config": {"workload": "configs[1]: per-frame CFEAR-3 odometry registration (k=40, z_min=60, r=3, P2P, "
                               "4-keyframe window%s), polar image -> pose, %d streams in flight per GPU, %d frames per step"
                               % (", + covariance by cost sampling" if args.cov_sampling else "", B, FPS),
                   "streams_per_gpu": B, "frames_per_step": FPS, "sequences_per_gpu": S, "ring_frames": F,
                   "rows": ROWS, "cols": COLS, "parallelism": "replicas x%d" % D.world,
                   "registrations_per_step": B * FPS * D.world},
        "ms_per_frame_batch": elapsed / (K * FPS) * 1e3,
        "timed_region_s": elapsed,
        "rccl_ranks": len(per_rank),
        "per_rank_value": per_rank,
        "roofline": roof,
        "roofline_whole_path": whole_path_of(nf, tot["cells"] / max(tot["frames"], 1), 4, value / D.world),
        "kernel_breakdown": breakdown,
        "kernel_breakdown_note": "hipEvents around every kernel in an extra pass of %d frames AFTER the timed region; inside it only "
                                 "the polar sweep is bracketed (roofline.avg_launch_ms)" % n_all,
        "mean_cells_per_scan": tot["cells"] / max(tot["frames"], 1),
        "failed_registrations": int(bad_total),
        "input_generation_s": t_gen,
    }
    out.update(sharded)
    if D.world > 1:
        return emit(D, out)

    # =====================================================================================================
    # 1 GPU only, after the timed region: extra passes and the CPU baseline
    # =====================================================================================================
    od.close()
    skip = set(filter(None, os.environ.get("BENCH_SKIP", "").split(",")))   # debugging: leave extras out (dense,single,host,mulran)
    if not args.no_extras and not (args.bins_major or args.keep_nodes or args.cov_sampling):
        # ---- dense rows: every azimuth holds >= k bins >= z_min (N_f ~ 16 000, the reference's upper bound) ----
        if not args.dense and "dense" not in skip:
            # as many streams in flight as the headline (4096); the 1024-stream figure of the earlier rounds is kept beside it
            Sd, nfr = 64, 24
            drings = make_rings(Sd, F, 50000, dev, dense=True)
            cand = float((drings[0, :4] >= 60).sum(dim=2).float().mean().item())
            rd = None
            for Bd in (min(B, Sd * F), 1024):
                if rd is not None and Bd >= min(B, Sd * F):
                    break
                dss = StreamSet(Bd, Sd, F)
                odd = new_fuser(Bd)
                run_odometry(odd, drings, dss, 16)               # (until every stream's keyframe window is full: the steady state)
                D.barrier()
                ctx.profile_enable(True); ctx.profile_read(reset=True)
                td = time.perf_counter()
                dst = run_odometry(odd, drings, dss, nfr, 16)
                D.barrier()
                td = time.perf_counter() - td
                dprof = ctx.profile_read(reset=True); ctx.profile_enable(False)
                odd.close()
                if rd is None:
                    rd = roofline_of(dprof, Bd, dst["points"] / dst["frames"])
                    rd.update({"full_path_value": Bd * nfr / td, "full_path_unit": "registrations/s (filter -> pose on the dense scenes)",
                               "ms_per_frame_batch": td / nfr * 1e3, "streams": Bd, "frames": nfr,
                               "mean_candidates_per_row": cand, "mean_cells_per_scan": dst["cells"] / dst["frames"],
                               "failed_registrations": dst["bad"],
                               "kernel_breakdown": {k: v[0] / max(v[1], 1) for k, v in dprof.items()},
                               "data": "scene_dense: %d worlds x ring %d, %d streams" % (Sd, F, Bd)})
                else:
                    rd["at_1024_streams"] = {"full_path_value": Bd * nfr / td, "ms_per_frame_batch": td / nfr * 1e3,
                                             "failed_registrations": dst["bad"],
                                             "kernel_breakdown": {k: v[0] / max(v[1], 1) for k, v in dprof.items()}}
            out["roofline_dense"] = rd
            del drings
        # ---- one sequence alone: the literal configs[1] case (latency-bound) ---------------------------------
        if "single" not in skip:
            od1 = new_fuser(1)
            ss1 = StreamSet(1, 1, F)
            run_odometry(od1, rings, ss1, 8)
            D.barrier()
            n1 = 400
            ts = time.perf_counter()
            st1 = run_odometry(od1, rings, ss1, n1, 8)
            D.barrier()
            ts = time.perf_counter() - ts
            out["single_stream"] = {"value": n1 / ts, "unit": "registrations/s", "ms_per_frame": ts / n1 * 1e3, "frames": n1,
                                    "failed_registrations": st1["bad"],
                                    "note": "one sequence, frame t needs pose t-1: the rate is 1 / latency"}
            od1.close()
        # ---- host-resident input: pinned images, the H2D copy inside the timed region --------------------------
        if "host" not in skip:
            Bh, Gh, nh = 512, 4, 12
            hod = new_fuser(Bh)
            hss = StreamSet(Bh, S, F)
            host = torch.empty((Gh, Bh, ROWS, COLS), dtype=torch.uint8).pin_memory()
            for g in range(Gh):
                idx = torch.from_numpy(hss.seq * F + (hss.start + g) % F).to(dev)
                host[g].copy_(rings.view(S * F, ROWS, COLS).index_select(0, idx))
            torch.cuda.synchronize()
            for g in range(Gh):
                hod.process(host[g], host[(g + 1) % Gh])
            D.barrier()
            th = time.perf_counter()
            for t in range(nh):
                hod.process(host[t % Gh], host[(t + 1) % Gh])
            D.barrier()
            th = time.perf_counter() - th
            out["host_input"] = {"value": Bh * nh / th, "unit": "registrations/s", "ms_per_frame_batch": th / nh * 1e3, "streams": Bh,
                                 "GBs_over_pcie": Bh * nh * IMG / th / 1e9,
                                 "note": "images in pinned host memory, one H2D copy per frame batch inside the timed region "
                                         "(PCIe-bound; never `value`); a ring of %d frame batches cycles" % Gh}
            hod.close()
            del host
        # ---- the other sensor setups of BASELINE.json, each on its own synthetic worlds ----------------------------
        # configs[2]: MulRan (sweeps arrive [range bins][azimuths], 0.0595 m bins, counter-clockwise, 5-keyframe window);
        # configs[4]: CA-CFAR on the Kvarntorp setup (0.175 m bins: 588 m range, ~15 000 detections per sweep)
        def side_config(params, seed0, range_res, bins_major, Bs, Ss, nfr, scene_kw=None):
            from tbv_slam_public_amd import synth
            Fs = F                                               # the whole closed lap: frame Fs continues into frame 0
            sr = torch.empty((Ss, Fs, ROWS, COLS), dtype=torch.uint8, device=dev)
            for q in range(Ss):
                scn = synth.Scene(seed0 + q, circle_frames=F, range_res=range_res, ccw=True, cols=COLS, **(scene_kw or {}))
                sr[q] = synth.render_frames_torch(scn, list(range(Fs)), dev)
            sod = api.OdometryKeyframeFuser(Bs, COLS if bins_major else ROWS, ROWS if bins_major else COLS, params, ctx=ctx)
            seq = torch.arange(Bs, device=dev) % Ss
            start = (torch.arange(Bs, device=dev) // Ss) * 7 % Fs   # streams of one world start at different frames of its lap

            def frame(t):                                        # stream b = world b % Ss at frame (start_b + t) % Fs
                x = sr.view(Ss * Fs, ROWS, COLS).index_select(0, seq * Fs + (start + t) % Fs)
                return torch.rot90(x, -1, dims=(1, 2)).contiguous() if bins_major else x
            batches = [frame(t) for t in range(Fs)]               # every frame of the lap as a gathered batch (Bs x 1.3 MB each)
            del sr
            bad = 0
            torch.cuda.synchronize()                             # the library's stream is not torch's: the batches must exist
            for t in range(4):
                sod.process(batches[t % Fs], batches[(t + 1) % Fs])
            D.barrier()
            ctx.profile_enable(True); ctx.profile_read(reset=True)
            t0s = time.perf_counter()
            pts = cells = 0.0
            for t in range(4, 4 + nfr):
                info = sod.process(batches[t % Fs], batches[(t + 1) % Fs])
                bad += int((info["reg_status"] < 0).sum())
                pts += float(info["n_points"].mean()); cells += float(info["n_cells"].mean())
            D.barrier()
            dts = time.perf_counter() - t0s
            sp = ctx.profile_read(reset=True); ctx.profile_enable(False)
            sod.close()
            del batches
            return {"value": Bs * nfr / dts, "unit": "registrations/s", "ms_per_frame_batch": dts / nfr * 1e3, "streams": Bs,
                    "frames": nfr, "timed_s": dts, "distinct_frames": Ss * Fs,
                    "mean_points_per_scan": pts / nfr, "mean_cells_per_scan": cells / nfr,
                    "failed_registrations": bad, "kernel_breakdown": {k: v[0] / max(v[1], 1) for k, v in sp.items()},
                    "note": "%d worlds x a closed lap of %d sweeps (%d distinct frames), every stream walks its lap" % (Ss, Fs, Ss * Fs)}
        note("extra: config2_mulran")
        if "mulran" not in skip:
            out["config2_mulran"] = side_config(api.odometry_preset("CFEAR-3", "mulran", submap_scan_size=5), 70000, 0.0595238, True, 1024, 32, 320)
        # Kvarntorp / Volvo sweeps reach Process() through radarDriver::Callback's rotate(.., ROTATE_90_COUNTERCLOCKWISE)
        # (radar_driver.cpp:74-90), i.e. they ARRIVE [range bins][azimuths]: that is the layout timed here (the decode fused into
        # the filter: cacfar_cols); the same run on pre-rotated sweeps -- work skipped -- is kept beside it as `prerotated_value`
        c4_par = api.odometry_params(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=20.0, cacfar_nb_guard_cells=10,
                                     cacfar_window_size=40, cacfar_false_alarm_rate=0.01, radar_ccw=1, kstrong_range_res=0.175,
                                     rotate_ccw=1)
        # The worlds of this configuration: the 0.175 m bins reach 588 m, so synth.Scene spreads its default 60 walls over a 1.2 km
        # square and a quarter of the worlds yield fewer surface points than any real scan (SURVEY 8d's realism gate: 184-484 per
        # scan, the 5-95 % band of combined.txt; seed 80002: 12-47).  150 walls + 500 scatterers give ~290 per scan on average
        # (tools/cfar_realism.py) and no failed registration; the sparse worlds stay as a labelled extra (`feature_poor`).
        # A quarter of these sweeps exceed 16 384 detections: they run through the 64-points-per-thread instantiation of surface_sort_mixed_kernel.
        c4_world = dict(n_walls=150, n_scatter=500)
        note("extra: config4_cacfar_kvarntorp ([bins][azimuths] input)")
        c4 = side_config(c4_par, 80000, 0.175, True, 512, 32, 480, c4_world)
        note("extra: config4_cacfar_kvarntorp (pre-rotated input)")
        c4_pre = side_config(api.odometry_params(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=20.0, cacfar_nb_guard_cells=10,
                                                 cacfar_window_size=40, cacfar_false_alarm_rate=0.01, radar_ccw=1,
                                                 kstrong_range_res=0.175), 80000, 0.175, False, 512, 32, 160, c4_world)
        note("extra: config4_cacfar_kvarntorp (feature-poor worlds)")
        c4_poor = side_config(c4_par, 80000, 0.175, True, 512, 32, 96)
        c4["feature_poor"] = {k: c4_poor[k] for k in ("value", "mean_cells_per_scan", "mean_points_per_scan", "failed_registrations", "frames")}
        c4["feature_poor"]["note"] = ("the default 60-wall worlds at this range (round 4's config4 data): below the realism gate; their "
                                      "failed registrations are CFEAR_ERR_TOO_FEW_RESIDUALS (n_scan_normal.cpp:368-369) on worlds with 12-47 "
                                      "surface points per sweep, the CPU oracle fails on the same (world, frame) steps -- "
                                      "tests/test_gpu_odometry.py::test_cacfar_pipeline_kvarntorp_preset")
        c4["input_layout"] = "[range bins][azimuths] (radar_driver.cpp:74-90): decode fused into the CA-CFAR sweep"
        c4["prerotated_value"] = c4_pre["value"]
        c4["prerotated_kernel_breakdown"] = c4_pre["kernel_breakdown"]
        # the CA-CFAR kernels read only the bins their arithmetic can reach (400 m cap, radar_driver.cpp:54: bin 2286 + guard
        # + window of 3360); both the R*C figure and the bytes really requested are given
        need_cols = min(COLS, (int(np.ceil(400.0 / 0.175)) + 10 + 40 + 15) // 16 * 16)
        for kname, brk in (("cacfar_cols", c4["kernel_breakdown"]), ("cacfar_rows", c4_pre["kernel_breakdown"])):
            cr_ms = brk.get(kname, 0.0)
            if cr_ms > 0:
                c4["roofline_" + kname] = {"bound": "hbm", "avg_launch_ms": cr_ms, "scans_per_launch": 512,
                                           "achieved": IMG * 512 / (cr_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": IMG * 512 / (cr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "bytes_requested_per_scan": ROWS * need_cols,
                                           "frac_of_bytes_requested": ROWS * need_cols * 512 / (cr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out["config4_cacfar_kvarntorp"] = c4
        # ---- loop-closure candidates from cached features (configs[3]) ---------------------------------------
        note("extra: loopclosure")
        lc = loopclosure_run(D, args.candidates, 20, 3, world_obj=lc_world)
        out["loopclosure"] = {k: lc[k] for k in ("value", "unit", "ms_per_step", "ms_per_4096", "ok_fraction", "mean_outer_iters",
                                                 "fixed_cost_ms", "fixed_cost_kernel_ms", "fixed_overhead_ms", "kernel_ms", "gather_ms",
                                                 "pipelined_ms_per_step", "pipelined_value", "graph_slots", "collective",
                                                 "python_mirror_ms_per_step", "distinct_scans", "projected_strong_scaling")}
        out["loopclosure"]["candidates"] = args.candidates
    if lc_world is not None:
        lc_world.close()

    # ---- CPU baseline: the oracle (port), 1 thread, bounded sample of the same frames -----------------
    if not args.no_cpu_baseline and not (args.bins_major or args.keep_nodes):
        from oracle import pyoracle as O
        n_frames_cpu = min((W + K) * FPS, max(8, (args.cpu_frames or 1600) // n_cmp))
        host_rings = rings[:n_cmp].cpu().numpy()                   # sequences 0 .. n_cmp-1, streams 0 .. n_cmp-1 start at frame 0
        reg = O.reg_params(cost="P2P", loss="Huber", loss_limit=0.1, weight_opt=4, regularization=0.0)
        done, tc0 = 0, time.perf_counter()
        err_xy, err_th = 0.0, 0.0
        t_filter, stage1, stage1_frames = 0.0, {}, 0
        for sd in range(n_cmp):
            fz = O.Fuser(reg, res=3.0, submap_scan_size=4, weight_intensity=True)
            for f in range(n_frames_cpu):
                tf0 = time.perf_counter()
                sr, si, scn = O.kstrongest(host_rings[sd, f % F], K_STRONGEST, 60)
                cloud = O.kstrongest_cloud(sr, si, scn, 0.0438, 2.5)
                t_filter += time.perf_counter() - tf0
                pose, oi = fz.process(cloud)
                done += 1
                d = np.abs(poses[f, sd] - pose)
                d[2] = abs((d[2] + np.pi) % (2 * np.pi) - np.pi)
                err_xy, err_th = max(err_xy, d[:2].max()), max(err_th, d[2])
            st_sec, st_n = fz.stage_times()
            for kk, vv in st_sec.items():
                stage1[kk] = stage1.get(kk, 0.0) + vv
            stage1_frames += st_n
        tc = time.perf_counter() - tc0
        stage1["Filtering"] = t_filter
        # per-stage CPU time per frame, named like the reference's `timing` keys (radar_driver.cpp:87, 111:
        # "Filtering"; odometrykeyframefuser.cpp:253-255: "compensate", "build_normals", "register")
        stage_ms_1 = {kk: vv / max(stage1_frames, 1) * 1e3 for kk, vv in stage1.items()}
        out["cpu_baseline"] = {"value": done / tc, "unit": "registrations/s", "cores": 1, "kind": "port",
                               "stage_ms": stage_ms_1,
                               "sample": "%d frames = the first %d frames of %d of this run's streams, full path filter->pose, "
                                         "oracle/liboracle.so g++ -O3, 1 thread, %.1f s; host has %d cores"
                                         % (done, n_frames_cpu, n_cmp, tc, os.cpu_count())}
        out["pose_error_vs_cpu"] = {"max_abs_xy_m": err_xy, "max_abs_theta_rad": err_th, "frames_compared": done}
        # the same oracle on the host's cores, one sequence per thread at a time like the reference's NR_WORKERS
        # processes (one native call per sequence, GIL released; the oracle has no shared state)
        from concurrent.futures import ThreadPoolExecutor
        T = _usable_cpus()
        nseq_mt = 40

        def one_sequence(i):
            fz = O.Fuser(reg, res=3.0, submap_scan_size=4, weight_intensity=True)
            fz.run_sequence(host_rings[i % n_cmp, :nseq_mt], K_STRONGEST, 60, 0.0438, 2.5)   # one native call per sequence
            return nseq_mt, fz.stage_times()[0]
        n_tasks = 3 * T
        tm0 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            res_mt = list(ex.map(one_sequence, range(n_tasks)))
        tm = time.perf_counter() - tm0
        done_mt = sum(r[0] for r in res_mt)
        stage_mt = {kk: sum(r[1][kk] for r in res_mt) / max(done_mt, 1) * 1e3 for kk in res_mt[0][1]}
        out["cpu_baseline_all_threads"] = {
            "value": done_mt / tm, "unit": "registrations/s", "cores": T, "kind": "port",
            "stage_ms": stage_mt,
            "stage_ms_note": "mean wall time per frame inside a worker thread while all %d threads run (shared caches and "
                             "memory bandwidth included)" % T,
            "sample": "%d sequences x %d frames on %d threads (cgroup quota / affinity of this host: %d of %d CPUs), %.1f s"
                      % (n_tasks, nseq_mt, T, T, os.cpu_count() or 0, tm)}
    emit(D, out)


if __name__ == "__main__":
    main()
