"""CPU, world_size 2, gloo: the candidate-batch sharding + result all_gather of dist.py.  The per-rank
compute is injected (here: the CPU oracle) because the product has no CPU path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tbv_slam_public_amd import _lib as L
from tbv_slam_public_amd import dist as cdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_jobs(n):
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    imgs, gt, _ = synth.scene_v1(9, 3)
    cells = []
    for f in range(3):
        sr, si, sc = O.kstrongest(imgs[f], 12, 60)
        cells.append(O.surface_points(O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5), 3.5, 1.0, (0, 0), False))
    rng = np.random.default_rng(5)
    jobs = []
    for i in range(n):
        a, b = rng.choice(3, size=2, replace=False)
        guess = np.array([2.5 * (b - a), 0.0, 0.0]) + rng.normal(0, 0.3, 3) * [1, 1, 0.05]
        jobs.append(([cells[a], cells[b]], np.array([[0, 0, 0], guess])))
    return jobs


def _oracle_fn(jobs):
    from oracle import pyoracle as O
    par = O.reg_params(cost="P2L", max_outer=4, max_inner=10)
    out = np.zeros(len(jobs), L.RESULT_DTYPE)
    for i, (cells, T) in enumerate(jobs):
        ok, p, r = O.register(cells, T, par)
        out[i]["pose"] = p[-1]
        out[i]["score"], out[i]["final_cost"] = r.score, r.final_cost
        out[i]["num_residuals"], out[i]["outer_iters"], out[i]["lm_iters"] = r.num_residuals, r.outer_iters, r.lm_iters
        out[i]["status"] = 0 if ok else L.ERR_SOLVER
    return out


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jobs = _make_jobs(n)
    calls = []

    def fn(local):
        calls.append(len(local))
        return _oracle_fn(local)
    out = cdist.register_candidates_sharded(jobs, fn)
    q.put((rank, calls[0], out.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8])
def test_sharded_candidates_match_serial(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    serial = _oracle_fn(_make_jobs(n))
    per = (n + world - 1) // world
    for rank, ncalls, raw in got:
        out = np.frombuffer(raw, L.RESULT_DTYPE)
        assert ncalls == min(per, n - rank * per)              # each rank computed only its block
        assert out.shape[0] == n
        np.testing.assert_array_equal(out["pose"], serial["pose"])          # candidate order preserved
        np.testing.assert_array_equal(out["outer_iters"], serial["outer_iters"])
        assert (out["status"] == 0).all()


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi, per = cdist.shard_range(n, world, r)
                assert hi - lo <= per
                got += list(range(lo, hi))
            assert got == list(range(n))


def test_single_process_passthrough():
    jobs = _make_jobs(3)
    out = cdist.register_candidates_sharded(jobs, _oracle_fn)
    assert out.shape[0] == 3 and (out["status"] == 0).all()


# ---- full candidate verification: 480-byte records, ApplyConstratins after the gather ----------------------
def _make_cands(n):
    from oracle import pyoracle as O
    from tbv_slam_public_amd import synth
    imgs, gt, _ = synth.scene_v1(9, 3)
    nodes = []
    for f in range(3):
        sr, si, sc = O.kstrongest(imgs[f], 12, 60)
        pk = O.peaks(imgs[f], 12, sr, sc)
        nodes.append(dict(cells=O.surface_points(O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5), 3.5, 1.0, (0, 0), False),
                          peaks=O.kstrongest_cloud(sr, si, sc, 0.0438, 2.5, mask=pk), T=gt[f]))
    rng = np.random.default_rng(6)
    cands = []
    for i in range(n):
        a, b = rng.choice(3, size=2, replace=False)
        t_true = O.xyt_compose(O.xyt_inverse(nodes[a]["T"]), nodes[b]["T"])
        err = rng.normal(0, 0.3, 3) * [1, 1, 0.05] if i % 3 else np.array([15.0, 9.0, 0.8])
        cands.append(dict(frm=nodes[a], to=nodes[b], t_be_guess=t_true + err, sc_sim=0.2, odom_bounds=0.0, group=i // 3))
    return cands


def _oracle_verify_fn(cands):
    from oracle import pyoracle as O
    out = np.zeros(len(cands), L.VERIFY_RESULT_DTYPE)
    for i, c in enumerate(cands):
        r = O.verify_loop_candidate(c["frm"]["cells"], c["frm"]["peaks"], c["frm"]["T"], c["to"]["cells"], c["to"]["peaks"],
                                    c["t_be_guess"], c["sc_sim"], c["odom_bounds"])
        out[i]["t_be"], out[i]["cov"], out[i]["coral"], out[i]["cfear"] = r["t_be"], r["cov"], r["coral"], r["cfear"]
        out[i]["alignment_quality"], out[i]["probability"], out[i]["reg_ok"] = r["alignment_quality"], r["probability"], r["reg_ok"]
    return out


def _verify_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def fn(local):
        calls.append(len(local))
        return _oracle_verify_fn(local)
    out = cdist.verify_candidates_sharded(_make_cands(n), fn, 0.8, False)
    q.put((rank, calls[0], out.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_verification_selects_after_the_gather():
    n, world = 7, 2                                   # groups {0,1,2}, {3,4,5}, {6}: group 1 straddles the rank boundary
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import pyoracle as O
    cands = _make_cands(n)
    serial = cdist.verify_candidates_sharded(cands, _oracle_verify_fn, 0.8, False)        # no process group: passthrough
    groups = [c["group"] for c in cands]
    np.testing.assert_array_equal(serial["accepted"].astype(bool), O.apply_constraints(serial["probability"], groups, 0.8, False))
    assert serial["accepted"].sum() >= 2 and all(serial["accepted"][[i for i in range(n) if groups[i] == g]].sum() <= 1 for g in set(groups))
    for rank, ncalls, raw in got:
        out = np.frombuffer(raw, L.VERIFY_RESULT_DTYPE)
        assert ncalls == (4 if rank == 0 else 3)
        for f in ("t_be", "probability", "accepted", "rank", "coral", "cfear"):
            np.testing.assert_array_equal(out[f], serial[f])


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` must become 2 ranks (torch.distributed.run, one per GPU) without an external launcher;
    the dry run rendezvous over gloo and reports what an all_gather sees."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "7"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["ranks"] == [0.0, 1.0] and out["steps"] == 7
    assert len(out["per_rank_value"]) == 2
    # the sharded workloads are part of the line at every N: strong = 4096 candidates over the ranks, weak = 4096 per rank;
    # `ranks` is read off the gathered records (which rank produced each), not off the launcher's environment
    sh = out["loopclosure_sharded"]
    assert sh["ranks"] == 2
    assert sh["strong"]["candidates"] == 4096 and sh["strong"]["candidates_per_rank"] == 2048
    assert sh["weak"]["candidates"] == 8192 and sh["weak"]["candidates_per_rank"] == 4096


def test_bench_refuses_more_gpus_than_visible():
    """Without 2 GPUs `--gpus 2` must fail loudly instead of measuring one GPU and calling it two."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this host really has 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--streams", "64"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "refusing to run" in r.stderr
    # and a rank count that contradicts --gpus is an error too
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env2)
    assert r.returncode != 0 and "rank(s)" in (r.stderr + r.stdout)


def test_c_abi_gather_with_a_mock_two_rank_callback():
    """cfear_gather_records (the collective step of cfear_register_batch_sharded / cfear_verify_loop_candidates_sharded): a
    C++ host hands in its own all_gather as a callback.  Here a mock 2-rank callback: each "rank" contributes its padded
    block; the library must send exactly its block (zero-padded to ceil(n / world) records) and return all n records in
    candidate order without the padding."""
    import ctypes as C
    lib = L.lib()
    n, world, rec = 7, 2, 72
    records = np.arange(n * rec, dtype=np.uint8).reshape(n, rec) % 251
    lo, hi, per = (C.c_int32(), C.c_int32(), C.c_int32())
    blocks, sent = {}, {}
    import struct
    for r in range(world):
        assert lib.cfear_shard_range(n, world, r, C.byref(lo), C.byref(hi), C.byref(per)) == 0
        assert (lo.value, hi.value, per.value) == cdist.shard_range(n, world, r)
        pad = np.zeros((per.value, rec), np.uint8)
        pad[:hi.value - lo.value] = records[lo.value:hi.value]
        blocks[r] = pad.tobytes() + struct.pack("<ii", 0, hi.value - lo.value)   # + the trailer {rank status, records}
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    for me in range(world):
        def gather(user, send, recv, nbytes, me=me):
            sent[me] = C.string_at(send, nbytes)
            C.memmove(recv, b"".join(sent[me] if r == me else blocks[r] for r in range(world)), nbytes * world)
            return 0
        cb = CB(gather)
        lo_, hi_, _ = cdist.shard_range(n, world, me)
        local = np.ascontiguousarray(records[lo_:hi_])
        out = np.zeros((n, rec), np.uint8)
        rc = lib.cfear_gather_records(local.ctypes.data, n, rec, world, me, C.cast(cb, C.c_void_p), None, out.ctypes.data)
        assert rc == 0
        assert sent[me] == blocks[me]                       # the rank's own block, zero-padded, + its trailer
        np.testing.assert_array_equal(out, records)
    # a peer whose own work failed still takes part; every rank then returns THAT rank's status (nobody hangs, one verdict)
    def gather_peer_failed(user, send, recv, nbytes):
        mine = C.string_at(send, nbytes)
        peer = bytes(nbytes - 8) + struct.pack("<ii", L.ERR_CAPACITY, per.value)
        C.memmove(recv, mine + peer, 2 * nbytes)
        return 0
    cbf = CB(gather_peer_failed)
    out = np.zeros((n, rec), np.uint8)
    lo_, hi_, _ = cdist.shard_range(n, world, 0)
    rc = lib.cfear_gather_records(np.ascontiguousarray(records[lo_:hi_]).ctypes.data, n, rec, world, 0, C.cast(cbf, C.c_void_p), None, out.ctypes.data)
    assert rc == L.ERR_CAPACITY
    np.testing.assert_array_equal(out[lo_:hi_], records[lo_:hi_])   # this rank's own records still arrive
    # a callback given at world 1 is called (a 1-rank communicator runs the same collective as 8)
    calls = []
    def gather_one(user, send, recv, nbytes):
        calls.append(nbytes)
        C.memmove(recv, send, nbytes)
        return 0
    cb1 = CB(gather_one)
    out = np.zeros((n, rec), np.uint8)
    assert lib.cfear_gather_records(records.ctypes.data, n, rec, 1, 0, C.cast(cb1, C.c_void_p), None, out.ctypes.data) == 0
    assert calls == [n * rec + 8]
    np.testing.assert_array_equal(out, records)
    # world 1 needs no callback; a failing callback is an error status
    out = np.zeros((n, rec), np.uint8)
    assert lib.cfear_gather_records(records.ctypes.data, n, rec, 1, 0, None, None, out.ctypes.data) == 0
    np.testing.assert_array_equal(out, records)
    bad = CB(lambda user, send, recv, nbytes: 5)
    assert lib.cfear_gather_records(records[:4].ctypes.data, n, rec, 2, 0, C.cast(bad, C.c_void_p), None, out.ctypes.data) != 0
    assert lib.cfear_shard_range(5, 2, 2, C.byref(lo), C.byref(hi), None) == L.ERR_INVALID_ARGUMENT
