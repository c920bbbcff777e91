"""Loop-closure candidate batches sharded over the GPUs of one node.

TBV registers every loop-closure candidate independently (tbv_slam/src/tbv_slam/loopclosure.cpp:658-721,
each candidate builds its own n_scan_normal_reg at :56), so a batch shards trivially: contiguous
blocks of ceil(n / world) candidates per rank, no exchange on the data path, and ONE collective at the
end -- an all_gather of fixed 72-byte result records (cfear_reg_result), rank order = candidate order,
so the host can apply its ApplyConstratins-style selection deterministically.  One process per GPU,
torch.distributed backend "nccl" (= RCCL over xGMI on MI355X); the message is 72 B x ceil(n/world) per
rank (36 KiB for 4096 candidates on 8 ranks): latency-bound, far below the per-link ring bound.

Full candidate verification (registration + CorAl + CFEAR quality + classifiers, api.verify_loop_candidates)
shards the same way with 480-byte records; its accept / reject step runs after the gather because the candidates
of one query may straddle a rank boundary.

`register_fn(local_jobs) -> RESULT_DTYPE array` is the per-rank compute; the default runs
libcfear_hip.so on this rank's GPU.  The CPU tests inject another function (there is no CPU path in
the product).
"""
import numpy as np

from . import _lib as L


def shard_range(n, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; every rank gets ceil(n / world) slots, the tail is short."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def default_register_fn(reg):
    def fn(jobs):
        return reg.RegisterBatch(jobs) if jobs else np.zeros(0, L.RESULT_DTYPE)

    def into(jobs, device_ptr):                          # records stay on the GPU (stream-ordered, not synchronised)
        return reg.RegisterBatchInto(jobs, device_ptr) if jobs else 0
    fn.into, fn.ctx = into, reg.ctx
    return fn


def _unpad(allr, n, world):
    out = np.concatenate([allr[r, :shard_range(n, world, r)[1] - shard_range(n, world, r)[0]] for r in range(world)])
    assert out.shape[0] == n
    return out


_BUFFERS = {}


def _buffers(nbytes, world, dev):
    """Send / receive tensors (+ pinned host mirrors on the GPU path) reused from step to step: a loop-closure thread gathers
    the same few hundred KiB every time."""
    import torch
    key = (nbytes, world, str(dev))
    b = _BUFFERS.get(key)
    if b is None:
        pin = dev.type == "cuda"
        b = dict(send_h=torch.zeros(nbytes, dtype=torch.uint8, pin_memory=pin),
                 recv_h=torch.zeros(world * nbytes, dtype=torch.uint8, pin_memory=pin))
        if pin:
            b["send_d"] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            b["recv_d"] = torch.zeros(world * nbytes, dtype=torch.uint8, device=dev)
        if len(_BUFFERS) > 8:
            _BUFFERS.clear()
        _BUFFERS[key] = b
    return b


def _gather_records(local, n, group):
    """all_gather of one fixed-size record per candidate: every rank contributes ceil(n / world) records (its block,
    zero-padded), rank order = candidate order; returns the n real records on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi, per = shard_range(n, world, rank)
    assert local.shape[0] == hi - lo
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    rec = local.dtype.itemsize
    b = _buffers(per * rec, world, dev)
    sh = b["send_h"].numpy()                            # (zeroed when it was made; padding slots are never returned)
    sh[:(hi - lo) * rec] = local.view(np.uint8).reshape(-1)
    if dev.type == "cuda":
        b["send_d"].copy_(b["send_h"], non_blocking=True)
        dist.all_gather_into_tensor(b["recv_d"], b["send_d"], group=group)
        b["recv_h"].copy_(b["recv_d"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    else:
        dist.all_gather_into_tensor(b["recv_h"], b["send_h"], group=group)
    allr = b["recv_h"].numpy().view(local.dtype).reshape(world, per)
    if world == 1:
        return allr[0, :n].copy()
    return _unpad(allr, n, world)


def register_candidates_sharded(jobs, register_fn, group=None):
    """jobs: the FULL candidate list (same on every rank).  Returns the results of all candidates, in
    candidate order, on every rank.  With the RCCL backend and a register_fn that can leave its records on the GPU
    (default_register_fn: .into / .ctx) the block never visits the host: the kernel writes into the send tensor,
    all_gather_into_tensor moves it over xGMI, ONE device-to-host copy returns all n records.  A process group of ONE
    rank still runs the collective (the driver's N = 1 run exercises the same code as N = 8)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return register_fn(jobs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(jobs)
    lo, hi, per = shard_range(n, world, rank)
    if dist.get_backend(group) == "nccl" and hasattr(register_fn, "into"):
        import torch
        rec = L.RESULT_DTYPE.itemsize
        dev = torch.device("cuda", torch.cuda.current_device())
        b = _buffers(per * rec, world, dev)
        send = b["send_d"]                               # (zeroed when it was made; padding slots are never returned)
        got = register_fn.into(jobs[lo:hi], send.data_ptr())
        assert got == hi - lo
        # The collective is enqueued on torch's current stream.  When the context enqueues on that very stream (a context
        # created on a non-default torch stream; checked here, not taken from the caller) it is simply ordered behind the
        # kernels; otherwise the library ran on another stream and torch must wait for the records.
        if not register_fn.ctx.shares_torch_stream():
            register_fn.ctx.synchronize()
        dist.all_gather_into_tensor(b["recv_d"], send, group=group)
        b["recv_h"].copy_(b["recv_d"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        allr = b["recv_h"].numpy().view(L.RESULT_DTYPE).reshape(world, per)
        return allr[0, :n].copy() if world == 1 else _unpad(allr, n, world)
    local = register_fn(jobs[lo:hi])
    assert local.dtype == L.RESULT_DTYPE
    return _gather_records(local, n, group)


def apply_constraints(results, groups, model_threshold=0.8, all_candidates=True):
    """loopclosure::ApplyConstratins (tbv_slam/src/tbv_slam/loopclosure.cpp:261-274) over gathered records: per
    query (group) sort by probability, larger first; accept above the threshold, every candidate or only the best.
    Rewrites results["accepted"] / results["rank"] in place -- a query's candidates may sit on different ranks, so
    the selection the library made inside one rank's block is redone over the whole list."""
    groups = np.asarray(groups)
    if results.shape[0] == 0:
        return results
    prob = results["probability"]
    order = np.lexsort((-prob, groups))                 # stable: by query, then probability descending
    gs = groups[order]
    first = np.concatenate([[True], gs[1:] != gs[:-1]])
    start = np.maximum.accumulate(np.where(first, np.arange(order.shape[0]), 0))
    rank = np.arange(order.shape[0]) - start
    results["rank"][order] = rank
    results["accepted"][order] = ((rank == 0) | bool(all_candidates)) & (prob[order] > model_threshold)
    return results


def verify_candidates_sharded(cands, verify_fn, model_threshold=0.8, all_candidates=True, group=None, fn_selects=False):
    """Loop-candidate verification (api.verify_loop_candidates) for the FULL candidate list `cands` (dicts with a
    "group" key, same on every rank): each rank verifies its contiguous block, one all_gather of the 480-byte
    cfear_verify_result records, then ApplyConstratins over the whole list.  `verify_fn(local_cands) ->
    VERIFY_RESULT_DTYPE array`; fn_selects: verify_fn applies the selection itself with the same threshold / policy
    (api.verify_loop_candidates does), so a single process need not redo it."""
    import torch.distributed as dist
    groups_of = lambda: [int(c.get("group", 0)) for c in cands]   # (only the paths that select here walk the list)
    if not (dist.is_available() and dist.is_initialized()):
        out = verify_fn(cands)                                # one rank sees every candidate of every query: a verify_fn
        if fn_selects:                                        # that already ran ApplyConstratins (the library does) is final
            return out
        return apply_constraints(out, groups_of(), model_threshold, all_candidates)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(cands)
    lo, hi, per = shard_range(n, world, rank)
    if dist.get_backend(group) == "nccl" and hasattr(verify_fn, "into"):
        # the records never visit the host before the gather: the chain's last kernel writes them into the send tensor,
        # all_gather_into_tensor moves them over xGMI, ONE device-to-host copy returns all n -- and the selection runs once, over
        # the whole list (verify_fn.select(records): the library's ApplyConstratins on host records)
        import torch
        rec = L.VERIFY_RESULT_DTYPE.itemsize
        dev = torch.device("cuda", torch.cuda.current_device())
        b = _buffers(per * rec, world, dev)
        got = verify_fn.into(cands[lo:hi], b["send_d"].data_ptr())     # (waits for the chain: its error flag comes back)
        assert got == hi - lo
        dist.all_gather_into_tensor(b["recv_d"], b["send_d"], group=group)
        b["recv_h"].copy_(b["recv_d"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        allr = b["recv_h"].numpy().view(L.VERIFY_RESULT_DTYPE).reshape(world, per)
        out = allr[0, :n].copy() if world == 1 else _unpad(allr, n, world)
        if hasattr(verify_fn, "select"):                      # (a caller that keeps the query ids as an array: no walk over the list)
            return verify_fn.select(out)
        return apply_constraints(out, groups_of(), model_threshold, all_candidates)
    local = verify_fn(cands[lo:hi])
    assert local.dtype == L.VERIFY_RESULT_DTYPE
    out = _gather_records(local, n, group)
    if world == 1 and fn_selects:                             # one rank saw every candidate of every query and has selected
        return out
    return apply_constraints(out, groups_of(), model_threshold, all_candidates)
