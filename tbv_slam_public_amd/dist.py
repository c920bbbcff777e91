"""Loop-closure candidate batches sharded over the GPUs of one node.

TBV registers every loop-closure candidate independently (tbv_slam/src/tbv_slam/loopclosure.cpp:658-721,
each candidate builds its own n_scan_normal_reg at :56), so a batch shards trivially: contiguous
blocks of ceil(n / world) candidates per rank, no exchange on the data path, and ONE collective at the
end -- an all_gather of fixed 72-byte result records (cfear_reg_result), rank order = candidate order,
so the host can apply its ApplyConstratins-style selection deterministically.  One process per GPU,
torch.distributed backend "nccl" (= RCCL over xGMI on MI355X); the message is 72 B x ceil(n/world) per
rank (36 KiB for 4096 candidates on 8 ranks): latency-bound, far below the per-link ring bound.

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
    return fn


def register_candidates_sharded(jobs, register_fn, group=None):
    """jobs: the FULL candidate list (same on every rank).  Returns the results of all candidates, in
    candidate order, on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return register_fn(jobs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(jobs)
    lo, hi, per = shard_range(n, world, rank)
    local = register_fn(jobs[lo:hi])
    assert local.dtype == L.RESULT_DTYPE and local.shape[0] == hi - lo
    padded = np.zeros(per, L.RESULT_DTYPE)
    padded["status"] = L.ERR_INVALID_ARGUMENT          # padding slots are never returned
    padded[:hi - lo] = local
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    send = torch.from_numpy(padded.view(np.uint8).reshape(-1).copy()).to(dev)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    allr = recv.cpu().numpy().view(L.RESULT_DTYPE).reshape(world, per)
    out = np.concatenate([allr[r, :shard_range(n, world, r)[1] - shard_range(n, world, r)[0]] for r in range(world)])
    assert out.shape[0] == n
    return out
