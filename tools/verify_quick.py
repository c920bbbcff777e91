"""cfear_verify_loop_candidates alone (no collective, no Python gather): step time, per-kernel event times, host marshal share."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tbv_slam_public_amd import api


class D:
    world, rank, local_rank, dist, dev = 1, 0, 0, None, torch.device("cuda", 0)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
V = bench.VerifyWorld(D)
prepared = api.prepare_verify_batch(V.candidates(n))
for _ in range(3):
    out = api.verify_loop_candidates(prepared, V.par, V.ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = api.verify_loop_candidates(prepared, V.par, V.ctx)
dt = (time.perf_counter() - t0) / 20 * 1e3
V.ctx.profile_enable(True); V.ctx.profile_read(reset=True)
for _ in range(10):
    out = api.verify_loop_candidates(prepared, V.par, V.ctx)
prof = V.ctx.profile_read(reset=True); V.ctx.profile_enable(False)
print("verify_loop_candidates: %.3f ms per %d candidates (%.3g/s); kernels per step: %s = %.3f ms" % (
    dt, n, n / dt * 1e3, {k: round(v[0] / 10, 4) for k, v in prof.items()}, sum(v[0] for v in prof.values()) / 10))
