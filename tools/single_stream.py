"""One sequence alone (the literal configs[1] case): wall time per frame and hipEvent time per kernel -- where the latency
of a frame goes (context option HOST_TIMELINE = 1 adds the host timeline).  Round 3: 0.255 ms per frame of which the kernels take 0.244
(register 0.139, surface_sort 0.061, surface_finish 0.016, surface_prep 0.014, surface_points 0.007, sweep 0.008): the frame
is a chain of single-workgroup latencies, not launch overhead."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
sc = synth.Scene(100000, circle_frames=64)
ring = synth.render_frames_torch(sc, list(range(64)), "cuda")
od = api.OdometryKeyframeFuser(1, 400, 3360, api.odometry_params())
ctx = od.ctx
torch.cuda.synchronize()
for t in range(16): od.process(ring[t % 64:t % 64 + 1], ring[(t + 1) % 64:(t + 1) % 64 + 1])
ctx.profile_enable(True); ctx.profile_read(reset=True)
t0 = time.perf_counter()
N = 512
for t in range(16, 16 + N): od.process(ring[t % 64:t % 64 + 1], ring[(t + 1) % 64:(t + 1) % 64 + 1])
dt = time.perf_counter() - t0
prof = ctx.profile_read(reset=True)
print("ms/frame", dt / N * 1e3, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()}, "sum us", round(sum(v[0] / max(v[1], 1) for v in prof.values()) * 1e3, 1))
ctx.profile_enable(False)
t0 = time.perf_counter()
for t in range(16, 16 + N): od.process(ring[t % 64:t % 64 + 1], ring[(t + 1) % 64:(t + 1) % 64 + 1])
print("ms/frame without events", (time.perf_counter() - t0) / N * 1e3)
