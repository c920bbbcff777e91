"""One MulRan-layout sequence alone ([range bins][azimuths] sweeps): wall time per frame and hipEvent time per kernel with the
fused decode and (context option FUSED_DECODE = 0) with the rotation kernel + row sweep."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
sc = synth.Scene(100000, circle_frames=64, range_res=0.0595238, ccw=True)
ring = torch.rot90(synth.render_frames_torch(sc, list(range(64)), "cuda"), -1, dims=(1, 2)).contiguous()
od = api.OdometryKeyframeFuser(1, 3360, 400, api.odometry_preset("CFEAR-3", "mulran", submap_scan_size=5))
ctx = od.ctx
torch.cuda.synchronize()
for t in range(16): od.process(ring[t % 64:t % 64 + 1], ring[(t + 1) % 64:(t + 1) % 64 + 1])
ctx.profile_enable(True); ctx.profile_read(reset=True)
N = 512
t0 = time.perf_counter()
for t in range(16, 16 + N): od.process(ring[t % 64:t % 64 + 1], ring[(t + 1) % 64:(t + 1) % 64 + 1])
dt = time.perf_counter() - t0
prof = ctx.profile_read(reset=True)
print("ms/frame", dt / N * 1e3, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()})
ctx.profile_enable(False)
t0 = time.perf_counter()
for t in range(16, 16 + N): od.process(ring[t % 64:t % 64 + 1], ring[(t + 1) % 64:(t + 1) % 64 + 1])
print("ms/frame without events", (time.perf_counter() - t0) / N * 1e3)
