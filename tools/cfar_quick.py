import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
B, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 8
dev = "cuda"
rings = torch.empty((B, F, 400, 3360), dtype=torch.uint8, device=dev)
for b in range(B):  # distinct scenes up to 64, then repeats
    if b >= 64: rings[b] = rings[b % 64]; continue
    sc = synth.Scene(500 + b, circle_frames=64, range_res=0.175, ccw=True)
    rings[b] = synth.render_frames_torch(sc, list(range(F)), dev)
par = api.odometry_params(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=20.0, cacfar_nb_guard_cells=10,
                          cacfar_window_size=40, cacfar_false_alarm_rate=0.01, radar_ccw=1, kstrong_range_res=0.175)
od = api.OdometryKeyframeFuser(B, 400, 3360, par)
ctx = od.ctx
ctx.profile_enable(True)
for t in range(F):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = od.process(rings[:, t].contiguous())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("frame", t, "ms", round(dt * 1e3, 2), "points", int(info["n_points"].mean()), int(info["n_points"].max()), "cells", int(info["n_cells"].mean()), "bad", int((info["reg_status"] < 0).sum()))
prof = ctx.profile_read(reset=True)
print({k: (round(v[0] / max(v[1], 1) * 1e3, 3) if isinstance(v, (tuple, list)) else v) for k, v in prof.items()} if isinstance(prof, dict) else prof)
