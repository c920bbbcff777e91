# per-kernel times of the k-strongest pipeline on MulRan-like sweeps (range_res 0.0595 m: 200 m range, 134 x 134 voxels)
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
B, F = 256, 6
rings = torch.empty((B, F, 400, 3360), dtype=torch.uint8, device="cuda")
for b in range(B):
    sc = synth.Scene(700 + b % 32, circle_frames=64, range_res=0.0595238, ccw=True)
    rings[b] = synth.render_frames_torch(sc, list(range(F)), "cuda")
od = api.OdometryKeyframeFuser(B, 400, 3360, api.odometry_params(kstrong_range_res=0.0595238, radar_ccw=1))
od.ctx.profile_enable(True)
for t in range(F):
    info = od.process(rings[:, t].contiguous())
print("points", int(info["n_points"].mean()), "cells", int(info["n_cells"].mean()), "bad", int((info["reg_status"] < 0).sum()))
prof = od.ctx.profile_read(reset=True)
print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()})
