"""Which synthetic worlds give Kvarntorp-preset CA-CFAR scans inside SURVEY 8d's realism gate (184-484 surface points per scan,
the 5-95 % band of the real rows)?  Renders a lap of each candidate world, runs the batched CA-CFAR odometry and prints the cell
counts and failed registrations per world.  python tools/cfar_realism.py [n_walls ...]"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
walls = [int(a) for a in sys.argv[1:]] or [60, 120, 200]
F, S = 64, 16
par = api.odometry_params(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=20.0, cacfar_nb_guard_cells=10,
                          cacfar_window_size=40, cacfar_false_alarm_rate=0.01, radar_ccw=1, kstrong_range_res=0.175)
for nw in walls:
    sr = torch.empty((S, F, 400, 3360), dtype=torch.uint8, device="cuda")
    for q in range(S):
        sr[q] = synth.render_frames_torch(synth.Scene(80000 + q, n_walls=nw, n_scatter=200 * nw // 60, circle_frames=F, range_res=0.175, ccw=True), list(range(F)), "cuda")
    od = api.OdometryKeyframeFuser(S, 400, 3360, par)
    torch.cuda.synchronize()
    cells = np.zeros((F, S)); bad = np.zeros(S, int); pts = 0.0
    for t in range(F):
        info = od.process(sr[:, t].contiguous())
        cells[t] = info["n_cells"]; bad += (info["reg_status"] < 0); pts += float(info["n_points"].mean())
    od.close(); del sr
    print("n_walls", nw, "points/scan %.0f" % (pts / F), "cells mean %.0f" % cells.mean(), "per world min", cells.min(0).astype(int).tolist(),
          "mean", cells.mean(0).astype(int).tolist(), "failed", bad.tolist(), flush=True)
