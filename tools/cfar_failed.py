"""Which registrations of bench.py's config4 pass (CA-CFAR, Kvarntorp setup) fail, and does the CPU oracle fail on the same
(world, frame)?  Replays the side configuration (32 worlds x 16 frames walked forth and back, 512 streams) and then feeds the
failing world's frames, in the same order, through the oracle's fuser."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
from oracle import pyoracle as O
from bench import pingpong
B, Ss, Fs, F = 512, 32, 16, 64
dev = "cuda"
sr = torch.empty((Ss, Fs, 400, 3360), dtype=torch.uint8, device=dev)
for q in range(Ss):
    scn = synth.Scene(80000 + q, circle_frames=F, range_res=0.175, ccw=True)
    sr[q] = synth.render_frames_torch(scn, list(range(Fs)), dev)
par = api.odometry_params(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=20.0, cacfar_nb_guard_cells=10,
                          cacfar_window_size=40, cacfar_false_alarm_rate=0.01, radar_ccw=1, kstrong_range_res=0.175)
od = api.OdometryKeyframeFuser(B, 400, 3360, par)
seq = torch.arange(B, device=dev) % Ss
batches = [sr[:, t].index_select(0, seq).contiguous() for t in range(Fs)]
torch.cuda.synchronize()
fails = {}
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for t in range(T):
    info = od.process(batches[pingpong(t, Fs)], batches[pingpong(t + 1, Fs)])
    bad = np.nonzero(info["reg_status"] < 0)[0]
    for b in bad:
        fails.setdefault((t, int(b) % Ss), []).append((int(b), int(info["reg_status"][b]), int(info["n_cells"][b])))
print("failing (step, world): streams", {k: (len(v), v[0][1:]) for k, v in fails.items()})
worlds = sorted({w for (_, w) in fails})[:2]
reg = O.reg_params(cost="P2P", loss="Huber", loss_limit=0.1, weight_opt=4, regularization=0.0)
for w in worlds:
    fz = O.Fuser(reg, res=3.0, submap_scan_size=4, weight_intensity=True, radar_ccw=True)
    imgs = sr[w].cpu().numpy()
    ofail = []
    for t in range(T):
        cloud, _ = O.cacfar(imgs[pingpong(t, Fs)], 40, 10, 0.01, 0.175, 20.0, 2.5)
        pose, oi = fz.process(cloud)
        if t > 0 and oi[2] != 1:
            ofail.append(t)
    gfail = sorted(t for (t, ww) in fails if ww == w)
    print("world", w, "GPU fails at steps", gfail, "oracle fails at steps", ofail, "same:", gfail == ofail)
