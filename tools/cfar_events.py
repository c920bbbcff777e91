"""hipEvent timing of the CA-CFAR pipeline in profile mode 1 (every kernel) and mode 2 (row kernel only), with prefetch like
bench.py; compare with rocprofv3 --kernel-trace of the same command.  --bins-major: the sweeps arrive [range bins][azimuths]
(decode fused into the filter: cacfar_cols_kernel)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
BM = '--bins-major' in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith('--')]
B, F = (int(args[0]) if args else 512), 8
dev = "cuda"
rings = torch.empty((min(B, 64), F, 400, 3360), dtype=torch.uint8, device=dev)
for b in range(rings.shape[0]):
    sc = synth.Scene(500 + b, circle_frames=64, range_res=0.175, ccw=True)
    rings[b] = synth.render_frames_torch(sc, list(range(F)), dev)
seq = torch.arange(B, device=dev) % rings.shape[0]
batches = [rings[:, t].index_select(0, seq).contiguous() for t in range(F)]
if BM:
    batches = [torch.rot90(x, -1, dims=(1, 2)).contiguous() for x in batches]
par = api.odometry_params(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=20.0, cacfar_nb_guard_cells=10,
                          cacfar_window_size=40, cacfar_false_alarm_rate=0.01, radar_ccw=1, kstrong_range_res=0.175, rotate_ccw=int(BM))
od = api.OdometryKeyframeFuser(B, 3360 if BM else 400, 400 if BM else 3360, par)
ctx = od.ctx
torch.cuda.synchronize()
pp = lambda t: (t % (2 * F - 2)) if (t % (2 * F - 2)) < F else 2 * F - 2 - (t % (2 * F - 2))
for mode in (0, 2, 1):
    ctx.profile_enable(mode); ctx.profile_read(reset=True)
    for t in range(4):
        od.process(batches[pp(t)], batches[pp(t + 1)])
    ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 24
    for t in range(4, 4 + N):
        info = od.process(batches[pp(t)], batches[pp(t + 1)])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True)
    print("mode", mode, "ms/frame", round(dt / N * 1e3, 3), {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()},
          "bad", int((info["reg_status"] < 0).sum()), "pts", int(info["n_points"].mean()))
