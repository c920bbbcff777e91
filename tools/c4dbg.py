"""Debug driver: the bench's config2 (MulRan) and config4 (CA-CFAR, [bins][azimuths] input) side runs, alone or in sequence, on
one context: python tools/c4dbg.py [frames_c4] [frames_mulran] [--torch-stream]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tbv_slam_public_amd import api, synth
dev = "cuda"
ROWS, COLS = 400, 3360
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n4 = int(args[0]) if args else 80
n2 = int(args[1]) if len(args) > 1 else 0
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream) if "--torch-stream" in sys.argv else None

def side(params, seed0, range_res, Bs, Ss, nfr, name):
    Fs = 64
    sr = torch.empty((Ss, Fs, ROWS, COLS), dtype=torch.uint8, device=dev)
    for q in range(Ss):
        scn = synth.Scene(seed0 + q, circle_frames=Fs, range_res=range_res, ccw=True)
        sr[q] = synth.render_frames_torch(scn, list(range(Fs)), dev)
    sod = api.OdometryKeyframeFuser(Bs, COLS, ROWS, params, ctx=ctx)
    seq = torch.arange(Bs, device=dev) % Ss
    start = (torch.arange(Bs, device=dev) // Ss) * 7 % Fs
    batches = [torch.rot90(sr.view(Ss * Fs, ROWS, COLS).index_select(0, seq * Fs + (start + t) % Fs), -1, dims=(1, 2)).contiguous() for t in range(Fs)]
    del sr
    torch.cuda.synchronize()
    for t in range(nfr):
        try:
            info = sod.process(batches[t % Fs], batches[(t + 1) % Fs])
        except Exception as e:
            print(name, "FAIL at frame", t, str(e)[:160]); sys.exit(1)
        if t % 40 == 0: print(name, "frame", t, "pts", int(info["n_points"].mean()), "cells", int(info["n_cells"].mean()), "max cells", int(info["n_cells"].max()), flush=True)
    sod.close()
    print(name, "ok")

if n2:
    side(api.odometry_preset("CFEAR-3", "mulran", submap_scan_size=5), 70000, 0.0595238, 1024, 32, n2, "mulran")
side(api.odometry_params(filter_type=1, cacfar_range_res=0.175, cacfar_z_min=20.0, cacfar_nb_guard_cells=10, cacfar_window_size=40,
                         cacfar_false_alarm_rate=0.01, radar_ccw=1, kstrong_range_res=0.175, rotate_ccw=1), 80000, 0.175, 512, 32, n4, "c4")
