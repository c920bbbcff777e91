"""Which form of matcher_kernel (wavefronts per registration x LDS per workgroup) serves which batch: kernel time (hipEvents) of
two-scan loop-closure candidates (P2L 4x10) and five-scan odometry registrations (P2P 8x20) at several batch sizes, every form
forced through the context options.  python tools/form_sweep.py [pairs|window]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth
from tbv_slam_public_amd import _lib as L

what = sys.argv[1] if len(sys.argv) > 1 else "both"
ctx = api.default_context()
lap = 64
scans, gts = [], []
for wd in range(4):
    sc = synth.Scene(3000 + wd, circle_frames=lap)
    imgs = synth.render_frames_torch(sc, list(range(lap)), "cuda")
    torch.cuda.synchronize()
    r = api.filter_kstrongest(imgs, 40, 60, 0.0438, 2.5)
    ctx.synchronize()
    xyzi, npts = r["xyzi"].cpu().numpy(), r["n_points"].cpu().numpy()
    for f in range(lap):
        gts.append(sc.pose_at(f, lap))
        scans.append(api.MapPointNormal(xyzi[f, :int(npts[f])], 3.0, (0, 0), True))
gt = np.stack(gts)
print("cells per scan: mean %.0f max %d" % (np.mean([s.GetSize() for s in scans]), max(s.GetSize() for s in scans)))
rng = np.random.default_rng(1)

def rel(a, b):
    c, s = np.cos(a[2]), np.sin(a[2]); d = b[:2] - a[:2]
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], b[2] - a[2]])

def timed(fn, reps=5):
    fn(); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    for _ in range(reps): fn()
    prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
    return sum(v[0] for k, v in prof.items() if k.startswith("register")) / reps

def sweep(name, make_jobs, reg, sizes, forms):
    for n in sizes:
        jobs = make_jobs(n)
        row = []
        for waves, kb in forms:
            ctx.set_option(L.OPT_MATCHER_WAVES, waves); ctx.set_option(L.OPT_MATCHER_LDS_KB, kb)
            try:
                ms = timed(lambda: reg.RegisterBatch(jobs))
                out = reg.RegisterBatch(jobs)
                row.append("%dx%d: %.3f ms%s" % (waves, kb, ms, "" if (out["status"] == 0).all() else " (%d failed)" % int((out["status"] != 0).sum())))
            except Exception as e:
                row.append("%dx%d: %s" % (waves, kb, type(e).__name__))
        ctx.set_option(L.OPT_MATCHER_WAVES, 0); ctx.set_option(L.OPT_MATCHER_LDS_KB, 0)
        print(name, n, " | ".join(row), flush=True)

if what in ("pairs", "both"):
    def pairs(n):
        jobs = []
        for _ in range(n):
            base = int(rng.integers(0, 4)) * lap
            i = base + int(rng.integers(0, lap - 7)); j = i + int(rng.integers(2, 7))
            guess = rel(gt[i], gt[j]) + np.concatenate([rng.normal(0, 1.0, 2), rng.normal(0, np.deg2rad(3.0), 1)])
            jobs.append(([scans[i], scans[j]], np.array([[0.0, 0.0, 0.0], guess])))
        return jobs
    reg = api.n_scan_normal_reg("P2L"); reg.SetParameters(4, 10)
    sweep("pairs", pairs, reg, (64, 256, 512, 1024, 4096),
          ((0, 0), (2, 20), (2, 26), (2, 40), (2, 80), (4, 40), (4, 80), (4, 160), (8, 80), (8, 160)))
if what in ("window", "both"):
    def window(n):
        jobs = []
        for _ in range(n):
            base = int(rng.integers(0, 4)) * lap
            i = base + int(rng.integers(0, lap - 5))
            idx = [i, i + 1, i + 2, i + 3, i + 4]
            T = np.array([rel(gt[i], gt[k]) for k in idx])
            T[-1] += np.concatenate([rng.normal(0, 0.3, 2), rng.normal(0, 0.01, 1)])
            jobs.append(([scans[k] for k in idx], T))
        return jobs
    reg = api.n_scan_normal_reg("P2P", "Huber", 0.1, 4)
    sweep("window", window, reg, (1, 64, 128, 256, 512, 1024, 4096),
          ((0, 0), (4, 40), (4, 52), (4, 80), (4, 160), (8, 80), (8, 160), (16, 160)))
