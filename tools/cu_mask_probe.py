"""Can the polar sweep (HBM-bound, VGPR-heavy) and the matcher (latency-bound, LDS-heavy) share the chip by CU MASKS instead of
by interleaving on the same CUs (tried three times, EXPERIMENTS.md)?  Step 1: how fast is each kernel on a stream restricted
to N compute units (hipExtStreamCreateWithCUMask)?  Prints the sweep's time over 2048 images and a 4096-job matcher batch for
several masks; step 2 (--pair): both at once on complementary masks."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from tbv_slam_public_amd import api, synth

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s.value

def mask_every(frac_num, frac_den, offset=0):          # `frac_num` of every `frac_den` consecutive CU bits
    return {i for i in range(256) if ((i + offset) % frac_den) < frac_num}

torch.cuda.init()
sc = synth.Scene(0)
base = torch.from_numpy(np.stack([sc.render(f, 8) for f in range(8)])).cuda()
B = 2048
imgs = base.repeat((B // 8, 1, 1)).contiguous()
# matcher batch: five-scan windows
lap = 64
scans, gts = [], []
ctx0 = api.default_context()
for wd in range(2):
    s2 = synth.Scene(3000 + wd, circle_frames=lap)
    im = synth.render_frames_torch(s2, list(range(lap)), "cuda")
    torch.cuda.synchronize()
    r = api.filter_kstrongest(im, 40, 60, 0.0438, 2.5)
    ctx0.synchronize()
    xyzi, npts = r["xyzi"].cpu().numpy(), r["n_points"].cpu().numpy()
    for f in range(lap):
        gts.append(s2.pose_at(f, lap)); scans.append((xyzi[f, :int(npts[f])]))
gt = np.stack(gts)
def rel(a, b):
    c, s = np.cos(a[2]), np.sin(a[2]); d = b[:2] - a[:2]
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], b[2] - a[2]])

def setup(stream):
    ctx = api.Context(0, stream=stream)
    maps = [api.MapPointNormal(c, 3.0, (0, 0), True, ctx=ctx) for c in scans]
    rng = np.random.default_rng(1)
    jobs = []
    for _ in range(4096):
        base_i = int(rng.integers(0, 2)) * lap
        i = base_i + int(rng.integers(0, lap - 5)); idx = [i, i + 1, i + 2, i + 3, i + 4]
        T = np.array([rel(gt[i], gt[k]) for k in idx]); T[-1] += np.concatenate([rng.normal(0, 0.1, 2), rng.normal(0, 0.004, 1)])
        jobs.append(([maps[k] for k in idx], T))
    reg = api.n_scan_normal_reg("P2P", "Huber", 0.1, 4, ctx=ctx)
    prepared = reg.PrepareBatch(jobs)
    return ctx, reg, prepared, maps

def time_sweep(ctx, n=6):
    api.filter_kstrongest(imgs, 40, 60, 0.0438, 2.5, ctx=ctx)
    ctx.synchronize(); ctx.profile_enable(True); ctx.profile_read(reset=True)
    for _ in range(n): api.filter_kstrongest(imgs, 40, 60, 0.0438, 2.5, ctx=ctx)
    p = ctx.profile_read(reset=True); ctx.profile_enable(False)
    return p["kstrongest_rows"][0] / p["kstrongest_rows"][1]

def time_reg(ctx, reg, prepared, n=4):
    reg.RegisterBatch(prepared); ctx.synchronize(); ctx.profile_enable(True); ctx.profile_read(reset=True)
    for _ in range(n): reg.RegisterBatch(prepared)
    p = ctx.profile_read(reset=True); ctx.profile_enable(False)
    return sum(v[0] for k, v in p.items() if k.startswith("register")) / n

if "--pair" not in sys.argv:
    for name, bits in (("all 256", set(range(256))), ("first 128", set(range(128))), ("first 64", set(range(64))), ("1 of 2", mask_every(1, 2)), ("1 of 4", mask_every(1, 4)),
                       ("first 96", set(range(96))), ("first 160", set(range(160))), ("first 192", set(range(192))), ("last 192", set(range(64, 256)))):
        st = masked_stream(bits)
        ctx, reg, prepared, maps = setup(st)
        ts = time_sweep(ctx); tr = time_reg(ctx, reg, prepared)
        print("%-10s (%3d CUs): sweep %.3f ms per %d images = %.2f TB/s | matcher %.3f ms per 4096" % (name, len(bits), ts, B, B * 1.344e6 / ts / 1e9, tr), flush=True)
        del reg, prepared, maps; ctx.close()
else:
    # (mask bit i = CU slot i / 8 of XCC i % 8 -- the driver deals the bits round-robin over the 8 XCCs; an XCC left without a bit
    #  runs unrestricted, so "every 4th bit" restricts nothing: the first k bits = k / 8 CUs of EVERY XCC)
    for k in (64, 96, 112, 128, 144):
        num, den = k, 256
        sa, sb = masked_stream(set(range(k))), masked_stream(set(range(k, 256)))
        ctx_s = api.Context(0, stream=sa)
        ctx_m, reg, prepared, maps = setup(sb)
        ts = time_sweep(ctx_s); tr = time_reg(ctx_m, reg, prepared)
        ctx_s.synchronize(); ctx_m.synchronize()
        t0 = time.perf_counter()
        N = 6
        for _ in range(N):
            api.filter_kstrongest(imgs, 40, 60, 0.0438, 2.5, ctx=ctx_s); api.filter_kstrongest(imgs, 40, 60, 0.0438, 2.5, ctx=ctx_s)   # 2 x 2048 images = one 4096-stream sweep
            reg.RegisterBatch(prepared)
        ctx_s.synchronize(); ctx_m.synchronize()
        both = (time.perf_counter() - t0) / N * 1e3
        print("sweep on %d of %d CUs, matcher on the rest: alone sweep %.3f ms (x2 per frame batch) matcher %.3f ms | together %.3f ms per (4096-image sweep + 4096 registrations)" % (num, den, ts, tr, both), flush=True)
        del reg, prepared, maps; ctx_s.close(); ctx_m.close()
