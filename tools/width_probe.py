"""kstrongest_rows (the fused key output of the batched odometry) on 512 sweeps of the SAME world at 3360 / 3768 bins, with the
rows dense and padded to a 16-byte pitch: what Oxford's native width costs and why (bytes, row alignment, ragged tail, candidates)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tbv_slam_public_amd import api, synth

B = 512
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for cols, pitch in ((3360, 3360), (3760, 3760), (3768, 3768), (3768, 3776), (3776, 3776)):
    sc = synth.Scene(0, cols=cols)
    base = torch.from_numpy(np.stack([sc.render(f, 8) for f in range(8)])).cuda()
    buf = torch.zeros((B, 400, pitch), dtype=torch.uint8, device="cuda")
    for b in range(B):
        buf[b, :, :cols] = base[b % 8]
    view = buf[:, :, :cols]                       # rows `pitch` bytes apart
    cand = float((base >= 60).sum(dim=2).float().mean())
    torch.cuda.synchronize()
    api.filter_kstrongest_rowkeys(view, 40, 60, 0.0438, 2.5, ctx=ctx)
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    for _ in range(20):
        keys, cnt = api.filter_kstrongest_rowkeys(view, 40, 60, 0.0438, 2.5, ctx=ctx)
    torch.cuda.synchronize()
    prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
    ms = sum(v[0] for v in prof.values()) / 20
    print("cols %d pitch %d: %.4f ms per %d sweeps = %.2f TB/s of R*C; candidates per row %.1f, kept per sweep %.0f  %s"
          % (cols, pitch, ms, B, B * 400 * cols / (ms * 1e-3) / 1e12, cand, float(cnt[:, :, 0].sum(dim=1).float().mean()), list(prof)))
