#!/usr/bin/env python
"""Micro-benchmark of stage F alone (kstrongest_rows + kstrong_cloud) on a device-resident batch.
Used for rocprofv3 kernel-trace / PMC passes on the polar sweep (profiles/)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--data", choices=["scene", "dense", "uniform"], default="scene")
    ap.add_argument("--k", type=int, default=40)
    ap.add_argument("--peaks", type=int, default=0)
    args = ap.parse_args()
    import torch
    from tbv_slam_public_amd import api, synth
    if args.data == "scene":
        sc = synth.Scene(0)
        base = np.stack([sc.render(f, 8) for f in range(8)])
    elif args.data == "dense":
        sc = synth.Scene(0, **synth.DENSE_KW)
        base = np.stack([sc.render(f, 8) for f in range(8)])
    else:
        base = synth.uniform_v1(0, batch=8)
    d = torch.from_numpy(base).cuda()
    imgs = torch.empty((args.batch, 400, 3360), dtype=torch.uint8, device="cuda")
    for b in range(args.batch):
        imgs[b] = d[b % 8]
    ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    api.filter_kstrongest(imgs, args.k, 60, 0.0438, 2.5, bool(args.peaks), ctx=ctx)
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        r = api.filter_kstrongest(imgs, args.k, 60, 0.0438, 2.5, bool(args.peaks), ctx=ctx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    prof = ctx.profile_read()
    nf = float(r["n_points"].float().mean())
    for name, (ms, n) in prof.items():
        avg = ms / max(n, 1)
        gbs = args.batch * (400 * 3360) / (avg * 1e-3) / 1e9
        alg = args.batch * (400 * 3360 + 16 * nf + 4 * 400) / (avg * 1e-3) / 1e9      # R*C + 16 N_f + 4 R
        print("%-18s avg %.4f ms  polar-read %.0f GB/s  algorithmic %.0f GB/s = %.3f of 8 TB/s" % (name, avg, gbs, alg, alg / 8000.0))
    print("wall %.4f ms per call, %d images, mean points %.0f" % (dt * 1e3, args.batch, nf))


if __name__ == "__main__":
    main()
