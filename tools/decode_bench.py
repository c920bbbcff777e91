"""[range bins][azimuths] sweeps: the fused decode + sweep (kstrongest_cols_kernel) against the two-kernel route
(rotate_ccw_rows_kernel, then kstrongest_rows_kernel) on N distinct MulRan-shaped images resident in HBM.
    python tools/decode_bench.py [N] [--two-pass | --tile | --lists | --image] [--iters K] [--zmin Z] [--dense]
Prints the average time of one pass (hipEvents on the context's stream).  Under rocprofv3 (--kernel-trace --stats, or
--pmc FETCH_SIZE) it is the workload behind profiles/r0N/decode_*.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from tbv_slam_public_amd import api, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 512
    two = "--two-pass" in sys.argv
    tile = "--tile" in sys.argv                      # every tile through the LDS transposition (kstrongest_cols_kernel)
    route = 1 if "--lists" in sys.argv else (2 if "--image" in sys.argv else 0)   # lists in global memory / one workgroup per image, whatever the batch size
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    zmin = int(sys.argv[sys.argv.index("--zmin") + 1]) if "--zmin" in sys.argv else 60
    gen = synth.scene_dense if "--dense" in sys.argv else synth.scene_v1       # --dense: every azimuth holds >= 40 bins >= z_min
    base = np.concatenate([gen(sd, 8, range_res=0.0595238, ccw=True)[0] for sd in range(8)])      # 64 x [400][3360]
    src = torch.from_numpy(np.ascontiguousarray(np.rot90(base, -1, axes=(1, 2)))).cuda()                  # [64][3360][400]
    imgs = src.repeat((n + 63) // 64, 1, 1)[:n].contiguous()
    noise = torch.randint(0, 8, imgs.shape, dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))                             # distinct copies
    imgs = torch.clamp(imgs.to(torch.int16) + noise.to(torch.int16), 0, 255).to(torch.uint8)
    del noise
    ctx = api.default_context()
    run = lambda: api.filter_kstrongest_rowkeys(imgs, 12, zmin, 0.0595238, 2.5, bins_major=True, two_pass=two, tile_sweep=tile, route=route)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        keys, cnt = run()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    gb = n * 3360 * 400 / 1e9
    print("%s: %d images, %.3f ms per pass, %.2f TB/s of image bytes, %d points kept" %
          ("two-pass" if two else "tile sweep" if tile else "global lists" if route == 1 else "image per workgroup" if route == 2 else "fused", n, ms, gb / ms, int(cnt[:, :, 0].sum())))


if __name__ == "__main__":
    main()
