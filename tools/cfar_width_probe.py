"""cacfar_rows at Oxford's native width (3768 bins: rows the kernel cannot read in aligned 16-byte pieces) against 3760 / 3776."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tbv_slam_public_amd import api, synth

B = 512
ctx = api.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for cols in (3360, 3760, 3768, 3776):
    sc = synth.Scene(0, cols=cols, range_res=0.0438)
    base = torch.from_numpy(np.stack([sc.render(f, 8) for f in range(8)])).cuda()
    imgs = base.repeat(B // 8, 1, 1).contiguous()
    par = api.odometry_params(filter_type=1, cacfar_range_res=0.0438, cacfar_z_min=20.0, cacfar_nb_guard_cells=10, cacfar_window_size=40,
                              cacfar_false_alarm_rate=0.01, kstrong_range_res=0.0438)
    od = api.OdometryKeyframeFuser(B, 400, cols, par, ctx=ctx)
    od.process(imgs)
    torch.cuda.synchronize()
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    for _ in range(6):
        info = od.process(imgs)
    prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
    print("cols %d: cacfar_rows %.4f ms per %d sweeps, points per sweep %.0f" % (cols, prof["cacfar_rows"][0] / prof["cacfar_rows"][1], B, float(info["n_points"].mean())))
    od.close()
