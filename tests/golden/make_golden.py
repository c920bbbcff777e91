#!/usr/bin/env python
"""Generates tests/golden/*.npz from the CPU oracle (the reference ships no golden vectors for this
path and cannot be built here -- SURVEY.md 8c -- so these pin the ORACLE, and through it the HIP path,
against silent drift).  Re-run only when the oracle's semantics are deliberately changed:
    python tests/golden/make_golden.py
Inputs are synthetic (tbv_slam_public_amd/synth.py) and stored alongside the expected outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O                     # noqa: E402
from tbv_slam_public_amd import synth                # noqa: E402


def main():
    # ---- filters on a reduced sweep (64 azimuths x 1200 bins) ----------------------------------------
    sc = synth.Scene(77, rows=64, cols=1200, n_walls=30)
    img = sc.render(0, 1)
    img[5, 300:360] = 255                              # plateau wider than k: tie rule
    img[9, 1194:] = 250                                # kept bins at the far edge (peaks border case)
    sr, si, cnt = O.kstrongest(img, 12, 60)
    pk = O.peaks(img, 12, sr, cnt)
    cloud = O.kstrongest_cloud(sr, si, cnt, 0.0438, 2.5)
    cloud_pk = O.kstrongest_cloud(sr, si, cnt, 0.0438, 2.5, mask=pk)
    cf_cloud, cf_rc = O.cacfar(img, 20, 5, 0.01, 0.0438, 40, 2.5)
    np.savez_compressed(os.path.join(HERE, "filters.npz"), img=img, sel_range=sr, sel_intensity=si, sel_count=cnt,
                        is_peak=pk, cloud=cloud, cloud_peaks=cloud_pk, cfar_cloud=cf_cloud, cfar_rc=cf_rc)
    # ---- surface points + registration on a full-size pair ---------------------------------------------
    imgs, gt, _ = synth.scene_v1(78, 3)
    clouds = []
    for f in range(3):
        a, b, c = O.kstrongest(imgs[f], 40, 60)
        clouds.append(O.kstrongest_cloud(a, b, c, 0.0438, 2.5))
    mot = np.array([2.45, 0.03, 0.011])
    comp = O.compensate(clouds[1], mot, False)
    cells = [O.surface_points(clouds[0], 3.0, 1.0, (0, 0), True), O.surface_points(comp, 3.0, 1.0, (0, 0), True),
             O.surface_points(clouds[2], 3.0, 1.0, (0, 0), True)]
    poses = np.array([[0, 0, 0], [2.3, 0.2, 0.02], [5.2, -0.3, 0.01]])
    out = {}
    for name, kw, mo, mi in [("p2l_4x10", dict(cost="P2L"), 4, 10), ("p2p_w4", dict(cost="P2P", weight_opt=4), 8, 20),
                             ("p2d", dict(cost="P2D"), 8, 20), ("p2l_cauchy", dict(cost="P2L", loss="Cauchy", weight_opt=4), 8, 20)]:
        par = O.reg_params(max_outer=mo, max_inner=mi, **kw)
        ok, p, r = O.register(cells, poses, par)
        pairs, w = O.associate(cells, poses, par, 1)
        okc, cost, res, score = O.get_cost(cells, poses, par)
        out[name + "_pose"] = p[-1]
        out[name + "_meta"] = np.array([ok, r.outer_iters, r.lm_iters, r.num_residuals], np.int64)
        out[name + "_cost"] = np.array([r.final_cost, r.score, cost, score])
        out[name + "_pairs"] = pairs
        out[name + "_weights"] = w
    np.savez_compressed(os.path.join(HERE, "registration.npz"), cloud0=clouds[0], cloud1=clouds[1], cloud2=clouds[2],
                        mot=mot, comp1=comp, cells0=cells[0], cells1=cells[1], cells2=cells[2], poses=poses, **out)
    # ---- verification features: covariance by cost sampling + CorAl / CFEAR alignment quality ---------------
    ver = {}
    for name, kw, mo, mi in [("p2l_4x10", dict(cost="P2L"), 4, 10), ("p2p_w4", dict(cost="P2P", weight_opt=4), 8, 20)]:
        par = O.reg_params(max_outer=mo, max_inner=mi, **kw)
        ok, p, r = O.register(cells, poses, par)
        par.first_itr = r.outer_iters
        cok, cov, smp = O.cov_by_sampling(cells, p, par, r.final_cost, r.num_residuals, 0.4, 0.0043625, 3, 4.0)
        ver[name + "_cov_ok"] = np.array([cok, r.outer_iters, r.num_residuals], np.int64)
        ver[name + "_cov"] = cov
        ver[name + "_samples"] = smp
        ver[name + "_reg"] = np.array([r.final_cost] + list(p[-1]))
    peaks = []
    for f in range(2):
        a, b, c = O.kstrongest(imgs[f], 40, 60)
        peaks.append(O.kstrongest_cloud(a, b, c, 0.0438, 2.5, mask=O.peaks(imgs[f], 40, a, c)))
    offs = np.array([[0, 0, 0], [0.5, 0, 0.0087], [0, -2.0, 0.26]])
    src_pose = np.array([2.45, 0.03, 0.011])
    cq, cv = [], []
    for o in offs:
        okq, q, _ = O.coral_quality(peaks[0], peaks[1], np.zeros(3), src_pose, o, 1.0)
        cq.append(q); cv.append(okq)
    np.savez_compressed(os.path.join(HERE, "verification.npz"), peaks0=peaks[0], peaks1=peaks[1], offsets=offs,
                        src_pose=src_pose, coral_quality=np.array(cq), coral_valid=np.array(cv), **ver)
    for f in ("filters.npz", "registration.npz", "verification.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
