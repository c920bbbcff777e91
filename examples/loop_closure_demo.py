#!/usr/bin/env python
"""TBV's back end on one synthetic lap, every arithmetic step on the GPU through libcfear_hip.so:

  radar sweeps -> CFEAR-3 odometry (OdometryKeyframeFuser)                      cfear_radarodometry
  graph nodes  -> peaks cloud + oriented surface points per keyframe (RadarScan, types.h:119-122)
  per node     -> radar Scan Context of the local map, candidate retrieval      place_recognition_radar, loopclosure.cpp:552-600
  per candidate-> RegisterLoopCandidate + VerifyLoopCandidate + ApplyConstratins loopclosure.cpp:658-725

The sensor drives a closed circle, so the last nodes revisit the first ones: the demo prints the loop constraints the
verifier accepts and how far their registered transforms are from the ground truth.  `run(backend)` is shared with
tests/test_gpu_tbv_loop.py, which runs it a second time with the CPU oracle behind the same host logic.
    python examples/loop_closure_demo.py [--frames 68]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbv_slam_public_amd import synth          # noqa: E402

RANGE_RES, K, Z_MIN, MIN_DISTANCE, RES = 0.0438, 40, 60, 2.5, 3.0


def circle_scene(seed=21, yaw_rate=0.45):
    """scene_v1 world, but driven on a closed circle (10 m/s, radius 10 / yaw_rate m)."""
    sc = synth.Scene(seed)
    sc._yaw_rates[:] = yaw_rate
    return sc


def xyt_compose(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([c * b[0] - s * b[1] + a[0], s * b[0] + c * b[1] + a[1], a[2] + b[2]])


def xyt_inverse(a):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([-(c * a[0] + s * a[1]), s * a[0] - c * a[1], -a[2]])


def transform_cloud(xyzi, T):
    """pcl::transformPointCloud with an Affine3d: double products, float result."""
    c, s = np.cos(T[2]), np.sin(T[2])
    x, y = xyzi[:, 0].astype(np.float64), xyzi[:, 1].astype(np.float64)
    out = xyzi.copy()
    out[:, 0] = ((c * x + (-s) * y) + 0.0 * xyzi[:, 2]) + T[0]
    out[:, 1] = ((s * x + c * y) + 0.0 * xyzi[:, 2]) + T[1]
    return out


class HipBackend:
    """The product: every call lands in libcfear_hip.so."""

    def __init__(self):
        from tbv_slam_public_amd import api
        self.api = api
        self.driver = api.radarDriver(api.radarDriverParameters(k_strongest=K, z_min=Z_MIN, range_res=RANGE_RES,
                                                                min_distance=MIN_DISTANCE))

    def sequence(self, imgs):
        """Odometry and graph nodes in one pass: with keep_nodes the pipeline hands out every frame's RadarScan
        (surface points + compensated peaks cloud), so nothing is filtered or featurised twice."""
        od = self.api.OdometryKeyframeFuser(1, imgs.shape[1], imgs.shape[2],
                                            self.api.odometry_preset("CFEAR-3", "oxford", keep_nodes=1))
        poses, nodes = [], []
        for f in range(imgs.shape[0]):
            info = od.process(imgs[f:f + 1], imgs[f + 1:f + 2] if f + 1 < imgs.shape[0] else None)
            poses.append(info["pose"][0].copy())
            nd = od.node(0)                                                      # every frame is a keyframe here
            nodes.append(dict(scan=nd["scan"], peaks=nd["peaks"]))
        od.close()
        return np.array(poses), nodes

    def scan_context(self):
        return self.api.RSCManagerNative()          # database in HBM, retrieval policy in the library

    def odom_bounds(self, rel):
        return self.api.VerifyByOdometry(rel)

    def verify(self, nodes, cands):
        jobs = [dict(from_scan=nodes[c["from"]]["scan"], to_scan=nodes[c["to"]]["scan"], from_peaks=nodes[c["from"]]["peaks"],
                     to_peaks=nodes[c["to"]]["peaks"], from_pose=c["from_pose"], t_be_guess=c["t_be_guess"], sc_sim=c["sc_sim"],
                     odom_bounds=c["odom_bounds"], group=c["from"]) for c in cands]
        r = self.api.verify_loop_candidates(jobs)
        return [dict(t_be=r["t_be"][i].copy(), probability=float(r["probability"][i]), accepted=bool(r["accepted"][i]),
                     reg_ok=bool(r["reg_ok"][i]), alignment_quality=float(r["alignment_quality"][i])) for i in range(len(cands))]


def run(backend, n_frames=68, scene=None, log=None):
    """-> dict(poses, gt, candidates [dict], results [dict]) ; candidates[i] / results[i] belong together."""
    sc = scene or circle_scene()
    imgs = np.stack([sc.render(f, n_frames) for f in range(n_frames)])
    gt = np.stack([sc.pose_at(f, n_frames) for f in range(n_frames)])
    gt = np.array([xyt_compose(xyt_inverse(gt[0]), g) for g in gt])             # odometry starts at the identity
    # graph nodes: every frame is a keyframe here (2.5 m between sweeps > 1.5 m)
    poses, nodes = backend.sequence(imgs)
    rsc = backend.scan_context()
    cands = []
    for i in range(n_frames - 1):                                                # the closure thread trails the odometry by one node
        merged = []
        for j in (i - 1, i, i + 1):                                              # ScansToLocalMap, N_aggregate = 1 (loopclosure.cpp:552-570)
            if 0 <= j < n_frames:
                merged.append(transform_cloud(nodes[j]["peaks"], poses[j]))
        local = transform_cloud(np.concatenate(merged), xyt_inverse(poses[i]))
        rsc.makeAndSaveScancontextAndKeysRadarCloud(local, poses[i])
        for c in rsc.detectLoopClosureID():
            to = c["nn_idx"]
            rel = [xyt_compose(xyt_inverse(poses[k]), poses[k + 1]) for k in range(to, i)]
            # Tsrcguess = Taug^-1 * Rz(sc yaw) (loopclosure.cpp:693-697)
            guess = xyt_compose(xyt_inverse(np.asarray(c["Taug"], np.float64)), np.array([0.0, 0.0, c["yaw_diff_rad"]]))
            cands.append({"from": i, "to": to, "from_pose": poses[i], "t_be_guess": guess, "sc_sim": c["min_dist"],
                          "odom_bounds": backend.odom_bounds(np.array(rel).reshape(-1, 3)), "sc_yaw": c["yaw_diff_rad"]})
    results = backend.verify(nodes, cands)
    if log:
        for c, r in zip(cands, results):
            if r["accepted"]:
                true = xyt_compose(xyt_inverse(gt[c["from"]]), gt[c["to"]])
                d = r["t_be"] - true
                d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi
                log("loop %2d -> %2d  p = %.3f  t_be = (%.2f, %.2f, %.3f)  error vs ground truth (%.2f m, %.4f rad)"
                    % (c["from"], c["to"], r["probability"], *r["t_be"], np.hypot(d[0], d[1]), abs(d[2])))
    return dict(poses=poses, gt=gt, candidates=cands, results=results)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=68)
    a = ap.parse_args()
    out = run(HipBackend(), a.frames, log=print)
    acc = sum(r["accepted"] for r in out["results"])
    drift = np.abs(out["poses"][-1] - out["gt"][-1])
    print("%d candidates verified, %d loop constraints accepted; odometry drift after the lap: %.2f m" %
          (len(out["candidates"]), acc, np.hypot(drift[0], drift[1])))
