#!/usr/bin/env python
"""Writes a simple_graph.sgh (types.cpp:103-113) from the batched fuser: one synthetic sequence, every keyframe a node
(pose, compensated clouds, surface points) with AddToGraph's odometry constraint.  Needs an MI355X.
    python tools/make_graph.py --frames 60 --out gpurun_out/simple_graph.sgh
    python bench.py --workload loopclosure --graph gpurun_out/simple_graph.sgh"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--out", default="simple_graph.sgh")
    args = ap.parse_args()
    import torch
    from tbv_slam_public_amd import api, synth
    sc = synth.Scene(args.seed)
    od = api.OdometryKeyframeFuser(1, 400, 3360, api.odometry_params(keep_nodes=1))
    nodes = []
    for f in range(args.frames):
        info = od.process(torch.from_numpy(sc.render(f, args.frames)[None]).cuda())
        if not info["keyframe_added"][0]:
            continue
        nd = od.node(0)
        c = od.constraint(0)
        nodes.append(dict(T=info["pose"][0], idx=len(nodes), stamp=int(1e9 * 0.25 * f), cloud_peaks=nd["peaks"], cloud_nopeaks=nd["cloud"],
                          cells=nd["scan"].GetCells(), radius=3.0, weight_intensity=1, constraints=[c] if c else []))
    api.SaveSimpleGraph(args.out, nodes)
    print("wrote %s: %d nodes, %d bytes" % (args.out, len(nodes), os.path.getsize(args.out)))


if __name__ == "__main__":
    main()
