"""Where a sharded candidate step's time goes (cfear_candidate_pipe): host time of submit / collect, latency and pipelined step,
for direct launches vs the captured graph, with / without the one-rank ncclComm_t, with / without exchange timing events."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tbv_slam_public_amd import api, _lib as L


class D:
    world, rank, local_rank, dist, dev = 1, 0, 0, None, torch.device("cuda", 0)
    @staticmethod
    def barrier(): torch.cuda.synchronize()
    @staticmethod
    def max_over_ranks(x): return float(x)


W = bench.LoopClosureWorld(D)
comm1 = api.RcclComm(W.ctx, 1, 0)
steps = 200
for n in (512, 1, 4096):
    cands = W.candidates(n)
    for graph in (0, 1):
        for comm in (None, comm1):
            for timing in (0, 1):
                for depth in (2, 3):
                    pipe = api.CandidatePipe(W.reg, W.table, n, comm, 0, 1, depth=depth, graph=bool(graph), timing=bool(timing))
                    out = np.empty(n, L.RESULT_DTYPE)
                    for _ in range(8):
                        pipe.collect(pipe.submit(cands), out)
                    torch.cuda.synchronize()
                    ts = tc = 0.0
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        a = time.perf_counter(); tk = pipe.submit(cands); b = time.perf_counter(); pipe.collect(tk, out); c = time.perf_counter()
                        ts += b - a; tc += c - b
                    lat = (time.perf_counter() - t0) / steps * 1e3
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tks = [pipe.submit(cands) for _ in range(depth - 1)]
                    for _ in range(steps - (depth - 1)):
                        tks.append(pipe.submit(cands)); pipe.collect(tks.pop(0), out)
                    while tks:
                        pipe.collect(tks.pop(0), out)
                    pl = (time.perf_counter() - t0) / steps * 1e3
                    print("n %4d graph %d comm %d timing %d depth %d: latency %.4f ms (submit %.4f collect-wait %.4f)  pipelined %.4f ms"
                          % (n, graph, comm is not None, timing, depth, lat, ts / steps * 1e3, tc / steps * 1e3, pl), flush=True)
                    pipe.close()
