#!/usr/bin/env python3
"""Experiment: G independent particle groups (P / G particles each) stepped on G HIP streams from one host thread, against
all P particles on one stream.  No cross-stream dependency at all (each group normalises its own weights), so this is the
UPPER bound of what overlapping groups can buy.  python tools/exp_streams.py [P] [steps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W = 8
cfg = bench.WORKLOADS[os.environ.get("WL", "config2")]
dev = torch.device("cuda", 0)
for G in (1, 2, 4):
    scen = [bench.Scenario(cfg, P // G, K + W, seed=0, rank=g) for g in range(G)]
    streams = [torch.cuda.Stream(dev) for _ in range(G)] if G > 1 else [torch.cuda.current_stream(dev)]
    hots = []
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            hots.append(bench.HotPath(cfg, P // G, scen[g], dev))
    torch.cuda.synchronize()
    import ctypes as C
    E = hots[0].E
    handles = [C.c_void_p(st.cuda_stream) for st in streams]
    def run(first, n):
        for s in range(first, first + n):
            for g in range(G):
                E._PINNED_STREAM = handles[g]          # every library call of this step goes to the group's stream
                hots[g].step(s)
        E._PINNED_STREAM = None
    run(0, W)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(W, K)
        torch.cuda.synchronize(); best.append(time.perf_counter() - t0)
    el = sorted(best)[1]
    for h in hots:
        h.eng.take_flags()
    print(f"G={G}: {P} particles, {1e3 * el / K:.4f} ms per scan of all groups, {P * K / el:.0f} particle-scans/s")
