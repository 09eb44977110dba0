#!/usr/bin/env python3
"""Experiment: the device-side bound of G independent particle groups with NO host cost per launch -- every group's K steps
are captured into one hipGraph on the group's stream (torch stream capture of the library's launches) and the G graphs are
replayed side by side.  Eager figures of the same objects beside it.  python tools/exp_graph.py [P] [steps]"""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W = 8
cfg = bench.WORKLOADS[os.environ.get("WL", "config2")]
dev = torch.device("cuda", 0)
for G in (1, 2, 4, 8):
    if P % G or (P // G) < 8:
        continue
    scen = [bench.Scenario(cfg, P // G, K + W, seed=0, rank=g) for g in range(G)]
    streams = [torch.cuda.Stream(dev) for _ in range(G)]
    hots = []
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            hots.append(bench.HotPath(cfg, P // G, scen[g], dev))
    torch.cuda.synchronize()
    E = hots[0].E
    handles = [C.c_void_p(st.cuda_stream) for st in streams]

    def run(first, n, groups=range(G)):
        for s in range(first, first + n):
            for g in groups:
                E._PINNED_STREAM = handles[g]
                hots[g].step(s)
        E._PINNED_STREAM = None
    run(0, W)
    torch.cuda.synchronize()
    t = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(W, K)
        torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    eager = sorted(t)[1]
    graphs = []
    try:
        for g in range(G):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[g]):
                run(W, K, groups=[g])
            graphs.append(gr)
        torch.cuda.synchronize()
        t = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    graphs[g].replay()
            torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
        rep = sorted(t)[2]
        msg = f"graph replay {1e3 * rep / K:.4f} ms per scan ({P * K / rep:.0f} particle-scans/s)"
    except Exception as exc:
        msg = "graph capture failed: " + repr(exc)[:200]
    for h in hots:
        h.eng.take_flags()
    print(f"G={G}: {P} particles: eager {1e3 * eager / K:.4f} ms per scan ({P * K / eager:.0f}/s); {msg}", flush=True)
