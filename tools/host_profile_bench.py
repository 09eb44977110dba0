#!/usr/bin/env python3
"""Development aid: where the host spends its time enqueueing one bench step (cProfile over N steps of the config-2 hot path,
one particle group).  Run on the GPU box: python tools/host_profile_bench.py [workload] [P] [groups]"""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = bench.WORKLOADS[wl]
scen = bench.Scenario(cfg, P, 40)
hot = bench.make_hot_path(cfg, P, scen, torch.device("cuda", 0), G)
for s in range(40):
    hot.step(s)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
for i in range(N):
    hot.step(i % 40)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/step, with drain %.1f us/step" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    hot.step(i % 40)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
