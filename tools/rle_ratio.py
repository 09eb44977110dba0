"""Run-length compression ratio of the coarse cell lists (k_bound_lds RLE) of a bench workload: python tools/rle_ratio.py config5 128"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch, bench
wl, P = sys.argv[1], int(sys.argv[2])
cfg = bench.WORKLOADS[wl]
scen = bench.Scenario(cfg, P, 4)
hot = bench.HotPath(cfg, P, scen, torch.device("cuda", 0))
for s in range(3):
    hot.step(s)
torch.cuda.synchronize()
lv = hot.coarse
pc = lv.t["pcells"].cpu().numpy(); kc = lv.t["kcount"].cpu().numpy()
tot = runs = 0
for p in range(min(P, 8)):
    for it in range(lv.ntheta):
        K = kc[p, it]; a = pc[p, it, :K]
        for b0 in range(0, K, 64):
            c = a[b0:b0 + 64]
            tot += len(c); runs += 1 + int((np.diff(c) != 0).sum())
print(wl, "cells", tot, "runs", runs, "ratio %.2f" % (tot / runs), "mean K %.0f" % kc.mean())
