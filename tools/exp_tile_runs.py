#!/usr/bin/env python3
"""Experiment: how many tiles of the blur list have their left neighbour in the list too (they could reuse half of the axis-0
pass)?  python tools/exp_tile_runs.py [workload]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
cfg = bench.WORKLOADS[wl]
P = bench.WORKLOAD_PARTICLES.get(wl, 64)
scen = bench.Scenario(cfg, P, 30)
hot = bench.HotPath(cfg, P, scen, torch.device("cuda", 0))
for s in range(30):
    hot.step(s)
    if s in (10, 20, 29):
        torch.cuda.synchronize()
        for name, lv in (("coarse", hot.coarse), ("fine", hot.fine)):
            if lv is None: continue
            tc = lv.t["tilecount"].cpu().numpy(); tl = lv.t["tilelist"].cpu().numpy()
            st = lv.t["tilestate"].cpu().numpy().reshape(P, -1)
            left = tot = free = left_real = 0
            for p in range(P):
                lst = tl[p, 0, :tc[p, 0]]
                tiles = set(lst.tolist())
                real = set(t for t in tiles if st[p, t] != 0)          # the blur really ran (the halo holds an occupied cell)
                tot += len(tiles); left += sum(1 for t in tiles if (t % lv.tmax) and (t - 1) in tiles)
                free += len(tiles) - len(real); left_real += sum(1 for t in real if (t % lv.tmax) and (t - 1) in real)
            print(f"{wl} scan {s} {name}: {tot / P:.1f} listed tiles per particle, {100 * left / max(tot, 1):.1f} % with their left neighbour listed; "
                  f"{100 * free / max(tot, 1):.1f} % of the listed tiles turn out free (constant path); of the really blurred ones "
                  f"{100 * left_real / max(tot - free, 1):.1f} % have a really blurred left neighbour")
