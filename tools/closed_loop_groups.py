#!/usr/bin/env python3
"""Closed loop (BASELINE config 3, 64 particles x 910 scans) in G particle groups: seconds per leg.  python tools/closed_loop_groups.py [G ...]"""
import importlib, os, sys, time, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
pkg = importlib.import_module("slam-2d-lidar-scan_amd")
dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
readings = dataio.read_npz(os.path.join(REPO, "tests", "golden", "intel_gfs.npz"))
u = 0.02
ogP = [50.0, 50.0, readings[0], u, math.pi, 10, 180, 5 * u]
smP = [1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5]
for G in [int(a) for a in sys.argv[1:]] or [1, 2]:
    best = None
    for rep in range(3):
        pf = pkg.ParticleFilter(64, ogP, smP, rng=np.random.RandomState(0), groups=G)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pf.run(readings)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    print(f"groups {pf.n_groups}: best of 3 {best:.4f} s = {910 / best:.0f} scans/s  (aborted {pf.stats.get('aborted', 0)}, redo {pf.stats['redo']})", flush=True)
