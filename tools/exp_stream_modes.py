#!/usr/bin/env python3
"""Is the two-group step bimodal across HotPathGroups instances of one process (each takes three fresh torch streams)?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
cfg = bench.WORKLOADS["config2"]; dev = torch.device("cuda", 0)
scen = bench.Scenario(cfg, 64, 60)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    hot = bench.make_hot_path(cfg, 64, scen, dev, 2)
    for s in range(8): hot.step(s)
    hot.take_flags()
    ms = 1e3 * statistics.median([bench.timed_run(hot, 8, 50)[0] for _ in range(3)]) / 50
    print(f"instance {i}: {ms:.4f} ms per step; streams {[hex(st.cuda_stream) for st in hot.streams]} norm {hex(hot.norm.cuda_stream)}", flush=True)
    del hot
    torch.cuda.empty_cache()
