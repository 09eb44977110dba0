#!/usr/bin/env python3
"""Development aid: how long the HOST needs to enqueue one bench step (no synchronisation inside the loop) against the
device time of the same steps -- is the step host-bound?  python tools/host_enqueue_rate.py [workload]  (SLAM2D_FORCE_DIST=1
for the sharded path at one rank)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, torch.distributed as dist
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
cfg = bench.WORKLOADS[wl]; P = bench.WORKLOAD_PARTICLES.get(wl, 64); K = 200
if os.environ.get("SLAM2D_FORCE_DIST") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
scen = bench.Scenario(cfg, P, K + 10)
hot = bench.HotPath(cfg, P, scen, torch.device("cuda", 0))
for s in range(10):
    hot.step(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(10, 10 + K):
    hot.step(s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{wl}: host enqueue {1e6 * (t1 - t0) / K:.1f} us per step, until the device is done {1e6 * (t2 - t0) / K:.1f} us per step")
if dist.is_initialized():
    dist.destroy_process_group()
