import sys, os, gc, importlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
pkg = importlib.import_module("slam-2d-lidar-scan_amd")
gc.disable()
r0 = {"x": 0.0, "y": 0.0, "theta": 0.0, "range": [3.0] * 180}
for i in range(3):
    pf = pkg.ParticleFilter(16, [50.0, 50.0, r0, 0.02, np.pi, 10, 180, 0.1], [1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5], rng=np.random.RandomState(0))
    pf.updateParticles(r0, 1)
    w = pf.particles[3].weight; og = pf.particles[2].og.mapXLim
    del pf
    torch.cuda.synchronize()
    print("allocated MB after del (gc disabled):", round(torch.cuda.memory_allocated() / 1e6, 1))
