import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
for i in range(2):
    d = bench.dropin_serial(64, 53, torch.device("cuda", 0))
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k != "note"})
