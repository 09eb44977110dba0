import os
print("start", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4])
import torch
print("after import torch", len(os.sched_getaffinity(0)))
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
print("after cuda init", len(os.sched_getaffinity(0)))
import importlib, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
L = importlib.import_module("slam-2d-lidar-scan_amd._lib")
print(L.group_policy())
print(open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split()[0])
