import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch, json, statistics
cfg = bench.WORKLOADS["config2"]; dev = torch.device("cuda", 0)
scen = bench.Scenario(cfg, 64, 80)
hot = bench.make_hot_path(cfg, 64, scen, dev, 2)
for s in range(8): hot.step(s)
hot.take_flags()
base = 1e3 * statistics.median([bench.timed_run(hot, 8, 60)[0] for _ in range(3)]) / 60
print("base", base)
print(json.dumps(bench.normaliser_probe(cfg, 64, scen, dev, 60, 8, 2, base), indent=1))
