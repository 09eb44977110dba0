"""bench.py end to end on the GPU box: the N-rank path in its one-GPU dry mode (`--gpus 2 --backend gloo --share-gpu`: two
ranks, 64 particles each, both on cuda:0, the 24-byte all-gather through gloo) must report two ranks; the driver's short
form must print one well-formed line with the roofline and CPU-baseline blocks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_dry_mode():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "6", "--warmup", "3", "--repeats", "2", "--no-variants", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["ranks"]["ranks_seen"] == 2 and d["config"]["total_particles"] == 128
    assert len(d["ranks"]["ms_per_step_per_rank"]) == 2 and d["fault_flags"] == 0 and d["scaling"] == "weak"
    one = _bench(["--gpus", "1", "--steps", "6", "--warmup", "3", "--repeats", "2", "--no-variants", "--no-cpu-baseline"])
    assert one["n_gpus"] == 1 and one["config"]["total_particles"] == 64 and "ranks" not in one
    # same per-rank work in both runs: the processed-bytes accounting of a step must agree
    a, b = d["roofline"]["whole_step"]["whole_array_bytes_per_particle_scan"], one["roofline"]["whole_step"]["whole_array_bytes_per_particle_scan"]
    assert a == b


def test_driver_form_prints_the_contract_line():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-variants", "--cpu-seconds", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["event_pairs"]["launches_timed"] >= 3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert d["steps"] == 5 and d["timed_blocks"]["repeats"] == 5 and d["cpu_baseline"]["kind"] == "port"


def test_closed_loop_workload_sharded_dry_mode():
    """BASELINE configs 3 / 4 as one command (`--workload config3`): the closed loop over the Intel log through
    ParticleFilter.run(), here 16 particles over two ranks on the one GPU (gloo), 200 scans with a forced resample (weights
    all-gathered, whole particles migrated between the ranks), and the same on one rank: the job must finish with one line,
    strong scaling declared, and the resample counted."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    two = _bench(["--workload", "config3", "--gpus", "2", "--backend", "gloo", "--share-gpu", "--total-particles", "16", "--steps", "200"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["particles_per_gpu"] == 8
    assert two["resamples"] == 1 and two["steps"] == 200 and two["value"] > 0
    one = _bench(["--workload", "config3", "--particles", "16", "--steps", "200"])
    assert one["n_gpus"] == 1 and one["config"]["total_particles"] == 16 and one["resamples"] == 1
    # the shared seeded stream makes the sharded run the same filter: the same map growth for particle 0
    assert one["final_map_of_particle_0"] == two["final_map_of_particle_0"]
