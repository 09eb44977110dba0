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


def _bench(args, timeout=900, prefix=(), extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    res = subprocess.run(list(prefix) + [sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
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


def test_ranks_survive_a_core_quota():
    """8 ranks on a 16-core quota is what the driver's 8-GPU box offers: 2 cores per rank.  The grouped scan calls keep polling
    host threads only where a rank has >= 3 cores (include/slam2d.h: slam2d_group_policy).  (a) One rank pinned to ONE core must
    step within 15 % of the unrestricted run -- the library's own host threads are what this measures.  (b) Two ranks sharing the
    GPU (gloo dry mode) pinned to 4 cores -- the driver's 2 per rank -- likewise; on 2 cores (1 per rank) the dry mode's per-scan
    host round trip (synchronise, gloo all-gather over loopback, copy back) no longer fits beside the other rank -- measured 3-5x,
    with no thread of ours polling -- so that case is only required to finish with the right policy; numbers go to
    gpurun_out/core_quota.json."""
    import shutil
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if shutil.which("taskset") is None or len(os.sched_getaffinity(0)) < 4:
        pytest.skip("needs taskset and 4 cores")
    common = ["--steps", "40", "--warmup", "6", "--repeats", "3", "--no-variants", "--no-cpu-baseline"]
    rows = {}
    rows["1 rank, unrestricted"] = one = _bench(["--gpus", "1"] + common)
    rows["1 rank, 1 core"] = pinned = _bench(["--gpus", "1"] + common, prefix=("taskset", "-c", "0"))
    pol = pinned["timed_blocks"]["host_issue"]
    if pol["cores"] != 1:                                   # (seen on some boxes of the pool: the pin does not reach the rank)
        pytest.skip(f"taskset -c 0 had no effect here: {pol}")
    assert not pol["threads"], pol
    assert pinned["ms_per_step"] <= 1.15 * one["ms_per_step"], (pinned["ms_per_step"], one["ms_per_step"])
    two = ["--gpus", "2", "--backend", "gloo", "--share-gpu"] + common
    rows["2 ranks, unrestricted"] = free = _bench(two)
    assert free["timed_blocks"]["host_issue"]["local_ranks"] == 2
    for cores in ("0-3", "0-1"):
        rows[f"2 ranks, cores {cores}"] = d = _bench(two, prefix=("taskset", "-c", cores))
        pol = d["timed_blocks"]["host_issue"]
        if pol["cores"] != (4 if cores == "0-3" else 2):
            pytest.skip(f"taskset -c {cores} had no effect here: {pol}")
        assert not pol["threads"], pol                      # < 3 cores per rank: no polling workers
        assert d["fault_flags"] == 0 and d["ranks"]["ranks_seen"] == 2
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "core_quota.json"), "w") as f:
        json.dump({k: {"ms_per_step": v["ms_per_step"], "host_issue": v["timed_blocks"]["host_issue"],
                       "host_enqueue_ms_per_step": v["timed_blocks"]["host_enqueue_ms_per_step"]} for k, v in rows.items()}, f, indent=1)
    assert rows["2 ranks, cores 0-3"]["ms_per_step"] <= 1.15 * free["ms_per_step"] + 0.010, (rows["2 ranks, cores 0-3"]["ms_per_step"], free["ms_per_step"])
