"""bench.py's launch plumbing without a GPU: `python bench.py --gpus N` must start N ranks itself, and a rank count that
does not match --gpus must fail instead of silently measuring one GPU (Algorithm/FastSlam.py:25-27 is the sharding axis the
N-GPU number is about).  `--spawn-check` stops before any GPU work."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    res = _run(["--gpus", "2", "--spawn-check"])
    assert res.returncode == 0, res.stderr[-2000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["gpus_requested"] == 2


def test_bench_single_rank_needs_no_launcher():
    res = _run(["--gpus", "1", "--spawn-check"])
    assert res.returncode == 0, res.stderr[-2000:]
    assert _json_line(res.stdout)["n_gpus"] == 1


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    res = _run(["--gpus", "8", "--spawn-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0
    assert "--gpus 8 but WORLD_SIZE=1" in res.stderr


def test_traffic_file_is_refused_when_the_kernels_changed(tmp_path, monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    entry = {"hbm_bytes_corrected": 1.0, "launches_per_step": 1.0, "in_step": True}
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    json.dump({"source_sha256": "0" * 64, "entries": {"config2:k_blur_clamp": entry}}, open(tmp_path / "profiles" / "traffic.json", "w"))
    got, note = bench.load_traffic("config2")
    assert got == {} and "refused" in note
    json.dump({"source_sha256": bench.source_sha256(), "entries": {"config2:k_blur_clamp": entry, "config5:k_bound1": entry}},
              open(tmp_path / "profiles" / "traffic.json", "w"))
    got, note = bench.load_traffic("config2")
    assert note is None and list(got) == ["k_blur_clamp"]
