#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/gpu_r4.sh (stages `kstat` and `pmc`) under gpurun_out/ into the small summaries
committed under profiles/:

  profiles/<round>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --workload <wl>`
  profiles/<round>_pmc_<workload>.txt            per-launch HBM traffic of every kernel (FETCH_SIZE / WRITE_SIZE passes)
  profiles/<round>_counters_<workload>.txt       SQ / TCP counters per launch
  profiles/traffic.json                          what bench.py's roofline block reads: per-launch HBM bytes per kernel,
                                                 keyed "<workload>:<kernel>", with the SHA-256 of the slam2d.hip they
                                                 were measured on (bench.py refuses the file when the source has changed)

FETCH_SIZE / WRITE_SIZE are KB counters; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports half of the bytes of
wide coalesced reads on gfx950, so HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (both raw figures are kept)."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil

ROUND = os.environ.get("ROUND", "r04")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
os.makedirs("profiles", exist_ok=True)
SRC = os.path.join("slam-2d-lidar-scan_amd", "csrc", "slam2d.hip")
sha = hashlib.sha256(open(SRC, "rb").read()).hexdigest()


def short(name):
    """`void k_blur_clamp<8>(Slam2dLevel)` -> ('k_blur_clamp', '<8>')."""
    n = name.replace("void ", "").split("(")[0]
    base = n.split("<")[0]
    return base, n[len(base):]


for wl in ("config2", "ref2level", "config5"):
    one = glob.glob(f"gpurun_out/kstat_{wl}_1group/**/k_kernel_stats.csv", recursive=True)
    if one:                                              # all particles in ONE launch sequence (SLAM2D_BENCH_GROUPS=1)
        shutil.copy(one[0], f"profiles/{ROUND}_{wl}_kernel_stats_1group.csv")
    found = glob.glob(f"gpurun_out/kstat_{wl}/**/k_kernel_stats.csv", recursive=True)
    if found:
        shutil.copy(found[0], f"profiles/{ROUND}_{wl}_kernel_stats.csv")
        print(f"== {found[0]}")
        for i, row in enumerate(csv.DictReader(open(found[0]))):
            if i < 14:
                print(f"  {row['Name'][:60]:60s} calls {row['Calls']:>4s} avg {float(row['AverageNs']) / 1e3:9.2f} us  {row['Percentage']:>6s} %")

entries = {}
for wl in ("config2", "ref2level", "config5"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc3/{wl}/*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            base, targs = short(row["Kernel_Name"])
            if base.startswith("k_"):
                agg[(base, targs)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if not agg:
        continue
    # launches per step: relative to the map update (exactly one per step)
    n_steps = max(len(v) for v in agg.get(("k_grid_update", ""), {"x": [0]}).values()) or 1
    lines, extra = [], []
    per_kernel = collections.defaultdict(lambda: dict(fetch=0.0, write=0.0, hit=0.0, miss=0.0, launches=0))
    for (base, targs), d in sorted(agg.items()):
        skip = {c: len(v) // 3 for c, v in d.items()}                       # steady state: drop the warm-up launches
        mean = {c: sum(v[skip[c]:]) / max(1, len(v[skip[c]:])) for c, v in d.items()}
        n = max(len(v) for v in d.values())
        fetch, write = mean.get("FETCH_SIZE", 0.0) * 1024, mean.get("WRITE_SIZE", 0.0) * 1024
        hit, miss = mean.get("TCC_HIT_sum", 0.0), mean.get("TCC_MISS_sum", 0.0)
        lps = n / n_steps
        k = per_kernel[{"k_bound_lds": "k_bound"}.get(base, base)]      # (the LDS-staged bounds are the same stage of the step)
        k["fetch"] += fetch * lps; k["write"] += write * lps; k["hit"] += hit * lps; k["miss"] += miss * lps; k["launches"] += n
        lines.append(f"{base + targs:34s} launches/step {lps:5.2f}  FETCH {fetch / 1e6:9.2f} MB  WRITE {write / 1e6:9.2f} MB  2*FETCH+WRITE {(2 * fetch + write) / 1e6:9.2f} MB"
                     f"  L2 hit {100 * hit / (hit + miss) if hit + miss else float('nan'):5.1f} %")
        keys = [c for c in ("SQ_WAVES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES",
                            "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_ACCESSES_sum") if c in mean]
        if keys:
            rd = mean.get("SQ_INSTS_VMEM_RD")
            tail = f"  L1 lines per wave-load {mean['TCP_TOTAL_CACHE_ACCESSES_sum'] / rd:6.1f}" if rd and "TCP_TOTAL_CACHE_ACCESSES_sum" in mean else ""
            extra.append(f"{(base + targs)[:40]:40s} " + "  ".join(f"{c} {mean[c]:.4g}" for c in keys) + tail)
    for base, k in per_kernel.items():
        lps = k["launches"] / n_steps
        in_step = lps >= 0.5
        per_launch = 1.0 / lps if lps else 0.0
        entries[f"{wl}:{base}"] = {"fetch_bytes_raw": k["fetch"] * per_launch, "write_bytes_raw": k["write"] * per_launch,
                                   "hbm_bytes_corrected": (2 * k["fetch"] + k["write"]) * per_launch,
                                   "l2_hit_rate": k["hit"] / (k["hit"] + k["miss"]) if k["hit"] + k["miss"] else None,
                                   "launches_per_step": lps, "in_step": in_step, "launches_sampled": k["launches"]}
    # what the SAME run really processed (bench.py's own accounting, from the JSON line it printed under the profiler): the
    # tile counts move along the trajectory, so `wasted` = traffic / processed must use the bytes of the scans measured
    try:
        import sys
        sys.path.insert(0, ROOT)
        import bench
        line = [ln for ln in open(f"gpurun_out/pmc3/{wl}/fetch.log").read().splitlines() if ln.startswith("{")][-1]
        doc = json.loads(line)
        pb, P = doc["roofline"]["processed_bytes_per_particle_scan"], doc["config"]["particles_per_gpu"]
        nprobe = min(8, doc["steps"])
        lps_of = {k: v["launches"] / nprobe for k, v in doc["stages_probe"].items()}
        for base in per_kernel:
            stage = {"k_exact_select": "k_exact"}.get(base, base)
            if base not in bench.STAGE_OF_KERNEL and base != "k_grid_update":
                continue
            # launches per step of the kernel: the probe's count where the stage has events, else one per level and group (the blur's)
            lps = lps_of.get(stage, lps_of.get("k_blur_clamp", 1.0))
            proc = bench.stage_bytes_per_launch(base, pb, P, lps)
            entries[f"{wl}:{base}"]["processed_bytes_at_measurement"] = proc
            entries[f"{wl}:{base}"]["tile_stats_at_measurement"] = doc["roofline"]["tile_stats"]
    except Exception as exc:                               # the summaries stand without it
        print("no processed-bytes accounting for", wl, repr(exc))
    try:                                                   # rocprofv3 --kernel-trace --stats averages of the same command (tools/gpu_r5.sh kstat)
        import csv as _csv
        ks = glob.glob(f"gpurun_out/kstat_{wl}/**/k_kernel_stats.csv", recursive=True)
        if ks:
            dur = collections.defaultdict(lambda: [0.0, 0])
            for r in _csv.DictReader(open(ks[0])):
                nm = r["Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
                nm = {"k_bound_lds": "k_bound"}.get(nm, nm)
                dur[nm][0] += float(r["AverageNs"]) * int(r["Calls"]); dur[nm][1] += int(r["Calls"])
            for base in per_kernel:
                if dur[base][1]:
                    entries[f"{wl}:{base}"]["avg_us_rocprof"] = dur[base][0] / dur[base][1] / 1e3
    except Exception as exc:
        print("no kernel-stats durations for", wl, repr(exc))
    step_total = sum(v["hbm_bytes_corrected"] * v["launches_per_step"] for kk, v in entries.items() if kk.startswith(wl + ":") and v["in_step"])
    head = (f"rocprofv3 --pmc {{FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum}} --kernel-trace -- python bench.py --workload {wl} --steps 12 --warmup 6 "
            "--repeats 1 --no-cpu-baseline --no-variants\nper-launch means after the warm-up launches; FETCH/WRITE_SIZE are KB counters (x1024 here); "
            f"slam2d.hip sha256 {sha[:16]}\nlaunches/step is relative to k_grid_update (one launch per scan and particle group: bench.py steps its particles in "
            f"groups on separate streams, so a 64-particle scan is four launch sequences of 16 -- config 5: two of 64)\n"
            f"HBM bytes of one launch sequence (all kernels of one scan of one particle group): {step_total / 1e6:.1f} MB\n")
    open(f"profiles/{ROUND}_pmc_{wl}.txt", "w").write(head + "\n".join(lines) + "\n")
    print(head + "\n".join(lines))
    if extra:
        open(f"profiles/{ROUND}_counters_{wl}.txt", "w").write(
            f"rocprofv3 --pmc <SQ_* | TCP_*> --kernel-trace -- python bench.py --workload {wl} --steps 12 --warmup 6 --repeats 1 --no-cpu-baseline --no-variants\n"
            "per-launch means after the warm-up launches (raw counter values)\n" + "\n".join(extra) + "\n")
        print("\n".join(extra))

if entries:
    json.dump({"source_sha256": sha, "round": ROUND, "entries": entries}, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
