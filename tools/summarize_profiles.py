#!/usr/bin/env python3
"""Turn the rocprofv3 outputs under gpurun_out/ into the small summaries committed under profiles/:
per-kernel stats of the bench command and per-launch HBM traffic from the PMC passes
(FETCH_SIZE / WRITE_SIZE in KB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced
reads by 2x on gfx950, other widths uncalibrated -- both raw and corrected figures are kept)."""
import collections
import csv
import glob
import json
import os
import shutil

ROUND = os.environ.get("ROUND", "r01")
os.makedirs("profiles", exist_ok=True)
for wl in ("config2", "ref2level", "config5"):
    src = f"gpurun_out/prof_{wl}/{wl}_kernel_stats.csv"
    if os.path.exists(src):
        shutil.copy(src, f"profiles/{ROUND}_{wl}_kernel_stats.csv")
        print(f"== {src}")
        for i, row in enumerate(csv.DictReader(open(src))):
            if i < 12:
                print(f"  {row['Name'][:60]:60s} calls {row['Calls']:>4s} avg {float(row['AverageNs']) / 1e3:9.2f} us  {row['Percentage']:>6s} %")

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
traffic, lines = {}, []
for k, d in sorted(agg.items()):
    if not k.startswith("k_"):
        continue
    # steady state: skip the warm-up launches (first 5 steps)
    mean = {c: sum(v[len(v) // 6:]) / max(1, len(v[len(v) // 6:])) for c, v in d.items()}
    fetch, write = mean.get("FETCH_SIZE", 0.0) * 1024, mean.get("WRITE_SIZE", 0.0) * 1024
    hit, miss = mean.get("TCC_HIT_sum", 0.0), mean.get("TCC_MISS_sum", 0.0)
    name = k.split("<")[0]
    targs = k[k.index("<") + 1:k.rindex(">")].replace(" ", "").split(",") if "<" in k else []
    mode = (targs[1] if name == "k_sweep" and len(targs) > 1 else targs[0] if name == "k_select" and targs else "0")
    if mode != "0":
        name += {"1": "_ring", "2": "_rest"}.get(mode, "_" + mode)             # passes of the prior-pruned variant
    traffic[f"config2:{name}"] = {"fetch_bytes_raw": fetch, "write_bytes_raw": write,
                                  "hbm_bytes_corrected": 2 * fetch + write,
                                  "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
                                  "launches_sampled": max(len(v) for v in d.values())}
    lines.append(f"{k:24s} FETCH {fetch / 1e6:9.2f} MB  WRITE {write / 1e6:9.2f} MB  2*FETCH+WRITE {(2 * fetch + write) / 1e6:9.2f} MB"
                 f"  L2 hit {100 * hit / (hit + miss) if hit + miss else float('nan'):5.1f} %")
if traffic:
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
    open(f"profiles/{ROUND}_pmc_config2.txt", "w").write(
        "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} --kernel-trace -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants\n"
        "per-launch means over the timed launches; FETCH/WRITE_SIZE are KB counters (x1024 here)\n" + "\n".join(lines) + "\n")
    print("\n".join(lines))

# SQ / TCP counters of every kernel (raw per-launch means; pmc_sq and pmc_tcp passes)
extra = []
for k, d in sorted(agg.items()):
    if not k.startswith("k_"):
        continue
    mean = {c: sum(v[len(v) // 4:]) / max(1, len(v[len(v) // 4:])) for c, v in d.items()}
    keys = [c for c in ("SQ_WAVES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES",
                        "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_ACCESSES_sum") if c in mean]
    if keys:
        rd = mean.get("SQ_INSTS_VMEM_RD")
        tail = f"  L1 lines per wave-load {mean['TCP_TOTAL_CACHE_ACCESSES_sum'] / rd:6.1f}" if rd and "TCP_TOTAL_CACHE_ACCESSES_sum" in mean else ""
        extra.append(f"{k[:40]:40s} " + "  ".join(f"{c} {mean[c]:.4g}" for c in keys) + tail)
if extra:
    open(f"profiles/{ROUND}_counters_config2.txt", "w").write(
        "rocprofv3 --pmc <SQ_* | TCP_*> --kernel-trace -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants\n"
        "per-launch means after the warm-up launches (raw counter values)\n" + "\n".join(extra) + "\n")
    print("\n".join(extra))
