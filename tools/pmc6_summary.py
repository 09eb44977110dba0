#!/usr/bin/env python3
"""Per-kernel means of the counter passes written by tools/gpu_r6.sh pmcp (gpurun_out/pmc6/<tag>/<pass>/...).
    python tools/pmc6_summary.py config2_p256 [more tags]        # prints a table; --write ROUND also writes profiles/<ROUND>_{pmc,counters}_<tag>.txt"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n


def load(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(ROOT, f"gpurun_out/pmc6/{tag}/*/**/*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k.startswith("k_"):
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, d in agg.items():
        out[k] = {c: sum(v[len(v) // 3:]) / max(1, len(v[len(v) // 3:])) for c, v in d.items()}
        out[k]["_launches"] = max(len(v) for v in d.values())
    return out


def kstat(tag):
    fs = glob.glob(os.path.join(ROOT, f"gpurun_out/kstat_{tag}/**/k_kernel_stats.csv"), recursive=True)
    dur = {}
    for r in (csv.DictReader(open(fs[0])) if fs else []):
        dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["Percentage"]))
    return dur


def table(tag):
    m, dur = load(tag), kstat(tag)
    lines = []
    order = sorted(m, key=lambda k: -dur.get(k, (0, 0, 0))[0] * dur.get(k, (0, 0, 0))[1])
    lines.append(f"{'kernel':28s} {'us':>7s} {'%':>5s} {'HBM MB':>8s} {'GB/s':>7s} {'L2hit':>6s} {'waves':>7s} {'VALU/w':>7s} {'SALU/w':>7s} {'LDS/w':>6s} {'VMRD/w':>6s} {'VMWR/w':>6s} "
                 f"{'cyc/w':>8s} {'valu_util':>9s} {'lines/ld':>8s} {'ldsbank%':>8s} {'TA busy%':>8s}")
    for k in order:
        c = m[k]
        us, calls, pct = dur.get(k, (float('nan'), 0, 0))
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        w = c.get("SQ_WAVES", 0) or float('nan')
        busy_cu = c.get("SQ_BUSY_CU_CYCLES", 0)
        # VALU issue utilisation: one VALU instruction keeps a SIMD's issue port for >= 1 cycle (wave64 on a 16-lane... gfx950: 4 cycles per wave64 op for fp32, more for fp64);
        # SQ_ACTIVE_INST_VALU counts those cycles (per SIMD); SQ_BUSY_CU_CYCLES counts cycles a CU is busy (x 4 SIMDs)
        act_valu = c.get("SQ_ACTIVE_INST_VALU", 0)
        busy = c.get("SQ_BUSY_CYCLES", 0)
        util = act_valu / (4 * busy_cu) if busy_cu else float('nan')
        rd = c.get("SQ_INSTS_VMEM_RD", 0)
        lines_per = c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / rd if rd else float('nan')
        bank = 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c.get("SQ_ACTIVE_INST_LDS", 1) if c.get("SQ_ACTIVE_INST_LDS") else float('nan')
        ta = c.get("TA_BUSY_avr", float('nan'))
        gui = c.get("GRBM_GUI_ACTIVE", 0)
        ta_pct = 100 * ta / gui if gui else float('nan')
        lines.append(f"{k[:28]:28s} {us:7.1f} {pct:5.1f} {hbm / 1e6:8.2f} {hbm / us / 1e3 if us == us else 0:7.0f} {100 * hit / (hit + miss) if hit + miss else float('nan'):6.1f} {w:7.0f} "
                     f"{c.get('SQ_INSTS_VALU', 0) / w:7.0f} {c.get('SQ_INSTS_SALU', 0) / w:7.0f} {c.get('SQ_INSTS_LDS', 0) / w:6.0f} {rd / w:6.1f} {c.get('SQ_INSTS_VMEM_WR', 0) / w:6.1f} "
                     f"{c.get('SQ_WAVE_CYCLES', 0) / w:8.0f} {util:9.3f} {lines_per:8.1f} {bank:8.1f} {ta_pct:8.1f}")
    return "\n".join(lines), m, dur


if __name__ == "__main__":
    tags = [a for a in sys.argv[1:] if not a.startswith("--")]
    for tag in tags:
        t, m, dur = table(tag)
        print(f"== {tag}")
        print(t)
        if "--raw" in sys.argv:
            for k, c in m.items():
                print(k, {a: round(b, 1) for a, b in c.items()})
