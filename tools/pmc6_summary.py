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


def alone_table(tag):
    """Standalone durations (the counter passes serialise the kernels) + LDS counters where collected."""
    import glob as g
    alone = collections.defaultdict(list)
    for f in g.glob(os.path.join(ROOT, f"gpurun_out/pmc6/{tag}/sq/**/pmc_kernel_trace.csv"), recursive=True) or g.glob(os.path.join(ROOT, f"gpurun_out/pmc6/{tag}/*/**/pmc_kernel_trace.csv"), recursive=True)[:1]:
        for r in csv.DictReader(open(f)):
            alone[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    m = load(tag)
    out = []
    for k, v in sorted(alone.items(), key=lambda kv: -sum(kv[1])):
        if len(v) < 10 or not k.startswith("k_"):
            continue
        v = v[len(v) // 3:]
        c = m.get(k, {})
        out.append(f"  {k[:30]:30s} alone {sum(v) / len(v):7.1f} us (min {min(v):6.1f})  " + "  ".join(f"{a} {c[a]:.4g}" for a in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_ANY") if a in c))
    return "\n".join(out)


if __name__ == "__main__":
    tags = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--alone" in sys.argv:
        for tag in tags:
            print(f"== {tag} (standalone durations under the counter passes)")
            print(alone_table(tag))
    for tag in tags:
        t, m, dur = table(tag)
        print(f"== {tag}")
        print(t)
        if "--raw" in sys.argv:
            for k, c in m.items():
                print(k, {a: round(b, 1) for a, b in c.items()})


def write_profiles(tag, rnd):
    """profiles/<rnd>_<tag>_kernel_stats.csv (rocprofv3 --stats, copied), _pmc.txt (HBM traffic per launch), _counters.txt (SQ / TCP / TA
    per launch + the derived columns of table())."""
    import shutil, hashlib, glob as g
    fs = g.glob(os.path.join(ROOT, f"gpurun_out/kstat_{tag}/**/k_kernel_stats.csv"), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(ROOT, f"profiles/{rnd}_{tag}_kernel_stats.csv"))
    t, m, dur = table(tag)
    wl, p = tag.rsplit("_p", 1)
    p = p.split("_")[0]                                     # (a tag suffix behind the particle count: config2_p256_final)
    sha = hashlib.sha256(open(os.path.join(ROOT, "slam-2d-lidar-scan_amd/csrc/slam2d.hip"), "rb").read()).hexdigest()[:16]
    # standalone durations: the counter passes serialise the kernels (one group's launch at a time)
    alone = collections.defaultdict(list)
    for f in g.glob(os.path.join(ROOT, f"gpurun_out/pmc6/{tag}/fetch/**/pmc_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            alone[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    head = (f"rocprofv3 --pmc <one pass per line of tools/gpu_r6.sh pmcp> --kernel-trace -- python bench.py --workload {wl} --particles {p} --steps 12 --warmup 6 --repeats 1 "
            f"--no-cpu-baseline --no-variants\nper-launch means after the warm-up launches; slam2d.hip sha256 {sha}\n"
            f"a launch serves one particle group = {int(p) // 2} particles (bench.py runs {p} particles per GPU in two groups on two streams)\n")
    lines = [head, "HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are KB counters; gfx950: HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md)",
             f"{'kernel':28s} {'FETCH MB':>9s} {'WRITE MB':>9s} {'HBM MB':>8s} {'L2 hit %':>8s} {'us alone':>9s} {'us in step':>10s} {'GB/s alone':>10s}"]
    tot = 0.0
    for k, c in sorted(m.items(), key=lambda kv: -dur.get(kv[0], (0, 0, 0))[0] * dur.get(kv[0], (0, 0, 0))[1]):
        if c["_launches"] < 10:
            continue
        f_, w_ = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
        hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        a = alone.get(k, [])
        a = sum(a[len(a) // 3:]) / max(1, len(a[len(a) // 3:])) if a else float("nan")
        tot += 2 * f_ + w_
        lines.append(f"{k[:28]:28s} {f_ / 1e6:9.2f} {w_ / 1e6:9.2f} {(2 * f_ + w_) / 1e6:8.2f} {100 * hit / (hit + miss) if hit + miss else float('nan'):8.1f} {a:9.1f} "
                     f"{dur.get(k, (float('nan'),))[0]:10.1f} {(2 * f_ + w_) / a / 1e3 if a == a else 0:10.0f}")
    lines.append(f"HBM bytes of one launch sequence (one scan of one group of {int(p) // 2} particles): {tot / 1e6:.1f} MB = {tot / (int(p) // 2) / 1e3:.0f} KB per particle-scan")
    open(os.path.join(ROOT, f"profiles/{rnd}_{tag}_pmc.txt"), "w").write("\n".join(lines) + "\n")
    cl = [head, "derived per launch (us / % from rocprofv3 --kernel-trace --stats of the same command, two groups overlapping; /w = per wave; cyc/w = SQ_WAVE_CYCLES (quad-cycles) per wave;",
          "valu_util = SQ_ACTIVE_INST_VALU / (4 SQ_BUSY_CU_CYCLES); lines/ld = TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD; TA busy% = TA_BUSY_avr / GRBM_GUI_ACTIVE)", t, "",
          "raw counter means per launch"]
    for k, c in m.items():
        if c["_launches"] >= 10:
            cl.append(f"{k[:34]:34s} " + "  ".join(f"{a} {b:.5g}" for a, b in sorted(c.items()) if not a.startswith("_")))
    open(os.path.join(ROOT, f"profiles/{rnd}_{tag}_counters.txt"), "w").write("\n".join(cl) + "\n")


if __name__ == "__main__" and "--write" in sys.argv:
    rnd = sys.argv[sys.argv.index("--write") + 1]
    for tag in [a for a in sys.argv[1:] if not a.startswith("--") and a != rnd]:
        write_profiles(tag, rnd)
