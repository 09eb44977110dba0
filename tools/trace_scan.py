#!/usr/bin/env python3
"""One stretch of a rocprofv3 kernel trace as a timeline per hardware queue, plus per-kernel averages and per-queue sums.
python tools/trace_scan.py <k_kernel_trace.csv> [window us] [offset us into the busiest second half]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:26]) for r in rows)
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 480e3
avg = collections.defaultdict(list)
for s, e, q, n in ev[len(ev) // 3:]:
    avg[n].append(e - s)
print("per kernel: " + "; ".join(f"{n} x{len(v)} {1e-3 * sum(v) / len(v):.1f} us" for n, v in sorted(avg.items(), key=lambda kv: -sum(kv[1]))[:14]))
ups = [x for x in ev if x[3].startswith("k_grid_update")]
mid = ups[(3 * len(ups)) // 4]                       # an update launch three quarters into the trace: steady state of the last leg
t0 = mid[0] + (float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0.0)
sel = [x for x in ev if t0 <= x[0] < t0 + win]
qs = sorted({x[2] for x in sel})
print(f"window {win * 1e-3:.0f} us from an update launch; queues {qs}")
for q in qs:
    prev_end = None
    line = []
    for s, e, qq, n in sel:
        if qq != q:
            continue
        gap = "" if prev_end is None else f"(+{1e-3 * (s - prev_end):.1f})"
        line.append(f"{gap}{n.replace('k_', '')}@{1e-3 * (s - t0):.0f}+{1e-3 * (e - s):.0f}")
        prev_end = e
    print(f" q{q}: " + " ".join(line))
