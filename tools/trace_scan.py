#!/usr/bin/env python3
"""One scan of a rocprofv3 kernel trace as a timeline (steady state, from one k_grid_update wave of launches to the next) plus
per-kernel averages and per-queue sums.  python tools/trace_scan.py <k_kernel_trace.csv> [scan index from the middle]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]) for r in rows)
ev = ev[len(ev) // 3:]
avg = collections.defaultdict(list)
for s, e, q, n in ev:
    avg[n].append(e - s)
print("per kernel: " + "; ".join(f"{n} x{len(v)} {1e-3 * sum(v) / len(v):.1f} us (max {1e-3 * max(v):.0f})" for n, v in sorted(avg.items(), key=lambda kv: -sum(kv[1]))[:16]))
perq = collections.Counter()
for s, e, q, n in ev:
    perq[q] += e - s
print("per queue busy us:", {q: round(1e-3 * v) for q, v in perq.items()}, " wall us:", round(1e-3 * (ev[-1][1] - ev[0][0])))
ups = [i for i, x in enumerate(ev) if x[3].startswith("k_grid_update")]
nq = max(1, len({ev[i][2] for i in ups}))
mid = ups[(len(ups) // 2 // nq) * nq + int(sys.argv[2]) * nq if len(sys.argv) > 2 else (len(ups) // 2 // nq) * nq]
end = ups[ups.index(mid) + nq] if ups.index(mid) + nq < len(ups) else len(ev) - 1
t0 = ev[mid][0]
for s, e, q, n in ev[mid:end + nq]:
    print(f"  q{q:>3} {1e-3 * (s - t0):9.1f} us  +{1e-3 * (e - s):7.1f}  {n}")
