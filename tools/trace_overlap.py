#!/usr/bin/env python3
"""Overlap of kernels in a rocprofv3 kernel trace: sum of durations, union of busy time, per-queue sums.  python tools/trace_overlap.py <k_kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:28]) for r in rows]
ev.sort()
n = len(ev)
skip = n // 3                                       # steady state
ev = ev[skip:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
total = sum(e[1] - e[0] for e in ev)
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
perq = collections.Counter()
for s, e, q, _ in ev: perq[q] += e - s
print(f"kernels {len(ev)}  wall {1e-3 * (t1 - t0):.0f} us  sum of durations {1e-3 * total:.0f} us  union busy {1e-3 * busy:.0f} us  concurrency {total / busy:.2f}  idle {100 * (1 - busy / (t1 - t0)):.1f} %")
print("per queue:", {q: round(1e-3 * v) for q, v in perq.items()})
