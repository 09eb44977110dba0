#!/usr/bin/env python3
"""Idle time in front of every kernel of a single-stream rocprofv3 kernel trace: per kernel name, mean duration and mean gap
(start - end of the previous kernel, whatever it was).  python tools/trace_gaps.py <k_kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows)
skip = int(len(ev) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.5))
ev = ev[skip:]
dur, gap, prevname = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(collections.Counter)
split = len(sys.argv) > 3 and sys.argv[3] == "split"          # "split": a kernel's launches told apart by the kernel in front of them
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    if split:
        n1 = n1 + " <- " + n0[:22]
    dur[n1].append(e1 - s1); gap[n1].append(s1 - e0); prevname[n1][n0] += 1
wall = ev[-1][1] - ev[0][0]
busy = sum(e - s for s, e, _ in ev)
print(f"kernels {len(ev)}  wall {wall / 1e3:.0f} us  busy {busy / 1e3:.0f} us  idle {100 * (1 - busy / wall):.1f} %")
for n in sorted(dur, key=lambda k: -sum(dur[k]) - sum(gap[k])):
    g = sorted(gap[n])
    print(f"  {n:40s} n {len(g):5d}  dur {sum(dur[n]) / len(g) / 1e3:7.2f} us  gap before: mean {sum(g) / len(g) / 1e3:7.2f}  median {g[len(g) // 2] / 1e3:7.2f}  p90 {g[int(0.9 * len(g))] / 1e3:7.2f}   after {prevname[n].most_common(1)[0][0]}")
