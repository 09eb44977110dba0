#!/usr/bin/env python3
"""Per-stage kernel times (HIP event pairs) of one workload / input mode, branch and bound on and off.  python tools/stage_times.py [mode] [workload]"""
import os, sys, ctypes as C, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
mode = sys.argv[1] if len(sys.argv) > 1 else "displaced"
wl = sys.argv[2] if len(sys.argv) > 2 else "config2"
cfg = bench.WORKLOADS[wl]; dev = torch.device("cuda", 0); P = bench.WORKLOAD_PARTICLES.get(wl, 64)
scen = bench.Scenario(cfg, P, 40, mode=mode)
hot = bench.make_hot_path(cfg, P, scen, dev, 2)
E = hot.E; lib = E._lib.lib()
stages = list(range(8))
for bnb in (True, False):
    saved = [(lv, lv.c.bnb) for lv in hot.levels()]
    if not bnb:
        for lv, _ in saved: lv.c.bnb = 0
    for s in range(10): hot.step(s)
    hot.take_flags()
    ms = 1e3 * statistics.median([bench.timed_run(hot, 10, 30)[0] for _ in range(3)]) / 30
    hot.take_flags()
    E._lib.check(lib.slam2d_prof_enable(sum(1 << st for st in stages), 400), "prof")
    for s in range(10, 30): hot.step(s)
    hot.take_flags()
    out = {}
    for st in stages:
        tot, n = C.c_double(0), C.c_int32(0)
        lib.slam2d_prof_collect(st, C.byref(tot), C.byref(n))
        if n.value: out[E._lib.STAGE_NAMES[st]] = round(1e3 * tot.value / n.value, 1)
    lib.slam2d_prof_disable()
    st = bench.level_stats(hot)["coarse"]
    kept = None
    if bnb:
        lv = hot.subs[0].coarse
        b = lv.t["bounds"].cpu().numpy(); best = lv.t["bnb_best"].cpu().numpy().view(np.uint64)
        bits = np.where(best >> np.uint64(63), best & np.uint64(0x7FFFFFFFFFFFFFFF), ~best).astype(np.uint64); m0 = bits.view(np.float64)
        k = (b >= (m0 - 30.0)[:, None, None, None]).reshape(len(m0), -1).sum(axis=1)
        kept = dict(min=int(k.min()), median=float(np.median(k)), max=int(k.max()))
    print(f"{wl} {mode} bnb={bnb}: {ms:.4f} ms/step; stage us (with event pairs): {out}; blur tiles {st['blur_tiles']:.0f} needed {st['needed_tiles']:.0f} kept {kept}", flush=True)
    for lv, v in saved: lv.c.bnb = v
