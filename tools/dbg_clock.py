#!/usr/bin/env python3
"""Development aid: build the library with -DSLAM2D_DEBUG_CLOCK into a scratch .so, run a few config-2 steps and print the
in-kernel phase stamps (100 MHz wall clock) of k_bound / k_exact_select.  Run on the GPU box: python tools/dbg_clock.py"""
import ctypes as C, importlib, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
_lib = importlib.import_module("slam-2d-lidar-scan_amd._lib")
_lib.HIPCC_FLAGS.append("-DSLAM2D_DEBUG_CLOCK")
_lib.build_library(force=True)
import numpy as np, torch
import bench
cfg = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "config2"]
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scen = bench.Scenario(cfg, P, 12)
hot = bench.HotPath(cfg, P, scen, torch.device("cuda", 0))
for s in range(12):
    hot.step(s)
torch.cuda.synchronize()
L = _lib.lib()
L.slam2d_debug_clock.argtypes = [C.c_void_p]
buf = (C.c_longlong * 64)()
assert L.slam2d_debug_clock(buf) == 0
t = np.array(buf[:], dtype=np.int64)
def us(a, b): return (t[b] - t[a]) / 100.0
print("k_blur_clamp (one tile): halo -> LDS %.2f us, axis-0 pass %.2f, axis-1 pass + stores %.2f, min/max + block minima %.2f; total %.2f" % (us(50, 51), us(51, 52), us(52, 53), us(53, 54), us(50, 54)))
print("k_tile_triage (particle 0): prologue + loads issued + LDS stores %.2f us, barrier %.2f, slices OR %.2f, classify %.2f, barrier-or %.2f, lists + fills %.2f, barrier + counts %.2f; total %.2f" % (
    us(20, 21), us(21, 22), us(22, 23), us(23, 24), us(24, 25), us(25, 26), us(26, 27), us(20, 27)))
print("k_endpoints (theta 0, particle 0): cells %.2f us, hash + tile marking %.2f, global marks + compaction %.2f" % (us(40, 41), us(41, 42), us(42, 43)))
if os.environ.get("SLAM2D_BOUND_LDS", "1") != "0":
    print("k_bound_lds (block 0, wave 0): staging loads+stores %.2f us, tiles/pmax + barrier %.2f, first angle's loop %.2f, its bounds + seed pick %.2f, its seed %.2f, remaining angles %.2f; total %.2f" % (
        us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 14), us(0, 14)))
print("k_bound: prologue->loop end %.2f us, bounds+argmax %.2f, seed tile %.2f, atomic %.2f" % (us(0, 1), us(1, 2), us(2, 3), us(3, 4)))
print("k_exact_select: scan %.2f us, list %.2f, tiles %.2f, max %.2f, exp %.2f, theta sums %.2f, select %.2f; total %.2f" % (
    us(8, 9), us(9, 10), us(10, 11), us(11, 29), us(29, 30), us(30, 12), us(12, 13), us(8, 13)))
for name, a in (("coarse", 32), ("fine", 44)):
    if t[a + 3] > t[a] > 0:
        print("k_sweep %s (particle 0, theta 0, chunk 0, wave 0): gather loop %.2f us, partial sums to LDS + barrier %.2f, scores + reduction + partial %.2f; total %.2f" % (
            name, us(a, a + 1), us(a + 1, a + 2), us(a + 2, a + 3), us(a, a + 3)))
print("k_grid_update (block of particle 0, beams 0-3): whole walk %.2f us; normaliser block %.2f us; update block start relative to the normaliser block's %.2f" % (us(60, 62), us(58, 59), us(58, 60)))
b = hot.coarse.t["bounds"].cpu().numpy(); best = hot.coarse.t["bnb_best"].cpu().numpy().view(np.uint64)
bits = np.where(best >> np.uint64(63), best & np.uint64(0x7FFFFFFFFFFFFFFF), ~best).astype(np.uint64); m0 = bits.view(np.float64)
kept = (b >= (m0 - 30.0)[:, None, None, None]).reshape(P, -1).sum(axis=1)
print("  kept tiles per particle: p0 %d, min %d, median %d, max %d" % (kept[0], kept.min(), np.median(kept), kept.max()))
print(hot.coarse.bnb_stats())
