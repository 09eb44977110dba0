#!/usr/bin/env python3
"""Closed loop (BASELINE config 3, 64 particles x 910 scans): how much of a run the host spends WAITING for a scan's report
(Event.synchronize in ParticleFilter.run()) and how much issuing -- a host that never waits is the bottleneck.
python tools/closed_loop_hostwait.py [G ...]"""
import importlib, os, sys, time, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
pkg = importlib.import_module("slam-2d-lidar-scan_amd")
dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
readings = dataio.read_npz(os.path.join(REPO, "tests", "golden", "intel_gfs.npz"))
u = 0.02
ogP = [50.0, 50.0, readings[0], u, math.pi, 10, 180, 5 * u]
smP = [1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5]
waited = [0.0, 0]
_sync = torch.cuda.Event.synchronize  # (the host link waits through slam2d_event_synchronize: not counted here)


def timed_sync(self):
    t0 = time.perf_counter()
    _sync(self)
    waited[0] += time.perf_counter() - t0
    waited[1] += 1


torch.cuda.Event.synchronize = timed_sync
filt = importlib.import_module("slam-2d-lidar-scan_amd.filter")
engine = importlib.import_module("slam-2d-lidar-scan_amd.engine")
slow = [0.0, 0]
_up = filt.ParticleFilter.updateParticles


def timed_up(self, reading, count):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _up(self, reading, count)
    torch.cuda.synchronize(); slow[0] += time.perf_counter() - t0; slow[1] += 1


filt.ParticleFilter.updateParticles = timed_up
_rw = filt._ReportWaiter.synchronize          # (round 5: the device-synced groups wait for a pushed report, not for an event)


def timed_rw(self):
    t0 = time.perf_counter()
    _rw(self)
    waited[0] += time.perf_counter() - t0
    waited[1] += 1


filt._ReportWaiter.synchronize = timed_rw
joined = [0.0, 0]
_join = filt.ParticleFilter._join_groups


def timed_join(self):
    t0 = time.perf_counter()
    _join(self)
    joined[0] += time.perf_counter() - t0; joined[1] += 1


filt.ParticleFilter._join_groups = timed_join
grow = [0.0, 0]
_mat = engine.MapState._materialise


def timed_mat(self):
    if self._pending is None:
        return
    t0 = time.perf_counter()
    _mat(self)
    grow[0] += time.perf_counter() - t0; grow[1] += 1


engine.MapState._materialise = timed_mat
refr = [0.0, 0]
_refresh = engine.ParticleEngine.refresh_maps


def timed_refresh(self):
    t0 = time.perf_counter()
    _refresh(self)
    refr[0] += time.perf_counter() - t0; refr[1] += 1


engine.ParticleEngine.refresh_maps = timed_refresh
for G in [int(a) for a in sys.argv[1:]] or [1, 2]:
    for rep in range(3):
        pf = pkg.ParticleFilter(64, ogP, smP, rng=np.random.RandomState(0), groups=G or None)
        waited[:] = [0.0, 0]; slow[:] = [0.0, 0]; grow[:] = [0.0, 0]; refr[:] = [0.0, 0]; joined[:] = [0.0, 0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pf.run(readings)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(f"groups {pf.n_groups}: {el:.4f} s = {910 / el:.0f} scans/s; host waited {waited[0]:.4f} s in {waited[1]} report waits "
              f"({1e3 * waited[0] / max(1, waited[1]):.3f} ms each), issued for {1e3 * (el - waited[0]) / 910:.3f} ms per scan "
              f"(aborted {pf.stats.get('aborted', 0)}, redo {pf.stats['redo']}, step by step {pf.stats['step_by_step']}); step-by-step scans: {slow[1]} in {slow[0]:.4f} s, "
              f"of which {grow[1]} map re-allocations {grow[0]:.4f} s (host); {refr[1]} descriptor refreshes {refr[0]:.4f} s; "
              f"{joined[1]} joins of the issuing threads {joined[0]:.4f} s", flush=True)
