"""Batched FastSLAM particle filter: the behaviour of the reference's
``ParticleFilter`` / ``Particle`` (Algorithm/FastSlam.py:10-140) with every particle's
scan match and map update done in one set of kernel launches.

The reference loops over particles serially (:25-27); here the particle index is the
batch axis of the device state.  Constructor arguments and method names follow the
reference.  Sharding over several GPUs (one process per GPU) is in ``parallel.py``;
this class handles the particles of one rank.
"""
import ctypes as C
import math
import os
import weakref

import numpy as np
import torch

from . import _lib, parallel
from .engine import MATCH_DOUBLES, LidarModel, MapState, ParticleEngine, SearchLevel, pinned_stream, require_gpu, _ptr, _stream


class ParticleView:
    """What the reference's callers read from a particle (Algorithm/FastSlam.py:164-177)."""

    def __init__(self, pf, i):
        # (a weak reference: the filter owns its views, and a cycle would keep gigabytes of maps alive until the cyclic collector
        # happens to run -- bench.py's "slow legs" of round 5)
        self._pf, self._i = weakref.proxy(pf), i

    @property
    def weight(self):
        return float(self._pf.weights[self._i])

    @property
    def xTrajectory(self):
        return [t[self._i, 0] for t in self._pf.trajectory]

    @property
    def yTrajectory(self):
        return [t[self._i, 1] for t in self._pf.trajectory]

    @property
    def prevMatchedReading(self):
        m = self._pf.prev_matched[self._i]
        return {"x": float(m[0]), "y": float(m[1]), "theta": float(m[2]), "range": self._pf.prev_raw["range"]}

    @property
    def og(self):
        return MapView(self._pf, self._i)


class MapView:
    def __init__(self, pf, i):
        self._pf, self._m = pf, pf.engine.maps[i]           # (short-lived: made per access by ParticleView.og)

    @property
    def occupancyGridVisited(self):
        return self._m.download()[0]

    @property
    def occupancyGridTotal(self):
        return self._m.download()[1]

    @property
    def mapXLim(self):
        return self._m.lim_x

    @property
    def mapYLim(self):
        return self._m.lim_y

    def convertRealXYToMapIdx(self, x, y):
        return self._m.to_map_idx(x, y, self._pf.lidar.unit)

    def mapImage(self, xRange, yRange, as_u8=False):
        """The frame the reference's driver saves per scan (Algorithm/FastSlam.py:171-177) without downloading the count
        arrays: ``np.flipud(1 - (visited / total)[yIdx[0]:yIdx[1], xIdx[0]:xIdx[1]])`` as a host array, computed on the
        device (slam2d_map_image); only the window crosses PCIe."""
        xIdx, yIdx = self.convertRealXYToMapIdx(xRange, yRange)
        return self._m.image(xIdx[0], xIdx[1], yIdx[0], yIdx[1], flipud=True, as_u8=as_u8).cpu().numpy()


# the host's bound on a scan's report: beyond the device-side waits' (word 59 of the sync block: 30 s unless SLAM2D_SYNC_TIMEOUT_MS says otherwise),
# so that a tripped device-side wait is what the caller gets to see
_HOST_WAIT_S = 15.0 + (int(os.environ["SLAM2D_SYNC_TIMEOUT_MS"]) / 1e3 if os.environ.get("SLAM2D_SYNC_TIMEOUT_MS", "").isdigit() else 30.0)


class _ReportWaiter:
    """Stands where a torch Event stood in run()'s pending tuple: the scan's report is PUSHED into the pinned host pack by the
    device (Slam2dScan.h_seq); synchronize() polls the sequence word from C (GIL released), bounded."""
    __slots__ = ("ptr", "seq")

    def __init__(self, ptr, seq):
        self.ptr, self.seq = ptr, seq

    def synchronize(self):
        _lib.check(_lib.lib().slam2d_host_wait_seq(self.ptr, self.seq, _HOST_WAIT_S), "slam2d_host_wait_seq")


def _heading(dx, dy, dist):
    return math.acos(dx / dist) if dy > 0 else -math.acos(dx / dist)


class _ReadingWindow:
    """``readings[i]`` over any iterable without materialising it: ParticleFilter.run() consumes its readings in order and steps back
    at most two (a voided scan and its successor are re-issued), so the last few items suffice -- an unbounded iterator (a live
    sensor) is read one item at a time."""

    def __init__(self, iterable, keep=4):
        self._it, self._buf, self._n, self._done, self._keep = iter(iterable), {}, 0, False, keep

    def _fill(self, i):
        while not self._done and self._n <= i:
            try:
                self._buf[self._n] = next(self._it)
                self._n += 1
            except StopIteration:
                self._done = True
        for k in [k for k in self._buf if k < i - self._keep]:
            del self._buf[k]

    def has(self, i):
        self._fill(i)
        return i < self._n

    def __getitem__(self, i):
        self._fill(i)
        return self._buf[i]


class ParticleFilter:
    """``ParticleFilter(numParticles, ogParameters, smParameters)`` as in
    Algorithm/FastSlam.py:11,197-207.

    growable=True reproduces the reference's map growth (host check + device
    re-allocation per particle; one extra synchronisation per scan).  growable=False
    skips the growth checks: maps must be pre-sized, and a window that leaves the
    map raises.  ``updateParticles`` is synchronous per scan (the odometry prior of the
    next scan needs the matched poses on the host); ``bench.py`` drives the same
    kernels without host round trips.
    ``total_particles`` / ``first_index`` describe this rank's slice when sharded;
    ``rng`` defaults to the legacy global NumPy stream like the reference.
    ``match_max``: take the arg-max at the coarse level instead of the soft-max draw (``matchScan(matchMax=True)``, what
    the reference's single-trajectory driver does, Utils/ScanMatcher_OGBased.py:244); no uniform is drawn then.  With one
    particle, ``ParticleFilter(1, ..., match_max=True).run(readings)`` is that driver (``:226-256``) on the batched path.
    ``bnb``: score the pose cubes by branch and bound over 4x4 pose tiles (include/slam2d.h) -- None: wherever
    the cube is large enough for it to pay (engine.bnb_default), True / False: wherever applicable / nowhere.
    ``groups``: ``run()`` steps the particles in this many groups, each on its own HIP stream, joined only by the weight
    normaliser (slam2d_groups_match_begin / slam2d_groups_commit) -- None: ``auto_groups`` (SLAM2D_FILTER_GROUPS overrides); at
    most four (the GPU runs four compute queues side by side: a fifth group makes them take turns -- and four groups get a queue each
    only when the application has set GPU_MAX_HW_QUEUES >= 8 before its first HIP call; on the runtime's default of 4 hardware queues
    two groups share one with the default stream's, engine.group_streams).
    Through round 4 the closed loop gained nothing from groups (ten event packets and two copy-engine transfers per scan tied the
    groups together); round 5's grouped calls need none of them (include/slam2d.h, ABI 16) and the closed loop runs 10-12 % faster in
    two or four groups than in one.  Results are those of one group."""

    def __init__(self, numParticles, ogParameters, smParameters, device=None, growable=True, rng=None,
                 total_particles=None, first_index=0, group=None, bnb=None, match_max=False, groups=None, force_sharded=False):
        (mapX, mapY, initXY, unit, fov, max_range, beams, wall) = ogParameters            # :66
        (sr, half_rad, sigma, move_sigma, max_dev, turn_sigma, miss, cf) = smParameters   # :67-68
        self.device = require_gpu(device or "cuda:0")
        self.numParticles = numParticles
        self.total_particles = total_particles or numParticles
        self.first_index = first_index
        self.growable = growable
        self.match_max = bool(match_max)
        self.rng = rng
        self.group = group
        self.sharded = self.total_particles != numParticles or bool(force_sharded)     # (force_sharded: the sharded code path on ONE rank)
        if self.sharded:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise _lib.Slam2dError("a sharded ParticleFilter needs torch.distributed to be initialised")
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
            if parallel.shard_range(self.total_particles, self.world, self.rank) != (first_index, numParticles):
                raise ValueError("first_index / numParticles do not match parallel.shard_range for this rank")
            if rng is None:
                # every rank must draw the same uniforms and resample indices: an unseeded per-process stream
                # would give inconsistent send / receive plans (deadlock or silent divergence)
                raise ValueError("a sharded ParticleFilter needs rng=np.random.RandomState(seed), the same seed on every rank")
        self.lidar = LidarModel.get(unit, max_range, fov, beams, wall)
        maps = [MapState.create(mapX, mapY, initXY, unit, self.device) for _ in range(numParticles)]
        self.engine = ParticleEngine(self.lidar, maps, self.device)
        P, dev = numParticles, self.device
        common = dict(search_radius_ctor=sr, half_rad=half_rad, move_sigma=move_sigma, max_move_dev=max_dev,
                      turn_sigma=turn_sigma, bnb=bnb)
        cstep = cf * unit                                                                 # ScanMatcher_OGBased.py:54
        self.coarse = SearchLevel(self.lidar, P, dev, step=cstep, sigma=sigma / cf, miss_prob=miss, radius=sr,
                                  fine=False, **common)
        self.fine = SearchLevel(self.lidar, P, dev, step=unit, sigma=sigma, miss_prob=miss ** (2 / cf),
                                radius=cstep, fine=True, **common)                        # :66-73
        self.m_coarse = self.engine.match_buffer("coarse")
        self.m_fine = self.engine.match_buffer("fine")
        self.d_pose = torch.zeros((P, 3), dtype=torch.float64, device=dev)     # prevMatchedReading poses
        self.d_head = torch.full((P,), float("nan"), dtype=torch.float64, device=dev)   # prevMatchedMovingTheta
        self.d_est = torch.zeros((P, 3), dtype=torch.float64, device=dev)
        self.d_psi = torch.zeros((P, 2), dtype=torch.float64, device=dev)
        # the scan's inputs sit in ONE device buffer ([beams ranges | P uniforms]: one H2D copy per scan), staged through two
        # pinned host buffers (the pipelined driver, run(), stages scan s while scan s-1 may still be in flight)
        self._d_in = torch.zeros(beams + P, dtype=torch.float64, device=dev)
        self.d_ranges, self.d_uniform = self._d_in[:beams], self._d_in[beams:]
        self._h_in = [torch.zeros(beams + P, dtype=torch.float64).pin_memory() for _ in range(2)]
        # everything the host reads per scan sits in ONE device buffer (one D2H copy, one synchronisation):
        # [P x 5 report: x, y, theta, confidence, log-confidence | P normalised weights | variance, log of the weight sum | P fault-bit words]
        self._npack = 6 * P + 2
        self._d_pack = torch.zeros(self._npack + (P + 1) // 2, dtype=torch.float64, device=dev)
        self._h_pack = torch.zeros(self._npack + (P + 1) // 2, dtype=torch.float64).pin_memory()
        self._d_flagsnap = self._d_pack[self._npack:].view(torch.int32)[:P]
        self._h_flagsnap = self._h_pack[self._npack:].view(torch.int32)[:P]
        self.d_report = self._d_pack[:5 * P].view(P, 5)
        self.d_w = self._d_pack[5 * P:6 * P]
        self.d_stats = self._d_pack[6 * P:6 * P + 2]
        self.d_w.fill_(1.0)
        self.d_logw = torch.zeros(P, dtype=torch.float64, device=dev)     # log of weight = 1 (:75)
        self._normalized_step = -1
        self.weights = np.ones(P)
        self.trajectory = []
        self.prev_matched = None
        self.prev_raw = None
        self.prev_raw_heading = None
        self.particles = [ParticleView(self, i) for i in range(P)]
        self.last_confidence = np.ones(P)
        self.last_variance = None
        self._normalizer = None
        self.lazy_field = True
        # coarse level: poses the motion prior rules out are not scored (SLAM2D_MATCH_PRUNE_BY_PRIOR) -- where the cube is large
        # enough for the two extra launches (ring, then the rest for unsettled particles) to pay: at the reference's 30 x 27 x 27
        # cube the pruned coarse level takes 51 us against 31 us for the plain sweep (closed loop over the Intel log, round 3)
        import os
        from .engine import BNB_MIN_WORK
        env = os.environ.get("SLAM2D_PRUNE", "auto")
        self.prune_by_prior = env == "1" or (env != "0" and self.coarse.ntheta * self.coarse.nx ** 2 * beams >= BNB_MIN_WORK)
        self.step = 0
        env_g = os.environ.get("SLAM2D_FILTER_GROUPS", "")
        g = int(env_g) if env_g.isdigit() else (groups if groups is not None else self.auto_groups(P, self.sharded))
        if g > 4:                                        # (the GPU runs four queues side by side: five groups and more take turns --
            g = 4 if P % 4 == 0 else 2                   #  open loop 0.265-0.30 ms per scan against 0.10-0.11, closed loop 0.91 s against 0.20)
        if self.sharded:
            # every rank gathers the same number of partials per scan: the same G everywhere, so it has to divide every rank's share
            counts = [parallel.shard_range(self.total_particles, self.world, r)[1] for r in range(self.world)]
            if not env_g.isdigit() and groups is None:
                g = self.auto_groups(min(counts), True)
            self.n_groups = g if (g > 1 and all(c % g == 0 for c in counts)) else 1
        else:
            self.n_groups = g if (g > 1 and P % g == 0) else 1
        # run() goes through the grouped calls even with ONE group (their event-free closed loop: ranges pulled, report pushed:
        # 0.2215 s against 0.2284 s for the 910 Intel scans at 64 particles); SLAM2D_FILTER_GROUPED1=0: the one-stream calls.
        # A sharded filter too (round 6): its commit ends in slam2d_norm_gate + the all-gather of the partials +
        # slam2d_weights_merge_publish_report on the normaliser's stream (the event path, SLAM2D_FILTER_EVENTS=1, is one rank's only)
        self.grouped_single = os.environ.get("SLAM2D_FILTER_GROUPED1", "1") == "1"
        if self.sharded and os.environ.get("SLAM2D_FILTER_EVENTS", "0") == "1":
            self.n_groups, self.grouped_single = 1, False
        self._grp = None                                 # streams, events, level views: built by the first grouped run()
        # run(): scans redone step by step (discarded speculative match); resample(): all / those that moved any state
        self.stats = {"redo": 0, "aborted": 0, "reissued": 0, "step_by_step": 0, "resamples": 0, "state_moving_resamples": 0}

    # ---- odometry prior (Algorithm/FastSlam.py:77-106) ----
    @staticmethod
    def auto_groups(P, sharded=False):
        """Particle groups of run() when the caller names none: four from 32 particles (a multiple of 4), two from 16, where the
        host has cores for the threads that issue them (slam2d_group_policy); else one.  Round 5, 64 particles x the 910 Intel
        scans, legs interleaved in one process: one group 0.2284 s, one through the event-free calls 0.2215, two 0.2064, four
        0.2043 (the event path of rounds 3-4: 0.238 in two, 0.42-0.60 in four)."""
        if not _lib.group_policy()["threads"]:
            return 1
        if sharded:          # (a sharded rank: two groups + the normaliser's stream + the collective's make four busy queues, the most the
            return 2 if (P >= 16 and P % 2 == 0) else 1      #  GPU runs side by side: bench.py, one-rank RCCL group, 0.1211 ms in two groups, 0.1286 in four)
        if P >= 32 and P % 4 == 0:
            return 4
        return 2 if (P >= 16 and P % 2 == 0) else 1

    def _raw_odometry(self, raw, prev_raw=None, prev_raw_heading="same"):
        """The particle-independent part of updateEstimatedPose: distance and heading of the raw
        odometry step (:81-104).  Returns (estMovingDist, rawMovingTheta, has_turn, raw_turn)."""
        pr = self.prev_raw if prev_raw is None else prev_raw
        if prev_raw_heading == "same":
            prev_raw_heading = self.prev_raw_heading
        dx, dy = raw['x'] - pr['x'], raw['y'] - pr['y']
        dist = math.sqrt(dx ** 2 + dy ** 2)
        raw_heading, has_turn, turn = None, 0, 0.0
        if dist > 0.3:
            raw_heading = _heading(dx, dy, dist)
            if prev_raw_heading is not None:
                has_turn, turn = 1, raw_heading - prev_raw_heading
        return dist, raw_heading, has_turn, turn

    def _prior(self, raw):
        """Host mirror of the device prior (tests): (est [P,3], dist, psi list, raw heading)."""
        dist, raw_heading, has_turn, turn = self._raw_odometry(raw)
        pm = self.prev_matched
        est = np.empty((self.numParticles, 3))
        est[:, 0], est[:, 1] = pm[:, 0], pm[:, 1]
        est[:, 2] = pm[:, 2] + raw['theta'] - self.prev_raw['theta']
        # (the reference raises TypeError if a particle's last matched move was exactly 0)
        psi = [None if (not has_turn or h is None) else h + turn for h in self.prev_matched_heading]
        return est, dist, psi, raw_heading

    def _draw_uniforms(self):
        """One uniform per particle of the WHOLE filter, in particle order, from the legacy
        stream -- the order np.random.choice consumes it in the reference's serial loop
        (ScanMatcher_OGBased.py:138); a rank uses its own slice."""
        src = self.rng if self.rng is not None else np.random
        u = src.random_sample(self.total_particles)
        return u[self.first_index:self.first_index + self.numParticles]

    def _stage_inputs(self, which, ranges, uniforms=None):
        """The scan's ranges (and uniforms) to the device through pinned buffer `which`: one asynchronous copy."""
        h, B = self._h_in[which], self.lidar.beams
        h.numpy()[:B] = ranges
        if uniforms is None:
            self.d_ranges.copy_(h[:B], non_blocking=True)
        else:
            h.numpy()[B:] = uniforms
            self._d_in.copy_(h, non_blocking=True)

    # ---- Particle.update for all particles (Algorithm/FastSlam.py:25-27,122-135) ----
    def updateParticles(self, reading, count):
        """One scan for every particle.  Everything between the uploads (ranges, uniforms) and the
        single download at the end (poses, confidences) runs on the device without a host round
        trip; the host only decides map growth from the poses it already has."""
        eng, P, L = self.engine, self.numParticles, _lib.lib()
        rng_in = np.asarray(reading['range'], dtype=np.float64)
        if count == 1 or self.match_max:
            self._stage_inputs(0, rng_in)
        if count == 1:
            self.prev_raw_heading = None
            self.d_pose.copy_(torch.tensor([[reading['x'], reading['y'], reading['theta']]] * P, dtype=torch.float64))
            self.d_head.fill_(float("nan"))
            matched = np.tile([reading['x'], reading['y'], reading['theta']], (P, 1)).astype(np.float64)
            conf = np.ones(P)
            d_shift = self._grow_for_first_update(reading) if self.growable else None
            eng.grid_update(self.d_pose, 3, self.d_ranges, d_shift)          # :133
            eng.take_flags()
        else:
            dist, raw_heading, has_turn, turn = self._raw_odometry(reading)
            est_xy = self.prev_matched                              # estimate = previous matched x, y (:79)
            if self.growable:
                self._grow_for_windows(est_xy[:, 0], est_xy[:, 1], self.coarse.reach)
            if not self.match_max:
                self._stage_inputs(0, rng_in, self._draw_uniforms())
            _lib.check(L.slam2d_prior(_ptr(self.d_pose), float(reading['theta']), float(self.prev_raw['theta']),
                                      has_turn, float(turn), _ptr(self.d_head), P, _ptr(self.d_est),
                                      _ptr(self.d_psi), _stream()), "slam2d_prior")
            self._match(self.coarse, self.d_est, 3, dist, self.d_psi, None if self.match_max else self.d_uniform, self.m_coarse)
            if self.growable and not self._fine_window_cannot_grow(est_xy, (self.coarse.ncell + 1) * self.coarse.step):
                eng.take_flags()
                c = eng.read_matches(self.m_coarse)
                self._grow_for_windows(c["x"], c["y"], self.fine.reach)
            self._match(self.fine, self.m_coarse, MATCH_DOUBLES, dist, None, None, self.m_fine)
            _lib.check(L.slam2d_post_match(_ptr(self.m_fine), _ptr(self.m_coarse), P, _ptr(self.d_pose),
                                           _ptr(self.d_head), _ptr(self.d_logw), _ptr(self.d_report), _stream()),
                       "slam2d_post_match")
            eng.grid_update(self.d_pose, 3, self.d_ranges)          # :133 (the update window lies inside the
            #                                                         search window that was grown for)
            self._normalize_on_device()                             # Algorithm/FastSlam.py:43-48, ahead of weightUnbalanced()
            self._h_pack.copy_(self._d_pack, non_blocking=True)     # poses, confidences, weights, variance
            eng.take_flags()                                        # the one synchronisation of the scan
            while self.sharded and np.isnan(self._h_pack.numpy()[6 * P]) and self._h_pack.numpy()[6 * P + 1] == -1.0:
                # another rank voided this scan in its pipelined run() (its groups left the void partial: slam2d.h,
                # slam2d_weights_merge_publish_report): nothing was merged; gather again -- that rank comes back with the scan
                self.stats["peer_voids"] = self.stats.get("peer_voids", 0) + 1
                self._normalizer(self.d_logw, None, 1, self.d_w, self.d_stats, local_done=True)
                self._h_pack.copy_(self._d_pack, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            rep = self._h_pack.numpy()[:5 * P].reshape(P, 5)
            matched, conf = rep[:, 0:3].copy(), rep[:, 3].copy()
            self._normalized_step = self.step + 1
            self.prev_raw_heading = raw_heading
        self.trajectory.append(matched[:, :2].copy())
        self.prev_matched, self.prev_raw = matched, reading
        self.last_confidence = conf
        self.step += 1

    # ---- the reference's driver loop, pipelined (Algorithm/FastSlam.py:152-162) ----
    def run(self, readings, first_count=1, force_resample=(), on_scan=None):
        """``for reading: updateParticles(reading, count); if weightUnbalanced(): resample()`` -- the same sequence of
        decisions and results as calling those methods one by one, but the host never waits for the scan it has just
        enqueued.  Scan s's match (prior, coarse and fine level: it only reads the maps) is enqueued BEFORE the host has
        seen scan s-1's results, on the assumption -- true for all but a handful of scans of a run -- that scan s-1
        triggers neither a resample nor a map growth; the host then reads scan s-1's packed report (already complete: it
        precedes the match in stream order), and either commits scan s (bookkeeping, map update, normaliser, report
        download) or, if the assumption failed, discards the speculative match and redoes the scan step by step.  The
        legacy random stream is consumed exactly as by the unpipelined calls (its state is restored on a discard).
        ``force_resample``: scan counts after which to resample regardless (tests).  ``on_scan(count, self, unbalanced)``
        is called once a scan's results are on the host.  Returns the list of (count, resample indices)."""
        with pinned_stream():
            try:
                return self._run(readings, first_count, force_resample, on_scan)
            finally:
                # whatever ended the run (a fault raised from a scan's report, the caller's on_scan): no worker thread may still be
                # issuing a match over descriptors this object owns
                try:
                    self._join_groups()
                except _lib.Slam2dError:
                    pass

    def _run(self, readings, first_count, force_resample, on_scan):
        eng, P = self.engine, self.numParticles
        # pending = (count, reading, raw_heading, event, state of the random stream before the scan's uniforms) of the scan in flight
        resamples, pending = [], None
        events = [torch.cuda.Event(), torch.cuda.Event()]
        grouped = (self.n_groups > 1 or self.grouped_single) and self.lazy_field
        if grouped and self._grp is None:
            self._setup_groups()

        stream_rng = self.rng if self.rng is not None else np.random
        # A scan is speculated without knowing whether one of its search windows leaves a map (the reference would grow that map
        # first, Utils/ScanMatcher_OGBased.py:27): the match raises SLAM2D_F_WINDOW_OUTSIDE_MAP for such a particle and the
        # commit, told to treat that bit as fatal for the WHOLE scan (abort_mask), does nothing at all on the device.  The host
        # finds the bit in the scan's report, grows the maps and runs the scan again, step by step.  (Round 2 did not speculate
        # while any window was within 2.5 m of a map's edge: 39 % of the Intel log's scans.)  Sharded filters commit in three
        # calls around a collective and keep the conservative rule.
        # ... Sharded filters on the grouped calls (round 6) void a scan RANK BY RANK: particles are independent, so a rank whose
        # windows were inside commits its share, and only the normaliser waits -- the voiding rank's groups leave a void partial,
        # every rank's merge reports the scan as voided and changes nothing; the voiding rank grows and re-issues, the others
        # answer with another all-gather + merge over the partials they already hold (slam2d_weights_merge_publish_report).  On the
        # one-stream calls a sharded filter keeps the conservative rule (no speculation near a map's edge).
        abortable = self.growable and (not self.sharded or (grouped and self._grp.devsync))
        abort_mask = _lib.F_WINDOW_OUTSIDE_MAP if abortable else 0

        sync_tripped = [False]

        def was_aborted(p):
            """Wait for scan p's report; True if its commit was a no-op on the device (a window had left a map)."""
            p[3].synchronize()
            snap = self._h_flagsnap.numpy().view(np.uint32)
            if abortable and (snap & abort_mask).any():
                return True
            if abortable and (snap & _lib.F_SCAN_VOIDED).any() and (snap & _lib.F_SYNC_TIMEOUT).any():
                # a commit's gate gave up waiting for the match's arrivals (a device-side wait ran into its bound: the groups' queues
                # starved behind another process, a debugger, a hung peer) BEFORE anything of the scan was written: the scan is intact.
                # It is run again through the calls that wait for nothing on the device, and the sync words start afresh.
                self.stats["sync_timeouts"] = self.stats.get("sync_timeouts", 0) + 1
                sync_tripped[0] = True
                return True
            if self.sharded and abortable:
                while np.isnan(self._h_pack.numpy()[6 * P]) and self._h_pack.numpy()[6 * P + 1] == -1.0:
                    # voided on ANOTHER rank: this rank's commit stands; meet the re-issued scan's partials in another all-gather
                    self.stats["peer_voids"] = self.stats.get("peer_voids", 0) + 1
                    self._sharded_regather().synchronize()
            self._check_flag_snapshot()
            return False

        def finish(p):
            """Scan p's results are on the host: bookkeeping + the reference's degeneracy test.  Returns whether the
            reference resamples after this scan (the caller does it: the random stream may have to be rewound first)."""
            count, reading, raw_heading = p[0], p[1], p[2]
            rep = self._h_pack.numpy()[:5 * P].reshape(P, 5)
            matched, conf = rep[:, 0:3].copy(), rep[:, 3].copy()
            self.trajectory.append(matched[:, :2].copy())
            self.prev_matched, self.prev_raw, self.last_confidence = matched, reading, conf
            self.prev_raw_heading = raw_heading
            self.step += 1
            self._normalized_step = self.step
            unb = self.weightUnbalanced()
            if on_scan is not None:
                on_scan(count, self, unb)
            return unb or count in force_resample

        # the next scan's pose prior rides in this scan's commit (slam2d_scan_commit_next) when the next reading is at hand:
        # prior_ready = count of the scan whose prior the last commit wrote (its match then skips the prior's launch)
        fold_prior = ((not grouped and not self.sharded) or (grouped and self._grp.devsync)) and os.environ.get("SLAM2D_FILTER_FOLD_PRIOR", "1") != "0"
        prior_ready = [None]

        def plain(count, reading):
            """One scan through the unpipelined calls."""
            prior_ready[0] = None
            self.stats["step_by_step"] += 1
            self._quiesce_groups()
            self.updateParticles(reading, count)
            unb = self.weightUnbalanced()
            if on_scan is not None:
                on_scan(count, self, unb)
            if unb or count in force_resample:
                resamples.append((count, self.resample()))

        def discard_speculation(rng_state):
            """Throw away what is in flight: its fault flags and its draws from the random stream."""
            prior_ready[0] = None
            if grouped:
                self._join_groups()
                torch.cuda.synchronize(self.device)
                self._grp.flags2.zero_()
                self._grp.active = False
                if sync_tripped[0] and self._grp.devsync:          # (every stream is idle: the counters and tickets start from zero again)
                    bound = int(self._grp.sync[59].item())
                    self._grp.sync.zero_()
                    self._grp.sync[59] = bound
                    self._grp.gate_seq = 0
                    torch.cuda.synchronize(self.device)
            else:
                torch.cuda.current_stream().synchronize()
                eng.flags.zero_()
            if rng_state is not None:
                stream_rng.set_state(rng_state)

        def redo_aborted(p):
            """Scan p was voided on the device (and so is everything enqueued after it): back to the random stream's state
            before its uniforms, then the scan again through the calls that grow the maps."""
            self.stats["aborted"] = self.stats.get("aborted", 0) + 1
            discard_speculation(p[4])
            plain(p[0], p[1])

        # A voided scan is re-issued THROUGH THE PIPELINE once the maps have grown for it, as the step-by-step path grows them
        # (Utils/ScanMatcher_OGBased.py:27, once per level): first for the coarse windows, from the poses the host holds; if
        # those were inside already, for the fine windows, from the coarse matched poses the voided commit leaves in the scan's
        # report (its coarse match was the one the step-by-step path would compute: same maps, same inputs, same uniforms).
        # The re-issued coarse match then runs on the grown maps, the reference's on the maps before the growth: the same cells
        # -- growth pads a map with unobserved cells and shifts its low limit by a whole number of cells, and a window's first
        # map column rint((x_lo - lim) / unit) sits an integer away from rounding ties for poses that move in multiples of the
        # match step.  A scan voided three times goes through the step-by-step calls.  (Round 3 ran every voided scan AND its
        # successor step by step: 44 of the Intel log's 910 scans at 0.85 ms each.)  SLAM2D_FILTER_REISSUE=0 restores that.
        reissue = os.environ.get("SLAM2D_FILTER_REISSUE", "1") != "0"
        retried = (None, 0)                                  # (count of the scan last re-issued, how often)
        if not isinstance(readings, (list, tuple)):
            readings = _ReadingWindow(readings)              # a generator (a live sensor) stays lazy: the loop steps back two items at most

        def has_reading(k):
            return (k < len(readings)) if isinstance(readings, (list, tuple)) else readings.has(k)

        i = 0
        while has_reading(i):
            count, reading = first_count + i, readings[i]
            i += 1
            if count == 1 or (pending is None and self.prev_raw is None) or not self.lazy_field:
                assert pending is None
                plain(count, reading)
                continue
            if self.growable and not abortable and self.prev_matched is not None:
                # (no device-side abort: no speculation while some particle's window hugs its map's edge, judged from the poses
                # the host already has, one or two scans old, with 1 m of slack for the motion since)
                slack = (self.coarse.ncell + 1) * self.coarse.step + 1.0
                if self._outside(self.prev_matched[:, 0], self.prev_matched[:, 1], self.coarse.reach + slack).size:
                    if pending is not None:
                        prev_count = pending[0]
                        was_aborted(pending)
                        if finish(pending):
                            self._quiesce_groups()
                            resamples.append((prev_count, self.resample()))
                        pending = None
                    plain(count, reading)
                    continue
            parity = count & 1
            if pending is None:
                prev_raw, prev_raw_heading = self.prev_raw, self.prev_raw_heading
            else:
                prev_raw, prev_raw_heading = pending[1], pending[2]
            dist, raw_heading, has_turn, turn = self._raw_odometry(reading, prev_raw, prev_raw_heading)
            rng_state = stream_rng.get_state()
            state_before = rng_state
            if grouped:
                self._stage_inputs_group(parity, np.asarray(reading['range'], dtype=np.float64), None if self.match_max else self._draw_uniforms())
                self._enqueue_match_groups(reading, prev_raw, dist, has_turn, turn, parity, prior_ready[0] == count)
            else:
                self._stage_inputs(parity, np.asarray(reading['range'], dtype=np.float64), None if self.match_max else self._draw_uniforms())
                self._enqueue_match(reading, prev_raw, dist, has_turn, turn, prior_ready[0] == count)   # speculative: scan count-1 not seen yet
            redo = False
            if pending is not None:
                prev_count = pending[0]
                if was_aborted(pending):
                    # scan count-1 did not happen on the device: redo it (with the growth it needs), then this scan, whose
                    # speculative match started from a state that never was
                    p, pending = pending, None
                    self.stats["redo"] += 1
                    tries = retried[1] if retried[0] == p[0] else 0
                    if sync_tripped[0]:                        # a gate's timeout: this scan and its successor through the waiting-free calls
                        redo_aborted(p)
                        sync_tripped[0] = False
                    elif reissue and tries < 2:
                        self.stats["aborted"] = self.stats.get("aborted", 0) + 1
                        coarse_xy = self._h_pack.numpy()[:5 * P].reshape(P, 5)[:, :2].copy()     # (a voided commit reports the coarse poses)
                        discard_speculation(p[4])
                        grew = self._grow_for_windows(self.prev_matched[:, 0], self.prev_matched[:, 1], self.coarse.reach)
                        if not grew:
                            grew = self._grow_for_windows(coarse_xy[:, 0], coarse_xy[:, 1], self.fine.reach)
                        if grew:
                            retried = (p[0], tries + 1)
                            self.stats["reissued"] += 1
                            i -= 2                              # both scans again, speculatively
                            continue
                        plain(p[0], p[1])
                    else:
                        redo_aborted(p)
                    plain(count, reading)
                    continue
                if finish(pending):
                    # the reference draws the resample indices BEFORE this scan's uniforms: rewind, resample, redo the scan
                    stream_rng.set_state(rng_state)
                    self._quiesce_groups()                      # (the speculative match of this scan reads the maps a resample moves)
                    idx = self.resample()
                    resamples.append((prev_count, idx))
                    rng_state = None
                    # ... unless the draw moved nothing (always so with one particle, whose degeneracy test is always true,
                    # Algorithm/FastSlam.py:37) and the speculative match consumed no uniform (match_max): it stands as it is
                    redo = not (self.match_max and np.array_equal(np.asarray(idx), np.arange(self.total_particles)))
                    if not redo:
                        state_before = stream_rng.get_state()
                pending = None
            if not abortable:
                est_xy = self.prev_matched
                margin = (self.coarse.ncell + 1) * self.coarse.step
                if not redo and self.growable and self._outside(est_xy[:, 0], est_xy[:, 1], self.coarse.reach + margin).size:
                    redo = True                                                     # a window may leave a map: grow, step by step
            if redo:
                self.stats["redo"] += 1
                discard_speculation(rng_state)
                plain(count, reading)
                continue
            nxt = None
            if fold_prior and has_reading(i):
                _, _, n_has_turn, n_turn = self._raw_odometry(readings[i], reading, raw_heading)
                nxt = (float(readings[i]['theta']), float(reading['theta']), int(n_has_turn), float(n_turn))
                prior_ready[0] = count + 1
            if grouped:
                ev = self._enqueue_commit_groups(abort_mask, parity, nxt, readings[i]['range'] if nxt is not None else None)
            else:
                self._enqueue_commit(abort_mask, nxt)
                ev = events[parity]
                ev.record()
            pending = (count, reading, raw_heading, ev, state_before)
        if pending is not None:
            if was_aborted(pending):
                redo_aborted(pending)
            elif finish(pending):
                self._quiesce_groups()
                resamples.append((pending[0], self.resample()))
        self._quiesce_groups()
        if grouped:             # both buffers of fault bits (a bit an update raised after its launch's snapshot is still there)
            f = self._grp.flags2.cpu().numpy().view(np.uint32)
            self._grp.flags2.zero_()
            bad = np.argwhere(f & _lib.FATAL_FLAGS)
            if bad.size:
                b, pp = bad[0]
                raise _lib.Slam2dError(f"particle {int(pp)}: {_lib.describe_flags(int(f[b, pp]) & _lib.FATAL_FLAGS)}")
        else:
            eng.take_flags()    # a bit the last update raised after its launch's snapshot (slam2d_scan_commit) is still there
        return resamples

    # ---- particle groups on streams (run() only): slam2d_groups_match / slam2d_groups_commit ----
    def _setup_groups(self):
        """Streams, events, offset views of both levels and the C descriptors of the groups.  The fault bits get a second buffer:
        a group may be a scan ahead of another, and the commit of scan s decides its abort over ALL groups' bits of scan s --
        scans alternate between the two buffers (include/slam2d.h, Slam2dScan.d_abort_flags)."""
        L, eng, P, G, dev = _lib.lib(), self.engine, self.numParticles, self.n_groups, self.device
        per = P // G
        grp = type("Groups", (), {})()
        grp.per = per
        grp.flags2 = torch.zeros((2, P), dtype=torch.int32, device=dev)
        eng.flags = grp.flags2[0]                        # (the call-by-call path keeps using buffer 0)
        grp.d_in = [self._d_in, torch.zeros_like(self._d_in)]
        from .engine import group_streams
        pool = group_streams(dev, G + 1)
        grp.streams, grp.norm = pool[1:], pool[0]
        grp.ev_matched = [C.c_void_p(L.slam2d_event_create()) for _ in range(G)]
        grp.ev_done = [C.c_void_p(L.slam2d_event_create()) for _ in range(G)]
        grp.ev_merged, grp.ev_inputs = C.c_void_p(L.slam2d_event_create()), C.c_void_p(L.slam2d_event_create())
        grp.ready = [torch.cuda.Event(), torch.cuda.Event()]
        grp.parts = torch.zeros((G, 3), dtype=torch.float64, device=dev)
        grp.coarse = [self.coarse.view(g * per, (g + 1) * per) for g in range(G)]
        grp.fine = [self.fine.view(g * per, (g + 1) * per) for g in range(G)]
        grp.c = (_lib.Slam2dGroup * G)()
        grp.scan = _lib.Slam2dScan()
        for g in range(G):
            cg = grp.c[g]
            cg.coarse, cg.fine = C.pointer(grp.coarse[g]), C.pointer(grp.fine[g])
            cg.P, cg.est_stride = per, 3
            cg.stream = C.c_void_p(grp.streams[g].cuda_stream)
            cg.ev_matched, cg.ev_done = grp.ev_matched[g], grp.ev_done[g]
            cg.d_part = grp.parts[g].data_ptr()
        sc = grp.scan
        sc.n_local, sc.n_parts, sc.total_particles = P, G, P
        sc.d_parts = grp.parts.data_ptr()
        sc.norm_stream, sc.ev_merged, sc.ev_inputs = C.c_void_p(grp.norm.cuda_stream), grp.ev_merged, grp.ev_inputs
        sc.merge = 1
        if self.sharded:
            # the groups only arrive and wait (merge == 0); the rank's G partials are all-gathered on the normaliser's stream and merged
            # there in (rank, group) order (_sharded_gather_merge)
            import torch.distributed as dist
            grp.parts_all = torch.zeros((G * self.world, 3), dtype=torch.float64, device=dev)
            sc.merge, sc.n_parts, sc.total_particles = 0, G * self.world, self.total_particles
            sc.d_parts = grp.parts_all.data_ptr()
            grp.via_host = dist.get_backend(self.group) == "gloo"
            grp.rccl = None if grp.via_host else parallel.DirectRccl.create(dev, self.group)
        grp.merged_once, grp.active = False, False
        # Round 5: the grouped closed loop WITHOUT events and copies (include/slam2d.h, ABI 16).  Every group's prior launch pulls the
        # scan's inputs from the pinned staging buffer; the abort decision over all groups' fault bits sits behind an arrival counter
        # and a one-wave gate kernel; the normaliser merges on the device; the block that finishes a scan pushes the report into the
        # pinned host pack and publishes the scan's number, which the host polls; the match is issued by worker threads while this
        # thread goes on (slam2d_groups_match_begin).  SLAM2D_FILTER_EVENTS=1: the event path of rounds 3-4.
        grp.devsync = os.environ.get("SLAM2D_FILTER_EVENTS", "0") != "1"
        if grp.devsync:
            B = self.lidar.beams
            grp.sync = torch.zeros(64, dtype=torch.int32, device=dev)
            grp.h_seq = torch.zeros(16, dtype=torch.int32).pin_memory()
            grp.d_pull = torch.zeros((2, G, B), dtype=torch.float64, device=dev)      # (scan parity: a commit leaves the next scan's ranges)
            grp.gate_seq = grp.report_seq = 0
            for g in range(G):
                grp.c[g].ev_matched = grp.c[g].ev_done = None
            sc.ev_inputs = sc.ev_merged = sc.norm_stream = None
            sc.d_norm_sync = grp.sync.data_ptr()
            ms = os.environ.get("SLAM2D_SYNC_TIMEOUT_MS", "")
            if ms.isdigit():                             # the bound of the device-side waits (word 59; default 30 s)
                grp.sync[59] = int(ms)
            if not self.sharded:                         # (sharded: the merge on the normaliser's stream pushes the report, slam2d_weights_merge_publish_report)
                sc.h_seq, sc.h_pack, sc.d_pack = grp.h_seq.data_ptr(), self._h_pack.data_ptr(), self._d_pack.data_ptr()
                sc.pack_doubles = self._d_pack.numel()
            torch.cuda.synchronize(dev)
        self._grp = grp

    def _join_groups(self):
        """Wait for the worker threads of a match left running (slam2d_groups_match_begin) -- before the device is synchronised or
        anything that match reads is touched."""
        if self._grp is not None and self._grp.devsync:
            _lib.check(_lib.lib().slam2d_groups_join(), "slam2d_groups_join")

    def _bind_groups(self, parity):
        """Per-scan pointers of the group descriptors (resampling replaces pose / heading tensors, growth the map descriptors)."""
        grp, eng, per = self._grp, self.engine, self._grp.per
        B = self.lidar.beams
        d_in = grp.d_in[parity]
        flags = grp.flags2[parity]
        sm = C.sizeof(_lib.Slam2dMap)
        for lv, views in ((self.coarse, grp.coarse), (self.fine, grp.fine)):
            for v in views:
                lv.sync_view(v)
        for g in range(self.n_groups):
            cg, p0 = grp.c[g], g * per
            cg.d_maps = eng.d_maps.data_ptr() + p0 * sm
            cg.d_uniform = None if self.match_max else d_in.data_ptr() + (B + p0) * 8
            cg.d_prev_pose, cg.d_heading = self.d_pose.data_ptr() + p0 * 24, self.d_head.data_ptr() + p0 * 8
            cg.d_est_out, cg.d_psi_out = self.d_est.data_ptr() + p0 * 24, self.d_psi.data_ptr() + p0 * 16
            cg.d_coarse, cg.d_fine = self.m_coarse.data_ptr() + p0 * MATCH_DOUBLES * 8, self.m_fine.data_ptr() + p0 * MATCH_DOUBLES * 8
            cg.d_flags = flags.data_ptr() + p0 * 4
            cg.d_logw = self.d_logw.data_ptr() + p0 * 8
            cg.d_report = self.d_report.data_ptr() + p0 * 40
            cg.d_flag_snapshot = self._d_flagsnap.data_ptr() + p0 * 4
        sc = grp.scan
        sc.d_ranges = d_in.data_ptr()
        if grp.devsync:
            h = self._h_in[parity].data_ptr()
            sc.d_ranges, sc.h_ranges = None, h
            for g in range(self.n_groups):
                cg = grp.c[g]
                cg.d_uniform = None
                cg.h_uniform = None if self.match_max else h + (B + g * per) * 8
                cg.d_pull, cg.d_pull_next = grp.d_pull[parity, g].data_ptr(), grp.d_pull[parity ^ 1, g].data_ptr()
        sc.d_abort_flags, sc.n_abort_flags = flags.data_ptr(), self.numParticles
        sc.d_logw_all, sc.d_w, sc.d_stats = self.d_logw.data_ptr(), self.d_w.data_ptr(), self.d_stats.data_ptr()

    def _quiesce_groups(self):
        """Before anything that runs on the main stream over all particles (the call-by-call path, a resample, a growth): wait
        for the group streams."""
        if self._grp is not None and self._grp.active:
            self._join_groups()
            torch.cuda.synchronize(self.device)
            self._grp.active = False

    def _stage_inputs_group(self, parity, ranges, uniforms):
        """Scan inputs through pinned buffer `parity` into the device buffer of the same parity (a group may still be reading the
        other one); the groups' streams wait for the copy (Slam2dScan.ev_inputs)."""
        h, B = self._h_in[parity], self.lidar.beams
        h.numpy()[:B] = ranges
        if uniforms is not None:
            h.numpy()[B:] = uniforms
        if not self._grp.devsync:                        # (device-synced groups pull from the pinned buffer themselves)
            self._grp.d_in[parity].copy_(h, non_blocking=True)

    def _enqueue_match_groups(self, reading, prev_raw, dist, has_turn, turn, parity, prior_ready=False):
        eng, grp, L = self.engine, self._grp, _lib.lib()
        self._join_groups()
        eng.refresh_bits()
        for lv in (self.coarse, self.fine):
            if lv.c.occ_gen >= 254:                      # the stamp wraps: the occupancy images are zeroed -- with every stream idle
                torch.cuda.synchronize(self.device)
            lv.next_generation()
        self._bind_groups(parity)
        sc = grp.scan
        if grp.devsync:
            # nothing orders the group streams behind the main stream any more (the event path's ev_inputs did): whatever the main
            # stream still holds -- a bit refresh, an image zeroed, a growth, a step-by-step scan -- is finished first (one query
            # per scan; idle in the steady state)
            ms = torch.cuda.current_stream(self.device)
            if not ms.query():
                ms.synchronize()
            if self.n_groups > 1:                        # (every particle's selecting wave counts itself in: P per such call)
                grp.gate_seq = ((grp.gate_seq + 1) & 0xFFFFFFFF) or 1
            sc.match_seq = grp.gate_seq or 1
        else:
            L.slam2d_event_record(grp.ev_inputs, _stream())  # behind the staging copy (and a bit refresh) on the main stream
        sc.est_moving_dist, sc.raw_theta, sc.prev_raw_theta = float(dist), float(reading['theta']), float(prev_raw['theta'])
        sc.has_turn, sc.raw_turn = int(has_turn), float(turn)
        sc.options = _lib.MATCH_PRUNE_BY_PRIOR if self.prune_by_prior else 0
        if prior_ready and grp.devsync:                  # (the previous commit wrote this scan's prior and pulled its ranges)
            sc.options |= _lib.MATCH_PRIOR_READY
        sc.abort_mask = 0
        if grp.devsync:
            _lib.check(L.slam2d_groups_match_begin(C.byref(eng.lidar_c), grp.c, self.n_groups, C.byref(sc)), "slam2d_groups_match_begin")
        else:
            _lib.check(L.slam2d_groups_match(C.byref(eng.lidar_c), grp.c, self.n_groups, C.byref(sc)), "slam2d_groups_match")
        grp.active = True

    def _enqueue_commit_groups(self, abort_mask, parity, next_prior=None, next_ranges=None):
        eng, grp, L = self.engine, self._grp, _lib.lib()
        self._join_groups()                              # (the match's launches are all enqueued before anything below synchronises)
        # A promotion to 64-bit cells re-allocates the promoted maps' cells and replaces the descriptor array, on the main stream, while the
        # group streams may still run this scan's match over the old ones: they are idled BEFORE anything is freed (the caching
        # allocator could hand the old blocks to the very allocations that follow).  The test is _before_update's own.
        if eng._max_bound + 2 * (eng._pending_updates + 1) > _lib.COUNT_LIMIT:
            torch.cuda.synchronize(self.device)
        seen = eng.maps_version
        eng._before_update()
        if eng.maps_version != seen:                     # a map was promoted (new arrays, new descriptors)
            torch.cuda.synchronize(self.device)
            self._bind_groups(parity)
        sc = grp.scan
        sc.abort_mask, sc.wait_merged = int(abort_mask), int(grp.merged_once)
        if os.environ.get("SLAM2D_FILTER_NO_ABORT") == "1":      # timing experiment only (a window leaving a map is then fatal)
            sc.abort_mask = 0
        if grp.devsync:
            # as in the match: nothing orders the group streams behind the main stream -- a resample that moved nothing leaves the
            # speculative match standing and its weight reset (d_logw / d_w fills) on the main stream, which this commit's normaliser
            # blocks rewrite (idle in the steady state: one query)
            ms = torch.cuda.current_stream(self.device)
            if not ms.query():
                ms.synchronize()
        if grp.devsync:
            sc.match_seq = grp.gate_seq or 1             # (the match call's: its particles are what this commit's gates wait for)
            if next_prior is not None:
                # the next scan's prior and ranges ride in this commit (Slam2dScan.h_next_ranges): its ranges go into the OTHER
                # staging buffer now -- the one that scan's uniforms will follow into
                hn = self._h_in[parity ^ 1]
                hn.numpy()[:self.lidar.beams] = np.asarray(next_ranges, dtype=np.float64)
                sc.h_next_ranges = hn.data_ptr()
                sc.next_raw_theta, sc.next_prev_raw_theta, sc.next_has_turn, sc.next_raw_turn = next_prior
            else:
                sc.h_next_ranges = None
            grp.report_seq = (grp.report_seq + 1) & 0xFFFFFFFF
            sc.report_seq = grp.report_seq
            _lib.check(L.slam2d_groups_commit(C.byref(eng.lidar_c), grp.c, self.n_groups, C.byref(sc)), "slam2d_groups_commit")
            if self.sharded:
                self._sharded_gather_merge(grp.report_seq)
            grp.merged_once = True
            return _ReportWaiter(grp.h_seq.data_ptr(), grp.report_seq)
        _lib.check(L.slam2d_groups_commit(C.byref(eng.lidar_c), grp.c, self.n_groups, C.byref(sc)), "slam2d_groups_commit")
        grp.merged_once = True
        with torch.cuda.stream(grp.norm):                # behind the merge: report, weights, variance and the fault-bit snapshot
            self._h_pack.copy_(self._d_pack, non_blocking=True)
            grp.ready[parity].record()
        return grp.ready[parity]

    def _sharded_regather(self):
        """Another all-gather + merge of the scan just committed, over the partials this rank already holds (the scan was voided on
        another rank and comes again there): no gate -- this rank's groups do not arrive a second time."""
        grp = self._grp
        grp.report_seq = (grp.report_seq + 1) & 0xFFFFFFFF
        self._sharded_gather_merge(grp.report_seq, gate=False)
        return _ReportWaiter(grp.h_seq.data_ptr(), grp.report_seq)

    def _sharded_gather_merge(self, report_seq, gate=True):
        """The sharded half of a grouped commit, on the normaliser's stream: a one-wave gate waits for the groups' normaliser blocks
        (their partials), ONE all-gather of the rank's G x 3 doubles, then the merge over all (rank, group) partials in that order --
        which publishes the word the groups' next normaliser blocks wait for and pushes the scan's report to the host
        (slam2d_weights_merge_publish_report).  No event, no copy; with RCCL nothing here waits on the host."""
        import torch.distributed as dist
        grp, L, G = self._grp, _lib.lib(), self.n_groups
        ns = C.c_void_p(grp.norm.cuda_stream)
        if gate:
            _lib.check(L.slam2d_norm_gate(C.c_void_p(grp.sync.data_ptr()), G, ns), "slam2d_norm_gate")
        if grp.rccl is not None:
            grp.rccl.all_gather(grp.parts.data_ptr(), grp.parts_all.data_ptr(), 3 * G, ns)
        else:
            with torch.cuda.stream(grp.norm):
                if grp.via_host:                         # gloo (tests, dry runs): the partials hop through host memory
                    mine = grp.parts.reshape(-1).cpu()
                    got = [torch.empty_like(mine) for _ in range(self.world)]
                    dist.all_gather(got, mine, group=self.group)
                    grp.parts_all.copy_(torch.cat(got).reshape(-1, 3))
                else:
                    dist.all_gather_into_tensor(grp.parts_all.reshape(-1), grp.parts.reshape(-1), group=self.group)
        _lib.check(L.slam2d_weights_merge_publish_report(
            _ptr(self.d_logw), self.numParticles, _ptr(grp.parts_all), G * self.world, self.total_particles, _ptr(self.d_w), _ptr(self.d_stats),
            C.c_void_p(grp.sync.data_ptr()), _ptr(self._d_pack), _ptr(self._h_pack), self._d_pack.numel(), C.c_void_p(grp.h_seq.data_ptr()),
            int(report_seq), ns), "slam2d_weights_merge_publish_report")

    def _enqueue_match(self, reading, prev_raw, dist, has_turn, turn, prior_ready=False):
        """prior + coarse + fine match of one scan for all particles (slam2d_scan_match: one library call); reads the
        maps, changes no filter state.  prior_ready: the previous commit wrote this scan's prior already (_enqueue_commit)."""
        eng, P = self.engine, self.numParticles
        eng.refresh_bits()
        self.coarse.next_generation()
        self.fine.next_generation()
        _lib.check(_lib.lib().slam2d_scan_match(
            C.byref(eng.lidar_c), C.byref(self.coarse.c), C.byref(self.fine.c), _ptr(eng.d_maps), P, _ptr(self.d_pose),
            float(reading['theta']), float(prev_raw['theta']), has_turn, float(turn), _ptr(self.d_head), _ptr(self.d_ranges),
            float(dist), None if self.match_max else _ptr(self.d_uniform), _ptr(self.d_est), _ptr(self.d_psi), _ptr(self.m_coarse),
            _ptr(self.m_fine),
            _ptr(eng.flags), (_lib.MATCH_PRUNE_BY_PRIOR if self.prune_by_prior else 0) | (_lib.MATCH_PRIOR_READY if prior_ready else 0),
            _stream()), "slam2d_scan_match")

    def _enqueue_commit(self, abort_mask=0, next_prior=None):
        """Bookkeeping, map update, normaliser and the (asynchronous) download of everything the host reads.  abort_mask: fault
        bits of the match that turn the whole commit into a device-side no-op (slam2d_scan_commit).  next_prior = (theta of the
        next raw reading, theta of this one, has_turn, raw turn): the next scan's pose prior is written in the same launch
        (slam2d_scan_commit_next; Algorithm/FastSlam.py:77-106 needs nothing else that is not on the device then)."""
        eng, P = self.engine, self.numParticles
        own_norm = not self.sharded
        eng._before_update()
        args = (C.byref(eng.lidar_c), _ptr(eng.d_maps), P, _ptr(self.m_fine), _ptr(self.m_coarse), _ptr(self.d_pose),
                _ptr(self.d_head), _ptr(self.d_logw), _ptr(self.d_report), _ptr(self.d_ranges), _ptr(eng.flags),
                _ptr(self.d_w) if own_norm else None, _ptr(self.d_stats) if own_norm else None,
                _ptr(self._d_flagsnap) if own_norm else None, abort_mask if own_norm else 0)
        if next_prior is not None and own_norm:
            _lib.check(_lib.lib().slam2d_scan_commit_next(*args, *next_prior, _ptr(self.d_est), _ptr(self.d_psi), _stream()),
                       "slam2d_scan_commit_next")
        else:
            _lib.check(_lib.lib().slam2d_scan_commit(*args, _stream()), "slam2d_scan_commit")
        if not own_norm:
            self._normalize_on_device()                 # two launches around the one all-gather of the scan
            self._d_flagsnap.copy_(eng.flags)
            eng.flags.zero_()
        self._h_pack.copy_(self._d_pack, non_blocking=True)           # report, weights, variance AND the fault-bit snapshot

    def _check_flag_snapshot(self):
        f = self._h_flagsnap.numpy().view(np.uint32)
        bad = np.flatnonzero(f & _lib.FATAL_FLAGS)
        if bad.size:
            p = int(bad[0])
            raise _lib.Slam2dError(f"particle {p}: {_lib.describe_flags(int(f[p]) & _lib.FATAL_FLAGS)}")

    @property
    def prev_matched_heading(self):
        """prevMatchedMovingTheta per particle (None where the pose did not move), from the device."""
        return [None if math.isnan(v) else float(v) for v in self.d_head.cpu().numpy()]

    def _limits(self):
        """[P, 4] array (x0, x1, y0, y1) of the particles' map limits, rebuilt only after a growth or a
        resample (the engine's map list is replaced then)."""
        maps = self.engine.maps
        key = (id(maps), self.engine.maps_version)      # refresh_maps() bumps the version after every growth / resample
        if getattr(self, "_lims_key", None) != key:
            self._lims = np.array([[m.lim_x[0], m.lim_x[1], m.lim_y[0], m.lim_y[1]] for m in maps])
            self._lims_key = key
        return self._lims

    def _outside(self, xs, ys, reach):
        """Indices of the particles whose window [x +- reach] x [y +- reach] leaves their map."""
        L = self._limits()
        xs, ys = np.asarray(xs), np.asarray(ys)
        bad = (xs - reach < L[:, 0]) | (xs + reach > L[:, 1]) | (ys - reach < L[:, 2]) | (ys + reach > L[:, 3])
        return np.flatnonzero(bad)

    def _grow_for_windows(self, xs, ys, reach):
        """checkAndExapndOG of every particle's search window (ScanMatcher_OGBased.py:27).  The
        common case (every window inside its map) is one vectorised comparison."""
        grew = False
        for i in self._outside(xs, ys, reach):
            m = self.engine.maps[i]
            n = len(m.growth_log)
            m.ensure_contains([xs[i] - reach, xs[i] + reach], [ys[i] - reach, ys[i] + reach], self.lidar.unit)
            grew |= len(m.growth_log) != n
        if grew:
            self.engine.refresh_maps()
        return grew

    def _fine_window_cannot_grow(self, est, margin):
        """True when every particle's coarse window, widened by `margin` (the largest coarse
        displacement), lies inside its map: the fine window then needs no growth whatever the
        coarse result is, and the mid-scan synchronisation can be skipped."""
        return self._outside(est[:, 0], est[:, 1], self.coarse.reach + margin).size == 0

    def _grow_for_first_update(self, reading):
        """The first scan's update may leave a small initial map; the reference then grows the map beam by
        beam INSIDE the update and writes with stale indices (Utils/OccupancyGrid.py:144-152).  Every
        particle has the same pose and the same map at that point, so the growth sequence and the per-beam
        index shifts are computed once (LidarModel.grow_for_update) and replayed on the other maps.
        Returns the [P, beams, 2] shift tensor for slam2d_grid_update, or None."""
        x, y, R = reading['x'], reading['y'], self.lidar.max_range
        maps = self.engine.maps
        m0 = maps[0]
        if not (x - R < m0.lim_x[0] or x + R > m0.lim_x[1] or y - R < m0.lim_y[0] or y + R > m0.lim_y[1]):
            return None
        n0 = len(m0.growth_log)
        shifts = self.lidar.grow_for_update(m0, x, y, reading['theta'], np.asarray(reading['range'], dtype=np.float64))
        for m in maps[1:]:
            with m.deferred_growth():                  # the whole sequence, one re-allocation per map
                for side, _ in m0.growth_log[n0:]:
                    m._grow(side, self.lidar.unit)
        if len(m0.growth_log) != n0:
            self.engine.refresh_maps()
        if shifts is None:
            return None
        return self.engine.to_device(np.repeat(shifts[None], self.numParticles, axis=0), dtype=np.int32)

    def _match(self, level, d_est, stride, dist, d_psi, d_uniform, d_out):
        """One level of matchScan for all particles.  lazy_field: blur only the field tiles the sweep reads
        (slam2d_match) -- same results, the full probSP image is not materialised."""
        eng = self.engine
        if self.lazy_field:
            eng.match(level, d_est, stride, self.d_ranges, dist, d_psi, d_uniform, d_out, prune=self.prune_by_prior)
        else:
            eng.field_build(level, d_est, stride)
            eng.sweep(level, d_est, stride, self.d_ranges, dist, d_psi, d_uniform, d_out)

    # ---- weights (Algorithm/FastSlam.py:30-48) ----
    # The reference's resample trigger (:37) sits at total degeneracy: variance > ((N-1)/N)^2 + (N-1-1e-15)/N^2,
    # i.e. within ~1e-15 of the largest value sum (w - 1/N)^2 can take.  A variance more than this far below
    # that maximum cannot trigger, whatever the rounding of the reference's sequential sum:
    _DEGENERACY_BAND = 1e-9

    def normalizeWeights(self):
        """weights <- weights / sum (:43-48) in the log domain.  Sharded: ONE collective per scan -- the
        all-gather of every rank's three partial sums (parallel.ShardedNormalizer), which yields the
        normalised weights of this rank's particles and sum (w - 1/N)^2 over all N, identical on every rank.
        The N weights themselves are gathered only when the filter is within _DEGENERACY_BAND of total
        degeneracy (the only place the reference's trigger can fire) or when a resample needs them.
        ``self.weights`` holds this rank's particles; ``self.all_weights`` all N (None until gathered)."""
        n, P = self.total_particles, self.numParticles
        if self._normalized_step == self.step:          # updateParticles already ran the normaliser and downloaded its results
            host = self._h_pack.numpy()
        else:
            self._normalize_on_device()
            host = self._d_pack.cpu().numpy()
        self._normalized_step = -1
        self.weights = host[5 * P:6 * P].copy()
        if self.sharded:
            self.all_weights = None
            self.last_variance = float(host[6 * P])                  # sum w^2 - 1/N, same bits on every rank
            if self.last_variance >= (n - 1) / n - self._DEGENERACY_BAND:
                self._gather_all_weights()
        else:
            self.all_weights = self.weights
            self._sequential_variance()
        if np.isnan(self.weights).any():
            # the reference fails loudly here: np.random.choice raises on NaN probabilities (a NaN cube entry,
            # e.g. the arccos argument of the heading prior rounding above 1, ScanMatcher_OGBased.py:107,138)
            raise _lib.Slam2dError("a particle weight is NaN (NaN confidence from the scan matcher)")

    def _normalize_on_device(self):
        """log-weights -> normalised weights + variance on the device (no host round trip)."""
        if self.sharded:
            if self._normalizer is None:
                # (partials per rank: as many as run()'s grouped commits gather, so that a rank that takes a scan step by step and a
                # rank that takes it through the groups meet in the same all-gather)
                slots = self.n_groups if (self.n_groups > 1 or self.grouped_single) and self.lazy_field else 1
                self._normalizer = parallel.ShardedNormalizer(_lib.lib(), _lib.check, self.device, self.total_particles, self.group, slots=slots)
            self._normalizer(self.d_logw, None, 1, self.d_w, self.d_stats)
        else:
            _lib.check(_lib.lib().slam2d_weights_normalize(_ptr(self.d_logw), None, 1, self.numParticles, _ptr(self.d_w),
                                                           _ptr(self.d_stats), _stream()), "slam2d_weights_normalize")

    def _sequential_variance(self):
        # sum (w_i - 1/N)^2 in the reference's sequential order (:32-35): at total degeneracy the outcome of
        # its trigger is decided by the rounding of this very sum
        n = self.total_particles
        self.last_variance = float(np.cumsum((self.all_weights - 1 / n) ** 2)[-1])

    def _gather_all_weights(self):
        """All N normalised weights on every rank (second collective; near-degeneracy and resampling only)."""
        if self.all_weights is None:
            self.all_weights = parallel.gather_weights(self.d_w, self.total_particles, self.world, self.group).cpu().numpy()
            self._sequential_variance()

    def weightUnbalanced(self):
        self.normalizeWeights()
        n = self.total_particles
        return self.last_variance > ((n - 1) / n) ** 2 + (n - 1.000000000000001) * (1 / n) ** 2     # :37

    # ---- resample (Algorithm/FastSlam.py:50-62) ----
    def resample(self):
        """np.random.choice(N, N, p=weights) (:59) on every rank from the shared seeded stream, then the
        state movement: local clones, and -- sharded -- point-to-point transfers of the particles that change
        rank (parallel.migrate_ragged)."""
        n = self.total_particles
        if self.sharded:
            self._gather_all_weights()
        src = self.rng if self.rng is not None else np.random
        idx = src.choice(np.arange(n), n, p=self.all_weights)                            # :59
        self.apply_resample(idx)
        return idx

    def _gather_maps(self, maps, idx):
        """new[i] = copy of maps[idx[i]].  Maps of one extent are copied by a single gather kernel
        (slam2d_gather_maps); ragged extents fall back to per-map device copies."""
        ref = maps[0]
        same = all((m.rows, m.cols, m.pitch, m.wide) == (ref.rows, ref.cols, ref.pitch, ref.wide) for m in maps)
        if not same:
            return [maps[j].clone() for j in idx]
        new = []
        for j in idx:
            src = maps[j]
            m = MapState.__new__(MapState)
            m.device, m.X, m.Y = src.device, src.X.copy(), src.Y.copy()
            m.rows, m.cols, m.pitch = src.rows, src.cols, src.pitch
            m.cells = torch.empty_like(src.cells)
            m._pending, m._defer = None, False
            m.wide, m.count_bound = src.wide, src.count_bound
            m._alloc_bits()
            m._sync_coords()
            m.growth_log = list(src.growth_log)
            new.append(m)
        from .engine import upload_map_descs
        d_src, d_dst = upload_map_descs(maps, self.device), upload_map_descs(new, self.device)
        d_idx = torch.as_tensor(np.asarray(idx, dtype=np.int32), device=self.device)
        _lib.check(_lib.lib().slam2d_gather_maps(_ptr(d_src), _ptr(d_dst), _ptr(d_idx), len(new),
                                                 ref.rows * ref.pitch * (2 if ref.wide else 1), _stream()), "slam2d_gather_maps")
        torch.cuda.current_stream().synchronize()      # the descriptor uploads must outlive the kernel
        return new

    def apply_resample(self, idx):
        n, P, first = self.total_particles, self.numParticles, self.first_index
        maps = self.engine.maps
        moved = not np.array_equal(np.asarray(idx), np.arange(n))
        self.engine.sync_bounds()                     # the maps are about to be copied: their count bounds go with them
        self.stats["resamples"] += 1
        self.stats["state_moving_resamples"] += int(moved)
        if not moved:
            pass        # every particle is replaced by a copy of itself: no map, pose or trajectory moves (only the weights reset)
        elif not self.sharded:
            self.engine.maps = self._gather_maps(maps, idx)         # deepcopy of the chosen particles (:61)
            local = np.asarray(idx)
            tidx = torch.as_tensor(local.astype(np.int64), device=self.device)
            self.d_pose = self.d_pose[tidx].contiguous()
            self.d_head = self.d_head[tidx].contiguous()
            self.prev_matched = self.prev_matched[local].copy()
            self.trajectory = [t[local].copy() for t in self.trajectory]
        else:
            # every particle travels whole: count map + (coordinate vectors, growth log, pose, heading,
            # trajectory), whatever extent its map has grown to (parallel.migrate_ragged)
            head = self.d_head.cpu().numpy()
            traj = np.stack(self.trajectory, axis=1) if self.trajectory else np.zeros((P, 0, 2))
            aux = [parallel.pack_particle(m.X, m.Y, m.growth_log, self.prev_matched[i], head[i], traj[i], m.count_bound).to(self.device)
                   for i, m in enumerate(maps)]
            cells, aux = parallel.migrate_ragged([m.cells for m in maps], aux, idx, n, self.world, self.rank, self.group)
            new_maps, poses, heads, trajs = [], [], [], []
            for c, a in zip(cells, aux):
                o = parallel.unpack_particle(a)
                m = MapState(o["X"], o["Y"], self.device, cells=c.contiguous())
                m.growth_log, m.count_bound = o["growth_log"], o["count_bound"]
                new_maps.append(m)
                poses.append(o["pose"]); heads.append(o["heading"]); trajs.append(o["trajectory"])
            self.engine.maps = new_maps
            self.prev_matched = np.array(poses).reshape(P, 3)
            self.d_pose = torch.as_tensor(self.prev_matched, device=self.device).contiguous()
            self.d_head = torch.as_tensor(np.array(heads, dtype=np.float64), device=self.device)
            T = len(self.trajectory)
            tr = np.array(trajs).reshape(P, T, 2)
            self.trajectory = [tr[:, t].copy() for t in range(T)]
        if moved:
            self.engine.refresh_maps()
        self.weights = np.full(P, 1 / n)                                                 # :62
        self.all_weights = np.full(n, 1 / n)
        self.d_logw.fill_(math.log(1 / n))
        self.d_w.fill_(1 / n)

    def best_particle(self):
        return self.particles[int(np.argmax(self.weights))]
