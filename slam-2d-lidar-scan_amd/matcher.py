"""``ScanMatcher`` with the reference's class surface
(Utils/ScanMatcher_OGBased.py:8-176), running the field build and the pose-cube
sweep on the GPU for one particle.

``matchScan`` is synchronous like the reference's (the caller needs the pose
before its next statement).  The batched, all-particles path is
``filter.ParticleFilter``.
"""
import copy
import math

import numpy as np
import torch

from . import _lib
from .engine import MATCH_DOUBLES, _MATCH_DTYPE, ParticleEngine, SearchLevel, pinned_stream

_level_cache = {}


class _CallBuffers:
    """Host / device staging of one synchronous matchScan call, kept with the grid's one-particle engine: a pinned input
    buffer (one H2D copy per call) and ONE device buffer holding both levels' results and the fault bits (one D2H copy and one
    synchronisation per call; the engine's flag word and match buffers are views of it)."""

    def __init__(self, eng, beams):
        dev = eng.device
        self.h_in = torch.zeros(6 + beams, dtype=torch.float64).pin_memory()
        self.d_in = torch.zeros(6 + beams, dtype=torch.float64, device=dev)
        self.d_out = torch.zeros(2 * MATCH_DOUBLES + 1, dtype=torch.float64, device=dev)
        self.h_out = torch.zeros(2 * MATCH_DOUBLES + 1, dtype=torch.float64).pin_memory()
        self.m_coarse = self.d_out[:MATCH_DOUBLES].view(1, MATCH_DOUBLES)
        self.m_fine = self.d_out[MATCH_DOUBLES:2 * MATCH_DOUBLES].view(1, MATCH_DOUBLES)
        eng.match_buf["coarse"], eng.match_buf["fine"] = self.m_coarse, self.m_fine
        eng.flags = self.d_out[2 * MATCH_DOUBLES:].view(torch.int32)[:1]         # (take_flags and the kernels use this word from now on)

    def download(self, eng):
        """Both levels' results + the fault word in one copy; raises on a fatal bit like ParticleEngine.take_flags."""
        self.h_out.copy_(self.d_out, non_blocking=True)
        torch.cuda.current_stream(eng.device).synchronize()
        host = self.h_out.numpy()
        bits = int(host[2 * MATCH_DOUBLES:].view(np.uint32)[0])
        if bits:
            eng.flags.zero_()
            if bits & _lib.FATAL_FLAGS:
                raise _lib.Slam2dError(f"particle 0: {_lib.describe_flags(bits & _lib.FATAL_FLAGS)}")
        m = host[:2 * MATCH_DOUBLES].view(_MATCH_DTYPE)
        return m[0], m[1]


def _call_buffers(eng, beams):
    io = getattr(eng, "_call_io", None)
    if io is None:
        io = eng._call_io = _CallBuffers(eng, beams)
    return io


class _Exclusive:
    """A shared workspace may serve one call at a time: the drop-in methods are synchronous (nothing is live between
    calls), and this guard turns any overlap -- two threads, a re-entrant callback -- into an error instead of silent
    corruption of another matcher's results."""

    def __init__(self, *levels):
        self.levels = levels

    def __enter__(self):
        for lv in self.levels:
            if getattr(lv, "_busy", False):
                raise _lib.Slam2dError("a ScanMatcher call is already using this configuration's shared workspace "
                                       "(the drop-in classes are synchronous and not re-entrant)")
        for lv in self.levels:
            lv._busy = True

    def __exit__(self, *exc):
        for lv in self.levels:
            lv._busy = False
        return False


def _shared_level(og, key_extra, **kw):
    """Workspaces are shared by all matchers of one configuration on one device:
    calls are synchronous and sequential, so nothing is live between calls (_Exclusive enforces it)."""
    key = (str(og.device), id(og.lidar)) + key_extra
    if key not in _level_cache:
        _level_cache[key] = SearchLevel(og.lidar, 1, og.device, bnb=False, **kw)     # whole cubes: brute-force sweep
    return _level_cache[key]


class ScanMatcher:
    def __init__(self, og, searchRadius, searchHalfRad, scanSigmaInNumGrid, moveRSigma, maxMoveDeviation, turnSigma,
                 missMatchProbAtCoarse, coarseFactor):
        self.searchRadius = searchRadius
        self.searchHalfRad = searchHalfRad
        self.og = og
        self.scanSigmaInNumGrid = scanSigmaInNumGrid
        self.coarseFactor = coarseFactor
        self.moveRSigma = moveRSigma
        self.turnSigma = turnSigma
        self.missMatchProbAtCoarse = missMatchProbAtCoarse
        self.maxMoveDeviation = maxMoveDeviation

    # ---- level plans ----
    def _level(self, step, sigma, miss, radius, half, fine):
        key = (step, sigma, miss, self.searchRadius, radius, half, bool(fine), self.moveRSigma, self.maxMoveDeviation,
               self.turnSigma)
        return _shared_level(self.og, key, step=step, sigma=sigma, miss_prob=miss,
                             search_radius_ctor=self.searchRadius, radius=radius, half_rad=half, fine=fine,
                             move_sigma=self.moveRSigma, max_move_dev=self.maxMoveDeviation,
                             turn_sigma=self.turnSigma)

    def coarse_level(self):
        step = self.coarseFactor * self.og.unitGridSize                        # :54
        sigma = self.scanSigmaInNumGrid / self.coarseFactor                    # :55
        return self._level(step, sigma, self.missMatchProbAtCoarse, self.searchRadius, self.searchHalfRad, False)

    def fine_level(self):
        step = self.og.unitGridSize                                            # :66
        miss = self.missMatchProbAtCoarse ** (2 / self.coarseFactor)           # :69
        cstep = self.coarseFactor * self.og.unitGridSize
        return self._level(step, self.scanSigmaInNumGrid, miss, cstep, self.searchHalfRad, True)

    # ---- the two halves, on the device ----
    def _build_field(self, eng, level, d_centre, stride, cx, cy):
        """checkAndExapndOG on the host (growth is a re-allocation), then the kernels."""
        self.og.checkAndExapndOG([cx - level.reach, cx + level.reach], [cy - level.reach, cy + level.reach])   # :27
        eng = self.og.engine()
        eng.field_build(level, d_centre, stride)
        return eng

    def frameSearchSpace(self, estimatedX, estimatedY, unitLength, sigma, missMatchProbAtCoarse):
        """(:20-39)  Returns xRangeList, yRangeList, probSP (host float64 array decoded from
        the device's fixed-point field: values exact to 2^-32 relative to probMin)."""
        level = self._level(unitLength, sigma, missMatchProbAtCoarse, self.searchRadius, self.searchHalfRad, False)
        with _Exclusive(level):
            eng = self.og.engine()
            d_c = eng.to_device([[estimatedX, estimatedY]])
            eng = self._build_field(eng, level, d_c, 2, estimatedX, estimatedY)
            self.last_flags = int(eng.take_flags()[0])
            fr = level.frames()[0]
            return [fr["xlo"], fr["xhi"]], [fr["ylo"], fr["yhi"]], level.field(0)

    def matchScan(self, reading, estMovingDist, estMovingTheta, count, matchMax=True):
        """(:47-79)  Coarse then fine; returns (matchedReading, coarse confidence)."""
        rMeasure = np.asarray(reading['range'])
        if count == 1:
            return reading, 1                                                  # :51-52
        ex, ey, eth = reading['x'], reading['y'], reading['theta']
        coarse, fine = self.coarse_level(), self.fine_level()
        with _Exclusive(coarse, fine), pinned_stream():
            og = self.og
            og.checkAndExapndOG([ex - coarse.reach, ex + coarse.reach], [ey - coarse.reach, ey + coarse.reach])   # :27 (coarse window)
            eng = og.engine()
            io = _call_buffers(eng, og.numSamplesPerRev)
            # everything the call uploads in ONE pinned buffer and one copy: [x, y, theta | cos, sin of the heading | uniform | ranges]
            h = io.h_in.numpy()
            h[0:3] = (ex, ey, eth)
            h[3:5] = eng.psi_table([estMovingTheta])[0]
            if not matchMax:                        # one draw from the legacy global stream, like np.random.choice (:138)
                h[5] = np.random.random_sample()
            h[6:] = rMeasure
            io.d_in.copy_(io.h_in, non_blocking=True)
            d_est, d_psi, d_u, d_rng = io.d_in[0:3], io.d_in[3:5], None if matchMax else io.d_in[5:6], io.d_in[6:]
            m_coarse, m_fine = io.m_coarse, io.m_fine
            # The fine window is centred on the coarse result, at most (ncell + 1) coarse cells from the estimate: when the coarse
            # window widened by that much lies inside the map, the fine window cannot need growth (:27 at the fine level) and the
            # call needs no host round trip between the levels -- one synchronisation per matchScan instead of two (round 3:
            # 0.86 ms per call).  Only the tiles the sweep reads are blurred (slam2d_match): this method hands no field out.
            margin = (coarse.ncell + 1) * coarse.step
            m = og.map
            settled = not (ex - coarse.reach - margin < m.lim_x[0] or ex + coarse.reach + margin > m.lim_x[1] or
                           ey - coarse.reach - margin < m.lim_y[0] or ey + coarse.reach + margin > m.lim_y[1])
            eng.match(coarse, d_est, 3, d_rng, estMovingDist, d_psi, d_u, m_coarse)
            if not settled:
                c = io.download(eng)[0]
                og.checkAndExapndOG([float(c["x"]) - fine.reach, float(c["x"]) + fine.reach],
                                    [float(c["y"]) - fine.reach, float(c["y"]) + fine.reach])
                eng = og.engine()
            eng.match(fine, m_coarse, MATCH_DOUBLES, d_rng, estMovingDist, None, None, m_fine)
            c, f = io.download(eng)
            og._update_pending = False                                            # (the download has looked at the last update's fault bits too)
            matched = {"x": float(f["x"]), "y": float(f["y"]), "theta": float(f["theta"]), "range": rMeasure}
            self.last = dict(coarse=c.copy(), fine=f.copy())
            # the matched pose and the scan's ranges are still on the device: an updateOccupancyGrid(matched) that follows on this
            # grid (Algorithm/FastSlam.py:129-133) launches from them -- no second upload (grid.updateOccupancyGrid)
            og._last_match = dict(ref=matched, pose=(matched["x"], matched["y"], matched["theta"]), rng=rMeasure, d_pose=m_fine, d_rng=d_rng, eng=eng)
            return matched, np.float64(c["confidence"])                            # :79

    def searchToMatch(self, probSP, estimatedX, estimatedY, estimatedTheta, rMeasure, xRangeList, yRangeList,
                      searchRadius, searchHalfRad, unitLength, estMovingDist, estMovingTheta, fineSearch=False,
                      matchMax=True):
        """(:91-151) on a caller-supplied field (re-quantised to the device's fixed-point format)."""
        rMeasure = np.asarray(rMeasure)
        level = self._level(unitLength, 1.0, 0.5, searchRadius, searchHalfRad, fineSearch)
        with _Exclusive(level):
            return self._search_to_match(level, probSP, estimatedX, estimatedY, estimatedTheta, rMeasure, xRangeList,
                                         yRangeList, estMovingDist, estMovingTheta, matchMax)

    def _search_to_match(self, level, probSP, estimatedX, estimatedY, estimatedTheta, rMeasure, xRangeList, yRangeList,
                         estMovingDist, estMovingTheta, matchMax):
        eng = self.og.engine()
        fh, fw = probSP.shape
        if fh > level.fmax or fw > level.fmax:
            raise ValueError("field larger than this matcher's search window")
        level.set_field(probSP, 0)
        fr = level.frames()
        fr[0]["xlo"], fr[0]["xhi"], fr[0]["ylo"], fr[0]["yhi"] = xRangeList[0], xRangeList[1], yRangeList[0], yRangeList[1]
        fr[0]["fh"], fr[0]["fw"] = fh, fw
        level.t["frames"].copy_(torch.from_numpy(fr.view(np.uint8).reshape(1, -1)))
        d_est = eng.to_device([[estimatedX, estimatedY, estimatedTheta]])
        d_rng = eng.to_device(rMeasure)
        d_psi = eng.to_device(eng.psi_table([estMovingTheta]))
        d_u = None if matchMax else eng.to_device([np.random.random_sample()])
        out = eng.match_buffer("adhoc")
        eng.sweep(level, d_est, 3, d_rng, estMovingDist, d_psi, d_u, out)
        eng.take_flags()
        m = eng.read_matches(out)[0]
        matched = {"x": float(m["x"]), "y": float(m["y"]), "theta": float(m["theta"]), "range": rMeasure}
        px, py = self.covertMeasureToXY(estimatedX, estimatedY, estimatedTheta, rMeasure)
        dth = matched["theta"] - estimatedTheta
        mpx, mpy = self.rotate((estimatedX, estimatedY), (px, py), dth)
        dx, dy = matched["x"] - estimatedX, matched["y"] - estimatedY
        self.last = dict(adhoc=m.copy())
        return mpx + dx, mpy + dy, matched, level.cube(0), np.float64(m["confidence"])

    # ---- small host helpers of the reference's surface ----
    def covertMeasureToXY(self, estimatedX, estimatedY, estimatedTheta, rMeasure):      # :81-89
        og = self.og
        rads = np.linspace(estimatedTheta - og.lidarFOV / 2, estimatedTheta + og.lidarFOV / 2, num=og.numSamplesPerRev)
        keep = rMeasure < og.lidarMaxRange
        return estimatedX + np.cos(rads[keep]) * rMeasure[keep], estimatedY + np.sin(rads[keep]) * rMeasure[keep]

    def rotate(self, origin, point, angle):                                             # :162-171
        ox, oy = origin
        px, py = point
        return (ox + np.cos(angle) * (px - ox) - np.sin(angle) * (py - oy),
                oy + np.sin(angle) * (px - ox) + np.cos(angle) * (py - oy))

    def convertXYToSearchSpaceIdx(self, px, py, beginX, beginY, unitLength):            # :173-176
        return (((px - beginX) / unitLength)).astype(int), (((py - beginY) / unitLength)).astype(int)

    def __deepcopy__(self, memo):
        new = ScanMatcher.__new__(ScanMatcher)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, copy.deepcopy(v, memo))       # og: clones the device map, alias preserved via memo
        return new
