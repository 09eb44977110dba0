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
from .engine import MATCH_DOUBLES, ParticleEngine, SearchLevel

_level_cache = {}


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
        with _Exclusive(coarse, fine):
            eng = self.og.engine()
            d_est = eng.to_device([[ex, ey, eth]])
            d_rng = eng.to_device(rMeasure)
            d_psi = eng.to_device(eng.psi_table([estMovingTheta]))
            d_u = None
            if not matchMax:                        # one draw from the legacy global stream, like np.random.choice (:138)
                d_u = eng.to_device([np.random.random_sample()])
            m_coarse, m_fine = eng.match_buffer("coarse"), eng.match_buffer("fine")
            eng = self._build_field(eng, coarse, d_est, 3, ex, ey)
            eng.sweep(coarse, d_est, 3, d_rng, estMovingDist, d_psi, d_u, m_coarse)
            eng.take_flags()
            c = eng.read_matches(m_coarse)[0]
            eng = self._build_field(eng, fine, m_coarse, MATCH_DOUBLES, float(c["x"]), float(c["y"]))
            eng.sweep(fine, m_coarse, MATCH_DOUBLES, d_rng, estMovingDist, None, None, m_fine)
            eng.take_flags()
            f = eng.read_matches(m_fine)[0]
            matched = {"x": float(f["x"]), "y": float(f["y"]), "theta": float(f["theta"]), "range": rMeasure}
            self.last = dict(coarse=c.copy(), fine=f.copy())
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
