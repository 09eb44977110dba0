"""``OccupancyGrid`` with the reference's class surface (Utils/OccupancyGrid.py:6-175),
backed by a device-resident packed count map and the HIP update kernel.

Constructor arguments, method names, attribute names and semantics follow the
reference so that ``Algorithm/FastSlam.py`` and the single-trajectory driver run on
it unchanged.  ``occupancyGridVisited`` / ``occupancyGridTotal`` are properties that
download the counts as float64 NumPy arrays (the reference stores float64 arrays).
"""
import copy

import numpy as np
import torch

from . import _lib
from .engine import MATCH_DOUBLES, LidarModel, MapState, ParticleEngine, pinned_stream, require_gpu

DEFAULT_DEVICE = "cuda:0"


class OccupancyGrid:
    def __init__(self, mapXLength, mapYLength, initXY, unitGridSize, lidarFOV, numSamplesPerRev, lidarMaxRange,
                 wallThickness, device=None):
        self.device = require_gpu(device or DEFAULT_DEVICE)
        self.unitGridSize = unitGridSize
        self.lidarFOV = lidarFOV
        self.lidarMaxRange = lidarMaxRange
        self.wallThickness = wallThickness
        self.numSamplesPerRev = numSamplesPerRev
        self.lidar = LidarModel.get(unitGridSize, lidarMaxRange, lidarFOV, numSamplesPerRev, wallThickness)
        self.angularStep = self.lidar.angular_step                 # Utils/OccupancyGrid.py:22
        self.numSpokes = self.lidar.num_spokes                     # :23
        self.spokesStartIdx = self.lidar.spoke_start               # :30
        self.map = MapState.create(mapXLength, mapYLength, initXY, unitGridSize, self.device)
        self.version = 0            # bumped whenever the device map is re-allocated
        self._engine = None

    # ---- attributes of the reference ----
    @property
    def mapXLim(self):
        return self.map.lim_x

    @property
    def mapYLim(self):
        return self.map.lim_y

    @property
    def occupancyGridVisited(self):
        self.flush()
        return self.map.download()[0]

    @occupancyGridVisited.setter
    def occupancyGridVisited(self, value):
        self.map.upload(value, self.map.download()[1])

    @property
    def occupancyGridTotal(self):
        self.flush()
        return self.map.download()[1]

    @occupancyGridTotal.setter
    def occupancyGridTotal(self, value):
        self.map.upload(self.map.download()[0], value)

    def set_counts(self, visited, total):
        """Upload both count arrays at once (tests / synthetic worlds)."""
        self.map.upload(visited, total)

    @property
    def OccupancyGridX(self):
        return np.meshgrid(self.map.X, self.map.Y)[0]

    @property
    def OccupancyGridY(self):
        return np.meshgrid(self.map.X, self.map.Y)[1]

    # ---- engine (one particle) ----
    def engine(self):
        if self._engine is None:
            self._engine = ParticleEngine(self.lidar, [self.map], self.device)
            self._engine_version = self.version
            from .matcher import _call_buffers              # the engine's fault word and match buffers live in ONE download buffer
            _call_buffers(self._engine, self.numSamplesPerRev)
        elif self._engine_version != self.version or self._engine.maps[0] is not self.map:
            self._engine.maps = [self.map]
            self._engine.refresh_maps()
            self._engine_version = self.version
        return self._engine

    # ---- index conversion and growth (Utils/OccupancyGrid.py:59-125) ----
    def convertRealXYToMapIdx(self, x, y):
        return self.map.to_map_idx(x, y, self.unitGridSize)

    def checkMapToExpand(self, x, y):
        return self.map._side_to_grow(x, y)

    def expandOccupancyGrid(self, expandDirection):
        self.map._grow(expandDirection if expandDirection in (1, 2, 3) else 4, self.unitGridSize)
        self.version += 1

    def checkAndExapndOG(self, x, y):
        before = len(self.map.growth_log)
        shift = self.map.ensure_contains(x, y, self.unitGridSize)
        if len(self.map.growth_log) != before:
            self.version += 1
        return shift

    # ---- per-scan update (Utils/OccupancyGrid.py:127-159) ----
    def _beam_cells(self, theta, rng):
        return self.lidar.beam_cells(theta, rng)

    def _grow_for_update(self, x, y, theta, rng):
        """Per-beam growth exactly in the reference's order (:147), returning the
        [beams, 2] low-side shifts that make the kernel reproduce its stale-index
        writes (:144-152), or None if nothing grew."""
        before = len(self.map.growth_log)
        shifts = self.lidar.grow_for_update(self.map, x, y, theta, rng)
        if len(self.map.growth_log) != before:
            self.version += 1
        return shifts

    def updateOccupancyGrid(self, reading, dTheta=0, update=True):
        x, y, theta = reading['x'], reading['y'], reading['theta'] + dTheta
        rng = np.asarray(reading['range'], dtype=np.float64)
        if not update:
            return self._update_points(x, y, theta, rng)
        R = self.lidarMaxRange
        m = self.map
        shifts = None
        if x - R < m.lim_x[0] or x + R > m.lim_x[1] or y - R < m.lim_y[0] or y + R > m.lim_y[1]:
            shifts = self._grow_for_update(x, y, theta, rng)      # rare: first scan of a small map
        eng = self.engine()
        lm, self._last_match = getattr(self, "_last_match", None), None
        if (shifts is None and lm is not None and lm["ref"] is reading and dTheta == 0 and lm["eng"] is eng and reading['range'] is lm["rng"]
                and (x, y, theta) == lm["pose"]):
            # the very dict this grid's matcher has just returned, untouched: its pose (the first doubles of the fine match) and the
            # scan's ranges are on the device already -- one launch, nothing copied (0.075 -> 0.0x ms of host time per call, round 6)
            with pinned_stream():
                eng.grid_update(lm["d_pose"], MATCH_DOUBLES, lm["d_rng"], None)
            self._update_pending = True
            return
        if shifts is not None:                                      # (rare: synchronous, with its own uploads)
            eng.grid_update(eng.to_device([[x, y, theta]]), 3, eng.to_device(rng), eng.to_device(shifts[None], dtype=np.int32))
            eng.take_flags()
            return
        # pose + ranges in one pinned buffer and one copy; the launch is NOT waited for: the reference's method returns nothing,
        # and everything that reads the map afterwards is ordered behind it on the stream.  Its fault bits are looked at by the next
        # synchronising call on this grid (matchScan's download, a count download, flush()): round 3 paid 0.08 ms per call here.
        io = self._update_io(eng, len(rng))
        io["ev"].synchronize()                                      # the previous call's copy has left the pinned buffer
        h = io["h"].numpy()
        h[0:3] = (x, y, theta)
        h[3:] = rng
        with pinned_stream():
            io["d"].copy_(io["h"], non_blocking=True)
            io["ev"].record()
            eng.grid_update(io["d"][0:3], 3, io["d"][3:], None)
        self._update_pending = True

    def _update_io(self, eng, beams):
        io = getattr(eng, "_update_io", None)
        if io is None or io["h"].numel() != 3 + beams:
            io = eng._update_io = dict(h=torch.zeros(3 + beams, dtype=torch.float64).pin_memory(),
                                       d=torch.zeros(3 + beams, dtype=torch.float64, device=self.device), ev=torch.cuda.Event())
        return io

    def flush(self):
        """Wait for the last updateOccupancyGrid and raise on a fault it flagged (a cell outside the map, a count overflow)."""
        if getattr(self, "_update_pending", False):
            self._update_pending = False
            if self._engine is not None:
                self._engine.take_flags()

    def _update_points(self, x, y, theta, rng):
        """update=False variant (:153-159): world coordinates of the empty / occupied cells."""
        W, xs = self.lidar.width, self.lidar.xs
        ex, ey, ox, oy = [], [], [], []
        for _, c, empty, occ in self._beam_cells(theta, rng):
            ex.extend(x + xs[c[empty] % W]); ey.extend(y + xs[c[empty] // W])
            ox.extend(x + xs[c[occ] % W]); oy.extend(y + xs[c[occ] // W])
        return np.asarray(ex), np.asarray(ey), np.asarray(ox), np.asarray(oy)

    def mapImage(self, xRange, yRange, as_u8=False):
        """``np.flipud(1 - (visited / total)[yIdx[0]:yIdx[1], xIdx[0]:xIdx[1]])`` -- what plotOccupancyGrid (:168-170) and the
        FastSLAM driver (Algorithm/FastSlam.py:171-177) draw -- computed on the device (slam2d_map_image)."""
        self.flush()                                                # (a fault of the last update must not pass unseen into a picture)
        xIdx, yIdx = self.convertRealXYToMapIdx(xRange, yRange)
        return self.map.image(xIdx[0], xIdx[1], yIdx[0], yIdx[1], flipud=True, as_u8=as_u8).cpu().numpy()

    # ---- plotting (host Matplotlib, off the hot path; Utils/OccupancyGrid.py:161-175) ----
    def plotOccupancyGrid(self, xRange=None, yRange=None, plotThreshold=True):
        import matplotlib.pyplot as plt
        if xRange is None or xRange[0] < self.mapXLim[0] or xRange[1] > self.mapXLim[1]:
            xRange = self.mapXLim
        if yRange is None or yRange[0] < self.mapYLim[0] or yRange[1] > self.mapYLim[1]:
            yRange = self.mapYLim
        img = self.mapImage(xRange, yRange)
        extent = [xRange[0], xRange[1], yRange[0], yRange[1]]
        plt.imshow(img, cmap='gray', extent=extent)
        plt.show()
        if plotThreshold:
            plt.matshow(img >= 0.5, cmap='gray', extent=extent)
            plt.show()

    # ---- copy.deepcopy support (Algorithm/FastSlam.py:58,61) ----
    def __deepcopy__(self, memo):
        self.flush()                                                # the copy has no engine to ask: faults are raised here, on the original
        new = OccupancyGrid.__new__(OccupancyGrid)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("map", "_engine", "lidar", "device", "_last_match"):
                continue
            setattr(new, k, copy.deepcopy(v, memo))
        new.device, new.lidar = self.device, self.lidar
        if self._engine is not None:
            self._engine.sync_bounds()
        new.map = self.map.clone()
        new._engine = None
        new._last_match = None
        new._update_pending = False
        new.version = 0
        return new
