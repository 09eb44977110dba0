"""Host side of the HIP path: device-resident particle maps, search-level plans and
the calls into libslam2d_hip.so.

Everything numeric that decides a cell index is either done on the device in fp64
with the reference's operation order, or prepared here once with NumPy scalars
(Gaussian taps, theta tables, the spoke LUT) so it is bit-identical to what the
reference's NumPy computes.  PyTorch is used only as the device allocator, for
host<->device copies and for the stream handle.

``file:line`` citations are relative to the reference repository root.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import Slam2dFrame, Slam2dLevel, Slam2dLidar, Slam2dMap, Slam2dMatch, Slam2dPartial, check

MATCH_DOUBLES = C.sizeof(Slam2dMatch) // 8      # stride of a Slam2dMatch array viewed as double*
_MATCH_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("theta", "f8"), ("confidence", "f8"),
                         ("log_confidence", "f8"), ("best_score", "f8"), ("pick", "i4"), ("argmax", "i4")])
_FRAME_DTYPE = np.dtype([("xlo", "f8"), ("ylo", "f8"), ("xhi", "f8"), ("yhi", "f8"), ("cx", "f8"), ("cy", "f8"),
                         ("field_min", "f8"), ("fh", "i4"), ("fw", "i4"), ("mx0", "i4"), ("mx1", "i4"),
                         ("my0", "i4"), ("my1", "i4"), ("redo", "i4"), ("min_known", "i4"), ("field_max", "f8")])
assert _MATCH_DTYPE.itemsize == C.sizeof(Slam2dMatch) and _FRAME_DTYPE.itemsize == C.sizeof(Slam2dFrame)


def require_gpu(device):
    """The HIP path needs the library and a GPU; there is no CPU fallback."""
    _lib.lib()
    if not torch.cuda.is_available():
        raise _lib.Slam2dError("no HIP device visible: the scan-matching path runs on an MI355X only "
                               "(there is no CPU fallback)")
    return torch.device(device)


_PINNED_STREAM = None


def _stream():
    """The current torch stream's handle.  torch.cuda.current_stream() costs ~8 us of host time per call, twenty
    times per scan: hot loops pin the handle for their duration (``with pinned_stream(): ...``)."""
    if _PINNED_STREAM is not None:
        return _PINNED_STREAM
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class pinned_stream:
    """Context manager: resolve the current stream once; every library call inside uses that handle."""

    def __enter__(self):
        global _PINNED_STREAM
        self._prev = _PINNED_STREAM
        _PINNED_STREAM = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        _PINNED_STREAM = self._prev
        return False


class _on_launch_stream:
    """Context for the few torch operations of the launch path (occupancy-stamp reset, index upload of the bit refresh): when
    the library calls are pinned to a stream other than torch's current one, torch must enqueue there too."""

    def __enter__(self):
        self._ctx = None
        pinned = None if _PINNED_STREAM is None else (_PINNED_STREAM.value or 0)          # (c_void_p(0).value is None)
        if pinned is not None and pinned != (torch.cuda.current_stream().cuda_stream or 0):
            self._ctx = torch.cuda.stream(torch.cuda.ExternalStream(pinned))
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


_GROUP_STREAMS = {}


def group_streams(device, n):
    """``n`` HIP streams for particle groups (+ the normaliser's), created ONCE per device -- the first call makes a batch of
    nine through the library (slam2d_streams_create: created and first used one after the other, so that they land on distinct
    hardware queues) -- and handed out again to every later caller.  Which hardware queue a stream sits on decides how well the
    groups overlap, and streams taken one by one out of torch's round-robin pool of 32 end up sharing queues: the twelfth set of
    three fresh streams in one process measured 0.146 ms per step against 0.130 for the first eleven (round 4), the closed loop in
    four groups 0.42-0.50 s against 0.21 s when two of its streams were taken from the pool later than the others (round 5)."""
    dev = torch.device(device)
    key = str(dev)
    if n > 3 and "GPU_MAX_HW_QUEUES" not in os.environ and not _GROUP_STREAMS:
        # more than two groups (+ the normaliser's stream) need more than the runtime's default of 4 hardware queues, and the
        # runtime reads the variable when it initialises: this is the last place where this library can still set it -- if no HIP
        # call has been made yet.  Otherwise the application has to (before importing torch.cuda work): say so, once.
        if not torch.cuda.is_initialized():
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
        else:
            import warnings
            warnings.warn("slam2d: %d particle-group streams on the HIP runtime's default of 4 hardware queues (HIP is already initialised): "
                          "groups that share a queue take turns; set GPU_MAX_HW_QUEUES=8 before the process's first HIP call" % n, RuntimeWarning, stacklevel=2)
    have = _GROUP_STREAMS.setdefault(key, [])
    while len(have) < n:
        k = max(9 if not have else 0, n - len(have))
        out = (C.c_void_p * k)()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().slam2d_streams_create(out, k), "slam2d_streams_create")
        have.extend(torch.cuda.ExternalStream(int(p), device=dev) for p in out)
    return have[:n]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


# ----------------------------------------------------------------------------
# Gaussian taps / analytic field floor
# ----------------------------------------------------------------------------
def gaussian_taps(sigma):
    """Normalised taps of SciPy's gaussian_filter (truncate 4.0) as the reference
    calls it (Utils/ScanMatcher_OGBased.py:42)."""
    sd = float(sigma)
    radius = int(4.0 * sd + 0.5)
    x = np.arange(-radius, radius + 1)
    w = np.exp(-0.5 / (sd * sd) * x ** 2)
    return w / w.sum(), radius


def blurred_free_value(log_miss, taps, radius):
    """Value the separable blur gives a cell whose whole neighbourhood is free, in the
    device kernel's (= SciPy's) operation order.  It is the field minimum whenever
    such a cell exists; k_floor_check verifies that on the device."""
    L = np.float64(log_miss)

    def one_pass(v):
        acc = v * taps[radius]
        for j in range(-radius, 0):
            acc = acc + (v + v) * taps[radius + j]
        return acc
    return float(one_pass(one_pass(L)))


def cost_scale_for(min_value):
    """2^k with -min_value * 2^k < 2^32 (k <= 31): the fixed-point scale of a search field
    whose values lie in [min_value, 0] (include/slam2d.h, "Search field format")."""
    m = abs(float(min_value))
    if not m > 0.0 or math.isinf(m) or math.isnan(m):
        return float(2 ** 31)
    k = min(31, int(math.floor(math.log2(4294967295.0 / m))))
    return float(2.0 ** k)


def bnb_default(nx, ntheta, beams, tmax):
    """Whether slam2d_match scores a level by branch and bound (include/slam2d.h): it pays once the cube is
    large enough that 1/16 of the gathers + a few per cent of the tiles + three more launches beat the brute-force
    sweep.  SLAM2D_BNB=0 / 1 forces it off / on wherever it is applicable (cube edge 9..64, <= 20000 field tiles)."""
    import os
    ok = 9 <= nx <= 64
    env = os.environ.get("SLAM2D_BNB", "auto")
    if env == "0" or not ok:
        return False
    if env == "1":
        return True
    return ntheta * nx * nx * beams >= BNB_MIN_WORK


BNB_MIN_WORK = 4_000_000      # poses x beams per particle-scan; config 2: 10.9 M, reference coarse: 3.9 M, fine: 0.65 M


def g2b_pitch(gp):
    """Row pitch (bytes) of Slam2dLevel.gmin2b for rows of gp entries: a multiple of 16 whose dword stride mod 32 puts the tile
    rows a lane group of k_bound_lds reads on distinct LDS banks (include/slam2d.h)."""
    pitch = -(-gp // 16) * 16
    while (pitch // 4) % 32 not in (8, 12, 20, 24):
        pitch += 16
    return pitch


def encode_cost(prob, scale):
    """probSP (values in [min, 0]) -> uint32 fixed-point cost, as the blur kernel stores it."""
    c = np.rint(-np.asarray(prob, dtype=np.float64) * scale)
    if c.min() < 0 or c.max() > 4294967295:
        raise ValueError("search field values must lie in [min, 0] with -min * scale < 2^32")
    return c.astype(np.uint32)


# ----------------------------------------------------------------------------
# Lidar model + polar spoke LUT (Utils/OccupancyGrid.py:22-57)
# ----------------------------------------------------------------------------
class LidarModel:
    """Lidar parameters and the cell-major spoke lookup table shared by all
    particles of one configuration."""

    _cache = {}

    @classmethod
    def get(cls, unit, max_range, fov, beams, wall_thickness):
        key = (float(unit), float(max_range), float(fov), int(beams), float(wall_thickness))
        if key not in cls._cache:
            cls._cache[key] = cls(*key)
        return cls._cache[key]

    def __init__(self, unit, max_range, fov, beams, wall_thickness):
        if not 2 <= beams <= _lib.MAX_BEAMS:
            raise ValueError(f"numSamplesPerRev must be in [2, {_lib.MAX_BEAMS}]")
        self.unit, self.max_range, self.fov, self.beams = unit, max_range, fov, int(beams)
        self.wall_thickness = wall_thickness
        self.angular_step = fov / beams                                           # :22
        self.num_spokes = int(np.rint(2 * np.pi / self.angular_step))             # :23
        self.spoke_start = int(((self.num_spokes / 2 - beams) / 2) % self.num_spokes)   # :30
        half = int(max_range / unit)                                              # :34
        self.half, self.width = half, 2 * half + 1
        self.xs = np.linspace(-max_range, max_range, self.width)                  # :36
        self.bin, self.r = self._build()
        self._dev = {}

    def _build(self):
        """spokesGrid (:32-45), cell-major: spoke bin and radius of each window cell."""
        S, h, W = self.num_spokes, self.half, self.width
        x = self.xs[None, :]
        y = self.xs[:, None]
        bins = np.zeros((W, W), dtype=np.int64)
        with np.errstate(divide="ignore", invalid="ignore"):
            ang = np.arctan(y / x[:, h + 1:])
            bins[:, h + 1:] = np.rint((np.pi / 2 + ang) / np.pi / 2 * S - 0.5).astype(int)
        bins[:, :h] = bins[::-1, ::-1][:, :h] + int(S / 2)        # point mirror, half a turn on
        bins[h + 1:, h] = int(S / 2)                              # upper half of the centre column
        r = np.sqrt(x ** 2 + y ** 2)
        return bins.astype(np.uint16), np.ascontiguousarray(r)

    def xs_step(self):
        """Step of the window coordinates if np.linspace's arithmetic (j * step + start, last element = stop)
        reproduces them bit for bit -- the update kernel then computes them instead of gathering; else 0."""
        W, R = self.width, float(self.max_range)
        if W < 2:
            return 0.0
        step = (R - (-R)) / (W - 1)
        y = np.arange(W, dtype=np.float64) * step + (-R)
        y[-1] = R
        return float(step) if np.array_equal(y, self.xs) else 0.0

    def spoke_lists(self):
        """itemizeSpokesGrid (:47-57) beam-major for the update kernel: the cells of a spoke ordered by
        radial band (SPOKE_BAND values of floor(r / unit)) and row-major inside a band.  Returns
        (band_ptr int32 [S, nb + 1] absolute start of every band, cells uint32 [W*W] = row << 16 | column,
        radius float64 [W*W], nb)."""
        W, S = self.width, self.num_spokes
        assert W <= 65535
        flat_bin, flat_r = self.bin.ravel().astype(np.int64), self.r.ravel()
        band = np.floor(flat_r / self.unit).astype(np.int64) // _lib.SPOKE_BAND
        nb = int(band.max()) + 1
        order = np.lexsort((np.arange(W * W), band, flat_bin))
        counts = np.bincount(flat_bin * nb + band, minlength=S * nb).reshape(S, nb)
        starts = np.concatenate(([0], np.cumsum(counts.ravel())))
        band_ptr = np.empty((S, nb + 1), dtype=np.int32)
        band_ptr[:, :nb] = starts[:-1].reshape(S, nb)
        band_ptr[:, nb] = starts[nb::nb]
        cells = ((order // W).astype(np.uint32) << np.uint32(16)) | (order % W).astype(np.uint32)
        return band_ptr, cells, np.ascontiguousarray(flat_r[order]), nb

    # -- host mirror of the per-beam walk (Utils/OccupancyGrid.py:133-147), used only where the reference
    #    grows the map INSIDE an update (first scan of a small map) and by the update=False variant --
    def _spoke_lists_host(self):
        if not hasattr(self, "_spoke_ptr"):
            flat = self.bin.ravel().astype(np.int64)
            self._spoke_cells = np.argsort(flat, kind="stable")
            self._spoke_ptr = np.concatenate(([0], np.cumsum(np.bincount(flat, minlength=self.num_spokes))))
        return self._spoke_cells, self._spoke_ptr

    def beam_cells(self, theta, rng):
        """Per beam: (beam index, flat LUT cells of its spoke, empty mask, occupied mask), reference order."""
        cells, ptr = self._spoke_lists_host()
        S = self.num_spokes
        offset = int(np.rint(theta / (2 * np.pi) * S))                        # :131
        r_flat = self.r.ravel()
        half_w = self.wall_thickness / 2
        for i in range(self.beams):
            spoke = int(np.rint((self.spoke_start + offset + i) % S))         # :134
            c = cells[ptr[spoke]:ptr[spoke + 1]]
            rs = r_flat[c]
            empty = rs < rng[i] - half_w if rng[i] < self.max_range else np.zeros(rs.shape, dtype=bool)   # :138-141
            occ = (rs > rng[i] - half_w) & (rs < rng[i] + half_w)             # :142-143
            yield i, c, empty, occ

    def grow_for_update(self, m, x, y, theta, rng):
        """Per-beam growth of MapState ``m`` exactly in the reference's order (:147).  The reference computes a beam's
        cell indices BEFORE that beam's growth and writes with them afterwards (:144-152): a low-side growth inside
        beam i leaves its indices stale by the inserted block, a negative stale index wraps (Python semantics) against
        the array as it is at that moment, and later low-side growths move what was written.  Returns, for
        k_grid_update, int32 [beams, 6] = (dc_i, dr_i: low-side shift of beam i's own growth; ac_i, ar_i: sum of the
        low-side shifts of all LATER beams; cols_i, rows_i: map shape right after beam i's growth) -- or None if
        nothing grew.  A cell with final index (mx, my) is then written at wrap(mx - dc_i - ac_i, cols_i) + ac_i."""
        W, xs = self.width, self.xs
        rec = np.zeros((self.beams, 6), dtype=np.int32)
        grew = False
        with m.deferred_growth():          # the per-beam growth steps are planned on the host, the array follows once
            for i, c, _, occ in self.beam_cells(theta, rng):
                if occ.any():
                    before = len(m.growth_log)
                    dc, dr = m.ensure_contains(x + xs[c[occ] % W], y + xs[c[occ] // W], self.unit)
                    rec[i, 0:2] = (dc, dr)
                    grew |= len(m.growth_log) != before
                rec[i, 4:6] = (m.cols, m.rows)
        if not grew:
            return None
        # low-side shifts of the beams after i
        rec[:, 2] = np.concatenate((np.cumsum(rec[::-1, 0])[::-1][1:], [0]))
        rec[:, 3] = np.concatenate((np.cumsum(rec[::-1, 1])[::-1][1:], [0]))
        return rec

    def on(self, device):
        """Device copies + the C struct (kept alive with the tensors)."""
        key = str(device)
        if key not in self._dev:
            ptr, cells, radii, nb = self.spoke_lists()
            t = dict(xs=_dev(self.xs, device), sptr=_dev(ptr, device), scells=_dev(cells.view(np.int32), device),
                     sr=_dev(radii, device))
            s = Slam2dLidar(unit=self.unit, max_range=self.max_range, fov=self.fov,
                            wall_half=self.wall_thickness / 2, beams=self.beams, num_spokes=self.num_spokes,
                            spoke_start=self.spoke_start, lut_w=self.width, lut_xs=t["xs"].data_ptr(),
                            spoke_band=t["sptr"].data_ptr(), spoke_cells=t["scells"].data_ptr(),
                            spoke_r=t["sr"].data_ptr(), num_bands=nb, lut_xs_step=self.xs_step())
            self._dev[key] = (s, t)
        return self._dev[key][0]


# ----------------------------------------------------------------------------
# One particle's map (Utils/OccupancyGrid.py:7-20, 59-125)
# ----------------------------------------------------------------------------
class MapState:
    """Device-resident count map of one particle + host copies of its coordinate
    vectors and limits.  Cells are uint32 (visited << 16 | total), stored in an
    int32 tensor."""

    PITCH_ALIGN = 16
    layout_version = 0      # (class default: maps assembled field by field -- resample's gather, clone -- start at 0 too)

    def __init__(self, X, Y, device, cells=None):
        self.device = device
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.rows, self.cols = len(self.Y), len(self.X)
        self.pitch = -(-self.cols // self.PITCH_ALIGN) * self.PITCH_ALIGN
        fresh = cells is None
        if fresh:
            cells = torch.full((self.rows, self.pitch), _lib.INIT_CELL, dtype=torch.int32, device=device)
        self.cells = cells
        self.wide = cells.dtype == torch.int64          # 64-bit cells (visited << 32 | total): counts beyond 16 bits
        # upper bound of any cell's `total` (2 initially, every update adds at most 2): the engine promotes a map to wide
        # cells before an update could overflow a narrow one.  Cells handed in from elsewhere: unknown, assume the worst
        self.count_bound = 2 if fresh else (2 ** 31 if self.wide else _lib.COUNT_LIMIT)
        self._pending, self._defer = None, False
        self.layout_version = 0          # bumped whenever `cells` / `bits` are re-allocated (growth, promotion): descriptors on the device are stale then
        self._alloc_bits()
        self._sync_coords()
        self.growth_log = []

    @classmethod
    def create(cls, mapXLength, mapYLength, initXY, unit, device):
        """Initial extent exactly as the reference's constructor (:8-14), including its
        use of xNum for both axes; count arrays are (xNum+1, yNum+1) there, which is
        only consistent for square maps -- rectangular maps are rejected here."""
        xNum, yNum = int(mapXLength / unit), int(mapYLength / unit)
        if xNum != yNum:
            raise ValueError("the reference's OccupancyGrid is only self-consistent for square maps")
        X = np.linspace(-xNum * unit / 2, xNum * unit / 2, num=xNum + 1) + initXY['x']
        Y = np.linspace(-xNum * unit / 2, xNum * unit / 2, num=yNum + 1) + initXY['y']
        return cls(X, Y, device)

    def _alloc_bits(self):
        """1 bit per cell (occupied), [rows][bits_pitch] words; rebuilt on the device by
        ParticleEngine.refresh_bits() whenever the host wrote `cells` (bits_valid False)."""
        self.bits_pitch = -(-self.cols // 32)
        self.bits = torch.zeros((self.rows, self.bits_pitch), dtype=torch.int32, device=self.device)
        self.bits_valid = False

    def _sync_coords(self):
        self.lim_x = [self.X[0], self.X[-1]]           # mapXLim (:19, :86-87)
        self.lim_y = [self.Y[0], self.Y[-1]]           # mapYLim (:20, :88-89)
        xy = _dev(np.concatenate((self.X, self.Y)), self.device)          # (one upload)
        self.dX, self.dY = xy[:len(self.X)], xy[len(self.X):]

    def desc(self):
        self._materialise()
        return Slam2dMap(cells=self.cells.data_ptr(), X=self.dX.data_ptr(), Y=self.dY.data_ptr(),
                         rows=self.rows, cols=self.cols, pitch=self.pitch, bits_pitch=self.bits_pitch,
                         lim_x0=self.lim_x[0], lim_x1=self.lim_x[1], lim_y0=self.lim_y[0], lim_y1=self.lim_y[1],
                         occ_bits=self.bits.data_ptr(), wide=1 if self.wide else 0)

    def promote(self):
        """To 64-bit cells (visited << 32 | total), the format of a map whose counts may exceed 16 bits (Slam2dMap.wide).
        The reference's float64 counts never saturate (Utils/OccupancyGrid.py:148-152); this is how a long stationary log
        keeps running.  The occupancy bits are unaffected."""
        if self.wide:
            return
        self._materialise()
        with _on_launch_stream():           # (ordered with the launches of a pinned group stream)
            c = self.cells.to(torch.int64)
            self.cells = (((c >> 16) & 0xFFFF) << 32) | (c & 0xFFFF)
        self.wide = True
        self.layout_version += 1            # another array in another format: every uploaded descriptor of this map is stale

    # -- growth: expandOccupancyGridHelper (:59-89) --
    def _side_to_grow(self, x, y):                                            # :108-118
        x, y = np.asarray(x), np.asarray(y)
        if np.any(x < self.lim_x[0]):
            return 1
        if np.any(x > self.lim_x[1]):
            return 2
        if np.any(y < self.lim_y[0]):
            return 3
        if np.any(y > self.lim_y[1]):
            return 4
        return -1

    def _plan_grow(self, side, unit):
        """One 20 % growth step on one side, host bookkeeping only: coordinate vectors, limits, shape, growth log.  The count
        array follows in ONE re-allocation + block copy when the growth sequence is complete (``_materialise``): the first
        update of a small map grows it seventeen times (CSAIL), and every step used to allocate and copy the whole map.
        Returns (d_col, d_row): how far existing content moves (non-zero only for low-side growth)."""
        if self._pending is None:
            self._pending = [self.cells, self.rows, self.cols, 0, 0]       # the array as it was + accumulated low-side shift
        rows, cols = self.rows, self.cols
        if side in (1, 2):
            n = int(cols / 5)                                                 # :72
            if side == 1:      # low side: exact spacing (:75-76)
                new = np.linspace(self.lim_x[0] - n * unit, self.lim_x[0], num=n, endpoint=False)
                self.X = np.concatenate((new, self.X))
                shift = (n, 0)
            else:              # high side: spacing unit*(n-1)/n, as the reference (:79-80)
                new = np.linspace(self.lim_x[1] + unit, self.lim_x[1] + n * unit, num=n, endpoint=False)
                self.X = np.concatenate((self.X, new))
                shift = (0, 0)
        else:
            n = int(rows / 5)                                                 # :62
            if side == 3:
                new = np.linspace(self.lim_y[0] - n * unit, self.lim_y[0], num=n, endpoint=False)
                self.Y = np.concatenate((new, self.Y))
                shift = (0, n)
            else:
                new = np.linspace(self.lim_y[1] + unit, self.lim_y[1] + n * unit, num=n, endpoint=False)
                self.Y = np.concatenate((self.Y, new))
                shift = (0, 0)
        self.rows, self.cols = len(self.Y), len(self.X)
        self.lim_x = [self.X[0], self.X[-1]]           # mapXLim (:86-87)
        self.lim_y = [self.Y[0], self.Y[-1]]           # mapYLim (:88-89)
        self._pending[3] += shift[0]
        self._pending[4] += shift[1]
        self.growth_log.append((side, n))
        return shift

    def _materialise(self):
        """The planned growth on the device: one fresh array of the final extent, the old content copied to its place."""
        if self._pending is None:
            return
        old, rows, cols, dc, dr = self._pending
        self._pending = None
        old_desc = Slam2dMap(cells=old.data_ptr(), rows=rows, cols=cols, pitch=self.pitch, wide=1 if self.wide else 0)
        self.pitch = -(-self.cols // self.PITCH_ALIGN) * self.PITCH_ALIGN
        with _on_launch_stream():
            # one pass on the device writes the new counts (old content shifted, fresh cells elsewhere) AND the new occupancy bits
            # (slam2d_map_grow; round 3: fill + strided copy + zeroed bits + a refresh pass over the whole map, ~100 us per map)
            self.cells = torch.empty((self.rows, self.pitch), dtype=old.dtype, device=self.device)
            self.bits_pitch = -(-self.cols // 32)
            self.bits = torch.empty((self.rows, self.bits_pitch), dtype=torch.int32, device=self.device)
            self._sync_coords()
            new_desc = Slam2dMap(cells=self.cells.data_ptr(), occ_bits=self.bits.data_ptr(), rows=self.rows, cols=self.cols,
                                 pitch=self.pitch, bits_pitch=self.bits_pitch, wide=1 if self.wide else 0)
            _lib.check(_lib.lib().slam2d_map_grow(C.byref(old_desc), C.byref(new_desc), dr, dc, _stream()), "slam2d_map_grow")
            self.bits_valid = True
        self.layout_version += 1

    def _grow(self, side, unit):
        """One growth step, on the device at once (expandOccupancyGrid)."""
        shift = self._plan_grow(side, unit)
        if not self._defer:
            self._materialise()
        return shift

    def deferred_growth(self):
        """Context manager: growth steps inside are planned on the host and materialised once at the end."""
        m = self

        class _Ctx:
            def __enter__(self):
                self.prev, m._defer = m._defer, True
                return m

            def __exit__(self, *exc):
                m._defer = self.prev
                if not m._defer:
                    m._materialise()
                return False
        return _Ctx()

    def ensure_contains(self, x, y, unit):
        """checkAndExapndOG (:120-125).  Returns the total (d_col, d_row) content shift."""
        dc = dr = 0
        side = self._side_to_grow(x, y)
        while side != -1:
            s = self._plan_grow(side, unit)
            dc += s[0]; dr += s[1]
            side = self._side_to_grow(x, y)
        if not self._defer:
            self._materialise()
        return dc, dr

    def to_map_idx(self, x, y, unit):                                         # :102-106
        xi = np.rint((np.asarray(x) - self.lim_x[0]) / unit).astype(int)
        yi = np.rint((np.asarray(y) - self.lim_y[0]) / unit).astype(int)
        return xi, yi

    def image(self, x0, x1, y0, y1, flipud=True, as_u8=False):
        """``np.flipud(1 - (visited / total)[y0:y1, x0:x1])`` (Algorithm/FastSlam.py:173-176) computed on the device
        (slam2d_map_image): a float64 (or uint8, rint(255 v)) tensor [y1 - y0, x1 - x0] on the map's device.  Python slice
        semantics for the bounds (negative indices, clipping)."""
        self._materialise()
        xa, xb, _ = slice(int(x0), int(x1)).indices(self.cols)
        ya, yb, _ = slice(int(y0), int(y1)).indices(self.rows)
        h, w = max(0, yb - ya), max(0, xb - xa)
        out = torch.empty((h, w), dtype=torch.uint8 if as_u8 else torch.float64, device=self.device)
        if h == 0 or w == 0:
            return out
        d = upload_map_descs([self], self.device)
        check(_lib.lib().slam2d_map_image(_ptr(d), 0, xa, xb, ya, yb, 1 if flipud else 0, None if as_u8 else _ptr(out),
                                          _ptr(out) if as_u8 else None, _stream()), "slam2d_map_image")
        torch.cuda.current_stream(self.device).synchronize()      # the descriptor upload must outlive the kernel
        return out

    def download(self):
        """(visited, total) as float64 host arrays, like the reference's attributes."""
        self._materialise()
        if self.wide:
            raw = self.cells[:, :self.cols].cpu().numpy().view(np.uint64)
            return (raw >> np.uint64(32)).astype(np.float64), (raw & np.uint64(0xFFFFFFFF)).astype(np.float64)
        raw = self.cells[:, :self.cols].cpu().numpy().view(np.uint32)
        return (raw >> np.uint32(16)).astype(np.float64), (raw & np.uint32(0xFFFF)).astype(np.float64)

    def upload(self, visited, total):
        self._materialise()
        v = np.asarray(visited)
        t = np.asarray(total)
        if v.shape != (self.rows, self.cols) or t.shape != v.shape:
            raise ValueError("count arrays do not match the map shape")
        if (v < 0).any() or (t < 0).any() or max(v.max(), t.max()) >= 2 ** 31 or \
                not (np.array_equal(v, np.rint(v)) and np.array_equal(t, np.rint(t))):
            raise ValueError("counts must be integers in [0, 2^31)")
        if max(v.max(), t.max()) > _lib.COUNT_LIMIT:
            self.promote()
        if self.wide:
            packed = (v.astype(np.uint64) << np.uint64(32)) | t.astype(np.uint64)
            self.cells[:, :self.cols] = torch.from_numpy(packed.view(np.int64)).to(self.device)
        else:
            packed = (v.astype(np.uint32) << np.uint32(16)) | t.astype(np.uint32)
            self.cells[:, :self.cols] = torch.from_numpy(packed.view(np.int32)).to(self.device)
        self.count_bound = max(int(t.max()), 2)
        self.bits_valid = False

    def clone(self):
        self._materialise()                 # (first: a pending growth changes pitch, rows and cols)
        m = MapState.__new__(MapState)
        m.device = self.device
        m.X, m.Y = self.X.copy(), self.Y.copy()
        m.rows, m.cols, m.pitch = self.rows, self.cols, self.pitch
        m.cells = self.cells.clone()
        m._pending, m._defer = None, False
        m.layout_version = 0
        m.wide, m.count_bound = self.wide, self.count_bound
        m.bits_pitch, m.bits, m.bits_valid = self.bits_pitch, self.bits.clone(), self.bits_valid
        m._sync_coords()
        m.growth_log = list(self.growth_log)
        return m


def upload_map_descs(maps, device):
    arr = (Slam2dMap * len(maps))(*[m.desc() for m in maps])
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


# ----------------------------------------------------------------------------
# One search level (Utils/ScanMatcher_OGBased.py:53-60 coarse, :65-73 fine)
# ----------------------------------------------------------------------------
class SearchLevel:
    """Parameters and device workspaces of one level for P particles."""

    def __init__(self, lidar, P, device, step, sigma, miss_prob, search_radius_ctor, radius, half_rad, fine,
                 move_sigma, max_move_dev, turn_sigma, bnb=None):
        self.lidar, self.P, self.device = lidar, P, device
        self.step, self.sigma, self.miss_prob, self.fine = step, sigma, miss_prob, bool(fine)
        self.radius, self.half_rad = radius, half_rad
        self.reach = 1.1 * lidar.max_range + search_radius_ctor                   # :21
        self.taps, self.blur_radius = gaussian_taps(sigma)
        if self.blur_radius > _lib.MAX_BLUR_RADIUS:
            raise ValueError(f"blur radius {self.blur_radius} exceeds {_lib.MAX_BLUR_RADIUS}")
        self.log_miss = math.log(miss_prob)                                       # :25
        self.floor_value = blurred_free_value(self.log_miss, self.taps, self.blur_radius)
        self.cost_scale = cost_scale_for(self.floor_value)
        self.ncell = int(radius / step)                                           # :94
        self.nx = 2 * self.ncell + 1
        self.thetas = np.arange(-half_rad, half_rad + lidar.angular_step, lidar.angular_step)   # :114
        self.ntheta = len(self.thetas)
        self.fmax = int(2 * self.reach / step) + 2
        self.fpitch = -(-self.fmax // 16) * 16
        self.wmax = int(2 * self.reach / lidar.unit) + 3
        self.kmax = lidar.beams
        self.rv_coef = -(1 / (2 * move_sigma ** 2))                               # :101
        self.tw_coef = -1 / (2 * turn_sigma ** 2)                                 # :108
        self.max_move_dev = max_move_dev
        npose = self.nx * self.nx
        self.npartial = self.ntheta * (-(-npose // 64))
        self.tmax = -(-self.fmax // 16)             # 16x16-cell tiles of the blur
        nbt_ = (self.nx + 3) // 4
        # limits of k_exact_select: one thread scans <= 32 tile bounds, per-theta sums live in LDS (256 thetas)
        applicable = 9 <= self.nx <= 64 and self.ntheta <= 256 and self.ntheta * nbt_ * 4 * ((nbt_ + 3) // 4) <= 32768
        self.bnb = (bnb_default(self.nx, self.ntheta, lidar.beams, self.tmax) if bnb is None else bool(bnb)) and applicable
        # two-level bounds pay where one load instruction per cell and theta is the bottleneck (k_bound at its texture-path
        # bound: ~1000-cell lists); SLAM2D_BNB_LEVELS=1 / 2 forces the choice
        import os
        lv_env = os.environ.get("SLAM2D_BNB_LEVELS", "auto")
        # (round 6: with the bound image staged in LDS -- k_bound_lds, where it fits in 64 KB beside the run words -- one level of
        # bounds beats two at 1081 beams as well: config 5 1.203 -> 1.130 ms per 128-particle step)
        gp_ = 4 * self.tmax
        lds_one_level = os.environ.get("SLAM2D_BOUND_LDS", "1") != "0" and gp_ * g2b_pitch(gp_) + nbt_ * g2b_pitch(gp_) + 16 < 65536
        self.bnb_levels = 0 if not self.bnb else (int(lv_env) if lv_env in ("1", "2") else (2 if lidar.beams >= 512 and not lds_one_level else 1))
        if self.bnb_levels == 2 and self.nx < 17:
            self.bnb_levels = 1
        # angle bounds (Slam2dLevel.bnb == 3): a cube of <= 5 x 5 poses per angle is too small for pose tiles, but behind a long
        # cell list one gmin2 entry per cell bounds the angle's whole plane -- most angles are then never scored.  Pays where
        # the exact sweep of the plane is expensive, i.e. ~1000-cell lists; SLAM2D_ABOUND=0 / 1 forces it off / on
        ab_env = os.environ.get("SLAM2D_ABOUND", "auto")
        self.abound = (not self.bnb and self.nx <= 5 and self.ntheta < (1 << 14) and bnb is not False and ab_env != "0"
                       and (ab_env == "1" or bnb is True or lidar.beams >= 512))
        if self.abound:
            self.bnb_levels = 3
        # angles per k_endpoints block (the block dilates and stores its tile marks once): SLAM2D_EP_GROUP overrides
        g_env = os.environ.get("SLAM2D_EP_GROUP", "")
        self.ep_group = int(g_env) if g_env.isdigit() and int(g_env) >= 1 else (4 if lidar.beams >= 512 else 2)      # measured: 1081 beams 189 / 173 / 163 / 165 us for 1 / 2 / 4 / 8, 180 beams 22.4 / 18.7 / 19.6 for 1 / 2 / 4
        nbt = (self.nx + 3) // 4
        nbq4 = 4 * ((nbt + 3) // 4)
        i32, f64 = torch.int32, torch.float64
        t = self.t = dict(
            blur_w=_dev(self.taps, device),
            thetas=_dev(self.thetas, device),
            # scalar np.cos / np.sin per angle, as the reference's rotate() sees them (:169-170)
            cos=_dev(np.array([np.cos(a) for a in self.thetas]), device),
            sin=_dev(np.array([np.sin(a) for a in self.thetas]), device),
            frames=torch.zeros((P, C.sizeof(Slam2dFrame)), dtype=torch.uint8, device=device),
            axis_x=torch.zeros((P, self.wmax), dtype=i32, device=device),
            axis_y=torch.zeros((P, self.wmax), dtype=i32, device=device),
            # occupied-cell image of every particle, followed by the flags of its 8 x 8-cell blocks (generation-stamped bytes)
            occ=torch.zeros(P * self.fmax * self.fpitch + P * 2 * self.tmax * ((2 * self.tmax + 17) & ~15), dtype=torch.uint8, device=device),
            field=torch.zeros((P, self.fmax, self.fpitch), dtype=i32, device=device),     # uint32 costs
            cells=torch.zeros((P, self.ntheta, self.kmax), dtype=i32, device=device),
            kcount=torch.zeros((P, self.ntheta), dtype=i32, device=device),
            prior=torch.zeros((P, 2, npose), dtype=f64, device=device),
            cube=torch.zeros((P, self.ntheta, npose), dtype=f64, device=device),
            partials=torch.zeros((P, self.npartial, C.sizeof(Slam2dPartial)), dtype=torch.uint8, device=device),
            tilestate=torch.ones((P, self.tmax, self.tmax), dtype=torch.uint8, device=device),   # all dirty
            tilemin=torch.zeros((P, self.tmax, self.tmax), dtype=f64, device=device),
            tilemax=torch.zeros((P, self.tmax, self.tmax), dtype=f64, device=device),
            tilelist=torch.zeros((P, 2, self.tmax * self.tmax), dtype=i32, device=device),
            tilecount=torch.zeros((P, 2), dtype=i32, device=device),
            tileneed=torch.zeros((P, -(-self.ntheta // self.ep_group), (self.tmax * self.tmax + 31) // 32), dtype=i32, device=device),   # one slice per group of angles
            freerow=torch.zeros((P, 64), dtype=torch.int64, device=device),
            ring=torch.zeros(1 + self.nx * ((self.nx + 3) // 4), dtype=i32, device=device),
            prune_state=torch.zeros(P, dtype=i32, device=device),
            beam_xy=torch.zeros((P, lidar.beams, 2), dtype=f64, device=device),
            sync=torch.zeros((P, _lib.SYNC_WORDS), dtype=i32, device=device),       # arrival counters (zero between launches)
        )
        if self.abound:     # angle bounds: block minima of the field, cell offsets into them, one bound per (particle, theta)
            t.update(
                gmin=torch.zeros((P, 4 * self.tmax, 4 * self.tmax), dtype=i32, device=device),
                gmin2=torch.zeros((P, 4 * self.tmax, 4 * self.tmax), dtype=i32, device=device),
                pcells=torch.zeros((P, self.ntheta, self.kmax), dtype=i32, device=device),
                bounds=torch.zeros((P, self.ntheta), dtype=f64, device=device),
                bnb_best=torch.zeros(P, dtype=torch.int64, device=device),
                seed_key=torch.zeros(P, dtype=torch.int64, device=device),
            )
        if self.bnb:        # branch and bound over 4x4 pose tiles (include/slam2d.h)
            t.update(
                gmin=torch.zeros((P, 4 * self.tmax, 4 * self.tmax), dtype=i32, device=device),
                gmin2=torch.zeros((P, 4 * self.tmax, 4 * self.tmax), dtype=i32, device=device),
                pcells=torch.zeros((P, self.ntheta, self.kmax), dtype=i32, device=device),
                bounds=torch.zeros((P, self.ntheta, nbt, nbq4), dtype=f64, device=device),
                tile_pmax=torch.zeros((P, nbt, nbq4), dtype=f64, device=device),
                bnb_best=torch.zeros(P, dtype=torch.int64, device=device),
            )
            if self.bnb_levels == 1:    # the bounds' byte image for k_bound_lds (one block per particle stages it in LDS)
                self.g2b_pitch = g2b_pitch(4 * self.tmax)
                t.update(gmin2b=torch.zeros((P, 4 * self.tmax, self.g2b_pitch), dtype=torch.uint8, device=device),
                         theta_umax=torch.zeros((P, self.ntheta), dtype=f64, device=device))
            if self.bnb_levels == 2:    # long cell lists: 8x8-pose tiles first
                t.update(
                    gmin3d=torch.zeros((P, 4, 2 * self.tmax, 2 * self.tmax), dtype=i32, device=device),
                    p3cells=torch.zeros((P, self.ntheta, self.kmax), dtype=i32, device=device),
                    bounds1=torch.zeros((P, self.ntheta, 64), dtype=f64, device=device),
                    seed_key=torch.zeros(P, dtype=torch.int64, device=device),
                )
        self.c = Slam2dLevel(
            step=step, reach=self.reach, log_miss=self.log_miss, floor_value=self.floor_value,
            cost_scale=self.cost_scale, blur_radius=self.blur_radius, fmax=self.fmax, fpitch=self.fpitch, wmax=self.wmax,
            blur_w=t["blur_w"].data_ptr(), ncell=self.ncell, ntheta=self.ntheta, fine=int(self.fine),
            kmax=self.kmax, thetas=t["thetas"].data_ptr(), theta_cos=t["cos"].data_ptr(),
            theta_sin=t["sin"].data_ptr(), rv_coef=self.rv_coef, tw_coef=self.tw_coef,
            max_move_dev=max_move_dev, frames=t["frames"].data_ptr(), axis_x=t["axis_x"].data_ptr(),
            axis_y=t["axis_y"].data_ptr(), occ=t["occ"].data_ptr(), field=t["field"].data_ptr(),
            cells=t["cells"].data_ptr(), kcount=t["kcount"].data_ptr(), prior=t["prior"].data_ptr(),
            cube=t["cube"].data_ptr(), partials=t["partials"].data_ptr(), npartial=self.npartial, tmax=self.tmax,
            tilemask=t["occ"].data_ptr() + P * self.fmax * self.fpitch, tilestate=t["tilestate"].data_ptr(),
            tilemin=t["tilemin"].data_ptr(), tilemax=t["tilemax"].data_ptr(), tilelist=t["tilelist"].data_ptr(),
            tilecount=t["tilecount"].data_ptr(),
            tileneed=t["tileneed"].data_ptr(), freerow=t["freerow"].data_ptr(), ring=t["ring"].data_ptr(), prune_state=t["prune_state"].data_ptr(),
            ring_cap=self.nx * ((self.nx + 3) // 4), bnb=self.bnb_levels, ep_group=self.ep_group, beam_xy=t["beam_xy"].data_ptr(),
            sync=t["sync"].data_ptr(),
            **({k: t[k].data_ptr() for k in ("gmin3d", "p3cells", "bounds1", "seed_key")} if self.bnb_levels == 2 else {}),
            **(dict(gmin2b=t["gmin2b"].data_ptr(), g2b_pitch=self.g2b_pitch, theta_umax=t["theta_umax"].data_ptr()) if "gmin2b" in t else {}),
            **({k: t[k].data_ptr() for k in ("gmin", "gmin2", "pcells", "bounds", "bnb_best", "seed_key")} if self.abound else {}),
            **({k: t[k].data_ptr() for k in ("gmin", "gmin2", "pcells", "bounds", "tile_pmax", "bnb_best")} if self.bnb else {}))

    _PER_PARTICLE = ("frames", "axis_x", "axis_y", "field", "cells", "kcount", "prior", "cube", "partials", "tilestate", "tilemin",
                     "tilemax", "tilelist", "tilecount", "tileneed", "freerow", "prune_state", "beam_xy", "sync", "gmin", "gmin2",
                     "pcells", "bounds", "tile_pmax", "bnb_best", "gmin3d", "p3cells", "bounds1", "seed_key", "gmin2b", "theta_umax")

    def view(self, p0, p1):
        """A Slam2dLevel describing particles [p0, p1) of this level: the same parameters, every per-particle pointer advanced to
        particle p0 (the C ABI takes base pointers, so a group of particles is an offset view of every array; include/slam2d.h,
        slam2d_groups_*).  The ring of the prior pruning is written per call and therefore the view's own.  The view's
        generation stamp follows the parent's: call ``sync_view(view)`` after ``next_generation()``."""
        v = Slam2dLevel.from_buffer_copy(self.c)
        for k in self._PER_PARTICLE:
            tz = self.t.get(k)
            if tz is not None and getattr(self.c, k):
                setattr(v, k, tz.data_ptr() + p0 * tz.stride(0) * tz.element_size())
        img = self.fmax * self.fpitch
        fb = 2 * self.tmax * ((2 * self.tmax + 17) & ~15)
        v.occ = self.t["occ"].data_ptr() + p0 * img
        v.tilemask = self.t["occ"].data_ptr() + self.P * img + p0 * fb
        ring = torch.zeros_like(self.t["ring"])
        self._view_rings = getattr(self, "_view_rings", []) + [ring]
        v.ring = ring.data_ptr()
        return v

    def sync_view(self, v):
        v.occ_gen = self.c.occ_gen

    def next_generation(self):
        """Advance the occupancy-image generation stamp (Slam2dLevel.occ_gen) for the next build: the
        image is zeroed once per 254 builds instead of at every build."""
        g = self.c.occ_gen + 1
        if g > 254:
            with _on_launch_stream():
                self.t["occ"].zero_()
            g = 1
        self.c.occ_gen = g

    # -- results --
    def frames(self):
        return self.t["frames"].cpu().numpy().view(_FRAME_DTYPE).reshape(-1)

    def field_cost(self, p=0):
        """Fixed-point cost image of particle p, uint32 [fh, fw]."""
        fr = self.frames()[p]
        return self.t["field"][p, :fr["fh"], :fr["fw"]].cpu().numpy().view(np.uint32)

    def field(self, p=0):
        """probSP of particle p as a float64 host array [fh, fw] (= -cost / cost_scale)."""
        return -(self.field_cost(p).astype(np.float64) / self.c.cost_scale)

    def set_field(self, prob, p=0):
        """Load a caller-supplied probSP for particle p (re-quantised at a scale fitted to it)."""
        prob = np.asarray(prob, dtype=np.float64)
        scale = cost_scale_for(prob.min())
        cost = encode_cost(prob, scale)
        self.c.cost_scale = scale
        fh, fw = prob.shape
        self.t["field"][p, :fh, :fw] = torch.from_numpy(cost.view(np.int32)).to(self.device)
        self.t["tilestate"][p].fill_(1)          # the buffer no longer holds what field_build left there
        self.t["freerow"][p].zero_()             # ... and no tile of it is known to hold the constant
        return scale

    def bnb_stats(self):
        """Diagnostics of the last slam2d_match: mean tiles per particle in the two work lists (blur, fill) and,
        with branch and bound, the fraction of pose tiles that were scored exactly."""
        tc = self.t["tilecount"].cpu().numpy().astype(np.float64).mean(axis=0)
        out = dict(blur_tiles=tc[0], fill_tiles=tc[1])
        if self.bnb:
            nbt = (self.nx + 3) // 4
            b = self.t["bounds"].cpu().numpy()[..., :nbt]
            best = self.t["bnb_best"].cpu().numpy().view(np.uint64)
            bits = np.where(best >> np.uint64(63), best & np.uint64(0x7FFFFFFFFFFFFFFF), ~best).astype(np.uint64)
            m0 = bits.view(np.float64)
            kept = b >= (m0 - _lib.BNB_MARGIN)[:, None, None, None]
            out.update(kept_fraction=float(kept.mean()), kept_per_particle=float(kept.reshape(len(m0), -1).sum(axis=1).mean()),
                       kept_max_per_theta=int(kept.reshape(len(m0), self.ntheta, -1).sum(axis=2).max()))
        return out

    def cube(self, p=0):
        return self.t["cube"][p].cpu().numpy().reshape(self.ntheta, self.nx, self.nx)

    def cells_of(self, p, it):
        k = int(self.t["kcount"][p, it].item())
        off = self.t["cells"][p, it, :k].cpu().numpy()
        return off // self.fpitch + self.ncell, off % self.fpitch + self.ncell      # (cy, cx)

    def algorithmic_bytes(self, n_window_cells=None):
        """SURVEY.md 8(d) per particle-scan at this level: field build
        2*c*Wm^2 + 4*Fh*Fw (c = 2 bytes per count: packed uint32 cell holds both),
        sweep 4*Fh*Fw + 8*B + 8*Ntheta*Ny*Nx (the cube is float64 here)."""
        wm = int(2 * self.reach / self.lidar.unit) if n_window_cells is None else n_window_cells
        f = (self.fmax - 1) ** 2
        field_build = 2 * 2 * wm * wm + 4 * f
        sweep = 4 * f + 8 * self.lidar.beams + 8 * self.ntheta * self.nx * self.nx
        return dict(field_build=field_build, sweep=sweep)


# ----------------------------------------------------------------------------
# The engine: P particles, their maps, two search levels, the kernel calls
# ----------------------------------------------------------------------------
class ParticleEngine:
    """Batched scan matching + map update for P particles on one GPU."""

    def __init__(self, lidar, maps, device):
        self.device = require_gpu(device)
        self.L = _lib.lib()
        self.lidar = lidar
        self.lidar_c = lidar.on(self.device)
        self.maps = list(maps)
        self.P = len(self.maps)
        self.flags = torch.zeros(self.P, dtype=torch.int32, device=self.device)
        self.match_buf = {}
        self.refresh_maps()

    def sync_bounds(self):
        """Credit the updates launched since the last call to the maps that received them (every update adds at most 2 to a
        cell's `total`): to be called before maps are copied, replaced or handed to somebody else."""
        n = getattr(self, "_pending_updates", 0)
        if n:
            for m in self._bound_maps:
                m.count_bound += 2 * n
            self._pending_updates = 0

    def _rebound(self):
        self._bound_maps = list(self.maps)
        # the largest bound among the maps that can still overflow (a wide map cannot: one promoted map must not make every
        # later scan re-check -- and re-upload -- the others)
        self._max_bound = max([m.count_bound for m in self.maps if not m.wide], default=0)

    def _before_update(self):
        """Move every map that the coming update could overflow (16-bit counts) to 64-bit cells first (MapState.promote) --
        the reference's float64 counts never saturate.  O(1) per scan until then."""
        self._pending_updates += 1
        if self._max_bound + 2 * self._pending_updates > _lib.COUNT_LIMIT:
            self.sync_bounds()
            promoted = False
            for m in self.maps:                     # (sync_bounds has credited the coming update as well)
                if not m.wide and m.count_bound > _lib.COUNT_LIMIT:
                    m.promote()
                    promoted = True
            if promoted:
                self.refresh_maps()                 # (new arrays, new cell format: new descriptors)
            else:
                self._rebound()

    def refresh_maps(self):
        self.sync_bounds()
        self._pending_updates = 0
        self._rebound()
        with _on_launch_stream():
            self.d_maps = upload_map_descs(self.maps, self.device)
        self._layouts = [(id(m), m.layout_version) for m in self.maps]
        self.maps_version = getattr(self, "maps_version", 0) + 1      # limits / descriptors changed (growth, resample)
        self.refresh_bits()

    def refresh_bits(self):
        """Rebuild the occupancy bits of every map the host has written since the last build -- and upload fresh descriptors
        first when a map's arrays were re-allocated behind the engine's back (a growth or a promotion to 64-bit cells through
        the map's own methods: OccupancyGrid.set_counts beyond 16 bits, checkAndExapndOG)."""
        stale, moved = [], False
        for i, (m, seen) in enumerate(zip(self.maps, self._layouts)):
            if not m.bits_valid:
                stale.append(i)
            if seen[1] != m.layout_version or seen[0] != id(m):
                moved = True
        if moved or len(self.maps) != len(self._layouts):
            for m in self.maps:
                m._materialise()
            return self.refresh_maps()
        if not stale:
            return
        self.sync_bounds()                 # the host wrote these maps (upload, growth, copy): their count bounds may have changed
        self._rebound()
        with _on_launch_stream():
            idx = _dev(np.asarray(stale, dtype=np.int32), self.device)
            check(self.L.slam2d_map_refresh_bits(_ptr(self.d_maps), _ptr(idx), len(stale), _stream()),
                  "slam2d_map_refresh_bits")
        for i in stale:
            self.maps[i].bits_valid = True

    def match_buffer(self, name):
        if name not in self.match_buf:
            self.match_buf[name] = torch.zeros((self.P, MATCH_DOUBLES), dtype=torch.float64, device=self.device)
        return self.match_buf[name]

    @staticmethod
    def read_matches(buf):
        return buf.cpu().numpy().view(_MATCH_DTYPE).reshape(-1)

    # -- kernels --
    def field_build(self, level, d_centre, stride):
        self.refresh_bits()
        level.next_generation()
        check(self.L.slam2d_field_build(C.byref(self.lidar_c), C.byref(level.c), _ptr(self.d_maps), self.P,
                                        _ptr(d_centre), stride, _ptr(self.flags), _stream()), "slam2d_field_build")

    def sweep(self, level, d_est, stride, d_ranges, est_moving_dist, d_psi_cs, d_uniform, d_out):
        check(self.L.slam2d_sweep(C.byref(self.lidar_c), C.byref(level.c), self.P, _ptr(d_est), stride,
                                  _ptr(d_ranges), float(est_moving_dist), _ptr(d_psi_cs), _ptr(d_uniform),
                                  _ptr(d_out), _ptr(self.flags), _stream()), "slam2d_sweep")

    def match(self, level, d_est, stride, d_ranges, est_moving_dist, d_psi_cs, d_uniform, d_out, prune=False):
        """field_build + sweep of one level in one call, blurring only the field tiles the sweep reads
        (slam2d_match): same matches and cube, level.field() is left incomplete.  prune: score the poses
        inside the motion prior's ring first and skip the rest when they cannot matter
        (SLAM2D_MATCH_PRUNE_BY_PRIOR; coarse level only; level.cube() then holds only the ring)."""
        self.refresh_bits()
        level.next_generation()
        check(self.L.slam2d_match(C.byref(self.lidar_c), C.byref(level.c), _ptr(self.d_maps), self.P, _ptr(d_est),
                                  stride, _ptr(d_ranges), float(est_moving_dist), _ptr(d_psi_cs), _ptr(d_uniform),
                                  _ptr(d_out), _ptr(self.flags), _lib.MATCH_PRUNE_BY_PRIOR if prune else 0, _stream()),
              "slam2d_match")

    def grid_update(self, d_pose, stride, d_ranges, d_beam_shift=None):
        self._before_update()
        self.refresh_bits()
        check(self.L.slam2d_grid_update(C.byref(self.lidar_c), _ptr(self.d_maps), self.P, _ptr(d_pose), stride,
                                        _ptr(d_ranges), _ptr(d_beam_shift),
                                        _ptr(self.flags), _stream()), "slam2d_grid_update")
        if d_beam_shift is not None:                   # the stale-index path does not keep the occupancy bits in step
            for m in self.maps:
                m.bits_valid = False

    def grid_update_weights(self, d_pose, stride, d_ranges, d_logw, logconf_ptr, logconf_stride, d_w, d_stats):
        """Map update + (weight *= confidence, normalise) of all particles in one launch (slam2d_grid_update_weights)."""
        self._before_update()
        self.refresh_bits()
        check(self.L.slam2d_grid_update_weights(C.byref(self.lidar_c), _ptr(self.d_maps), self.P, _ptr(d_pose), stride,
                                                _ptr(d_ranges), _ptr(self.flags), _ptr(d_logw), C.c_void_p(logconf_ptr),
                                                logconf_stride, _ptr(d_w), _ptr(d_stats), _stream()), "slam2d_grid_update_weights")

    def grid_update_weights_local(self, d_pose, stride, d_ranges, d_logw, logconf_ptr, logconf_stride, d_part, normalizer=None):
        """Map update + the rank-local half of the sharded normaliser in one launch (slam2d_grid_update_weights_local);
        the all-gather and slam2d_weights_merge follow (parallel.ShardedNormalizer(..., local_done=True)).  ``normalizer``:
        that ShardedNormalizer (``d_part`` should be its ``part``): the launch is ordered behind its previous, possibly
        still overlapped, merge first -- the launch writes the log-weights and the partials that merge works on."""
        if normalizer is not None:
            normalizer.pre_local()
        self._before_update()
        self.refresh_bits()
        check(self.L.slam2d_grid_update_weights_local(C.byref(self.lidar_c), _ptr(self.d_maps), self.P, _ptr(d_pose), stride,
                                                      _ptr(d_ranges), _ptr(self.flags), _ptr(d_logw), C.c_void_p(logconf_ptr),
                                                      logconf_stride, _ptr(d_part), _stream()), "slam2d_grid_update_weights_local")

    def take_flags(self, fatal=_lib.FATAL_FLAGS):
        """Synchronise, fetch and clear the per-particle fault bits; raise on fatal ones."""
        f = self.flags.cpu().numpy().view(np.uint32).copy()
        self.flags.zero_()
        bad = np.flatnonzero(f & fatal)
        if bad.size:
            p = int(bad[0])
            raise _lib.Slam2dError(f"particle {p}: {_lib.describe_flags(int(f[p]) & fatal)}")
        return f

    # -- uploads --
    def to_device(self, a, dtype=np.float64):
        return _dev(np.asarray(a, dtype=dtype), self.device)

    @staticmethod
    def psi_table(psi):
        """(math.cos, math.sin) per particle of estMovingTheta; NaN rows for None
        (Utils/ScanMatcher_OGBased.py:104-107)."""
        out = np.full((len(psi), 2), np.nan)
        for i, v in enumerate(psi):
            if v is not None and not (isinstance(v, float) and math.isnan(v)):
                out[i] = (math.cos(v), math.sin(v))
        return out
