"""Seeded synthetic worlds and lidar scans for the benchmark configurations.

BASELINE.json's configs 2-5 are synthetic parametrisations of the reference's
functions: a rectilinear world (outer wall, corridors,
rectangular obstacles), ranges ray-cast from it, and a seeded random-walk
trajectory.  Pure NumPy, host only; used by ``bench.py``, the tests and the
golden-vector generator.  Nothing here is on the timed path.
"""
import numpy as np


def make_world(size_m, unit, seed=0, n_boxes=40, wall_cells=2):
    """Boolean occupancy image [n, n] of a ``size_m`` x ``size_m`` world at
    ``unit`` metres per cell: a closed outer wall, two corridor walls with
    door gaps, and ``n_boxes`` seeded rectangular obstacles (outlines only,
    like a lidar-mapped building)."""
    rs = np.random.RandomState(seed)
    n = int(round(size_m / unit)) + 1
    w = np.zeros((n, n), dtype=bool)
    t = max(1, int(wall_cells))
    m = max(t + 1, int(0.04 * n))                 # margin of the outer wall
    w[m:m + t, m:n - m] = True
    w[n - m - t:n - m, m:n - m] = True
    w[m:n - m, m:m + t] = True
    w[m:n - m, n - m - t:n - m] = True
    # corridor walls with doors
    for frac in (0.38, 0.62):
        r = int(frac * n)
        w[r:r + t, m:n - m] = True
        for _ in range(6):
            c = rs.randint(m + 5, n - m - 5 - int(1.2 / unit))
            w[r:r + t, c:c + int(1.2 / unit)] = False
    # box outlines
    for _ in range(n_boxes):
        h = rs.randint(int(0.6 / unit), int(4.0 / unit) + 2)
        ww = rs.randint(int(0.6 / unit), int(4.0 / unit) + 2)
        r0 = rs.randint(m + t + 1, n - m - t - h - 1)
        c0 = rs.randint(m + t + 1, n - m - t - ww - 1)
        w[r0:r0 + t, c0:c0 + ww] = True
        w[r0 + h - t:r0 + h, c0:c0 + ww] = True
        w[r0:r0 + h, c0:c0 + t] = True
        w[r0:r0 + h, c0 + ww - t:c0 + ww] = True
    # keep the centre free so a robot can stand there
    c = n // 2
    k = int(1.5 / unit)
    w[c - k:c + k, c - k:c + k] = False
    return w


def raycast(world, unit, origin_xy, pose, fov, beams, max_range, no_return=None):
    """Ranges of ``beams`` rays over ``fov`` centred on pose heading, beam
    angles as the matcher draws them (linspace, Utils/ScanMatcher_OGBased.py:82).
    ``origin_xy`` is the world coordinate of cell [0, 0].  A ray that leaves
    the world or exceeds ``max_range`` returns ``no_return`` (default
    1.5 * max_range, like a max-range reading)."""
    x, y, th = pose
    if no_return is None:
        no_return = 1.5 * max_range
    ang = np.linspace(th - fov / 2, th + fov / 2, beams)
    ds = unit * 0.5
    steps = np.arange(ds, max_range, ds)
    px = x + np.outer(np.cos(ang), steps)
    py = y + np.outer(np.sin(ang), steps)
    ci = np.rint((px - origin_xy[0]) / unit).astype(np.int64)
    ri = np.rint((py - origin_xy[1]) / unit).astype(np.int64)
    n = world.shape[0]
    inside = (ci >= 0) & (ci < n) & (ri >= 0) & (ri < n)
    hit = np.zeros(px.shape, dtype=bool)
    hit[inside] = world[ri[inside], ci[inside]]
    first = np.argmax(hit, axis=1)
    any_hit = hit.any(axis=1)
    # quantise like a real sensor log (cm)
    return np.where(any_hit, np.round(steps[first], 2), no_return)


def free_pose_near(world, unit, origin_xy, rs, centre=None, spread=3.0):
    """A seeded pose in free space near ``centre`` (default: world centre)."""
    n = world.shape[0]
    if centre is None:
        centre = (origin_xy[0] + unit * (n // 2), origin_xy[1] + unit * (n // 2))
    for _ in range(1000):
        x = centre[0] + rs.uniform(-spread, spread)
        y = centre[1] + rs.uniform(-spread, spread)
        c = int(round((x - origin_xy[0]) / unit))
        r = int(round((y - origin_xy[1]) / unit))
        k = max(1, int(0.3 / unit))
        if 0 <= r - k and r + k < n and 0 <= c - k and c + k < n and not world[r - k:r + k + 1, c - k:c + k + 1].any():
            return x, y, rs.uniform(-np.pi, np.pi)
    raise RuntimeError("no free pose found")


def random_walk(world, unit, origin_xy, n_scans, seed=0, step=0.4, turn=0.15, max_radius=None):
    """Seeded trajectory of ``n_scans`` poses that stays in free space (and, if
    ``max_radius`` is given, within that distance of its start); each pose sits on
    the map lattice (multiples of ``unit`` from the first), like the poses the
    matcher emits."""
    rs = np.random.RandomState(seed)
    x, y, th = free_pose_near(world, unit, origin_xy, rs, spread=1.0)
    x = origin_xy[0] + unit * round((x - origin_xy[0]) / unit)
    y = origin_xy[1] + unit * round((y - origin_xy[1]) / unit)
    poses = [(x, y, th)]
    x0, y0 = x, y
    n = world.shape[0]
    k = max(1, int(0.3 / unit))
    while len(poses) < n_scans:
        for _ in range(50):
            nth = th + rs.normal(0, turn)
            nx = x + unit * round(step * np.cos(nth) / unit)
            ny = y + unit * round(step * np.sin(nth) / unit)
            c = int(round((nx - origin_xy[0]) / unit))
            r = int(round((ny - origin_xy[1]) / unit))
            inside = max_radius is None or (nx - x0) ** 2 + (ny - y0) ** 2 <= max_radius ** 2
            if inside and k <= r < n - k and k <= c < n - k and not world[r - k:r + k + 1, c - k:c + k + 1].any():
                x, y, th = nx, ny, nth
                break
            th = th + rs.uniform(-1.0, 1.0)
        poses.append((x, y, th))
    return poses


def counts_from_world(world):
    """(visited, total) count arrays of a 'well-mapped' world: walls observed
    occupied three times (visited 7, total 8 -> ratio 0.875), free cells observed
    empty three times (visited 1, total 5).  float64 like the reference's arrays
    (Utils/OccupancyGrid.py:13-14,148-152)."""
    visited = np.where(world, 7.0, 1.0)
    total = np.where(world, 8.0, 5.0)
    return visited, total
