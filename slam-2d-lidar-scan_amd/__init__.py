"""MI355X-native correlative scan matching and occupancy-grid update for 2-D
lidar FastSLAM (drop-in for the hot path of xiaofeng419/SLAM-2D-LIDAR-SCAN).

The directory name is the project's (it is not a Python identifier): import it
with ``importlib.import_module("slam-2d-lidar-scan_amd")`` or through the
``slam2d_amd`` alias module at the repository root.

    from slam2d_amd import OccupancyGrid, ScanMatcher, ParticleFilter

``OccupancyGrid`` / ``ScanMatcher`` keep the reference's class surface
(Utils/OccupancyGrid.py, Utils/ScanMatcher_OGBased.py); ``ParticleFilter`` is the
batched counterpart of Algorithm/FastSlam.py's.  All three need the HIP library
(``libslam2d_hip.so``, built by ``__graft_entry__.build()``) and a GPU; there is no
CPU fallback.
"""
__version__ = "0.1.0"

# (Importing this package changes nothing in the process's environment.  Particle groups run on their own HIP streams, and streams
# that share a hardware queue take turns: the HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES = 4 queues unless the
# APPLICATION sets the variable before its first HIP call -- bench.py and the examples set 8; engine.group_streams() sets it only
# when HIP has not been initialised yet and otherwise says so once.  INTEGRATION.md, "environment".)

_LAZY = {
    "OccupancyGrid": ("grid", "OccupancyGrid"),
    "ScanMatcher": ("matcher", "ScanMatcher"),
    "ParticleFilter": ("filter", "ParticleFilter"),
    "ParticleEngine": ("engine", "ParticleEngine"),
    "SearchLevel": ("engine", "SearchLevel"),
    "LidarModel": ("engine", "LidarModel"),
    "MapState": ("engine", "MapState"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(name)
