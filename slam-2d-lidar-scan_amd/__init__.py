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

# Particle groups run on their own HIP streams (ParticleFilter(groups=G), slam2d_groups_*), and streams that share a hardware queue
# serialise: the HIP runtime's default of 4 queues holds two groups beside the default stream.  The runtime reads this when it
# initialises, i.e. at the process's first HIP call -- import this package (or set the variable) before that for more than two groups.
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
