"""MI355X-native correlative scan matching and occupancy-grid update for 2-D
lidar FastSLAM (drop-in for the hot path of xiaofeng419/SLAM-2D-LIDAR-SCAN).

The directory name is the project's (it is not a Python identifier): import it
with ``importlib.import_module("slam-2d-lidar-scan_amd")`` or through the
``slam2d_amd`` alias module at the repository root.
"""
__version__ = "0.1.0"
