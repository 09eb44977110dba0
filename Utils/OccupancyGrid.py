"""Drop-in for the reference's ``Utils/OccupancyGrid.py``: put this repository ahead of
the reference on ``sys.path`` and ``from Utils.OccupancyGrid import OccupancyGrid``
(Algorithm/FastSlam.py:5) binds the MI355X implementation."""
import importlib

OccupancyGrid = importlib.import_module("slam-2d-lidar-scan_amd.grid").OccupancyGrid
