"""Drop-in for the reference's ``Utils/ScanMatcher_OGBased.py``
(``from Utils.ScanMatcher_OGBased import ScanMatcher``, Algorithm/FastSlam.py:6)."""
import importlib

ScanMatcher = importlib.import_module("slam-2d-lidar-scan_amd.matcher").ScanMatcher
