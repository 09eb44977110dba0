"""Importable alias for the ``slam-2d-lidar-scan_amd`` package (whose directory
name is not a Python identifier).  ``import slam2d_amd`` yields that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("slam-2d-lidar-scan_amd")
sys.modules[__name__] = _pkg
