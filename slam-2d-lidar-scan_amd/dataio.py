"""Scan logs: the reference's JSON format (`{"map": {"<timestamp>": {"x", "y", "theta", "range": [...]}}}`,
DataSet/PreprocessedData/*, read by `readJson` at Utils/ScanMatcher_OGBased.py:265-268 and iterated in
sorted-key order, :232) and the compact `.npz` re-encoding used by this repository's fixtures
(`range_cm` uint16 [scans, beams], `pose` float64 [scans, 3])."""
import json

import numpy as np


def read_json(path):
    """The reference's `readJson` + `sorted(keys)` iteration: a time-ordered list of reading dicts."""
    with open(path, "r") as f:
        scans = json.load(f)["map"]
    return [scans[k] for k in sorted(scans.keys())]


def read_npz(path):
    """Readings from the compact fixture encoding (ranges in centimetres, exact to the log's 2 decimals)."""
    z = np.load(path)
    ranges = z["range_cm"].astype(np.float64) / 100.0
    return [{"x": float(p[0]), "y": float(p[1]), "theta": float(p[2]), "range": r} for p, r in zip(z["pose"], ranges)]


def write_npz(path, readings):
    rng = np.array([r["range"] for r in readings], dtype=np.float64)
    cm = np.rint(rng * 100)
    if not np.array_equal(cm / 100.0, rng) or cm.max() >= 65536:
        raise ValueError("ranges are not representable in whole centimetres below 655.36 m")
    pose = np.array([[r["x"], r["y"], r["theta"]] for r in readings], dtype=np.float64)
    np.savez_compressed(path, range_cm=cm.astype(np.uint16), pose=pose)
