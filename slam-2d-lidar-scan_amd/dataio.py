"""Scan logs: CARMEN / GMapping text logs (`FLASER`, `LASER_READING` records -- what the reference's
DataPreprocess/preprocess_log_intel.py:22-55 and preprocess_gfs.py:7-22 convert), the reference's JSON format (`{"map": {"<timestamp>": {"x", "y", "theta", "range": [...]}}}`,
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


def _parse_laser_line(tokens, stamp_offset):
    """`<TAG> n r_1 .. r_n x y theta ...`: ranges, the laser pose that follows them, and the time stamp
    `stamp_offset` tokens after the ranges (GFS LASER_READING: 3, CARMEN FLASER: 6 -- the odometry pose sits
    in between, DataPreprocess/preprocess_log_intel.py:29,50)."""
    n = int(tokens[1])
    rng = np.array(tokens[2:n + 2], dtype=np.float64)
    x, y, theta = (float(v) for v in tokens[n + 2:n + 5])
    return float(tokens[n + 2 + stamp_offset]), {"x": x, "y": y, "theta": theta, "range": rng}


def read_text_log(path, record=None):
    """Readings of a CARMEN (`FLASER`) or GMapping (`LASER_READING`) text log in the order the reference's pipeline
    processes them: its preprocessors key the records by the float time stamp in a dict (a repeated stamp: the last record
    wins, DataPreprocess/preprocess_gfs.py:17), dump it as JSON (keys become ``repr(float)`` strings) and the drivers iterate
    ``sorted(sensorData.keys())`` -- the STRING order of those keys (Utils/ScanMatcher_OGBased.py:232), which equals time
    order only while all stamps have the same number of integer digits.  `record`: which tag to read; default: whichever
    the file holds (LASER_READING wins if both occur, as in the reference's Intel pipeline).  Returns (readings, stamps)."""
    offsets = {"LASER_READING": 3, "FLASER": 6}
    found = {k: [] for k in offsets}
    with open(path, "r") as f:
        for line in f:
            tag = line.split(" ", 1)[0]
            if tag in offsets and (record is None or tag == record):
                found[tag].append(_parse_laser_line(line.split(), offsets[tag]))
    rows = found["LASER_READING"] or found["FLASER"]
    if not rows:
        raise ValueError(f"{path}: no FLASER / LASER_READING records")
    by_key = {}
    for stamp, reading in rows:
        by_key[repr(float(stamp))] = (stamp, reading)          # last record of a stamp wins, as in the reference's dict
    order = sorted(by_key)                                     # string order of the JSON keys
    return [by_key[k][1] for k in order], np.array([by_key[k][0] for k in order])


def text_log_to_npz(src, dst, record=None):
    """Text log -> the compact binary fixture format (ranges must be whole centimetres, as the bundled logs are)."""
    readings, _ = read_text_log(src, record)
    write_npz(dst, readings)
    return len(readings)


def read_relations(path):
    """A ``*.relations`` file (ground-truth relative poses between pairs of scans; DataSet/RawData/intel.relations and
    friends) as the reference's DataPreprocess/preprocess_relation.py:1-22 reads it: per line
    ``stamp1 stamp2 x y z roll pitch yaw`` -> x = token 2, y = token 3, theta = token 7, keyed once by the first and once by
    the second time stamp (float keys; a repeated stamp: the last line wins).  Returns
    ``{'relation_timeStamp1': {t1: {x, y, theta, timeStamp2}}, 'relation_timeStamp2': {t2: {x, y, theta, timeStamp1}}}``."""
    by1, by2 = {}, {}
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            x, y, theta, t1, t2 = float(tok[2]), float(tok[3]), float(tok[7]), float(tok[0]), float(tok[1])
            by1[t1] = {"x": x, "y": y, "theta": theta, "timeStamp2": t2}
            by2[t2] = {"x": x, "y": y, "theta": theta, "timeStamp1": t1}
    return {"relation_timeStamp1": by1, "relation_timeStamp2": by2}


def write_relations_json(path, relations):
    """The processed-relations JSON exactly as the reference writes it (sorted keys, indent 4, :20-21)."""
    with open(path, "w") as fp:
        json.dump(relations, fp, sort_keys=True, indent=4)
