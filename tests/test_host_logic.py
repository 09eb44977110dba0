"""Host-side logic of the product (no kernels): spoke LUT, Gaussian taps and the analytic
field floor, level geometry, map growth emulation, odometry prior -- against the oracle.
CPU only (MapState works on torch's CPU device; nothing here calls the HIP library)."""
import importlib
import os
import math

import numpy as np
import pytest
import torch

import codec
from conftest import load_golden
from oracle import slam_oracle as so

eng = importlib.import_module("slam-2d-lidar-scan_amd.engine")
flt = importlib.import_module("slam-2d-lidar-scan_amd.filter")
synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")

CPU = torch.device("cpu")


@pytest.mark.parametrize("cfg", [(0.02, 10, np.pi, 180), (0.1, 10, np.pi, 180), (0.05, 16.0, 1.5 * np.pi, 1081),
                                 (0.1, 34.5, np.pi, 180)])
def test_lidar_model_equals_oracle_lut(cfg):
    unit, R, fov, beams = cfg
    lm = eng.LidarModel(unit, R, fov, beams, 5 * unit)
    o = so.SpokeLUT(unit, R, fov, beams)
    assert np.array_equal(lm.bin.astype(np.int64), o.bin)
    assert np.array_equal(lm.r, o.r) and np.array_equal(lm.xs, o.xs)
    assert (lm.num_spokes, lm.spoke_start, lm.angular_step) == (o.num_spokes, o.start_idx, o.angular_step)


@pytest.mark.parametrize("sigma,miss", [(0.4, 0.15), (2, 0.15 ** 0.4), (2, 0.15), (1.3, 0.3)])
def test_taps_and_floor(sigma, miss):
    w, r = eng.gaussian_taps(sigma)
    wo, ro = so.gaussian_weights(sigma)
    assert r == ro and np.array_equal(w, wo)
    L = math.log(miss)
    free = np.full((6 * r + 9, 6 * r + 7), L)
    blurred = so.blur_reflect(free, sigma)
    floor = eng.blurred_free_value(L, w, r)
    assert np.all(blurred == floor)                 # bit-exact, borders (reflect) included
    free[3 * r + 2, 3 * r + 1] = 0.0                # one occupied cell: every value >= floor
    b2 = so.blur_reflect(free, sigma)
    assert b2.min() == floor and (b2 >= floor).all()


def test_level_geometry_bounds():
    """fmax / wmax really bound the per-particle field and window sizes (incl. the 248/249
    jitter the reference shows)."""
    z = load_golden("levels.npz")
    reach = 1.1 * 10 + 1.4
    for scan in z["scans"]:
        for level, step in (("coarse", 0.1), ("fine", 0.02)):
            fh, fw = z[f"s{scan}_{level}_field_prob_cls"].shape
            fmax = int(2 * reach / step) + 2
            assert fh <= fmax and fw <= fmax and fmax - min(fh, fw) <= 3
            xr = z[f"s{scan}_{level}_field_xr"]
            X = z[f"s{scan}_{level}_field_X"]
            i0, i1 = np.rint((xr - X[0]) / 0.02).astype(int)
            assert i1 - i0 <= int(2 * reach / 0.02) + 3


def test_map_growth_emulation_equals_oracle():
    """MapState growth, the HOST side (the device pass -- slam2d_map_grow: new counts and occupancy bits -- is compared with the
    oracle on the GPU, tests/test_gpu_wide_counts.py): the same sequence, block sizes, coordinates (high-side compression
    included), limits, index conversion, and where the old content is to sit in the new array."""
    init = {"x": 0.698, "y": -0.015}
    m = eng.MapState.create(10, 10, init, 0.02, CPU)
    og = so.GridOracle(10, 10, init, 0.02, np.pi, 180, 10, 0.1, lut=so.SpokeLUT(0.5, 4, np.pi, 180))
    rs = np.random.RandomState(0)
    v = rs.randint(1, 9, og.visited.shape).astype(np.float64)
    t = v + rs.randint(1, 9, og.visited.shape)
    og.visited[:], og.total[:] = v, t
    m.upload(v, t)
    c = m.clone()
    c.cells += 1
    assert not torch.equal(c.cells, m.cells)
    rows0, cols0 = m.rows, m.cols
    m._defer = True                                   # plan only: nothing is materialised without the HIP library
    for (x, y) in [([-12.4, 12.4], [-12.4, 12.4]), ([-20, 3], [1, 2]), ([0, 1], [-30, 22]), ([31, 32], [0, 1])]:
        og.checkAndExapndOG(x, y)
        m.ensure_contains(x, y, 0.02)
        assert m.growth_log == og.growth_log
        assert np.array_equal(m.X, og.X) and np.array_equal(m.Y, og.Y)
        assert m.lim_x == og.mapXLim and m.lim_y == og.mapYLim
        assert (m.rows, m.cols) == og.visited.shape
        dc = sum(n for side, n in m.growth_log if side == 1)
        dr = sum(n for side, n in m.growth_log if side == 3)
        assert m._pending[1:] == [rows0, cols0, dc, dr]
        assert np.array_equal(og.visited[dr:dr + rows0, dc:dc + cols0], v) and np.array_equal(og.total[dr:dr + rows0, dc:dc + cols0], t)
        xi, yi = m.to_map_idx(x, y, 0.02)
        xo, yo = og.convertRealXYToMapIdx(x, y)
        assert np.array_equal(xi, xo) and np.array_equal(yi, yo)


def test_map_upload_download_roundtrip_and_validation():
    m = eng.MapState.create(2, 2, {"x": 0.0, "y": 0.0}, 0.1, CPU)
    v, t = m.download()
    assert np.all(v == 1) and np.all(t == 2) and v.shape == (21, 21)
    rs = np.random.RandomState(1)
    v = rs.randint(0, 65536, v.shape).astype(np.float64); t = rs.randint(0, 65536, v.shape).astype(np.float64)
    m.upload(v, t)
    v2, t2 = m.download()
    assert np.array_equal(v, v2) and np.array_equal(t, t2)
    with pytest.raises(ValueError):
        m.upload(v + 0.5, t)
    with pytest.raises(ValueError):
        m.upload(v[:-1], t[:-1])
    with pytest.raises(ValueError):
        eng.MapState.create(2, 3, {"x": 0.0, "y": 0.0}, 0.1, CPU)


def test_psi_table():
    t = eng.ParticleEngine.psi_table([None, 0.25, float("nan"), -2.0])
    assert np.isnan(t[0]).all() and np.isnan(t[2]).all()
    assert t[1, 0] == math.cos(0.25) and t[3, 1] == math.sin(-2.0)


def test_batched_prior_equals_reference_prior(intel_readings):
    """ParticleFilter._prior vectorises Particle.updateEstimatedPose (FastSlam.py:77-106)."""
    class Dummy(flt.ParticleFilter):
        prev_matched_heading = None      # plain attribute instead of the device-backed property

        def __init__(self):      # no device
            self.numParticles = 3
    pf = Dummy()
    rs = np.random.RandomState(3)
    prev_raw_heading = None
    heads = [None, None, None]
    prev_matched = np.array([[r["x"], r["y"], r["theta"]] for r in intel_readings[:1]] * 3) + rs.normal(0, 0.01, (3, 3))
    for k in range(1, 40):
        raw, prev_raw = intel_readings[k], intel_readings[k - 1]
        pf.prev_matched, pf.prev_raw = prev_matched, prev_raw
        pf.prev_raw_heading, pf.prev_matched_heading = prev_raw_heading, heads
        est, dist, psi, raw_heading = pf._prior(raw)
        for i in range(3):
            pm = {"x": prev_matched[i, 0], "y": prev_matched[i, 1], "theta": prev_matched[i, 2]}
            if prev_raw_heading is not None and heads[i] is None and dist > 0.3:
                continue     # the reference raises TypeError here
            e, d, p, rh = so.odometry_prior(raw, pm, prev_raw, prev_raw_heading, heads[i])
            assert (e["x"], e["y"], e["theta"]) == tuple(est[i]) and d == dist and rh == raw_heading
            assert p == psi[i]
        prev_raw_heading = raw_heading
        heads = [rs.uniform(-3, 3) if rs.rand() > 0.2 else None for _ in range(3)]
        prev_matched = est + rs.normal(0, 0.02, (3, 3))


def test_synthetic_world_is_seeded_and_sane():
    w1 = synth.make_world(40, 0.1, seed=3)
    w2 = synth.make_world(40, 0.1, seed=3)
    assert np.array_equal(w1, w2) and w1.shape == (401, 401) and 0.01 < w1.mean() < 0.2
    origin = (-20.0, -20.0)
    pose = synth.free_pose_near(w1, 0.1, origin, np.random.RandomState(0))
    r = synth.raycast(w1, 0.1, origin, pose, np.pi, 180, 15.0)
    assert r.shape == (180,) and (r > 0).all() and 0 <= (r >= 15.0).mean() < 0.6
    poses = synth.random_walk(w1, 0.1, origin, 20, seed=1)
    assert len(poses) == 20


def test_dataio_roundtrip(tmp_path, intel_readings):
    dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
    import json
    sub = intel_readings[:25]
    jpath = tmp_path / "log.json"
    json.dump({"map": {f"{1000 + i:.3f}": {"x": r["x"], "y": r["y"], "theta": r["theta"], "range": list(r["range"])}
                       for i, r in enumerate(sub)}}, open(jpath, "w"))
    back = dataio.read_json(str(jpath))
    assert [(r["x"], r["y"], r["theta"]) for r in back] == [(r["x"], r["y"], r["theta"]) for r in sub]
    npath = str(tmp_path / "log.npz")
    dataio.write_npz(npath, back)
    again = dataio.read_npz(npath)
    assert all(np.array_equal(a["range"], np.asarray(b["range"])) and a["theta"] == b["theta"] for a, b in zip(again, sub))


def test_text_log_ingest(tmp_path):
    """CARMEN FLASER / GMapping LASER_READING records -> readings in time order -> the compact npz and back
    (DataPreprocess/preprocess_gfs.py:7-22, preprocess_log_intel.py:22-55)."""
    import importlib
    dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
    gfs = tmp_path / "a.gfs"
    gfs.write_text("# comment\n"
                   "LASER_READING 3 1.25 2.50 81.91 0.5 -0.25 1.5 976052892.5 extra\n"
                   "ODOM 1 2 3\n"
                   "LASER_READING 3 1.00 2.00 3.00 0.1 0.2 0.3 976052890.25 extra\n")
    readings, stamps = dataio.read_text_log(str(gfs))
    assert list(stamps) == [976052890.25, 976052892.5]
    assert readings[0]["x"] == 0.1 and readings[0]["theta"] == 0.3 and list(readings[1]["range"]) == [1.25, 2.5, 81.91]
    clf = tmp_path / "a.log"
    clf.write_text("FLASER 2 4.50 5.75 1.0 2.0 0.5 9.0 9.0 9.0 12.5 host 0.1\n"
                   "FLASER 2 1.50 2.25 3.0 4.0 0.25 8.0 8.0 8.0 11.5 host 0.1\n")
    readings, stamps = dataio.read_text_log(str(clf))
    assert list(stamps) == [11.5, 12.5] and (readings[0]["x"], readings[0]["y"], readings[0]["theta"]) == (3.0, 4.0, 0.25)
    assert dataio.text_log_to_npz(str(clf), str(tmp_path / "a.npz")) == 2
    back = dataio.read_npz(str(tmp_path / "a.npz"))
    assert list(back[1]["range"]) == [4.5, 5.75] and back[1]["x"] == 1.0
    ref_raw, ref_json = "/root/reference/DataSet/RawData/csail.corrected.gfs", "/root/reference/DataSet/PreprocessedData/csail_gfs"
    if os.path.exists(ref_raw) and os.path.exists(ref_json):          # the reference's own conversion, when it is around
        got, _ = dataio.read_text_log(ref_raw)
        want = dataio.read_json(ref_json)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert (a["x"], a["y"], a["theta"]) == (b["x"], b["y"], b["theta"]) and list(a["range"]) == list(b["range"])


def test_text_log_order_and_duplicates_follow_the_reference(tmp_path):
    """The reference iterates the STRING-sorted JSON keys of a dict keyed by the float stamp (Utils/ScanMatcher_OGBased.py:232,
    DataPreprocess/preprocess_gfs.py:17): relative stamps that change their digit count come out of time order, and of two
    records with one stamp the last survives."""
    dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
    gfs = tmp_path / "rel.gfs"
    gfs.write_text("LASER_READING 1 1.00 0.0 0.0 0.0 9.8 x\n"
                   "LASER_READING 1 2.00 0.0 0.0 0.0 10.2 x\n"
                   "LASER_READING 1 3.00 0.0 0.0 0.0 100.5 x\n"
                   "LASER_READING 1 4.00 0.0 0.0 0.0 9.8 x\n")
    readings, stamps = dataio.read_text_log(str(gfs))
    import json
    as_reference = json.loads(json.dumps({9.8: 4.0, 10.2: 2.0, 100.5: 3.0}, sort_keys=True))         # float keys -> repr strings
    assert [as_reference[k] for k in sorted(as_reference.keys())] == [float(r["range"][0]) for r in readings] == [2.0, 3.0, 4.0]
    assert list(stamps) == [10.2, 100.5, 9.8]        # '10.2' < '100.5' < '9.8' as strings


def test_relations_reader_matches_the_reference(tmp_path):
    """dataio.read_relations / write_relations_json against the JSON the reference's DataPreprocess/preprocess_relation.py
    wrote for the committed excerpt of intel.relations (tests/golden/make_golden_relations.py), byte for byte."""
    from conftest import GOLDEN
    dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
    rel = dataio.read_relations(os.path.join(GOLDEN, "relations_excerpt.txt"))
    assert len(rel["relation_timeStamp1"]) == len(rel["relation_timeStamp2"]) == 40      # 41 lines, one pair of stamps repeated
    out = tmp_path / "processed.json"
    dataio.write_relations_json(str(out), rel)
    assert out.read_text() == open(os.path.join(GOLDEN, "relations_excerpt_processed.json")).read()
    first = open(os.path.join(GOLDEN, "relations_excerpt.txt")).readline().split()
    e = rel["relation_timeStamp1"][float(first[0])]
    assert (e["x"], e["y"], e["theta"], e["timeStamp2"]) == (float(first[2]), float(first[3]), float(first[7]), float(first[1]))


def test_reference_caller_binds_to_the_shims():
    """The reference's unchanged Algorithm/FastSlam.py, imported with this repository ahead of the reference on
    sys.path, must bind OccupancyGrid / ScanMatcher to this implementation, and the signatures it calls must match
    the reference's (Utils/OccupancyGrid.py:7, Utils/ScanMatcher_OGBased.py:9,47).  Development container only."""
    import importlib
    import inspect
    import subprocess
    import sys
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference not present")
    code = r'''
import sys, inspect, importlib, os
os.environ["MPLBACKEND"] = "Agg"
sys.dont_write_bytecode = True
repo, ref = sys.argv[1], sys.argv[2]
sys.path[:0] = [repo, os.path.join(ref, "Algorithm"), ref]
import FastSlam
assert FastSlam.__file__.startswith(ref)
mine_g = importlib.import_module("slam-2d-lidar-scan_amd.grid").OccupancyGrid
mine_m = importlib.import_module("slam-2d-lidar-scan_amd.matcher").ScanMatcher
assert FastSlam.OccupancyGrid is mine_g and FastSlam.ScanMatcher is mine_m, (FastSlam.OccupancyGrid, FastSlam.ScanMatcher)
# the reference's own classes, loaded from their files, for the signature comparison
import importlib.util
def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
rg = load("ref_og", os.path.join(ref, "Utils", "OccupancyGrid.py")).OccupancyGrid
sys.modules.setdefault("Utils.OccupancyGrid", sys.modules["Utils.OccupancyGrid"])
rm = load("ref_sm", os.path.join(ref, "Utils", "ScanMatcher_OGBased.py")).ScanMatcher
def params(f): return [p for p in inspect.signature(f).parameters]
assert params(mine_g.__init__)[:9] == params(rg.__init__), (params(mine_g.__init__), params(rg.__init__))
assert params(mine_m.__init__) == params(rm.__init__)
assert params(mine_m.matchScan) == params(rm.matchScan)
assert params(mine_g.updateOccupancyGrid) == params(rg.updateOccupancyGrid)
assert params(mine_g.convertRealXYToMapIdx) == params(rg.convertRealXYToMapIdx)
for name in ("plotOccupancyGrid", "checkAndExapndOG", "occupancyGridVisited", "occupancyGridTotal", "OccupancyGridX", "OccupancyGridY"):
    assert hasattr(mine_g, name), name
print("bound")
'''
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code, repo, ref], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "bound" in res.stdout, res.stdout + res.stderr


def test_shared_workspace_guard():
    """The drop-in ScanMatcher's shared per-configuration workspaces serve one call at a time: an overlapping call raises
    (matcher._Exclusive) instead of corrupting another matcher's results."""
    import importlib
    matcher = importlib.import_module("slam-2d-lidar-scan_amd.matcher")
    lib = importlib.import_module("slam-2d-lidar-scan_amd._lib")

    class Level:
        pass
    a, b = Level(), Level()
    with matcher._Exclusive(a, b):
        with pytest.raises(lib.Slam2dError, match="already using"):
            with matcher._Exclusive(b):
                pass
        assert a._busy and b._busy
    assert not a._busy and not b._busy
    with pytest.raises(RuntimeError):
        with matcher._Exclusive(a):
            raise RuntimeError("inside")
    assert not a._busy


def test_division_free_truncation_matches_the_division():
    """csrc/slam2d.hip `trunc_div_fast` (the scatter role's column / row index of a window cell,
    Utils/ScanMatcher_OGBased.py:32-36): where the quotient sits on an integer n -- every column of a window whose pose lies
    on the map's lattice -- (int)(v / step) is decided from the exact remainder r = v - n * step and the spacing of the
    doubles below n instead of the fp64 division.  The same decision restated here (the remainder in exact rationals,
    which is what the device's single fma returns) must equal NumPy's division on quotients a few ulp either side of an
    integer, powers of two included."""
    from fractions import Fraction

    def fast(v, step):
        t = v * (1.0 / step)
        n = float(np.rint(t))
        if abs(t - n) >= 1e-6 and abs(t) < 1e9:
            return int(t)
        if not (1.0 <= n < 2.0 ** 31):
            return int(np.float64(v) / np.float64(step))
        r_exact = Fraction(v) - Fraction(n) * Fraction(step)
        r = float(r_exact)
        assert Fraction(r) == r_exact                      # representable: one fma returns it unrounded
        ni = int(n)
        e, pow2 = ni.bit_length() - 1, (ni & (ni - 1)) == 0
        ghalf = 2.0 ** (e - 53 - pow2)
        return ni if r >= -ghalf * step else ni - 1

    rs = np.random.RandomState(0)
    steps = [0.02, 0.05, 0.1, 0.25, 0.02 * 5, 0.05 * 2, 0.3, 1 / 3, 0.07]
    checked = 0
    for _ in range(60000):
        step = steps[rs.randint(len(steps))]
        n = int([rs.randint(0, 4), rs.randint(1, 2100), 2 ** rs.randint(0, 12), 2 ** rs.randint(1, 12) + rs.randint(-1, 2)][rs.randint(4)])
        v = np.float64(n * step)
        k = rs.randint(-6, 7)
        for _ in range(abs(k)):
            v = np.nextafter(v, np.inf if k > 0 else -np.inf)
        if rs.rand() < 0.2:
            v = np.float64(n * step + rs.uniform(-1e-7, 1e-7) * step)
        if rs.rand() < 0.1:
            v = np.float64(rs.uniform(0, 200))
        v = float(v)
        if v < 0:
            continue
        assert fast(v, step) == int(np.float64(v) / np.float64(step)), (v, step, n)
        checked += 1
    assert checked > 50000


def test_group_issue_policy_follows_the_cores_per_rank():
    """include/slam2d.h slam2d_group_policy: the grouped scan calls keep a polling worker thread per group only where every
    process on the host has at least 3 cores of its own (scheduler affinity capped by the cgroup quota, divided by the local
    ranks) -- 8 ranks on a 16-core quota issue from the calling thread.  No GPU needed: the policy is host-side."""
    import subprocess, sys, json, shutil
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib, json; L = importlib.import_module('slam-2d-lidar-scan_amd._lib'); print(json.dumps(L.group_policy()))")

    def ask(env=None, prefix=()):
        e = dict(os.environ)
        for k in ("SLAM2D_GROUP_THREADS", "SLAM2D_LOCAL_RANKS", "LOCAL_WORLD_SIZE"):
            e.pop(k, None)
        e.update(env or {})
        out = subprocess.run(list(prefix) + [sys.executable, "-c", code], env=e, capture_output=True, text=True, cwd=REPO, timeout=120)
        assert out.returncode == 0, out.stderr[-400:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    base = ask()
    assert base["cores"] >= 1 and base["local_ranks"] == 1
    assert base["threads"] == (base["cores"] >= 3)
    crowded = ask({"LOCAL_WORLD_SIZE": str(max(1, base["cores"]))})      # one core per rank
    assert crowded["local_ranks"] == max(1, base["cores"]) and not crowded["threads"]
    assert ask({"SLAM2D_LOCAL_RANKS": "1", "LOCAL_WORLD_SIZE": "64"})["local_ranks"] == 1      # the explicit variable wins
    assert ask({"LOCAL_WORLD_SIZE": "64", "SLAM2D_GROUP_THREADS": "1"})["threads"]            # forced on
    assert not ask({"SLAM2D_GROUP_THREADS": "0"})["threads"]                                   # forced off
    if shutil.which("taskset") and base["cores"] >= 2:
        two = ask(prefix=("taskset", "-c", "0-1"))
        assert two["cores"] <= 2 and not two["threads"]


def test_reading_window_keeps_iterators_lazy():
    """ParticleFilter.run() over a generator must not exhaust it up front (a live sensor never ends): the window adaptor reads one
    item at a time and still serves the two steps back a re-issued scan needs."""
    filt = importlib.import_module("slam-2d-lidar-scan_amd.filter")
    pulled = []

    def sensor():
        k = 0
        while True:                                         # unbounded
            pulled.append(k)
            yield {"k": k}
            k += 1
    w = filt._ReadingWindow(sensor())
    assert w.has(0) and w[0]["k"] == 0 and pulled == [0]
    for i in range(1, 50):
        assert w.has(i) and w[i]["k"] == i
        assert w[i - 1]["k"] == i - 1 and (i < 2 or w[i - 2]["k"] == i - 2)      # the re-issue's two steps back
        assert len(pulled) == i + 1                          # never ahead of the consumer
    assert len(w._buf) <= 6
    short = filt._ReadingWindow(iter([{"k": 0}, {"k": 1}]))
    assert short.has(0) and short.has(1) and not short.has(2) and not short.has(3)


def test_traffic_json_is_consistent():
    """profiles/traffic.json (what bench.py's roofline replays): every kernel of a step that carries both figures must process no
    more than 4x its counter bytes (a processed figure far above the measured traffic is an accounting slip -- round 4's
    k_grid_update entry held a whole step's bytes against one launch's counters; the gather kernels are exempt: their processed
    figure counts every gather and L2 serves most), launches per step must be whole multiples of
    a half, and the file must name the source it was measured on."""
    import json
    doc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")))
    assert len(doc["source_sha256"]) == 64 and doc["entries"]
    seen = 0
    for key, e in doc["entries"].items():
        if not e.get("in_step"):
            continue
        assert abs(e["launches_per_step"] * 2 - round(e["launches_per_step"] * 2)) < 1e-9, (key, e["launches_per_step"])
        if "processed_bytes_at_measurement" in e and e.get("hbm_bytes_corrected"):
            seen += 1
            if key.split(":")[1] in ("k_exact_select", "k_sweep", "k_bound"):
                continue          # gather kernels: every gather is counted as processed, most are served by L2 (97 % hits at config 5)
            assert e["processed_bytes_at_measurement"] <= 4.0 * e["hbm_bytes_corrected"], (key, e["processed_bytes_at_measurement"], e["hbm_bytes_corrected"])
    assert seen >= 6
    # k_grid_update: one launch per particle group -- its processed bytes are those of the group's particles, not the step's
    for wl in ("config2", "ref2level"):
        g = doc["entries"].get(f"{wl}:k_grid_update")
        if g and "processed_bytes_at_measurement" in g:
            assert 0.3 < g["hbm_bytes_corrected"] / g["processed_bytes_at_measurement"] < 4.0, (wl, g)


def test_auto_groups_follow_particles_and_cores(monkeypatch):
    """ParticleFilter.auto_groups: the particle groups run() steps in when the caller names none -- four from 32 particles (a
    multiple of 4), two from 16, one otherwise; one whenever the host has no cores for the issuing threads.  A sharded rank stays
    at two (round 6: its commit goes through the grouped calls too; with the normaliser's stream and the collective's, two groups make
    the four queues the GPU runs side by side)."""
    import importlib
    filt = importlib.import_module("slam-2d-lidar-scan_amd.filter")
    pol = {"cores": 16, "local_ranks": 1, "threads": True, "polite": False}
    monkeypatch.setattr(filt._lib, "group_policy", lambda: pol)
    auto = filt.ParticleFilter.auto_groups
    assert [auto(p) for p in (1, 6, 15, 16, 30, 32, 64, 66, 256)] == [1, 1, 1, 2, 2, 4, 4, 2, 4]
    assert [auto(p, sharded=True) for p in (3, 15, 16, 64, 65, 256)] == [1, 1, 2, 2, 1, 2]
    pol["threads"] = False
    assert [auto(p) for p in (16, 64, 256)] == [1, 1, 1] and auto(64, sharded=True) == 1


def test_bound_image_pitch_keeps_tile_rows_on_distinct_lds_banks():
    """engine.g2b_pitch (Slam2dLevel.g2b_pitch): a multiple of 16 bytes (the image is staged with 16-byte copies), >= the row, and its
    dword stride mod 32 leaves the three 11-byte tile rows a 32-lane group of k_bound_lds reads (4 dwords each, any byte phase) on
    distinct banks -- for every tile-grid size a level can have."""
    import importlib
    E = importlib.import_module("slam-2d-lidar-scan_amd.engine")
    for tmax in range(4, 140):
        gp = 4 * tmax
        pitch = E.g2b_pitch(gp)
        assert pitch % 16 == 0 and gp <= pitch < gp + 128
        sd = (pitch // 4) % 32
        for phase in range(4):                                  # byte phase of the window's first column inside its dword
            for nbt in (11, 16):
                ndw = (phase + nbt + 3) // 4                    # dwords a tile row spans
                rows = -(-32 // nbt)
                banks = [(r * sd + k) % 32 for r in range(rows) for k in range(ndw)]
                assert len(set(banks)) == len(banks), (tmax, pitch, phase, nbt)
