"""Parity of the HIP path (through the C ABI, via the drop-in classes) against the
golden vectors captured from the reference and against the CPU oracle on the same
seeded inputs.  Needs an MI355X: run with ``-m gpu``.

Bars (BASELINE.json north_star): arg-max pose indices identical; scores,
confidences and weights within 1e-5 relative.  The device stores the search field as
a 32-bit fixed-point cost (include/slam2d.h) and sums it exactly in uint64, so observed
errors are ~1e-9.  Integer work (map counts, cell indices, field dimensions) and the
quantised field itself are compared bit-exactly.
"""
import copy
import importlib

import numpy as np
import pytest

import codec
from conftest import load_golden
from oracle import slam_oracle as so

pytestmark = pytest.mark.gpu
E = importlib.import_module("slam-2d-lidar-scan_amd.engine")

REF_SM = (1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5)
RTOL = 1e-5          # the stated bar
RTOL_TIGHT = 1e-8    # what the 32-bit fixed-point field + exact integer accumulation delivers
# relative_motion_error of the REFERENCE's 6-particle run over the same log (flow_fastslam_long.npz, lineage of its
# best particle): 0.067 m per 10 scans, 0.115 m per 50 scans; raw odometry: 0.243 / 5.23
CONFIG3_REL_ERR_BOUND = {10: 0.10, 50: 0.25}


@pytest.fixture(scope="module")
def pkg():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return importlib.import_module("slam-2d-lidar-scan_amd")


def _grid_with_state(pkg, packed, X, Y, unit=0.02, fov=np.pi, beams=180, R=10, wall=0.1):
    """OccupancyGrid whose device map is the captured state (counts + coordinate vectors)."""
    og = pkg.OccupancyGrid(1, 1, {"x": 0.0, "y": 0.0}, unit, fov, beams, R, wall)
    og.map = pkg.MapState(X, Y, og.device)
    v, t = codec.unpack_counts(packed)
    og.map.upload(v, t)
    og.version += 1
    return og


LEVEL_SCANS = [2, 3, 12, 15, 16, 40, 150, 234]


@pytest.mark.parametrize("scan", LEVEL_SCANS)
@pytest.mark.parametrize("level", ["coarse", "fine"])
def test_field_build_matches_reference(pkg, scan, level):
    """frameSearchSpace + generateProbSearchSpace: quantised probSP and probMin bit-exact, no clamp flips."""
    z = load_golden("levels.npz")
    pre = f"s{scan}_{level}_field_"
    og = _grid_with_state(pkg, z[pre + "map"], z[pre + "X"], z[pre + "Y"])
    sm = pkg.ScanMatcher(og, *REF_SM)
    ex, ey, step, sigma, miss = z[pre + "args"]
    xr, yr, prob = sm.frameSearchSpace(ex, ey, step, sigma, miss)
    want = codec.decode_field(z[pre + "prob_cls"], z[pre + "prob_floor"], z[pre + "prob_other"])
    assert og.map.growth_log == []
    assert np.array_equal(np.array(xr), z[pre + "xr"]) and np.array_equal(np.array(yr), z[pre + "yr"])
    assert prob.shape == want.shape
    level = sm._level(step, sigma, miss, sm.searchRadius, sm.searchHalfRad, False)
    flips = int(((prob == 0) != (want == 0)).sum())
    assert flips == 0, f"{flips} clamp flips"
    assert level.frames()[0]["field_min"] == want.min()                     # probMin, bit-exact
    assert np.array_equal(level.field_cost(0), E.encode_cost(want, level.c.cost_scale)), \
        f"max abs diff {np.abs(prob - want).max():.3e}"
    assert np.abs(prob - want).max() <= 0.5 / level.c.cost_scale


@pytest.mark.parametrize("scan", LEVEL_SCANS)
@pytest.mark.parametrize("level", ["coarse", "fine"])
def test_sweep_matches_reference(pkg, scan, level):
    """searchToMatch on the reference's own probSP: arg-max identical, cube and confidence
    within the bar, matched pose identical."""
    z = load_golden("levels.npz")
    fpre, pre = f"s{scan}_{level}_field_", f"s{scan}_{level}_sweep_"
    prob = codec.decode_field(z[fpre + "prob_cls"], z[fpre + "prob_floor"], z[fpre + "prob_other"])
    og = pkg.OccupancyGrid(1, 1, {"x": 0.0, "y": 0.0}, 0.02, np.pi, 180, 10, 0.1)
    sm = pkg.ScanMatcher(og, *REF_SM)
    ex, ey, eth = z[pre + "est"]
    radius, half, step, dist, psi, fine, mm = z[pre + "args"]
    _, _, matched, cube, conf = sm.searchToMatch(prob, ex, ey, eth, z[pre + "ranges"], z[pre + "xr"], z[pre + "yr"],
                                                 radius, half, step, dist, codec.none_if_nan(psi),
                                                 fineSearch=bool(fine), matchMax=True)
    want = z[pre + "cube"]
    assert cube.shape == want.shape
    np.testing.assert_allclose(cube, want, rtol=RTOL_TIGHT, atol=0)
    assert int(sm.last["adhoc"]["argmax"]) == int(z[pre + "pick"])
    np.testing.assert_allclose(conf, z[pre + "conf"], rtol=RTOL)
    assert [matched["x"], matched["y"], matched["theta"]] == list(z[pre + "matched"])


def test_unique_cells_match_oracle(pkg):
    """np.unique(axis=0) of the rotated endpoint cells, per theta (bit-exact integer work)."""
    z = load_golden("levels.npz")
    fpre, pre = "s150_coarse_field_", "s150_coarse_sweep_"
    prob = codec.decode_field(z[fpre + "prob_cls"], z[fpre + "prob_floor"], z[fpre + "prob_other"])
    og = pkg.OccupancyGrid(1, 1, {"x": 0.0, "y": 0.0}, 0.02, np.pi, 180, 10, 0.1)
    sm = pkg.ScanMatcher(og, *REF_SM)
    ex, ey, eth = z[pre + "est"]
    radius, half, step, dist, psi, fine, mm = z[pre + "args"]
    sm.searchToMatch(prob, ex, ey, eth, z[pre + "ranges"], z[pre + "xr"], z[pre + "yr"], radius, half, step, dist,
                     codec.none_if_nan(psi), fineSearch=bool(fine), matchMax=True)
    level = sm._level(step, 1.0, 0.5, radius, half, bool(fine))
    ogo = so.GridOracle(1, 1, {"x": 0.0, "y": 0.0}, 0.02, np.pi, 180, 10, 0.1, lut=so.SpokeLUT(0.5, 4, np.pi, 180))
    smo = so.MatcherOracle(ogo, *REF_SM)
    px, py = smo.covertMeasureToXY(ex, ey, eth, z[pre + "ranges"])
    for it, th in enumerate(smo.theta_range(half)):
        cells = smo.unique_cells(ex, ey, px, py, th, z[pre + "xr"][0], z[pre + "yr"][0], step)
        cy, cx = level.cells_of(0, it)
        got = set(zip(cx.tolist(), cy.tolist()))
        assert got == set(map(tuple, cells.tolist())), f"theta {it}"


@pytest.mark.parametrize("scan", [1, 2, 40])
def test_update_matches_reference(pkg, scan):
    """updateOccupancyGrid: counts bit-exact, incl. the scan-1 growth inside the update."""
    z = load_golden("update.npz")
    mapx, mapy, unit, fov, beams, R, wall = z["cfg"]
    og = pkg.OccupancyGrid(mapx, mapy, {"x": float(z["init"][0]), "y": float(z["init"][1])}, unit, fov, int(beams),
                           R, wall)
    before = z[f"s{scan}_before"]
    if before.shape != (og.map.rows, og.map.cols):
        xl0, xl1, yl0, yl1 = z[f"s{scan}_lim_after"]
        X = np.linspace(xl0, xl1, before.shape[1]); Y = np.linspace(yl0, yl1, before.shape[0])
        X[0], X[-1], Y[0], Y[-1] = xl0, xl1, yl0, yl1
        og.map = pkg.MapState(X, Y, og.device)
        og.version += 1
    og.map.upload(*codec.unpack_counts(before))
    x, y, th = z[f"s{scan}_pose"]
    og.updateOccupancyGrid({"x": x, "y": y, "theta": th, "range": z[f"s{scan}_ranges"]})
    after = codec.pack_counts(og.occupancyGridVisited, og.occupancyGridTotal)
    assert after.shape == z[f"s{scan}_after"].shape
    assert np.array_equal(after, z[f"s{scan}_after"])
    assert np.array_equal(np.array([og.mapXLim[0], og.mapXLim[1], og.mapYLim[0], og.mapYLim[1]]),
                          z[f"s{scan}_lim_after"])


def test_scanmatch_flow_matches_reference(pkg, intel_readings):
    """Config 1 plumbing on the GPU classes: the single-trajectory flow
    (Utils/ScanMatcher_OGBased.py:226-256) over ALL 320 golden Intel scans, map growth 501^2 -> final
    extent included: poses identical to the reference's, confidences within the bar, final map (shape,
    limits, SHA-256 of the counts) identical to the reference's."""
    import hashlib
    z = load_golden("flow_scanmatch.npz")
    n = len(z["poses"])
    assert n == 320
    r0 = intel_readings[0]
    og = pkg.OccupancyGrid(10, 10, r0, 0.02, np.pi, 180, 10, 0.1)
    sm = pkg.ScanMatcher(og, *REF_SM)
    out, confs = so.run_scanmatch_flow(intel_readings, og, sm, max_scans=n)
    got = np.array([[m["x"], m["y"], m["theta"]] for m in out])
    bad = np.flatnonzero((got != z["poses"][:n]).any(axis=1))
    assert bad.size == 0, f"first differing scan {bad[0] + 1}: {got[bad[0]]} vs {z['poses'][bad[0]]}"
    np.testing.assert_allclose(np.array(confs, dtype=np.float64), z["confs"][:n], rtol=RTOL)
    visited, total = og.occupancyGridVisited, og.occupancyGridTotal
    assert list(visited.shape) == list(z["final_shape"])
    assert [og.mapXLim[0], og.mapXLim[1], og.mapYLim[0], og.mapYLim[1]] == list(z["final_lims"])
    assert hashlib.sha256(codec.pack_counts(visited, total).tobytes()).digest() == z["final_map_sha"].tobytes()


def test_scanmatch_flow_csail_matches_reference(pkg, csail_readings):
    """The same flow over the reference's second bundled log (CSAIL, 361 beams: 722 spokes, 59 search angles, two beams
    per endpoint thread), all 80 golden scans: poses identical, confidences within the bar, final map identical."""
    import hashlib
    z = load_golden("flow_scanmatch_csail.npz")
    n = len(z["poses"])
    r0 = csail_readings[0]
    og = pkg.OccupancyGrid(10, 10, r0, 0.02, np.pi, int(z["beams"]), 10, 0.1)
    sm = pkg.ScanMatcher(og, *REF_SM)
    out, confs = so.run_scanmatch_flow(csail_readings, og, sm, max_scans=n)
    got = np.array([[m["x"], m["y"], m["theta"]] for m in out])
    bad = np.flatnonzero((got != z["poses"][:n]).any(axis=1))
    assert bad.size == 0, f"first differing scan {bad[0] + 1}: {got[bad[0]]} vs {z['poses'][bad[0]]}"
    np.testing.assert_allclose(np.array(confs, dtype=np.float64), z["confs"][:n], rtol=RTOL)
    visited, total = og.occupancyGridVisited, og.occupancyGridTotal
    assert list(visited.shape) == list(z["final_shape"])
    assert [og.mapXLim[0], og.mapXLim[1], og.mapYLim[0], og.mapYLim[1]] == list(z["final_lims"])
    assert hashlib.sha256(codec.pack_counts(visited, total).tobytes()).digest() == z["final_map_sha"].tobytes()
    # the same driver on the batched path: ParticleFilter(1, match_max=True).run() -- lazy field build, prior pruning /
    # branch and bound, pipelined scans -- must walk the same trajectory and build the same map
    u = 0.02
    ogP = [10, 10, r0, u, np.pi, 10, int(z["beams"]), 5 * u]
    for bnb in (False, True):
        pf = pkg.ParticleFilter(1, ogP, list(REF_SM), rng=np.random.RandomState(0), bnb=bnb, match_max=True)
        pf.run(csail_readings[:n])
        traj = np.array([t[0] for t in pf.trajectory])
        assert np.array_equal(traj, z["poses"][:n, :2]), f"bnb={bnb}"
        assert np.array_equal(pf.prev_matched[0], z["poses"][n - 1])
        m = pf.engine.maps[0]
        assert [m.lim_x[0], m.lim_x[1], m.lim_y[0], m.lim_y[1]] == list(z["final_lims"])
        assert hashlib.sha256(codec.pack_counts(*m.download()).tobytes()).digest() == z["final_map_sha"].tobytes()


def test_single_trajectory_driver_on_the_batched_path(pkg, intel_readings):
    """Utils/ScanMatcher_OGBased.py:226-256 (processSensorData) as a product-side driver: one particle, arg-max matches,
    ParticleFilter.run().  All 320 golden Intel scans: trajectory, last pose and final map identical to the reference's."""
    import hashlib
    z = load_golden("flow_scanmatch.npz")
    n = len(z["poses"])
    u = 0.02
    ogP = [10, 10, intel_readings[0], u, np.pi, 10, 180, 5 * u]
    pf = pkg.ParticleFilter(1, ogP, list(REF_SM), match_max=True)
    confs = []
    pf.run(intel_readings[:n], on_scan=lambda count, f, unb: confs.append(float(f.last_confidence[0])))
    traj = np.array([t[0] for t in pf.trajectory])
    assert np.array_equal(traj, z["poses"][:n, :2])
    assert np.array_equal(pf.prev_matched[0], z["poses"][n - 1])
    np.testing.assert_allclose(np.array(confs), z["confs"][:n], rtol=RTOL)
    m = pf.engine.maps[0]
    assert [m.rows, m.cols] == list(z["final_shape"])
    assert [m.lim_x[0], m.lim_x[1], m.lim_y[0], m.lim_y[1]] == list(z["final_lims"])
    assert hashlib.sha256(codec.pack_counts(*m.download()).tobytes()).digest() == z["final_map_sha"].tobytes()


def test_scanmatch_flow_other_parameters_matches_reference(pkg, intel_readings):
    """The reference's flow with non-default constructor parameters (unit 0.04, coarse factor 4: blur radii 3 -- no
    specialised kernel -- and 12; 8 m lidar; 21 x 15 x 15 / 21 x 9 x 9 cubes): drop-in classes, and the batched path with
    and without branch and bound."""
    import hashlib
    z = load_golden("flow_scanmatch_params.npz")
    mx, my, unit, fov, beams, R, wall = z["og_args"]
    a = z["sm_args"]
    smP = [a[0], a[1], a[2], a[3], a[4], a[5], a[6], int(a[7])]
    n = len(z["poses"])
    og = pkg.OccupancyGrid(mx, my, intel_readings[0], unit, fov, int(beams), R, wall)
    sm = pkg.ScanMatcher(og, *smP)
    out, confs = so.run_scanmatch_flow(intel_readings, og, sm, max_scans=n)
    got = np.array([[m["x"], m["y"], m["theta"]] for m in out])
    bad = np.flatnonzero((got != z["poses"]).any(axis=1))
    assert bad.size == 0, f"first differing scan {bad[0] + 1}: {got[bad[0]]} vs {z['poses'][bad[0]]}"
    np.testing.assert_allclose(np.array(confs, dtype=np.float64), z["confs"], rtol=RTOL)
    assert hashlib.sha256(codec.pack_counts(og.occupancyGridVisited, og.occupancyGridTotal).tobytes()).digest() == z["final_map_sha"].tobytes()
    ogP = [mx, my, intel_readings[0], unit, fov, R, int(beams), wall]
    for bnb in (False, True):
        pf = pkg.ParticleFilter(1, ogP, smP, bnb=bnb, match_max=True)
        pf.run(intel_readings[:n])
        assert np.array_equal(np.array([t[0] for t in pf.trajectory]), z["poses"][:, :2]), f"bnb={bnb}"
        m = pf.engine.maps[0]
        assert [m.lim_x[0], m.lim_x[1], m.lim_y[0], m.lim_y[1]] == list(z["final_lims"])
        assert hashlib.sha256(codec.pack_counts(*m.download()).tobytes()).digest() == z["final_map_sha"].tobytes()


def test_dropin_under_fastslam_caller(pkg, intel_readings):
    """The reference's Particle / ParticleFilter caller logic (restated in the oracle module,
    Algorithm/FastSlam.py:10-140) driving the HIP OccupancyGrid / ScanMatcher classes
    unchanged, copy.deepcopy resample included, against the golden FastSLAM run."""
    z = load_golden("flow_fastslam.npz")
    n_particles, n_scans, seed, map_m = (int(v) for v in z["cfg"])      # all 40 golden scans
    u = 0.02
    ogP = [map_m, map_m, intel_readings[0], u, np.pi, 10, 180, 5 * u]
    np.random.seed(seed)       # the drop-in matcher draws from the legacy global stream, like the reference
    pf = so.ParticleFilterOracle(n_particles, ogP, list(REF_SM), rng=None, grid_cls=pkg.OccupancyGrid,
                                 matcher_cls=pkg.ScanMatcher)
    for count, raw in enumerate(intel_readings[:n_scans], start=1):
        pf.updateParticles(raw, count)
        np.testing.assert_allclose(np.array([p.weight for p in pf.particles], dtype=np.float64),
                                   z["raw_weights"][count - 1], rtol=RTOL)
        unb = pf.weightUnbalanced()
        np.testing.assert_allclose(np.array([p.weight for p in pf.particles], dtype=np.float64),
                                   z["weights"][count - 1], rtol=RTOL)
        got = np.array([[p.prevMatchedReading[k] for k in ("x", "y", "theta")] for p in pf.particles])
        assert np.array_equal(got, z["matched"][count - 1])
        if unb or count in z["force_resample"]:
            draw = pf.resample()
            want = z["resamples"][[r[0] == count for r in z["resamples"]]][0][1:]
            assert np.array_equal(draw, want)
            assert pf.particles[0].sm.og is pf.particles[0].og          # alias survives deepcopy


def test_batched_filter_matches_reference(pkg, intel_readings):
    """The batched ParticleFilter (all particles in one launch set) against the golden
    FastSLAM run: same uniforms from the seeded legacy stream, matched poses identical,
    weights / variance within the bar, resample draws identical, final maps identical."""
    import hashlib
    z = load_golden("flow_fastslam.npz")
    n_particles, n_scans, seed, map_m = (int(v) for v in z["cfg"])
    u = 0.02
    ogP = [map_m, map_m, intel_readings[0], u, np.pi, 10, 180, 5 * u]
    rng = np.random.RandomState(seed)
    pf = pkg.ParticleFilter(n_particles, ogP, list(REF_SM), rng=rng)
    resamples = []
    for count, raw in enumerate(intel_readings[:n_scans], start=1):
        pf.updateParticles(raw, count)
        unb = pf.weightUnbalanced()
        assert unb == bool(z["unbalanced"][count - 1])
        np.testing.assert_allclose(pf.weights, z["weights"][count - 1], rtol=RTOL)
        np.testing.assert_allclose(pf.last_variance, z["variance"][count - 1], rtol=RTOL, atol=1e-12)
        assert np.array_equal(pf.prev_matched, z["matched"][count - 1]), f"scan {count}"
        if unb or count in z["force_resample"]:
            resamples.append(np.concatenate(([count], pf.resample())))
    assert np.array_equal(np.array(resamples), z["resamples"])
    for p, sha in zip(pf.particles, z["maps_sha"]):
        packed = codec.pack_counts(p.og.occupancyGridVisited, p.og.occupancyGridTotal)
        assert hashlib.sha256(packed.tobytes()).digest() == sha.tobytes()


def _batched_filter_against(pkg, z, readings, n_scans=None, **kw):
    """The batched ParticleFilter replaying a golden FastSLAM run of the reference (seeded legacy stream):
    the natural weightUnbalanced() decision (Algorithm/FastSlam.py:37), weights / variance within the bar,
    matched poses identical at every scan, resample draws identical, every change of a particle's map
    shape at the same scan, final maps (limits + SHA-256 of the counts) identical."""
    import hashlib
    n_particles, total_scans, seed, map_m = (int(v) for v in z["cfg"])
    n_scans = n_scans or total_scans
    u = 0.02
    ogP = [map_m, map_m, readings[0], u, np.pi, 10, int(z["beams"]) if "beams" in z.files else 180, 5 * u]
    pf = pkg.ParticleFilter(n_particles, ogP, list(REF_SM), rng=np.random.RandomState(seed), **kw)
    resamples, events, last = [], [], [None] * n_particles
    for count, raw in enumerate(readings[:n_scans], start=1):
        pf.updateParticles(raw, count)
        unb = pf.weightUnbalanced()
        assert unb == bool(z["unbalanced"][count - 1]), f"scan {count}: variance {pf.last_variance!r}"
        np.testing.assert_allclose(pf.weights, z["weights"][count - 1], rtol=RTOL, atol=1e-290, err_msg=f"scan {count}")
        np.testing.assert_allclose(pf.last_variance, z["variance"][count - 1], rtol=RTOL, atol=1e-12)
        assert np.array_equal(pf.prev_matched, z["matched"][count - 1]), f"scan {count}"
        for i, m in enumerate(pf.engine.maps):
            if (m.rows, m.cols) != last[i]:
                last[i] = (m.rows, m.cols)
                events.append([count, i, m.rows, m.cols])
        if unb or count in z["force_resample"]:
            draw = pf.resample()
            resamples.append(np.concatenate(([count], draw)))
            last = [last[j] for j in draw]
    assert np.array_equal(np.array(resamples).reshape(-1, n_particles + 1), z["resamples"][z["resamples"][:, 0] <= n_scans])
    assert np.array_equal(np.array(events), z["shape_events"][z["shape_events"][:, 0] <= n_scans])
    if n_scans == total_scans:
        for p, m, sha, lim in zip(pf.particles, pf.engine.maps, z["maps_sha"], z["final_lims"]):
            assert [m.lim_x[0], m.lim_x[1], m.lim_y[0], m.lim_y[1]] == list(lim)
            packed = codec.pack_counts(p.og.occupancyGridVisited, p.og.occupancyGridTotal)
            assert hashlib.sha256(packed.tobytes()).digest() == sha.tobytes()
    return pf


def _pipelined_run_against(pkg, z, readings, **kw):
    """The same replay through ParticleFilter.run(): the pipelined driver (scan s's match enqueued before scan s-1's
    results are read, discarded and redone when that scan turns out to resample or to need a map growth) must give the
    reference's results scan for scan, and consume the random stream exactly as the reference does."""
    import hashlib
    n_particles, n_scans, seed, map_m = (int(v) for v in z["cfg"])
    u = 0.02
    ogP = [map_m, map_m, readings[0], u, np.pi, 10, int(z["beams"]) if "beams" in z.files else 180, 5 * u]
    rng = np.random.RandomState(seed)
    pf = pkg.ParticleFilter(n_particles, ogP, list(REF_SM), rng=rng, **kw)
    seen = []

    def on_scan(count, f, unb):
        assert unb == bool(z["unbalanced"][count - 1]), f"scan {count}"
        np.testing.assert_allclose(f.weights, z["weights"][count - 1], rtol=RTOL, atol=1e-290, err_msg=f"scan {count}")
        np.testing.assert_allclose(f.last_variance, z["variance"][count - 1], rtol=RTOL, atol=1e-12)
        assert np.array_equal(f.prev_matched, z["matched"][count - 1]), f"scan {count}"
        seen.append(count)
    resamples = pf.run(readings[:n_scans], force_resample=set(int(v) for v in z["force_resample"]), on_scan=on_scan)
    assert seen == list(range(1, n_scans + 1))
    got = np.array([np.concatenate(([c], idx)) for c, idx in resamples]).reshape(-1, n_particles + 1)
    assert np.array_equal(got, z["resamples"])
    for p, m, sha, lim in zip(pf.particles, pf.engine.maps, z["maps_sha"], z["final_lims"]):
        assert [m.lim_x[0], m.lim_x[1], m.lim_y[0], m.lim_y[1]] == list(lim)
        packed = codec.pack_counts(p.og.occupancyGridVisited, p.og.occupancyGridTotal)
        assert hashlib.sha256(packed.tobytes()).digest() == sha.tobytes()
    # the stream was consumed exactly as by the step-by-step calls: the next draw agrees with a fresh replay's
    ref_rng = np.random.RandomState(seed)
    for count in range(2, n_scans + 1):
        ref_rng.random_sample(n_particles)
        if count in set(int(r[0]) for r in z["resamples"]):
            ref_rng.choice(np.arange(n_particles), n_particles, p=z["weights"][count - 1])
    assert rng.random_sample() == ref_rng.random_sample()
    return pf


@pytest.mark.parametrize("golden", ["flow_fastslam_growth.npz", "flow_fastslam_long.npz"])
def test_pipelined_driver_reproduces_reference_runs(pkg, intel_readings, golden, monkeypatch):
    pf = _pipelined_run_against(pkg, load_golden(golden), intel_readings)
    # (round 5: one group goes through the grouped, event-free calls -- ranges pulled from pinned memory, the next prior and ranges in
    # the commit, the report pushed by the device)
    assert pf._grp is not None and pf._grp.devsync and pf.n_groups == 1
    # scans voided on the device (a window left its map) went through the pipeline again after the growth -- for the coarse
    # windows from the host's poses, for the fine windows from the coarse poses the voided commit reports -- not step by step
    assert pf.stats["aborted"] > 0 and pf.stats["reissued"] > 0, pf.stats
    # ... the one-stream calls of rounds 3-4 (what a sharded filter still runs) give the same run
    monkeypatch.setenv("SLAM2D_FILTER_GROUPED1", "0")
    pf1 = _pipelined_run_against(pkg, load_golden(golden), intel_readings)
    assert pf1._grp is None and pf1.stats["aborted"] == pf.stats["aborted"] and pf1.stats["reissued"] == pf.stats["reissued"]
    # ... and round 3's handling (the voided scan and its successor step by step) gives the same run
    monkeypatch.setenv("SLAM2D_FILTER_REISSUE", "0")
    pf0 = _pipelined_run_against(pkg, load_golden(golden), intel_readings)
    assert pf0.stats["reissued"] == 0 and pf0.stats["step_by_step"] > pf.stats["step_by_step"]


@pytest.mark.parametrize("golden,groups,events", [("flow_fastslam_growth.npz", 3, False), ("flow_fastslam_long.npz", 2, False),
                                                  ("flow_fastslam_long.npz", 3, False), ("flow_fastslam_long.npz", 2, True)])
def test_grouped_pipelined_driver_reproduces_reference_runs(pkg, intel_readings, golden, groups, events, monkeypatch):
    """The pipelined driver with the particles in groups on their own HIP streams (slam2d_groups_match / slam2d_groups_commit:
    one library call each per scan, the groups joined only by the normaliser's merge; the abort of a scan whose window left a
    map decided over ALL groups' fault bits): the reference's results scan for scan -- growth inside speculated scans, natural and
    forced resamples (every one a full stop of the group streams), the random stream's state at the end."""
    # events: the event path of rounds 3-4 (SLAM2D_FILTER_EVENTS=1: staging copy + ev_inputs, ev_matched across groups, the merge
    # launch on a third stream, a download) -- round 5's default needs none of them (Slam2dScan.h_ranges / match_seq / h_seq)
    if events:
        monkeypatch.setenv("SLAM2D_FILTER_EVENTS", "1")
    pf = _pipelined_run_against(pkg, load_golden(golden), intel_readings, groups=groups)
    assert pf.n_groups == groups and pf._grp is not None and pf._grp.merged_once and pf._grp.devsync == (not events)
    assert pf.stats["aborted"] > 0 or golden != "flow_fastslam_long.npz"


@pytest.mark.parametrize("driver", ["calls", "run", "run_bnb", "run_groups"])
def test_batched_filter_csail_matches_reference(pkg, csail_readings, driver):
    """The reference's FastSLAM on its second log (CSAIL, 361 beams; 3 particles x 60 scans from a 10 m map, two forced
    resamples): per-call loop, pipelined driver, and the latter with branch and bound forced on."""
    z = load_golden("flow_fastslam_csail.npz")
    if driver == "calls":
        _batched_filter_against(pkg, z, csail_readings)
    else:
        _pipelined_run_against(pkg, z, csail_readings, bnb=(driver == "run_bnb") or None, groups=3 if driver == "run_groups" else 1)


def test_batched_filter_growth_matches_reference(pkg, intel_readings):
    """3 particles x 150 scans from a 10 m initial map: the per-beam growth inside the first update with its
    stale-index writes (Utils/OccupancyGrid.py:144-152), search-window growth at both levels (501^2 ->
    3096 x 2580), particles whose maps grow at different scans, three resamples over ragged extents."""
    _batched_filter_against(pkg, load_golden("flow_fastslam_growth.npz"), intel_readings)


def test_batched_filter_long_run_matches_reference(pkg, intel_readings):
    """BASELINE config 3's closed loop at the reference's own particle count: 6 particles x the whole
    910-scan Intel log, seed 0, 50 m map (Algorithm/FastSlam.py:197-207).  The resamples are the ones the
    reference's degeneracy test fires by itself; maps grow beyond the initial 2501^2."""
    z = load_golden("flow_fastslam_long.npz")
    assert int(z["cfg"][1]) == 910 and len(z["resamples"]) >= 3
    _batched_filter_against(pkg, z, intel_readings)


def relative_motion_error(xy, gt_xy, k=10):
    """Frame-independent trajectory metric against the ground-truth log: mean | |p[i+k] - p[i]| - |g[i+k] - g[i]| |
    over all i (metres per k scans).  The corrected log and the matcher's trajectory live in different frames."""
    d = np.hypot(*(xy[k:] - xy[:-k]).T)
    g = np.hypot(*(gt_xy[k:] - gt_xy[:-k]).T)
    return float(np.abs(d - g).mean())


def test_config3_full_run_64_particles(pkg, intel_readings):
    """BASELINE config 3 at its stated size: FastSLAM, 64 particles, the whole 910-scan Intel log, reference
    defaults, 50 m map with growth on.  No golden exists at 64 particles (the reference needs ~6.4 s per scan);
    asserted: completion without a fault flag, finite normalised weights, the maps grew (the resample trigger sits at total
    degeneracy and need not fire with 64 particles -- the 6-particle golden run covers it),
    and the best particle's trajectory agrees with the ground-truth log (intel_corrected_log) as well as the
    reference's own 6-particle run does and far better than raw odometry (bounds calibrated on that run, see
    tests/golden/make_golden_long.py)."""
    import time
    gt = load_golden("intel_corrected_pose.npz")["pose"][:, :2]
    raw_xy = np.array([[r["x"], r["y"]] for r in intel_readings])
    u = 0.02
    ogP = [50, 50, intel_readings[0], u, np.pi, 10, 180, 5 * u]
    pf = pkg.ParticleFilter(64, ogP, list(REF_SM), rng=np.random.RandomState(0))
    resamples, t0 = [], time.perf_counter()
    for count, raw in enumerate(intel_readings, start=1):
        pf.updateParticles(raw, count)                      # raises Slam2dError on any fatal fault flag
        if pf.weightUnbalanced():
            pf.resample()
            resamples.append(count)
    elapsed = time.perf_counter() - t0
    assert np.isfinite(pf.weights).all() and abs(pf.weights.sum() - 1) < 1e-9
    best = int(np.argmax(pf.weights))
    traj = np.array([t[best] for t in pf.trajectory])
    assert traj.shape == (910, 2)
    m = pf.engine.maps[best]
    assert m.rows > 2501 or m.cols > 2501                  # the Intel log leaves the initial 50 m map
    for k, bound in CONFIG3_REL_ERR_BOUND.items():
        err, err_raw = relative_motion_error(traj, gt, k), relative_motion_error(raw_xy, gt, k)
        print(f"config 3: relative motion error {err:.4f} m / {k} scans (raw odometry {err_raw:.4f}, bound {bound})")
        assert err < bound and err < 0.5 * err_raw
    print(f"config 3: 910 scans x 64 particles in {elapsed:.2f} s; resamples {resamples}; map {m.rows}x{m.cols}")
    assert elapsed < 120


@pytest.mark.parametrize("name", ["synth_cfg2.npz", "synth_cfg5s.npz"])
def test_synthetic_shapes(pkg, name):
    """BASELINE config-2 shape (field 801^2, cube 36x41x41, 180 beams) and the reduced
    config-5 shape (1081 beams over 1.5 pi)."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    z = load_golden(name)
    size_m, unit, R, fov, beams, sr, sh, sigma, miss, dist, psi, wall_cells = z["cfg"]
    world = synth.make_world(size_m, unit, seed=int(z["world_seed"]), wall_cells=int(wall_cells))
    og = pkg.OccupancyGrid(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, int(beams), R, 5 * unit)
    og.set_counts(*synth.counts_from_world(world))
    sm = pkg.ScanMatcher(og, sr, sh, sigma, 0.1, 0.25, 0.3, miss, 1)
    ex, ey, eth = z["est"]
    xr, yr, prob = sm.frameSearchSpace(ex, ey, unit, sigma, miss)
    want = codec.decode_field(z["prob_cls"], z["prob_floor"], z["prob_other"])
    assert og.map.growth_log == []
    level = sm._level(unit, sigma, miss, sm.searchRadius, sm.searchHalfRad, False)
    assert np.array_equal(level.field_cost(0), E.encode_cost(want, level.c.cost_scale))
    _, _, matched, cube, conf = sm.searchToMatch(want, ex, ey, eth, z["ranges"], xr, yr, sr, sh, unit, dist,
                                                 codec.none_if_nan(psi), fineSearch=False, matchMax=True)
    assert tuple(cube.shape) == tuple(z["cube_shape"])
    assert int(sm.last["adhoc"]["argmax"]) == int(z["pick"])
    np.testing.assert_allclose(conf, z["conf"], rtol=RTOL)
    np.testing.assert_allclose(float(sm.last["adhoc"]["log_confidence"]), np.log(z["conf"]), rtol=1e-9)
    if "cube" in z.files:
        np.testing.assert_allclose(cube, z["cube"], rtol=RTOL_TIGHT)
    else:
        np.testing.assert_allclose(cube[::4, ::2, ::2], z["cube_sub"], rtol=RTOL_TIGHT)
    assert [matched["x"], matched["y"], matched["theta"]] == list(z["matched"])


def test_floor_redo_path(pkg):
    """Rare branch: no cell of the field has an all-free neighbourhood, so the field minimum is
    NOT the analytic floor and the clamp pass is redone with the measured minimum.  Forced with a
    map whose occupied cells form a dense lattice; field and probMin must still be bit-exact."""
    lib = importlib.import_module("slam-2d-lidar-scan_amd._lib")
    unit, R, size_m = 0.1, 5.0, 20
    og = pkg.OccupancyGrid(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, 90, R, 0.5)
    n = og.map.rows
    v = np.ones((n, n)); t = np.full((n, n), 5.0)
    v[::5, ::5] = 7.0; t[::5, ::5] = 8.0                 # occupied every 5th cell in both directions
    og.set_counts(v, t)
    sm = pkg.ScanMatcher(og, 1.0, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 1)
    xr, yr, prob = sm.frameSearchSpace(0.3, -0.2, unit, 2, 0.15)
    assert sm.last_flags & lib.F_FLOOR_REDO
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, 90, R, 0.5,
                        lut=so.SpokeLUT(0.5, 4, np.pi, 90))
    ogo.visited[:], ogo.total[:] = v, t
    smo = so.MatcherOracle(ogo, 1.0, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 1)
    xro, yro, want = smo.frameSearchSpace(0.3, -0.2, unit, 2, 0.15)
    level = sm._level(unit, 2, 0.15, sm.searchRadius, sm.searchHalfRad, False)
    assert want.min() > level.floor_value                                   # the premise of the test
    assert level.frames()[0]["field_min"] == want.min()
    assert prob.shape == want.shape and int(((prob == 0) != (want == 0)).sum()) == 0
    assert np.array_equal(level.field_cost(0), E.encode_cost(want, level.c.cost_scale))
    # and the next ordinary build on the same (shared) workspace is unaffected by the redo
    og.set_counts(np.ones((n, n)), np.full((n, n), 5.0))
    ogo.visited[:], ogo.total[:] = 1.0, 5.0
    v2 = np.ones((n, n)); v2[40:44, 30:90] = 7.0
    t2 = np.full((n, n), 5.0); t2[40:44, 30:90] = 8.0
    og.set_counts(v2, t2); ogo.visited[:], ogo.total[:] = v2, t2
    _, _, prob2 = sm.frameSearchSpace(0.3, -0.2, unit, 2, 0.15)
    _, _, want2 = smo.frameSearchSpace(0.3, -0.2, unit, 2, 0.15)
    assert not (sm.last_flags & lib.F_FLOOR_REDO)
    assert np.array_equal(level.field_cost(0), E.encode_cost(want2, level.c.cost_scale))


def test_softmax_draw_matches_numpy(pkg):
    """matchMax=False: the index drawn for a given uniform equals
    cdf.searchsorted(u, 'right') on the reference's cube (np.random.choice semantics)."""
    z = load_golden("levels.npz")
    fpre, pre = "s40_coarse_field_", "s40_coarse_sweep_"
    prob = codec.decode_field(z[fpre + "prob_cls"], z[fpre + "prob_floor"], z[fpre + "prob_other"])
    og = pkg.OccupancyGrid(1, 1, {"x": 0.0, "y": 0.0}, 0.02, np.pi, 180, 10, 0.1)
    sm = pkg.ScanMatcher(og, *REF_SM)
    ex, ey, eth = z[pre + "est"]
    radius, half, step, dist, psi, fine, mm = z[pre + "args"]
    flat = z[pre + "cube"].reshape(-1)
    p = np.exp(flat) / np.exp(flat).sum()
    cdf = np.cumsum(p); cdf /= cdf[-1]
    for seed in range(6):
        np.random.seed(seed)
        u = np.random.RandomState(seed).random_sample()
        sm.searchToMatch(prob, ex, ey, eth, z[pre + "ranges"], z[pre + "xr"], z[pre + "yr"], radius, half, step,
                         dist, codec.none_if_nan(psi), fineSearch=False, matchMax=False)
        want = int(cdf.searchsorted(u, side="right"))
        assert int(sm.last["adhoc"]["pick"]) == want, f"seed {seed}"


def test_weights_kernel(pkg):
    import torch
    from importlib import import_module
    flt = import_module("slam-2d-lidar-scan_amd.filter")
    rs = np.random.RandomState(5)
    for n in (1, 4, 64, 1000):
        logw = rs.uniform(-300, -5, n)
        logc = rs.uniform(-200, 1, n)
        d_lw = torch.from_numpy(logw.copy()).cuda()
        d_lc = torch.from_numpy(logc).cuda()
        d_w = torch.zeros(n, dtype=torch.float64, device="cuda")
        d_s = torch.zeros(2, dtype=torch.float64, device="cuda")
        L = flt._lib.lib()
        flt._lib.check(L.slam2d_weights_normalize(flt._ptr(d_lw), flt._ptr(d_lc), 1, n, flt._ptr(d_w), flt._ptr(d_s),
                                                   flt._stream()), "weights")
        s = logw + logc
        w = np.exp(s - s.max()); w /= w.sum()
        np.testing.assert_allclose(d_w.cpu().numpy(), w, rtol=1e-12)
        np.testing.assert_allclose(d_s[0].item(), ((w - 1 / n) ** 2).sum(), rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(np.exp(d_lw.cpu().numpy()), w, rtol=1e-10)


def test_sharded_weights_kernels(pkg):
    """slam2d_weights_local + slam2d_weights_merge over emulated ranks (ragged shards) give the
    weights, log-weights and variance of the single-process normaliser over all particles, and
    agree with the torch-op restatement parallel.normalize_sharded."""
    import torch
    from importlib import import_module
    flt = import_module("slam-2d-lidar-scan_amd.filter")
    par = import_module("slam-2d-lidar-scan_amd.parallel")
    L = flt._lib.lib()
    rs = np.random.RandomState(11)
    for n, world in ((7, 1), (64, 2), (130, 3), (1000, 8)):
        logw = rs.uniform(-300, -5, n)
        logc = rs.uniform(-200, 1, (n, 3))                          # strided log-confidences
        shards = [par.shard_range(n, world, r) for r in range(world)]
        d_lw = [torch.from_numpy(logw[f:f + c].copy()).cuda() for f, c in shards]
        d_lc = [torch.from_numpy(logc[f:f + c].copy()).cuda() for f, c in shards]
        parts = torch.zeros(3 * world, dtype=torch.float64, device="cuda")
        for r in range(world):
            flt._lib.check(L.slam2d_weights_local(flt._ptr(d_lw[r]), d_lc[r].data_ptr() + 8, 3, shards[r][1],
                                                  parts.data_ptr() + 24 * r, flt._stream()), "local")
        s = logw + logc[:, 1]
        w = np.exp(s - s.max()); w /= w.sum()
        for r, (f, c) in enumerate(shards):
            d_w = torch.zeros(c, dtype=torch.float64, device="cuda")
            d_s = torch.zeros(2, dtype=torch.float64, device="cuda")
            flt._lib.check(L.slam2d_weights_merge(flt._ptr(d_lw[r]), c, flt._ptr(parts), world, n, flt._ptr(d_w),
                                                  flt._ptr(d_s), flt._stream()), "merge")
            np.testing.assert_allclose(d_w.cpu().numpy(), w[f:f + c], rtol=1e-12)
            np.testing.assert_allclose(np.exp(d_lw[r].cpu().numpy()), w[f:f + c], rtol=1e-10)
            np.testing.assert_allclose(d_s[0].item(), ((w - 1 / n) ** 2).sum(), rtol=1e-9, atol=1e-15)
            np.testing.assert_allclose(d_s[1].item(), s.max() + np.log(np.exp(s - s.max()).sum()), rtol=1e-13)
    # the host-side wrapper without a process group == the torch-op version
    norm = par.ShardedNormalizer(L, flt._lib.check, "cuda", 64)
    lw = torch.from_numpy(rs.uniform(-50, 0, 64)).cuda()
    tw, tlw, tvar = par.normalize_sharded(lw.clone(), 64)
    d_w = torch.zeros(64, dtype=torch.float64, device="cuda")
    d_s = torch.zeros(2, dtype=torch.float64, device="cuda")
    norm(lw, None, 1, d_w, d_s)
    np.testing.assert_allclose(d_w.cpu().numpy(), tw.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(lw.cpu().numpy(), tlw.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(d_s[0].item(), tvar.item(), rtol=1e-9, atol=1e-15)
    assert L.slam2d_weights_local(None, None, 1, 4, None, None) < 0
    assert L.slam2d_weights_merge(flt._ptr(lw), 64, flt._ptr(d_s), 1, 8, flt._ptr(d_w), flt._ptr(d_s), None) < 0


def test_large_synthetic_properties(pkg):
    """Size-independent properties at BASELINE config-2 size with 8 particles: identical
    particles give identical results (batch determinism), the update kernel touches each
    window cell at most once (counts rise by exactly the number of owning beams' hits),
    and a second identical update doubles the increments (linearity of the counts)."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    eng_mod = importlib.import_module("slam-2d-lidar-scan_amd.engine")
    unit, R, fov, beams, size_m = 0.1, 34.5, np.pi, 180, 90
    world = synth.make_world(size_m, unit, seed=0)
    origin = (-size_m / 2, -size_m / 2)
    rs = np.random.RandomState(1)
    pose = synth.free_pose_near(world, unit, origin, rs, spread=1.5)
    pose = (origin[0] + unit * round((pose[0] - origin[0]) / unit), origin[1] + unit * round((pose[1] - origin[1]) / unit), pose[2])
    ranges = synth.raycast(world, unit, origin, pose, fov, beams, R)
    P = 8
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, beams, 5 * unit]
    smP = [2.05, 0.30, 2, 0.1, 0.25, 0.3, 0.15, 1]
    pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
    v, t = synth.counts_from_world(world)
    for m in pf.engine.maps:
        m.upload(v, t)
    eng = pf.engine
    est = np.tile([pose[0] + 0.2, pose[1] - 0.1, pose[2] + 0.03], (P, 1))
    d_est, d_rng = eng.to_device(est), eng.to_device(ranges)
    eng.field_build(pf.coarse, d_est, 3)
    eng.sweep(pf.coarse, d_est, 3, d_rng, 0.3, None, None, pf.m_coarse)
    eng.take_flags()
    m = eng.read_matches(pf.m_coarse)
    assert len(set(m["pick"].tolist())) == 1 and len(set(m["confidence"].tolist())) == 1
    cubes = pf.coarse.t["cube"].cpu().numpy()
    assert all(np.array_equal(cubes[0], cubes[i]) for i in range(1, P))
    # oracle on the same inputs (one particle)
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, beams, R, 5 * unit)
    ogo.visited[:], ogo.total[:] = v, t
    smo = so.MatcherOracle(ogo, *smP)
    xr, yr, prob = smo.frameSearchSpace(est[0, 0], est[0, 1], unit, 2, 0.15)
    _, cube_o, conf_o = smo.searchToMatch(prob, est[0, 0], est[0, 1], est[0, 2], ranges, xr, yr, 2.05, 0.30, unit,
                                          0.3, None, fineSearch=False, matchMax=True)
    assert int(m["argmax"][0]) == int(cube_o.argmax())
    np.testing.assert_allclose(cubes[0].reshape(cube_o.shape), cube_o, rtol=RTOL_TIGHT)
    np.testing.assert_allclose(m["log_confidence"][0], np.log(conf_o), rtol=1e-7)
    # update: linearity + oracle
    d_pose = eng.to_device(np.tile(pose, (P, 1)))
    eng.grid_update(d_pose, 3, d_rng)
    eng.take_flags()
    v1, t1 = pf.engine.maps[3].download()
    ogo.update_cell_major({"x": pose[0], "y": pose[1], "theta": pose[2], "range": ranges})
    assert np.array_equal(v1, ogo.visited) and np.array_equal(t1, ogo.total)
    eng.grid_update(d_pose, 3, d_rng)
    eng.take_flags()
    v2, t2 = pf.engine.maps[3].download()
    assert np.array_equal(v2 - v1, v1 - v) and np.array_equal(t2 - t1, t1 - t)


def _tile_bits(level, p):
    """Needed-tile bitmap of slam2d_match for particle p as a bool [tmax, tmax] array."""
    words = np.bitwise_or.reduce(level.t["tileneed"][p].cpu().numpy().view(np.uint32), axis=0)    # over the angles' slices
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:level.tmax * level.tmax]
    return bits.reshape(level.tmax, level.tmax).astype(bool)


@pytest.mark.parametrize("levels", ["single", "two", "radius12", "radius5"])
def test_lazy_match_equals_full_build(pkg, levels):
    """slam2d_match (blur only the tiles the sweep reads) against slam2d_field_build + slam2d_sweep on
    separate, identically driven workspaces over a sequence of scans (the persistent tile state must
    stay truthful while tiles are skipped): matches, cubes and partial reductions bit-identical;
    the lazily built field equals the full one on every tile marked as needed; the needed set is
    a strict subset of the wall tiles (otherwise the test does not test anything)."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    if levels == "single":
        unit, R, fov, beams, size_m, wall = 0.1, 34.5, np.pi, 180, 90, 0.5
        smP = [2.05, 0.30, 2, 0.1, 0.25, 0.3, 0.15, 1]
    elif levels in ("radius12", "radius5"):
        # blur radii beside the reference's: 12 (the generic blur kernel; the triage's block flags reach TWO 8-cell blocks
        # beyond a tile) and 5 (generic kernel, one block)
        unit, R, fov, beams, size_m, wall = 0.1, 20.0, np.pi, 180, 60, 0.5
        smP = [1.45, 0.25, 3.0 if levels == "radius12" else 1.3, 0.1, 0.25, 0.3, 0.15, 1]
    else:
        unit, R, fov, beams, size_m, wall = 0.05, 8.0, np.pi, 180, 30, 0.25
        smP = [1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5]
    world = synth.make_world(size_m, unit, seed=3)
    origin = (-size_m / 2, -size_m / 2)
    poses = synth.random_walk(world, unit, origin, 7, seed=5, step=0.4, max_radius=2.0)
    P = 3
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, beams, wall]
    v, t = synth.counts_from_world(world)
    pfs = []
    for _ in range(2):
        pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
        for m in pf.engine.maps:
            m.upload(v, t)
        pfs.append(pf)
    full, lazy = pfs
    rs = np.random.RandomState(9)
    skipped_total = 0
    for s in range(1, len(poses)):
        ranges = synth.raycast(world, unit, origin, poses[s], fov, beams, R)
        est = np.array([[poses[s - 1][0] + unit * rs.randint(-2, 3), poses[s - 1][1] + unit * rs.randint(-2, 3),
                         poses[s][2] + rs.normal(0, 0.02)] for _ in range(P)])
        uni = rs.random_sample(P)
        psi = np.tile([np.cos(0.3), np.sin(0.3)], (P, 1))
        out = []
        for pf, is_lazy in ((full, False), (lazy, True)):
            eng = pf.engine
            d_est, d_rng, d_u, d_psi = eng.to_device(est), eng.to_device(ranges), eng.to_device(uni), eng.to_device(psi)
            chain = [(pf.coarse, d_est, 3, d_psi, d_u, pf.m_coarse)]
            if levels == "two":
                chain.append((pf.fine, pf.m_coarse, E.MATCH_DOUBLES, None, None, pf.m_fine))
            for level, centre, stride, dp, du, buf in chain:
                if is_lazy:
                    eng.match(level, centre, stride, d_rng, 0.4, dp, du, buf)
                else:
                    eng.field_build(level, centre, stride)
                    eng.sweep(level, centre, stride, d_rng, 0.4, dp, du, buf)
            eng.take_flags()
            out.append([(lv, buf.cpu().numpy().copy(), lv.t["cube"].cpu().numpy().copy(),
                         lv.t["partials"].cpu().numpy().copy()) for lv, _, _, _, _, buf in chain])
        for (lf, mf, cf, pf_), (ll, ml, cl, pl) in zip(*out):
            assert np.array_equal(mf.view(np.uint8), ml.view(np.uint8))
            assert np.array_equal(cf.view(np.uint8), cl.view(np.uint8))
            assert np.array_equal(pf_, pl)
            for p in range(P):
                need = _tile_bits(ll, p)
                assert need.any()
                a, b = lf.field_cost(p), ll.field_cost(p)
                fh, fw = a.shape
                mask = np.kron(need, np.ones((16, 16), dtype=bool))[:fh, :fw]
                assert np.array_equal(a[mask], b[mask])
                walls = lf.t["tilestate"][p].cpu().numpy().astype(bool)         # 1 = blurred by the full build
                skipped_total += int((walls & ~need).sum())
                assert np.array_equal(lf.frames()[p]["field_min"], ll.frames()[p]["field_min"])
    assert skipped_total > 0


def test_lazy_match_falls_back_without_free_tile(pkg):
    """A frame in which every 16x16 tile has an occupied cell nearby has no analytically known minimum:
    slam2d_match must build everything, and agree with the full path (incl. the measured-minimum redo)."""
    unit, R, size_m = 0.1, 5.0, 16
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, R, 90, 0.5]
    smP = [1.0, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 1]
    n = int(size_m / unit) + 1
    v, t = np.ones((n, n)), np.full((n, n), 2.0)
    v[::6, ::6] += 4; t[::6, ::6] += 4                                   # an occupied cell every 6 cells
    res = []
    rng = np.full(90, 2.0) + 0.3 * np.sin(np.arange(90))
    for is_lazy in (False, True):
        pf = pkg.ParticleFilter(2, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
        for m in pf.engine.maps:
            m.upload(v, t)
        eng = pf.engine
        d_est = eng.to_device([[0.1, -0.2, 0.05], [0.3, 0.1, -0.1]])
        if is_lazy:
            eng.match(pf.coarse, d_est, 3, eng.to_device(rng), 0.2, None, None, pf.m_coarse)
        else:
            eng.field_build(pf.coarse, d_est, 3)
            eng.sweep(pf.coarse, d_est, 3, eng.to_device(rng), 0.2, None, None, pf.m_coarse)
        flags = eng.take_flags()
        assert all(int(f) & 0x20 for f in flags)                         # SLAM2D_F_FLOOR_REDO: the measured minimum was used
        res.append((pf.m_coarse.cpu().numpy().copy(), pf.coarse.t["cube"].cpu().numpy().copy(),
                    [pf.coarse.field_cost(p).copy() for p in range(2)], pf.coarse.frames()["field_min"].copy()))
    assert np.array_equal(res[0][0].view(np.uint8), res[1][0].view(np.uint8))
    assert np.array_equal(res[0][1].view(np.uint8), res[1][1].view(np.uint8))
    assert all(np.array_equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    assert np.array_equal(res[0][3], res[1][3]) and np.all(res[0][3] > pf.coarse.floor_value)


def _prior_nan_direction(ncell, step, dist, max_dev):
    """A unit vector (c, s) for which the reference's thetaWeight is NaN (|arg| > 1 by rounding,
    Utils/ScanMatcher_OGBased.py:105-108) at some pose OUTSIDE the motion prior's ring."""
    for a in range(1, 40):
        for b in range(1, 40):
            n = float(np.hypot(a, b))
            c, s_ = a / n, b / n
            for k in range(1, ncell // max(a, b) + 1):
                xv, yv = k * a, k * b
                arg = (xv * c + yv * s_) / np.sqrt(float(xv * xv + yv * yv))
                outside = abs(np.hypot(xv * step, yv * step) - dist) > max_dev
                if arg > 1.0 and outside:
                    return c, s_
    raise AssertionError("no NaN direction found")


def test_prior_pruning_equals_full_sweep(pkg):
    """SLAM2D_MATCH_PRUNE_BY_PRIOR against the unpruned slam2d_match on identically driven workspaces:
    arg-max, drawn index and matched pose identical; confidence within 1e-10 relative (the poses that are
    not scored add < 1e-12); ring poses of the cube bit-identical.  Particles the ring cannot settle --
    estimate so far off that the best ring pose scores below -60, or a NaN prior outside the ring (the
    reference's argmax returns the first NaN) -- must come back bit-identical to the full sweep."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    lib = importlib.import_module("slam-2d-lidar-scan_amd._lib")
    unit, R, fov, beams, size_m, wall = 0.1, 34.5, np.pi, 180, 90, 0.5
    smP = [2.05, 0.30, 2, 0.1, 0.25, 0.3, 0.15, 1]
    world = synth.make_world(size_m, unit, seed=3, wall_cells=6)     # thick walls: matched endpoints score ~0
    origin = (-size_m / 2, -size_m / 2)
    poses = synth.random_walk(world, unit, origin, 6, seed=5, step=0.4, max_radius=2.0)
    P = 5
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, beams, wall]
    v, t = synth.counts_from_world(world)
    pfs = []
    for _ in range(2):
        pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
        for m in pf.engine.maps:
            m.upload(v, t)
        pfs.append(pf)
    full, pruned = pfs
    lv = pruned.coarse
    nan_c, nan_s = _prior_nan_direction(lv.ncell, lv.step, 0.4, 0.25)
    rs = np.random.RandomState(9)
    settled_total = 0
    for s in range(1, len(poses)):
        ranges = synth.raycast(world, unit, origin, poses[s], fov, beams, R)
        ranges = np.where(ranges < R, ranges + 0.3, ranges)   # returns from INSIDE the wall band, as mapped walls give
        est = np.array([[poses[s - 1][0] + unit * rs.randint(-1, 2), poses[s - 1][1] + unit * rs.randint(-1, 2),
                         poses[s][2] + rs.normal(0, 0.02)] for _ in range(P)])
        est[3, 0] += 1.7                                      # particle 3: 1.7 m off -- no ring pose fits
        psi = np.tile([np.cos(0.3), np.sin(0.3)], (P, 1))
        psi[1] = (np.nan, np.nan)                             # estMovingTheta = None
        psi[4] = (nan_c, nan_s)                               # particle 4: NaN thetaWeight outside the ring
        uni = rs.random_sample(P)
        res = []
        for pf, prune in ((full, False), (pruned, True)):
            eng = pf.engine
            eng.match(pf.coarse, eng.to_device(est), 3, eng.to_device(ranges), 0.4, eng.to_device(psi),
                      eng.to_device(uni), pf.m_coarse, prune=prune)
            eng.take_flags()
            res.append((eng.read_matches(pf.m_coarse), pf.coarse.t["cube"].cpu().numpy().copy()))
        (mf, cf), (mp, cp) = res
        state = pruned.coarse.t["prune_state"].cpu().numpy()
        ring = pruned.coarse.t["ring"].cpu().numpy()
        assert 0 < ring[0] < 0.2 * lv.nx * ((lv.nx + 3) // 4)
        assert state[3] == 1 and state[4] == 1           # particles 0-2 settle when their best ring pose reaches -60
        settled_total += int((state == 0).sum())
        nq = (lv.nx + 3) // 4
        for p in range(P):
            assert mf["argmax"][p] == mp["argmax"][p] and mf["pick"][p] == mp["pick"][p]
            assert (mf["x"][p], mf["y"][p], mf["theta"][p]) == (mp["x"][p], mp["y"][p], mp["theta"][p])
            if state[p]:
                assert np.array_equal(mf[p:p + 1].view(np.uint8), mp[p:p + 1].view(np.uint8))
                assert np.array_equal(cf[p].view(np.uint8), cp[p].view(np.uint8))
            else:
                assert mf["best_score"][p] == mp["best_score"][p] >= -100.0 + lib.PRUNE_MARGIN - 170 * 2.0
                np.testing.assert_allclose(mp["confidence"][p], mf["confidence"][p], rtol=1e-10)
                np.testing.assert_allclose(mp["log_confidence"][p], mf["log_confidence"][p], rtol=1e-12)
                for u in ring[1:1 + ring[0]]:
                    iy, dx = divmod(int(u), nq)
                    q = iy * lv.nx + 4 * dx
                    n = min(4, lv.nx - 4 * dx)
                    assert np.array_equal(cf[p].reshape(lv.ntheta, -1)[:, q:q + n], cp[p].reshape(lv.ntheta, -1)[:, q:q + n])
    assert settled_total >= 5


@pytest.mark.parametrize("case", ["no_returns", "one_return", "two_beams"])
def test_degenerate_scans_match_oracle(pkg, case):
    """Ragged / empty inputs: a scan without a single return (every range >= lidarMaxRange: no endpoint,
    the cube is the motion prior alone and the update marks nothing free), a scan with one return, and the
    smallest lidar the reference's linspace supports (2 beams).  All three through every match path
    (full, lazy, lazy + pruned) against the oracle, then the map update against the oracle's."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    unit, R, size_m, wall = 0.1, 5.0, 16, 0.5
    beams = 2 if case == "two_beams" else 90
    smP = [1.0, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 1]
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, R, beams, wall]
    world = synth.make_world(size_m, unit, seed=2, n_boxes=8)
    v, t = synth.counts_from_world(world)
    ranges = synth.raycast(world, unit, (-size_m / 2, -size_m / 2), (0.2, -0.1, 0.4), np.pi, beams, R)
    if case == "no_returns":
        ranges = np.full(beams, R + 1.0)
    elif case == "one_return":
        keep = int(np.argmin(ranges))
        ranges = np.where(np.arange(beams) == keep, ranges, R)
    est = np.array([[0.1, -0.2, 0.35], [0.3, 0.0, 0.45]])
    psi = np.array([[np.cos(0.3), np.sin(0.3)], [np.nan, np.nan]])
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, beams, R, wall)
    ogo.visited[:], ogo.total[:] = v, t
    smo = so.MatcherOracle(ogo, *smP)
    want = []
    for p in range(2):
        xr, yr, prob = smo.frameSearchSpace(est[p, 0], est[p, 1], unit, 2, 0.15)
        want.append(smo.searchToMatch(prob, est[p, 0], est[p, 1], est[p, 2], ranges, xr, yr, 1.0, 0.25, unit, 0.3,
                                      0.3 if p == 0 else None, fineSearch=False, matchMax=True))
    for path in ("full", "lazy", "pruned"):
        pf = pkg.ParticleFilter(2, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
        for m in pf.engine.maps:
            m.upload(v, t)
        eng = pf.engine
        d_est, d_rng, d_psi = eng.to_device(est), eng.to_device(ranges), eng.to_device(psi)
        if path == "full":
            eng.field_build(pf.coarse, d_est, 3)
            eng.sweep(pf.coarse, d_est, 3, d_rng, 0.3, d_psi, None, pf.m_coarse)
        else:
            eng.match(pf.coarse, d_est, 3, d_rng, 0.3, d_psi, None, pf.m_coarse, prune=path == "pruned")
        eng.take_flags()
        got = eng.read_matches(pf.m_coarse)
        for p in range(2):
            matched, cube, conf = want[p]
            assert int(got["argmax"][p]) == int(cube.argmax()), path
            assert (got["x"][p], got["y"][p], got["theta"][p]) == (matched["x"], matched["y"], matched["theta"]), path
            np.testing.assert_allclose(got["confidence"][p], conf, rtol=RTOL_TIGHT, err_msg=path)
            if path != "pruned":
                np.testing.assert_allclose(pf.coarse.cube(p), cube, rtol=RTOL_TIGHT, atol=0)
        # the update with the same scan
        pose = np.array([[0.2, -0.1, 0.4], [0.2, -0.1, 0.4]])
        eng.grid_update(eng.to_device(pose), 3, d_rng)
        eng.take_flags()
        ogu = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, beams, R, wall)
        ogu.visited[:], ogu.total[:] = v, t
        ogu.update_cell_major({"x": 0.2, "y": -0.1, "theta": 0.4, "range": ranges})
        gv, gt = pf.engine.maps[1].download()
        assert np.array_equal(gv, ogu.visited) and np.array_equal(gt, ogu.total), path


def test_sweep_skipping_constant_patches_matches_oracle(pkg):
    """Long cell lists (kmax >= 512: here 1081 beams over 1.5 pi) take the sweep variant that does not load
    patches lying entirely in free-space tiles and adds the constant instead: cube, arg-max and confidence must
    still agree with the oracle, with skipping actually in play (free tiles exist inside the frame)."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    unit, R, fov, beams, size_m, wall = 0.1, 8.0, 1.5 * np.pi, 1081, 30, 0.5
    smP = [1.0, 0.1, 2, 0.1, 0.25, 0.3, 0.15, 1]
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, beams, wall]
    world = synth.make_world(size_m, unit, seed=4, n_boxes=12)
    v, t = synth.counts_from_world(world)
    pose = (0.3, -0.2, 0.5)
    ranges = synth.raycast(world, unit, (-size_m / 2, -size_m / 2), pose, fov, beams, R)
    est = np.array([[0.2, -0.1, 0.47], [0.4, -0.3, 0.52]])
    pf = pkg.ParticleFilter(2, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
    for m in pf.engine.maps:
        m.upload(v, t)
    eng = pf.engine
    assert pf.coarse.kmax >= 512 and pf.coarse.tmax <= 64
    eng.match(pf.coarse, eng.to_device(est), 3, eng.to_device(ranges), 0.3, None, None, pf.m_coarse)
    eng.take_flags()
    got = eng.read_matches(pf.m_coarse)
    masks = pf.coarse.t["freerow"].cpu().numpy()
    assert (masks != 0).any()
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, beams, R, wall)
    ogo.visited[:], ogo.total[:] = v, t
    smo = so.MatcherOracle(ogo, *smP)
    for p in range(2):
        xr, yr, prob = smo.frameSearchSpace(est[p, 0], est[p, 1], unit, 2, 0.15)
        matched, cube, conf = smo.searchToMatch(prob, est[p, 0], est[p, 1], est[p, 2], ranges, xr, yr, 1.0, 0.1, unit,
                                                0.3, None, fineSearch=False, matchMax=True)
        assert int(got["argmax"][p]) == int(cube.argmax())
        np.testing.assert_allclose(pf.coarse.cube(p), cube, rtol=RTOL_TIGHT, atol=0)
        np.testing.assert_allclose(got["log_confidence"][p], np.log(conf), rtol=1e-9)
        assert (got["x"][p], got["y"][p], got["theta"][p]) == (matched["x"], matched["y"], matched["theta"])


def test_config5_full_size_properties(pkg):
    """BASELINE config 5 at full size (2000^2 map @ 0.05 m, 1081 beams over 1.5 pi, coarse 139x41x41 + fine
    139x5x5 cubes), too large for the oracle in a test: size-independent properties instead.  Identical
    particles give identical results; the pruned and the unpruned two-level match agree (arg-max, pose,
    confidence); the matched pose is the planted one; a second identical map update doubles the count
    increments and no cell is touched twice within one update."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    unit, R, fov, beams, size_m, wall = 0.05, 30.0, 1.5 * np.pi, 1081, 100, 0.25
    smP = [2.05, 0.30, 2, 0.1, 0.25, 0.3, 0.15, 2]
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, beams, wall]
    world = synth.make_world(size_m, unit, seed=0, n_boxes=60, wall_cells=6)
    origin = (-size_m / 2, -size_m / 2)
    v, t = synth.counts_from_world(world)
    rs = np.random.RandomState(3)
    true = synth.free_pose_near(world, unit, origin, rs, spread=1.0)
    true = (origin[0] + unit * round((true[0] - origin[0]) / unit), origin[1] + unit * round((true[1] - origin[1]) / unit), true[2])
    ranges = synth.raycast(world, unit, origin, true, fov, beams, R)
    ranges = np.where(ranges < R, ranges + 0.15, ranges)               # returns from inside the wall band
    P = 3
    est = np.tile([true[0] - 0.3, true[1] + 0.2, true[2] + 0.01], (P, 1))     # 0.36 m off: inside the coarse window
    dist = float(np.hypot(0.3, 0.2))
    psi = np.tile([np.cos(2.5), np.sin(2.5)], (P, 1))
    out = []
    for prune in (False, True):
        pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
        for m in pf.engine.maps:
            m.upload(v, t)
        eng = pf.engine
        d_rng = eng.to_device(ranges)
        eng.match(pf.coarse, eng.to_device(est), 3, d_rng, dist, eng.to_device(psi), None, pf.m_coarse, prune=prune)
        eng.match(pf.fine, pf.m_coarse, E.MATCH_DOUBLES, d_rng, dist, None, None, pf.m_fine)
        eng.take_flags()
        c, f = eng.read_matches(pf.m_coarse).copy(), eng.read_matches(pf.m_fine).copy()
        assert len(set(c["argmax"].tolist())) == 1 and len(set(f["argmax"].tolist())) == 1          # batch determinism
        assert len(set(c["confidence"].tolist())) == 1
        out.append((c, f, pf))
    (c0, f0, pf0), (c1, f1, _) = out
    assert c0["argmax"][0] == c1["argmax"][0] and f0["argmax"][0] == f1["argmax"][0]
    assert (f0["x"][0], f0["y"][0], f0["theta"][0]) == (f1["x"][0], f1["y"][0], f1["theta"][0])
    np.testing.assert_allclose(c1["log_confidence"], c0["log_confidence"], rtol=1e-10)
    assert abs(f0["x"][0] - true[0]) <= 2 * unit and abs(f0["y"][0] - true[1]) <= 2 * unit      # the planted pose
    eng = pf0.engine
    d_pose = eng.to_device(np.tile(true, (P, 1)))
    d_rng = eng.to_device(ranges)
    eng.grid_update(d_pose, 3, d_rng); eng.take_flags()
    v1, t1 = eng.maps[2].download()
    eng.grid_update(d_pose, 3, d_rng); eng.take_flags()
    v2, t2 = eng.maps[2].download()
    assert np.array_equal(v2 - v1, v1 - v) and np.array_equal(t2 - t1, t1 - t)
    dv, dt = v1 - v, t1 - t
    assert set(np.unique(dt).tolist()) <= {0.0, 1.0, 2.0} and set(np.unique(dv).tolist()) <= {0.0, 2.0}
    assert dt.sum() > 10000


def test_config5_full_size_matches_oracle(pkg):
    """BASELINE config 5 at FULL size against the oracle: one two-level matchScan (2000^2 map @ 0.05 m, 1081
    beams over 1.5 pi, coarse cube 139x41x41 = 233 659 poses + fine 139x5x5), through the batched lazy path the
    filter uses.  Arg-max identical at both levels, both cubes within 1e-8, confidences within the bar,
    matched pose identical, soft-max draw identical, map counts after the update identical."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    unit, R, fov, beams, size_m, wall = 0.05, 30.0, 1.5 * np.pi, 1081, 100, 0.25
    smP = [2.05, 0.30, 2, 0.1, 0.25, 0.3, 0.15, 2]
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, beams, wall]
    world = synth.make_world(size_m, unit, seed=0, n_boxes=60, wall_cells=6)
    origin = (-size_m / 2, -size_m / 2)
    v, t = synth.counts_from_world(world)
    rs = np.random.RandomState(3)
    true = synth.free_pose_near(world, unit, origin, rs, spread=1.0)
    true = (origin[0] + unit * round((true[0] - origin[0]) / unit), origin[1] + unit * round((true[1] - origin[1]) / unit), true[2])
    ranges = synth.raycast(world, unit, origin, true, fov, beams, R)
    ranges = np.where(ranges < R, ranges + 0.15, ranges)
    est = {"x": true[0] - 0.3, "y": true[1] + 0.2, "theta": true[2] + 0.01, "range": ranges}
    dist, psi, u01 = float(np.hypot(0.3, 0.2)), 2.5, 0.37
    # oracle (one particle; ~10 s of NumPy)
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, beams, R, wall)
    ogo.visited[:], ogo.total[:] = v, t
    smo = so.MatcherOracle(ogo, *smP)
    smo.trace = []
    want_max, want_conf = smo.matchScan(est, dist, psi, 2, matchMax=True)
    tr_max = [e for e in smo.trace if "cube" in e]
    smo.trace = []
    want_draw, _ = smo.matchScan(est, dist, psi, 2, matchMax=False, uniform=u01)
    tr_draw = [e for e in smo.trace if "cube" in e]
    # HIP: particle 0 = arg-max (uniform ignored via matchMax path below), particles 1.. = soft-max draw
    P = 2
    pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
    for m in pf.engine.maps:
        m.upload(v, t)
    eng = pf.engine
    d_rng = eng.to_device(ranges)
    d_est = eng.to_device(np.tile([est["x"], est["y"], est["theta"]], (P, 1)))
    d_psi = eng.to_device(np.tile([np.cos(psi), np.sin(psi)], (P, 1)))
    for d_u, want, tr in ((None, want_max, tr_max), (eng.to_device(np.full(P, u01)), want_draw, tr_draw)):
        eng.match(pf.coarse, d_est, 3, d_rng, dist, d_psi, d_u, pf.m_coarse, prune=False)
        eng.match(pf.fine, pf.m_coarse, E.MATCH_DOUBLES, d_rng, dist, None, None, pf.m_fine)
        eng.take_flags()
        c, f = eng.read_matches(pf.m_coarse).copy(), eng.read_matches(pf.m_fine).copy()
        for p in range(P):
            assert int(c["argmax"][p]) == int(tr[0]["cube"].argmax()) and int(c["pick"][p]) == int(tr[0]["pick"])
            assert int(f["argmax"][p]) == int(tr[1]["cube"].argmax())
            np.testing.assert_allclose(c["confidence"][p], tr[0]["confidence"], rtol=RTOL)
            np.testing.assert_allclose(c["log_confidence"][p], np.log(tr[0]["confidence"]), rtol=1e-9)
            assert (f["x"][p], f["y"][p], f["theta"][p]) == (want["x"], want["y"], want["theta"])
        np.testing.assert_allclose(pf.coarse.cube(1), tr[0]["cube"], rtol=RTOL_TIGHT)
        np.testing.assert_allclose(pf.fine.cube(0), tr[1]["cube"], rtol=RTOL_TIGHT)
    np.testing.assert_allclose(c["confidence"][0], want_conf, rtol=RTOL)
    # branch and bound (what the batched filter runs at this size): same picks, poses and confidences
    pfb = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=True)
    assert pfb.coarse.bnb
    for m in pfb.engine.maps:
        m.upload(v, t)
    eb = pfb.engine
    for d_u, want, tr in ((None, want_max, tr_max), (eb.to_device(np.full(P, u01)), want_draw, tr_draw)):
        eb.match(pfb.coarse, d_est, 3, d_rng, dist, d_psi, d_u, pfb.m_coarse, prune=False)
        eb.match(pfb.fine, pfb.m_coarse, E.MATCH_DOUBLES, d_rng, dist, None, None, pfb.m_fine)
        eb.take_flags()
        cb, fb = eb.read_matches(pfb.m_coarse).copy(), eb.read_matches(pfb.m_fine).copy()
        for p in range(P):
            assert int(cb["argmax"][p]) == int(tr[0]["cube"].argmax()) and int(cb["pick"][p]) == int(tr[0]["pick"])
            np.testing.assert_allclose(cb["log_confidence"][p], np.log(tr[0]["confidence"]), rtol=1e-9)
            assert (fb["x"][p], fb["y"][p], fb["theta"][p]) == (want["x"], want["y"], want["theta"])
    # pruned by the motion prior: same pick, same pose, confidence within 1e-10
    eng.match(pf.coarse, d_est, 3, d_rng, dist, d_psi, eng.to_device(np.full(P, u01)), pf.m_coarse, prune=True)
    eng.take_flags()
    cp = eng.read_matches(pf.m_coarse).copy()
    assert int(cp["pick"][0]) == int(tr_draw[0]["pick"]) and int(cp["argmax"][0]) == int(tr_draw[0]["cube"].argmax())
    np.testing.assert_allclose(cp["log_confidence"], np.log(tr_draw[0]["confidence"]), rtol=1e-9)
    # map update at the matched pose
    eng.grid_update(pf.m_fine, E.MATCH_DOUBLES, d_rng)
    eng.take_flags()
    ogo.updateOccupancyGrid(want_draw)
    got_v, got_t = eng.maps[1].download()
    assert np.array_equal(got_v, ogo.visited) and np.array_equal(got_t, ogo.total)


# ------------------------------------------------------------------------------------------------
# branch and bound over 4x4 pose tiles (include/slam2d.h): same results as the brute-force sweep
# ------------------------------------------------------------------------------------------------
def _synthetic_filter(pkg, cfg, P, bnb, seed=0, n_boxes=60):
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    unit, size_m = cfg["unit"], cfg["map_m"]
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, cfg["fov"], cfg["max_range"], cfg["beams"], cfg["wall"]]
    smP = [cfg["search_radius"], cfg["half_rad"], cfg["sigma_cells"], 0.1, 0.25, 0.3, cfg["miss"], cfg["coarse_factor"]]
    world = synth.make_world(size_m, unit, seed=seed, n_boxes=n_boxes)
    pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=bnb)
    v, t = synth.counts_from_world(world)
    pf.engine.maps[0].upload(v, t)
    for m in pf.engine.maps[1:]:
        m.cells.copy_(pf.engine.maps[0].cells); m.bits_valid = False
    return pf, world


BNB_CASES = {
    # BASELINE config 2: 41 x 41 x 36 cube, 180 beams
    "config2": dict(unit=0.1, max_range=34.5, fov=np.pi, beams=180, map_m=100.0, search_radius=2.05, half_rad=0.30,
                    sigma_cells=2, miss=0.15, coarse_factor=1, wall=0.5),
    # the reference's defaults at a coarser unit: 27 x 27 coarse cube, 11 x 11 fine cube (3 x 3 tiles, partial tiles)
    "ref": dict(unit=0.05, max_range=10.0, fov=np.pi, beams=180, map_m=60.0, search_radius=1.4 * 2.5, half_rad=0.25,
                sigma_cells=2, miss=0.15, coarse_factor=5, wall=0.25),
    # 1081 beams over 1.5 pi (config 5's scan), 29 x 29 coarse cube
    "hokuyo": dict(unit=0.05, max_range=20.0, fov=1.5 * np.pi, beams=1081, map_m=60.0, search_radius=1.45, half_rad=0.1,
                   sigma_cells=2, miss=0.15, coarse_factor=2, wall=0.25),
}


# launch-shape choices the engine normally makes by itself (angles per k_endpoints block, one- or two-level bounds)
BNB_VARIANTS = {"default": {}, "group1": {"SLAM2D_EP_GROUP": "1"},
                "group3_two_level": {"SLAM2D_EP_GROUP": "3", "SLAM2D_BNB_LEVELS": "2"},
                "group5_one_level": {"SLAM2D_EP_GROUP": "5", "SLAM2D_BNB_LEVELS": "1"}}


@pytest.mark.parametrize("variant", sorted(BNB_VARIANTS))
@pytest.mark.parametrize("case", sorted(BNB_CASES))
def test_branch_and_bound_equals_brute_force(pkg, case, variant, monkeypatch):
    """slam2d_match with Slam2dLevel.bnb against the brute-force sweep of the same call, both levels, arg-max and
    soft-max draw, a spread-out particle cloud over several scans: arg-max, drawn index and matched pose identical,
    log-confidence within 1e-10, the cube identical at the chosen poses; also with a NaN heading prior (np.argmax
    returns the first NaN) and with a scan without returns.  Variants force the engine's launch-shape choices: angle
    groups that do not divide the number of angles, two-level bounds on short cell lists, one level on long ones."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    for k, v in BNB_VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    cfg = BNB_CASES[case]
    P, unit = 5, cfg["unit"]
    origin = (-cfg["map_m"] / 2, -cfg["map_m"] / 2)
    out = {}
    for bnb in (False, True):
        pf, world = _synthetic_filter(pkg, cfg, P, bnb)
        assert pf.coarse.bnb == bnb and pf.fine.bnb == (bnb and pf.fine.nx >= 9)
        assert pf.fine.abound == (bnb and pf.fine.nx <= 5)             # 5 x 5 fine cubes: angle bounds (Slam2dLevel.bnb == 3)
        if bnb and "SLAM2D_BNB_LEVELS" in BNB_VARIANTS[variant]:
            assert pf.coarse.bnb_levels == (int(BNB_VARIANTS[variant]["SLAM2D_BNB_LEVELS"]) if pf.coarse.nx >= 17 else 1)
        if "SLAM2D_EP_GROUP" in BNB_VARIANTS[variant]:
            assert pf.coarse.ep_group == int(BNB_VARIANTS[variant]["SLAM2D_EP_GROUP"])
        eng = pf.engine
        poses = synth.random_walk(world, unit, origin, 5, seed=3, step=0.3, max_radius=6.0)
        rs = np.random.RandomState(7)
        res = []
        for s in range(1, 5):
            ranges = synth.raycast(world, unit, origin, poses[s], cfg["fov"], cfg["beams"], cfg["max_range"])
            if s == 4:
                ranges = np.full_like(ranges, 1.5 * cfg["max_range"])            # no returns: priors only
            k = rs.randint(-3, 4, size=(P, 2))
            est = np.column_stack((poses[s - 1][0] + k[:, 0] * unit, poses[s - 1][1] + k[:, 1] * unit,
                                   poses[s][2] + rs.normal(0, 0.03, P)))
            d = np.hypot(poses[s][0] - poses[s - 1][0], poses[s][1] - poses[s - 1][1])
            psi = np.arctan2(poses[s][1] - poses[s - 1][1], poses[s][0] - poses[s - 1][0]) + 0.013
            cs = (np.cos(psi), np.sin(psi))
            if s == 3:                      # exactly lattice-aligned heading: arccos argument rounds above 1 -> NaN prior
                cs = _prior_nan_direction(pf.coarse.ncell, pf.coarse.step, d, 0.25)
            d_psi = eng.to_device(np.tile(cs, (P, 1)))
            d_rng = eng.to_device(ranges)
            for d_u in (None, eng.to_device(rs.random_sample(P))):
                eng.match(pf.coarse, eng.to_device(est), 3, d_rng, float(d), d_psi, d_u, pf.m_coarse, prune=False)
                eng.match(pf.fine, pf.m_coarse, E.MATCH_DOUBLES, d_rng, float(d), None, None, pf.m_fine)
                eng.take_flags()
                c, f = eng.read_matches(pf.m_coarse).copy(), eng.read_matches(pf.m_fine).copy()
                cube = pf.coarse.t["cube"].cpu().numpy().reshape(P, -1)
                res.append((c, f, cube[np.arange(P), c["pick"]], cube[np.arange(P), c["argmax"]]))
        out[bnb] = res
    assert len(out[True]) == len(out[False]) == 8
    saw_nan = False
    for i, ((c0, f0, pk0, am0), (c1, f1, pk1, am1)) in enumerate(zip(out[False], out[True])):
        assert np.array_equal(c0["argmax"], c1["argmax"]), f"step {i}: coarse arg-max"
        assert np.array_equal(c0["pick"], c1["pick"]), f"step {i}: coarse draw"
        assert np.array_equal(f0["argmax"], f1["argmax"]), f"step {i}: fine arg-max"
        for k in ("x", "y", "theta"):
            assert np.array_equal(f0[k], f1[k]) and np.array_equal(c0[k], c1[k]), f"step {i}: {k}"
        np.testing.assert_allclose(c1["log_confidence"], c0["log_confidence"], rtol=1e-10, equal_nan=True)
        np.testing.assert_allclose(f1["log_confidence"], f0["log_confidence"], rtol=1e-10, equal_nan=True)
        np.testing.assert_array_equal(pk1, pk0)
        np.testing.assert_array_equal(am1, am0)
        saw_nan |= bool(np.isnan(c0["log_confidence"]).any())
    assert saw_nan                         # the NaN-prior scan really produced NaN scores


def test_branch_and_bound_matches_reference_golden(pkg):
    """The reference's own config-2 call (synth_cfg2.npz: field 801^2, cube 36 x 41 x 41) through slam2d_match
    with branch and bound: arg-max and matched pose identical, confidence within the bar."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    z = load_golden("synth_cfg2.npz")
    size_m, unit, R, fov, beams, sr, sh, sigma, miss, dist, psi, wall_cells = z["cfg"]
    world = synth.make_world(size_m, unit, seed=int(z["world_seed"]), wall_cells=int(wall_cells))
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, int(beams), 5 * unit]
    for bnb in (True, False):
        pf = pkg.ParticleFilter(3, ogP, [sr, sh, sigma, 0.1, 0.25, 0.3, miss, 1], growable=False,
                                rng=np.random.RandomState(0), bnb=bnb)
        assert pf.coarse.bnb == bnb
        for m in pf.engine.maps:
            m.upload(*synth.counts_from_world(world))
        eng = pf.engine
        d_est = eng.to_device(np.tile(z["est"], (3, 1)))
        d_psi = eng.to_device(np.tile([np.cos(psi), np.sin(psi)], (3, 1)))
        eng.match(pf.coarse, d_est, 3, eng.to_device(z["ranges"]), float(dist), d_psi, None, pf.m_coarse, prune=False)
        eng.take_flags()
        c = eng.read_matches(pf.m_coarse)
        for p in range(3):
            assert int(c["argmax"][p]) == int(z["pick"])
            np.testing.assert_allclose(c["confidence"][p], z["conf"], rtol=RTOL)
            np.testing.assert_allclose(c["log_confidence"][p], np.log(z["conf"]), rtol=1e-9)
            assert [c["x"][p], c["y"][p], c["theta"][p]] == list(z["matched"])


@pytest.mark.parametrize("golden", ["flow_fastslam_growth.npz", "flow_fastslam_long.npz"])
def test_branch_and_bound_reproduces_reference_runs(pkg, intel_readings, golden):
    """The reference's closed-loop FastSLAM runs (natural resamples, growth, 910 scans) with branch and bound forced
    on at both levels (27 x 27 and 11 x 11 cubes): every matched pose, weight, resample draw and final map as before."""
    pf = _batched_filter_against(pkg, load_golden(golden), intel_readings, bnb=True)
    assert pf.coarse.bnb and pf.fine.bnb


def test_device_sincos_vs_numpy(pkg, intel_readings):
    """Deviation (i) of DESIGN.md, measured: the endpoint kernel evaluates cos / sin of the beam angles with ocml,
    the reference with NumPy (Utils/ScanMatcher_OGBased.py:87-88).  Over every beam angle of the Intel log's
    headings (|theta| up to ~31 rad over the run, +- the search half-width) the two agree to <= 1 ulp, and an
    endpoint's cell index -- trunc of (x + r cos a - begin) / step -- is the same for every beam of the log at both
    levels."""
    import torch
    lib = importlib.import_module("slam-2d-lidar-scan_amd._lib")
    L = lib.lib()
    th = np.array([r["theta"] for r in intel_readings])
    th = np.concatenate([th * k for k in (1.0, 5.0, 13.0)])                # the run winds up to ~31 rad; go well past it
    fov, B = np.pi, 180
    ang = np.concatenate([np.linspace(t - fov / 2, t + fov / 2, B) for t in th])
    d_a = torch.from_numpy(ang).cuda()
    d_c, d_s = torch.empty_like(d_a), torch.empty_like(d_a)
    lib.check(L.slam2d_device_sincos(d_a.data_ptr(), ang.size, d_c.data_ptr(), d_s.data_ptr(), E._stream()), "sincos")
    c, s_ = d_c.cpu().numpy(), d_s.cpu().numpy()

    def ulps(a, b):
        ia, ib = a.view(np.int64), b.view(np.int64)
        return np.abs(np.where(ia < 0, np.int64(-2 ** 63) - ia, ia) - np.where(ib < 0, np.int64(-2 ** 63) - ib, ib))
    uc, us = ulps(c, np.cos(ang)), ulps(s_, np.sin(ang))
    print(f"ocml vs NumPy over {ang.size} beam angles: cos exact {np.mean(uc == 0):.4f}, max {uc.max()} ulp; "
          f"sin exact {np.mean(us == 0):.4f}, max {us.max()} ulp")
    assert uc.max() <= 1 and us.max() <= 1
    # endpoint cells of the real log with either pair of tables: identical
    n_diff = 0
    for r in intel_readings[::7]:
        rng = np.asarray(r["range"])
        a = np.linspace(r["theta"] - fov / 2, r["theta"] + fov / 2, B)
        d_a = torch.from_numpy(a).cuda(); d_c = torch.empty_like(d_a); d_s = torch.empty_like(d_a)
        lib.check(L.slam2d_device_sincos(d_a.data_ptr(), B, d_c.data_ptr(), d_s.data_ptr(), E._stream()), "sincos")
        cd, sd = d_c.cpu().numpy(), d_s.cpu().numpy()
        for step in (0.1, 0.02):
            lo_x, lo_y = r["x"] - 12.4, r["y"] - 12.4
            for cc, ss in ((np.cos(a), np.sin(a)),):
                ref_x = ((r["x"] + cc * rng - lo_x) / step).astype(int); ref_y = ((r["y"] + ss * rng - lo_y) / step).astype(int)
            dev_x = ((r["x"] + cd * rng - lo_x) / step).astype(int); dev_y = ((r["y"] + sd * rng - lo_y) / step).astype(int)
            n_diff += int((ref_x != dev_x).sum() + (ref_y != dev_y).sum())
    assert n_diff == 0


def test_fault_flags_instead_of_out_of_bounds(pkg):
    """Data-dependent faults are reported, never executed: a search window or an update window
    that leaves a non-growable map raises, and a 16-bit count that would overflow raises."""
    lib = importlib.import_module("slam-2d-lidar-scan_amd._lib")
    unit, R, size_m = 0.1, 5.0, 14
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, R, 90, 0.5]
    smP = [1.0, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 1]
    pf = pkg.ParticleFilter(3, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)     # P not a multiple of 8
    eng = pf.engine
    rng = np.full(90, 2.0)
    # (a) search window outside the map (reach = 6.5 m, map half-size 7 m, pose 1 m off centre)
    d_est = eng.to_device(np.tile([1.0, 0.0, 0.0], (3, 1)))
    eng.field_build(pf.coarse, d_est, 3)
    with pytest.raises(lib.Slam2dError, match="window outside"):
        eng.take_flags()
    # (b) update window outside the map: pose 3 m from the edge with a 5 m lidar
    d_pose = eng.to_device(np.tile([4.0, 0.0, 0.3], (3, 1)))
    before = pf.engine.maps[1].download()
    eng.grid_update(d_pose, 3, eng.to_device(rng + 4.0))
    with pytest.raises(lib.Slam2dError, match="outside the map"):
        eng.take_flags()
    # (c) count overflow: cells already at the 16-bit limit are left alone and reported
    v = np.full((pf.engine.maps[0].rows,) * 2, 1.0); t = np.full(v.shape, 65535.0)
    for m in pf.engine.maps:
        m.upload(v, t)
    eng.grid_update(eng.to_device(np.zeros((3, 3))), 3, eng.to_device(rng))
    with pytest.raises(lib.Slam2dError, match="overflow"):
        eng.take_flags()
    v2, t2 = pf.engine.maps[2].download()
    assert t2.max() == 65535 and np.array_equal(v2, v)
    # (d) the engine is usable afterwards and the three particles agree
    for m in pf.engine.maps:
        m.upload(np.ones(v.shape), np.full(v.shape, 2.0))
    eng.grid_update(eng.to_device(np.zeros((3, 3))), 3, eng.to_device(rng))
    assert not eng.take_flags().any()
    maps = [m.download() for m in pf.engine.maps]
    assert all(np.array_equal(maps[0][0], mm[0]) and np.array_equal(maps[0][1], mm[1]) for mm in maps[1:])
    assert maps[0][1].sum() > 2.0 * v.size


@pytest.mark.parametrize("sigma", [1.3, 3.0, 0.2, 1.0])
def test_generic_blur_radius(pkg, sigma):
    """Blur radii other than the reference's two (2 and 8): the generic kernel path (sigma 1.3 -> radius 5,
    3.0 -> 12, 0.2 -> 1) and the radius-4 specialisation (sigma 1.0, BASELINE config 5's coarse level):
    quantised field and probMin still bit-exact, and the match on it agrees with the oracle."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    unit, R, size_m, beams = 0.1, 8.0, 30, 120
    world = synth.make_world(size_m, unit, seed=5, n_boxes=20)
    v, t = synth.counts_from_world(world)
    og = pkg.OccupancyGrid(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, beams, R, 0.5)
    og.set_counts(v, t)
    smP = (1.0, 0.2, sigma, 0.1, 0.25, 0.3, 0.2, 1)
    sm = pkg.ScanMatcher(og, *smP)
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, beams, R, 0.5)
    ogo.visited[:], ogo.total[:] = v, t
    smo = so.MatcherOracle(ogo, *smP)
    origin = (-size_m / 2, -size_m / 2)
    pose = synth.free_pose_near(world, unit, origin, np.random.RandomState(2), spread=1.0)
    pose = (origin[0] + unit * round((pose[0] - origin[0]) / unit), origin[1] + unit * round((pose[1] - origin[1]) / unit), pose[2])
    ranges = synth.raycast(world, unit, origin, pose, np.pi, beams, R)
    ex, ey = pose[0] + 0.2, pose[1] - 0.1
    xr, yr, prob = sm.frameSearchSpace(ex, ey, unit, sigma, 0.2)
    xro, yro, want = smo.frameSearchSpace(ex, ey, unit, sigma, 0.2)
    level = sm._level(unit, sigma, 0.2, sm.searchRadius, sm.searchHalfRad, False)
    assert level.blur_radius == int(4 * sigma + 0.5) and level.blur_radius not in (2, 8)
    assert level.frames()[0]["field_min"] == want.min()
    assert np.array_equal(level.field_cost(0), E.encode_cost(want, level.c.cost_scale))
    est = {"x": ex, "y": ey, "theta": pose[2] + 0.03, "range": ranges}
    _, _, matched, cube, conf = sm.searchToMatch(want, ex, ey, est["theta"], ranges, xr, yr, 1.0, 0.2, unit, 0.3, 0.4)
    mo, cube_o, conf_o = smo.searchToMatch(want, ex, ey, est["theta"], ranges, xro, yro, 1.0, 0.2, unit, 0.3, 0.4)
    assert int(sm.last["adhoc"]["argmax"]) == int(cube_o.argmax())
    np.testing.assert_allclose(cube, cube_o, rtol=RTOL_TIGHT)
    np.testing.assert_allclose(conf, conf_o, rtol=RTOL)


@pytest.mark.parametrize("P", [1, 5, 9])
def test_particle_counts_not_multiple_of_eight(pkg, P):
    """The sweep pins particle p to XCD p % 8; odd particle counts must still cover everyone."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    unit, R, size_m, beams = 0.1, 6.0, 24, 90
    world = synth.make_world(size_m, unit, seed=7, n_boxes=15)
    v, t = synth.counts_from_world(world)
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, R, beams, 0.5]
    smP = [0.8, 0.2, 2, 0.1, 0.25, 0.3, 0.15, 1]
    pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=False)
    for m in pf.engine.maps:
        m.upload(v, t)
    origin = (-size_m / 2, -size_m / 2)
    pose = synth.free_pose_near(world, unit, origin, np.random.RandomState(3), spread=1.0)
    pose = (origin[0] + unit * round((pose[0] - origin[0]) / unit), origin[1] + unit * round((pose[1] - origin[1]) / unit), pose[2])
    ranges = synth.raycast(world, unit, origin, pose, np.pi, beams, R)
    eng = pf.engine
    rs = np.random.RandomState(P)
    est = np.tile([pose[0], pose[1], pose[2]], (P, 1)) + np.column_stack([unit * rs.randint(-2, 3, P), unit * rs.randint(-2, 3, P), rs.normal(0, 0.02, P)])
    d_est, d_rng = eng.to_device(est), eng.to_device(ranges)
    eng.field_build(pf.coarse, d_est, 3)
    eng.sweep(pf.coarse, d_est, 3, d_rng, 0.2, None, None, pf.m_coarse)
    eng.take_flags()
    got = eng.read_matches(pf.m_coarse)
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, np.pi, beams, R, 0.5)
    ogo.visited[:], ogo.total[:] = v, t
    smo = so.MatcherOracle(ogo, *smP)
    for p in range(P):
        xr, yr, prob = smo.frameSearchSpace(est[p, 0], est[p, 1], unit, 2, 0.15)
        mo, cube_o, conf_o = smo.searchToMatch(prob, est[p, 0], est[p, 1], est[p, 2], ranges, xr, yr, 0.8, 0.2, unit, 0.2, None)
        assert int(got["argmax"][p]) == int(cube_o.argmax()), f"particle {p}"
        assert (got["x"][p], got["y"][p], got["theta"][p]) == (mo["x"], mo["y"], mo["theta"])
        np.testing.assert_allclose(got["confidence"][p], conf_o, rtol=RTOL)


def test_map_fill_gather_and_timer_entry_points(pkg):
    """The remaining C-ABI entry points: slam2d_map_fill, slam2d_gather_maps, slam2d_map_refresh_bits,
    slam2d_timer_*."""
    import ctypes as C
    import torch
    lib = importlib.import_module("slam-2d-lidar-scan_amd._lib")
    L = lib.lib()
    dev = torch.device("cuda:0")
    maps = [pkg.MapState.create(6, 6, {"x": 0.0, "y": 0.0}, 0.1, dev) for _ in range(5)]
    rs = np.random.RandomState(0)
    contents = []
    for m in maps:
        v = rs.randint(1, 50, (m.rows, m.cols)).astype(np.float64)
        t = v + rs.randint(0, 60, v.shape)
        m.upload(v, t)
        contents.append((v, t))
    eng = pkg.ParticleEngine(pkg.LidarModel.get(0.1, 3.0, np.pi, 60, 0.3), maps, dev)
    for m, (v, t) in zip(maps, contents):                      # bits rebuilt from the uploaded counts
        bits = m.bits.cpu().numpy().view(np.uint32)
        occ = np.zeros((m.rows, m.bits_pitch * 32), dtype=bool)
        occ[:, :m.cols] = 2 * v > t
        assert np.array_equal(bits, np.packbits(occ.reshape(m.rows, -1, 32)[:, :, ::-1], axis=2).view(">u4").reshape(m.rows, -1))
    idx = np.array([3, 3, 0, 4, 1], dtype=np.int32)
    dst = [pkg.MapState.create(6, 6, {"x": 0.0, "y": 0.0}, 0.1, dev) for _ in range(5)]
    E2 = importlib.import_module("slam-2d-lidar-scan_amd.engine")
    d_src, d_dst = E2.upload_map_descs(maps, dev), E2.upload_map_descs(dst, dev)
    d_idx = torch.as_tensor(idx, device=dev)
    timer = L.slam2d_timer_create()
    assert timer
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.check(L.slam2d_timer_start(timer, stream), "timer_start")
    lib.check(L.slam2d_gather_maps(C.c_void_p(d_src.data_ptr()), C.c_void_p(d_dst.data_ptr()), C.c_void_p(d_idx.data_ptr()),
                                   5, maps[0].rows * maps[0].pitch, stream), "gather")
    lib.check(L.slam2d_timer_stop(timer, stream), "timer_stop")
    ms = C.c_float(-1)
    lib.check(L.slam2d_timer_elapsed_ms(timer, C.byref(ms)), "timer_elapsed")
    L.slam2d_timer_destroy(timer)
    assert 0 <= ms.value < 1000
    for i, j in enumerate(idx):
        v, t = dst[i].download()
        assert np.array_equal(v, contents[j][0]) and np.array_equal(t, contents[j][1])
    lib.check(L.slam2d_map_fill(C.c_void_p(dst[0].cells.data_ptr()), dst[0].rows * dst[0].pitch, lib.INIT_CELL, stream), "fill")
    v, t = dst[0].download()
    assert (v == 1).all() and (t == 2).all()


def test_update_launch_with_the_normaliser_equals_separate_calls(pkg):
    """slam2d_grid_update_weights / slam2d_grid_update_weights_local (the normaliser, or its rank-local half, as block 0 of
    the map update's launch) against slam2d_grid_update followed by slam2d_weights_normalize / slam2d_weights_local:
    maps, occupancy bits, log-weights, weights, statistics and partials bit-identical, over a few scans with carried
    log-weights."""
    import torch
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    lib = E._lib
    cfg = BNB_CASES["config2"]
    P, unit = 7, cfg["unit"]
    origin = (-cfg["map_m"] / 2, -cfg["map_m"] / 2)
    L = lib.lib()
    pfs = [_synthetic_filter(pkg, cfg, P, False)[0] for _ in range(3)]
    world = _synthetic_filter(pkg, cfg, 1, False)[1]
    poses = synth.random_walk(world, unit, origin, 5, seed=4, step=0.3, max_radius=6.0)
    rs = np.random.RandomState(11)
    st = [dict(logw=torch.full((P,), -np.log(P), dtype=torch.float64, device=pf.device),
               w=torch.zeros(P, dtype=torch.float64, device=pf.device), stats=torch.zeros(2, dtype=torch.float64, device=pf.device),
               part=torch.zeros(3, dtype=torch.float64, device=pf.device)) for pf in pfs]
    for s in range(1, 5):
        ranges = synth.raycast(world, unit, origin, poses[s], cfg["fov"], cfg["beams"], cfg["max_range"])
        pose = np.column_stack((poses[s][0] + rs.normal(0, 0.2, P), poses[s][1] + rs.normal(0, 0.2, P), poses[s][2] + rs.normal(0, 0.05, P)))
        conf = rs.normal(-30.0, 5.0, P)
        for k, (pf, t) in enumerate(zip(pfs, st)):
            eng = pf.engine
            d_pose, d_rng, d_conf = eng.to_device(pose), eng.to_device(ranges), eng.to_device(conf)
            if k == 0:      # separate calls; both normalisers on copies of the same log-weights
                eng.grid_update(d_pose, 3, d_rng)
                lw_local = t["logw"].clone()
                lib.check(L.slam2d_weights_local(E._ptr(lw_local), E._ptr(d_conf), 1, P, E._ptr(t["part"]), E._stream()), "local")
                lib.check(L.slam2d_weights_normalize(E._ptr(t["logw"]), E._ptr(d_conf), 1, P, E._ptr(t["w"]), E._ptr(t["stats"]),
                                                     E._stream()), "normalize")
                t["lw_local"] = lw_local
            elif k == 1:
                eng.grid_update_weights(d_pose, 3, d_rng, t["logw"], d_conf.data_ptr(), 1, t["w"], t["stats"])
            else:
                t["lw_before"] = t["logw"].clone()
                eng.grid_update_weights_local(d_pose, 3, d_rng, t["logw"], d_conf.data_ptr(), 1, t["part"])
            eng.take_flags()
        a, b, c = st
        assert torch.equal(a["logw"], b["logw"]) and torch.equal(a["w"], b["w"]) and torch.equal(a["stats"], b["stats"]), f"scan {s}"
        assert torch.equal(a["part"], c["part"]) and torch.equal(a["lw_local"], c["logw"]), f"scan {s}: rank-local half"
        c["logw"].copy_(a["logw"])                       # (the merge half is not under test: carry the normalised weights)
        for p in range(P):
            for other in pfs[1:]:
                assert torch.equal(pfs[0].engine.maps[p].cells, other.engine.maps[p].cells), f"scan {s}: map {p}"
                assert torch.equal(pfs[0].engine.maps[p].bits, other.engine.maps[p].bits), f"scan {s}: bits {p}"
    assert abs(float(st[1]["w"].sum()) - 1.0) < 1e-12


def test_split_exact_select_stress(pkg):
    """k_exact_select over several blocks per particle hands its scores to the particle's last-arriving block through
    agent-scope 8-byte atomics and an arrival ticket, with no cache fence on either side (csrc/slam2d.hip at the ticket;
    MI355X_MICROARCH.md "inter-workgroup visibility": "8-B agent atomics both sides").  1 000 launches at 64 particles
    (4 blocks per particle, every XCD), estimates scattered from spot-on to 2 m off so that the surviving tiles range from
    dozens to thousands per particle, arg-max and soft-max draw: the whole match record must equal, bit for bit, what ONE
    block per particle (Slam2dLevel.sync = NULL) computes from the same inputs."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    cfg = BNB_CASES["config2"]
    P, unit = 64, cfg["unit"]
    origin = (-cfg["map_m"] / 2, -cfg["map_m"] / 2)
    pf, world = _synthetic_filter(pkg, cfg, P, True)
    assert pf.coarse.bnb and pf.coarse.c.sync
    eng = pf.engine
    poses = synth.random_walk(world, unit, origin, 6, seed=5, step=0.3, max_radius=6.0)
    rs = np.random.RandomState(11)
    sync_ptr = pf.coarse.c.sync
    import torch
    single = torch.zeros_like(pf.m_coarse)
    kept = []
    launches = 0
    for rep in range(125):
        s = 1 + rep % 5
        ranges = synth.raycast(world, unit, origin, poses[s], cfg["fov"], cfg["beams"], cfg["max_range"])
        spread = (0, 2, 6, 20)[rep % 4]
        k = rs.randint(-spread, spread + 1, size=(P, 2))
        est = np.column_stack((poses[s][0] + k[:, 0] * unit, poses[s][1] + k[:, 1] * unit, poses[s][2] + rs.normal(0, 0.05, P)))
        d = float(np.hypot(poses[s][0] - poses[s - 1][0], poses[s][1] - poses[s - 1][1]))
        psi = np.arctan2(poses[s][1] - poses[s - 1][1], poses[s][0] - poses[s - 1][0])
        d_psi, d_rng, d_est = eng.to_device(np.tile((np.cos(psi), np.sin(psi)), (P, 1))), eng.to_device(ranges), eng.to_device(est)
        for d_u in (None, eng.to_device(rs.random_sample(P))):
            pf.coarse.c.sync = None                          # one block per particle: no hand-off
            eng.match(pf.coarse, d_est, 3, d_rng, d, d_psi, d_u, single, prune=False)
            eng.take_flags()
            want = single.cpu().numpy().copy()
            pf.coarse.c.sync = sync_ptr
            for _ in range(4):                               # the split launch, repeatedly: placement and arrival order vary
                eng.match(pf.coarse, d_est, 3, d_rng, d, d_psi, d_u, pf.m_coarse, prune=False)
                launches += 1
            eng.take_flags()
            got = pf.m_coarse.cpu().numpy()
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), f"launch {launches}: split result differs from the one-block path"
        kept.append(float(pf.coarse.bnb_stats()["kept_per_particle"]))
    assert launches == 1000 and min(kept) < 100 and max(kept) > 1000, (launches, min(kept), max(kept))    # few survivors to thousands
    assert int(pf.coarse.t["sync"].abs().sum().item()) == 0         # every launch left the arrival counters at zero


# ---- corners the reference's code defines but its data never reaches (tests/golden/make_golden_edges.py) ----
def _nan_golden_filter(pkg, z, P, bnb):
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    size_m, unit, R, fov, beams, wall = z["nan_cfg"]
    world = synth.make_world(size_m, unit, seed=int(z["nan_world_seed"]), n_boxes=8)
    ogP = [size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, R, int(beams), wall]
    smP = list(z["nan_sm"][:7]) + [int(z["nan_sm"][7])]
    pf = pkg.ParticleFilter(P, ogP, smP, growable=False, rng=np.random.RandomState(0), bnb=bnb)
    for m in pf.engine.maps:
        m.upload(*synth.counts_from_world(world))
    return pf, world


@pytest.mark.parametrize("case", ["outside", "inside"])
def test_nan_first_argmax_adhoc_matches_reference_golden(pkg, case):
    """searchToMatch on the reference's own probSP with a heading prior that holds NaNs (arccos argument rounded past 1,
    Utils/ScanMatcher_OGBased.py:105-108): the cube carries NaN exactly where the reference's does and agrees within the bar
    elsewhere, the arg-max is the reference's (np.argmax: the FIRST NaN, :134), the confidence is NaN (:141), the matched pose
    is that pose's."""
    z = load_golden("edges.npz")
    size_m, unit, R, fov, beams, wall = z["nan_cfg"]
    prob = codec.decode_field(z["nan_prob_cls"], z["nan_prob_floor"], z["nan_prob_other"])
    og = pkg.OccupancyGrid(1, 1, {"x": 0.0, "y": 0.0}, unit, fov, int(beams), R, wall)
    sm = pkg.ScanMatcher(og, *z["nan_sm"][:7], int(z["nan_sm"][7]))
    pre = f"nan_{case}_"
    ex, ey, eth = z["nan_est"]
    _, _, matched, cube, conf = sm.searchToMatch(prob, ex, ey, eth, z["nan_ranges"], z["nan_xr"], z["nan_yr"], z["nan_sm"][0], z["nan_sm"][1],
                                                 unit, float(z[pre + "dist"]), float(z[pre + "psi"]), fineSearch=False, matchMax=True)
    want = z[pre + "cube"]
    assert np.isnan(want).any() and np.array_equal(np.isnan(cube), np.isnan(want))
    np.testing.assert_allclose(cube, want, rtol=RTOL_TIGHT, atol=0, equal_nan=True)
    assert int(sm.last["adhoc"]["argmax"]) == int(z[pre + "pick"]) == int(np.argmax(cube))
    assert np.isnan(conf) and np.isnan(sm.last["adhoc"]["log_confidence"])
    assert [matched["x"], matched["y"], matched["theta"]] == list(z[pre + "matched"])


NAN_PATHS = {"sweep": (False, False, {}), "pruned": (False, True, {}), "bnb": (True, False, {"SLAM2D_BNB_LEVELS": "1"}),
             "bnb_pruned": (True, True, {"SLAM2D_BNB_LEVELS": "1"}), "bnb_two_level": (True, False, {"SLAM2D_BNB_LEVELS": "2"})}


@pytest.mark.parametrize("path", sorted(NAN_PATHS))
@pytest.mark.parametrize("case", ["outside", "inside"])
def test_nan_first_argmax_every_scoring_path_matches_reference_golden(pkg, case, path, monkeypatch):
    """The same two calls through slam2d_match (field built on the device from the golden's world) on every scoring path --
    k_sweep (+ k_select), the prior-pruned sweep, k_bound + k_exact_select, both with pruning, the two-level bounds
    (k_bound1 / k_seed / k_bound2) -- against the reference golden AND the oracle: arg-max = the reference's first NaN,
    matched pose identical, confidence NaN; particles beside it with no heading (psi None) keep their finite results,
    identical to the oracle's."""
    bnb, prune, env = NAN_PATHS[path]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    z = load_golden("edges.npz")
    pre = f"nan_{case}_"
    P = 3
    pf, world = _nan_golden_filter(pkg, z, P, bnb)
    assert pf.coarse.bnb == bnb
    if bnb:
        assert pf.coarse.bnb_levels == int(env["SLAM2D_BNB_LEVELS"])
    eng = pf.engine
    psi, dist = float(z[pre + "psi"]), float(z[pre + "dist"])
    cs = np.tile([np.cos(psi), np.sin(psi)], (P, 1))
    import math
    cs[:, 0], cs[:, 1] = math.cos(psi), math.sin(psi)             # the reference's own expressions (:107)
    cs[1] = (np.nan, np.nan)                                       # particle 1: estMovingTheta = None
    d_est = eng.to_device(np.tile(z["nan_est"], (P, 1)))
    eng.match(pf.coarse, d_est, 3, eng.to_device(z["nan_ranges"]), dist, eng.to_device(cs), None, pf.m_coarse, prune=prune)
    eng.take_flags()
    c = eng.read_matches(pf.m_coarse)
    # the oracle on the same world
    size_m, unit, R, fov, beams, wall = z["nan_cfg"]
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    ogo = so.GridOracle(size_m, size_m, {"x": 0.0, "y": 0.0}, unit, fov, int(beams), R, wall)
    ogo.visited[:], ogo.total[:] = synth.counts_from_world(world)
    smo = so.MatcherOracle(ogo, *z["nan_sm"][:7], int(z["nan_sm"][7]))
    ex, ey, eth = z["nan_est"]
    xr, yr, prob = smo.frameSearchSpace(ex, ey, unit, z["nan_sm"][2], z["nan_sm"][6])
    for p in range(P):
        has_psi = p != 1
        mo, cube_o, conf_o = smo.searchToMatch(prob, ex, ey, eth, z["nan_ranges"], xr, yr, z["nan_sm"][0], z["nan_sm"][1], unit, dist,
                                               psi if has_psi else None, fineSearch=False, matchMax=True)
        assert int(c["argmax"][p]) == int(cube_o.argmax())
        assert (c["x"][p], c["y"][p], c["theta"][p]) == (mo["x"], mo["y"], mo["theta"])
        if has_psi:
            assert int(c["argmax"][p]) == int(z[pre + "pick"])                      # the reference's own first NaN
            assert [c["x"][p], c["y"][p], c["theta"][p]] == list(z[pre + "matched"])
            assert np.isnan(c["confidence"][p]) and np.isnan(c["log_confidence"][p]) and np.isnan(conf_o)
        else:
            np.testing.assert_allclose(c["confidence"][p], conf_o, rtol=RTOL_TIGHT)


def _wrap_grid(pkg, z):
    unit, R, fov, beams, wall = z["wrap_cfg"]
    return _grid_with_state(pkg, z["wrap_map"], z["wrap_X"], z["wrap_Y"], unit=unit, fov=fov, beams=int(beams), R=R, wall=wall)


@pytest.mark.parametrize("case", ["cols", "rows_cols", "coarse"])
def test_wrapped_field_index_matches_reference_golden(pkg, case):
    """frameSearchSpace on a map that grew >= 2 times on the high side, the window's first columns / rows occupied: the
    truncated field index of those cells is negative and the reference's NumPy scatter wraps it to the field's LAST
    columns / rows (Utils/ScanMatcher_OGBased.py:36-37).  Quantised probSP and probMin bit-exact against the reference."""
    z = load_golden("edges.npz")
    og = _wrap_grid(pkg, z)
    sm = pkg.ScanMatcher(og, *z["wrap_sm"][:7], int(z["wrap_sm"][7]))
    pre = f"wrap_{case}_"
    ex, ey, step, sigma, miss = z[pre + "args"]
    xr, yr, prob = sm.frameSearchSpace(ex, ey, step, sigma, miss)
    want = codec.decode_field(z[pre + "prob_cls"], z[pre + "prob_floor"], z[pre + "prob_other"])
    assert int(z[pre + "wrapped_cells"]) > 0 and og.map.growth_log == []
    assert np.array_equal(np.array(xr), z[pre + "xr"]) and np.array_equal(np.array(yr), z[pre + "yr"])
    assert prob.shape == want.shape
    level = sm._level(step, sigma, miss, sm.searchRadius, sm.searchHalfRad, False)
    assert int(((prob == 0) != (want == 0)).sum()) == 0
    assert level.frames()[0]["field_min"] == want.min()
    assert np.array_equal(level.field_cost(0), E.encode_cost(want, level.c.cost_scale))
    assert (want[:, -3:] == 0).sum() + (want[-3:, :] == 0).sum() > 0                  # the wrapped cells are there


@pytest.mark.parametrize("bnb", [False, True])
def test_wrapped_field_index_through_the_lazy_match(pkg, bnb):
    """The same maps through slam2d_match (lazy field build, merged scatter) with a scan whose endpoints reach the field's
    last tiles: every tile the scoring needed equals the reference's field there, arg-max / matched pose / confidence equal the
    oracle's on the oracle's (bit-equal to the reference's) field."""
    z = load_golden("edges.npz")
    unit, R, fov, beams, wall = z["wrap_cfg"]
    beams = int(beams)
    smP = list(z["wrap_sm"][:7]) + [int(z["wrap_sm"][7])]
    cases = ["cols", "rows_cols"]
    P = len(cases)
    pf = pkg.ParticleFilter(P, [10, 10, {"x": 0.0, "y": 0.0}, unit, fov, R, beams, wall], smP, growable=False,
                            rng=np.random.RandomState(0), bnb=bnb)
    v, t = codec.unpack_counts(z["wrap_map"])
    for p in range(P):
        pf.engine.maps[p] = pkg.MapState(z["wrap_X"], z["wrap_Y"], pf.engine.device)
        pf.engine.maps[p].upload(v, t)
    eng = pf.engine
    est = np.array([[z[f"wrap_{c}_args"][0], z[f"wrap_{c}_args"][1], 0.3 * (i + 1)] for i, c in enumerate(cases)])
    ranges = 4.93 - 0.4 * (np.arange(beams) % 5 == 0)                 # endpoints at the far tiles of the field
    eng.match(pf.coarse, eng.to_device(est), 3, eng.to_device(ranges), 0.3, None, None, pf.m_coarse)
    eng.take_flags()
    c = eng.read_matches(pf.m_coarse)
    ogo = so.GridOracle(1, 1, {"x": 0.0, "y": 0.0}, unit, fov, beams, R, wall)
    ogo.visited, ogo.total = v.copy(), t.copy()
    ogo.X, ogo.Y = z["wrap_X"].copy(), z["wrap_Y"].copy()
    ogo.mapXLim, ogo.mapYLim = [ogo.X[0], ogo.X[-1]], [ogo.Y[0], ogo.Y[-1]]
    smo = so.MatcherOracle(ogo, *smP)
    far_needed = 0
    for p, name in enumerate(cases):
        want = codec.decode_field(z[f"wrap_{name}_prob_cls"], z[f"wrap_{name}_prob_floor"], z[f"wrap_{name}_prob_other"])
        need = _tile_bits(pf.coarse, p)
        fh, fw = want.shape
        mask = np.kron(need, np.ones((16, 16), dtype=bool))[:fh, :fw]
        got = pf.coarse.field_cost(p)
        assert np.array_equal(got[mask], E.encode_cost(want, pf.coarse.c.cost_scale)[mask])
        far_needed += int(mask[:, -19:].any()) + int(mask[-19:, :].any())
        xr, yr, prob = smo.frameSearchSpace(est[p, 0], est[p, 1], unit, smP[2], smP[6])
        assert np.array_equal(prob, want)
        mo, cube_o, conf_o = smo.searchToMatch(prob, est[p, 0], est[p, 1], est[p, 2], ranges, xr, yr, smP[0], smP[1], unit, 0.3, None,
                                               fineSearch=False, matchMax=True)
        assert int(c["argmax"][p]) == int(cube_o.argmax())
        assert (c["x"][p], c["y"][p], c["theta"][p]) == (mo["x"], mo["y"], mo["theta"])
        np.testing.assert_allclose(c["confidence"][p], conf_o, rtol=RTOL_TIGHT)
    assert far_needed > 0                                              # the scoring did read tiles the wrapped cells reach


def test_large_launches_take_the_frame_kernel_and_agree_with_small_ones(pkg):
    """From 128 particles per launch slam2d_match runs k_frame_axis (frame, axis tables, beam endpoints tabulated once per particle)
    and k_occ_scatter as their own launches instead of inside k_endpoints' (round 6: on a filled machine the merged launch's repeated
    cos / sin and frame arithmetic cost more than two launches).  The same particles matched 130 at a time and 65 at a time -- both
    levels, soft-max draw -- give identical matches, cubes at the chosen poses and fields on the needed tiles."""
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    cfg = BNB_CASES["ref"]
    unit = cfg["unit"]
    origin = (-cfg["map_m"] / 2, -cfg["map_m"] / 2)
    big, world = _synthetic_filter(pkg, cfg, 130, None)
    small, _ = _synthetic_filter(pkg, cfg, 65, None)
    poses = synth.random_walk(world, unit, origin, 3, seed=3, step=0.3, max_radius=6.0)
    rs = np.random.RandomState(11)
    for s in (1, 2):
        ranges = synth.raycast(world, unit, origin, poses[s], cfg["fov"], cfg["beams"], cfg["max_range"])
        k = rs.randint(-3, 4, size=(130, 2))
        est = np.column_stack((poses[s - 1][0] + k[:, 0] * unit, poses[s - 1][1] + k[:, 1] * unit, poses[s][2] + rs.normal(0, 0.03, 130)))
        uni = rs.random_sample(130)
        psi = np.tile([np.cos(0.4), np.sin(0.4)], (130, 1))
        out = []
        for pf, n in ((big, 130), (small, 65)):
            eng = pf.engine
            d_rng = eng.to_device(ranges)
            eng.match(pf.coarse, eng.to_device(est[:n]), 3, d_rng, 0.3, eng.to_device(psi[:n]), eng.to_device(uni[:n]), pf.m_coarse)
            eng.match(pf.fine, pf.m_coarse, E.MATCH_DOUBLES, d_rng, 0.3, None, None, pf.m_fine)
            eng.take_flags()
            c, f = eng.read_matches(pf.m_coarse).copy(), eng.read_matches(pf.m_fine).copy()
            out.append((c, f, [pf.coarse.frames()[p]["field_min"] for p in range(65)], [pf.fine.field_cost(p).copy() for p in (0, 64)],
                        [_tile_bits(pf.fine, p) for p in (0, 64)]))
        (cb, fb, mb, fieldb, needb), (cs, fs, ms, fields, needs) = out
        assert np.array_equal(cb[:65].view(np.uint8), cs.view(np.uint8)) and np.array_equal(fb[:65].view(np.uint8), fs.view(np.uint8))
        assert mb == ms
        for a, b, na, nb in zip(fieldb, fields, needb, needs):
            assert np.array_equal(na, nb)
            fh, fw = a.shape
            mask = np.kron(na, np.ones((16, 16), dtype=bool))[:fh, :fw]
            assert mask.any() and np.array_equal(a[mask], b[mask])


def test_a_gate_timeout_is_survived(pkg, intel_readings, monkeypatch):
    """A device-side wait that runs into its bound (here: the commit's gate, made to wait for match arrivals that never come by
    taking some away behind the filter's back, twice in a run; bound 0.3 s) must not end the run: the gate voids the scan before
    anything is written (the report carries SLAM2D_F_SCAN_VOIDED | SLAM2D_F_SYNC_TIMEOUT), the driver runs that scan and its successor
    through the calls that wait for nothing on the device, resets the sync words and goes on in groups -- and the whole 910-scan
    golden run still comes out pose for pose, draw for draw."""
    import hashlib
    monkeypatch.setenv("SLAM2D_SYNC_TIMEOUT_MS", "300")
    z = load_golden("flow_fastslam_long.npz")
    n_particles, n_scans, seed, map_m = (int(v) for v in z["cfg"])
    u = 0.02
    ogP = [map_m, map_m, intel_readings[0], u, np.pi, 10, 180, 5 * u]
    pf = pkg.ParticleFilter(n_particles, ogP, list(REF_SM), rng=np.random.RandomState(seed), groups=3)
    seen = []

    def on_scan(count, f, unb):
        assert unb == bool(z["unbalanced"][count - 1]), f"scan {count}"
        np.testing.assert_allclose(f.weights, z["weights"][count - 1], rtol=RTOL, atol=1e-290, err_msg=f"scan {count}")
        assert np.array_equal(f.prev_matched, z["matched"][count - 1]), f"scan {count}"
        seen.append(count)
        if count in (200, 600) and f._grp is not None:
            f._grp.sync[61] -= 2                      # two arrivals fewer than the next gates will wait for
    resamples = pf.run(intel_readings[:n_scans], force_resample=set(int(v) for v in z["force_resample"]), on_scan=on_scan)
    assert seen == list(range(1, n_scans + 1))
    assert pf.stats.get("sync_timeouts", 0) >= 1, pf.stats
    assert pf._grp.devsync and pf.n_groups == 3
    got = np.array([np.concatenate(([c], idx)) for c, idx in resamples]).reshape(-1, n_particles + 1)
    assert np.array_equal(got, z["resamples"])
    for p, m, sha, lim in zip(pf.particles, pf.engine.maps, z["maps_sha"], z["final_lims"]):
        assert [m.lim_x[0], m.lim_x[1], m.lim_y[0], m.lim_y[1]] == list(lim)
        packed = codec.pack_counts(p.og.occupancyGridVisited, p.og.occupancyGridTotal)
        assert hashlib.sha256(packed.tobytes()).digest() == sha.tobytes()


@pytest.mark.parametrize("case", ["config2", "hokuyo"])
def test_lds_bounds_are_deterministic(pkg, case, monkeypatch):
    """k_bound_lds launch after launch on the same inputs (config 2's short lists; 1081 beams: run-length compressed lists, one
    seed per wave): the bounds of every pose tile, the threshold bnb_best and the match come out bit for bit the same 150 times --
    seeds are skipped by a racy read of bnb_best, which may change WHICH seeds are scored, never the final threshold."""
    monkeypatch.setenv("SLAM2D_BNB_LEVELS", "1")
    synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
    cfg = BNB_CASES[case]
    P, unit = 10, cfg["unit"]
    origin = (-cfg["map_m"] / 2, -cfg["map_m"] / 2)
    pf, world = _synthetic_filter(pkg, cfg, P, True)
    assert pf.coarse.bnb_levels == 1 and "gmin2b" in pf.coarse.t
    eng = pf.engine
    poses = synth.random_walk(world, unit, origin, 3, seed=3, step=0.3, max_radius=6.0)
    rs = np.random.RandomState(5)
    ranges = synth.raycast(world, unit, origin, poses[1], cfg["fov"], cfg["beams"], cfg["max_range"])
    k = rs.randint(-3, 4, size=(P, 2))
    est = np.column_stack((poses[0][0] + k[:, 0] * unit, poses[0][1] + k[:, 1] * unit, poses[1][2] + rs.normal(0, 0.03, P)))
    d_est, d_rng = eng.to_device(est), eng.to_device(ranges)
    d_psi, d_u = eng.to_device(np.tile([np.cos(0.2), np.sin(0.2)], (P, 1))), eng.to_device(rs.random_sample(P))
    first = None
    for _ in range(150):
        eng.match(pf.coarse, d_est, 3, d_rng, 0.3, d_psi, d_u, pf.m_coarse, prune=False)
        eng.take_flags()
        got = (pf.coarse.t["bounds"].cpu().numpy().copy(), pf.coarse.t["bnb_best"].cpu().numpy().copy(), pf.m_coarse.cpu().numpy().copy())
        if first is None:
            first = got
            assert np.isfinite(first[0]).any()
        else:
            for a, b in zip(first, got):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
