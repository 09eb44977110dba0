"""The oracle against the LIVE reference, when it is present (development container only;
skipped on the GPU box, where /root/reference does not exist).  The committed golden vectors
(tests/test_oracle_golden.py) are the portable form of the same pinning."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Utils")), reason="reference not present")

REF_SM = (1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5)


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    saved = list(sys.path)
    shadow = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "Utils" or k.startswith("Utils.")}
    sys.path[:0] = [REF, os.path.join(REF, "Algorithm")]
    try:
        from Utils.OccupancyGrid import OccupancyGrid
        from Utils.ScanMatcher_OGBased import ScanMatcher
        import Utils.ScanMatcher_OGBased as sm_mod
        import FastSlam
        assert OccupancyGrid.__module__ == "Utils.OccupancyGrid" and REF in sm_mod.__file__
        yield dict(OccupancyGrid=OccupancyGrid, ScanMatcher=ScanMatcher, sm_mod=sm_mod, FastSlam=FastSlam)
    finally:
        sys.path[:] = saved
        for k in [k for k in sys.modules if k == "Utils" or k.startswith("Utils.") or k == "FastSlam"]:
            sys.modules.pop(k)
        sys.modules.update(shadow)


def test_blur_is_scipy_bit_for_bit():
    from scipy.ndimage import gaussian_filter
    from oracle import slam_oracle as so
    rs = np.random.RandomState(0)
    for shape, sigma in [((249, 249), 0.4), ((300, 211), 2.0), ((64, 50), 1.3), ((40, 33), 3.7)]:
        a = np.where(rs.rand(*shape) < 0.05, 0.0, np.log(0.15))
        assert np.array_equal(gaussian_filter(a, sigma=sigma), so.blur_reflect(a, sigma))


def test_scanmatch_flow_equals_reference(ref, intel_readings):
    from oracle import slam_oracle as so
    n = 14
    r0 = intel_readings[0]
    og_r = ref["OccupancyGrid"](10, 10, r0, 0.02, np.pi, 180, 10, 0.1)
    sm_r = ref["ScanMatcher"](og_r, *REF_SM)
    sm_mod = ref["sm_mod"]
    xs, ys, poses, confs = [], [], [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for count, raw in enumerate(intel_readings[:n], start=1):
            if count == 1:
                pr = pm = None
                matched, conf = raw, 1
            else:
                est, dist, psi, rawth = sm_mod.updateEstimatedPose(raw, prev_m, prev_r, pr, pm)
                matched, conf = sm_r.matchScan(est, dist, psi, count)
                pr, pm = rawth, sm_mod.getMovingTheta(matched, xs, ys)
            og_r.updateOccupancyGrid(matched)
            xs.append(matched["x"]); ys.append(matched["y"])
            prev_m, prev_r = matched, raw
            poses.append((matched["x"], matched["y"], matched["theta"])); confs.append(conf)
    og_o = so.GridOracle(10, 10, r0, 0.02, np.pi, 180, 10, 0.1)
    sm_o = so.MatcherOracle(og_o, *REF_SM)
    out, oc = so.run_scanmatch_flow(intel_readings, og_o, sm_o, max_scans=n)
    assert [(m["x"], m["y"], m["theta"]) for m in out] == poses and list(oc) == confs
    assert np.array_equal(og_r.occupancyGridVisited, og_o.visited) and np.array_equal(og_r.occupancyGridTotal, og_o.total)
    assert np.array_equal(og_r.OccupancyGridX[0], og_o.X) and np.array_equal(og_r.OccupancyGridY[:, 0], og_o.Y)


def test_fastslam_flow_equals_reference(ref, intel_readings):
    from oracle import slam_oracle as so
    u, n_particles, n = 0.02, 3, 9
    ogP = [20, 20, intel_readings[0], u, np.pi, 10, 180, 5 * u]
    np.random.seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        pf = ref["FastSlam"].ParticleFilter(n_particles, ogP, list(REF_SM))
        wr = []
        for c, r in enumerate(intel_readings[:n], start=1):
            pf.updateParticles(r, c)
            pf.weightUnbalanced()
            wr.append([p.weight for p in pf.particles])
            if c == 6:
                pf.resample()
    np.random.seed(3)
    po = so.ParticleFilterOracle(n_particles, ogP, list(REF_SM))
    wo = []
    for c, r in enumerate(intel_readings[:n], start=1):
        po.updateParticles(r, c)
        po.weightUnbalanced()
        wo.append([p.weight for p in po.particles])
        if c == 6:
            po.resample()
    assert np.array_equal(np.array(wr, dtype=float), np.array(wo, dtype=float))
    for a, b in zip(pf.particles, po.particles):
        assert np.array_equal(a.og.occupancyGridVisited, b.og.visited) and a.xTrajectory == b.xTrajectory
