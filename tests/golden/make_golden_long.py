#!/usr/bin/env python3
"""Long closed-loop golden runs of the reference's FastSLAM (development container only:
needs /root/reference).  Companion of make_golden.py; writes

  flow_fastslam_long.npz    6 particles x 910 scans (the whole Intel log), seed 0, 50 m map
                            (Algorithm/FastSlam.py:197-207 defaults): the NATURAL
                            weightUnbalanced() trigger (:37) fires and the maps grow.
  flow_fastslam_growth.npz  3 particles x 150 scans, seed 1, 10 m initial map: per-beam growth
                            inside the first update (Utils/OccupancyGrid.py:144-147), heavy
                            search-window growth, forced resamples between particles whose maps
                            have grown differently.

    python tests/golden/make_golden_long.py [long|growth|csail|params|all]

Per scan and particle the fixtures hold the uniform consumed by the soft-max draw, the matched
pose, the raw and the normalised weight, the variance / unbalanced decision, the resample
draws, every change of a particle's map shape, and the SHA-256 + limits of the final maps.
Data only -- no reference source text.
"""
import hashlib
import io
import sys
import time

import numpy as np

import make_golden as mg       # imports the reference (sys.path set up there)

ref_fs = mg.ref_fs
codec = mg.codec


def run(name, readings, n_particles, n_scans, seed, map_m, force_resample=()):
    u = 0.02
    ogP = [map_m, map_m, readings[0], u, np.pi, 10, len(readings[0]['range']), 5 * u]      # Algorithm/FastSlam.py:204 order
    smP = list(mg.REF_DEFAULT_SM)
    np.random.seed(seed)
    with mg.quiet():
        pf = ref_fs.ParticleFilter(n_particles, ogP, smP)
    W, V, M, C, U, RS, UNB, SHAPES = [], [], [], [], [], [], [], []
    last_shape = [None] * n_particles
    t = time.time()
    for count, raw in enumerate(readings[:n_scans], start=1):
        us = []
        for p in pf.particles:
            st = np.random.get_state()
            with mg.quiet():
                p.update(raw, count)
            probe = np.random.RandomState(); probe.set_state(st)
            now = np.random.get_state()
            changed = now[2] != st[2] or not np.array_equal(now[1], st[1])
            us.append(probe.random_sample() if changed else np.nan)
        C.append([p.weight for p in pf.particles])        # pre-normalisation weights
        with mg.quiet():
            unb = pf.weightUnbalanced()
        n = pf.numParticles
        V.append(sum((p.weight - 1 / n) ** 2 for p in pf.particles))
        W.append([p.weight for p in pf.particles])
        M.append([[p.prevMatchedReading["x"], p.prevMatchedReading["y"], p.prevMatchedReading["theta"]]
                  for p in pf.particles])
        U.append(us); UNB.append(unb)
        for i, p in enumerate(pf.particles):             # shape after this scan's update, before a resample
            sh = p.og.occupancyGridVisited.shape
            if sh != last_shape[i]:
                SHAPES.append([count, i, sh[0], sh[1]])
                last_shape[i] = sh
        if unb or count in force_resample:
            st = np.random.get_state()
            with mg.quiet():
                pf.resample()
            probe = np.random.RandomState(); probe.set_state(st)
            draw = probe.choice(np.arange(n), n, p=np.array(W[-1]))
            RS.append(np.concatenate(([count], draw)))
            last_shape = [last_shape[j] for j in draw]
        if count % 50 == 0:
            print(f"    {name}: scan {count}/{n_scans}, {time.time() - t:.0f} s, resamples {[int(r[0]) for r in RS]}",
                  flush=True)
    print(f"  reference FastSLAM {n_particles} x {n_scans}: {time.time() - t:.1f} s")
    maps_sha = [np.frombuffer(hashlib.sha256(codec.pack_counts(
        p.og.occupancyGridVisited, p.og.occupancyGridTotal).tobytes()).digest(), dtype=np.uint8)
        for p in pf.particles]
    lims = [[p.og.mapXLim[0], p.og.mapXLim[1], p.og.mapYLim[0], p.og.mapYLim[1]] for p in pf.particles]
    shapes = [list(p.og.occupancyGridVisited.shape) for p in pf.particles]
    mg.save(name, weights=np.array(W, dtype=np.float64), raw_weights=np.array(C, dtype=np.float64),
            variance=np.array(V), matched=np.array(M), uniforms=np.array(U), unbalanced=np.array(UNB),
            resamples=np.array(RS).reshape(-1, n_particles + 1).astype(np.int64),
            cfg=np.array([n_particles, n_scans, seed, map_m]), beams=np.int64(len(readings[0]['range'])),
            force_resample=np.array(force_resample, dtype=np.int64), maps_sha=np.array(maps_sha),
            final_lims=np.array(lims), final_shapes=np.array(shapes), shape_events=np.array(SHAPES, dtype=np.int64))


def csail(n_scans=80):
    """A second dataset through the reference's single-trajectory flow (Utils/ScanMatcher_OGBased.py:226-256): the
    bundled CSAIL log, 361 beams over pi (722 spokes, 59 search angles) -- a beam count other than the Intel log's 180.
    Writes the log re-encoded (csail_gfs.npz) and flow_scanmatch_csail.npz."""
    import json
    import os
    d = json.load(open(os.path.join(mg.REF, "DataSet/PreprocessedData/csail_gfs")))["map"]
    readings = [d[k] for k in sorted(d.keys())]
    rng = np.array([r["range"] for r in readings])
    cm = np.rint(rng * 100)
    assert np.array_equal(cm / 100.0, rng) and cm.max() < 65536
    mg.save("csail_gfs.npz", range_cm=cm.astype(np.uint16), pose=np.array([[r["x"], r["y"], r["theta"]] for r in readings]))
    beams = len(readings[0]["range"])
    og = mg.OccupancyGrid(10, 10, readings[0], 0.02, np.pi, beams, 10, 0.1)
    sm = mg.ScanMatcher(og, *mg.REF_DEFAULT_SM)
    t = time.time()
    with mg.quiet():
        poses, confs = mg.flow(readings, og, sm, n_scans)
    print(f"  reference flow over {n_scans} CSAIL scans ({beams} beams): {time.time() - t:.1f} s")
    run("flow_fastslam_csail.npz", readings, 3, 60, 2, 10, force_resample=(20, 40))
    mg.save("flow_scanmatch_csail.npz", poses=poses, confs=confs, beams=np.int64(beams),
            final_shape=np.array(og.occupancyGridVisited.shape),
            final_lims=np.array([og.mapXLim[0], og.mapXLim[1], og.mapYLim[0], og.mapYLim[1]]),
            final_map_sha=np.frombuffer(hashlib.sha256(
                codec.pack_counts(og.occupancyGridVisited, og.occupancyGridTotal).tobytes()).digest(), dtype=np.uint8))


def params(n_scans=45):
    """The reference's single-trajectory flow over the Intel log with constructor parameters OTHER than its defaults, so
    that the parametrised code paths are pinned too: unit 0.04, coarse factor 4 (coarse sigma 0.75 -> blur radius 3, a
    radius without a specialised kernel; fine radius 12), lidar range 8 m, search radius 1.2 / half angle 0.2 (cubes
    21 x 15 x 15 and 21 x 9 x 9), wall 0.12, miss probability 0.2, other sigmas."""
    readings = mg.load_intel()
    og_args = (12, 12, readings[0], 0.04, np.pi, len(readings[0]["range"]), 8, 0.12)
    sm_args = (1.2, 0.2, 3, 0.15, 0.3, 0.25, 0.2, 4)
    og = mg.OccupancyGrid(*og_args)
    sm = mg.ScanMatcher(og, *sm_args)
    t = time.time()
    with mg.quiet():
        poses, confs = mg.flow(readings, og, sm, n_scans)
    print(f"  reference flow, alternative parameters, {n_scans} scans: {time.time() - t:.1f} s")
    mg.save("flow_scanmatch_params.npz", poses=poses, confs=confs,
            og_args=np.array([12, 12, 0.04, np.pi, len(readings[0]["range"]), 8, 0.12]), sm_args=np.array(sm_args, dtype=np.float64),
            final_shape=np.array(og.occupancyGridVisited.shape),
            final_lims=np.array([og.mapXLim[0], og.mapXLim[1], og.mapYLim[0], og.mapYLim[1]]),
            final_map_sha=np.frombuffer(hashlib.sha256(
                codec.pack_counts(og.occupancyGridVisited, og.occupancyGridTotal).tobytes()).digest(), dtype=np.uint8))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("params", "all"):
        params()
    if what == "params":
        return
    if what in ("csail", "all"):
        csail()
    if what == "csail":
        return
    readings = mg.load_intel()
    if what in ("growth", "all"):
        run("flow_fastslam_growth.npz", readings, 3, 150, 1, 10, force_resample=(30, 75, 120))
    if what in ("long", "all"):
        run("flow_fastslam_long.npz", readings, 6, len(readings), 0, 50)


if __name__ == "__main__":
    main()
