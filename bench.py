#!/usr/bin/env python3
"""Benchmark of the scan-matching / map-update hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|ref2level|config5] [--particles P]

A "step" is one lidar scan processed for every particle of the rank: search-field build
from each particle's own map, pose-cube sweep with soft-max pose draw and confidence,
occupancy-grid update at the matched pose, weight update + normalisation (sharded runs:
one RCCL all-gather of the normaliser).  All inputs of all steps (ranges, pose estimates,
uniforms) are resident in HBM before the timed region; nothing is copied or synchronised
inside it.

Default workload = BASELINE.json configs[1] ("config2": 800x800 @ 0.1 m search field,
36x41x41 pose cube, 180 beams) with the north-star's 64 particles per GPU.  Prints ONE JSON
line (rank 0).  The K-step timed block (barrier + synchronize on both sides) is repeated
--repeats times; value / ms_per_step are the median block.
N > 1: one rank per GPU, particles sharded P per rank (weak scaling), value = whole-job
particle-scans/s.  Under a launcher (torch.distributed.run sets WORLD_SIZE) the ranks are
taken from the environment; `python bench.py --gpus N` on its own re-executes itself under
torch.distributed.run with N ranks.  Either way the run FAILS unless exactly N ranks joined.
`--backend gloo --share-gpu` is a dry mode for boxes with fewer GPUs than ranks.
"""
import argparse
import ctypes as C
import importlib
import json
import math
import hashlib
import os
import socket
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# Particle groups run on their own HIP streams, and streams that share a hardware queue serialise: the runtime's default of 4
# queues holds two groups + the default stream; four groups need more (read by the HIP runtime when it initialises: before torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)

WORKLOADS = {
    # BASELINE.json configs[1]; SURVEY.md 8(d) "config 2": single-level match at 0.1 m
    "config2": dict(unit=0.1, max_range=34.5, fov=math.pi, beams=180, map_m=100.0, search_radius=2.05,
                    half_rad=0.30, sigma_cells=2, miss=0.15, coarse_factor=1, levels=1, wall=0.5,
                    move_sigma=0.1, max_dev=0.25, turn_sigma=0.3,
                    note="800x800@0.1m search field, 36x41x41 pose cube, 180 beams, single-level match + map update"),
    # reference defaults (Algorithm/FastSlam.py:197-199): two-level match at 0.02 m
    "ref2level": dict(unit=0.02, max_range=10.0, fov=math.pi, beams=180, map_m=50.0, search_radius=1.4,
                      half_rad=0.25, sigma_cells=2, miss=0.15, coarse_factor=5, levels=2, wall=0.1,
                      move_sigma=0.1, max_dev=0.25, turn_sigma=0.3,
                      note="reference defaults: coarse 249^2/30x27x27 + fine 1241^2/30x11x11, 180 beams, map update"),
    # BASELINE.json configs[4] (per-GPU slice): 2000x2000 @ 0.05 m map, 1081-beam 270-degree scans, 128 particles/GPU
    "config5": dict(unit=0.05, max_range=30.0, fov=1.5 * math.pi, beams=1081, map_m=100.0, search_radius=2.05,
                    half_rad=0.30, sigma_cells=2, miss=0.15, coarse_factor=2, levels=2, wall=0.25,
                    move_sigma=0.1, max_dev=0.25, turn_sigma=0.3,
                    note="2000x2000@0.05m map, 1081 beams over 1.5 pi, coarse 702^2/139x41x41 + fine 1403^2/139x5x5"),
}
WORKLOAD_PARTICLES = {"config5": 128}
PROF_EVERY_MAX = int(os.environ.get("SLAM2D_BENCH_PROF_EVERY", "41"))  # (round 4: 7 -> 41: measured 0.1312 ms per step with a pair round every 7th launch, 0.1280 with every 51st, 0.1419 with every launch)
# the dominant kernel keeps a HIP event pair around every n-th of its launches inside the timed region (n <= 41, chosen so
# that at least 13 launches are timed: an event pair holds the stream for ~6 us on each side of the kernel)
KERNEL_SOURCE = os.path.join(REPO, "slam-2d-lidar-scan_amd", "csrc", "slam2d.hip")


def source_sha256():
    with open(KERNEL_SOURCE, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS) + ["config3"],
                    help="config3: the CLOSED loop over the bundled Intel log (ParticleFilter.run, host decisions, growth, resampling; "
                         "sharded over the ranks with --gpus N: BASELINE config 4 is --gpus 8 --workload config3 --total-particles 512)")
    ap.add_argument("--resample-every", type=int, default=100, help="config3: force a resample every this many scans (0: only the reference's own trigger)")
    ap.add_argument("--particles", type=int, default=None, help="particles per GPU (default 64; config5: 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the variant measurements (brute-force sweep, prior-pruned "
                    "sweep, reference defaults, config-5 slice, config-3 closed loop)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of each CPU baseline leg (two legs: one process, all cores)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="processes of the all-cores CPU baseline (0: every host core)")
    ap.add_argument("--repeats", type=int, default=5, help="the K-step timed block is run this many times; value / ms_per_step are the "
                    "median block (every block is bracketed by barrier + synchronize)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of a multi-rank run "
                    "(nccl = RCCL over xGMI; gloo: dry mode, the 24-byte all-gather hops through host memory)")
    ap.add_argument("--share-gpu", action="store_true", help="dry mode: ranks beyond the visible GPUs share them (rank r -> GPU r %% count)")
    ap.add_argument("--groups", type=int, default=None, help="particle groups per GPU, each on its own HIP stream (default: 2 from 32 "
                    "particles per GPU up, SLAM2D_BENCH_GROUPS overrides; 1 = one stream)")
    ap.add_argument("--total-particles", type=int, default=None, help="STRONG scaling: this many particles over all ranks (e.g. 512 = "
                    "BASELINE config 4, 1024 with --workload config5 = config 5); default: --particles per rank (weak scaling)")
    ap.add_argument("--spawn-check", action="store_true", help="launch plumbing only (no GPU work): every rank joins a gloo group, "
                    "rank 0 prints the rank count it saw")
    return ap.parse_args()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU on
    127.0.0.1 (the form the driver itself uses for N > 1).  Returns the launcher's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


class Scenario:
    """Synthetic world, trajectory, scans and per-particle pose estimates of one workload."""

    MODES = ("tracked", "worst", "displaced")

    def __init__(self, cfg, P, n_scans, seed=0, rank=0, mode="tracked"):
        """mode "tracked": the scans are ray-cast from the very world the maps hold and every estimate lies within two cells of
        the true pose -- the steady state of a converged filter, and the BEST case for the branch and bound (1 % of the pose tiles
        survive).  "worst": SURVEY 8(d)'s structure-free input -- ranges ~ U(1, 0.999 R), seed 0, every endpoint cell unique,
        nothing fits the map (the input behind the reference's 148.6 ms CPU figure).  "displaced": the tracked scans, but every
        estimate 1.5 m and 0.2 rad off the true pose: the match lies far from the motion prior's ring."""
        assert mode in self.MODES
        synth = importlib.import_module("slam-2d-lidar-scan_amd.synth")
        self.cfg, self.P, self.mode = cfg, P, mode
        u = cfg["unit"]
        self.world = synth.make_world(cfg["map_m"], u, seed=seed, n_boxes=60)
        self.origin = (-cfg["map_m"] / 2, -cfg["map_m"] / 2)
        self.visited, self.total = synth.counts_from_world(self.world)
        poses = synth.random_walk(self.world, u, self.origin, n_scans + 1, seed=seed + 1, step=0.4, max_radius=7.0)
        self.poses = np.array(poses)
        self.ranges = np.stack([synth.raycast(self.world, u, self.origin, p, cfg["fov"], cfg["beams"],
                                              cfg["max_range"]) for p in poses[1:]])
        rs = np.random.RandomState(1000 + rank)
        # per-particle estimates: previous true pose + a small lattice offset (a spread-out particle cloud)
        k = rs.randint(-2, 3, size=(n_scans, P, 2))
        self.est = np.empty((n_scans, P, 3))
        self.est[:, :, 0] = self.poses[:-1, None, 0] + k[:, :, 0] * u
        self.est[:, :, 1] = self.poses[:-1, None, 1] + k[:, :, 1] * u
        self.est[:, :, 2] = self.poses[1:, None, 2] + rs.normal(0, 0.02, size=(n_scans, P))
        d = self.poses[1:, :2] - self.poses[:-1, :2]
        self.dist = np.hypot(d[:, 0], d[:, 1])
        self.psi = np.arctan2(d[:, 1], d[:, 0])
        self.uniform = rs.random_sample((n_scans, P))
        # heading of the odometry step with a little odometry noise (an exactly lattice-aligned heading makes the
        # reference's arccos argument round above 1 -> NaN thetaWeight along that direction, Q-list in SURVEY App. A)
        self.psi = self.psi + rs.normal(0.0, 0.01, size=self.psi.shape)
        if mode == "worst":
            self.ranges = np.random.RandomState(0).uniform(1.0, 0.999 * cfg["max_range"], size=self.ranges.shape)
        elif mode == "displaced":
            rd = np.random.RandomState(2000 + rank)
            ang = rd.uniform(-np.pi, np.pi, size=(n_scans, P))
            k = int(round(1.5 / u))
            self.est[:, :, 0] += u * np.rint(k * np.cos(ang))
            self.est[:, :, 1] += u * np.rint(k * np.sin(ang))
            self.est[:, :, 2] += 0.2 * rd.choice([-1.0, 1.0], size=(n_scans, P))


class HotPath:
    """Device state + the per-step launch sequence (no host round trips)."""

    def __init__(self, cfg, P, scen, device):
        pkg = importlib.import_module("slam-2d-lidar-scan_amd")
        E = importlib.import_module("slam-2d-lidar-scan_amd.engine")
        self.E, self.cfg, self.P = E, cfg, P
        u = cfg["unit"]
        self.lidar = E.LidarModel.get(u, cfg["max_range"], cfg["fov"], cfg["beams"], cfg["wall"])
        seed_map = E.MapState.create(cfg["map_m"], cfg["map_m"], {"x": 0.0, "y": 0.0}, u, device)
        seed_map.upload(scen.visited, scen.total)
        maps = [seed_map] + [seed_map.clone() for _ in range(P - 1)]
        self.eng = E.ParticleEngine(self.lidar, maps, device)
        common = dict(search_radius_ctor=cfg["search_radius"], half_rad=cfg["half_rad"], move_sigma=cfg["move_sigma"],
                      max_move_dev=cfg["max_dev"], turn_sigma=cfg["turn_sigma"])
        cf = cfg["coarse_factor"]
        self.coarse = E.SearchLevel(self.lidar, P, device, step=cf * u, sigma=cfg["sigma_cells"] / cf,
                                    miss_prob=cfg["miss"], radius=cfg["search_radius"], fine=False, **common)
        self.fine = None
        if cfg["levels"] == 2:
            self.fine = E.SearchLevel(self.lidar, P, device, step=u, sigma=cfg["sigma_cells"],
                                      miss_prob=cfg["miss"] ** (2 / cf), radius=cf * u, fine=True, **common)
        e = self.eng
        self.d_ranges = e.to_device(scen.ranges)
        self.d_est = e.to_device(scen.est)
        self.d_uniform = e.to_device(scen.uniform)
        psi_cs = np.stack([np.cos(scen.psi), np.sin(scen.psi)], axis=1)            # shared heading prior
        self.d_psi = e.to_device(np.repeat(psi_cs[:, None, :], P, axis=1))
        self.dist = scen.dist
        self.m_coarse, self.m_fine = e.match_buffer("coarse"), e.match_buffer("fine")
        self.d_logw = torch.zeros(P, dtype=torch.float64, device=device)
        self.d_w = torch.zeros(P, dtype=torch.float64, device=device)
        self.d_stats = torch.zeros(2, dtype=torch.float64, device=device)
        self.L = E._lib.lib()
        self.sharded = dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SLAM2D_FORCE_DIST") == "1")
        self.par = importlib.import_module("slam-2d-lidar-scan_amd.parallel")
        self.total_particles = P * (dist.get_world_size() if dist.is_initialized() else 1)
        self.normalizer = None
        self.lazy = os.environ.get("SLAM2D_BENCH_FULL_FIELD", "0") != "1"   # slam2d_match vs field_build + sweep
        self.prune = False      # SLAM2D_MATCH_PRUNE_BY_PRIOR: measured separately ("variants" in the JSON line)

    # -- what tests and the accounting read, the same on HotPathGroups --
    @property
    def groups(self):
        return [self]

    def levels(self):
        return [lv for lv in (self.coarse, self.fine) if lv is not None]

    def maps(self):
        return list(self.eng.maps)

    def matches(self, which):
        return self.eng.read_matches(self.m_coarse if which == "coarse" else self.m_fine).copy()

    def weights(self):
        return self.d_w.cpu().numpy().copy()

    def take_flags(self):
        return self.eng.take_flags()

    def match(self, s):
        """Both levels of the scan match of this object's particles for scan s; returns the buffer that holds the matched poses."""
        e, E = self.eng, self.E
        est, rng = self.d_est[s], self.d_ranges[s]
        if not self.lazy:
            e.field_build(self.coarse, est, 3)
        if self.lazy:
            e.match(self.coarse, est, 3, rng, float(self.dist[s]), self.d_psi[s], self.d_uniform[s], self.m_coarse,
                    prune=self.prune)
        else:
            e.sweep(self.coarse, est, 3, rng, float(self.dist[s]), self.d_psi[s], self.d_uniform[s], self.m_coarse)
        final = self.m_coarse
        if self.fine is not None:
            if not self.lazy:
                e.field_build(self.fine, self.m_coarse, E.MATCH_DOUBLES)
            (e.match if self.lazy else e.sweep)(self.fine, self.m_coarse, E.MATCH_DOUBLES, rng, float(self.dist[s]),
                                                None, None, self.m_fine)
            final = self.m_fine
        return final

    def update_local(self, s, final, part):
        """Map update at the matched poses + the local half of a split normaliser ([max, sum, sum of squares] of these
        particles' log-weights -> the three doubles of `part`) in one launch."""
        self.eng.grid_update_weights_local(final, self.E.MATCH_DOUBLES, self.d_ranges[s], self.d_logw, self.m_coarse.data_ptr() + 32,
                                           self.E.MATCH_DOUBLES, part)

    def match_and_update(self, s):
        """Both levels of the scan match and the map update of all particles for scan s."""
        e, E = self.eng, self.E
        rng = self.d_ranges[s]
        final = self.match(s)
        if self.sharded:   # the rank-local half of the normaliser rides in the update's launch; collective + merge follow
            if self.normalizer is None:
                self.normalizer = self.par.ShardedNormalizer(self.L, E._lib.check, self.d_logw.device, self.total_particles,
                                                             overlap=os.environ.get("SLAM2D_BENCH_OVERLAP", "0") == "1")
                # (overlap: collective + merge on a side stream.  Bit-identical, but at one rank its events and stream
                # switches cost the host more than the collective's latency: 0.207 vs 0.187 ms/step -- off by default)
            e.grid_update_weights_local(final, E.MATCH_DOUBLES, rng, self.d_logw, self.m_coarse.data_ptr() + 32, E.MATCH_DOUBLES,
                                        self.normalizer.part, normalizer=self.normalizer)
        else:       # one launch: the normaliser rides beside the update (it reads only what the match wrote)
            e.grid_update_weights(final, E.MATCH_DOUBLES, rng, self.d_logw, self.m_coarse.data_ptr() + 32, E.MATCH_DOUBLES,
                                  self.d_w, self.d_stats)         # +32: log_confidence

    def normalise(self):
        """weight *= confidence, then the normaliser over all particles of the job."""
        E = self.E
        if self.sharded:
            self.normalizer(self.d_logw, self.m_coarse.data_ptr() + 32, E.MATCH_DOUBLES, self.d_w, self.d_stats, local_done=True)
        # (one GPU: the normaliser went out with the map update, match_and_update)

    def step(self, s):
        self.match_and_update(s)
        self.normalise()

    def algorithmic_bytes(self, scen):
        """SURVEY.md 8(d) per particle-scan, with the build's real storage: packed uint32 cell
        (both counts) => 8 B per updated cell, 1 bit per map cell read by the field build; uint32 fixed-point field;
        float64 cube."""
        lid, out = self.lidar, {}
        for name, lv in (("coarse", self.coarse), ("fine", self.fine)):
            if lv is None:
                continue
            f = int(2 * lv.reach / lv.step) + 1
            wm = int(2 * lv.reach / lid.unit)
            nbt = (lv.nx + 3) // 4
            out[name] = dict(scatter=wm * wm // 8 + f * f, blur=f * f + 4 * f * f,      # map bits in, occupied image out / in, field out
                             sweep=4 * f * f + 8 * lid.beams + 8 * lv.ntheta * lv.nx * lv.nx,
                             # branch and bound: tile bounds (gmin2, 1/16 of the field's cells, in; cell lists in; one
                             # double per pose tile out); surviving tiles + selection (field in, cell lists in, one
                             # partial per theta out)
                             bound=f * f // 4 + 4 * lid.beams * lv.ntheta + 8 * lv.ntheta * nbt * 4 * ((nbt + 3) // 4),
                             exact=4 * f * f + 4 * lid.beams * lv.ntheta + 24 * lv.ntheta, bnb=bool(lv.bnb))
        # update: touched cells of one representative scan (cell-major classification on the host LUT)
        rng = scen.ranges[0]
        S, B = lid.num_spokes, lid.beams
        beam = (lid.bin.astype(np.int64) - lid.spoke_start) % S
        own = beam < B
        rb = rng[np.where(own, beam, 0)]
        hw = lid.wall_thickness / 2
        empty = own & (rb < lid.max_range) & (lid.r < rb - hw)
        occ = own & (lid.r > rb - hw) & (lid.r < rb + hw)
        out["update"] = dict(touched_cells=int(empty.sum() + occ.sum()),
                             per_particle=8 * int(empty.sum() + occ.sum()), shared_lut=12 * lid.width ** 2)    # RMW of 4-byte cells; spoke table (4 + 8 B per window cell)
        return out


class ScenarioView:
    """The inputs of particles [p0, p1) of a Scenario (scan data shared)."""

    def __init__(self, scen, p0, p1):
        self.__dict__.update(scen.__dict__)
        self.P, self.est, self.uniform = p1 - p0, scen.est[:, p0:p1], scen.uniform[:, p0:p1]


class HotPathGroups:
    """The same step with the rank's particles in G groups, each on its own HIP stream.

    Every kernel of the step is latency-bound at 64 particles: a launch ramps up, works at a fraction of the
    machine and drains, and on ONE stream the next launch waits for the drain.  Particles are independent during a scan
    (Algorithm/FastSlam.py:25-27), so two half-size launch sequences on two streams fill each other's ramps and drains --
    measured 0.152 -> 0.137 ms per scan at 64 particles (tools/exp_streams.py; four groups are host-bound).  The one thing
    the groups share is the weight normaliser (Algorithm/FastSlam.py:30-48): every group's update launch leaves its
    [max, sum, sum of squares] (slam2d_grid_update_weights_local), a third stream waits for all groups' updates of the scan,
    (all-gathers the partials across ranks when sharded,) and folds them in fixed order (slam2d_weights_merge) -- the
    sharded normaliser with groups in the role of ranks.  A group's next update waits for that merge (it rewrites the
    log-weights the merge works on), so the groups never drift more than a scan apart; nothing else synchronises them."""

    def __init__(self, cfg, P, scen, device, G):
        assert P % G == 0 and G >= 2
        self.cfg, self.P, self.G = cfg, P, G
        per = P // G
        self.subs = [HotPath(cfg, per, ScenarioView(scen, g * per, (g + 1) * per), device) for g in range(G)]
        h0 = self.subs[0]
        self.E, self.L, self.lidar, self.coarse, self.fine, self.lazy = h0.E, h0.L, h0.lidar, h0.coarse, h0.fine, h0.lazy
        self.par = h0.par
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.sharded = dist.is_initialized() and (self.world > 1 or os.environ.get("SLAM2D_FORCE_DIST") == "1")
        self.total_particles = P * self.world
        self.d_logw = torch.zeros(P, dtype=torch.float64, device=device)
        self.d_w = torch.zeros(P, dtype=torch.float64, device=device)
        self.d_stats = torch.zeros(2, dtype=torch.float64, device=device)
        for g, sub in enumerate(self.subs):
            sub.d_logw, sub.d_w = self.d_logw[g * per:(g + 1) * per], self.d_w[g * per:(g + 1) * per]
        self.parts_local = torch.zeros(3 * G, dtype=torch.float64, device=device)
        self.part_of = [self.parts_local[3 * g:3 * g + 3] for g in range(G)]
        self.parts_all = torch.zeros(3 * G * self.world, dtype=torch.float64, device=device) if self.sharded else self.parts_local
        self.via_host = self.sharded and dist.get_backend() == "gloo"
        self.rccl = self.par.DirectRccl.create(device) if self.sharded and not self.via_host else None    # None: through torch.distributed
        pool = self.E.group_streams(device, G + 1)           # the same streams for every hot path of the process (see there)
        self.streams, self.norm = pool[1:], pool[0]
        self.handles = [C.c_void_p(st.cuda_stream) for st in self.streams]
        self.norm_handle = C.c_void_p(self.norm.cuda_stream)
        self.ev_done = [C.c_void_p(self.L.slam2d_event_create()) for _ in range(G)]       # events without timing, through the C ABI
        self.ev_merged = C.c_void_p(self.L.slam2d_event_create())
        self.merged_once = False
        self.prune = False
        # round 5: on one rank the groups' normaliser blocks merge among themselves on the device (Slam2dScan.d_norm_sync): no merge
        # launch, no event packet between a group's kernels (SLAM2D_BENCH_DEVICE_MERGE=0: the merge launch on a third stream, rounds 3-4)
        self.device_merge = os.environ.get("SLAM2D_BENCH_DEVICE_MERGE", "1") != "0"
        self.norm_sync = torch.zeros(64, dtype=torch.int32, device=device)
        # round 4: the whole step is issued by ONE library call (slam2d_groups_step); SLAM2D_BENCH_PYSTEP=1 keeps round 3's
        # call-by-call issue from Python (~7 us of interpreter + ctypes per launch: 0.103 of the 0.131 ms step)
        self.c_step = os.environ.get("SLAM2D_BENCH_PYSTEP", "0") != "1"
        _lib = self.E._lib
        self.cgroups = (_lib.Slam2dGroup * G)()
        self.cscan = _lib.Slam2dScan()
        self._maps_seen = [None] * G
        self._bases = []
        for g, sub in enumerate(self.subs):
            cg = self.cgroups[g]
            cg.coarse = C.pointer(sub.coarse.c)
            if sub.fine is not None:
                cg.fine = C.pointer(sub.fine.c)
            cg.P, cg.est_stride = per, 3
            cg.d_coarse, cg.d_fine = sub.m_coarse.data_ptr(), sub.m_fine.data_ptr()
            cg.d_flags = sub.eng.flags.data_ptr()
            cg.d_logw, cg.d_part = sub.d_logw.data_ptr(), self.part_of[g].data_ptr()
            cg.stream, cg.ev_done = self.handles[g], self.ev_done[g]
            self._bases.append((sub.d_est.data_ptr(), sub.d_est.stride(0) * 8, sub.d_psi.data_ptr(), sub.d_psi.stride(0) * 8,
                                sub.d_uniform.data_ptr(), sub.d_uniform.stride(0) * 8))
        sc = self.cscan
        sc.d_logw_all, sc.n_local, sc.n_parts = self.d_logw.data_ptr(), P, G * self.world
        sc.total_particles, sc.d_w, sc.d_stats = self.total_particles, self.d_w.data_ptr(), self.d_stats.data_ptr()
        sc.norm_stream, sc.ev_merged = self.norm_handle, self.ev_merged
        self._ranges_base, self._ranges_stride = h0.d_ranges.data_ptr(), h0.d_ranges.stride(0) * 8
        self._lidar_ref = C.byref(h0.eng.lidar_c)
        torch.cuda.synchronize()

    @property
    def groups(self):
        return self.subs

    def levels(self):
        return [lv for sub in self.subs for lv in sub.levels()]

    def maps(self):
        return [m for sub in self.subs for m in sub.eng.maps]

    def matches(self, which):
        torch.cuda.synchronize()
        return np.concatenate([sub.matches(which) for sub in self.subs])

    def weights(self):
        torch.cuda.synchronize()
        return self.d_w.cpu().numpy().copy()

    def take_flags(self):
        torch.cuda.synchronize()
        f = np.concatenate([sub.eng.take_flags() for sub in self.subs])
        torch.cuda.synchronize()                 # (the flags are cleared on torch's stream, the groups run on their own)
        return f

    def algorithmic_bytes(self, scen):
        return self.subs[0].algorithmic_bytes(scen)

    def step(self, s):
        if self.c_step:
            return self.step_c(s)
        return self.step_py(s)

    def step_c(self, s):
        """One scan of all groups through ONE library call (slam2d_groups_step); sharded: the all-gather of the partials and
        the merge follow from here."""
        E, L = self.E, self.L
        try:
            for g, sub in enumerate(self.subs):
                E._PINNED_STREAM = self.handles[g]           # (the little torch does here -- stamp reset, bit refresh -- goes to the group's stream)
                eng = sub.eng
                eng._before_update()
                eng.refresh_bits()
                for lv in sub.levels():
                    lv.next_generation()
                cg = self.cgroups[g]
                if self._maps_seen[g] != eng.maps_version:
                    cg.d_maps, self._maps_seen[g] = eng.d_maps.data_ptr(), eng.maps_version
                e0, es, p0, ps, u0, us = self._bases[g]
                cg.d_est, cg.d_psi_cs, cg.d_uniform = e0 + s * es, p0 + s * ps, u0 + s * us
        finally:
            E._PINNED_STREAM = None
        sc = self.cscan
        sc.d_ranges, sc.est_moving_dist = self._ranges_base + s * self._ranges_stride, float(self.subs[0].dist[s])
        sc.options = E._lib.MATCH_PRUNE_BY_PRIOR if self.prune else 0
        sc.d_parts, sc.n_parts, sc.total_particles = self.parts_all.data_ptr(), self.G * self.world, self.total_particles
        sc.wait_merged, sc.merge = int(self.merged_once), 0 if self.sharded else 1
        sc.d_norm_sync = self.norm_sync.data_ptr() if self.device_merge else None      # (sharded: the groups only wait / arrive; _gather_and_merge follows)
        for g in range(self.G):                              # (nobody waits for a group's update on a stream then: no event packet behind it)
            self.cgroups[g].ev_done = None if sc.d_norm_sync else self.ev_done[g]
        if os.environ.get("SLAM2D_BENCH_UNCOUPLED") == "1":       # timing experiment: the groups never meet (no normaliser)
            sc.wait_merged, sc.merge, sc.d_norm_sync = 0, 0, None
        E._lib.check(L.slam2d_groups_step(self._lidar_ref, self.cgroups, self.G, C.byref(sc)), "slam2d_groups_step")
        if self.sharded:
            self._gather_and_merge()
        self.merged_once = True

    def _gather_and_merge(self):
        E, L = self.E, self.L
        dsync = bool(self.cscan.d_norm_sync)
        if dsync:        # the normaliser's stream waits for the groups' partials on the device (no ev_done behind the groups' updates)
            E._lib.check(L.slam2d_norm_gate(C.c_void_p(self.norm_sync.data_ptr()), self.G, self.norm_handle), "slam2d_norm_gate")
        if self.rccl is not None:                        # ONE ncclAllGather on the normaliser's stream, straight from librccl
            self.rccl.all_gather(self.parts_local.data_ptr(), self.parts_all.data_ptr(), 3 * self.G, self.norm_handle)
        else:
            with torch.cuda.stream(self.norm):
                if self.via_host:                        # gloo (dry mode): the partials hop through host memory
                    mine = self.parts_local.cpu()
                    got = [torch.empty_like(mine) for _ in range(self.world)]
                    dist.all_gather(got, mine)
                    self.parts_all.copy_(torch.cat(got))
                else:
                    dist.all_gather_into_tensor(self.parts_all, self.parts_local)
        if dsync:        # ... and the groups' next normaliser blocks for the merge (no ev_merged)
            E._lib.check(L.slam2d_weights_merge_publish(E._ptr(self.d_logw), self.P, E._ptr(self.parts_all), self.G * self.world, self.total_particles,
                                                        E._ptr(self.d_w), E._ptr(self.d_stats), C.c_void_p(self.norm_sync.data_ptr()), self.norm_handle),
                         "slam2d_weights_merge_publish")
            return
        E._lib.check(L.slam2d_weights_merge(E._ptr(self.d_logw), self.P, E._ptr(self.parts_all), self.G * self.world,
                                            self.total_particles, E._ptr(self.d_w), E._ptr(self.d_stats), self.norm_handle),
                     "slam2d_weights_merge")
        L.slam2d_event_record(self.ev_merged, self.norm_handle)

    def step_py(self, s):
        E, L = self.E, self.L
        try:
            for g, sub in enumerate(self.subs):
                E._PINNED_STREAM = self.handles[g]           # every library call of this group goes to its stream (so does the
                sub.prune = self.prune                       # little torch does on the launch path: engine._on_launch_stream)
                final = sub.match(s)
                if self.merged_once:
                    L.slam2d_stream_wait_event(self.handles[g], self.ev_merged)   # the previous scan's merge is done with the log-weights
                sub.update_local(s, final, self.part_of[g])
                L.slam2d_event_record(self.ev_done[g], self.handles[g])
            for g in range(self.G):
                L.slam2d_stream_wait_event(self.norm_handle, self.ev_done[g])
            if self.sharded:
                with torch.cuda.stream(self.norm):
                    if self.via_host:                            # gloo (dry mode): the partials hop through host memory
                        mine = self.parts_local.cpu()
                        got = [torch.empty_like(mine) for _ in range(self.world)]
                        dist.all_gather(got, mine)
                        self.parts_all.copy_(torch.cat(got))
                    else:
                        dist.all_gather_into_tensor(self.parts_all, self.parts_local)
            # one merge launch for all groups: their log-weights are consecutive slices of one array
            E._lib.check(L.slam2d_weights_merge(E._ptr(self.d_logw), self.P, E._ptr(self.parts_all), self.G * self.world,
                                                self.total_particles, E._ptr(self.d_w), E._ptr(self.d_stats), self.norm_handle),
                         "slam2d_weights_merge")
            L.slam2d_event_record(self.ev_merged, self.norm_handle)
            self.merged_once = True
        finally:
            E._PINNED_STREAM = None


def bench_groups(arg, P, sharded=None):
    """Particle groups per GPU: --groups, else SLAM2D_BENCH_GROUPS, else what was measured best (round 5, 8 hardware queues, the
    groups' normaliser merged on the device): 4 groups for 32-96 particles on ONE rank whose grouped calls issue from worker
    threads (64 particles: 0.1132 ms per scan against 0.1178 in two groups, reference defaults 0.1852 against 0.1967); 2 groups
    from 16 particles up otherwise -- sharded (the all-gather's enqueue joins the host's work per scan: 0.1286 in four groups
    against 0.1211 in two on a one-rank RCCL group), without worker threads (a rank with fewer than 3 cores: four groups are
    host-bound, 0.1205 against 0.1177), at 128 particles and more (no difference / 5 % slower) -- else 1.  Eight groups collapse
    (0.59 ms): more streams than queues that run side by side."""
    if arg is None:
        env = os.environ.get("SLAM2D_BENCH_GROUPS", "")
        if sharded is None:
            sharded = (dist.is_initialized() and dist.get_world_size() > 1) or os.environ.get("SLAM2D_FORCE_DIST") == "1" or \
                int(os.environ.get("WORLD_SIZE", "1")) > 1
        four = 32 <= P <= 96 and P % 4 == 0 and not sharded
        if four:
            four = importlib.import_module("slam-2d-lidar-scan_amd._lib").group_policy()["threads"]
        arg = int(env) if env.isdigit() else (4 if four else 2 if P >= 16 and P % 2 == 0 else 1)
    return max(1, int(arg))


def make_hot_path(cfg, P, scen, device, groups):
    """groups <= 1: everything on the current stream (HotPath); else HotPathGroups."""
    if groups and groups > 1 and P % groups == 0 and P // groups >= 8:
        return HotPathGroups(cfg, P, scen, device, groups)
    return HotPath(cfg, P, scen, device)


_CPU = {}


def effective_cores():
    """Host cores this process may really use: os.cpu_count() capped by the scheduler affinity and the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return n


def _cpu_worker_init(cfg, visited, total, est, ranges, dist, psi):
    """One oracle matcher + grid per worker process (the reference builds one per particle, Algorithm/FastSlam.py:69-70)."""
    from oracle import slam_oracle as so
    u = cfg["unit"]
    lut = so.SpokeLUT(u, cfg["max_range"], cfg["fov"], cfg["beams"])
    og = so.GridOracle(cfg["map_m"], cfg["map_m"], {"x": 0.0, "y": 0.0}, u, cfg["fov"], cfg["beams"],
                       cfg["max_range"], cfg["wall"], lut=lut)
    og.visited[:], og.total[:] = visited, total
    sm = so.MatcherOracle(og, cfg["search_radius"], cfg["half_rad"], cfg["sigma_cells"], cfg["move_sigma"],
                          cfg["max_dev"], cfg["turn_sigma"], cfg["miss"], cfg["coarse_factor"],
                          rng=np.random.RandomState(0))
    _CPU.update(cfg=cfg, og=og, sm=sm, est=est, ranges=ranges, dist=dist, psi=psi)


def _cpu_unit(unit):
    """One particle-scan of the workload through the oracle: field build(s), cube(s), soft-max draw, map update."""
    cfg, og, sm = _CPU["cfg"], _CPU["og"], _CPU["sm"]
    n_scans, P = _CPU["est"].shape[:2]
    s, p = unit // P % n_scans, unit % P
    x, y, th = _CPU["est"][s, p]
    ranges, u = _CPU["ranges"][s], cfg["unit"]
    if cfg["levels"] == 2:
        matched, _ = sm.matchScan({"x": x, "y": y, "theta": th, "range": ranges}, _CPU["dist"][s], _CPU["psi"][s], 2, matchMax=False)
    else:
        xr, yr, prob = sm.frameSearchSpace(x, y, u, cfg["sigma_cells"], cfg["miss"])
        matched, _, _ = sm.searchToMatch(prob, x, y, th, ranges, xr, yr, cfg["search_radius"], cfg["half_rad"], u,
                                         _CPU["dist"][s], _CPU["psi"][s], fineSearch=False, matchMax=False)
    og.update_cell_major(matched)
    return 1


def cpu_baseline(cfg, scen, target_seconds, workers):
    """The CPU oracle (a NumPy port of the reference; oracle/slam_oracle.py) on the same workload: (i) one process,
    as the reference runs; (ii) a process pool over particle-scans on every host core (particles are independent,
    Algorithm/FastSlam.py:25-27).  Bounded samples.  Runs BEFORE the GPU is initialised (the pool forks)."""
    import multiprocessing as mp
    args = (cfg, scen.visited, scen.total, scen.est, scen.ranges, scen.dist, scen.psi)
    n_scans, P = scen.est.shape[:2]
    t0 = time.perf_counter()
    _cpu_worker_init(*args)
    setup = time.perf_counter() - t0
    units, t_start = 0, time.perf_counter()
    while True:
        _cpu_unit(units)
        units += 1
        el = time.perf_counter() - t_start
        if el >= target_seconds or units >= 4 * P:
            break
    single = units / el
    out = dict(value=single, unit="particle-scans/s", cores=1, kind="port",
               sample=f"{units} particle-scans of the same workload ({el:.1f} s; NumPy port of the reference, one process; "
                      f"LUT / setup {setup:.1f} s excluded)", host_cores_available=os.cpu_count(), host_cores_effective=effective_cores())
    workers = workers or effective_cores()
    if workers > 1:
        try:
            ctx = mp.get_context("fork")
            with ctx.Pool(workers, initializer=_cpu_worker_init, initargs=args) as pool:
                pool.map(_cpu_unit, range(workers), chunksize=1)                   # workers warm (their matchers built)
                n, t1, rates = 0, time.perf_counter(), []
                while True:                                                         # rounds of one unit per worker, time-bounded
                    ta = time.perf_counter()
                    pool.map(_cpu_unit, range(n, n + workers), chunksize=1)
                    n += workers
                    tb = time.perf_counter()
                    rates.append(workers / (tb - ta))
                    el2 = tb - t1
                    if el2 >= target_seconds:
                        break
            out.update(value=n / el2, cores=workers, single_core_value=single,
                       spread={"rounds": len(rates), "min": min(rates), "median": statistics.median(rates), "max": max(rates),
                               "note": "particle-scans/s of every round of one unit per worker; the figure moves with the box's other tenants"},
                       sample=f"{n} particle-scans on a {workers}-process pool ({el2:.1f} s); single process: {units} in {el:.1f} s; "
                              "NumPy port of the reference; LUT / setup excluded")
        except Exception as exc:                                                    # the single-core leg stands
            out["pool_error"] = repr(exc)
    return out


TA_CLK_PER_WAVE_LOAD = 16.0     # measured: a vector-memory instruction occupies a CU's texture path for 16 clocks whatever its width
GPU_CLOCK_GHZ = 2.4
N_CU = 256


def _ctl(values, device):
    """A small tensor for the control collectives (barrier timing, rank census): on the GPU for nccl, on the host for gloo."""
    on_host = dist.is_initialized() and dist.get_backend() == "gloo"
    return torch.tensor(values, dtype=torch.float64, device="cpu" if on_host else device)


def timed_run(hot, first, K):
    """K steps starting at scan `first`, bracketed by barrier + synchronize on both sides; returns (seconds = MAX over
    ranks, this rank's own seconds)."""
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(first, first + K):
        hot.step(s)
    timed_run.host_enqueue_s = time.perf_counter() - t0      # how long the host took to enqueue the block (host-bound if ~ the block)
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    if dist.is_initialized():
        dist.barrier()
    el = time.perf_counter() - t0
    if dist.is_initialized():
        t = _ctl([el], hot.d_logw.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el, mine


def level_stats(hot):
    """What the last step really touched, per level (device state read back once, outside every timed region): tiles
    blurred / filled / needed per particle, pose tiles scored exactly, mean unique endpoint cells per angle, occupied
    field cells -- averaged over the particles of all groups."""
    torch.cuda.synchronize()
    out = {}
    for name in ("coarse", "fine"):
        acc, wsum = {}, 0
        for sub in hot.groups:
            lv = getattr(sub, name)
            if lv is None:
                continue
            st = {a: float(b) for a, b in lv.bnb_stats().items()}
            need = lv.t["tileneed"].cpu().numpy().view(np.uint32)              # [P, groups of angles, words]
            need = np.bitwise_or.reduce(need, axis=1)
            st["needed_tiles"] = float(np.unpackbits(need.view(np.uint8), axis=1).sum(axis=1).mean())
            st["kbar"] = float(lv.t["kcount"].double().mean().item())
            P, f = lv.P, lv.fmax * lv.fpitch
            occ = lv.t["occ"][:P * f].view(P, f)
            st["occupied_field_cells"] = float((occ == lv.c.occ_gen).sum(dim=1).double().mean().item())
            for a, v in st.items():
                acc[a] = (max(acc.get(a, 0.0), v) if a == "kept_max_per_theta" else acc.get(a, 0.0) + v * lv.P)
            wsum += lv.P
        if wsum:
            out[name] = {a: (v if a == "kept_max_per_theta" else v / wsum) for a, v in acc.items()}
    return out


def processed_bytes(hot, scen, stats):
    """Bytes one particle-scan REALLY processes, stage by stage -- SURVEY.md 8(d)'s convention (every array once per
    direction, real storage) applied to the parts of the arrays the lazy build and the branch and bound touch, not to the
    whole arrays: tiles built x bytes per tile, surviving pose tiles x K x 64 B, touched cells x 8 B."""
    lid, out = hot.lidar, {}
    for name, lv in (("coarse", hot.coarse), ("fine", hot.fine)):
        if lv is None:
            continue
        st = stats[name]
        wm = int(2 * lv.reach / lid.unit)
        nbt = (lv.nx + 3) // 4
        nneed = (lv.tmax * lv.tmax + 31) // 32
        groups = -(-lv.ntheta // lv.ep_group)
        d = dict(
            endpoints=8 * lid.beams + (8 if lv.bnb else 4) * lv.ntheta * st["kbar"] + 4 * groups * nneed,     # ranges in; cell lists + tile slices out
            scatter=wm * wm // 8 + 2 * st["occupied_field_cells"],                                              # map bits in; stamped bytes + tile flags out
            triage=2 * lv.tmax * lv.tmax + 4 * groups * nneed + 4 * (st["blur_tiles"] + st["fill_tiles"]),
            blur=256 * (1 + 4) * st["blur_tiles"] + 256 * 4 * st["fill_tiles"],                                 # occupied image in, field out
            check=(2 * 16 * 4 + (16 if "gmin2b" in lv.t else 0)) * (st["blur_tiles"] + st["fill_tiles"]) if lv.bnb else 0.0)   # block minima in, gmin2 (+ its byte image) out
        if lv.bnb:
            # tile bounds: the bound entries of the needed tiles in (k_bound_lds: the particle's whole byte image, staged in LDS once),
            # cell lists in, one double per pose tile out
            staged = 4 * lv.tmax * lv.g2b_pitch if "gmin2b" in lv.t and os.environ.get("SLAM2D_BOUND_LDS", "1") != "0" else 16 * 4 * st["needed_tiles"]
            d["bound"] = staged + 4 * lv.ntheta * st["kbar"] + 8 * lv.ntheta * nbt * 4 * ((nbt + 3) // 4)
            d["exact"] = 64 * st["kbar"] * st.get("kept_per_particle", 0.0) + 4 * lv.ntheta * st["kbar"] + 8 * 16 * st.get("kept_per_particle", 0.0)
        else:
            d["sweep"] = 256 * 4 * st["needed_tiles"] + 4 * lv.ntheta * st["kbar"] + 8 * lv.ntheta * lv.nx * lv.nx
        d["bnb"] = bool(lv.bnb)
        out[name] = d
    ab = hot.algorithmic_bytes(scen)["update"]
    out["update"] = dict(per_particle=ab["per_particle"], table_per_launch=12 * ab["touched_cells"], touched_cells=ab["touched_cells"])
    return out


STAGE_OF_KERNEL = {"k_sweep": "sweep", "k_blur_clamp": "blur", "k_occ_scatter": "scatter", "k_bound": "bound", "k_exact": "exact",
                   "k_exact_select": "exact", "k_endpoints": "endpoints", "k_tile_triage": "triage", "k_blur_check_redo": "check"}
# the kernels of one step (one scan of all particles), in launch order
STEP_KERNELS = ("k_endpoints", "k_tile_triage", "k_blur_clamp", "k_blur_check_redo", "k_bound", "k_exact_select", "k_sweep", "k_grid_update")
ROOFLINE_TARGET_NOTE = ("north_star asks for >= 40 % of the HBM roofline.  That figure is reachable only by SURVEY 8(d)'s whole-array bytes "
                        "(roofline.whole_step.whole_array_frac: every field, window and cube counted in full) -- bytes the lazy field build and "
                        "the branch and bound deliberately never touch (the reference scores every pose, Utils/ScanMatcher_OGBased.py:116-132; "
                        "this build scores ~1 %).  By the bytes really processed (roofline.frac) and by counters (roofline.measured_hbm_frac) the "
                        "step runs at 0.07-0.11 of peak: at 64 particles x 180 beams it is a chain of latency-bound launches, not a bandwidth "
                        "problem, and the 40 % target is not applicable as phrased")


def stage_bytes_per_launch(kernel, pb, P, launches_per_step, merged_scatter=True):
    """Processed bytes of ONE launch of `kernel`: the step's bytes of the levels it serves (all particles) divided over its
    launches of a step -- every kernel alike, k_grid_update included (one launch per particle group: its spoke table is read
    once per launch)."""
    lps = max(launches_per_step, 1e-9)
    if kernel == "k_grid_update":
        return pb["update"]["per_particle"] * P / lps + pb["update"]["table_per_launch"]
    key = STAGE_OF_KERNEL.get(kernel)
    if key is None:
        return 0.0
    tot = sum(v.get(key, 0.0) for k, v in pb.items() if k != "update")
    if kernel == "k_endpoints" and merged_scatter:              # the occupied-cell scatter rides in this launch
        tot += sum(v.get("scatter", 0.0) for k, v in pb.items() if k != "update")
    return tot * P / lps


def step_processed_bytes(pb):
    lev = [v for k, v in pb.items() if k != "update"]
    return sum(sum(x for kk, x in v.items() if kk != "bnb") for v in lev) + pb["update"]["per_particle"]


def load_traffic(workload):
    """Per-launch HBM bytes of every kernel from the PMC passes (profiles/traffic.json, written by
    tools/summarize_profiles.py from rocprofv3 --pmc runs of this very command).  Refused when the kernels have changed
    since: the file records the SHA-256 of slam2d.hip it was measured on."""
    path = os.path.join(REPO, "profiles", "traffic.json")
    if not os.path.exists(path):
        return {}, "profiles/traffic.json is missing"
    doc = json.load(open(path))
    if doc.get("source_sha256") != source_sha256():
        return {}, "profiles/traffic.json was measured on another build of slam2d.hip (source hash differs): refused"
    pre = workload + ":"
    return {k[len(pre):]: v for k, v in doc.get("entries", {}).items() if k.startswith(pre)}, None


def roofline_of(hot, scen, P, ms_per_step, stage_ms, launches_per_step_of, workload, overhead_us=0.0, probe_ms=None, pmc_input=True):
    """The roofline block of the JSON line.  Headline (`frac`, `achieved`, `traffic`): the WHOLE STEP against the HBM peak by the
    bytes it really processes, with the PMC-measured traffic beside it -- one stable figure instead of whichever kernel's summed
    event time happens to lead (k_bound and k_exact_select are within a few per cent of one another and 6x apart in bytes).
    `kernels`: one row per kernel of the step (time, processed bytes, counter bytes, their ratio); `dominant`: the kernel with the
    largest summed event time, timed live inside the timed region.  pmc_input: the scans are the ones profiles/traffic.json was
    measured on (the workload's tracked input); other inputs get no counter figures."""
    stats = level_stats(hot)
    pb = processed_bytes(hot, scen, stats)
    traffic_all, traffic_note = load_traffic(workload) if pmc_input else ({}, "no PMC passes for this input: counter figures are "
                                                                             "given only for the scans they were measured on")
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "whole step"}
    ngroups = len(hot.groups)
    lps_default = {"k_grid_update": float(ngroups)}
    per_level = launches_per_step_of.get("k_blur_clamp", float(ngroups * sum(1 for lv in (hot.coarse, hot.fine) if lv is not None)))      # one launch per level and group

    def lps_of(kernel):
        stage = {"k_exact_select": "k_exact"}.get(kernel, kernel)
        if stage in launches_per_step_of:
            return launches_per_step_of[stage]
        return lps_default.get(kernel, float(per_level))
    if stage_ms:
        dom = max(stage_ms, key=lambda k: stage_ms[k]["total_ms"])
        lps = launches_per_step_of.get(dom, 1.0)
        raw_us = stage_ms[dom]["avg_us"]
        t_us = max(raw_us - overhead_us, 1e-3)
        proc = stage_bytes_per_launch(dom, pb, P, lps)
        ab = hot.algorithmic_bytes(scen)
        lev = {k: v for k, v in ab.items() if k != "update"}
        whole_array = {"k_sweep": sum(v["sweep"] for v in lev.values() if not v["bnb"]), "k_blur_clamp": sum(v["blur"] for v in lev.values()),
                       "k_occ_scatter": sum(v["scatter"] for v in lev.values()), "k_bound": sum(v["bound"] for v in lev.values() if v["bnb"]),
                       "k_exact": sum(v["exact"] for v in lev.values() if v["bnb"]), "k_grid_update": ab["update"]["per_particle"]}.get(dom, 0) * P / lps
        name = {"k_exact": "k_exact_select"}.get(dom, dom)
        tr = (traffic_all.get(name) or {}).get("hbm_bytes_corrected")
        proc_at = (traffic_all.get(name) or {}).get("processed_bytes_at_measurement")      # of the scans the counters saw
        achieved = proc / (t_us * 1e-6) / 1e9
        dominant = dict(kernel=name, achieved=achieved, frac=achieved / HBM_PEAK_GBS, traffic=tr,
                        measured_hbm_frac=(tr / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if tr else None,
                        wasted=(tr / (proc_at or proc)) if tr and (proc_at or proc) else None,
                        processed_bytes_per_launch=proc, avg_launch_us=t_us, avg_launch_us_with_event_pair=raw_us,
                        event_pair_overhead_us=overhead_us,
                        note="the kernel with the largest summed event time in the timed region: HIP events on its launch stream around "
                             "every n-th launch, minus the measured cost of an empty event pair",
                        whole_array_convention={"bytes_per_launch": whole_array, "frac": whole_array / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                "note": "SURVEY 8(d) whole-array bytes / the time of a kernel that touches a fraction of those arrays: "
                                                        "evidence of work avoided, NOT a bandwidth figure"})
        cl = hot.coarse
        kbar = stats["coarse"]["kbar"]
        gathers = {"k_sweep": cl.ntheta * kbar * math.ceil(cl.nx * math.ceil(cl.nx / 4) / 64), "k_bound": cl.ntheta * kbar,
                   "k_exact": stats["coarse"].get("kept_per_particle", 0.0) * math.ceil(kbar / 16)}
        if dom in gathers:
            ta_us = gathers[dom] * P * TA_CLK_PER_WAVE_LOAD / N_CU / (GPU_CLOCK_GHZ * 1e3)
            dominant["gather"] = dict(wave_loads_per_launch=gathers[dom] * P, clk_per_wave_load=TA_CLK_PER_WAVE_LOAD, texture_path_bound_us=ta_us,
                                      frac_of_texture_path_bound=ta_us / t_us)
        out["dominant"] = dominant
        out["event_pairs"] = {"launches_timed": stage_ms[dom]["launches"], "region": "inside the timed steps, on the launch stream", "kernel": name}
    # one row per kernel of the step: time (the probe's HIP events where the stage has them, else the rocprofv3 average recorded with the
    # counters), processed bytes of one launch, counter bytes of one launch, their ratio
    rows = []
    for kname in STEP_KERNELS:
        stage = {"k_exact_select": "k_exact"}.get(kname, kname)
        ent = traffic_all.get(kname) or {}
        lps = lps_of(kname)
        proc = stage_bytes_per_launch(kname, pb, P, lps)
        if proc <= 0.0 and not ent:
            continue
        t_us, t_src = None, None
        if probe_ms and stage in probe_ms:
            t_us, t_src = max(probe_ms[stage]["avg_us"] - overhead_us, 1e-3), "HIP events (probe steps)"
        elif ent.get("avg_us_rocprof"):
            t_us, t_src = ent["avg_us_rocprof"], "rocprofv3 --kernel-trace (profiles/)"
        hbm = ent.get("hbm_bytes_corrected")
        rows.append({"kernel": kname, "launches_per_step": lps, "avg_us": round(t_us, 2) if t_us else None, "time_from": t_src,
                     "processed_MB": round(proc / 1e6, 3), "frac": round(proc / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if t_us else None,
                     "hbm_MB": round(hbm / 1e6, 3) if hbm else None,
                     "hbm_over_processed": round(hbm / ent["processed_bytes_at_measurement"], 2) if hbm and ent.get("processed_bytes_at_measurement") else None})
    out["kernels"] = rows
    # the whole step: processed bytes, whole-array bytes and -- when every kernel of the step has a PMC entry -- measured HBM bytes
    step_proc = step_processed_bytes(pb)
    ab = hot.algorithmic_bytes(scen)
    lev = {k: v for k, v in ab.items() if k != "update"}
    step_whole = sum(v["scatter"] + v["blur"] + (v["bound"] + v["exact"] if v["bnb"] else v["sweep"]) for v in lev.values()) + ab["update"]["per_particle"]
    t = ms_per_step * 1e-3
    meas = None
    if traffic_all:
        # (the PMC summary counts launches per k_grid_update launch, i.e. per scan of ONE particle group)
        meas = sum(v.get("hbm_bytes_corrected", 0.0) * v.get("launches_per_step", 1.0) for k, v in traffic_all.items() if v.get("in_step")) \
            * launches_per_step_of.get("k_grid_update", ngroups)
    achieved = step_proc * P / t / 1e9
    out.update(achieved=achieved, frac=achieved / HBM_PEAK_GBS, traffic=meas,
               measured_hbm_frac=(meas / t / 1e9 / HBM_PEAK_GBS) if meas else None,
               wasted=(meas / (step_proc * P)) if meas else None,
               traffic_note=traffic_note or "counter bytes replayed from profiles/traffic.json (builder-side rocprofv3 --pmc passes of this command, "
                                            "2 FETCH_SIZE + WRITE_SIZE per the gfx950 guide, locked to slam2d.hip's SHA-256), divided by THIS run's "
                                            "step time: not an independent measurement of this run",
               target_note=ROOFLINE_TARGET_NOTE)
    out["whole_step"] = {"processed_bytes_per_particle_scan": step_proc, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                         "measured_hbm_bytes_per_step": meas, "measured_hbm_frac": (meas / t / 1e9 / HBM_PEAK_GBS) if meas else None,
                         "whole_array_bytes_per_particle_scan": step_whole, "whole_array_frac": step_whole * P / t / 1e9 / HBM_PEAK_GBS}
    out["processed_bytes_per_particle_scan"] = {k: ({a: (round(b, 1) if not isinstance(b, bool) else b) for a, b in v.items()}) for k, v in pb.items()}
    out["tile_stats"] = {k: {a: round(b, 3) for a, b in v.items()} for k, v in stats.items()}
    return out


def side_workload(name, P, K, W, device, rank, mode="tracked", with_brute=False, groups=None):
    """A short run of another workload / another input for the `variants` block (same launch sequence, its own scenario).
    with_brute: the same scans again with every pose scored (k_sweep) -- what the branch and bound must not lose against."""
    cfg = WORKLOADS[name]
    scen = Scenario(cfg, P, K + W, seed=0, rank=rank, mode=mode)
    hot = make_hot_path(cfg, P, scen, device, bench_groups(groups, P))

    def run():
        for s in range(W):
            hot.step(s)
        hot.take_flags()
        blocks = []
        for _ in range(3):
            blocks.append(timed_run(hot, W, K)[0])
            hot.take_flags()
        return statistics.median(blocks)
    el = run()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rf = roofline_of(hot, scen, P, 1e3 * el / K, {}, {}, name, pmc_input=(mode == "tracked"))
    out = dict(value=P * world * K / el, unit="particle-scans/s", ms_per_step=1e3 * el / K, steps=K, repeats=3, particles_per_gpu=P,
               workload=cfg["note"], input=mode, pose_hypotheses_per_particle_scan=hot.coarse.ntheta * hot.coarse.nx ** 2 +
               (hot.fine.ntheta * hot.fine.nx ** 2 if hot.fine else 0),
               branch_and_bound={k: bool(lv.bnb) for k, lv in (("coarse", hot.coarse), ("fine", hot.fine)) if lv is not None},
               particle_groups_per_gpu=len(hot.groups), whole_step=rf["whole_step"], tile_stats=rf["tile_stats"])
    if with_brute and any(lv.bnb for lv in hot.levels()):
        saved = [(lv, lv.c.bnb) for lv in hot.levels()]
        for lv, _ in saved:
            lv.c.bnb = 0
        elb = run()
        for lv, v in saved:
            lv.c.bnb = v
        out["brute_force_ms_per_step"] = 1e3 * elb / K
        out["vs_brute_force"] = elb / el               # > 1: the default path is faster than scoring every pose
    return out


def p_sweep(name, counts, K, W, device, rank):
    """particle-scans/s of one workload against the particles per GPU (default groups): does P = 64 fill the machine?"""
    out = {}
    for P in counts:
        r = side_workload(name, P, K, W, device, rank)
        out[str(P)] = dict(value=r["value"], ms_per_step=r["ms_per_step"], us_per_particle_scan=1e3 * r["ms_per_step"] / P,
                           particle_groups_per_gpu=r["particle_groups_per_gpu"])
        torch.cuda.empty_cache()
    return out


def config3_closed_loop(P, device):
    """BASELINE config 3: FastSLAM over the bundled Intel log (910 scans x 180 beams), reference defaults, 50 m map with
    growth, CLOSED loop through ParticleFilter (host decides growth / resampling, one synchronisation per scan; scans
    are staged from host memory) -- the host- and PCIe-inclusive rate, never `value`."""
    pkg = importlib.import_module("slam-2d-lidar-scan_amd")
    dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
    path = os.path.join(REPO, "tests", "golden", "intel_gfs.npz")
    if not os.path.exists(path):
        return None
    readings = dataio.read_npz(path)
    u = 0.02
    ogP = [50.0, 50.0, readings[0], u, math.pi, 10, 180, 5 * u]
    smP = [1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5]
    pf = pkg.ParticleFilter(P, ogP, smP, device=device, rng=np.random.RandomState(0))
    for count, raw in enumerate(readings[:5], start=1):          # warm-up: first scans
        pf.updateParticles(raw, count)
        pf.weightUnbalanced()
    def leg(force=(), groups=None, P=P):
        # (the previous leg's filter sits in a reference cycle -- its particle views point back at it -- so its 1.6 GB of maps, its
        # streams and events live until Python's cyclic collector happens to run: inside this timed leg, or not at all -- then this
        # leg's growth re-allocations miss torch's cache.  The "slow legs" of round 5, 0.28 s against 0.20 s, were the ones with
        # 67-68 device allocations.  Collected here, every timed leg runs alike; the cold figure is the first leg's)
        import gc
        gc.collect()
        pf = pkg.ParticleFilter(P, ogP, smP, device=device, rng=np.random.RandomState(0), groups=groups)
        torch.cuda.synchronize()
        ms0 = torch.cuda.memory_stats(device)
        t0 = time.perf_counter()
        resamples = pf.run(readings, force_resample=set(force))      # the pipelined driver: same decisions as the per-call loop
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms1 = torch.cuda.memory_stats(device)
        # (hipMalloc / hipFree calls inside the leg: a growth re-allocation the caching allocator cannot serve costs ~0.5 ms)
        dev_allocs = [int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("num_device_alloc", "num_device_free")]
        m = pf.engine.maps[int(np.argmax(pf.weights))]
        return dict(value=P * len(readings) / el, unit="particle-scans/s", scans=len(readings), particles=P, seconds=el,
                    scans_per_sec=len(readings) / el, resamples=len(resamples), state_moving_resamples=pf.stats["state_moving_resamples"],
                    scans_voided_and_repeated=pf.stats.get("aborted", 0), scans_redone=pf.stats["redo"], scans_reissued_in_pipeline=pf.stats["reissued"],
                    scans_step_by_step=pf.stats["step_by_step"], particle_groups=pf.n_groups, device_allocs_and_frees=dev_allocs,
                    final_map=[m.rows, m.cols])
    cold = leg()                                       # (first leg: allocator warm-up, 1.6 GB of maps)
    legs = sorted((leg() for _ in range(3)), key=lambda r: r["seconds"])      # three timed legs, the median reported: a leg holds 143
    out = legs[1]                                                              # growth re-allocations and the host's per-scan work, and
    out["seconds_of_each_leg"] = [round(r["seconds"], 5) for r in legs]        # boxes of the pool differ by 25 % on it
    out["device_allocs_and_frees_of_each_leg"] = [r["device_allocs_and_frees"] for r in legs]
    out["first_leg_cold_allocator"] = {k: cold[k] for k in ("seconds", "scans_per_sec", "device_allocs_and_frees")}     # every growth re-allocation a hipMalloc
    out["note"] = ("closed loop through ParticleFilter.run(): host decisions (growth, resampling); the particles in groups on their own streams "
                   "(ParticleFilter.auto_groups), each scan's ranges pulled from pinned host memory and its report pushed there by the device (no copy, "
                   "no event); scan s is enqueued before scan s-1's results are read; a scan voided on the device (a search window left its map) is "
                   "issued again through the pipeline once the maps have grown (slam2d_map_grow: one device pass per map)")
    # the same in one group (through the same event-free calls) and through round 4's one-stream calls (staging copy, download, event)
    out["one_group"] = {k: v for k, v in leg(groups=1).items() if k in ("value", "seconds", "scans_per_sec", "particle_groups")}
    os.environ["SLAM2D_FILTER_GROUPED1"] = "0"
    try:
        out["one_stream_calls"] = {k: v for k, v in leg(groups=1).items() if k in ("value", "seconds", "scans_per_sec", "particle_groups")}
    finally:
        del os.environ["SLAM2D_FILTER_GROUPED1"]
    out["one_stream_calls"]["note"] = "slam2d_scan_match / slam2d_scan_commit_next on one stream: rounds 3-4's closed loop"
    # the same log with a resample forced every 100 scans (64 particles never degenerate by themselves on this log): the gather
    # of 64 maps, the stop of the group streams and the redone speculative scan are inside the timed figure
    forced = leg(force=range(100, len(readings), 100))
    out["forced_resample_every_100"] = {k: v for k, v in forced.items() if k in ("value", "seconds", "scans_per_sec", "resamples",
                                                                                "state_moving_resamples", "scans_redone", "device_allocs_and_frees")}
    # the closed loop at other particle counts (one leg each after a warm-up leg: 16 / 256 particles x 910 scans)
    out["p_sweep"] = {}
    for PP in (16, 256):
        leg(P=PP)
        r = leg(P=PP)
        out["p_sweep"][str(PP)] = {k: r[k] for k in ("value", "seconds", "scans_per_sec", "scans_voided_and_repeated")}
    return out


class _SerialParticle:
    """The reference's Particle (Algorithm/FastSlam.py:64-140) restated as a CALLER of the drop-in classes: one OccupancyGrid and
    one ScanMatcher per particle, one synchronous matchScan + one updateOccupancyGrid per scan -- the path a user takes who
    swaps only Utils/OccupancyGrid.py and Utils/ScanMatcher_OGBased.py under the unchanged driver."""

    def __init__(self, pkg, ogP, smP):
        mapX, mapY, init_xy, unit, fov, max_range, beams, wall = ogP
        self.og = pkg.OccupancyGrid(mapX, mapY, init_xy, unit, fov, beams, max_range, wall)
        self.sm = pkg.ScanMatcher(self.og, *smP)
        self.weight, self.traj = 1.0, []
        self.prev_matched = self.prev_raw = self.prev_raw_heading = self.prev_heading = None

    @staticmethod
    def _heading(dx, dy):
        d = math.sqrt(dx * dx + dy * dy)
        return (math.acos(dx / d) if dy > 0 else -math.acos(dx / d)) if d != 0 else None

    def update(self, reading, count):
        if count == 1:
            self.prev_raw_heading = self.prev_heading = None
            matched, conf = reading, 1
        else:
            pr, pm = self.prev_raw, self.prev_matched
            est = {"x": pm["x"], "y": pm["y"], "theta": pm["theta"] + reading["theta"] - pr["theta"], "range": reading["range"]}
            dx, dy = reading["x"] - pr["x"], reading["y"] - pr["y"]
            dist = math.sqrt(dx * dx + dy * dy)
            raw_heading = est_heading = None
            if dist > 0.3:
                raw_heading = self._heading(dx, dy)
                if self.prev_raw_heading is not None:
                    est_heading = self.prev_heading + (raw_heading - self.prev_raw_heading)
            matched, conf = self.sm.matchScan(est, dist, est_heading, count, matchMax=False)
            self.prev_raw_heading = raw_heading
            self.prev_heading = self._heading(matched["x"] - self.traj[-1][0], matched["y"] - self.traj[-1][1])
        self.traj.append((matched["x"], matched["y"]))
        self.og.updateOccupancyGrid(matched)
        self.prev_matched, self.prev_raw = matched, reading
        self.weight *= conf


def dropin_serial(P, n_scans, device):
    """The UNCHANGED-CALLER path (north_star: "drops in under FastSlam.py unchanged"): the reference's serial loop over P
    particles (Algorithm/FastSlam.py:25-27) on the drop-in classes, Intel log, reference defaults -- host- and PCIe-inclusive
    (one upload and one download per matchScan), never `value`."""
    pkg = importlib.import_module("slam-2d-lidar-scan_amd")
    dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
    path = os.path.join(REPO, "tests", "golden", "intel_gfs.npz")
    if not os.path.exists(path):
        return None
    readings = dataio.read_npz(path)[:n_scans]
    u = 0.02
    ogP = [50.0, 50.0, readings[0], u, math.pi, 10, 180, 5 * u]
    smP = [1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5]
    np.random.seed(0)
    parts = [_SerialParticle(pkg, ogP, smP) for _ in range(P)]
    for count, rd in enumerate(readings[:3], start=1):                    # warm-up (first builds, allocations)
        for pt in parts:
            pt.update(rd, count)
    torch.cuda.synchronize()
    t_match = t_upd = t_upd_host = 0.0
    t0 = time.perf_counter()
    for count, rd in enumerate(readings[3:], start=4):
        for pt in parts:
            pt.update(rd, count)
        w = np.array([pt.weight for pt in parts])
        s = w.sum()
        for pt in parts:
            pt.weight /= s                                                # normalizeWeights (:43-48); no resample in 50 scans
    for pt in parts:
        pt.og.flush()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n = len(readings) - 3
    # the two calls timed apart on one particle (synchronised around each)
    pt = parts[0]
    for count, rd in enumerate(readings[-8:], start=len(readings) + 1):
        pm, pr = pt.prev_matched, pt.prev_raw
        est = {"x": pm["x"], "y": pm["y"], "theta": pm["theta"] + rd["theta"] - pr["theta"], "range": rd["range"]}
        torch.cuda.synchronize(); a = time.perf_counter()
        matched, _ = pt.sm.matchScan(est, 0.1, None, count, matchMax=False)
        b = time.perf_counter()
        pt.og.updateOccupancyGrid(matched)
        b2 = time.perf_counter()
        pt.og.flush(); torch.cuda.synchronize()
        c = time.perf_counter()
        t_match += b - a; t_upd += c - b; t_upd_host += b2 - b
        pt.prev_matched = matched
    return dict(value=P * n / el, unit="particle-scans/s", scans=n, particles=P, seconds=el, scans_per_sec=n / el,
                ms_per_matchScan=1e3 * t_match / 8, ms_per_updateOccupancyGrid=1e3 * t_upd / 8,
                ms_per_updateOccupancyGrid_host=1e3 * t_upd_host / 8,      # the call alone (it does not wait for its launch); the figure before it includes the kernel and a synchronisation
                note="the reference's serial per-particle loop (Algorithm/FastSlam.py:25-27,122-135) on the drop-in OccupancyGrid / "
                     "ScanMatcher classes: one synchronous matchScan (one upload, one download) and one updateOccupancyGrid per "
                     "particle and scan; host- and PCIe-inclusive")


def par_mod():
    return importlib.import_module("slam-2d-lidar-scan_amd.parallel")


def normaliser_probe(workload, P, K, W, G, base_ms):
    """What SHARDING adds to a rank's step, measured on this one GPU: the same command in fresh processes with a one-rank RCCL
    group (SLAM2D_FORCE_DIST=1: the sharded code path -- the groups' normaliser blocks only count themselves in, a gate kernel on the
    normaliser's stream, the all-gather, the publishing merge) and, for the difference, unsharded in the sharded run's group count.
    Fresh processes because which hardware queue a stream lands on depends on the streams its process created before (the same
    legs INSIDE this process, behind the four-group headline, put the collective's stream on a group's queue: 0.25 ms per scan
    against 0.121) -- and a rank of an N-GPU job is a fresh process.  The data path has no other collective
    (Algorithm/FastSlam.py:25-27), so a rank of an N-GPU job runs this step plus the all-gather's latency over xGMI: a PREDICTION for
    the first multi-GPU run to be checked against, not a measurement of it."""
    def leg(env_extra):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "SLAM2D_BENCH_GROUPS")}
        env.update(env_extra)
        env["MASTER_PORT"] = str(_free_port())
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", workload, "--particles", str(P), "--steps", str(K),
               "--warmup", str(W), "--repeats", "5", "--no-variants", "--no-cpu-baseline"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if res.returncode != 0 or not lines:
            raise RuntimeError(f"probe leg {env_extra} failed: {res.stderr[-300:]}")
        d = json.loads(lines[-1])
        return d["timed_blocks"]["ms_per_step_of_each"], d["config"]["particle_groups_per_gpu"]
    try:
        g2 = str(bench_groups(None, P, sharded=True))
        base_blocks, _ = leg({"SLAM2D_BENCH_GROUPS": g2})
        c10d_blocks, groups = leg({"SLAM2D_FORCE_DIST": "1"})
        try:
            direct_blocks, _ = leg({"SLAM2D_FORCE_DIST": "1", "SLAM2D_DIRECT_RCCL": "1"})
        except Exception:
            direct_blocks = None
        base_here, ms = statistics.median(base_blocks), statistics.median(c10d_blocks)
        # the difference of two medians of five blocks each, WITH its sign, and the blocks' own spread beside it
        added = ms - base_here
        spread = max(max(c10d_blocks) - min(c10d_blocks), max(base_blocks) - min(base_blocks))
        # a rank of an N-GPU job runs the sharded step; the one-GPU run it is compared with runs the headline's step
        eff_lo, eff_hi = min(1.0, base_ms / (ms + spread)), min(1.0, base_ms / max(ms - spread, 1e-9))
        return dict(ms_per_step_sharded_one_rank=ms, ms_per_step_unsharded=base_here, ms_per_step_headline=base_ms,
                    normaliser_added_us=1e3 * added, normaliser_added_us_spread=1e3 * spread,
                    blocks_ms={"unsharded": base_blocks, "sharded": c10d_blocks},
                    collective="torch.distributed all_gather_into_tensor (the default)",
                    ms_per_step_with_direct_ncclAllGather=statistics.median(direct_blocks) if direct_blocks else None,
                    predicted_weak_scaling_efficiency=[eff_lo, eff_hi], predicted_speedup_at_8_gpus=[8 * eff_lo, 8 * eff_hi],
                    groups={"headline": G, "probe_legs": groups},
                    note="fresh processes on this GPU, one-rank RCCL group: added = median(sharded) - median(unsharded) of five blocks each (both in "
                         "the sharded run's group count), signed; efficiency = the headline's step / the sharded step -+ the blocks' spread, as a "
                         "SCALE record computes it.  The 8-rank all-gather of 8 x 48 bytes adds its xGMI latency (a few us) and eight processes "
                         "share the host: a PREDICTION for the first multi-GPU run to be checked against")
    except Exception as exc:
        return dict(error=repr(exc))


def predicted_strong_scaling(ps, total, added_us, spread_us=None):
    """`total` particles over N GPUs from the P sweep (ms per step at total / N particles per GPU) plus what the sharded
    normaliser adds to a step: the data path has no other exchange (Algorithm/FastSlam.py:25-27).  64 particles do not fill an
    MI355X (2.0 us per particle-scan at 64, 1.3 at 256), so a fixed particle count gains less than N from N GPUs -- the
    model the first SCALE record is to be checked against."""
    rows, base = {}, None
    for n in (1, 2, 4, 8):
        r = ps.get(str(total // n))
        if r is None:
            continue
        ms = r["ms_per_step"] + (1e-3 * added_us if n > 1 else 0.0)
        if base is None:
            base = ms * n if n > 1 else ms                 # (without the 1-GPU point: relative to N x the smallest N measured)
            base_n = n
        rows[str(n)] = dict(particles_per_gpu=total // n, ms_per_step=round(ms, 5), particle_scans_per_sec=round(total / ms * 1e3),
                            speedup_vs_1_gpu=round((base / ms) if base_n == 1 else float("nan"), 3))
    s8 = rows.get("8", {}).get("speedup_vs_1_gpu")
    verdict = (f"STRONG scaling of {total} particles: {s8:.1f}x predicted at 8 GPUs, {'below' if s8 < 6.0 else 'at or above'} the north-star's 6x bar "
               f"({total // 8} particles per GPU do not fill an MI355X); WEAK scaling (64 per GPU) is predicted at 8 x t(64) / (t(64) + normaliser) "
               f"= {8 * ps['64']['ms_per_step'] / (ps['64']['ms_per_step'] + 1e-3 * added_us):.1f}x"
               if s8 is not None and s8 == s8 and "64" in ps else "no 1-GPU and 8-GPU points in the sweep")
    return dict(total_particles=total, per_n_gpus=rows, normaliser_added_us=round(added_us, 2), normaliser_added_us_spread=spread_us and round(spread_us, 2),
                note="from variants.p_sweep (one GPU, the same kernels) + variants.sharded_normaliser_probe; xGMI latency of the 24-byte-per-group "
                     "all-gather and host contention between ranks not included.  " + verdict)


def closed_loop_sharded(args, world, rank, device):
    """BASELINE configs 3 / 4 as one command: FastSLAM over the bundled Intel log (910 scans x 180 beams, reference defaults, maps
    growing from 50 m) through ParticleFilter.run(), the particles sharded over the ranks -- per scan one all-gather of the
    normaliser's partials, per resample the weights' all-gather and the point-to-point migration of whole particles
    (parallel.migrate_ragged).  Host-, PCIe- and collective-inclusive: a `step` is one scan of all particles."""
    pkg = importlib.import_module("slam-2d-lidar-scan_amd")
    dataio = importlib.import_module("slam-2d-lidar-scan_amd.dataio")
    par = par_mod()
    readings = dataio.read_npz(os.path.join(REPO, "tests", "golden", "intel_gfs.npz"))
    K = min(args.steps, len(readings)) if args.steps != 100 else len(readings)        # (the default --steps means the whole log)
    readings = readings[:K]
    total = args.total_particles or (args.particles or 64) * world
    first, count = par.shard_range(total, world, rank)
    u = 0.02
    ogP = [50.0, 50.0, readings[0], u, math.pi, 10, 180, 5 * u]
    smP = [1.4, 0.25, 2, 0.1, 0.25, 0.3, 0.15, 5]
    force = set(range(args.resample_every, K, args.resample_every)) if args.resample_every > 0 else set()

    def leg():
        # (SLAM2D_FORCE_DIST=1 on one rank: the sharded closed loop -- gate, all-gather over RCCL, publishing merge -- with nobody to wait for)
        kw = dict(total_particles=total, first_index=first, force_sharded=True) if dist.is_initialized() else {}
        import gc
        gc.collect()                                   # (the previous leg's maps go back to torch's cache: closed_loop_intel's note)
        pf = pkg.ParticleFilter(count, ogP, smP, device=device, rng=np.random.RandomState(0), **kw)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pf.run(readings, force_resample=force)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        el = time.perf_counter() - t0
        if dist.is_initialized():
            t = _ctl([el], device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, pf, res
    leg()                                                  # warm-up (allocator, first builds)
    el, pf, res = min((leg() for _ in range(3)), key=lambda r: r[0]) if os.environ.get("SLAM2D_BENCH_LEGS", "3") != "1" else leg()
    if rank == 0:
        print(json.dumps({
            "metric": "scans/sec (180-beam) x particles, closed loop over the Intel log", "value": total * K / el, "unit": "particle-scans/s",
            "n_gpus": world, "steps": K, "warmup": 0, "ms_per_step": 1e3 * el / K, "higher_is_better": True,
            "scaling": "strong" if args.total_particles else "weak", "vs_baseline": None,
            "dtype": "u32 fixed-point field, u64 exact accumulate, f64 priors/scores, u32 packed counts", "data": "bundled Intel log (tests/golden/intel_gfs.npz)",
            "config": {"workload": "config3: FastSLAM closed loop, reference defaults, 910-scan Intel log, maps growing from 50 m",
                       "total_particles": total, "particles_per_gpu": count, "resample_every": args.resample_every,
                       "parallelism": (f"particles sharded x{world}, " if dist.is_initialized() else "single GPU, ") +
                                      f"{pf.n_groups} particle group(s) per GPU on the event-free calls" + (" (sharded commit: gate + all-gather + publishing merge)" if pf.sharded else "")},
            "scans_per_sec": K / el, "resamples": len(res), "state_moving_resamples": pf.stats["state_moving_resamples"],
            "scans_redone": pf.stats["redo"], "filter_stats": dict(pf.stats), "final_map_of_particle_0": [pf.engine.maps[0].rows, pf.engine.maps[0].cols]}), flush=True)


def spawn_check(args, world, rank):
    """Launch plumbing only (CPU): every rank joins a gloo group and adds 1; rank 0 prints what it saw."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"spawn_check": True, "n_gpus": world, "ranks_seen": seen, "gpus_requested": args.gpus}))
    return 0 if seen == max(1, args.gpus) else 3


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))                 # no launcher around us: become one (one rank per GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly one rank per GPU "
                         "(python bench.py --gpus N starts them itself)")
    if args.spawn_check:
        sys.exit(spawn_check(args, world, rank))
    force_dist = os.environ.get("SLAM2D_FORCE_DIST") == "1"     # exercise the sharded code path on one GPU
    cfg = WORKLOADS.get(args.workload, WORKLOADS["ref2level"])
    P, K, W, R = args.particles or WORKLOAD_PARTICLES.get(args.workload, 64), args.steps, args.warmup, max(1, args.repeats)
    if args.total_particles:
        if args.total_particles % world:
            raise SystemExit(f"bench.py: --total-particles {args.total_particles} is no multiple of the {world} ranks")
        P = args.total_particles // world
    nprobe = min(8, K)
    closed = args.workload == "config3"
    scen = None if closed else Scenario(cfg, P, W + max(K, nprobe), seed=0, rank=rank)
    cpu = None
    if not closed and not args.no_cpu_baseline and world == 1 and rank == 0:
        cpu = cpu_baseline(cfg, scen, args.cpu_seconds, args.cpu_workers)      # before HIP is initialised: the pool forks
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    ngpu = torch.cuda.device_count()
    if world > ngpu and not args.share_gpu:
        raise SystemExit(f"bench.py: {world} ranks but {ngpu} visible GPU(s); --share-gpu is the dry mode that lets ranks share one")
    dev_index = local_rank % ngpu
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    ranks_seen = 1
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        census = _ctl([1.0], device)
        dist.all_reduce(census)                       # every rank adds one over the backend the bench will use
        ranks_seen = int(census.item())
        if ranks_seen != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py: the {args.backend} group holds {ranks_seen} ranks, expected {world}")
    if args.workload == "config3":
        closed_loop_sharded(args, world, rank, device)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    E = importlib.import_module("slam-2d-lidar-scan_amd.engine")
    lib = E._lib.lib()
    G = bench_groups(args.groups, P)
    hot = make_hot_path(cfg, P, scen, device, G)

    stages = [E._lib.STAGE_SWEEP, E._lib.STAGE_BLUR, E._lib.STAGE_SCATTER, E._lib.STAGE_UPDATE,
              E._lib.STAGE_SELECT, E._lib.STAGE_ENDPOINTS, E._lib.STAGE_BOUND, E._lib.STAGE_EXACT]

    def collect():
        out = {}
        for st in stages:
            tot, n = C.c_double(0), C.c_int32(0)
            E._lib.check(lib.slam2d_prof_collect(st, C.byref(tot), C.byref(n)), "prof_collect")
            if n.value:
                out[E._lib.STAGE_NAMES[st]] = dict(total_ms=tot.value, launches=n.value, avg_us=1e3 * tot.value / n.value)
        return out

    # warm-up: the first steps build every field tile (nothing is known to hold the free-space constant yet).  Sharded: the ranks
    # start it together (a rank's groups wait ON THE DEVICE for the merge behind the all-gather, i.e. for the slowest rank -- bounded,
    # 30 s -- and ranks leave their set-up seconds apart)
    if dist.is_initialized():
        torch.cuda.synchronize()
        dist.barrier()
    for s in range(W):
        hot.step(s)
    flags = hot.take_flags()        # synchronises; raises on any fault
    # a few bracketed steps (outside the timed region: an event pair costs ~5 us of stream time) find the dominant kernel
    E._lib.check(lib.slam2d_prof_enable(sum(1 << st for st in stages), 4 * 8 + 8), "prof_enable")
    for s in range(W, W + nprobe):
        hot.step(s)
    hot.take_flags()
    probe_ms = collect()
    dom_stage = max(stages, key=lambda st: probe_ms.get(E._lib.STAGE_NAMES[st], {}).get("total_ms", 0.0))
    launches_per_step_of = {k: v["launches"] / nprobe for k, v in probe_ms.items()}
    # what an event pair adds to a bracketed launch: the same pair with nothing in between, recorded while the stream is busy
    timer = lib.slam2d_timer_create()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pair = []
    for i in range(12):
        hot.step(W + i % nprobe)
        lib.slam2d_timer_start(timer, stream)
        lib.slam2d_timer_stop(timer, stream)
        ms = C.c_float(0)
        E._lib.check(lib.slam2d_timer_elapsed_ms(timer, C.byref(ms)), "timer")
        pair.append(1e3 * ms.value)
    lib.slam2d_timer_destroy(timer)
    hot.take_flags()
    overhead_us = statistics.median(pair)
    # timed region: R blocks of K steps, each bracketed by barrier + synchronize; only the dominant kernel keeps an event pair,
    # and only around every n-th of its launches (an event pair holds the stream for ~6 us on each side of the kernel; odd n: a
    # stage with one launch per level alternates between the levels)
    dom_lps = launches_per_step_of.get(E._lib.STAGE_NAMES[dom_stage], 1.0)
    every = int(max(1, min(PROF_EVERY_MAX, (R * K * dom_lps) // 13)))       # (>= 13 timed launches: the driver's 5 x 20 steps bracket every 15th)
    if every > 1 and every % 2 == 0:
        every -= 1
    E._lib.check(lib.slam2d_prof_every(every), "prof_every")
    E._lib.check(lib.slam2d_prof_enable(1 << dom_stage, int(R * K * dom_lps) // every + 16), "prof_enable")
    blocks, mine, enq = [], [], []
    for _ in range(R):
        el, own = timed_run(hot, W, K)
        enq.append(timed_run.host_enqueue_s)
        flags = flags | hot.take_flags()
        blocks.append(el)
        mine.append(own)
    lib.slam2d_prof_disable()
    E._lib.check(lib.slam2d_prof_every(1), "prof_every")
    stage_ms = collect()
    elapsed = statistics.median(blocks)
    # N > 1: what the sharded normaliser (24-byte all-gather + merge launch) adds to a step on this rank
    per_rank, normaliser_added_us = None, None
    if dist.is_initialized():
        own_ms = 1e3 * statistics.median(mine) / K
        saved = hot.sharded
        hot.sharded = False                          # the same steps with every rank normalising on its own: no collective
        if isinstance(hot, HotPathGroups):
            hot.parts_all = hot.parts_local
            hot.world, total_saved, hot.total_particles = 1, hot.total_particles, hot.P
        local = statistics.median([timed_run(hot, W, K)[1] for _ in range(3)])
        hot.sharded = saved
        if isinstance(hot, HotPathGroups):
            hot.world, hot.total_particles = world, total_saved
            hot.parts_all = torch.zeros(3 * hot.G * world, dtype=torch.float64, device=device) if saved else hot.parts_local
        hot.take_flags()
        rec = _ctl([own_ms, 1e3 * (statistics.median(mine) - local) / K], device)
        got = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(got, rec)
        per_rank = [float(g[0]) for g in got]
        normaliser_added_us = [float(g[1]) for g in got]

    variants = {}
    if not args.no_variants and world == 1:          # (multi-GPU scaling runs: the headline only)
        def variant(bnb, prune, note):
            saved = [(lv, lv.c.bnb) for lv in hot.levels()]
            for lv, _ in saved:
                if not bnb:
                    lv.c.bnb = 0
            hot.prune = prune
            for s in range(W):
                hot.step(s)
            hot.take_flags()
            el = statistics.median([timed_run(hot, W, K)[0] for _ in range(3)])
            hot.take_flags()
            for lv, v in saved:
                lv.c.bnb = v
            hot.prune = False
            return dict(value=P * world * K / el, unit="particle-scans/s", ms_per_step=1e3 * el / K, repeats=3, note=note)
        rf_main = roofline_of(hot, scen, P, 1e3 * elapsed / K, stage_ms, launches_per_step_of, args.workload, overhead_us, probe_ms=probe_ms)
        if world == 1 and not dist.is_initialized():
            variants["sharded_normaliser_probe"] = normaliser_probe(args.workload, P, min(K, 60), W, G, 1e3 * elapsed / K)
        if any(lv is not None and lv.bnb for lv in (hot.coarse, hot.fine)):
            variants["brute_force_sweep"] = variant(False, False, "every pose of the cube scored (k_sweep), the whole cube materialised "
                                                    "in HBM: the round-1 headline path")
        variants["prior_pruned_sweep"] = variant(False, True, "brute-force sweep restricted to the motion prior's ring where that "
                                                 "settles the particle (SLAM2D_MATCH_PRUNE_BY_PRIOR)")
        for name, (p2, k2) in {"ref2level": (64, 40), "config5": (128, 12)}.items():
            if name != args.workload:
                variants[name] = side_workload(name, p2, k2, 4, device, rank)
        if args.workload == "config2":
            # inputs the branch and bound does NOT like (the headline's scans fit the maps, 1 % of the pose tiles survive):
            # SURVEY 8(d)'s structure-free scans and estimates far off the motion prior's ring, each next to the brute-force sweep
            variants["config2_worst"] = side_workload("config2", P, 40, 6, device, rank, mode="worst", with_brute=True)
            variants["config2_displaced"] = side_workload("config2", P, 40, 6, device, rank, mode="displaced", with_brute=True)
            variants["p_sweep"] = p_sweep("config2", (16, 32, 64, 128, 256, 512), 30, 6, device, rank)
        if world == 1 and args.workload == "config2":
            c3 = config3_closed_loop(64, device)
            if c3 is not None:
                variants["config3_closed_loop"] = c3
            ds = dropin_serial(64, 53, device)
            if ds is not None:
                variants["dropin_serial"] = ds
    else:
        rf_main = roofline_of(hot, scen, P, 1e3 * elapsed / K, stage_ms, launches_per_step_of, args.workload, overhead_us, probe_ms=probe_ms)

    if rank == 0:
        total_units = P * world * K
        rf_main.setdefault("event_pairs", {})["every"] = every
        out = {
            "metric": f"scans/sec ({cfg['beams']}-beam) x particles at fixed search volume",
            "value": total_units / elapsed, "unit": "particle-scans/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong" if args.total_particles else "weak", "vs_baseline": None,
            "dtype": "u32 fixed-point field, u64 exact accumulate, f64 priors/scores, u32 packed counts", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['note']}", "particles_per_gpu": P,
                       "total_particles": P * world, "pose_hypotheses_per_particle_scan":
                       hot.coarse.ntheta * hot.coarse.nx ** 2 + (hot.fine.ntheta * hot.fine.nx ** 2 if hot.fine else 0),
                       "pose_scoring": "branch and bound over 4x4 pose tiles (exact arg-max / draw, confidence within 2e-8)"
                       if hot.coarse.bnb else "brute-force sweep",
                       "particle_groups_per_gpu": G if isinstance(hot, HotPathGroups) else 1,
                       "parallelism": f"particles sharded x{world}, one 24-byte-per-rank {'RCCL' if args.backend == 'nccl' else 'gloo (dry mode)'} "
                                      "all-gather of the weight normaliser per scan" if world > 1 else
                                      (f"single GPU, {G} particle groups on {G} HIP streams, their normaliser merged on the device"
                                       if isinstance(hot, HotPathGroups) else "single GPU, one stream"),
                       "input": f"{scen.mode}: " + {"tracked": "scans ray-cast from the mapped world, every estimate within two cells of the true pose -- the "
                                                               "steady state of a converged filter and the BEST case of the branch and bound; the other inputs "
                                                               "are in config.other_measurements (worst_case_ms_per_step, displaced)",
                                                    "worst": "structure-free ranges (SURVEY 8d)", "displaced": "estimates 1.5 m / 0.2 rad off"}[scen.mode]},
            "scans_per_sec": K / elapsed,
            "timed_blocks": {"repeats": R, "steps_each": K, "ms_per_step_of_each": [round(1e3 * b / K, 5) for b in blocks], "reported": "median",
                             "host_enqueue_ms_per_step": round(1e3 * statistics.median(enq) / K, 5),
                             "host_issue": hot.E._lib.group_policy()},      # cores / local ranks -> worker threads or not (include/slam2d.h: slam2d_group_policy)
            "roofline": rf_main,
            "stages_probe": {k: {"avg_us": round(v["avg_us"], 2), "launches": v["launches"]} for k, v in probe_ms.items()},
            "algorithmic_bytes_per_particle_scan": hot.algorithmic_bytes(scen),
            "fault_flags": int(np.bitwise_or.reduce(flags)) if len(flags) else 0,
        }
        if dist.is_initialized():
            out["ranks"] = {"backend": args.backend, "ranks_seen": ranks_seen, "gpus_visible": ngpu, "shared_gpu": bool(args.share_gpu and world > ngpu),
                            "ms_per_step_per_rank": per_rank, "normaliser_added_us_per_rank": normaliser_added_us}
        if variants:
            out["variants"] = variants
            # what the driver's trimmed record keeps is `config` and `roofline`: the numbers a reader needs beside the headline go there
            om = {}
            bf, worst = variants.get("brute_force_sweep"), variants.get("config2_worst")
            if bf:
                om["every_pose_ms_per_step"] = round(bf["ms_per_step"], 5)          # brute-force sweep of the whole cube, same scans
            if worst:
                om["worst_case_ms_per_step"] = round(worst["ms_per_step"], 5)       # structure-free scans
                om["worst_case_vs_brute_force"] = round(worst.get("vs_brute_force", float("nan")), 3)
            c3, ds, ps = variants.get("config3_closed_loop"), variants.get("dropin_serial"), variants.get("p_sweep")
            if c3 and "scans_per_sec" in c3:
                om["closed_loop_scans_per_sec"] = round(c3["scans_per_sec"], 1)     # BASELINE config 3: 64 particles x the Intel log, host in the loop
            if ds and "scans_per_sec" in ds:
                om["dropin_serial_scans_per_sec"] = round(ds["scans_per_sec"], 1)   # the reference's unchanged caller over the drop-in classes
            if ps:
                om["p_sweep_us_per_particle_scan"] = {k: round(v["us_per_particle_scan"], 3) for k, v in ps.items() if k in ("64", "256", "512")}
                npb = variants.get("sharded_normaliser_probe") or {}
                # (what sharding adds to a rank's step: the sharded step of the probe against the headline's)
                shard_us = 1e3 * (npb["ms_per_step_sharded_one_rank"] - npb["ms_per_step_headline"]) if "ms_per_step_sharded_one_rank" in npb else 0.0
                out["predicted_strong_scaling"] = om["predicted_strong_scaling_512_particles"] = predicted_strong_scaling(
                    ps, 512, max(shard_us, 0.0), npb.get("normaliser_added_us_spread"))
            out["config"]["other_measurements"] = om
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if os.environ.get("SLAM2D_BENCH_THREAD_CPU") == "1":      # which host threads of this rank burned CPU (a diagnosis aid for core quotas)
        rows = []
        for t in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{t}/stat").read()
                name = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                rows.append((int(rest[11]) + int(rest[12]), name, t))          # utime + stime, clock ticks
            except (OSError, ValueError):
                pass
        rows.sort(reverse=True)
        print(f"[rank {rank}] thread cpu ticks: " + ", ".join(f"{n}:{c}" for c, n, _ in rows[:8]), file=sys.stderr, flush=True)
    if getattr(main, "hung_probe", False):
        os._exit(0)                                    # (a stuck helper thread must not keep the process)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
