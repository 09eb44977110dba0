"""ctypes binding of libslam2d_hip.so (the C ABI in include/slam2d.h).

There is no CPU fallback: if the shared library is missing or was not built,
``lib()`` raises.  ``build_library()`` compiles it in-tree with hipcc for
gfx950 (cross-compiles without a GPU).
"""
import ctypes as C
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("SLAM2D_LIB") or os.path.join(PKG_DIR, "libslam2d_hip.so")   # SLAM2D_LIB: another build of the same ABI
SRC_PATH = os.path.join(PKG_DIR, "csrc", "slam2d.hip")
INCLUDE_DIR = os.path.join(REPO_DIR, "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread"]

# fault bits (include/slam2d.h SLAM2D_F_*)
F_WINDOW_OUTSIDE_MAP = 0x01
F_FIELD_INDEX = 0x02
F_ENDPOINT_OUTSIDE = 0x04
F_UPDATE_OUTSIDE_MAP = 0x08
F_COUNT_OVERFLOW = 0x10
F_FLOOR_REDO = 0x20
F_SYNC_TIMEOUT = 0x40
F_SCAN_VOIDED = 0x80           # (snapshot only: the commit was a no-op on the device)
FATAL_FLAGS = F_WINDOW_OUTSIDE_MAP | F_FIELD_INDEX | F_ENDPOINT_OUTSIDE | F_UPDATE_OUTSIDE_MAP | F_COUNT_OVERFLOW | F_SYNC_TIMEOUT
FLAG_NAMES = {
    F_WINDOW_OUTSIDE_MAP: "search window outside the map (grow the map first)",
    F_FIELD_INDEX: "occupied cell mapped outside the search field",
    F_ENDPOINT_OUTSIDE: "beam endpoint +/- search radius left the search field",
    F_UPDATE_OUTSIDE_MAP: "map update touched a cell outside the map",
    F_COUNT_OVERFLOW: "16-bit cell count overflow",
    F_FLOOR_REDO: "field minimum differed from the analytic floor (clamp redone)",
    F_SYNC_TIMEOUT: "a device-side wait of the groups' normaliser gave up after its bound (30 s unless set: a producer never arrived)",
}
INIT_CELL = 0x00010002
INIT_CELL_WIDE = (1 << 32) | 2        # the same in the 64-bit cell format (Slam2dMap.wide)
COUNT_LIMIT = 65535                  # largest `total` a narrow cell holds
MAX_BLUR_RADIUS = 16
MAX_BEAMS = 2048
SPOKE_BAND = 16
SYNC_WORDS = 4
ABI_VERSION = 17
MATCH_PRUNE_BY_PRIOR = 1
MATCH_PRIOR_READY = 2
PRUNE_MARGIN = 40.0
BNB_MARGIN = 30.0

STAGE_SWEEP, STAGE_BLUR, STAGE_SCATTER, STAGE_UPDATE, STAGE_SELECT, STAGE_ENDPOINTS, STAGE_BOUND, STAGE_EXACT = range(8)
STAGE_NAMES = {STAGE_SWEEP: "k_sweep", STAGE_BLUR: "k_blur_clamp", STAGE_SCATTER: "k_occ_scatter",
               STAGE_UPDATE: "k_grid_update", STAGE_SELECT: "k_select", STAGE_ENDPOINTS: "k_endpoints",
               STAGE_BOUND: "k_bound", STAGE_EXACT: "k_exact"}

_vp = C.c_void_p


class Slam2dMap(C.Structure):
    _fields_ = [("cells", _vp), ("X", _vp), ("Y", _vp),
                ("rows", C.c_int32), ("cols", C.c_int32), ("pitch", C.c_int32), ("bits_pitch", C.c_int32),
                ("lim_x0", C.c_double), ("lim_x1", C.c_double), ("lim_y0", C.c_double), ("lim_y1", C.c_double),
                ("occ_bits", _vp), ("wide", C.c_int32), ("_pad", C.c_int32)]


class Slam2dLidar(C.Structure):
    _fields_ = [("unit", C.c_double), ("max_range", C.c_double), ("fov", C.c_double), ("wall_half", C.c_double),
                ("beams", C.c_int32), ("num_spokes", C.c_int32), ("spoke_start", C.c_int32), ("lut_w", C.c_int32),
                ("lut_xs", _vp),
                ("spoke_band", _vp), ("spoke_cells", _vp), ("spoke_r", _vp), ("num_bands", C.c_int32), ("_pad", C.c_int32),
                ("lut_xs_step", C.c_double)]


class Slam2dFrame(C.Structure):
    _fields_ = [("xlo", C.c_double), ("ylo", C.c_double), ("xhi", C.c_double), ("yhi", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("field_min", C.c_double),
                ("fh", C.c_int32), ("fw", C.c_int32), ("mx0", C.c_int32), ("mx1", C.c_int32),
                ("my0", C.c_int32), ("my1", C.c_int32), ("redo", C.c_int32), ("min_known", C.c_int32),
                ("field_max", C.c_double)]


class Slam2dPartial(C.Structure):
    _fields_ = [("max", C.c_double), ("sumexp", C.c_double), ("argmax", C.c_int32), ("has_nan", C.c_int32)]


class Slam2dLevel(C.Structure):
    _fields_ = [("step", C.c_double), ("reach", C.c_double), ("log_miss", C.c_double), ("floor_value", C.c_double),
                ("cost_scale", C.c_double),
                ("blur_radius", C.c_int32), ("fmax", C.c_int32), ("fpitch", C.c_int32), ("wmax", C.c_int32),
                ("blur_w", _vp),
                ("ncell", C.c_int32), ("ntheta", C.c_int32), ("fine", C.c_int32), ("kmax", C.c_int32),
                ("thetas", _vp), ("theta_cos", _vp), ("theta_sin", _vp),
                ("rv_coef", C.c_double), ("tw_coef", C.c_double), ("max_move_dev", C.c_double),
                ("frames", _vp), ("axis_x", _vp), ("axis_y", _vp), ("occ", _vp), ("field", _vp),
                ("cells", _vp), ("kcount", _vp), ("prior", _vp), ("cube", _vp),
                ("partials", _vp), ("npartial", C.c_int32), ("tmax", C.c_int32), ("tilemask", _vp),
                ("tilestate", _vp), ("tilemin", _vp), ("tilemax", _vp),
                ("tilelist", _vp), ("tilecount", _vp), ("tileneed", _vp), ("freerow", _vp),
                ("ring", _vp), ("prune_state", _vp), ("ring_cap", C.c_int32),
                ("gmin", _vp), ("gmin2", _vp), ("pcells", _vp), ("bounds", _vp), ("tile_pmax", _vp), ("bnb_best", _vp),
                ("gmin3d", _vp), ("p3cells", _vp), ("bounds1", _vp), ("seed_key", _vp),
                ("beam_xy", _vp), ("sync", _vp), ("bnb", C.c_int32), ("ep_group", C.c_int32), ("occ_gen", C.c_int32), ("arrive", _vp),
                ("gmin2b", _vp), ("g2b_pitch", C.c_int32), ("reserved0", C.c_int32), ("theta_umax", _vp)]


class Slam2dMatch(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("theta", C.c_double),
                ("confidence", C.c_double), ("log_confidence", C.c_double), ("best_score", C.c_double),
                ("pick", C.c_int32), ("argmax", C.c_int32)]


class Slam2dGroup(C.Structure):
    """One particle group of a multi-stream scan (include/slam2d.h): HOST pointers to its level descriptors, device pointers
    to its slices of the per-particle arrays, its stream and events."""
    _fields_ = [("coarse", C.POINTER(Slam2dLevel)), ("fine", C.POINTER(Slam2dLevel)), ("d_maps", _vp),
                ("P", C.c_int32), ("est_stride", C.c_int32),
                ("d_est", _vp), ("d_psi_cs", _vp), ("d_uniform", _vp),
                ("d_prev_pose", _vp), ("d_heading", _vp), ("d_est_out", _vp), ("d_psi_out", _vp),
                ("d_coarse", _vp), ("d_fine", _vp), ("d_flags", _vp), ("d_logw", _vp), ("d_part", _vp),
                ("d_report", _vp), ("d_flag_snapshot", _vp),
                ("stream", _vp), ("ev_matched", _vp), ("ev_done", _vp), ("d_pull", _vp), ("h_uniform", _vp), ("d_pull_next", _vp)]


class Slam2dScan(C.Structure):
    """What all groups of a scan share (include/slam2d.h)."""
    _fields_ = [("d_ranges", _vp), ("est_moving_dist", C.c_double),
                ("raw_theta", C.c_double), ("prev_raw_theta", C.c_double), ("raw_turn", C.c_double),
                ("has_turn", C.c_int32), ("options", C.c_uint32), ("abort_mask", C.c_uint32), ("n_abort_flags", C.c_int32),
                ("d_abort_flags", _vp), ("ev_inputs", _vp),
                ("d_logw_all", _vp), ("n_local", C.c_int32), ("n_parts", C.c_int32), ("d_parts", _vp),
                ("total_particles", C.c_int64), ("d_w", _vp), ("d_stats", _vp),
                ("norm_stream", _vp), ("ev_merged", _vp), ("wait_merged", C.c_int32), ("merge", C.c_int32), ("d_norm_sync", _vp),
                ("h_ranges", _vp), ("match_seq", C.c_uint32), ("report_seq", C.c_uint32), ("h_seq", _vp), ("h_pack", _vp), ("d_pack", _vp),
                ("pack_doubles", C.c_int32), ("h_next_ranges", _vp), ("next_raw_theta", C.c_double), ("next_prev_raw_theta", C.c_double),
                ("next_raw_turn", C.c_double), ("next_has_turn", C.c_int32)]


STRUCTS = {"Slam2dGroup": Slam2dGroup, "Slam2dScan": Slam2dScan, "Slam2dMap": Slam2dMap, "Slam2dLidar": Slam2dLidar, "Slam2dFrame": Slam2dFrame,
           "Slam2dLevel": Slam2dLevel, "Slam2dMatch": Slam2dMatch, "Slam2dPartial": Slam2dPartial}

# name -> (restype, argtypes); every symbol include/slam2d.h declares
SIGNATURES = {
    "slam2d_abi_version": (C.c_int, []),
    "slam2d_sizeof": (C.c_int, [C.c_char_p]),
    "slam2d_device_count": (C.c_int, []),
    "slam2d_field_build": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dLevel), _vp, C.c_int32, _vp, C.c_int32,
                                     _vp, _vp]),
    "slam2d_sweep": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dLevel), C.c_int32, _vp, C.c_int32, _vp,
                               C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "slam2d_match": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dLevel), _vp, C.c_int32, _vp, C.c_int32, _vp,
                               C.c_double, _vp, _vp, _vp, _vp, C.c_uint32, _vp]),
    "slam2d_grid_update": (C.c_int, [C.POINTER(Slam2dLidar), _vp, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "slam2d_prior": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int32, C.c_double, _vp, C.c_int32, _vp, _vp, _vp]),
    "slam2d_post_match": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "slam2d_weights_normalize": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "slam2d_grid_update_weights": (C.c_int, [C.POINTER(Slam2dLidar), _vp, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "slam2d_grid_update_weights_local": (C.c_int, [C.POINTER(Slam2dLidar), _vp, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "slam2d_scan_match": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dLevel), C.POINTER(Slam2dLevel), _vp, C.c_int32, _vp,
                                    C.c_double, C.c_double, C.c_int32, C.c_double, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp,
                                    _vp, C.c_uint32, _vp]),
    "slam2d_scan_commit": (C.c_int, [C.POINTER(Slam2dLidar), _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, C.c_uint32, _vp]),
    "slam2d_scan_commit_next": (C.c_int, [C.POINTER(Slam2dLidar), _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, C.c_uint32, C.c_double, C.c_double, C.c_int32, C.c_double, _vp, _vp, _vp]),
    "slam2d_groups_match": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dGroup), C.c_int32, C.POINTER(Slam2dScan)]),
    "slam2d_groups_commit": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dGroup), C.c_int32, C.POINTER(Slam2dScan)]),
    "slam2d_groups_step": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dGroup), C.c_int32, C.POINTER(Slam2dScan)]),
    "slam2d_group_policy": (C.c_int, [C.POINTER(C.c_int32)]),
    "slam2d_groups_match_begin": (C.c_int, [C.POINTER(Slam2dLidar), C.POINTER(Slam2dGroup), C.c_int32, C.POINTER(Slam2dScan)]),
    "slam2d_groups_join": (C.c_int, []),
    "slam2d_host_wait_seq": (C.c_int, [_vp, C.c_uint32, C.c_double]),
    "slam2d_norm_gate": (C.c_int, [_vp, C.c_int32, _vp]),
    "slam2d_weights_merge_publish": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp]),
    "slam2d_weights_merge_publish_report": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_uint32, _vp]),
    "slam2d_weights_local": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp, _vp]),
    "slam2d_weights_merge": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.c_int64, _vp, _vp, _vp]),
    "slam2d_gather_maps": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int64, _vp]),
    "slam2d_map_image": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "slam2d_map_fill": (C.c_int, [_vp, C.c_int64, C.c_uint32, _vp]),
    "slam2d_map_grow": (C.c_int, [C.POINTER(Slam2dMap), C.POINTER(Slam2dMap), C.c_int32, C.c_int32, _vp]),
    "slam2d_map_refresh_bits": (C.c_int, [_vp, _vp, C.c_int32, _vp]),
    "slam2d_device_sincos": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    "slam2d_prof_enable": (C.c_int, [C.c_uint32, C.c_int32]),
    "slam2d_prof_collect": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "slam2d_prof_every": (C.c_int, [C.c_int32]),
    "slam2d_prof_disable": (None, []),
    "slam2d_streams_create": (C.c_int, [C.POINTER(_vp), C.c_int32]),
    "slam2d_stream_destroy": (None, [_vp]),
    "slam2d_event_create": (_vp, []),
    "slam2d_event_destroy": (None, [_vp]),
    "slam2d_event_record": (C.c_int, [_vp, _vp]),
    "slam2d_stream_wait_event": (C.c_int, [_vp, _vp]),
    "slam2d_timer_create": (_vp, []),
    "slam2d_timer_destroy": (None, [_vp]),
    "slam2d_timer_start": (C.c_int, [_vp, _vp]),
    "slam2d_timer_stop": (C.c_int, [_vp, _vp]),
    "slam2d_timer_elapsed_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
}

_lib = None


class Slam2dError(RuntimeError):
    pass


def build_library(force=False, verbose=False):
    """Compile csrc/slam2d.hip for gfx950 into libslam2d_hip.so (in-tree)."""
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(
            os.path.getmtime(SRC_PATH), os.path.getmtime(os.path.join(INCLUDE_DIR, "slam2d.h"))):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise Slam2dError("hipcc not found: cannot build libslam2d_hip.so")
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", INCLUDE_DIR, SRC_PATH, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise Slam2dError("hipcc failed:\n" + res.stdout + res.stderr)
    global _lib
    _lib = None
    return LIB_PATH


def lib():
    """The loaded library with argtypes set.  Raises if it is not built: the
    product path has no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Slam2dError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the HIP path)")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if L.slam2d_abi_version() != ABI_VERSION:
        raise Slam2dError(f"{LIB_PATH} has ABI version {L.slam2d_abi_version()}, the binding expects {ABI_VERSION}: rebuild it")
    for name, st in STRUCTS.items():
        n = L.slam2d_sizeof(name.encode())
        if n != C.sizeof(st):
            raise Slam2dError(f"ABI mismatch: sizeof({name}) is {n} in the library, {C.sizeof(st)} in the binding")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        kind = "HIP error" if rc > 0 else "timeout" if rc == -3 else "argument error"
        raise Slam2dError(f"{what}: {kind} {rc}")


def describe_flags(bits):
    return "; ".join(msg for bit, msg in FLAG_NAMES.items() if bits & bit) or "none"


def group_policy():
    """How slam2d_groups_* issue their groups on this host (include/slam2d.h: slam2d_group_policy): dict of the host cores this
    process may use, the local ranks assumed to share them, whether a worker thread per group is used and whether the threads wait
    politely.  Decided once per process, at the first call of this function or of a grouped step."""
    out = (C.c_int32 * 4)()
    check(lib().slam2d_group_policy(out), "slam2d_group_policy")
    return dict(cores=out[0], local_ranks=out[1], threads=bool(out[2]), polite=bool(out[3]))
