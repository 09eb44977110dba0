/*
 * slam2d.h -- C ABI of libslam2d_hip.so: MI355X (gfx950) kernels for the 2-D lidar
 * FastSLAM hot path of xiaofeng419/SLAM-2D-LIDAR-SCAN.
 *
 * The reference is pure Python and has no FFI layer; its boundary for this path is
 * the class surface of Utils/OccupancyGrid.py:OccupancyGrid and
 * Utils/ScanMatcher_OGBased.py:ScanMatcher as driven by Algorithm/FastSlam.py
 * (SURVEY.md section 8b).  Each entry point below replaces the body of one of
 * those methods for a batch of P particles; the Python classes in
 * slam-2d-lidar-scan_amd/ keep the reference's constructor/method/attribute
 * surface and call these through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C: PODs, raw pointers, sizes; no C++ or torch types.
 *   - every pointer named d_* (and every pointer inside the structs) is DEVICE memory
 *     owned by the caller (the Python side allocates it as torch tensors);
 *     the library allocates no device memory and keeps no state between calls -- except the sleeping host threads of
 *     slam2d_groups_* (one per particle group beyond the first, created on first use; SLAM2D_GROUP_THREADS=0: none) and
 *     the event pairs of slam2d_prof_*.  The slam2d_groups_* calls are not re-entrant (one caller at a time).
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     without synchronising; results are ordered after the call on that stream.
 *   - return value: 0 on success, otherwise a hipError_t code (> 0) or a
 *     SLAM2D_E_* code (< 0).  No exceptions cross the boundary.
 *   - data-dependent faults (window outside the map, count overflow, cell list
 *     overflow) are reported by OR-ing SLAM2D_F_* bits into d_flags[p]; the offending
 *     access is skipped, never performed out of bounds.
 *   - row-major everywhere; images are [rows = y][cols = x].
 *
 * Search field format (Slam2dLevel.field, one uint32 per field cell): the reference's
 *   probSP (Utils/ScanMatcher_OGBased.py:41-45) holds values in [probMin, 0]; the field
 *   stores the non-negative fixed-point COST  c = rint(-probSP * cost_scale),
 *   cost_scale = 2^k chosen per level so that -probMin * 2^k < 2^32 (k = 31 for the
 *   reference's parameters: resolution 4.7e-10, ~30x finer than float32 at the same
 *   4 bytes).  Pose scores are then exact integer sums (uint64), so they do not depend
 *   on summation order and exact ties of the reference stay exact ties.
 *
 * Cell format of a particle's map (one uint32 per cell):
 *     bits 31..16 = occupancyGridVisited count, bits 15..0 = occupancyGridTotal count
 *   (reference: two float64 arrays initialised to 1 and 2, Utils/OccupancyGrid.py:13-14;
 *    hit: visited += 2, total += 2; miss: total += 1, :148-152).  A cell is occupied
 *   iff 2*visited > total (== visited/total > 0.5, Utils/ScanMatcher_OGBased.py:29-31).
 *   That format holds 32766 observations of one cell; a map that may exceed it is kept in 64-bit cells
 *   (Slam2dMap.wide) -- the Python side promotes a map before that can happen; on a narrow map the update
 *   kernel raises SLAM2D_F_COUNT_OVERFLOW instead of wrapping.
 */
#ifndef SLAM2D_H
#define SLAM2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM2D_ABI_VERSION 17
#define SLAM2D_SPOKE_BAND 16         /* radial band width of the beam-major spoke table, in cells */

/* library error codes (negative; positive values are hipError_t) */
#define SLAM2D_E_BADARG   (-1)
#define SLAM2D_E_TOOLARGE (-2)   /* a size exceeds a compiled-in limit */
#define SLAM2D_E_TIMEOUT  (-3)   /* slam2d_host_wait_seq: the device did not publish the awaited scan in time */

/* per-particle fault bits OR-ed into d_flags[p] */
#define SLAM2D_F_WINDOW_OUTSIDE_MAP 0x01u  /* search window not inside the map: grow first (checkAndExapndOG) */
#define SLAM2D_F_FIELD_INDEX        0x02u  /* an occupied cell mapped outside the field (reference: IndexError) */
#define SLAM2D_F_ENDPOINT_OUTSIDE   0x04u  /* a beam endpoint +/- search radius left the field */
#define SLAM2D_F_UPDATE_OUTSIDE_MAP 0x08u  /* map update touched a cell outside the map */
#define SLAM2D_F_COUNT_OVERFLOW     0x10u  /* a 16-bit count would overflow */
#define SLAM2D_F_FLOOR_REDO         0x20u  /* informational: field minimum != analytic floor, clamp pass redone */
#define SLAM2D_F_SYNC_TIMEOUT       0x40u  /* a device-side wait of the groups' normaliser / gates (d_norm_sync) gave up after its bound */
#define SLAM2D_F_SCAN_VOIDED        0x80u  /* informational, in a voided scan's fault-bit SNAPSHOT only (never in d_flags): the commit did nothing
                                              on the device (abort_mask).  With SLAM2D_F_SYNC_TIMEOUT beside it the scan is intact and may be
                                              run again (the gate gave up BEFORE anything was written); without this bit a timeout is fatal */

#define SLAM2D_INIT_CELL 0x00010002u       /* visited = 1, total = 2 */
#define SLAM2D_MAX_BLUR_RADIUS 16
#define SLAM2D_MAX_BEAMS 2048
#define SLAM2D_SYNC_WORDS 4            /* arrival counters per particle in Slam2dLevel.sync */

/* One particle's map (device-resident).  X[cols] / Y[rows] are the stored cell-centre
 * coordinates (reference OccupancyGridX[0, :] / OccupancyGridY[:, 0]; rank-1 by
 * construction, growth included -- Utils/OccupancyGrid.py:12,83-85).  lim_* mirror
 * mapXLim / mapYLim (:19-20,86-89). */
typedef struct {
    uint32_t*     cells;     /* [rows][pitch] */
    const double* X;         /* [cols] */
    const double* Y;         /* [rows] */
    int32_t rows, cols, pitch, bits_pitch;
    double  lim_x0, lim_x1, lim_y0, lim_y1;
    uint32_t*     occ_bits;  /* [rows][bits_pitch] 1 bit per cell: occupied (2*visited > total).  Kept in
                                step by slam2d_grid_update; after any other write to `cells` call
                                slam2d_map_refresh_bits.  The field build reads only these bits. */
    int32_t wide;            /* 0: cells are uint32 (visited << 16 | total).  1: cells are uint64 [rows][pitch] (visited << 32 | total),
                                `cells` points at them: the format of a map whose counts no longer fit 16 bits (a cell observed
                                more than ~32 000 times; the reference's float64 counts never saturate).  The host promotes a map
                                BEFORE an update could overflow it (it knows an upper bound: every update adds at most 2) */
    int32_t _pad;
} Slam2dMap;

/* Lidar + polar spoke lookup table shared by all particles
 * (Utils/OccupancyGrid.py:22-57). */
typedef struct {
    double unit;             /* unitGridSize */
    double max_range;        /* lidarMaxRange */
    double fov;              /* lidarFOV */
    double wall_half;        /* wallThickness / 2 */
    int32_t beams;           /* numSamplesPerRev */
    int32_t num_spokes;      /* numSpokes */
    int32_t spoke_start;     /* spokesStartIdx */
    int32_t lut_w;           /* W = 2*int(max_range/unit)+1 */
    const double*   lut_xs;  /* [W]  linspace(-R, R, W) */
    /* radByX / radByY / radByR (:47-57): the window cells of every spoke.  The cells of a spoke are
     * ordered by radial band -- SLAM2D_SPOKE_BAND consecutive values of floor(r / unit) -- and row-major
     * inside a band; a beam touches the bands up to the one that holds range + wallThickness/2 */
    const int32_t*  spoke_band;  /* [num_spokes][num_bands + 1] index of each band's first cell (absolute) */
    const uint32_t* spoke_cells; /* [W*W] (window row << 16) | window column */
    const double*   spoke_r;     /* [W*W] radius of that cell */
    int32_t num_bands;           /* floor(max r / unit) / SLAM2D_SPOKE_BAND + 1 */
    int32_t _pad;
    double lut_xs_step;          /* 2*max_range/(W-1) when lut_xs[j] == j*lut_xs_step - max_range bit for bit
                                    (last element max_range), as numpy.linspace computes it; 0: use lut_xs */
} Slam2dLidar;

/* Geometry of one particle's search field at one level, written by
 * slam2d_field_build and read by slam2d_sweep. */
typedef struct {
    double xlo, ylo;         /* xRangeList[0], yRangeList[0] */
    double xhi, yhi;
    double cx, cy;           /* window centre (the pose estimate) */
    double field_min;        /* min of the blurred field (probMin) */
    int32_t fh, fw;          /* field rows / cols actually used (<= fmax) */
    int32_t mx0, mx1;        /* map column window [mx0, mx1) */
    int32_t my0, my1;        /* map row window    [my0, my1) */
    int32_t redo;            /* 1 = the clamp was redone with the measured minimum */
    int32_t min_known;       /* 1 = some tile of the frame is free, so field_min is the analytic floor */
    double field_max;        /* largest value of the (clamped) field over the tiles built at this call */
} Slam2dFrame;

/* Reduction of the 64*R consecutive cube entries one wave of the sweep scored. */
typedef struct {
    double  max;             /* largest score of the chunk (NaN if the chunk holds a NaN) */
    double  sumexp;          /* sum exp(score - max) over the chunk */
    int32_t argmax;          /* flat cube index of max (lowest on ties; first NaN if any) */
    int32_t has_nan;
} Slam2dPartial;

/* One search level (coarse or fine) for P particles: parameters + workspaces.
 * Reference: the two halves of ScanMatcher.matchScan,
 * Utils/ScanMatcher_OGBased.py:53-60 (coarse) and :65-73 (fine). */
typedef struct {
    /* ---- field build (frameSearchSpace + generateProbSearchSpace, :20-45) ---- */
    double step;             /* unitLength: coarseFactor*unit or unit */
    double reach;            /* 1.1*lidarMaxRange + searchRadius (ctor value, both levels) */
    double log_miss;         /* log(missMatchProb) of this level */
    double floor_value;      /* analytic field minimum: blur of an all-free neighbourhood */
    double cost_scale;       /* 2^k: field stores rint(-probSP * cost_scale) as uint32 */
    int32_t blur_radius;     /* int(4*sigma + 0.5) */
    int32_t fmax;            /* max field rows/cols over particles: int(2*reach/step) + 2 */
    int32_t fpitch;          /* row pitch (elements) of field / occ images, >= fmax */
    int32_t wmax;            /* max map-window edge: int(2*reach/unit) + 3 */
    const double* blur_w;    /* [2*blur_radius+1] normalised Gaussian taps */
    /* ---- cube scoring (searchToMatch, :91-151) ---- */
    int32_t ncell;           /* int(searchRadius/step): cube is [ntheta][2*ncell+1][2*ncell+1] */
    int32_t ntheta;
    int32_t fine;            /* 1: priors are zero (fineSearch=True) */
    int32_t kmax;            /* capacity of a per-theta cell list (>= beams) */
    const double* thetas;    /* [ntheta] thetaRange (:114) */
    const double* theta_cos; /* [ntheta] np.cos(thetaRange) */
    const double* theta_sin; /* [ntheta] */
    double rv_coef;          /* -(1 / (2 * moveRSigma**2))  (:101) */
    double tw_coef;          /* -1 / (2 * turnSigma**2)     (:108) */
    double max_move_dev;     /* maxMoveDeviation            (:103) */
    /* ---- workspaces, all device, sized for P particles ---- */
    Slam2dFrame* frames;     /* [P] */
    int32_t* axis_x;         /* [P][wmax] field column of every window map column */
    int32_t* axis_y;         /* [P][wmax] */
    uint8_t* occ;            /* [P][fmax][fpitch] occupied field cells; then [P][2 tmax][fp] block flags (8 x 8 cells), fp = (2 tmax + 17) & ~15 */
    uint32_t* field;         /* [P][fmax][fpitch]  fixed-point cost of probSP (see above) */
    int32_t* cells;          /* [P][ntheta][kmax] unique endpoint cells (patch-corner offsets) */
    int32_t* kcount;         /* [P][ntheta] */
    double*  prior;          /* [P][2][ny][nx]  rv plane, thetaWeight plane */
    double*  cube;           /* [P][ntheta][ny][nx] convTotal */
    Slam2dPartial* partials; /* [P][npartial] per-wave reductions of the cube (sweep -> select) */
    int32_t npartial;        /* capacity per particle: ntheta * ceil(ny*nx / 64) */
    int32_t tmax;            /* ceil(fmax / 16): 16x16-cell tiles per field edge */
    uint8_t* tilemask;       /* [P][2 tmax][fp], fp = (2 tmax + 17) & ~15 (the bytes beyond 2 tmax of a row are never written): the 8 x 8-cell block holds an occupied field cell (stamped like occ; must follow
                                occ contiguously: one memset clears both when occ_gen == 0).  Four flags per blur tile: at a blur
                                radius of 8 a tile's halo is exactly its 4 x 4 blocks, so the triage lists exactly the tiles whose
                                halo holds a wall */
    uint8_t* tilestate;      /* [P][tmax][tmax] PERSISTENT across calls: 0 = the field tile already holds
                                the free-space constant (no rewrite needed), 1 = dirty/unknown.
                                Initialise to 1; set to 1 whenever the field buffer is written by
                                anything other than slam2d_field_build */
    double*  tilemin;        /* [P][tmax][tmax] scratch: per-tile minimum of the blurred field */
    double*  tilemax;        /* [P][tmax][tmax] scratch: per-tile maximum of the field as stored */
    int32_t* tilelist;       /* [P][2][tmax*tmax] scratch: work lists (tiles to blur, tiles to fill) */
    int32_t* tilecount;      /* [P][2] scratch: their lengths */
    uint32_t* tileneed;      /* [P][ceil(ntheta/ep_group)][ceil(tmax*tmax/32)] scratch of slam2d_match: bit t of slice (p, g) set
                                = the poses of that group of angles read field tile t; every call rewrites every word, the field build
                                ORs the slices of a particle (may be NULL when only slam2d_field_build is used) */
    unsigned long long* freerow; /* [P][64] scratch (used when tmax <= 64): bit tx of word ty = field tile (ty, tx)
                                holds the free-space constant; lets the sweep skip loads.  NULL disables */
    int32_t* ring;           /* [1 + ring_cap] scratch of SLAM2D_MATCH_PRUNE_BY_PRIOR: length, then the ascending
                                sweep slots (4 consecutive dx of one dy) that hold a pose inside the prior's ring;
                                NULL disables the option */
    int32_t* prune_state;    /* [P] scratch of the same option: 1 = particle needs the full sweep */
    int32_t ring_cap;        /* capacity of ring (ny * ceil(nx / 4) always suffices) */
    /* ---- branch and bound over 4x4 pose tiles (slam2d_match with bnb != 0; see "Branch and bound" below) ---- */
    uint32_t* gmin;          /* [P][4*tmax][4*tmax] minimum cost of every ALIGNED 4x4 block of field cells; written with the
                                field tiles (k_blur_clamp, the triage's constant fill), so it shares tilestate */
    uint32_t* gmin2;         /* [P][4*tmax][4*tmax] element [Y][X] = min(gmin[Y..Y+1][X..X+1]) >> 12: a lower bound of the
                                cost of every field cell in rows 4Y..4Y+7, columns 4X..4X+7 -- whatever 4x4 window starts in
                                block (Y, X) lies inside.  20-bit values: a sum over <= 2048 cells fits 32 bits */
    int32_t* pcells;         /* [P][ntheta][kmax] the endpoint cells again, as byte offsets into a particle's gmin2 */
    double*  bounds;         /* [P][ntheta][nb][4*ceil(nb/4)] upper bound of the score of every pose tile, nb = ceil(nx/4) */
    double*  tile_pmax;      /* [P][nb][4*ceil(nb/4)] largest rv + thetaWeight of a pose tile (+inf if one is NaN) */
    unsigned long long* bnb_best; /* [P] order-preserving bits of the best exact score of the seed tiles */
    /* two-level bounds (bnb == 2: long cell lists): */
    uint32_t* gmin3d;        /* [P][4][2*tmax][2*tmax] min(gmin[Y..Y+2][X..X+2]) >> 12, decimated by two in four phase planes
                                (plane (Y & 1) * 2 + (X & 1), element [Y >> 1][X >> 1]): bounds a tile of 8 x 8 poses */
    int32_t* p3cells;        /* [P][ntheta][kmax] the endpoint cells as byte offsets into a particle's gmin3d */
    double*  bounds1;        /* [P][ntheta][8][8] upper bounds of the 8 x 8-pose tiles */
    unsigned long long* seed_key; /* [P] best finite 8 x 8-tile bound of the particle, packed with its (theta, tile): the seed */
    double*  beam_xy;        /* [P][beams][2] scratch of slam2d_match (may be NULL): beam endpoints of the pose estimate
                                (covertMeasureToXY, Utils/ScanMatcher_OGBased.py:81-89), evaluated once per particle */
    uint32_t* sync;          /* [P][SLAM2D_SYNC_WORDS] arrival counters of the launches whose last-arriving block of a particle
                                finishes the particle's work (k_exact_select split over several blocks; the tile triage as the tail of the
                                scatter).  ZERO them once after allocation; every launch leaves them zero again.  NULL: those
                                launches fall back to one block per particle / separate launches */
    int32_t bnb;             /* 1: slam2d_match scores this level by branch and bound; 2: with two-level bounds; 3: angle bounds
                                (cubes of <= 5 x 5 poses per angle: needs gmin, gmin2, pcells, bounds [P][ntheta], bnb_best, seed_key) */
    int32_t ep_group;        /* angles per k_endpoints block (>= 1; 0 = 1): tileneed holds ceil(ntheta / ep_group) slices per particle */
    int32_t occ_gen;         /* 0: occ + tilemask are cleared at every build.  1..255: generation stamp -- an
                                occ / tilemask byte means "occupied" only when it equals occ_gen, so nothing is
                                cleared; the caller passes a value unused since the buffers were last zeroed
                                (count 1, 2, ... 254, zero the buffers, start again at 1) */
    uint32_t* arrive;        /* NULL, or a device word the wave that writes a particle's Slam2dMatch adds 1 to behind it (ABI 16):
                                slam2d_groups_match sets it on its own copy of a scan's last level (Slam2dScan.match_seq) */
    uint8_t* gmin2b;         /* NULL, or [P][4*tmax][g2b_pitch] (ABI 17): gmin2 >> 12 in bytes (cost >> 24), written beside gmin2.  With it
                                (bnb == 1) the tile bounds of a particle are taken by ONE block that stages this image in LDS
                                (k_bound_lds: 2 LDS cycles per gather instead of ~20 L1 tag lookups); the bounds are looser by
                                < 2^-7 per cell, so a few more tiles are scored exactly -- results unchanged */
    int32_t g2b_pitch;       /* row pitch of gmin2b in bytes: a multiple of 16, >= 4*tmax, (g2b_pitch / 4) mod 32 in [5, 13] or
                                [19, 27] (the tile rows of a lane group then fall on distinct LDS banks) */
    int32_t reserved0;
    double* theta_umax;      /* NULL, or [P][ntheta] (ABI 17, bnb == 1): the largest tile bound of every angle, written with `bounds` by k_bound /
                                k_bound_lds; k_exact_select then reads the bounds only of angles that can hold a surviving tile
                                (theta_umax >= bnb_best - margin): same list, a fraction of the loads */
} Slam2dLevel;

/* Result of one level for one particle. */
typedef struct {
    double x, y, theta;      /* matchedReading pose (:142-143) */
    double confidence;       /* sum(exp(convTotal)) (:141); 0 when it underflows */
    double log_confidence;   /* log of the same, finite when confidence underflows */
    double best_score;       /* max(convTotal) */
    int32_t pick;            /* flat cube index chosen (argmax or soft-max draw) */
    int32_t argmax;          /* flat cube index of the maximum (lowest index on ties) */
} Slam2dMatch;

/* ------------------------------------------------------------------------- */

int slam2d_abi_version(void);
/* sizeof() of the PODs above as compiled, so the binding can verify its mirror. */
int slam2d_sizeof(const char* type_name);
/* number of visible HIP devices (<= 0: none); never raises. */
int slam2d_device_count(void);

/* frameSearchSpace + generateProbSearchSpace for P particles
 * (Utils/ScanMatcher_OGBased.py:20-45).
 *   d_centre[p*centre_stride + 0..1] = (estimatedX, estimatedY) of particle p.
 * Writes level->frames[p] and level->field[p].  The window must already lie inside
 * each map (the caller performs checkAndExapndOG growth, :27); otherwise
 * SLAM2D_F_WINDOW_OUTSIDE_MAP is raised and the window is clipped. */
int slam2d_field_build(const Slam2dLidar* lidar, const Slam2dLevel* level, const Slam2dMap* d_maps,
                       int32_t P, const double* d_centre, int32_t centre_stride,
                       uint32_t* d_flags, void* stream);

/* searchToMatch for P particles (Utils/ScanMatcher_OGBased.py:91-151).
 *   d_est[p*est_stride + 0..2] = (estimatedX, estimatedY, estimatedTheta)
 *   d_ranges[beams]            = rMeasure (shared by all particles)
 *   est_moving_dist            = estMovingDist
 *   d_psi_cs[P][2]             = (math.cos, math.sin) of estMovingTheta, NaN pair for None
 *                                (NULL: None for every particle; ignored when level->fine)
 *   d_uniform[P]               = one uniform in [0,1) per particle for the soft-max draw
 *                                (matchMax=False, :136-139); NULL selects argmax (matchMax=True)
 * Writes level->cube[p] (convTotal), level->partials[p] and d_out[p]. */
int slam2d_sweep(const Slam2dLidar* lidar, const Slam2dLevel* level, int32_t P,
                 const double* d_est, int32_t est_stride, const double* d_ranges,
                 double est_moving_dist, const double* d_psi_cs, const double* d_uniform,
                 Slam2dMatch* d_out, uint32_t* d_flags, void* stream);

/* matchScan's work at ONE level for P particles: frameSearchSpace + generateProbSearchSpace +
 * searchToMatch (Utils/ScanMatcher_OGBased.py:20-45,91-151) in one call, arguments as for the two
 * calls above (the field is centred on d_est[p][0..1]).  Results (d_out, level->cube,
 * level->partials) are identical to slam2d_field_build followed by slam2d_sweep; the difference is
 * that only the 16x16 field tiles the sweep reads -- those within the search radius of a beam
 * endpoint at some theta -- are blurred (typically 15-30 % of the tiles that hold a wall), so
 * level->field is left incomplete: tiles outside that set keep stale content.  Falls back to the
 * full build for a frame without a single free tile (the field minimum, :43, is then not known
 * without computing everything).
 *
 * options: SLAM2D_MATCH_PRUNE_BY_PRIOR (coarse level only; ignored otherwise).  The reference adds
 * rv = -100 to every pose whose distance from the estimate is not within maxMoveDeviation of the odometry
 * step (:102-103), and no other term of a score is positive.  With this option the poses inside that ring
 * are scored first; when the best of them exceeds -100 + K * max(field) (K = fewest endpoint cells of any
 * theta: the best any pose outside could reach) by SLAM2D_PRUNE_MARGIN, the poses outside cannot be the
 * arg-max and change confidence and soft-max draw by < 1e-12 relative, so they are not scored and
 * level->cube holds only the ring.  Any particle the ring does not settle (best ring score too low, a NaN
 * prior outside the ring) is swept in full by the same call: d_out never differs in arg-max from the
 * unpruned result. */
#define SLAM2D_MATCH_PRUNE_BY_PRIOR 1u
#define SLAM2D_MATCH_PRIOR_READY    2u   /* slam2d_scan_match, slam2d_groups_match[_begin]: d_est / d_psi_cs (the groups: and d_pull) already hold
                                              this scan's prior (slam2d_scan_commit_next, Slam2dScan.h_next_ranges) */
#define SLAM2D_PRUNE_MARGIN 40.0
#define SLAM2D_BNB_MARGIN 30.0
/* Branch and bound (Slam2dLevel.bnb != 0; every level whose cube has 2*ncell+1 in [9, 64] may use it).
 * The cube is cut into tiles of 4 x 4 poses (dy, dx) of one theta.  The poses of a tile read, at endpoint cell k,
 * a 4 x 4 window of the field; that window lies inside the 8 x 8 block of cells gmin2 summarises, so no pose of
 * the tile can score more than
 *     U = -(sum_k gmin2[block of cell k's window] << 12) / cost_scale + max(rv + thetaWeight over the tile).
 * One launch computes U for every tile -- 1/16 of the brute-force gathers, from an image 1/16 the size of the
 * field -- and scores exactly, per theta, the tile with the largest U; the best of those exact scores, M0, is a
 * lower bound of the cube's maximum.  A second launch scores exactly every tile with U >= M0 -
 * SLAM2D_BNB_MARGIN (a fraction of a per cent to a few per cent of the tiles) and selects the pose.  A pose that is
 * skipped scores < max - 30: it cannot be the arg-max, and all skipped poses together (<= 2.4e5 of them at the
 * largest configuration) change confidence and the soft-max draw by < 2.4e5 * exp(-30) = 2.2e-8 relative -- the bar
 * is 1e-5.  A tile holding a NaN prior is always scored (np.argmax returns the first NaN).  level->cube
 * then holds only the scored tiles.
 * bnb == 2 (long cell lists, ~1000 beams): the bounds come in two levels.  Tiles of 8 x 8 poses are bounded first
 * through gmin3d (their 8 x 8 cell window lies inside 3 x 3 aligned 4 x 4 blocks); one exact seed per particle (the
 * best child of the tile with the best 8 x 8 bound, seed_key) gives a first threshold; the 4 x 4 children of the
 * 8 x 8 tiles that reach it get their gmin2 bounds (all other 4 x 4 tiles: -inf), and every theta whose best child
 * still reaches the running maximum is seeded exactly and raises it.  The maximum over all candidate seeds and the
 * tile set scored exactly do not depend on the order in which the waves run.
 * bnb == 3 (angle bounds; cubes of <= 5 x 5 poses per angle, e.g. the fine level behind a coarse factor of 2, with ~1000-cell
 * lists): all poses of one angle read, at endpoint cell k, a window of <= 5 x 5 field cells starting at the patch corner, which
 * lies inside the aligned 8 x 8 block gmin2 summarises there -- so U(theta) = -(sum_k gmin2[cell k] << 12) / cost_scale + the
 * largest prior bounds the angle's whole plane from ONE 4-byte load per cell.  The plane of the particle's best-bound angle is
 * scored exactly first (the seed, M0); planes with U < M0 - SLAM2D_BNB_MARGIN are not scored (level->cube keeps stale content
 * there, their partial reads "nothing"): none of them can hold the arg-max, together they change the confidence by
 * < ntheta * 25 * exp(-30) relative. */
int slam2d_match(const Slam2dLidar* lidar, const Slam2dLevel* level, const Slam2dMap* d_maps, int32_t P,
                 const double* d_est, int32_t est_stride, const double* d_ranges,
                 double est_moving_dist, const double* d_psi_cs, const double* d_uniform,
                 Slam2dMatch* d_out, uint32_t* d_flags, uint32_t options, void* stream);

/* updateOccupancyGrid for P particles (Utils/OccupancyGrid.py:127-159): one wave per beam walks
 * the cells of the beam's spoke up to the measured range (each window cell belongs to exactly
 * one spoke, each spoke to at most one beam, so the counts are updated without atomics).
 *   d_pose[p*pose_stride + 0..2] = matched (x, y, theta)
 *   d_beam_shift: NULL, or [P][beams][6] int32 reproducing the reference's stale-index writes when the map
 *                 grew during a beam (Utils/OccupancyGrid.py:144-152): (dc, dr) low-side shift of the beam's own
 *                 growth, (ac, ar) sum of the low-side shifts of the later beams, (cols, rows) map shape right
 *                 after the beam's growth; a cell with final index (mx, my) is written at
 *                 wrap(mx - dc - ac, cols) + ac (wrap: a negative index + cols, as Python).  Such writes can land
 *                 on cells of other beams (the reference adds both increments): with d_beam_shift the counts are
 *                 updated atomically and occ_bits is NOT maintained -- call slam2d_map_refresh_bits afterwards. */
int slam2d_grid_update(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P,
                       const double* d_pose, int32_t pose_stride, const double* d_ranges,
                       const int32_t* d_beam_shift, uint32_t* d_flags, void* stream);

/* Particle.updateEstimatedPose for P particles (Algorithm/FastSlam.py:77-106): the pose prior of the
 * next scan from the previous matched poses and the raw odometry increment.
 *   d_prev_pose[p*3 + 0..2]  previous matched pose (prevMatchedReading)
 *   raw_theta, prev_raw_theta  currentRawReading['theta'], prevRawReading['theta'] (:78)
 *   has_turn / raw_turn      rawMovingTheta - prevRawMovingTheta when both exist (:94), else has_turn = 0
 *   d_heading[P]             prevMatchedMovingTheta per particle, NaN for None
 * Writes d_est[p*3 + 0..2] (estimatedReading pose) and d_psi_cs[p*2 + 0..1] = (cos, sin) of
 * estMovingTheta (NaN pair for None), the inputs of slam2d_field_build / slam2d_sweep. */
int slam2d_prior(const double* d_prev_pose, double raw_theta, double prev_raw_theta, int32_t has_turn,
                 double raw_turn, const double* d_heading, int32_t P, double* d_est, double* d_psi_cs,
                 void* stream);

/* The bookkeeping after a match (Algorithm/FastSlam.py:108-120,131-135): heading of the matched step
 * (getMovingTheta), previous pose <- matched pose, log-weight += log coarse confidence.
 *   d_fine / d_coarse        Slam2dMatch[P] of the fine / coarse level
 *   d_prev_pose[P][3]        in: previous matched pose, out: this scan's matched pose
 *   d_heading[P]             out: prevMatchedMovingTheta (NaN when the pose did not move)
 *   d_logw[P]                in/out
 *   d_report[P][5]           out (may be NULL): matched x, y, theta, coarse confidence, log of it -- the one
 *                            buffer a caller needs to download per scan */
int slam2d_post_match(const Slam2dMatch* d_fine, const Slam2dMatch* d_coarse, int32_t P, double* d_prev_pose,
                      double* d_heading, double* d_logw, double* d_report, void* stream);

/* Particle.update's `weight *= confidence` in the log domain followed by
 * ParticleFilter.normalizeWeights / weightUnbalanced (Algorithm/FastSlam.py:30-48,135).
 *   d_logw[N]    in/out: log-weights of ALL N particles (after an all-gather when sharded)
 *   d_logconf    log-confidences to add first, element i at d_logconf[i*logconf_stride] (NULL: none)
 *   d_w[N]       out: normalised weights
 *   d_stats[2]   out: [variance = sum (w - 1/N)^2, log of the pre-normalisation weight sum] */
int slam2d_weights_normalize(double* d_logw, const double* d_logconf, int32_t logconf_stride, int32_t N,
                             double* d_w, double* d_stats, void* stream);
/* slam2d_grid_update and slam2d_weights_normalize in ONE launch (the normaliser is one extra block beside the
 * update's: it only reads the log-weights and the log-confidences, which the match wrote).  N = P. */
int slam2d_grid_update_weights(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const double* d_pose,
                               int32_t pose_stride, const double* d_ranges, uint32_t* d_flags, double* d_logw,
                               const double* d_logconf, int32_t logconf_stride, double* d_w, double* d_stats, void* stream);
/* The same for particles sharded over processes: the extra block is slam2d_weights_local (d_part[3] out); the caller's
 * all-gather and slam2d_weights_merge follow. */
int slam2d_grid_update_weights_local(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const double* d_pose,
                                     int32_t pose_stride, const double* d_ranges, uint32_t* d_flags, double* d_logw,
                                     const double* d_logconf, int32_t logconf_stride, double* d_part, void* stream);

/* One scan of Particle.update for P particles in two calls (Algorithm/FastSlam.py:122-135), so that a host loop pays two
 * library calls per scan instead of six:
 *   slam2d_scan_match   = slam2d_prior + slam2d_match(coarse; soft-max draw if d_uniform) + slam2d_match(fine, centred on
 *                         the coarse result, arg-max) -- reads the maps, changes no filter state; `options` as slam2d_match
 *                         (coarse level);
 *   slam2d_scan_commit  = slam2d_post_match + slam2d_grid_update at the matched poses + (d_w != NULL) slam2d_weights_normalize
 *                         over these P particles (in the update's launch), which also moves the scan's fault bits from
 *                         d_flags into d_flag_snapshot[P] with an atomic exchange, so that one asynchronous download
 *                         returns everything the host reads; a bit the update raises after the exchange stays in d_flags
 *                         and is reported with the next call.
 *                         abort_mask (d_w != NULL only; 0: none): SLAM2D_F_* bits that void the scan.  A driver that enqueues
 *                         the commit before it has seen the match's fault bits (the pipelined closed loop) passes
 *                         SLAM2D_F_WINDOW_OUTSIDE_MAP: if the match raised such a bit for ANY particle -- the reference would
 *                         have grown that map first (checkAndExapndOG, Utils/ScanMatcher_OGBased.py:27) -- the whole launch is
 *                         a no-op (no bookkeeping, no map update, no weights; d_flag_snapshot receives the bits, d_flags keeps
 *                         them) and the host grows the maps and runs the scan again.  ABI 13: the voided launch leaves the
 *                         COARSE matched poses in d_report[i][0..2] (the other columns keep their old content): what the host
 *                         needs to grow the maps for the fine windows (:27 at the fine level) without matching again.
 * Arguments as in the calls they bundle. */
int slam2d_scan_match(const Slam2dLidar* lidar, const Slam2dLevel* coarse, const Slam2dLevel* fine, const Slam2dMap* d_maps,
                      int32_t P, const double* d_prev_pose, double raw_theta, double prev_raw_theta, int32_t has_turn,
                      double raw_turn, const double* d_heading, const double* d_ranges, double est_moving_dist,
                      const double* d_uniform, double* d_est, double* d_psi_cs, Slam2dMatch* d_coarse, Slam2dMatch* d_fine,
                      uint32_t* d_flags, uint32_t options, void* stream);
int slam2d_scan_commit(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const Slam2dMatch* d_fine,
                       const Slam2dMatch* d_coarse, double* d_prev_pose, double* d_heading, double* d_logw, double* d_report,
                       const double* d_ranges, uint32_t* d_flags, double* d_w, double* d_stats, uint32_t* d_flag_snapshot,
                       uint32_t abort_mask, void* stream);

/* slam2d_scan_commit that ALSO writes the next scan's pose prior (slam2d_prior's work: Algorithm/FastSlam.py:77-106 needs only
 * this scan's matched poses and headings and the next reading's odometry, which a driver replaying a log -- or one scan behind a
 * live sensor -- holds when it commits): the bookkeeping block computes d_next_est[P][3] / d_next_psi_cs[P][2] behind each
 * particle's bookkeeping, and the next slam2d_scan_match is called with SLAM2D_MATCH_PRIOR_READY -- one launch less per scan.
 * A voided scan (abort_mask) writes no prior; the driver re-issues it without the option.  d_next_est NULL: slam2d_scan_commit. */
int slam2d_scan_commit_next(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const Slam2dMatch* d_fine,
                            const Slam2dMatch* d_coarse, double* d_prev_pose, double* d_heading, double* d_logw, double* d_report,
                            const double* d_ranges, uint32_t* d_flags, double* d_w, double* d_stats, uint32_t* d_flag_snapshot,
                            uint32_t abort_mask, double next_raw_theta, double next_prev_raw_theta, int32_t next_has_turn,
                            double next_raw_turn, double* d_next_est, double* d_next_psi_cs, void* stream);

/* The same normaliser for particles sharded over several processes (one per GPU).  Rank-local
 * half: d_logw[i] += d_logconf[i * logconf_stride], then d_part[3] = [max log w, sum exp(lw - max),
 * sum exp(2 (lw - max))] of this rank's N particles.  The caller all-gathers the 24 bytes of every
 * rank (the only collective of a scan) and hands the [world][3] array to slam2d_weights_merge,
 * which folds it in rank order, writes this rank's normalised weights / log-weights and
 * d_stats[2] = [sum over ALL particles of (w - 1/total)^2, log of the pre-normalisation sum]. */
int slam2d_weights_local(double* d_logw, const double* d_logconf, int32_t logconf_stride, int32_t N,
                         double* d_part, void* stream);
int slam2d_weights_merge(double* d_logw, int32_t N, const double* d_parts, int32_t world,
                         int64_t total_particles, double* d_w, double* d_stats, void* stream);
/* The sharded normaliser of particle GROUPS without events between the groups' streams and the normaliser's (round 5; see
 * Slam2dScan.d_norm_sync with merge == 0): slam2d_norm_gate enqueues a one-wave kernel that waits until the normaliser blocks of all
 * G groups of the scan issued just before have left their partials (they count themselves in d_norm_sync[0]); the caller's
 * all-gather of the partials follows on that stream, then slam2d_weights_merge_publish = slam2d_weights_merge + the word the groups'
 * next normaliser blocks wait for (d_norm_sync[1]).  Call both, in this order, once per scan, behind slam2d_groups_step / _commit. */
int slam2d_norm_gate(uint32_t* d_norm_sync, int32_t G, void* stream);
int slam2d_weights_merge_publish(double* d_logw, int32_t N, const double* d_parts, int32_t world, int64_t total_particles,
                                 double* d_w, double* d_stats, uint32_t* d_norm_sync, void* stream);
/* ... and the closed loop's report (ABI 17): slam2d_weights_merge_publish, after which the same block copies d_pack[0 .. pack_doubles)
 * to the pinned host buffer h_pack and stores report_seq into the pinned host word h_seq (system scope) -- what the merging
 * normaliser block does on one rank (Slam2dScan.h_seq).  A sharded rank's commit is slam2d_groups_commit (merge == 0, h_seq NULL),
 * slam2d_norm_gate, the all-gather of the partials, this call; the host waits with slam2d_host_wait_seq.  The NEXT commit may be
 * issued only after that wait has returned: its groups rewrite d_pack's report rows and fault-bit snapshot.
 * A scan VOIDED on some rank (abort_mask: a search window had left a map): that rank's groups leave the void partial (sum -1)
 * and still arrive, so the enqueued gate, collective and merge run on every rank; the merge then changes no weight, publishes
 * nothing to the groups and reports d_stats = [NaN, -1].  The voiding rank grows its maps and issues the scan again (match, commit,
 * gate, all-gather, this call); every OTHER rank -- its own commit stands, its partials are intact, its groups' next normaliser
 * blocks wait -- answers a voided report with the all-gather and this call alone (no gate).  One all-gather per attempt and rank. */
int slam2d_weights_merge_publish_report(double* d_logw, int32_t N, const double* d_parts, int32_t world, int64_t total_particles,
                                        double* d_w, double* d_stats, uint32_t* d_norm_sync, const double* d_pack, double* h_pack,
                                        int32_t pack_doubles, uint32_t* h_seq, uint32_t report_seq, void* stream);

/* ---- one scan for several particle GROUPS, each on its own HIP stream, issued from C in ONE call ----
 * Particles are independent during a scan (Algorithm/FastSlam.py:25-27); every kernel of the step is latency- or issue-bound
 * at a few dozen particles, so a host driver steps its particles in G groups on G streams and joins them only in the weight
 * normaliser (:30-48).  Issued call by call from Python that costs ~7 us of interpreter + ctypes time per launch (round 3:
 * 0.103 of a 0.131 ms step); these entry points issue the same launches for all groups from C, so the host pays one call per
 * scan (slam2d_groups_step) or two (match / commit: the pipelined closed loop reads the previous scan's report in between).
 *
 * A group is described by HOST pointers to its level descriptors -- either levels of its own or offset views of a larger
 * level (every per-particle pointer advanced by the group's first particle; `tilemask` then no longer follows `occ`
 * contiguously, which is accepted whenever occ_gen != 0: nothing is cleared) -- and by device pointers to its slices of the
 * per-particle arrays. */
typedef struct {
    const Slam2dLevel* coarse;   /* HOST pointer */
    const Slam2dLevel* fine;     /* HOST pointer; NULL: single-level match (the coarse result is the matched pose) */
    const Slam2dMap*   d_maps;   /* the group's P map descriptors */
    int32_t P;
    int32_t est_stride;          /* doubles per row of d_est (>= 3) */
    const double* d_est;         /* [P][est_stride] pose estimates handed in (open loop), or NULL: derive them from d_prev_pose
                                    (slam2d_prior) into d_est_out / d_psi_out -- the closed loop */
    const double* d_psi_cs;      /* [P][2] heading prior with d_est (NULL: None); ignored in the closed loop */
    const double* d_uniform;     /* [P] soft-max draw at the coarse level; NULL: arg-max */
    double* d_prev_pose;         /* closed loop: [P][3] previous matched poses, in/out (slam2d_post_match) */
    double* d_heading;           /* closed loop: [P] prevMatchedMovingTheta, in/out */
    double* d_est_out;           /* closed loop: [P][3] */
    double* d_psi_out;           /* closed loop: [P][2] */
    Slam2dMatch* d_coarse;       /* [P] */
    Slam2dMatch* d_fine;         /* [P]; unused when fine == NULL */
    uint32_t* d_flags;           /* [P] fault bits of this scan (the caller alternates two buffers between scans when groups may
                                    run a scan apart and abort_mask is used: see Slam2dScan.d_abort_flags) */
    double* d_logw;              /* [P] log-weights, a slice of Slam2dScan.d_logw_all */
    double* d_part;              /* [3] out: this group's [max log w, sum, sum of squares] (slam2d_weights_local) */
    double* d_report;            /* closed loop: [P][5] (slam2d_post_match) or NULL */
    uint32_t* d_flag_snapshot;   /* closed loop: [P] the scan's fault bits, moved out of d_flags by the commit, or NULL */
    void* stream;                /* the group's HIP stream */
    void* ev_matched;            /* slam2d_event_create(): recorded behind the group's match (needed with abort_mask) or NULL */
    void* ev_done;               /* recorded behind the group's map update */
    /* ABI 16, with Slam2dScan.h_ranges (closed loop only): */
    double* d_pull;              /* [beams] the group's own device copy of the scan's ranges, written by its prior launch (or by the
                                    previous commit, d_pull_next) and read by every later kernel of the group's scan (d_uniform and
                                    Slam2dScan.d_ranges are then ignored) */
    const double* h_uniform;     /* PINNED HOST [P]: the group's soft-max uniforms of this scan (read there by the selecting kernel,
                                    once per particle), or NULL: arg-max */
    double* d_pull_next;         /* [beams] with Slam2dScan.h_next_ranges: where the commit leaves the NEXT scan's ranges -- that
                                    scan's d_pull (the caller alternates two buffers) */
} Slam2dGroup;

/* What all groups of a scan share. */
typedef struct {
    const double* d_ranges;      /* [beams] */
    double est_moving_dist;
    double raw_theta, prev_raw_theta, raw_turn;   /* closed loop: arguments of slam2d_prior */
    int32_t has_turn;
    uint32_t options;            /* as slam2d_match (coarse level) */
    uint32_t abort_mask;         /* closed loop: as slam2d_scan_commit, decided over d_abort_flags[0 .. n_abort_flags): the fault
                                    bits of ALL groups (their d_flags are slices of that array).  Every group's commit waits for
                                    every group's ev_matched first */
    int32_t n_abort_flags;
    const uint32_t* d_abort_flags;
    void* ev_inputs;             /* NULL, or an event every group's stream waits for before its match (inputs staged elsewhere) */
    /* the weight normaliser over all groups (and, sharded, all ranks): slam2d_weights_merge on norm_stream behind every
     * group's update */
    double* d_logw_all;          /* [n_local] the groups' log-weights, consecutive */
    int32_t n_local;
    int32_t n_parts;             /* partials in d_parts: G x ranks */
    const double* d_parts;       /* [n_parts][3]; one rank: the groups' d_part are its rows */
    int64_t total_particles;
    double* d_w;                 /* [n_local] out */
    double* d_stats;             /* [2] out */
    void* norm_stream;
    void* ev_merged;             /* recorded behind the merge; a group's update waits for the previous scan's */
    int32_t wait_merged;         /* 0: first scan, nothing to wait for */
    int32_t merge;               /* 1: issue the merge here.  0: the caller does (sharded: an all-gather of the partials comes
                                    first); norm_stream is still ordered behind every group's ev_done */
    uint32_t* d_norm_sync;       /* NULL, or 64 device words, ZEROED ONCE, kept for the life of these groups (ABI 15): with merge == 1 and
                                    no abort_mask (ABI 16: or abort_mask with match_seq) the groups' normaliser blocks merge among themselves on the device -- the block
                                    that arrives last merges for all -- and a group's next update waits for that inside its own
                                    normaliser block: no merge launch, no norm_stream, no ev_merged / ev_done / wait_merged (they may
                                    be NULL / 0; ev_done is still recorded when given).  d_w / d_stats / the log-weights are final
                                    once EVERY group's stream has passed the scan.  Needs n_parts == G, the groups' d_part being
                                    rows 0 .. G-1 of d_parts.  With merge == 0 (sharded) the groups only wait and arrive through
                                    the words; the caller follows with slam2d_norm_gate, its collective, slam2d_weights_merge_publish.
                                    Use it for every scan of the groups or for none.  At most 56 groups.  The device-side waits are
                                    bounded: word 59 holds the bound in milliseconds (the caller may set it once, before the first
                                    scan; 0 = 30 s); after it a group's fault word 0 receives SLAM2D_F_SYNC_TIMEOUT (word 62 carries
                                    it from the gate) and, where an abort_mask is in use, the scan is voided so that its report
                                    still reaches the host */
    /* ---- ABI 16: the closed loop without events and without copies (all optional; 0 / NULL = as before) ----
     * Measured in round 5: an event packet between two kernels of a stream costs 3 us, a wait across streams 6-8 us, a copy-engine
     * transfer in front of a kernel ~10 us; a grouped closed-loop scan had ten of them. */
    const double* h_ranges;      /* PINNED HOST [beams], device-visible (hipHostMalloc / torch pin_memory): every group's prior launch
                                    pulls the ranges and Slam2dGroup.h_uniform into Slam2dGroup.d_pull -- no staging copy, no ev_inputs.
                                    The caller alternates two host buffers between scans (a group may still be pulling scan s
                                    while the host stages s + 1) */
    uint32_t match_seq;          /* != 0 (needs d_norm_sync): the number of slam2d_groups_match[_begin] calls with match_seq != 0 since
                                    d_norm_sync was zeroed, this one included, passed AGAIN to the commit of the same scan.  In the
                                    match, the wave that writes a particle's result at the last level counts the particle in (word
                                    61); every group's commit starts with a one-wave gate kernel that waits (bounded) for
                                    n_abort_flags * match_seq arrivals: the abort_mask decision over all groups' fault bits without
                                    ev_matched, and d_norm_sync's device-side merge stays usable with abort_mask (a voided scan has no
                                    normaliser: nobody arrives there, the words stay in step).  n_abort_flags = all groups' particles */
    uint32_t report_seq;         /* with h_seq: the value to publish for this scan (the caller counts its commits) */
    uint32_t* h_seq;             /* PINNED HOST word, or NULL.  (The caller waits for scan s's value before it issues the commit of scan
                                    s + 1: that commit's groups rewrite the report rows and the fault-bit snapshot inside d_pack.)
                                    With d_norm_sync and merge == 1: the block that finishes the scan for
                                    all groups (the merging normaliser block; for a voided scan the last group's block 0, counted in
                                    word 60) copies d_pack[0 .. pack_doubles) to h_pack and then stores report_seq here, system scope.
                                    The host waits with slam2d_host_wait_seq: no download, no event */
    double* h_pack;              /* PINNED HOST [pack_doubles] */
    const double* d_pack;        /* device [pack_doubles]: whatever the caller wants of the scan -- it lays d_report, d_w, d_stats and
                                    d_flag_snapshot out inside this buffer */
    int32_t pack_doubles;
    /* commit only, closed loop with h_ranges: the NEXT scan's prior (slam2d_prior's arguments for it) and ranges ride in this
     * commit's bookkeeping block -- into every group's d_est_out / d_psi_out and d_pull_next -- and the next match is called with
     * SLAM2D_MATCH_PRIOR_READY in `options`: one launch less per group and scan.  A voided scan writes neither: the caller
     * re-issues without the option.  NULL: no such thing */
    const double* h_next_ranges; /* PINNED HOST [beams] */
    double next_raw_theta, next_prev_raw_theta, next_raw_turn;
    int32_t next_has_turn;
} Slam2dScan;

/* match of every group (prior when d_est == NULL, coarse level, fine level) on its stream; records ev_matched */
int slam2d_groups_match(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan);
/* map update at the matched poses + bookkeeping + the group's normaliser partial in ONE launch per group (closed loop:
 * slam2d_scan_commit's work with the abort decided over all groups), then the merge.  */
int slam2d_groups_commit(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan);
/* slam2d_groups_match, but the call returns while worker threads still issue the launches (all groups on workers; the
 * descriptors are copied, the level descriptors they point to must stay untouched until the join).  The next slam2d_groups_* call
 * waits for them first and returns their error; slam2d_groups_join does only that -- call it before synchronising the device or
 * touching anything the match reads.  Without worker threads (slam2d_group_policy) the call is slam2d_groups_match. */
int slam2d_groups_match_begin(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan);
int slam2d_groups_join(void);
/* Host side of Slam2dScan.h_seq: spin until the word has reached `want` (wrap-safe); SLAM2D_E_TIMEOUT after timeout_s seconds. */
int slam2d_host_wait_seq(const uint32_t* h_seq, uint32_t want, double timeout_s);
/* both, group by group (a group's update is enqueued right behind its match): the open-loop step of bench.py */
int slam2d_groups_step(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan);
/* How these three calls issue their groups on this host (decided at the first call; one call runs at a time, a second caller
 * waits).  out4[0] = host cores this process may use (scheduler affinity capped by the cgroup CPU quota), out4[1] = processes
 * assumed to share them (SLAM2D_LOCAL_RANKS, else torchrun's LOCAL_WORLD_SIZE, else 1), out4[2] = 1: one worker thread per group
 * beyond the first issues that group's launches (needs >= 3 cores per process: a worker and the waiting caller both poll;
 * otherwise all groups are issued from the calling thread), out4[3] = 1: the threads poll briefly and yield (< 6 cores per
 * process).  SLAM2D_GROUP_THREADS=0 / 1 forces out4[2]. */
int slam2d_group_policy(int32_t* out4);

/* ParticleFilter.resample's state movement (Algorithm/FastSlam.py:56-61) for maps of
 * identical shape: dst[p] = src[d_index[p]].  Ragged maps are copied by the host with
 * plain device-to-device copies instead. */
int slam2d_gather_maps(const Slam2dMap* d_src, const Slam2dMap* d_dst, const int32_t* d_index,
                       int32_t P, int64_t cells_per_map, void* stream);

/* Recompute occ_bits from cells for the maps d_maps[d_index[0..n)] (d_index NULL: maps 0..n-1).
 * Needed after the caller wrote `cells` itself (upload, growth copy, resample copy). */
int slam2d_map_refresh_bits(const Slam2dMap* d_maps, const int32_t* d_index, int32_t n, void* stream);

/* The picture of particle p's map that the reference's driver draws per scan (Algorithm/FastSlam.py:172-177:
 * `ogMap = visited / total; ogMap = ogMap[y0:y1, x0:x1]; np.flipud(1 - ogMap)`), straight from the device state:
 *   d_out[i][j]    (float64, may be NULL) = 1 - visited / total of map cell (row, x0 + j), row = y0 + i, or -- flipud != 0 --
 *                  y1 - 1 - i; the window [y0, y1) x [x0, x1) must lie inside the map (the caller resolves Python's slice
 *                  clipping, as convertRealXYToMapIdx + slicing do there);
 *   d_out_u8[i][j] (may be NULL) the same as 8-bit grey, rint(255 * value). */
int slam2d_map_image(const Slam2dMap* d_maps, int32_t p, int32_t x0, int32_t x1, int32_t y0, int32_t y1, int32_t flipud,
                     double* d_out, uint8_t* d_out_u8, void* stream);

/* ABI 13: expandOccupancyGrid (Utils/OccupancyGrid.py:59-100) for the count array, a whole growth SEQUENCE at once: every cell
 * of the caller-allocated new map (descriptors in HOST memory; new_map->cells [rows][pitch], new_map->occ_bits [rows][bits_pitch])
 * is written in one pass -- the old map's cell (r, c) at (r + d_row, c + d_col) (d_row / d_col: rows / columns the sequence
 * inserted on the low sides, :70-71; the high sides are appended, :81-82), SLAM2D_INIT_CELL elsewhere, pitch padding included --
 * together with the new map's occupancy bits.  Both maps in the same cell format (`wide`). */
int slam2d_map_grow(const Slam2dMap* old_map, const Slam2dMap* new_map, int32_t d_row, int32_t d_col, void* stream);

/* Fill a map with SLAM2D_INIT_CELL (np.ones / 2*np.ones, Utils/OccupancyGrid.py:13-14). */
int slam2d_map_fill(uint32_t* d_cells, int64_t n, uint32_t value, void* stream);

/* cos / sin of n angles as the endpoint kernel evaluates them on the device (ocml, fp64).  Test aid: the
 * reference evaluates them with NumPy (Utils/ScanMatcher_OGBased.py:87-88); tests measure the distance. */
int slam2d_device_sincos(const double* d_angles, int32_t n, double* d_cos, double* d_sin, void* stream);

/* Per-stage timing for bench.py's roofline figure: when a stage's bit is enabled, every
 * launch of that stage's kernel is bracketed by a HIP event pair on the launch stream
 * (up to `capacity` launches).  slam2d_prof_collect synchronises on the recorded
 * events, returns their summed duration and launch count, and resets the stage. */
#define SLAM2D_STAGE_SWEEP     0   /* k_sweep: pose-cube scoring */
#define SLAM2D_STAGE_BLUR      1   /* k_blur_clamp: separable blur + clamp */
#define SLAM2D_STAGE_SCATTER   2   /* k_occ_scatter: map window -> occupied field cells */
#define SLAM2D_STAGE_UPDATE    3   /* k_grid_update: occupancy-grid update */
#define SLAM2D_STAGE_SELECT    4   /* k_select: argmax / soft-max draw / confidence */
#define SLAM2D_STAGE_ENDPOINTS 5   /* k_endpoints: unique endpoint cells */
#define SLAM2D_STAGE_BOUND     6   /* k_bound: tile upper bounds + seed tiles (branch and bound) */
#define SLAM2D_STAGE_EXACT     7   /* k_exact: exact scores of the surviving tiles + selection */
#define SLAM2D_STAGE_COUNT     8
int  slam2d_prof_enable(uint32_t stage_mask, int32_t capacity);
int  slam2d_prof_collect(int32_t stage, double* total_ms, int32_t* launches);
/* Sample: only every `every`-th launch of an enabled stage gets its event pair (default 1).  An event pair costs the
 * stream ~6 us on each side of the kernel; a timed region that must not carry that on every step samples instead. */
int  slam2d_prof_every(int32_t every);
void slam2d_prof_disable(void);

/* Ordering between streams for a host driver that runs groups of particles on several HIP streams (particles are
 * independent during a scan, Algorithm/FastSlam.py:25-27; only the weight normaliser, :30-48, joins them): events without
 * timing.  slam2d_event_record marks a point of `stream`; slam2d_stream_wait_event makes later work of `stream` wait for it. */
/* n non-blocking streams created in one batch and each used once, so that they sit on distinct hardware queues (up to the
 * runtime's GPU_MAX_HW_QUEUES): the streams of particle groups.  A host driver creates them ONCE per process and reuses them. */
int   slam2d_streams_create(void** out, int32_t n);
void  slam2d_stream_destroy(void* stream);
void* slam2d_event_create(void);
void  slam2d_event_destroy(void* event);
int   slam2d_event_record(void* event, void* stream);
int   slam2d_stream_wait_event(void* stream, void* event);

/* Timing helper for bench.py: HIP events on the caller's stream.
 * slam2d_timer_create -> opaque handle; _start/_stop record events on `stream`;
 * _elapsed_ms synchronises on the stop event and returns milliseconds. */
void* slam2d_timer_create(void);
void  slam2d_timer_destroy(void* timer);
int   slam2d_timer_start(void* timer, void* stream);
int   slam2d_timer_stop(void* timer, void* stream);
int   slam2d_timer_elapsed_ms(void* timer, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* SLAM2D_H */
