// slam2d.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI for the 2-D lidar FastSLAM hot
// path: search-field build, pose-cube sweep, arg-max / soft-max pose selection,
// occupancy-grid update, weight normalisation.  See include/slam2d.h for the
// contract and DESIGN.md for the layout and roofline of each kernel.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread
//   -ffp-contract=off is REQUIRED: every cell index is a trunc/rint of an fp64
//   expression that sits on a lattice point, and the blur reproduces SciPy's
//   operation order; a fused multiply-add anywhere changes results.
//
// All `file:line` citations are relative to the reference repository root.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <sched.h>
#include <stdio.h>

#include "slam2d.h"

#define WAVE 64
#define BLUR_TILE 16                 // field tile edge of the blur (power of two: BLUR_SHIFT)
#define BLUR_SHIFT 4
#define FLAG_SHIFT 3                 // occupancy flags are kept per 8 x 8-cell block (four per tile): with a blur radius of 8 the halo of
//                                      a tile is exactly its 4 x 4 blocks, so "a wall nearby" is decided exactly (3 x 3 tile flags listed
//                                      24-60 % of tiles whose halo then turned out empty)

// ------------------------------------------------------------------------------------
// stage profiling (bench.py): optional HIP-event pairs around selected kernels
// ------------------------------------------------------------------------------------
namespace {

struct StageProf {
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
    int capacity = 0;
    int used = 0;
    int seen = 0;                    // launches of the stage since slam2d_prof_enable
};
StageProf g_prof[SLAM2D_STAGE_COUNT];
unsigned g_prof_mask = 0;
int g_prof_every = 1;                // an event pair around every g_prof_every-th launch of an enabled stage

std::mutex g_prof_mutex;             // (the groups of slam2d_groups_* may be issued from several host threads)
struct StageScope {
    int stage; hipStream_t s; int slot = -1;
    StageScope(int st, hipStream_t stream) : stage(st), s(stream) {
        if (st >= 0 && ((g_prof_mask >> st) & 1u)) {
            std::lock_guard<std::mutex> lk(g_prof_mutex);
            StageProf& p = g_prof[st];
            if (p.seen++ % g_prof_every == 0 && p.used < p.capacity) { slot = p.used++; (void)hipEventRecord(p.start[slot], s); }
        }
    }
    ~StageScope() { if (slot >= 0) (void)hipEventRecord(g_prof[stage].stop[slot], s); }
};

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// Phase timing inside a kernel (development aid, -DSLAM2D_DEBUG_CLOCK): thread 0 of one block stamps the
// 100 MHz wall clock at numbered points; slam2d_debug_clock() copies the stamps out.
#ifdef SLAM2D_DEBUG_CLOCK
__device__ long long g_dbg_clock[64];
#define DBG_CLOCK(i, blk) do { if ((blk) && threadIdx.x == 0) g_dbg_clock[i] = wall_clock64(); } while (0)
#else
#define DBG_CLOCK(i, blk) do { } while (0)
#endif

// ------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long order_bits(double v) {
    // monotone map double -> uint64 (so that atomicMin on the bits is a min on the values)
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unorder_bits(unsigned long long k) {
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
// Block flags of one particle: [2 tmax][flag_pitch] bytes, the pitch a multiple of 16 with at least two unused (never
// written, hence never "set") bytes at the end of every row: the triage copies rows with 16-byte loads and reads the byte left
// of a row's first flag from the padding of the row above.
__host__ __device__ __forceinline__ int flag_pitch(const Slam2dLevel& lv) { return ((lv.tmax << 1) + 17) & ~15; }
__host__ __device__ __forceinline__ size_t flag_bytes(const Slam2dLevel& lv) { return (size_t)(lv.tmax << 1) * flag_pitch(lv); }
// The occupied-field image and the tile flags are not cleared between builds: a cell / tile is
// occupied when its byte equals the build's generation stamp (Slam2dLevel.occ_gen, 1..255).
__device__ __forceinline__ uint8_t occ_stamp(const Slam2dLevel& lv) { return (uint8_t)(lv.occ_gen ? lv.occ_gen : 1); }
// The needed-tile bitmap of slam2d_match: one slice per (particle, group of ep_group angles), written whole by that group's k_endpoints block
// with plain stores (every call overwrites every word: nothing to clear) and OR-ed over theta by the triage.  One shared
// bitmap per particle cost k_endpoints half of its time at 139 angles: every block's atomics met on the same few lines.
__device__ __forceinline__ int need_slices(const Slam2dLevel& lv) {
    const int G = max(lv.ep_group, 1);
    return (lv.ntheta + G - 1) / G;
}
__device__ __forceinline__ uint32_t* need_slice(const Slam2dLevel& lv, const int p, const int grp, const int nneed) {
    return lv.tileneed + ((size_t)p * need_slices(lv) + grp) * nneed;
}
// 0x01 in every byte of v that equals the stamp byte (exact per byte), 0x00 elsewhere
__device__ __forceinline__ uint32_t bytes_equal(const uint32_t v, const uint8_t stamp) {
    const uint32_t t = v ^ (0x01010101u * stamp);
    return (~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu)) >> 7;
}

// The sweep skips loads whose patch holds only the free-space constant when the cell lists are long enough to
// pay for it (k_sweep) and the tile grid fits 64-bit row masks; k_blur_check_redo then builds those masks.
__host__ __device__ __forceinline__ bool sweep_skips(const Slam2dLevel& lv) {
    return lv.freerow != nullptr && lv.tmax <= 64 && lv.kmax >= 512;
}

__device__ __forceinline__ int reflect_index(int i, int n) {
    // SciPy 'reflect' extension: d c b a | a b c d | d c b a
    int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - 1 - m;
}

// sum over the 16 lanes of a DPP row, in every lane of the row: quad butterflies, then the two mirrors (no LDS
// crossbar round trips: a ds_bpermute butterfly of 4 x 64-bit values costs ~1 us in a lone wave)
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(const unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long row16_sum_u64(unsigned long long t) {
    t += dpp_u64<0xB1>(t);           // quad_perm [1, 0, 3, 2]
    t += dpp_u64<0x4E>(t);           // quad_perm [2, 3, 0, 1]
    t += dpp_u64<0x141>(t);          // row_half_mirror
    t += dpp_u64<0x140>(t);          // row_mirror
    return t;
}
// u32 sum over the 16 lanes of a DPP row, in every lane
__device__ __forceinline__ unsigned row16_sum_u32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
    return v;
}
// the same for a double (returned as its bits): used for the 16 exp(score - M) of a tile
__device__ __forceinline__ unsigned long long row16_sum_f64(double v) {
    v += __longlong_as_double((long long)dpp_u64<0xB1>((unsigned long long)__double_as_longlong(v)));
    v += __longlong_as_double((long long)dpp_u64<0x4E>((unsigned long long)__double_as_longlong(v)));
    v += __longlong_as_double((long long)dpp_u64<0x141>((unsigned long long)__double_as_longlong(v)));
    v += __longlong_as_double((long long)dpp_u64<0x140>((unsigned long long)__double_as_longlong(v)));
    return (unsigned long long)__double_as_longlong(v);
}
// wave-wide minimum / maximum of a double without LDS round trips (all 64 lanes active): DPP inside the four rows,
// v_readlane across them
template <int CTRL>
__device__ __forceinline__ double dpp_f64(const double v) {
    return __longlong_as_double((long long)dpp_u64<CTRL>((unsigned long long)__double_as_longlong(v)));
}
__device__ __forceinline__ double readlane_f64(const double v, const int l) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave64_min(double v) {
    v = fmin(v, dpp_f64<0xB1>(v)); v = fmin(v, dpp_f64<0x4E>(v)); v = fmin(v, dpp_f64<0x141>(v)); v = fmin(v, dpp_f64<0x140>(v));
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double wave64_max(double v) {
    v = fmax(v, dpp_f64<0xB1>(v)); v = fmax(v, dpp_f64<0x4E>(v)); v = fmax(v, dpp_f64<0x141>(v)); v = fmax(v, dpp_f64<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Inclusive prefix sums over the 64 lanes of a wave without LDS-crossbar round trips (a __shfl_up scan of a double is twelve
// ds_bpermute): DPP row shifts inside the four rows of 16 lanes (lanes without a source add 0), then the row totals are carried
// across with row_bcast:15 (rows 1 and 3 <- lane 15 of the row below) and row_bcast:31 (rows 2 and 3 <- lane 31).  Fixed order.
__device__ __forceinline__ double wave_scan_incl_f64(double v) {
    v += dpp_f64<0x111>(v);          // row_shr:1
    v += dpp_f64<0x112>(v);          // row_shr:2
    v += dpp_f64<0x114>(v);          // row_shr:4
    v += dpp_f64<0x118>(v);          // row_shr:8
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x142, 0xA, 0xF, false);        // row_bcast:15
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x142, 0xA, 0xF, false);
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x143, 0xC, 0xF, false);        // row_bcast:31
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x143, 0xC, 0xF, false);
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return v;
}
__device__ __forceinline__ int wave_scan_incl_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// (int)(v / step) -- truncation, as astype(int) -- and rint(v / unit) without the fp64 division (~35 issue slots each): v * (1 / d)
// differs from v / d by < 4e-16 relative, so the two agree unless the quotient is within 1e-6 of an integer (a half-integer for
// rint), where the exact division decides.
__device__ __forceinline__ int trunc_div(const double v, const double step, const double inv_step) {
    const double t = v * inv_step;
    if (fabs(t - rint(t)) < 1e-6 || !(fabs(t) < 1e9)) return (int)(v / step);
    return (int)t;
}
__device__ __forceinline__ int rint_div(const double v, const double unit, const double inv_unit) {
    const double t = v * inv_unit;
    double rt = rint(t);
    if (fabs(fabs(t - rt) - 0.5) < 1e-6 || !(fabs(t) < 1e9)) rt = rint(v / unit);
    return (int)rt;
}

// trunc_div for non-negative quotients below 2^31 WITHOUT the division when the quotient sits on an integer n -- which is the rule,
// not the exception: the reference's poses stay on the map's lattice, so (coordinate - window edge) / step is an
// integer give or take an ulp for every column of the window, rounds either way, and trunc_div's guard sends every one of them
// through the fp64 division (~35 issue slots).  (int)(v / step) is n iff the correctly rounded quotient reaches n.  r = v - n * step
// is exact as one fma (v and n * step agree in all but their last bits).  r >= 0: the real quotient is >= n, so is its rounding.
// r < 0: the quotient rounds UP to n iff it lies within half a spacing g of the doubles just below n (g = ulp(n), half that when n is
// a power of two; the tie goes to n, whose mantissa is even): r >= -(g / 2) * step, a product with a power of two, exact.
// tests/test_host_logic.py restates it in Python and checks it against the division on quotients within a few ulp of an integer.
__device__ __forceinline__ int trunc_div_fast(const double v, const double step, const double inv_step) {
    const double t = v * inv_step, n = rint(t);
    if (!(fabs(t - n) < 1e-6) && fabs(t) < 1e9) return (int)t;
    if (!(n >= 1.0 && n < 2147483648.0)) return (int)(v / step);
    const double r = __builtin_fma(-n, step, v);
    const int ni = (int)n, e = 31 - __clz(ni), pow2 = (ni & (ni - 1)) == 0 ? 1 : 0;
    const double ghalf = __longlong_as_double((long long)(1023 + e - 53 - pow2) << 52);
    return r >= -ghalf * step ? ni : ni - 1;
}

// ------------------------------------------------------------------------------------
// K2a  frame geometry                      (Utils/ScanMatcher_OGBased.py:21-28)
// ------------------------------------------------------------------------------------
// Frame geometry of one particle (every thread of k_frame_axis evaluates it redundantly: a few fp64
// operations are cheaper than a kernel boundary).
__device__ __forceinline__ Slam2dFrame make_frame(const Slam2dLidar& lid, const Slam2dLevel& lv, const Slam2dMap& m,
                                                  const double ex, const double ey, uint32_t& f) {
    Slam2dFrame fr;
    fr.cx = ex; fr.cy = ey;
    fr.xlo = ex - lv.reach; fr.xhi = ex + lv.reach;            // :22-23
    fr.ylo = ey - lv.reach; fr.yhi = ey + lv.reach;
    int fw = (int)((fr.xhi - fr.xlo) / lv.step) + 1;            // :24-25
    int fh = (int)((fr.yhi - fr.ylo) / lv.step) + 1;
    f = 0;
    if (fw > lv.fmax) { fw = lv.fmax; f |= SLAM2D_F_FIELD_INDEX; }
    if (fh > lv.fmax) { fh = lv.fmax; f |= SLAM2D_F_FIELD_INDEX; }
    // checkMapToExpand (Utils/OccupancyGrid.py:108-118): the caller grows the map first
    if (fr.xlo < m.lim_x0 || fr.xhi > m.lim_x1 || fr.ylo < m.lim_y0 || fr.yhi > m.lim_y1)
        f |= SLAM2D_F_WINDOW_OUTSIDE_MAP;
    // convertRealXYToMapIdx (Utils/OccupancyGrid.py:102-106) + Python slice clipping (:29-33)
    int mx0 = (int)rint((fr.xlo - m.lim_x0) / lid.unit), mx1 = (int)rint((fr.xhi - m.lim_x0) / lid.unit);
    int my0 = (int)rint((fr.ylo - m.lim_y0) / lid.unit), my1 = (int)rint((fr.yhi - m.lim_y0) / lid.unit);
    mx0 = max(0, min(mx0, m.cols)); mx1 = max(mx0, min(mx1, m.cols));
    my0 = max(0, min(my0, m.rows)); my1 = max(my0, min(my1, m.rows));
    if (mx1 - mx0 > lv.wmax) { mx1 = mx0 + lv.wmax; f |= SLAM2D_F_WINDOW_OUTSIDE_MAP; }
    if (my1 - my0 > lv.wmax) { my1 = my0 + lv.wmax; f |= SLAM2D_F_WINDOW_OUTSIDE_MAP; }
    fr.fh = fh; fr.fw = fw; fr.mx0 = mx0; fr.mx1 = mx1; fr.my0 = my0; fr.my1 = my1;
    fr.field_min = lv.floor_value; fr.redo = 0; fr.min_known = 0;
    fr.field_max = 0.0;
    return fr;
}

// ------------------------------------------------------------------------------------
// K2a+b  frame geometry (:21-28) and the field index of every window column / row (:32-36,173-176)
// ------------------------------------------------------------------------------------
__global__ void k_frame_axis(Slam2dLidar lid, Slam2dLevel lv, const Slam2dMap* __restrict__ maps,
                             const double* __restrict__ centre, int cstride, uint32_t* flags,
                             const double* __restrict__ ranges) {
    const int p = blockIdx.y, axis = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (ranges && axis == 0) {
        // covertMeasureToXY (Utils/ScanMatcher_OGBased.py:81-89) once per particle: k_endpoints' ntheta blocks of the
        // particle would otherwise each evaluate the same cos / sin (a third of its time at 1081 beams x 139 angles)
        const double ex = centre[(size_t)p * cstride], ey = centre[(size_t)p * cstride + 1], eth = centre[(size_t)p * cstride + 2];
        const int B = lid.beams;
        const double a0 = eth - lid.fov / 2, a1 = eth + lid.fov / 2;       // np.linspace(theta - fov/2, theta + fov/2, num=B) (:82-83)
        const double astep = (a1 - a0) / (double)(B - 1);
        for (int b = j; b < B; b += gridDim.x * blockDim.x) {
            const double rg = ranges[b];
            const double a = (b == B - 1) ? a1 : (double)b * astep + a0;
            lv.beam_xy[((size_t)p * B + b) * 2] = ex + cos(a) * rg;                              // :87
            lv.beam_xy[((size_t)p * B + b) * 2 + 1] = ey + sin(a) * rg;                          // :88
        }
    }
    const Slam2dMap m = maps[p];
    uint32_t f;
    const Slam2dFrame fr = make_frame(lid, lv, m, centre[(size_t)p * cstride], centre[(size_t)p * cstride + 1], f);
    if (j == 0 && axis == 0) {
        lv.frames[p] = fr;
        lv.tilecount[2 * p] = 0; lv.tilecount[2 * p + 1] = 0;
        if (lv.bnb) lv.bnb_best[p] = order_bits(-INFINITY);
        if (lv.bnb >= 2) lv.seed_key[p] = 0ull;
        if (f) atomicOr(&flags[p], f);
    }
    const int n = axis == 0 ? fr.mx1 - fr.mx0 : fr.my1 - fr.my0;
    if (j >= n) return;
    const double coord = axis == 0 ? m.X[fr.mx0 + j] : m.Y[fr.my0 + j];
    const double lo = axis == 0 ? fr.xlo : fr.ylo;
    const int dim = axis == 0 ? fr.fw : fr.fh;
    int idx = (int)((coord - lo) / lv.step);       // astype(int): truncation toward zero
    if (idx < 0) idx += dim;                       // Python negative-index wrap (:37)
    if (idx < 0 || idx >= dim) { idx = -1; atomicOr(&flags[p], SLAM2D_F_FIELD_INDEX); }
    (axis == 0 ? lv.axis_x : lv.axis_y)[(size_t)p * lv.wmax + j] = idx;
}

// ------------------------------------------------------------------------------------
// K2c  occupied map cells -> occupied field cells  (Utils/ScanMatcher_OGBased.py:29-37)
//      HBM-bound: streams the map window once (4 B / map cell), byte scatter into occ.
// ------------------------------------------------------------------------------------
// The occupied / free state of every map cell is kept as one bit (Slam2dMap.occ_bits, maintained by
// the update kernel), so the field build reads 1/32 of the bytes the count map holds.
// One thread = one 32-cell word of the map window.
#define SCATTER_ROWS 32
__global__ __launch_bounds__(256) void k_occ_scatter(Slam2dLevel lv, const Slam2dMap* __restrict__ maps) {
    // wave = threadIdx.y walks 8 rows of the map window; lane = one 32-cell word of the row.  The bits
    // of a non-zero word are handled by 32 lanes at once (two words per step), so a horizontal wall --
    // words with up to 32 bits set -- costs one step, not 32 serial iterations of one lane.
    const int p = blockIdx.z;
    const Slam2dFrame fr = lv.frames[p];
    const int nrow = fr.my1 - fr.my0;
    if (fr.mx1 <= fr.mx0 || nrow <= 0) return;
    const int lane = threadIdx.x, wave = threadIdx.y;
    const int w0 = (fr.mx0 >> 5) + blockIdx.x * 64, wlast = (fr.mx1 - 1) >> 5;
    if (w0 > wlast) return;
    const int w = w0 + lane;
    const Slam2dMap m = maps[p];
    constexpr int NR = SCATTER_ROWS / 4;
    const int i0 = blockIdx.y * SCATTER_ROWS + wave;
    // the field column of every window column is staged in LDS and the rows' field indices are loaded with
    // the words: the expansion loop below then has no global load in its dependency chain
    extern __shared__ int32_t ax_s[];
    const int ncol = fr.mx1 - fr.mx0;
    {
        const int32_t* __restrict__ ax = lv.axis_x + (size_t)p * lv.wmax;
        for (int j = wave * 64 + lane; j < ncol; j += 256) ax_s[j] = ax[j];
    }
    uint32_t words[NR];
    int fyk[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int i = i0 + 4 * k;
        words[k] = (i < nrow && w <= wlast) ? m.occ_bits[(size_t)(fr.my0 + i) * m.bits_pitch + w] : 0u;
        fyk[k] = i < nrow ? lv.axis_y[(size_t)p * lv.wmax + i] : -1;
    }
    __syncthreads();
    const int col_base = w << 5;
    uint32_t edge = ~0u;
    if (col_base < fr.mx0) edge &= ~0u << (fr.mx0 - col_base);                    // window edges
    if (col_base + 32 > fr.mx1) edge &= ~0u >> (col_base + 32 - fr.mx1);
    const int fpad = flag_pitch(lv);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.occ + (size_t)p * lv.fmax * lv.fpitch), (short)0, (int)((size_t)lv.fmax * lv.fpitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.tilemask + (size_t)p * flag_bytes(lv)), (short)0, (int)flag_bytes(lv), 0x00020000);   // flags of 8 x 8-cell blocks, [2 tmax][flag_pitch]
    const uint8_t stamp = occ_stamp(lv);
    const int half = lane >> 5, bit = lane & 31;
    const int lbase = (w0 << 5) + bit - fr.mx0;            // the lane's column of word w0, relative to the window
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const uint32_t word = words[k] & edge;
        unsigned long long nz = __ballot(word != 0u);
        if (!nz) continue;
        const int fy = __builtin_amdgcn_readfirstlane(fyk[k]);                     // (the wave's row: uniform)
        if (fy < 0) continue;                                                      // :36-37
        const int so = fy * lv.fpitch, st = (fy >> FLAG_SHIFT) * fpad;
        while (nz) {
            const int sa = __ffsll((long long)nz) - 1;
            nz &= nz - 1;
            int sb = -1;
            if (nz) { sb = __ffsll((long long)nz) - 1; nz &= nz - 1; }
            // (sa, sb are wave-uniform: v_readlane, not a ds_bpermute round trip per step)
            const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)word, sa), wb = sb >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)word, sb) : 0u;
            const uint32_t wsel = half ? wb : wa;
            const int src = half ? sb : sa;
            if ((wsel >> bit) & 1u) {
                const int fx = ax_s[lbase + (src << 5)];
                if (fx >= 0) {                                                     // (buffer stores: see scatter_role)
                    __builtin_amdgcn_raw_buffer_store_b8(stamp, ro, fx, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b8(stamp, rt, fx >> FLAG_SHIFT, st, 0);
                }
            }
        }
    }
}

// k_occ_scatter's work as a block role of k_endpoints' launch (slam2d_match below 512 beams, where no k_frame_axis
// precedes it): the block derives the frame itself and evaluates the field index of its columns / rows inline (:32-37)
// instead of reading the axis tables the frame block of the same launch is still writing.  Block (sbx, sby) of NW waves:
// wave = 8 rows of the map window, lane = one 32-cell word.  Out-of-range indices are skipped here and flagged by the
// frame block.
#ifndef SCATTER_ROLE_NR
#define SCATTER_ROLE_NR 8           // map rows per wave of the scatter role (every block derives the field index of ALL window columns first)
#endif
template <int NW>
__device__ __forceinline__ void scatter_role(const Slam2dLidar& lid, const Slam2dLevel& lv, const Slam2dMap* __restrict__ maps,
                                             const double* __restrict__ centre, const int cstride, const int p,
                                             const int sbx, const int sby, int32_t* ax_s) {
    const Slam2dMap m = maps[p];
    uint32_t fignore;
    const Slam2dFrame fr = make_frame(lid, lv, m, centre[(size_t)p * cstride], centre[(size_t)p * cstride + 1], fignore);
    const int nrow = fr.my1 - fr.my0;
    if (fr.mx1 <= fr.mx0 || nrow <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w0 = (fr.mx0 >> 5) + sbx * 64, wlast = (fr.mx1 - 1) >> 5;
    if (w0 > wlast) return;
    const int w = w0 + lane;
    constexpr int NR = SCATTER_ROLE_NR;
    const int i0 = sby * (NW * NR) + wave;
    if (sby * (NW * NR) >= nrow) return;
    const int ncol = fr.mx1 - fr.mx0;
    const double inv_step = 1.0 / lv.step;
    // (the rows' words and coordinates are requested FIRST: they need the frame only, and their round trip then runs under the
    // column table's)
    uint32_t words[NR];
    int fyk[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int i = i0 + NW * k;
        words[k] = (i < nrow && w <= wlast) ? m.occ_bits[(size_t)(fr.my0 + i) * m.bits_pitch + w] : 0u;
    }
    const double ycoord = (lane < NR && i0 + NW * lane < nrow) ? m.Y[fr.my0 + i0 + NW * lane] : 0.0;
    for (int j = threadIdx.x; j < ncol; j += NW * 64) {
        int idx = trunc_div_fast(m.X[fr.mx0 + j] - fr.xlo, lv.step, inv_step);
        if (idx < 0) idx += fr.fw;                         // Python negative-index wrap (:37)
        ax_s[j] = (idx < 0 || idx >= fr.fw) ? -1 : idx;
    }
    int fy_lane = -1;                                      // lane k < NR evaluates the field row of the wave's k-th map row
    if (lane < NR) {
        const int i = i0 + NW * lane;
        if (i < nrow) {
            int idx = trunc_div_fast(ycoord - fr.ylo, lv.step, inv_step);
            if (idx < 0) idx += fr.fh;
            fy_lane = (idx < 0 || idx >= fr.fh) ? -1 : idx;
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) fyk[k] = __builtin_amdgcn_readlane(fy_lane, k);
    __syncthreads();
    const int col_base = w << 5;
    uint32_t edge = ~0u;
    if (col_base < fr.mx0) edge &= ~0u << (fr.mx0 - col_base);                    // window edges
    if (col_base + 32 > fr.mx1) edge &= ~0u >> (col_base + 32 - fr.mx1);
    // (buffer stores: the row's byte offset rides in the scalar offset, the column in the lane's -- no 64-bit address
    // arithmetic per set bit; the kernel's launch is bound by its vector instructions)
    const int fpad = flag_pitch(lv);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.occ + (size_t)p * lv.fmax * lv.fpitch), (short)0, (int)((size_t)lv.fmax * lv.fpitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.tilemask + (size_t)p * flag_bytes(lv)), (short)0, (int)flag_bytes(lv), 0x00020000);   // flags of 8 x 8-cell blocks, [2 tmax][flag_pitch]
    const uint8_t stamp = occ_stamp(lv);
    const int half = lane >> 5, bit = lane & 31;
    const int lbase = (w0 << 5) + bit - fr.mx0;            // the lane's column of word w0, relative to the window
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const uint32_t word = words[k] & edge;
        unsigned long long nz = __ballot(word != 0u);
        if (!nz) continue;
        const int fy = fyk[k];
        if (fy < 0) continue;                                                      // :36-37
        const int so = fy * lv.fpitch, st = (fy >> FLAG_SHIFT) * fpad;             // (wave-uniform)
        while (nz) {
            const int sa = __ffsll((long long)nz) - 1;
            nz &= nz - 1;
            int sb = -1;
            if (nz) { sb = __ffsll((long long)nz) - 1; nz &= nz - 1; }
            const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)word, sa), wb = sb >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)word, sb) : 0u;
            const uint32_t wsel = half ? wb : wa;
            const int src = half ? sb : sa;
            if ((wsel >> bit) & 1u) {
                const int fx = ax_s[lbase + (src << 5)];
                if (fx >= 0) {
                    __builtin_amdgcn_raw_buffer_store_b8(stamp, ro, fx, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b8(stamp, rt, fx >> FLAG_SHIFT, st, 0);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_refresh_bits(const Slam2dMap* __restrict__ maps, const int32_t* __restrict__ index) {
    // one wave = 64 consecutive cells of a row: coalesced 256-byte read, one ballot, two words out
    const Slam2dMap m = maps[index ? index[blockIdx.y] : blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int groups_per_row = (m.cols + 63) >> 6;
    const long long ngroups = (long long)m.rows * groups_per_row;
    for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < ngroups; g += (long long)gridDim.x * 4) {
        const int row = (int)(g / groups_per_row), c0 = (int)(g - (long long)row * groups_per_row) << 6;
        const int col = c0 + lane;
        bool occ = false;
        if (col < m.cols) {
            if (m.wide) {
                const unsigned long long v = reinterpret_cast<const unsigned long long*>(m.cells)[(size_t)row * m.pitch + col];
                occ = 2ull * (v >> 32) > (v & 0xffffffffull);
            } else {
                const uint32_t v = m.cells[(size_t)row * m.pitch + col];
                occ = 2u * (v >> 16) > (v & 0xffffu);                              // :29-31
            }
        }
        const unsigned long long mask = __ballot(occ);
        if (lane < 2 && (c0 >> 5) + lane < m.bits_pitch)
            m.occ_bits[(size_t)row * m.bits_pitch + (c0 >> 5) + lane] = (uint32_t)(mask >> (32 * lane));
    }
}

// ------------------------------------------------------------------------------------
// K2d  separable Gaussian blur + clamp           (Utils/ScanMatcher_OGBased.py:41-45)
//      fp64, SciPy's symmetric correlate1d operation order, axis 0 then axis 1,
//      'reflect' borders.  The clamp threshold is 0.5 * (field minimum); the minimum
//      is known analytically (floor_value) whenever some cell has an all-free
//      neighbourhood, which k_blur_check_redo verifies from the measured minimum and
//      mode 1 redoes the clamp with the measured minimum otherwise.
// ------------------------------------------------------------------------------------
#define BLUR_THREADS 64              // one wave per tile: no cross-wave barriers
#define SLAM2D_BLUR_BLOCKS_PER_PARTICLE 256
#define BLUR_EXT (BLUR_TILE + 2 * SLAM2D_MAX_BLUR_RADIUS)
// RAD > 0: radius known at compile time, RAD <= 8 (register sliding windows, fully unrolled, LDS sized
// for it); RAD == 0: any radius up to SLAM2D_MAX_BLUR_RADIUS (loops over LDS).
//
// Work avoidance (all bit-exact):
//  * a tile whose halo holds no occupied cell is all "free": every value equals the analytic
//    floor, so it is filled with that constant without arithmetic;
//  * lv.tilestate remembers, across scans, which tiles of the (reused) field buffer already
//    hold that constant: a tile that was free at the previous build and is free now is not
//    touched at all -- the field build then costs HBM traffic only where walls are.
// Tiles are 16x16: walls are thin, so small tiles keep the blurred area close to the area that
// actually differs from the constant (work ~ (T + 2r + 1) * (2 + 2r/T) per unit wall length has its
// minimum near T = 2r).
template <int RAD>
struct BlurLds {
    static constexpr int EXT = RAD > 0 ? BLUR_TILE + 2 * RAD : BLUR_EXT;
    uint8_t occ[EXT][EXT + 4];
    double mid[BLUR_TILE][EXT + 5];        // pitch 37 doubles at RAD = 8: conflict-free b64 row reads
    double red;
};

template <int RAD>
__device__ __forceinline__ void blur_tile(const Slam2dLevel& lv, BlurLds<RAD>& sm, const int p, const Slam2dFrame& fr,
                                          const int tby, const int tbx, const int mode, const bool use_flags) {
    const int fh = fr.fh, fw = fr.fw;
    const int ty0 = tby * BLUR_TILE, tx0 = tbx * BLUR_TILE;
    if (ty0 >= fh || tx0 >= fw) return;
    const int r = RAD > 0 ? RAD : lv.blur_radius;
    const int ext = BLUR_TILE + 2 * r;
    const int tid = threadIdx.x;
    const bool dbg = p == 0 && blockIdx.x == 0 && mode == 0;
    DBG_CLOCK(50, dbg);
    const uint8_t* occ = lv.occ + (size_t)p * lv.fmax * lv.fpitch;
    const uint8_t stamp = occ_stamp(lv);
    uint8_t* state = lv.tilestate + ((size_t)p * lv.tmax + tby) * lv.tmax + tbx;
    // activity: an occupied cell within the halo lies in one of the 8 x 8-cell blocks within ceil(r / 8) blocks of the tile's four
    int any = 1;
    if (use_flags) {
        any = 0;
        const uint8_t* tiles = lv.tilemask + (size_t)p * flag_bytes(lv);
        const int nsy = (fh + 7) >> FLAG_SHIFT, nsx = (fw + 7) >> FLAG_SHIFT, k = (r + 7) >> FLAG_SHIFT, e = 2 + 2 * k;
        if (tid < e * e) {
            const int yy = 2 * tby - k + tid / e, xx = 2 * tbx - k + tid % e;
            if (yy >= 0 && yy < nsy && xx >= 0 && xx < nsx) any = tiles[yy * flag_pitch(lv) + xx] == stamp;
        }
        any = __syncthreads_or(any);
    }
    if (any) {
        // word loads when the halo is interior and 4-byte aligned (r % 4 == 0), bytes + reflect otherwise
        const bool interior = ty0 - r >= 0 && ty0 + BLUR_TILE + r <= fh && tx0 - r >= 0 && tx0 + BLUR_TILE + r <= fw;
        int exact = 0;                 // the tile flags are conservative: re-test on the halo itself
        // every load of the halo is issued before the first use: the image is not cache-resident (one
        // HBM latency instead of one per 64-lane slice)
        if ((r & 3) == 0 && interior) {
            const int wpr = ext >> 2;
            constexpr int NW = RAD > 0 ? ((BLUR_TILE + 2 * RAD) * ((BLUR_TILE + 2 * RAD) / 4) + BLUR_THREADS - 1) / BLUR_THREADS : 1;
            if constexpr (RAD > 0) {
                uint32_t v[NW];
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int idx = min(tid + i * BLUR_THREADS, ext * wpr - 1);
                    const int ly = idx / wpr, lw = idx - ly * wpr;
                    v[i] = *reinterpret_cast<const uint32_t*>(occ + (size_t)(ty0 - r + ly) * lv.fpitch + (tx0 - r) + 4 * lw);
                }
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int idx = tid + i * BLUR_THREADS;
                    if (idx < ext * wpr) {
                        const int ly = idx / wpr, lw = idx - ly * wpr;
                        const uint32_t e = bytes_equal(v[i], stamp);
                        *reinterpret_cast<uint32_t*>(&sm.occ[ly][4 * lw]) = e ^ 0x01010101u;                 // (RAD > 0: the LDS image holds 1 = free)
                        exact |= (e != 0u);
                    }
                }
            } else {
                for (int idx = tid; idx < ext * wpr; idx += BLUR_THREADS) {
                    const int ly = idx / wpr, lw = idx - ly * wpr;
                    const uint32_t v = bytes_equal(*reinterpret_cast<const uint32_t*>(occ + (size_t)(ty0 - r + ly) * lv.fpitch + (tx0 - r) + 4 * lw), stamp);
                    *reinterpret_cast<uint32_t*>(&sm.occ[ly][4 * lw]) = v;
                    exact |= (v != 0u);
                }
            }
        } else if constexpr (RAD > 0) {
            constexpr int EXT_C = BLUR_TILE + 2 * RAD;
            constexpr int NB = (EXT_C * EXT_C + BLUR_THREADS - 1) / BLUR_THREADS;
            uint8_t v[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int idx = min(tid + i * BLUR_THREADS, EXT_C * EXT_C - 1);
                const int ly = idx / EXT_C, lx = idx - ly * EXT_C;
                const int gy = reflect_index(ty0 - r + ly, fh), gx = reflect_index(tx0 - r + lx, fw);
                v[i] = occ[(size_t)gy * lv.fpitch + gx];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int idx = tid + i * BLUR_THREADS;
                if (idx < EXT_C * EXT_C) {
                    const int ly = idx / EXT_C, lx = idx - ly * EXT_C;
                    const uint8_t o = v[i] == stamp;
                    sm.occ[ly][lx] = o ^ 1;
                    exact |= o;
                }
            }
        } else {
            for (int idx = tid; idx < ext * ext; idx += BLUR_THREADS) {
                const int ly = idx / ext, lx = idx - ly * ext;
                const int gy = reflect_index(ty0 - r + ly, fh), gx = reflect_index(tx0 - r + lx, fw);
                const uint8_t o = occ[(size_t)gy * lv.fpitch + gx] == stamp;
                sm.occ[ly][lx] = o;
                exact |= o;
            }
        }
        any = __syncthreads_or(exact);
    }
    DBG_CLOCK(51, dbg);
    const double L = lv.log_miss;
    const double fmin_used = mode == 1 ? fr.field_min : lv.floor_value;
    const double thr = 0.5 * fmin_used;                                            // :44
    uint32_t* field = lv.field + (size_t)p * lv.fmax * lv.fpitch;
    const double* __restrict__ w = lv.blur_w;
    if (!any) {
        const double v = lv.floor_value;
        if (mode == 0 && tid == 0) {
            lv.tilemin[((size_t)p * lv.tmax + tby) * lv.tmax + tbx] = v;
            lv.tilemax[((size_t)p * lv.tmax + tby) * lv.tmax + tbx] = v > thr ? 0.0 : v;
        }
        if (mode == 0 && *state == 0) return;          // already holds the constant: nothing to write
        const uint32_t c = v > thr ? 0u : (uint32_t)rint(-v * lv.cost_scale);
        for (int idx = tid; idx < BLUR_TILE * BLUR_TILE; idx += BLUR_THREADS) {
            const int y = idx / BLUR_TILE, x = idx - y * BLUR_TILE;
            if (tx0 + x < lv.fpitch && ty0 + y < lv.fmax) field[(size_t)(ty0 + y) * lv.fpitch + tx0 + x] = c;
            // the tile's 4 x 4 block minima, from inside this loop (a separate store after it costs the kernel 17
            // VGPRs and a third of its waves)
            if (lv.bnb && ((y | x) & 3) == 0) lv.gmin[((size_t)p * (lv.tmax << 2) + ((ty0 + y) >> 2)) * (lv.tmax << 2) + ((tx0 + x) >> 2)] = c;
        }
        if (tid == 0) *state = mode == 0 ? 0 : 1;
        return;
    }
    if (tid == 0) *state = 1;
    double lmin = INFINITY, lmax = -INFINITY;     // of the blurred values / of the values as stored (after the clamp)
    // SciPy symmetric correlate1d order: out = a[c]*w[c]; for j=-r..-1: out += (a[c+j] + a[c-j]) * w[j]
    if constexpr (RAD > 0) {
        {   // axis-0 pass: lane = (column lx, half of the tile's rows: 8 consecutive outputs)
            constexpr int EXT_C = BLUR_TILE + 2 * RAD;
            static_assert(2 * EXT_C <= BLUR_THREADS, "one wave covers the axis-0 pass only for RAD <= 8");
            const int lx = tid % EXT_C, g = tid / EXT_C;
            if (g < 2) {
                // The inputs are 0.0 (occupied) or L, so a tap's pair sum is 0, L or 2L and its product with the weight is 0, L w or
                // 2 (L w) EXACTLY: with f = 0 / 1 and lw = L * w the tap is fma(f_a + f_b, lw, acc) -- the product inside is exact,
                // the one rounding is the sum's, as in SciPy's acc + (a + b) * w -- two operations instead of three.  (The centre term
                // is selected, not multiplied: 0.0 * lw would be -0.0.)
                double f[8 + 2 * RAD], lw[RAD + 1];
#pragma unroll
                for (int k = 0; k < 8 + 2 * RAD; ++k) f[k] = __hiloint2double((int)((uint32_t)sm.occ[g * 8 + k][lx] * 0x3FF00000u), 0);     // 1 = free -> 1.0 (a
                    // v_cvt_f64_u32 here costs the kernel 17 VGPRs and a third of its waves)
#pragma unroll
                for (int j = 0; j <= RAD; ++j) lw[j] = L * w[j];
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    double acc = f[o + RAD] != 0.0 ? lw[RAD] : 0.0;
#pragma unroll
                    for (int j = -RAD; j < 0; ++j) acc = __builtin_fma(f[o + RAD + j] + f[o + RAD - j], lw[RAD + j], acc);
                    sm.mid[g * 8 + o][lx] = acc;
                }
            }
        }
        __syncthreads();
        DBG_CLOCK(52, dbg);
        {   // axis-1 pass: lane = (row y, 4 consecutive columns)
            const int y = tid >> 2, x0 = (tid & 3) * 4;
            double mrow[4 + 2 * RAD];
#pragma unroll
            for (int k = 0; k < 4 + 2 * RAD; ++k) mrow[k] = sm.mid[y][x0 + k];
            uint32_t cmin = 0xffffffffu;                   // of the lane's 4 stored costs: one row of an aligned 4x4 block
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                double acc = mrow[o + RAD] * w[RAD];
#pragma unroll
                for (int j = -RAD; j < 0; ++j) acc = acc + (mrow[o + RAD + j] + mrow[o + RAD - j]) * w[RAD + j];
                const int gy = ty0 + y, gx = tx0 + x0 + o;
                if (gy < fh && gx < fw) {
                    lmin = fmin(lmin, acc);
                    lmax = fmax(lmax, acc);                // (clamped after the loop: the clamp is monotone)
                    const uint32_t cst = acc > thr ? 0u : (uint32_t)rint(-acc * lv.cost_scale);
                    field[(size_t)gy * lv.fpitch + gx] = cst;
                    cmin = min(cmin, cst);
                }
            }
            if (lmax > thr) lmax = 0.0;
            if (lv.bnb) {                                  // rows y, y^1, y^2, y^3 of the block sit in lanes tid ^ 4, ^ 8
                cmin = min(cmin, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cmin, 0x124, 0xF, 0xF, false));   // row_ror:4
                cmin = min(cmin, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cmin, 0x128, 0xF, 0xF, false));   // row_ror:8
                if ((y & 3) == 0) {
                    const int gp = lv.tmax << 2;
                    lv.gmin[((size_t)p * gp + (ty0 >> 2) + (y >> 2)) * gp + (tx0 >> 2) + (tid & 3)] = cmin;
                }
            }
        }
    } else {
        for (int idx = tid; idx < BLUR_TILE * ext; idx += BLUR_THREADS) {
            const int y = idx / ext, lx = idx - y * ext;
            double acc = (sm.occ[y + r][lx] ? 0.0 : L) * w[r];
            for (int j = -r; j < 0; ++j) {
                const double a = sm.occ[y + r + j][lx] ? 0.0 : L;
                const double b = sm.occ[y + r - j][lx] ? 0.0 : L;
                acc = acc + (a + b) * w[r + j];
            }
            sm.mid[y][lx] = acc;
        }
        __syncthreads();
        for (int idx = tid; idx < BLUR_TILE * BLUR_TILE; idx += BLUR_THREADS) {
            const int y = idx / BLUR_TILE, x = idx - y * BLUR_TILE;
            double acc = sm.mid[y][x + r] * w[r];
            for (int j = -r; j < 0; ++j) acc = acc + (sm.mid[y][x + r + j] + sm.mid[y][x + r - j]) * w[r + j];
            const int gy = ty0 + y, gx = tx0 + x;
            uint32_t cmin = 0xffffffffu;
            if (gy < fh && gx < fw) {
                lmin = fmin(lmin, acc);
                lmax = fmax(lmax, acc > thr ? 0.0 : acc);
                cmin = acc > thr ? 0u : (uint32_t)rint(-acc * lv.cost_scale);
                field[(size_t)gy * lv.fpitch + gx] = cmin;
            }
            if (lv.bnb) {        // this pass covers rows 4i .. 4i+3 (i = idx / 64): lane = (row tid >> 4, column tid & 15)
                cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, 1));
                cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, 2));
                cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, 16));
                cmin = min(cmin, (uint32_t)__shfl_xor((int)cmin, 32));
                if ((tid & 0x33) == 0) {
                    const int gp = lv.tmax << 2;
                    lv.gmin[((size_t)p * gp + (ty0 >> 2) + (idx >> 6)) * gp + (tx0 >> 2) + ((tid & 15) >> 2)] = cmin;
                }
            }
        }
    }
    DBG_CLOCK(53, dbg);
    if (mode == 0) {
        lmin = wave64_min(lmin); lmax = wave64_max(lmax);       // (a ds_bpermute butterfly here was ~1 us of the tile's ~4.5)
        if (tid == 0) {                                // reduced by k_blur_check_redo
            lv.tilemin[((size_t)p * lv.tmax + tby) * lv.tmax + tbx] = lmin;
            lv.tilemax[((size_t)p * lv.tmax + tby) * lv.tmax + tbx] = lmax;
        }
    }
    __syncthreads();                                   // LDS is reused by the block's next tile
    DBG_CLOCK(54, dbg);
}

// A free tile is the free-space constant: one wave, 64 lanes x 16 bytes = the tile's 256 cells (the whole tile, also beyond
// the current frame, so that the tile stays valid when the frame grows by its +-1 jitter), and its 4 x 4 block minima.
__device__ __forceinline__ uint32_t fill_value(const Slam2dLevel& lv) {
    const double v = lv.floor_value;
    return v > 0.5 * v ? 0u : (uint32_t)rint(-v * lv.cost_scale);
}
__device__ __forceinline__ void fill_tile(const Slam2dLevel& lv, int p, int t, int lane, const uint32_t c) {
    uint32_t* field = lv.field + (size_t)p * lv.fmax * lv.fpitch;
    const int ty0 = (t / lv.tmax) * BLUR_TILE, tx0 = (t % lv.tmax) * BLUR_TILE;
    const int y = lane >> 2, x = (lane & 3) * 4;
    if (ty0 + y < lv.fmax && tx0 + x + 3 < lv.fpitch)
        *reinterpret_cast<uint4*>(field + (size_t)(ty0 + y) * lv.fpitch + tx0 + x) = make_uint4(c, c, c, c);
    if (lv.bnb && lane < 16) {                             // the tile's 4 x 4 block minima (branch and bound)
        const int gp = lv.tmax << 2;
        lv.gmin[((size_t)p * gp + (ty0 >> 2) + (lane >> 2)) * gp + (tx0 >> 2) + (lane & 3)] = c;
    }
}

// Tile triage, one 1024-thread block per particle looping over its 16x16 field tiles:
//  * tiles with an occupied cell inside their blur halo -- an occupied 8x8-cell block (level->tilemask) within
//    ceil(radius / 8) blocks of the tile's four -- go to the blur work list;
//  * free tiles whose part of the field buffer does not already hold the free-space constant (tilestate) are
//    filled with it by this same block (no separate launch).
// lazy != 0 (slam2d_match): only tiles the sweep will read (lv.tileneed, marked by k_endpoints) are
// blurred or filled; the others keep their stale content and their tilestate.  This needs the field
// minimum (:43) to be known without computing the whole field: it is the analytic floor as soon as
// ONE tile of the frame is free (every cell of such a tile equals the floor and no blurred value is
// below it).  A frame without any free tile falls back to the full build.
#define TRIAGE_THREADS 1024
__global__ __launch_bounds__(TRIAGE_THREADS) void k_tile_triage(Slam2dLevel lv, int lazy) {
    // tile flags, tile states and the needed-tile bitmap of the particle are staged in LDS with one batch
    // of coalesced loads; everything after that runs out of LDS (the kernel is pure latency otherwise)
    extern __shared__ __attribute__((aligned(16))) uint8_t tri_lds[];     // block flag bits, [ntile4] states, [nneed] words, [ntile] fill list
    __shared__ int base[2];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    DBG_CLOCK(20, p == 0);
    const Slam2dFrame fr = lv.frames[p];
    const int nty = (fr.fh + BLUR_TILE - 1) >> BLUR_SHIFT, ntx = (fr.fw + BLUR_TILE - 1) >> BLUR_SHIFT;
    const int ntile = lv.tmax * lv.tmax, ntile4 = (ntile + 3) & ~3, nneed = (ntile + 31) >> 5;
    const int iters = (ntile + TRIAGE_THREADS - 1) / TRIAGE_THREADS;          // <= 32 (checked by the host)
    // the block flags as BITS in LDS -- one 16-bit word per 16-byte load, rows of `rw` words with a zero word in front, kb zero
    // rows above and below -- and, per tile row, the OR of the 2 + 2 kb flag rows its halo spans: a tile's test is one shift of
    // two words of its row (byte flags and a 4 x 4 window of LDS reads per tile cost three times the kernel's compute)
    const int kb = (lv.blur_radius + 7) >> FLAG_SHIFT;     // blocks the blur radius reaches beyond a tile (<= 2)
    const int fp = flag_pitch(lv), frows = lv.tmax << 1;
    const int rw = (fp >> 4) + 1, rw1 = rw + 1;
    uint16_t* bits_s = reinterpret_cast<uint16_t*>(tri_lds);                     // [kb + frows + kb][rw]
    uint16_t* rows_s = bits_s + (((frows + 2 * kb) * rw + 1) & ~1);              // [tmax][rw + 1] (a zero word at the end)
    const int fbytes = (2 * ((((frows + 2 * kb) * rw + 1) & ~1) + lv.tmax * rw1) + 15) & ~15;
    uint8_t* state_s = tri_lds + fbytes;
    uint32_t* need_s = reinterpret_cast<uint32_t*>(tri_lds + fbytes + ntile4);
    uint16_t* fill_s = reinterpret_cast<uint16_t*>(need_s + nneed);             // [ntile] the fill list
    const uint8_t stamp = occ_stamp(lv);
    uint8_t* state = lv.tilestate + (size_t)p * ntile;
    // every global load of the kernel's first half is issued before the first barrier -- flags, states AND the first words of
    // the needed-tile slices: the kernel is a chain of round trips, this makes it one instead of two
    constexpr int PRE = 4;
    const uint32_t* __restrict__ sl = need_slice(lv, p, 0, nneed);
    const int nsl = lazy ? nneed * need_slices(lv) : 0;
    uint32_t pre[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) { const int i = tid + u * TRIAGE_THREADS; pre[u] = i < nsl ? sl[i] : 0u; }
    {
        const uint4* tq = reinterpret_cast<const uint4*>(lv.tilemask + (size_t)p * flag_bytes(lv));     // (rows of fp bytes, fp % 16 == 0)
        const int qpr = fp >> 4, nq = frows * qpr;
        for (int i = tid; i < nq; i += TRIAGE_THREADS) {
            const uint4 v = tq[i];
            // bytes 0 / 1 -> one bit each: the multiplier moves byte k's bit 0 to bit 24 + k (no two partial products meet)
            const uint32_t m = ((bytes_equal(v.x, stamp) * 0x01020408u) >> 24) | (((bytes_equal(v.y, stamp) * 0x01020408u) >> 24) << 4) |
                               (((bytes_equal(v.z, stamp) * 0x01020408u) >> 24) << 8) | (((bytes_equal(v.w, stamp) * 0x01020408u) >> 24) << 12);
            const int row = i / qpr;
            bits_s[(kb + row) * rw + 1 + (i - row * qpr)] = (uint16_t)m;
        }
        for (int i = tid; i < frows; i += TRIAGE_THREADS) bits_s[(kb + i) * rw] = 0;
        for (int i = tid; i < kb * rw; i += TRIAGE_THREADS) { bits_s[i] = 0; bits_s[(kb + frows) * rw + i] = 0; }
        for (int t = tid; t < ntile; t += TRIAGE_THREADS) state_s[t] = state[t];
        if (lazy) for (int w = tid; w < nneed; w += TRIAGE_THREADS) need_s[w] = 0u;
    }
    if (tid < 2) base[tid] = 0;
    DBG_CLOCK(21, p == 0);
    __syncthreads();
    DBG_CLOCK(22, p == 0);
    if (lazy) {                                            // the particle's needed tiles: OR over the theta slices
#pragma unroll
        for (int u = 0; u < PRE; ++u)
            if (pre[u]) { const int w = (tid + u * TRIAGE_THREADS) % nneed; if ((need_s[w] & pre[u]) != pre[u]) atomicOr(&need_s[w], pre[u]); }
        for (int i = tid + PRE * TRIAGE_THREADS; i < nsl; i += TRIAGE_THREADS) {
            const uint32_t v = sl[i];
            if (v) { const int w = i % nneed; if ((need_s[w] & v) != v) atomicOr(&need_s[w], v); }
        }
    }
    for (int i = tid; i < lv.tmax * rw1; i += TRIAGE_THREADS) {                // the flag rows of a tile row's halo, ORed
        const int ty = i / rw1, j = i - ty * rw1;
        uint32_t v = 0u;
        if (j < rw) for (int k = 0; k < 2 + 2 * kb; ++k) v |= bits_s[(2 * ty + k) * rw + j];
        rows_s[i] = (uint16_t)v;
    }
    __syncthreads();
    DBG_CLOCK(23, p == 0);
    uint32_t liveb = 0u, anyb = 0u;
    int has_free = 0;
    for (int it = 0; it < iters; ++it) {
        const int t = it * TRIAGE_THREADS + tid;
        const int ty = t / lv.tmax, tx = t - ty * lv.tmax;
        if (t < ntile && ty < nty && tx < ntx) {
            // a wall within the blur radius of the tile <=> an occupied 8 x 8 block within kb blocks of the tile's four (flags
            // beyond the frame and the border read 0)
            // flag columns 2 tx - kb .. 2 tx + 1 + kb of the tile row's OR (bit 16 of the row = flag column 0)
            const int bit = 2 * tx - kb + 16;
            const uint16_t* r16 = rows_s + ty * rw1 + (bit >> 4);
            const int any = (((uint32_t)r16[0] | ((uint32_t)r16[1] << 16)) >> (bit & 15)) & ((1u << (2 + 2 * kb)) - 1u);
            liveb |= 1u << it;
            if (any) anyb |= 1u << it; else has_free = 1;
        }
    }
    DBG_CLOCK(24, p == 0);
    has_free = __syncthreads_or(has_free);
    DBG_CLOCK(25, p == 0);
    // one free tile pins the field minimum (:43) to the analytic floor: k_blur_check_redo has nothing to do
    if (tid == 0) lv.frames[p].min_known = has_free;
    const bool everything = !lazy || !has_free;
    int* list = lv.tilelist + (size_t)p * 2 * ntile;
    // list positions from LDS counters (one aggregated atomic per wave and list): the order of a list does not
    // matter -- every tile is blurred / filled independently -- and the loop needs no barrier
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int it = 0; it < iters; ++it) {
        const int t = it * TRIAGE_THREADS + tid;
        const bool live = (liveb >> it) & 1u, any = (anyb >> it) & 1u;
        const bool wanted = live && (everything || ((need_s[t >> 5] >> (t & 31)) & 1u));
        bool to_fill = false;
        // (a free tile's minimum is the floor -- not recorded: k_blur_check_redo reads the per-tile minima only when the frame
        // has no free tile at all, and then every tile goes through the blur, which records its own)
        if (live && !any && wanted && state_s[t] != 0) { to_fill = true; state[t] = 0; }
        const bool mine[2] = {wanted && any, to_fill};
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const unsigned long long mask = __ballot(mine[which]);
            if (!mask) continue;
            int start = 0;
            if (lane == 0) start = atomicAdd(&base[which], __popcll(mask));
            start = __builtin_amdgcn_readfirstlane(start);
            if (mine[which]) {
                list[which * ntile + start + __popcll(mask & below)] = t;
                if (which) fill_s[start + __popcll(mask & below)] = (uint16_t)t;           // (t < 32 * 1024)
            }
        }
    }
    DBG_CLOCK(26, p == 0);
    __syncthreads();
    if (tid < 2) lv.tilecount[2 * p + tid] = base[tid];
    DBG_CLOCK(27, p == 0);
    // fill: one wave per tile, the tiles from the LDS copy of the list
    const int nfill = base[1];
    const uint32_t c = fill_value(lv);
    for (int b = wave; b < nfill; b += TRIAGE_THREADS / 64) fill_tile(lv, p, fill_s[b], lane, c);
}

// Blur of the work list: gridDim.x one-wave blocks per particle walk that particle's active tiles.
#ifndef BLUR_MIN_WAVES
#define BLUR_MIN_WAVES 1
#endif
// bpp > 0 (round 4): a 1-D grid in which block b runs on XCD b % 8 and all blocks of particle p on XCD p % 8, like the sweep's
// and the update's -- x-adjacent tiles share the 128-byte lines of their occupancy halo (8 tiles wide), and with the blocks of a
// particle dealt round the XCDs by blockIdx.x every XCD's L2 fetched its own copy of those lines (round 3: 34 MB of HBM
// traffic per 32-particle launch for 6.9 MB processed, L2 hit rate 56 %).  bpp == 0: the (blocks, P) grid of rounds 1-3.
// tail != 0 (round 4: levels without bounds, where that was all k_blur_check_redo was launched for): the minimum check of a frame
// WITHOUT a free tile -- rare: then every tile was listed and blurred against the analytic floor -- is done by the last of the
// particle's blur blocks to finish (arrival counter lv.sync[p][1]; plain stores + agent release before the ticket, agent acquire
// after it: the slow, always-valid form -- it runs once in a blue moon).  A frame with a free tile, the normal case, costs nothing.
template <int RAD>
__device__ __forceinline__ void blur_check_tail(const Slam2dLevel& lv, BlurLds<RAD>& sm, const int p, Slam2dFrame fr, uint32_t* flags) {
    const int tid = threadIdx.x;
    const int nty = (fr.fh + BLUR_TILE - 1) >> BLUR_SHIFT, ntx = (fr.fw + BLUR_TILE - 1) >> BLUR_SHIFT;
    const double* __restrict__ tm = lv.tilemin + (size_t)p * lv.tmax * lv.tmax;
    double m = INFINITY;
    for (int t = tid; t < nty * ntx; t += BLUR_THREADS) m = fmin(m, tm[(t / ntx) * lv.tmax + (t % ntx)]);
    m = wave64_min(m);
    if (tid == 0) {
        lv.frames[p].field_min = m;
        if (m != lv.floor_value) { lv.frames[p].redo = 1; atomicOr(&flags[p], SLAM2D_F_FLOOR_REDO); }
    }
    if (m == lv.floor_value) return;                   // (wave-uniform)
    fr.field_min = m;
    for (int t = 0; t < nty * ntx; ++t) blur_tile<RAD>(lv, sm, p, fr, t / ntx, t % ntx, 1, true);
}
template <int RAD>
__global__ __launch_bounds__(BLUR_THREADS, BLUR_MIN_WAVES) void k_blur_clamp(Slam2dLevel lv, int P, int bpp, uint32_t* flags, int tail) {
    __shared__ BlurLds<RAD> sm;
    int p = blockIdx.y, first = blockIdx.x, stride = gridDim.x;
    if (bpp > 0) {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        p = (slot / bpp) * 8 + xcd; first = slot % bpp; stride = bpp;
        if (p >= P) return;
    }
    const int n = lv.tilecount[2 * p];
    if (first >= n) return;
    const Slam2dFrame fr = lv.frames[p];
    const int* list = lv.tilelist + (size_t)p * 2 * lv.tmax * lv.tmax;
    for (int b = first; b < n; b += stride) {
        const int t = list[b];
        blur_tile<RAD>(lv, sm, p, fr, t / lv.tmax, t % lv.tmax, 0, false);
    }
    if (tail && !fr.min_known) {                       // (block-uniform; rare)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                       // this wave's per-tile minima leave the XCD's L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned ticket = 0u;
        if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(&lv.sync[p * SLAM2D_SYNC_WORDS + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
        if (ticket != (unsigned)(min(n, stride) - 1)) return;                   // (blocks of this particle that had a tile)
        if (threadIdx.x == 0) __hip_atomic_store(&lv.sync[p * SLAM2D_SYNC_WORDS + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        blur_check_tail<RAD>(lv, sm, p, fr, flags);
    }
}

#ifndef GMIN2_ROWWISE
#define GMIN2_ROWWISE 1
#endif
// gmin2 (branch and bound): element [Y][X] = min(gmin[Y..Y+1][X..X+1]) >> 12.  Blocks beyond the buffer are clamped
// (duplicates only).
__device__ __forceinline__ void gmin2_entry(const Slam2dLevel& lv, const uint32_t* __restrict__ G, uint32_t* __restrict__ G2,
                                            const int p, const int gp, const int Y, const int X) {
    const int Y1 = min(Y + 1, gp - 1), X1 = min(X + 1, gp - 1);
    const uint32_t v = min(min(G[(size_t)Y * gp + X], G[(size_t)Y * gp + X1]), min(G[(size_t)Y1 * gp + X], G[(size_t)Y1 * gp + X1]));
    G2[(size_t)Y * gp + X] = v >> 12;
    if (lv.gmin2b) lv.gmin2b[((size_t)p * gp + Y) * lv.g2b_pitch + X] = (uint8_t)(v >> 24);      // (k_bound_lds)
    if (lv.bnb == 2) {
        // two-level bounds: the 8x8 cell window of an 8x8-pose tile lies inside blocks Y..Y+2 x X..X+2; stored decimated by
        // two in four phase planes, so that the tiles of one row (block stride 2) are contiguous
        const int Y2 = min(Y + 2, gp - 1), X2 = min(X + 2, gp - 1);
        uint32_t v3 = min(v, min(G[(size_t)Y * gp + X2], G[(size_t)Y1 * gp + X2]));
        v3 = min(v3, min(min(G[(size_t)Y2 * gp + X], G[(size_t)Y2 * gp + X1]), G[(size_t)Y2 * gp + X2]));
        const int hp = gp >> 1;
        lv.gmin3d[(size_t)p * gp * gp + ((size_t)(((Y & 1) << 1) | (X & 1)) * hp + (Y >> 1)) * hp + (X >> 1)] = v3 >> 12;
    }
}
// ... over the whole frame: `part` of `parts` thread groups of `nthreads` threads each
__device__ __forceinline__ void gmin2_pass(const Slam2dLevel& lv, const int p, const Slam2dFrame& fr, const int tid,
                                           const int nthreads, const int part, const int parts) {
    const int gp = lv.tmax << 2;
    const int rows = min(gp, (fr.fh >> 2) + 2), cols = min(gp, (fr.fw >> 2) + 2);
    const uint32_t* __restrict__ G = lv.gmin + (size_t)p * gp * gp;
    uint32_t* __restrict__ G2 = lv.gmin2 + (size_t)p * gp * gp;
    for (int idx = part * nthreads + tid; idx < rows * cols; idx += parts * nthreads) {
        const int Y = idx / cols, X = idx - Y * cols;
        gmin2_entry(lv, G, G2, p, gp, Y, X);
    }
}
// ... over the entries that can have changed: a tile written at this build (the blur list and the fill list of the triage)
// changes the minima of its own 4 x 4 blocks, hence entries Y in [4 ty - 1, 4 ty + 3] (from 4 ty - 2 with gmin3d), likewise
// X.  An entry the bounds read covers cells of tiles that are all needed, hence current: either rebuilt now (listed) or
// holding the free-space constant since they were last written -- and every write of a tile has refreshed all the entries it
// touches, so the entry equals what the whole-frame pass would store.  (That pass read and wrote 2 x 166 KB per particle
// and scan for ~150 changed tiles: 21.7 MB of HBM traffic per launch at config 2.)
__device__ __forceinline__ void gmin2_dirty(const Slam2dLevel& lv, const int p, const int tid, const int nthreads,
                                            const int part, const int parts) {
    const int gp = lv.tmax << 2, ntile = lv.tmax * lv.tmax;
    const int nb = lv.tilecount[2 * p], nf = lv.tilecount[2 * p + 1];
    const int* __restrict__ list = lv.tilelist + (size_t)p * 2 * ntile;
    const int E = lv.bnb == 2 ? 6 : 5, per = E * E;
    const uint32_t* __restrict__ G = lv.gmin + (size_t)p * gp * gp;
    uint32_t* __restrict__ G2 = lv.gmin2 + (size_t)p * gp * gp;
    if (lv.bnb != 2 && GMIN2_ROWWISE) {
        // 5 x 5 entries per tile, one thread per (tile, entry row): the two block-minima rows it needs as 1 + 4 + 1 values each (the
        // middle four are one aligned 16-byte load), five entries out as 1 + 4 -- 6 loads and 2 (+ 2 byte-image) stores per row where
        // the entry-per-thread form below issued 20 and 5 (+ 5); same minima, same bits.  Tiles on the image's border go entry by entry.
        typedef unsigned int gu4 __attribute__((ext_vector_type(4)));
        for (int idx = part * nthreads + tid; idx < (nb + nf) * 5; idx += parts * nthreads) {
            const int k = idx / 5, r = idx - k * 5;
            const int t = k < nb ? list[k] : list[ntile + (k - nb)];
            const int ty = t / lv.tmax, tx = t - ty * lv.tmax;
            const int Y = 4 * ty - 1 + r;
            if (Y < 0 || Y >= gp) continue;
            if (tx == 0 || tx >= lv.tmax - 1 || Y + 1 >= gp) {
                for (int c = 0; c < 5; ++c) {
                    const int X = 4 * tx - 1 + c;
                    if (X >= 0 && X < gp) gmin2_entry(lv, G, G2, p, gp, Y, X);
                }
                continue;
            }
            const uint32_t* __restrict__ r0 = G + (size_t)Y * gp + 4 * tx;
            const uint32_t* __restrict__ r1 = r0 + gp;
            const uint32_t am = r0[-1], ap = r0[4], bm = r1[-1], bp = r1[4];
            const gu4 a = *reinterpret_cast<const gu4*>(r0), b = *reinterpret_cast<const gu4*>(r1);
            const uint32_t cm = min(am, bm), c0 = min(a.x, b.x), c1 = min(a.y, b.y), c2 = min(a.z, b.z), c3 = min(a.w, b.w), cp = min(ap, bp);
            const uint32_t em = min(cm, c0), e0 = min(c0, c1), e1 = min(c1, c2), e2 = min(c2, c3), e3 = min(c3, cp);
            uint32_t* __restrict__ o = G2 + (size_t)Y * gp + 4 * tx;
            o[-1] = em >> 12;
            gu4 ov; ov.x = e0 >> 12; ov.y = e1 >> 12; ov.z = e2 >> 12; ov.w = e3 >> 12;
            *reinterpret_cast<gu4*>(o) = ov;
            if (lv.gmin2b) {
                uint8_t* __restrict__ ob = lv.gmin2b + ((size_t)p * gp + Y) * lv.g2b_pitch + 4 * tx;
                ob[-1] = (uint8_t)(em >> 24);
                *reinterpret_cast<uint32_t*>(ob) = (e0 >> 24) | ((e1 >> 24) << 8) | ((e2 >> 24) << 16) | ((e3 >> 24) << 24);
            }
        }
        return;
    }
    for (int idx = part * nthreads + tid; idx < (nb + nf) * per; idx += parts * nthreads) {
        const int k = idx / per, e = idx - k * per;
        const int t = k < nb ? list[k] : list[ntile + (k - nb)];
        const int ty = t / lv.tmax, tx = t - ty * lv.tmax;
        const int Y = 4 * ty - (E - 4) + e / E, X = 4 * tx - (E - 4) + e % E;
        if (Y < 0 || X < 0 || Y >= gp || X >= gp) continue;
        gmin2_entry(lv, G, G2, p, gp, Y, X);
    }
}

// probMin (:43) = minimum over the per-tile minima; when it is not the analytic floor (rare: no cell
// of the field has an all-free neighbourhood) the clamp is redone with it.  Block (p, 0) does that; with branch
// and bound, blocks (p, 0 .. gridDim.y-1) also derive gmin2 from the block minima the blur just wrote -- all of them
// at once when a free tile pins the minimum (frames[p].min_known, set by the triage: no redo can follow), block
// (p, 0) alone after the check otherwise.
template <int RAD>
__global__ __launch_bounds__(256) void k_blur_check_redo(Slam2dLevel lv, uint32_t* flags, int dirty_only) {
    __shared__ BlurLds<RAD> sm;
    __shared__ double red_s[4];
    const int p = blockIdx.x, tid = threadIdx.x;
    Slam2dFrame fr = lv.frames[p];
    if (lv.bnb && fr.min_known) {
        if (dirty_only) gmin2_dirty(lv, p, tid, 256, blockIdx.y, gridDim.y);
        else gmin2_pass(lv, p, fr, tid, 256, blockIdx.y, gridDim.y);
    }
    if (blockIdx.y != 0) return;
    {   // largest value any pose can read: free tiles hold the floor, the blurred ones recorded theirs
        const int n = lv.tilecount[2 * p];
        const int* list = lv.tilelist + (size_t)p * 2 * lv.tmax * lv.tmax;
        double mx = lv.floor_value > 0.5 * lv.floor_value ? 0.0 : lv.floor_value;
        for (int i = tid; i < n; i += 256) mx = fmax(mx, lv.tilemax[(size_t)p * lv.tmax * lv.tmax + list[i]]);
        mx = wave64_max(mx);
        if ((tid & 63) == 0) red_s[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) lv.frames[p].field_max = fmax(fmax(red_s[0], red_s[1]), fmax(red_s[2], red_s[3]));
        __syncthreads();
    }
    // bit tx of freerow[ty]: tile (ty, tx) of the field buffer holds the free-space constant (k_sweep skips
    // loads whose whole patch lies in such tiles)
    if (sweep_skips(lv) && tid < 64) {
        unsigned long long bits = 0ull;
        if (tid < lv.tmax) {
            const uint8_t* st = lv.tilestate + ((size_t)p * lv.tmax + tid) * lv.tmax;
            for (int tx = 0; tx < lv.tmax; ++tx) bits |= (unsigned long long)(st[tx] == 0) << tx;
        }
        lv.freerow[(size_t)p * 64 + tid] = fr.min_known ? bits : 0ull;       // not when the clamp may be redone
    }
    if (fr.min_known) return;                          // a free tile exists: the minimum is the analytic floor
    const int nty = (fr.fh + BLUR_TILE - 1) >> BLUR_SHIFT, ntx = (fr.fw + BLUR_TILE - 1) >> BLUR_SHIFT;
    const double* __restrict__ tm = lv.tilemin + (size_t)p * lv.tmax * lv.tmax;
    double m = INFINITY;
    for (int t = tid; t < nty * ntx; t += 256) m = fmin(m, tm[(t / ntx) * lv.tmax + (t % ntx)]);
    m = wave64_min(m);
    if ((tid & 63) == 0) red_s[tid >> 6] = m;
    __syncthreads();
    m = fmin(fmin(red_s[0], red_s[1]), fmin(red_s[2], red_s[3]));
    if (tid == 0) {
        lv.frames[p].field_min = m;
        if (m != lv.floor_value) { lv.frames[p].redo = 1; atomicOr(&flags[p], SLAM2D_F_FLOOR_REDO); }
    }
    if (m == lv.floor_value) {                         // (block-uniform)
        if (lv.bnb) gmin2_pass(lv, p, fr, tid, 256, 0, 1);
        return;
    }
    if (tid >= BLUR_THREADS) return;                   // the redo itself is one wave's work
    fr.field_min = m;
    for (int t = 0; t < nty * ntx; ++t) blur_tile<RAD>(lv, sm, p, fr, t / ntx, t % ntx, 1, true);
    if (lv.bnb) gmin2_pass(lv, p, fr, tid, BLUR_THREADS, 0, 1);
}

// ------------------------------------------------------------------------------------
// K1b  motion priors rv / thetaWeight              (Utils/ScanMatcher_OGBased.py:97-110)
//      prior[p][0] = rv, prior[p][1] = thetaWeight, each [ny][nx]; computed by the extra block of
//      k_endpoints' extra block of the particle
// ------------------------------------------------------------------------------------
// Pruning by the motion prior (slam2d_match with SLAM2D_MATCH_PRUNE_BY_PRIOR, coarse level only).
// The reference forces rv = -100 wherever the pose's distance from the estimate differs from the
// odometry step by more than maxMoveDeviation (:102-103); field sums and thetaWeight are <= 0, so every
// pose outside that ring scores <= -100.  The sweep first scores the ring alone (a few per cent of the
// cube); if the best ring pose beats -100 + K * (largest value of the field) by SLAM2D_PRUNE_MARGIN = 40, no
// pose outside the ring can be the arg-max and all of them together add less than 1e-12 (relative) to the
// confidence and to the soft-max draw.  Otherwise -- or when a pose outside the ring has a NaN prior, which the reference's argmax would
// return -- the particle is swept in full by a second, normally empty, launch.
// prune_ring(): is pose offset (xv, yv) inside the ring?  The very expression that decides rv below.
__device__ __forceinline__ bool prune_ring(const Slam2dLevel& lv, const int xv, const int yv, const double est_dist) {
    const double mx = (double)xv * lv.step, my = (double)yv * lv.step;
    return !(fabs(sqrt(mx * mx + my * my) - est_dist) > lv.max_move_dev);
}

__device__ __forceinline__ void write_priors(const Slam2dLevel& lv, const int p, const double est_dist,
                                             const double* __restrict__ psi_cs, const int prune) {
    __shared__ int ring_cnt[4];
    __shared__ int ring_base;
    const int nx = 2 * lv.ncell + 1, np_ = nx * nx;
    double* out = lv.prior + (size_t)p * 2 * np_;
    const double cpsi = psi_cs ? psi_cs[2 * p] : NAN;
    const double spsi = psi_cs ? psi_cs[2 * p + 1] : NAN;
    int need_full = 0;
    for (int q = threadIdx.x; q < np_; q += blockDim.x) {
        double rv = 0.0, tw = 0.0;
        if (!lv.fine) {
            const int iy = q / nx, ix = q - iy * nx;
            const int xv = ix - lv.ncell, yv = iy - lv.ncell;
            const double mx = (double)xv * lv.step, my = (double)yv * lv.step;
            const double dist = sqrt(mx * mx + my * my);
            const double dev = dist - est_dist;
            rv = lv.rv_coef * (dev * dev);                                          // :101
            if (fabs(dev) > lv.max_move_dev) rv = -100.0;                           // :102-103
            if (!isnan(cpsi)) {                                                     // :104-108
                double dv = sqrt((double)(xv * xv + yv * yv));
                if (dv == 0.0) dv = 0.0001;
                const double arg = ((double)xv * cpsi + (double)yv * spsi) / dv;
                const double th = acos(arg);            // NaN when |arg| > 1, as np.arccos
                tw = lv.tw_coef * (th * th);
            }
            if (prune && isnan(tw) && !prune_ring(lv, xv, yv, est_dist)) need_full = 1;
        }
        out[q] = rv;
        out[np_ + q] = tw;
    }
    if (lv.bnb && lv.bnb != 3) {
        // largest rv + thetaWeight of every 4x4 pose tile (+inf when one of them is NaN: np.argmax returns the first
        // NaN, so such a tile is always scored); padding tiles get -inf
        __syncthreads();                                   // the planes above were written by this block
        const int nbt = (nx + 3) >> 2, nbq4 = ((nbt + 3) >> 2) << 2;
        double* pm = lv.tile_pmax + (size_t)p * nbt * nbq4;
        for (int t = threadIdx.x; t < nbt * nbq4; t += blockDim.x) {
            const int by = t / nbq4, bx = t - by * nbq4;
            double m = -INFINITY;
            if (bx < nbt)
                for (int r = 0; r < 4 && 4 * by + r < nx; ++r)
                    for (int e = 0; e < 4 && 4 * bx + e < nx; ++e) {
                        const int q = (4 * by + r) * nx + 4 * bx + e;
                        const double v = out[q] + out[np_ + q];
                        m = isnan(v) ? INFINITY : fmax(m, v);
                    }
            pm[t] = m;
        }
    }
    if (!prune) return;
    need_full = __syncthreads_or(need_full);
    if (threadIdx.x == 0) lv.prune_state[p] = need_full;
    if (p != 0) return;
    // the ring as an ascending list of sweep slots (4 consecutive dx of one dy), shared by all particles
    const int nq = (nx + 3) >> 2, nslot = nx * nq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) ring_base = 0;
    __syncthreads();
    for (int s0 = 0; s0 < nslot; s0 += (int)blockDim.x) {
        const int u = s0 + threadIdx.x;
        bool in = false;
        if (u < nslot) {
            const int iy = u / nq, dx = (u - iy * nq) * 4;
            for (int e = 0; e < 4 && dx + e < nx; ++e) in = in || prune_ring(lv, dx + e - lv.ncell, iy - lv.ncell, est_dist);
        }
        const unsigned long long mask = __ballot(in);
        if (lane == 0) ring_cnt[wave] = __popcll(mask);
        __syncthreads();
        int pos = ring_base + __popcll(mask & (lane ? (~0ull >> (64 - lane)) : 0ull));
        for (int w2 = 0; w2 < wave; ++w2) pos += ring_cnt[w2];
        if (in && pos < lv.ring_cap) lv.ring[1 + pos] = u;
        __syncthreads();
        if (threadIdx.x == 0) for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) ring_base += ring_cnt[w2];
        __syncthreads();
    }
    if (threadIdx.x == 0) lv.ring[0] = ring_base <= lv.ring_cap ? ring_base : -1;     // -1: does not fit, sweep in full
}

// k_frame_axis's per-particle work, done by one 256-thread block (the priors block of k_endpoints) when that
// kernel is not launched: frame, flags, axis tables (:21-37,173-176).
__device__ __forceinline__ void frame_duties(const Slam2dLidar& lid, const Slam2dLevel& lv, const Slam2dMap* __restrict__ maps,
                                             const double* __restrict__ centre, const int cstride, uint32_t* flags, const int p) {
    const Slam2dMap m = maps[p];
    uint32_t f;
    const Slam2dFrame fr = make_frame(lid, lv, m, centre[(size_t)p * cstride], centre[(size_t)p * cstride + 1], f);
    if (threadIdx.x == 0) {
        lv.frames[p] = fr;
        lv.tilecount[2 * p] = 0; lv.tilecount[2 * p + 1] = 0;
        if (lv.bnb) lv.bnb_best[p] = order_bits(-INFINITY);
        if (lv.bnb >= 2) lv.seed_key[p] = 0ull;
        if (f) atomicOr(&flags[p], f);
    }
    bool bad = false;
    for (int axis = 0; axis < 2; ++axis) {
        const int n = axis == 0 ? fr.mx1 - fr.mx0 : fr.my1 - fr.my0;
        const double lo = axis == 0 ? fr.xlo : fr.ylo;
        const int dim = axis == 0 ? fr.fw : fr.fh;
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const double coord = axis == 0 ? m.X[fr.mx0 + j] : m.Y[fr.my0 + j];
            int idx = (int)((coord - lo) / lv.step);       // astype(int): truncation toward zero
            if (idx < 0) idx += dim;                       // Python negative-index wrap (:37)
            if (idx < 0 || idx >= dim) { idx = -1; bad = true; }
            (axis == 0 ? lv.axis_x : lv.axis_y)[(size_t)p * lv.wmax + j] = idx;
        }
    }
    if (bad) atomicOr(&flags[p], SLAM2D_F_FIELD_INDEX);
}

// ------------------------------------------------------------------------------------
// K1a  beam endpoints -> unique field cells per theta   (Utils/ScanMatcher_OGBased.py:81-89,
//      117-121,162-176).  One block per (theta, particle); np.unique through an LDS hash set, ordered
//      compaction; plus one block per particle for the motion priors.
//      A cell is stored as the offset of the corner of its (2*ncell+1)^2 patch:
//      (cy - ncell) * fpitch + (cx - ncell).
// ------------------------------------------------------------------------------------
// NT threads: 256, or 192 for scans of up to 192 beams (one beam per thread; 10 blocks per CU instead of 8 hold all the
// 36 x 64 + 128 blocks of config 2 at once -- no second, nearly empty round)
template <int NT>
__global__ __launch_bounds__(NT) void k_endpoints(Slam2dLidar lid, Slam2dLevel lv, const double* __restrict__ est,
                                                   int estride, const double* __restrict__ ranges, uint32_t* flags,
                                                   double est_dist, const double* __restrict__ psi_cs, int mark, int prune,
                                                   int beam_table, const Slam2dMap* __restrict__ maps, int scatter_bx) {
    // maps != NULL: this launch is not preceded by k_frame_axis (slam2d_match below 512 beams): the theta blocks derive
    // the frame fields they need themselves, the priors block also writes frames[p], the axis tables and the flags.
    // np.unique (:120) through an LDS hash set: every beam inserts its cell; of the beams that hit one
    // cell the lowest beam index owns it (atomicMin), so the list keeps beam order -- which is spatially
    // coherent (neighbouring beams hit neighbouring cells) and deterministic.  Scores are exact
    // integer sums, so the order of the list cannot change a result.
    // mark != 0 (slam2d_match): the block also ORs the 16x16 field tiles its patches touch into
    // lv.tileneed (corner bits in LDS, dilated once per block), so that the field build can skip every other tile.
    extern __shared__ __attribute__((aligned(16))) int ep_lds[];     // [hsize] keys, [hsize] owners, [32] (step, wave) counts, tile-marking scratch
    // A block walks lv.ep_group adjacent angles one after the other: the beam endpoints (and, below 512 beams, their
    // cos / sin) are evaluated once per block, and the tile marks of all its angles are dilated and stored once.
    const int grp = blockIdx.x, p = blockIdx.y, tid = threadIdx.x;
    const int G = max(lv.ep_group, 1), ngrp = (lv.ntheta + G - 1) / G;
    if (grp == ngrp) {                                     // the extra block of every particle: motion priors (+ ring)
        write_priors(lv, p, est_dist, psi_cs, prune);
        return;
    }
    if (grp == ngrp + 1) {                                 // a second one (only when maps != NULL): what k_frame_axis would have done
        frame_duties(lid, lv, maps, est, estride, flags, p);
        return;
    }
    if (grp > ngrp + 1) {                                  // (scatter_bx > 0) the occupied-cell scatter, beside the angle blocks
        const int sb = grp - (ngrp + 2);
        scatter_role<NT / 64>(lid, lv, maps, est, estride, p, sb % scatter_bx, sb / scatter_bx, ep_lds);
        return;
    }
    const int it0 = grp * G, it1 = min(it0 + G, lv.ntheta);
    const double ex = est[(size_t)p * estride], ey = est[(size_t)p * estride + 1];
    Slam2dFrame fr;
    if (maps) {
        // no k_frame_axis ahead of this launch: the four frame fields used here, in make_frame's very expressions (:22-25)
        fr.xlo = ex - lv.reach; fr.ylo = ey - lv.reach;
        fr.fw = min((int)(((ex + lv.reach) - fr.xlo) / lv.step) + 1, lv.fmax);
        fr.fh = min((int)(((ey + lv.reach) - fr.ylo) / lv.step) + 1, lv.fmax);
    } else {
        fr = lv.frames[p];
    }
    const int B = lid.beams;
    DBG_CLOCK(40, grp == 0 && p == 0);
    constexpr int NW = NT / 64;
    int hsize = 512;                                       // power of two >= 1.5 * beams (load factor <= 2/3)
    while (hsize < B + (B >> 1)) hsize <<= 1;
    const int hmask = hsize - 1;
    int* hkey = ep_lds;
    int* hown = ep_lds + hsize;
    int* cnt_s = ep_lds + 2 * hsize;
    for (int i = tid; i < hsize; i += NT) { hkey[i] = INT_MAX; hown[i] = INT_MAX; }
    // tile marking scratch: the patch of a beam covers tiles [tx0, tx0 + n + cx] x [ty0, ty0 + n + cy] with cx, cy in {0, 1}
    // (n = (lead + span) / 16), so a beam sets ONE bit -- its corner tile, in the bitmap of its class (cy, cx) -- and the
    // block dilates the four bitmaps afterwards (rows padded to whole words).  Walking the tile rows per beam was half of
    // this kernel's time at 1081 beams.
    const int nneed = (lv.tmax * lv.tmax + 31) >> 5;
    const int wp = (lv.tmax + 31) >> 5;                    // words per tile row
    const bool lds_mark = mark != 0;                       // (tmax^2 <= 28000, check_field_args: the scratch is <= 28 KB)
    uint32_t* corner_s = reinterpret_cast<uint32_t*>(ep_lds + 2 * hsize + 32);       // [2 cy][2 cx][tmax][wp]
    uint32_t* hd_s = corner_s + 4 * lv.tmax * wp;                                   // [2 cy][tmax][wp] after the horizontal pass
    uint32_t* lin_s = hd_s + 2 * lv.tmax * wp;                                      // [nneed] the block's bitmap (bit = ty * tmax + tx)
    uint32_t* const need_g = mark ? need_slice(lv, p, grp, nneed) : nullptr;
    if (lds_mark) for (int i = tid; i < 6 * lv.tmax * wp + nneed; i += NT) corner_s[i] = 0u;
    // branch and bound: gmin2 summarises the aligned 8x8 blocks around the 4x4 windows of the pose tiles, which
    // reach from 3 cells before the patch to 4 * ceil(nx / 4) + 3 cells after its corner
    // (two-level bounds: 8x8-pose tiles, 3x3 blocks: up to 8 * ceil(nx / 8) + 3)
    // mark == 2 (round 4): only the cells the POSES read, (2 ncell + 1)^2 at the patch, as without bounds.  A block minimum taken
    // partly over cells of tiles that were not rebuilt is a minimum over MORE values than the poses can read: never larger than
    // the true one, so the bound it enters stays an upper bound of the scores -- only looser where a pose tile hangs over the
    // window's edge -- and whatever is scored exactly reads needed cells only.
    const bool wide_need = lv.bnb && mark != 2;
    const int lead = wide_need ? 3 : 0;
    const int span = !wide_need ? 2 * lv.ncell : lv.bnb == 2 ? 8 * ((2 * lv.ncell + 8) >> 3) + 3 : 4 * ((2 * lv.ncell + 4) >> 2) + 3;
    const int ntl = (lead + span) >> BLUR_SHIFT;
    const int nc = lv.ncell;
    const int per = NT == 256 ? (B + 255) / 256 : 1;       // beams per thread (5 at 1081 beams -- rounding the beams up to a power of
    //                                                        two first made it 8: three fully masked passes through every loop below,
    //                                                        a third of the kernel's vector instructions), interleaved: beam = q * NT + tid, so that a
    //                                                        wave's loads and stores are contiguous (at 1081 beams the
    //                                                        thread-contiguous mapping cost 32 cache lines per wave-load)
    // np.linspace(theta - fov/2, theta + fov/2, num=B)  (:82-83)
    const double eth = est[(size_t)p * estride + 2];
    const double a0 = eth - lid.fov / 2, a1 = eth + lid.fov / 2;
    const double astep = (a1 - a0) / (double)(B - 1);
    constexpr int QMAX = NT == 256 ? SLAM2D_MAX_BEAMS / 256 : 1;
    int key[QMAX], slot[QMAX];
    double bdx[QMAX], bdy[QMAX];                           // beam endpoint - estimate (:167-168), NaN: no return (:84)
    bool bad = false;
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        bdx[q] = NAN; bdy[q] = NAN;
        const int b = q * NT + tid;
        if (q < per && b < B) {
            const double rg = ranges[b];
            if (rg < lid.max_range) {                                               // :84
                double px, py;
                if (beam_table) {                           // k_frame_axis evaluated :87-88 once for the particle
                    px = lv.beam_xy[((size_t)p * B + b) * 2]; py = lv.beam_xy[((size_t)p * B + b) * 2 + 1];
                } else {
                    const double a = (b == B - 1) ? a1 : (double)b * astep + a0;
                    px = ex + cos(a) * rg; py = ey + sin(a) * rg;                   // :87-88
                }
                bdx[q] = px - ex; bdy[q] = py - ey;
            }
        }
    }
  for (int it = it0; it < it1; ++it) {                     // (indentation kept: the body is one angle's pass)
    const double c = lv.theta_cos[it], s = lv.theta_sin[it];
    if (it > it0) {                                        // the previous angle's compaction is past its barrier: the hash
        for (int i = tid; i < hsize; i += NT) { hkey[i] = INT_MAX; hown[i] = INT_MAX; }      // set is free again
    }
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        key[q] = INT_MAX; slot[q] = 0;
        if (q < per) {
            if (!isnan(bdx[q])) {
                const double dx = bdx[q], dy = bdy[q];
                const double qx = ex + c * dx - s * dy;                             // :169
                const double qy = ey + s * dx + c * dy;                             // :170
                // (a reciprocal multiply with an exact-division guard, as rint_div below, measured no faster: 173.9 vs 166.9 us
                // at 1081 beams -- the kernel is a chain of barriers and LDS round trips, not instruction issue)
                const int cx = (int)((qx - fr.xlo) / lv.step);                      // :174
                const int cy = (int)((qy - fr.ylo) / lv.step);                      // :175
                const int x0 = cx - nc, y0 = cy - nc;
                if (x0 < 0 || y0 < 0 || cx + nc >= fr.fw || cy + nc >= fr.fh) bad = true;
                else key[q] = (y0 << 16) | x0;              // (fmax < 2^15 since fmax * fpitch < 2^29; no integer division to get them back)
            }
        }
    }
    __syncthreads();
    DBG_CLOCK(41, it == 0 && p == 0);
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        if (key[q] == INT_MAX) continue;
        int h = (int)(((unsigned)key[q] * 2654435761u) >> 7) & hmask;
        for (;;) {
            const int prev = atomicCAS(&hkey[h], INT_MAX, key[q]);
            if (prev == INT_MAX || prev == key[q]) break;
            h = (h + 1) & hmask;
        }
        slot[q] = h;
        atomicMin(&hown[h], q * NT + tid);
        if (mark) {                                        // tiles of the (2 nc + 1)^2 patch at (x0, y0)
            const int y0 = key[q] >> 16, x0 = key[q] & 0xFFFF;
            // (a patch clipped by the field's low edge keeps its full extent: at most one tile row / column too many)
            const int xa = max(x0 - lead, 0), ya = max(y0 - lead, 0);
            if (lds_mark) {
                const int tx0 = xa >> BLUR_SHIFT, ty0 = ya >> BLUR_SHIFT;
                const int cx = (((xa & (BLUR_TILE - 1)) + lead + span) >> BLUR_SHIFT) - ntl;
                const int cy = (((ya & (BLUR_TILE - 1)) + lead + span) >> BLUR_SHIFT) - ntl;
                uint32_t* w = corner_s + ((cy * 2 + cx) * lv.tmax + ty0) * wp + (tx0 >> 5);
                const uint32_t m = 1u << (tx0 & 31);
                // neighbouring beams share the corner tile: a plain read first, the atomic only for a new bit
                if (!(*w & m)) atomicOr(w, m);
            }
        }
    }
    __syncthreads();
    DBG_CLOCK(42, it == 0 && p == 0);
    // ordered compaction, beam order = (q, wave, lane): per (q, wave) survivor counts through ballots, one barrier, then
    // every survivor's position = survivors of the steps / waves before + survivors of lower lanes of its own ballot
    const int wv = tid >> 6, lane = tid & 63;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    int* qcnt = cnt_s;                                      // [per][4] counts (per <= 8: 32 ints)
    unsigned owner = 0u;                                    // bit q: this thread's beam of step q owns its cell
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        if (q < per) {
            const bool k = key[q] != INT_MAX && hown[slot[q]] == q * NT + tid;
            owner |= (k ? 1u : 0u) << q;
            const unsigned long long km = __ballot(k);
            if (lane == 0) qcnt[q * NW + wv] = __popcll(km);
        }
    }
    __syncthreads();
    int* out = lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    int* pout = lv.bnb ? lv.pcells + ((size_t)p * lv.ntheta + it) * lv.kmax : nullptr;
    const int gp = lv.tmax << 2;
    int run = 0, pos_end = 0;
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        if (q < per) {
            int before = run;
            for (int w2 = 0; w2 < NW; ++w2) { const int c2 = qcnt[q * NW + w2]; if (w2 < wv) before += c2; run += c2; }
            const unsigned long long km = __ballot((owner >> q) & 1u);
            if ((owner >> q) & 1u) {
                const int pos = before + __popcll(km & below);
                if (pos < lv.kmax) {
                    const int y0 = key[q] >> 16, x0 = key[q] & 0xFFFF;
                    out[pos] = y0 * lv.fpitch + x0;
                    if (pout) {                                // the block of the patch corner, as a byte offset into gmin2
                        const int Y0 = y0 >> 2, X0 = x0 >> 2;
                        pout[pos] = (Y0 * gp + X0) * 4;
                        if (lv.bnb == 2) {                     // ... and into the phase planes of gmin3d
                            const int hp = gp >> 1;
                            lv.p3cells[((size_t)p * lv.ntheta + it) * lv.kmax + pos] = ((((Y0 & 1) << 1 | (X0 & 1)) * hp + (Y0 >> 1)) * hp + (X0 >> 1)) * 4;
                        }
                    }
                }
            }
        }
    }
    pos_end = run;
    const int pos = pos_end;
    DBG_CLOCK(43, it == 0 && p == 0);
    if (tid == NT - 1) {
        int K = pos;                                       // the last thread ends at the total
        if (K > lv.kmax) { K = lv.kmax; bad = true; }
        lv.kcount[p * lv.ntheta + it] = K;
    }
  }
    if (bad) atomicOr(&flags[p], SLAM2D_F_ENDPOINT_OUTSIDE);
    if (!lds_mark) return;                                 // (block-uniform)
    __syncthreads();                                       // the corner bits of the block's angles are all set
    {
        const int nitem = lv.tmax * wp;
        for (int i = tid; i < 2 * nitem; i += NT) {      // horizontal: class cx covers tx0 .. tx0 + ntl + cx
            const int cy = i / nitem, rj = i - cy * nitem, j = rj % wp;
            unsigned long long d = 0ull;
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
                const uint32_t* c = corner_s + (cy * 2 + cx) * nitem + rj;
                unsigned long long v = ((unsigned long long)c[0] << 32) | (j ? c[-1] : 0u);
                for (int k = 0; k < ntl + cx; ++k) v |= v << 1;
                d |= v;
            }
            hd_s[i] = (uint32_t)(d >> 32);
        }
        __syncthreads();
        for (int i = tid; i < nitem; i += NT) {          // vertical: class cy covers ty0 .. ty0 + ntl + cy; then to the particle's bitmap
            const int row = i / wp, j = i - row * wp;
            uint32_t v = 0u;
            for (int dy = 0; dy <= ntl + 1 && dy <= row; ++dy) {
                if (dy <= ntl) v |= hd_s[i - dy * wp];
                v |= hd_s[nitem + i - dy * wp];
            }
            const int left = lv.tmax - 32 * j;             // tiles of this word inside the row
            if (left < 32) v &= (1u << left) - 1u;
            if (v) {
                const int start = row * lv.tmax + 32 * j, sh = start & 31;
                const uint32_t lo = v << sh, hi = sh ? v >> (32 - sh) : 0u;
                if (lo) atomicOr(&lin_s[start >> 5], lo);
                if (hi) atomicOr(&lin_s[(start >> 5) + 1], hi);
            }
        }
        __syncthreads();
        for (int i = tid; i < nneed; i += NT) need_g[i] = lin_s[i];
    }
}

// ------------------------------------------------------------------------------------
// K1c  pose-cube sweep                              (Utils/ScanMatcher_OGBased.py:116-132)
//      score[theta][dy][dx] = sum_k field[cy_k + dy][cx_k + dx] + rv + thetaWeight
//
//      One BLOCK scores 64*RQ slots of 4 consecutive dx of one (particle, theta): lanes run along the
//      flattened (dy, dx/4) plane, so every gather of a wave is a few contiguous row segments of the
//      fixed-point uint32 field; its 4 waves split the unique-cell list.  The list is wave-uniform: it is
//      read with scalar loads and the field through a buffer resource as (per-lane constant VGPR offset) +
//      (scalar cell offset), 16 bytes per lane.  Costs are summed exactly in 64-bit integers (as lo / hi
//      halves), so the sum does not depend on the order and argmax ties in the reference stay ties here.
//      Blocks of one particle are pinned to one XCD (block b runs on XCD b % 8) so that
//      particle's field stays in that XCD's 4 MiB L2 while its cube is swept.
// ------------------------------------------------------------------------------------
struct Best { double v; int i; int nan; };
__device__ __forceinline__ bool better(const Best& a, const Best& b) {
    // np.argmax semantics: first NaN wins; otherwise the largest value, lowest index on ties
    if (a.nan != b.nan) return a.nan > b.nan;
    if (a.nan) return a.i < b.i;
    return a.v > b.v || (a.v == b.v && a.i < b.i);
}
__device__ __forceinline__ Best wave_best(Best me) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        Best ot{__shfl_xor(me.v, o), __shfl_xor(me.i, o), __shfl_xor(me.nan, o)};
        if (better(ot, me)) me = ot;
    }
    return me;
}
__device__ __forceinline__ double wave_sum(double v) {
    // fixed order: DPP butterflies inside the four rows of 16 lanes, then (row 0 + row 1) + (row 2 + row 3) -- deterministic,
    // and no LDS-crossbar round trips (all 64 lanes must be active)
    v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
// wave_best without the 6-step butterfly of three values (24 ds_bpermute) in the common case: DPP maximum, one ballot; only
// when several lanes tie for the maximum, or a NaN is present, the full comparison decides.  Same result as wave_best.  All 64
// lanes must be active.
__device__ __forceinline__ Best wave_best_fast(const Best me) {
    if (__ballot(me.nan != 0)) return wave_best(me);
    const double m = wave64_max(me.i != INT_MAX ? me.v : -INFINITY);
    const unsigned long long eq = __ballot(me.v == m && me.i != INT_MAX);
    if (!eq) return Best{-INFINITY, INT_MAX, 0};
    if ((eq & (eq - 1)) == 0ull) return Best{m, __builtin_amdgcn_readlane(me.i, __ffsll((long long)eq) - 1), 0};
    return wave_best(me);
}
// wave_best for candidates whose index ascends with the lane (so "lowest index" = lowest lane): ballots and readlanes
// instead of a 6-step butterfly of three values.  np.argmax semantics as `better`.  All 64 lanes must be active.
__device__ __forceinline__ Best wave_best_ordered(const Best me) {
    const unsigned long long nanm = __ballot(me.nan != 0);
    if (nanm) {
        const int l = __ffsll((long long)nanm) - 1;
        return Best{NAN, __builtin_amdgcn_readlane(me.i, l), 1};
    }
    const double m = wave64_max(me.v);
    const unsigned long long eq = __ballot(me.v == m && me.i != INT_MAX);
    if (!eq) return Best{-INFINITY, INT_MAX, 0};
    return Best{m, __builtin_amdgcn_readlane(me.i, __ffsll((long long)eq) - 1), 0};
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A "slot" is 4 consecutive dx of one dy row (the last slot of a row is partly padding); one lane
// scores RQ slots, i.e. each gather is one 16-byte buffer load -- a quarter of the vector-memory
// instructions of a dword-per-lane sweep for the same bytes.
// One BLOCK = 64*RQ slots of one (particle, theta); its 4 waves score the SAME slots against
// interleaved quarters of the cell list (wave s takes cells s, s+4, ...), so four waves stream through
// the same field rows at the same time -- one L1 working set per block (measured 124 -> 117 us; the
// gathers run at L1 delivery rate, bypassing L1 costs 1.6x) -- and their exact integer partial sums
// meet in LDS.
#define SWEEP_REST_CHUNKS 4
#ifndef SWEEP_MAIN_CHUNKS
#define SWEEP_MAIN_CHUNKS 1
#endif
#ifndef SWEEP_DEPTH
#define SWEEP_DEPTH 8
#endif
// arg-max, confidence and matched pose of particle p from the sweep's partials (k_select<0> without the soft-max draw: the very
// expressions and summation order), by ONE wave -- the sweep's last-arriving block of the particle (k_sweep, sel_out).  The
// partials were published with write-through stores by other blocks: they are read past this CU's L1.
__device__ __forceinline__ Slam2dPartial load_partial_through(const Slam2dPartial* src) {
    const unsigned long long* u = reinterpret_cast<const unsigned long long*>(src);
    const unsigned long long a = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long c = __hip_atomic_load(u + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Slam2dPartial pt;
    pt.max = __longlong_as_double((long long)a); pt.sumexp = __longlong_as_double((long long)b);
    pt.argmax = (int)(unsigned)c; pt.has_nan = (int)(c >> 32);
    return pt;
}
__device__ __forceinline__ void select_argmax_from_partials(const Slam2dLevel& lv, const int p, const int nW, const double* __restrict__ est,
                                                            const int estride, Slam2dMatch* out) {
    const int lane = threadIdx.x & 63;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const Slam2dPartial* pt0 = lv.partials + (size_t)p * lv.npartial;
    Best me{-INFINITY, INT_MAX, 0};
    for (int w = lane; w < nW; w += WAVE) {
        const Slam2dPartial pt = load_partial_through(pt0 + w);
        Best cand{pt.max, pt.argmax, pt.has_nan};
        if (better(cand, me)) me = cand;
    }
    me = wave_best_fast(me);
    const double M = me.v;
    const int per = (nW + WAVE - 1) / WAVE;
    const int w0 = lane * per, w1 = min(nW, w0 + per);
    double mine = 0.0;
    for (int w = w0; w < w1; ++w) { const Slam2dPartial pt = load_partial_through(pt0 + w); mine += pt.sumexp * exp(pt.max - M); }
    const double incl = wave_scan_incl_f64(mine);
    const double total = readlane_f64(incl, WAVE - 1);
    if (lane == 0) {
        const int pick = me.i;
        Slam2dMatch m;
        const int it = pick / npose, rem = pick - it * npose;
        const int iy = rem / nx, ix = rem - iy * nx;
        const double ex = est[(size_t)p * estride], ey = est[(size_t)p * estride + 1], eth = est[(size_t)p * estride + 2];
        m.x = ex + (double)(ix - lv.ncell) * lv.step;                               // :142-143
        m.y = ey + (double)(iy - lv.ncell) * lv.step;
        m.theta = eth + lv.thetas[it];
        m.confidence = exp(M) * total;                                              // :141
        m.log_confidence = M + log(total);
        m.best_score = M;
        m.pick = pick;
        m.argmax = me.i;
        out[p] = m;
        if (lv.arrive) __hip_atomic_fetch_add(lv.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (Slam2dScan.match_seq)
    }
}

template <int RQ, int mode, bool SKIP>
// (RQ = 1, whole cube, no skip test: 8 waves per SIMD -- 64 VGPRs instead of 66 -- hold the reference's fine level, 30 blocks per
// particle x 64 particles = 1 920 blocks of 4 waves, in ONE round of blocks instead of a full one and a sliver)
__global__ __launch_bounds__(256, (RQ == 1 && mode == 0 && !SKIP) ? 8 : 1) void k_sweep(Slam2dLevel lv, int P, int chunks, int bpp, const double* __restrict__ sel_est = nullptr,
                                               int sel_estride = 0, Slam2dMatch* sel_out = nullptr, int deep = 0) {
    // mode 0: the whole cube.  mode 1 (RQ = 1): only the slots of the prior's ring (lv.ring).  mode 2: the
    // whole cube, for the particles the ring pass could not settle (lv.prune_state[p] != 0) -- see write_priors.
    // SKIP (RQ = 1): a wave-load whose whole patch (its 6-7 pose rows x all dx, at the cell) lies in tiles that
    // hold the free-space constant is not issued; the constant is added once per skipped cell at the end
    // (integer sums: exact).  ~29 % of the loads at config 2.
    __shared__ unsigned long long part_s[3][WAVE * RQ * 4];
    __shared__ unsigned long long free_s[64];
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int p = (slot / bpp) * 8 + xcd;
    if (p >= P) return;
    if (mode == 2 && lv.prune_state[p] == 0) return;
    if (mode == 1 && lv.prune_state[p] != 0) return;
    if constexpr (SKIP) {
        if (threadIdx.x < 64) free_s[threadIdx.x] = lv.freerow[(size_t)p * 64 + threadIdx.x];
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // modes 0 / 1: one block per (theta, chunk).  mode 2: one block per (theta, SWEEP_REST_CHUNKS chunks), so that
    // the launch -- empty for every settled particle -- is a quarter of the blocks
    const int w = slot % bpp;
    constexpr int CPB = mode == 2 ? SWEEP_REST_CHUNKS : (mode == 0 ? SWEEP_MAIN_CHUNKS : 1);      // chunks per block
    const int groups = (chunks + CPB - 1) / CPB;
    const int it = w / groups;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const int nq = (nx + 3) >> 2, nslot = nx * nq;
    const uint32_t* __restrict__ F = lv.field + (size_t)p * lv.fmax * lv.fpitch;
    const int* __restrict__ cl = lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    const int K = lv.kcount[p * lv.ntheta + it];
    const int nring = mode == 1 ? lv.ring[0] : 0;
    if (mode == 1 && (w - it * groups) * WAVE >= nring) return;     // also nring = -1: the ring did not fit
    // Buffer addressing (SRSRC): address = field base + per-lane VGPR byte offset (constant
    // over k) + wave-uniform SGPR byte offset (the cell) -- no vector address arithmetic in
    // the loop, and out-of-range offsets read 0 instead of faulting.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)F, (short)0, (int)((size_t)lv.fmax * lv.fpitch * sizeof(uint32_t)), 0x00020000);
    const int dbg0 = lv.fine ? 44 : 32;
    auto sweep_chunk = [&](const int ch) {
    DBG_CLOCK(dbg0, p == 0 && it == 0 && ch == 0);
    const int u0 = ch * (WAVE * RQ) + lane;
    int off[RQ], q0[RQ], nv[RQ];          // byte offset, first pose index, valid poses (0..4) of each slot
    unsigned lo[RQ][4], hi[RQ][4];        // exact 64-bit integer sums as 32-bit halves
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
        int u = u0 + r * WAVE;
        if (mode == 1) u = u < nring ? lv.ring[1 + u] : nslot;
        const int uu = u < nslot ? u : 0;
        const int iy = uu / nq, dx = (uu - iy * nq) * 4;
        off[r] = (iy * lv.fpitch + dx) * 4;
        q0[r] = iy * nx + dx;
        nv[r] = u < nslot ? min(4, nx - dx) : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[r][e] = 0u; hi[r][e] = 0u; }
    }
    unsigned nfree = 0u;
    if constexpr (SKIP) {
        // This wave's cells (k = wave, wave + 4, ...) 64 at a time, one per lane; the loop then pops cells off a
        // ballot mask and fetches their offsets with v_readlane, SWEEP_DEPTH loads in flight per wave -- no
        // memory access and no divergent branch in the dependency chain.  The patch test (vector code, outside
        // the load loop) removes the cells whose patch holds only the free-space constant.
        const int r_lo = (ch * WAVE) / nq, r_hi = min(nx - 1, (ch * WAVE + WAVE - 1) / nq), ncols = 4 * nq;
        const double inv_pitch = 1.0 / (double)lv.fpitch;
        for (int base = wave; base < K; base += 4 * WAVE) {
            const int kk = base + 4 * lane;
            const bool valid = kk < K;
            const int cof = valid ? cl[kk] : 0;
            int y0 = (int)((double)cof * inv_pitch);                // cof / fpitch without the integer division
            if (y0 * lv.fpitch > cof) --y0;
            if ((y0 + 1) * lv.fpitch <= cof) ++y0;
            const int x0 = cof - y0 * lv.fpitch;
            bool load = valid;
            {
                const int tx0 = x0 >> BLUR_SHIFT, tx1 = (x0 + ncols - 1) >> BLUR_SHIFT;
                const unsigned long long need = ((2ull << (tx1 - tx0)) - 1ull) << tx0;
                const unsigned long long f = free_s[(y0 + r_lo) >> BLUR_SHIFT] & free_s[(y0 + r_hi) >> BLUR_SHIFT];
                load = valid && (~f & need) != 0ull;
                nfree += (unsigned)__popcll(__ballot(valid && !load));
            }
            unsigned long long todo = __ballot(load);
            const int boff = cof * 4;
            while (todo) {
                int c[SWEEP_DEPTH];
#pragma unroll
                for (int i = 0; i < SWEEP_DEPTH; ++i) {
                    c[i] = 0x7ffffff0;                             // beyond the buffer: reads zeros
                    if (todo) {
                        const int j = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        c[i] = __builtin_amdgcn_readlane(boff, j);
                    }
                }
                u32x4 v[SWEEP_DEPTH];
#pragma unroll
                for (int i = 0; i < SWEEP_DEPTH; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[0], c[i], 0);
#pragma unroll
                for (int i = 0; i < SWEEP_DEPTH; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned s2 = lo[0][e] + v[i][e];
                        hi[0][e] += s2 < v[i][e] ? 1u : 0u;
                        lo[0][e] = s2;
                    }
            }
        }
    } else if (RQ == 1 && deep) {
        // Round 4: a level whose launch is ONE round of blocks (the reference's 11 x 11 fine cube: 30 blocks per particle) is
        // bound by the latency of this loop, not by the gathers' throughput -- with two loads in flight a wave's ~40 cells were
        // ~20 round trips.  The wave's cells come in one vector load (a lane each), v_readlane hands them to the loop as scalar
        // offsets, SWEEP_DEPTH loads are in flight (the skip loop's structure without its patch test).
        for (int base = wave; base < K; base += 4 * WAVE) {
            const int kk = base + 4 * lane;
            const int boff = kk < K ? cl[kk] * 4 : 0x7ffffff0;                     // beyond the buffer: reads zeros
            const int n = min(WAVE, (K - base + 3) >> 2);                           // this wave's cells in this batch (wave-uniform)
            for (int j0 = 0; j0 < n; j0 += SWEEP_DEPTH) {
                int c[SWEEP_DEPTH];
#pragma unroll
                for (int i = 0; i < SWEEP_DEPTH; ++i) c[i] = __builtin_amdgcn_readlane(boff, min(j0 + i, WAVE - 1));
                u32x4 v[SWEEP_DEPTH];
#pragma unroll
                for (int i = 0; i < SWEEP_DEPTH; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[0], j0 + i < n ? c[i] : 0x7ffffff0, 0);
#pragma unroll
                for (int i = 0; i < SWEEP_DEPTH; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned s2 = lo[0][e] + v[i][e];
                        hi[0][e] += s2 < v[i][e] ? 1u : 0u;
                        lo[0][e] = s2;
                    }
            }
        }
    } else {
#pragma unroll 2
    for (int k = wave; k < K; k += 4) {
        const int cell = cl[k] * 4;
        u32x4 v[RQ];
#pragma unroll
        for (int r = 0; r < RQ; ++r) v[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[r], cell, 0);
#pragma unroll
        for (int r = 0; r < RQ; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned s = lo[r][e] + v[r][e];
                hi[r][e] += s < v[r][e] ? 1u : 0u;
                lo[r][e] = s;
            }
    }
    }
    if constexpr (SKIP) {                                  // the skipped cells: each adds the free-space cost
        const double fv = lv.floor_value;
        const unsigned long long add = (unsigned long long)nfree * (fv > 0.5 * fv ? 0u : (uint32_t)rint(-fv * lv.cost_scale));
#pragma unroll
        for (int r = 0; r < RQ; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long t = (((unsigned long long)hi[r][e] << 32) | lo[r][e]) + add;
                hi[r][e] = (unsigned)(t >> 32); lo[r][e] = (unsigned)t;
            }
    }
    DBG_CLOCK(dbg0 + 1, p == 0 && it == 0 && ch == 0);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < RQ; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) part_s[wave - 1][(r * 4 + e) * WAVE + lane] = ((unsigned long long)hi[r][e] << 32) | lo[r][e];
    }
    __syncthreads();
    DBG_CLOCK(dbg0 + 2, p == 0 && it == 0 && ch == 0);
    if (wave > 0) return;
    {
#pragma unroll
    for (int r = 0; r < RQ; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ix = (r * 4 + e) * WAVE + lane;
            const unsigned long long tot = (((unsigned long long)hi[r][e] << 32) | lo[r][e]) + part_s[0][ix] + part_s[1][ix] + part_s[2][ix];
            hi[r][e] = (unsigned)(tot >> 32); lo[r][e] = (unsigned)tot;
        }
    const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
    double* __restrict__ out = lv.cube + ((size_t)p * lv.ntheta + it) * npose;
    const double inv = 1.0 / lv.cost_scale;
    double sc[RQ][4], prv[RQ][4], ptw[RQ][4];
    Best me{-INFINITY, INT_MAX, 0};
    // both prior planes of the lane's poses in one batch of loads (one round trip, not eight)
#pragma unroll
    for (int r = 0; r < RQ; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = min(q0[r] + e, npose - 1);
            prv[r][e] = pr[q];
            ptw[r][e] = pr[npose + q];
        }
#pragma unroll
    for (int r = 0; r < RQ; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[r][e] = -INFINITY;
            if (e < nv[r]) {
                const int q = q0[r] + e;
                const unsigned long long acc = ((unsigned long long)hi[r][e] << 32) | lo[r][e];
                const double sum = -((double)acc * inv);                           // sum of probSP values
                sc[r][e] = (sum + prv[r][e]) + ptw[r][e];                          // :131
                out[q] = sc[r][e];
                Best cand{sc[r][e], it * npose + q, isnan(sc[r][e]) ? 1 : 0};
                if (better(cand, me)) me = cand;
            }
        }
    // per-wave reduction for k_select: max / argmax / sum exp(score - max)
    if constexpr (RQ == 1) me = wave_best_ordered(me);     // lane = one slot: indices ascend with the lane
    else me = wave_best(me);
    double ex = 0.0;
#pragma unroll
    for (int r = 0; r < RQ; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nv[r]) ex += exp(sc[r][e] - me.v);
    ex = wave_sum(ex);
    if (lane == 0) {
        Slam2dPartial pt;
        pt.max = me.v; pt.sumexp = ex; pt.argmax = me.i; pt.has_nan = me.nan;
        Slam2dPartial* dst = lv.partials + (size_t)p * lv.npartial + it * chunks + ch;
        if (mode == 0 && sel_out != nullptr) {             // (the particle's last block reads it in this launch: write-through)
            unsigned long long* u = reinterpret_cast<unsigned long long*>(dst);
            __hip_atomic_store(u, (unsigned long long)__double_as_longlong(pt.max), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(u + 1, (unsigned long long)__double_as_longlong(pt.sumexp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(u + 2, ((unsigned long long)(unsigned)pt.has_nan << 32) | (unsigned)pt.argmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            *dst = pt;
        }
    }
    DBG_CLOCK(dbg0 + 3, p == 0 && it == 0 && ch == 0);
    }
    };
    if constexpr (CPB > 1) {
        const int c0 = (w - it * groups) * CPB;
        for (int ch = c0; ch < min(chunks, c0 + CPB); ++ch) {
            sweep_chunk(ch);
            __syncthreads();                               // part_s is reused by the next chunk
        }
    } else {
        sweep_chunk(w - it * groups);
    }
    if constexpr (mode == 0) {
        // Round 4 (sel_out != NULL: a level matched by arg-max, i.e. no soft-max draw, which would need the cube): the block takes a
        // ticket from the particle's third arrival counter once its partials are out, and the particle's LAST block selects --
        // k_select's work without its launch.
        if (sel_out != nullptr && wave == 0) {
            // (publish / consume as in k_exact_select: the partials left as relaxed agent-scope 8-byte atomic stores, are drained here
            // and read back by the last block with load_partial_through's agent-scope atomic loads -- MI355X_MICROARCH.md's "8-B agent
            // atomics both sides"; no cache to write back or invalidate on either side)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            unsigned ticket = 0u;
            if (lane == 0) ticket = __hip_atomic_fetch_add(&lv.sync[p * SLAM2D_SYNC_WORDS + 2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
            if (ticket == (unsigned)(bpp - 1)) {
                if (lane == 0) __hip_atomic_store(&lv.sync[p * SLAM2D_SYNC_WORDS + 2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                select_argmax_from_partials(lv, p, lv.ntheta * chunks, sel_est, sel_estride, sel_out);
            }
        }
    }
}

// Small cubes (a plane of <= 32 slots of 4 dx: 2*ncell+1 <= 7, e.g. the 5 x 5 fine level behind a coarse factor of 2):
// k_sweep would leave most lanes of its waves without a slot.  Here ONE wave scores a whole (particle, theta) plane:
// lane = (slot, cell slice) -- floor(64 / slots) slices share the cell list, which is staged in LDS once (no
// vector-memory instruction per cell for the list), so a plane costs ~K / slices 16-byte gathers instead of K.
// Same exact integer sums, same cube / partial layout as k_sweep with one chunk per theta.
// One (particle, theta) plane of a small cube by one wave; returns the plane's reduction (to every lane) and stores the scores
// and, with `write`, the plane's Slam2dPartial.  small_lds: [64 * 4] partial sums, then [kmax] cells.
// NWAVES waves (one block) split the cell list; the scores, the reduction and the return value are wave 0's.
// small_lds: [NWAVES * 4 * 64] partial sums, then [kmax] cells.
template <int NWAVES>
__device__ __forceinline__ Best sweep_small_plane(const Slam2dLevel& lv, const int p, const int it, unsigned long long* small_lds,
                                                  const bool write) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const int nq = (nx + 3) >> 2, nslot = nx * nq;             // <= 32 (checked by the host)
    const int S = WAVE / nslot;                                 // cell slices per wave
    const int K = lv.kcount[p * lv.ntheta + it];
    const int* __restrict__ cl = lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    int* cells_s = reinterpret_cast<int*>(small_lds + NWAVES * WAVE * 4);
    for (int k = threadIdx.x; k < K; k += NWAVES * WAVE) cells_s[k] = cl[k];
    const size_t image = (size_t)lv.fmax * lv.fpitch;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.field + (size_t)p * image), (short)0, (int)(image * sizeof(uint32_t)), 0x00020000);
    const bool active = lane < S * nslot;
    const int u = lane % nslot, sl = lane / nslot;
    const int iy = u / nq, dx = (u - iy * nq) * 4;
    const int off = (iy * lv.fpitch + dx) * 4;
    const int ST = S * NWAVES;                                  // slices of the whole block
    unsigned lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};
    __syncthreads();
    for (int k = sl + S * wave; k < K; k += 8 * ST) {           // 8 gathers in flight
        int o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (active && k + i * ST < K) ? off + cells_s[k + i * ST] * 4 : 0x7ffffff0;
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o[i], 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned s2 = lo[e] + v[i][e];
                hi[e] += s2 < v[i][e] ? 1u : 0u;
                lo[e] = s2;
            }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) small_lds[(wave * 4 + e) * WAVE + lane] = ((unsigned long long)hi[e] << 32) | lo[e];
    __syncthreads();
    Best me{-INFINITY, INT_MAX, 0};
    if (NWAVES > 1 && wave != 0) return me;
    const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
    double* __restrict__ out = lv.cube + ((size_t)p * lv.ntheta + it) * npose;
    const double inv = 1.0 / lv.cost_scale;
    const int nv = lane < nslot ? min(4, nx - dx) : 0, q0 = iy * nx + dx;
    double sc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sc[e] = -INFINITY;
        if (e < nv) {
            unsigned long long acc = 0ull;
            for (int w2 = 0; w2 < NWAVES; ++w2)
                for (int s2 = 0; s2 < S; ++s2) acc += small_lds[(w2 * 4 + e) * WAVE + lane + s2 * nslot];
            const int q = q0 + e;
            sc[e] = (-((double)acc * inv) + pr[q]) + pr[npose + q];                 // :131, as k_sweep
            if (write) out[q] = sc[e];
            Best cand{sc[e], it * npose + q, isnan(sc[e]) ? 1 : 0};
            if (better(cand, me)) me = cand;
        }
    }
    me = wave_best_ordered(me);
    if (!write) return me;
    double ex = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (e < nv) ex += exp(sc[e] - me.v);
    ex = wave_sum(ex);
    if (lane == 0) {
        Slam2dPartial pt;
        pt.max = me.v; pt.sumexp = ex; pt.argmax = me.i; pt.has_nan = me.nan;
        lv.partials[(size_t)p * lv.npartial + it] = pt;
    }
    return me;
}
// pruned != 0 (Slam2dLevel.bnb == 3, "angle bounds"): a plane whose upper bound (k_abound) lies more than SLAM2D_BNB_MARGIN
// below the particle's seed score (k_aseed) is not scored -- it cannot hold the arg-max and all such planes together change
// the confidence by < ntheta * 25 * exp(-30) relative; its partial says "nothing here".
__global__ __launch_bounds__(64) void k_sweep_small(Slam2dLevel lv, int P, int pruned) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long small_lds[];   // [64 * 4] partial sums, then [kmax] cells
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int p = (slot / lv.ntheta) * 8 + xcd, it = slot % lv.ntheta;
    if (p >= P) return;
    if (pruned && !(lv.bounds[(size_t)p * lv.ntheta + it] >= unorder_bits(lv.bnb_best[p]) - SLAM2D_BNB_MARGIN)) {
        if (threadIdx.x == 0) {
            Slam2dPartial pt;
            pt.max = -INFINITY; pt.sumexp = 0.0; pt.argmax = INT_MAX; pt.has_nan = 0;
            lv.partials[(size_t)p * lv.npartial + it] = pt;
        }
        return;
    }
    sweep_small_plane<1>(lv, p, it, small_lds, true);
}

// Angle bounds (Slam2dLevel.bnb == 3; cubes of <= 7 x 7 poses per angle behind long cell lists, e.g. the 5 x 5 fine level of a
// 1081-beam scan).  All poses of one angle read, at endpoint cell k, a window of <= 7 x 7 field cells that starts at the patch
// corner; for windows of <= 5 x 5 cells it lies inside the aligned 8 x 8 block gmin2 summarises at (corner >> 2), so
//     U(theta) = -(sum_k gmin2[cell k] << 12) / cost_scale + max(rv + thetaWeight)
// bounds every pose of the plane: ONE 4-byte load per cell and angle, 64 cells per load instruction -- against one 16-byte
// gather per cell and 6 poses in the exact sweep, whose time is L1 tag lookups (30 sectors per load).  One wave per
// (particle, theta); the particle's best bound (packed with its theta) goes to seed_key for k_aseed.
#ifndef ABOUND_STRIDE
#define ABOUND_STRIDE 1
#endif
// (ABOUND_WAVES angles per block: a launch of one-wave blocks pays ~4.5 ns of dispatch per block whatever they do -- 8 896 blocks at
// config 5, 40-50 us for ten gathers per wave; round 6)
#ifndef ABOUND_WAVES
#define ABOUND_WAVES 16
#endif
__global__ __launch_bounds__(64 * ABOUND_WAVES) void k_abound(Slam2dLevel lv, int P) {
    const int b = blockIdx.x;
    const int ngrp = (lv.ntheta + ABOUND_WAVES - 1) / ABOUND_WAVES;
    const int xcd = b & 7, slot = b >> 3;
    const int p = (slot / ngrp) * 8 + xcd, it = (slot % ngrp) * ABOUND_WAVES + (threadIdx.x >> 6);
    if (p >= P || it >= lv.ntheta) return;
    const int lane = threadIdx.x & 63;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const int gp = lv.tmax << 2;
    const int K = lv.kcount[p * lv.ntheta + it];
    const int* __restrict__ pcl = lv.pcells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    const __amdgpu_buffer_rsrc_t rg2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.gmin2 + (size_t)p * gp * gp), (short)0, (int)((size_t)gp * gp * sizeof(uint32_t)), 0x00020000);
    // (Every term of the bound is <= 0, so leaving cells out keeps it an upper bound -- but using every 4th cell of the list
    // already made it too loose to prune anything at config 5: the exact sweep went from 32 back to 156 us.  Stride 1.)
    unsigned sum = 0u;                                     // 20-bit values, <= 2048 cells: no carry
    const int Ks = (K + ABOUND_STRIDE - 1) / ABOUND_STRIDE;
    for (int k0 = 0; k0 < Ks; k0 += WAVE * 8) {
        int off[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int k = k0 + i * WAVE + lane; off[i] = k < Ks ? pcl[k * ABOUND_STRIDE] : 0x7ffffff0; }
        unsigned v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(rg2, off[i], 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[i];
    }
    sum = row16_sum_u32(sum);
    const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)sum, 0) + (unsigned)__builtin_amdgcn_readlane((int)sum, 16) +
                         (unsigned)__builtin_amdgcn_readlane((int)sum, 32) + (unsigned)__builtin_amdgcn_readlane((int)sum, 48);
    // largest prior of the plane (+inf when one is NaN: np.argmax returns the first NaN, such a plane is always scored)
    double pmx = 0.0;
    if (!lv.fine) {
        const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
        pmx = -INFINITY;
        for (int q = lane; q < npose; q += WAVE) { const double v = pr[q] + pr[npose + q]; pmx = isnan(v) ? INFINITY : fmax(pmx, v); }
        pmx = wave64_max(pmx);
    }
    const double U = (-(((double)tot * 4096.0) / lv.cost_scale) + pmx) + 1e-9;
    if (lane == 0) {
        lv.bounds[(size_t)p * lv.ntheta + it] = U;
        if (U > -INFINITY && U < INFINITY) atomicMax(&lv.seed_key[p], (order_bits(U) & ~0x3FFFull) | (unsigned long long)it);
        else if (U == INFINITY) atomicMax(&lv.seed_key[p], (order_bits(1e300) & ~0x3FFFull) | (unsigned long long)it);
    }
}
// The seed: the plane of the particle's best-bound angle, exactly; its best score is a lower bound of the cube's maximum.  One
// 1024-thread block per particle (16 waves share the cell list: a lone wave took ~30 us for 900 cells).
#define ASEED_WAVES 16
__global__ __launch_bounds__(64 * ASEED_WAVES) void k_aseed(Slam2dLevel lv, int P) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long small_lds[];
    const int p = blockIdx.x;
    const unsigned long long key = lv.seed_key[p];
    double best = -INFINITY;                               // no finite bound anywhere: no threshold, every plane is scored
    if (key) {                                             // (block-uniform)
        const Best me = sweep_small_plane<ASEED_WAVES>(lv, p, (int)(key & 0x3FFF), small_lds, false);
        if (!me.nan && me.i != INT_MAX) best = me.v;
    }
    if (threadIdx.x == 0) lv.bnb_best[p] = order_bits(best);
}

// ------------------------------------------------------------------------------------
// K1d  arg-max / soft-max draw / confidence / matched pose   (Utils/ScanMatcher_OGBased.py:133-143)
//      one wave per particle, working on the per-wave partials of the sweep
// ------------------------------------------------------------------------------------
template <int mode>
__global__ __launch_bounds__(64) void k_select(Slam2dLevel lv, int chunks, int RQ, const double* __restrict__ est,
                                               int estride, const double* __restrict__ uniform, Slam2dMatch* out) {
    // mode as in k_sweep.  In mode 1 only the first ceil(ring length / 64) chunks of every theta hold
    // partials; the partial of (theta it, chunk ch) sits at it * chunks + ch in every mode.
    const int p = blockIdx.x, lane = threadIdx.x;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    if (mode == 2 && lv.prune_state[p] == 0) return;
    if (mode == 1 && lv.prune_state[p] != 0) return;
    const int nring = mode == 1 ? lv.ring[0] : 0;
    if (mode == 1 && nring < 0) { if (lane == 0) lv.prune_state[p] = 1; return; }
    const int vch = mode == 1 ? (nring + WAVE - 1) / WAVE : chunks;          // chunks that hold partials
    const int nW = lv.ntheta * vch;
    const Slam2dPartial* __restrict__ pt0 = lv.partials + (size_t)p * lv.npartial;
    auto part = [&](const int v) -> const Slam2dPartial& { return pt0[(v / vch) * chunks + (v % vch)]; };
    const double* __restrict__ c = lv.cube + (size_t)p * lv.ntheta * npose;
    // global max / argmax
    Best me{-INFINITY, INT_MAX, 0};
    for (int w = lane; w < nW; w += WAVE) {
        Best cand{part(w).max, part(w).argmax, part(w).has_nan};
        if (better(cand, me)) me = cand;
    }
    me = wave_best_fast(me);
    const double M = me.v;
    if (mode == 1) {
        // a pose outside the ring scores rv + (field sum) + thetaWeight <= -100 + K * (largest field value);
        // K = cells of its theta, so the fewest cells of any theta give the bound for all of them
        int kmin = INT_MAX;
        for (int it = lane; it < lv.ntheta; it += WAVE) kmin = min(kmin, lv.kcount[p * lv.ntheta + it]);
        for (int o = 1; o < WAVE; o <<= 1) kmin = min(kmin, __shfl_xor(kmin, o));
        const double outside = -100.0 + (double)kmin * lv.frames[p].field_max;
        if (!(M >= outside + SLAM2D_PRUNE_MARGIN)) {     // also NaN: the ring alone does not settle this particle
            if (lane == 0) lv.prune_state[p] = 1;
            return;
        }
    }
    // total = sum_w sumexp_w * exp(max_w - M): each lane owns a contiguous run of partials
    const int per = (nW + WAVE - 1) / WAVE;
    const int w0 = lane * per, w1 = min(nW, w0 + per);
    double mine = 0.0;
    for (int w = w0; w < w1; ++w) mine += part(w).sumexp * exp(part(w).max - M);
    const double incl = wave_scan_incl_f64(mine);     // inclusive scan over lanes
    const double total = readlane_f64(incl, WAVE - 1);
    int pick = me.i;
    if (uniform != nullptr && !isnan(total)) {
        // np.random.choice(n, 1, p): first index whose normalised cdf exceeds u (:137-138)
        const double target = uniform[p] * total;
        const unsigned long long ahead = __ballot(incl > target);
        const int lsel = ahead ? __ffsll((long long)ahead) - 1 : WAVE - 1;
        double run = readlane_f64(incl - mine, lsel);             // cdf before lane lsel's partials
        const int s0 = min(lsel * per, nW - 1), s1 = max(s0 + 1, min(nW, lsel * per + per));
        int wsel = s1 - 1;
        for (int w = s0; w < s1; ++w) {                           // wave-uniform loop
            const double t = part(w).sumexp * exp(part(w).max - M);
            if (run + t > target || w == s1 - 1) { wsel = w; break; }
            run += t;
        }
        const int it = wsel / vch, ch = wsel - it * vch;
        const int nq = (nx + 3) >> 2, nslot = nx * nq;
        if (mode == 1) {
            // ring chunk: lane l owns slot ring[ch*64 + l] = up to 4 consecutive poses; slots ascend with l,
            // so the lanes walk the chunk's poses in cube order
            const int idx = ch * WAVE + lane;
            const int u = idx < nring ? lv.ring[1 + idx] : nslot;
            const int iy = u / nq, dx = (u - iy * nq) * 4;
            const int q0 = iy * nx + dx, nvl = u < nslot ? min(4, nx - dx) : 0;
            double lsum = 0.0;
            for (int e = 0; e < nvl; ++e) lsum += exp(c[(size_t)it * npose + q0 + e] - M);
            const double linc = wave_scan_incl_f64(lsum);
            const unsigned long long hit = __ballot(nvl > 0 && run + linc > target);
            const unsigned long long have = __ballot(nvl > 0);
            const int llast = 63 - __clzll((long long)have);                      // chunk's last slot (have != 0)
            const int l2 = hit ? __ffsll((long long)hit) - 1 : llast;
            double r2 = run + readlane_f64(linc - lsum, l2);
            int found = -1;
            if (lane == l2) {
                for (int e = 0; e < nvl; ++e) {
                    r2 += exp(c[(size_t)it * npose + q0 + e] - M);
                    if (r2 > target) { found = it * npose + q0 + e; break; }
                }
                if (found < 0) found = it * npose + q0 + max(0, nvl - 1);         // rounding fallback: last pose
            }
            pick = __builtin_amdgcn_readlane(found, l2);
        } else {
        // inside chunk wsel: it covers slots [ch*64*RQ, ...) = a contiguous pose range [qlo, qhi);
        // lane l owns a contiguous run of `per2` poses of it
        const int s0c = min(nslot, ch * WAVE * RQ), s1c = min(nslot, (ch + 1) * WAVE * RQ);
        const int qlo = (s0c / nq) * nx + min(nx, 4 * (s0c % nq));
        const int qhi = (s1c / nq) * nx + min(nx, 4 * (s1c % nq));
        const int per2 = (qhi - qlo + WAVE - 1) / WAVE;
        const int a0 = qlo + lane * per2, a1 = min(qhi, a0 + per2);
        double lsum = 0.0;
        for (int q = a0; q < a1; ++q) lsum += exp(c[(size_t)it * npose + q] - M);
        const double linc = wave_scan_incl_f64(lsum);
        const unsigned long long hit = __ballot(run + linc > target);
        pick = it * npose + max(qlo, qhi - 1);                    // rounding fallback: chunk's last pose
        if (hit) {
            const int l2 = __ffsll((long long)hit) - 1;
            double r2 = run + readlane_f64(linc - lsum, l2);
            int found = -1;
            if (lane == l2) {
                for (int q = a0; q < a1; ++q) {
                    r2 += exp(c[(size_t)it * npose + q] - M);
                    if (r2 > target) { found = it * npose + q; break; }
                }
                if (found < 0) found = it * npose + max(a0, a1 - 1);
            }
            pick = __builtin_amdgcn_readlane(found, l2);
        }
        }
    }
    if (lane == 0) {
        Slam2dMatch m;
        const int it = pick / npose, rem = pick - it * npose;
        const int iy = rem / nx, ix = rem - iy * nx;
        const double ex = est[(size_t)p * estride], ey = est[(size_t)p * estride + 1], eth = est[(size_t)p * estride + 2];
        m.x = ex + (double)(ix - lv.ncell) * lv.step;                               // :142-143
        m.y = ey + (double)(iy - lv.ncell) * lv.step;
        m.theta = eth + lv.thetas[it];
        m.confidence = exp(M) * total;                                              // :141
        m.log_confidence = M + log(total);
        m.best_score = M;
        m.pick = pick;
        m.argmax = me.i;
        out[p] = m;
        if (lv.arrive) __hip_atomic_fetch_add(lv.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (Slam2dScan.match_seq)
    }
}

// ------------------------------------------------------------------------------------
// K1e  branch and bound over 4x4 pose tiles (include/slam2d.h, "Branch and bound"): the same scores as k_sweep
//      for every pose that can matter, from 1/16 of the gathers plus the exact scores of a few per cent of the tiles.
//        blur / triage  write gmin (minimum of every aligned 4x4 block of the cost image) with the field tiles;
//                  k_blur_check_redo derives gmin2 = min over 2x2 blocks (>> 12): a lower bound of the cost anywhere
//                  in the aligned 8x8 block that contains a pose tile's 4x4 window at a cell.
//        k_bound   one wave per (particle, theta): upper bound U of every tile (lane = one pose-tile row x 4
//                  consecutive tiles: ONE 16-byte load per cell from an image 1/16 the size of the field, 16 in
//                  flight, plain 32-bit adds); then the tile with the largest U is scored exactly and its best
//                  score raises lv.bnb_best[p] (atomic max: order-independent, deterministic).
//        k_exact   one wave per (particle, theta): every tile with U >= bnb_best - SLAM2D_PRUNE_MARGIN is scored
//                  exactly (lane = pose row x 1/16 of the cell list); ONE partial {max, argmax, sum exp} per theta.
// ------------------------------------------------------------------------------------
struct Running { double m, s; int arg, nan; };
__device__ __forceinline__ void running_merge(Running& a, const double tm, const double ts, const int targ, const int tnan) {
    if (a.nan || tnan) {
        const int arg = (a.nan && tnan) ? min(a.arg, targ) : (tnan ? targ : a.arg);
        a.nan = 1; a.arg = arg; a.m = NAN; a.s = NAN;
        return;
    }
    if (tm == -INFINITY) return;
    if (tm > a.m) { a.s = a.s * exp(a.m - tm) + ts; a.m = tm; a.arg = targ; }
    else { a.s += ts * exp(tm - a.m); if (tm == a.m && targ < a.arg) a.arg = targ; }
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
    const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
    return ((unsigned long long)hi << 32) | lo;
}

// exact cost sums of the 16 poses of tile (by, bx): lane = (pose row r = lane / 16, cell slice s = lane % 16) scores
// the 4 poses (4 by + r, 4 bx .. 4 bx + 3) against cells k0 + s, k0 + s + kstep, ...; on return every lane of a
// row group holds the row's 4 sums over ALL the cells this wave walked (reduced over the 16 slices).
#define EXACT_DEPTH 8
#define SLAM2D_BEAM_TABLE_MIN 512   // beams from which k_frame_axis (with its per-particle beam-endpoint table) is worth its launch
// byte offsets of the first NPRE cells of lane slice s (cells s, s + 16, ...), beyond-the-buffer where the list ends
template <int NPRE>
__device__ __forceinline__ void tile_prefetch(const int* __restrict__ cl, const int K, int (&pre)[NPRE]) {
    const int s = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) pre[i] = s + 16 * i < K ? cl[s + 16 * i] * 4 : 0x7ffffff0;
}
template <int NPRE>
__device__ __forceinline__ void tile_exact(const Slam2dLevel& lv, const __amdgpu_buffer_rsrc_t rsrc, const int* __restrict__ cl,
                                           const int K, const int by, const int bx, const int (&pre)[NPRE],
                                           unsigned long long (&acc)[4]) {
    const int lane = threadIdx.x & 63, r = lane >> 4, s = lane & 15;
    const int lanepart = ((4 * by + r) * lv.fpitch + 4 * bx) * 4;
    unsigned lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};
    {
        u32x4 v[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lanepart + pre[i], 0, 0);
#pragma unroll
        for (int i = 0; i < NPRE; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned s2 = lo[e] + v[i][e];
                hi[e] += s2 < v[i][e] ? 1u : 0u;
                lo[e] = s2;
            }
    }
    for (int k = s + 16 * NPRE; k < K; k += EXACT_DEPTH * 16) {            // longer lists
        int off[EXACT_DEPTH];
#pragma unroll
        for (int i = 0; i < EXACT_DEPTH; ++i) {
            const int kk = k + i * 16;
            off[i] = kk < K ? lanepart + cl[kk] * 4 : 0x7ffffff0;      // beyond the buffer: reads zeros
        }
        u32x4 v[EXACT_DEPTH];
#pragma unroll
        for (int i = 0; i < EXACT_DEPTH; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], 0, 0);
#pragma unroll
        for (int i = 0; i < EXACT_DEPTH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned s2 = lo[e] + v[i][e];
                hi[e] += s2 < v[i][e] ? 1u : 0u;
                lo[e] = s2;
            }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = row16_sum_u64(((unsigned long long)hi[e] << 32) | lo[e]);
}

// One wave per (particle, theta); blocks of a particle pinned to one XCD like the sweep's.  Lane = one pose-tile row
// x 4 consecutive tiles: ONE 16-byte load per cell.  The loop is bound by the texture path's 16 clocks per wave
// instruction (a dword-per-tile mapping needs two instructions per cell and measured 2x slower), so what counts
// is the instruction count, 1/8 of the brute-force sweep's.  The cell list is read 64 cells at a time, one per
// lane, and handed to the loads with v_readlane: no scalar-memory round trip per batch.
#define BOUND_BATCH 32
#define BOUND_GROUP 2                // waves per block = adjacent angles of one particle (measured: 1 -> 31.4 us, 2 -> 30.2, 4 -> 33.5,
//                                      8 -> 38.3 at config 2: sharing L1 lines buys little, spreading over the CUs matters)
__global__ __launch_bounds__(64 * BOUND_GROUP) void k_bound(Slam2dLevel lv, int P) {
    const int b = blockIdx.x;
    const int ngroup = (lv.ntheta + BOUND_GROUP - 1) / BOUND_GROUP;
    const int xcd = b & 7, slot = b >> 3;
    const int p = (slot / ngroup) * 8 + xcd, it = (slot % ngroup) * BOUND_GROUP + (threadIdx.x >> 6);
    if (p >= P || it >= lv.ntheta) return;
    const int lane = threadIdx.x & 63;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const int nbt = (nx + 3) >> 2, nq = (nbt + 3) >> 2, nbq4 = nq << 2;
    const int gp = lv.tmax << 2;
    const int K = lv.kcount[p * lv.ntheta + it];
    const int* __restrict__ pcl = lv.pcells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    const int* __restrict__ cl = lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.gmin2 + (size_t)p * gp * gp), (short)0, (int)((size_t)gp * gp * sizeof(uint32_t)), 0x00020000);
    DBG_CLOCK(0, b == 0);
    const bool active = lane < nbt * nq;
    const int by = lane / nq, q = lane - by * nq;
    const int voff = active ? (by * gp + 4 * q) * 4 : 0x7ffffff0;
    const double* __restrict__ pm = lv.tile_pmax + (size_t)p * nbt * nbq4;
    double pmx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pmx[e] = active ? pm[by * nbq4 + 4 * q + e] : -INFINITY;     // in flight during the loop
    int pre[16];
    tile_prefetch<16>(cl, K, pre);                          // for the seed tile below: likewise
    unsigned sum[4] = {0u, 0u, 0u, 0u};                    // 20-bit values, <= 2048 cells: no carry
    int cv = lane < K ? pcl[lane] : 0x7ffffff0;
    for (int base = 0; base < K; base += WAVE) {
        const int cur = cv;
        if (base + WAVE < K) cv = base + WAVE + lane < K ? pcl[base + WAVE + lane] : 0x7ffffff0;     // next 64 cells
        // (straight-line double-buffered chunks that keep 16-32 loads in flight across the whole list measured SLOWER,
        // 31.9 vs 28.4 us: the loop is bound by the L1's tag lookups -- ~19 64-byte sectors per load, 89 % hits -- not by
        // latency, and the padding loads of the last chunk are not free)
#pragma unroll
        for (int j0 = 0; j0 < WAVE; j0 += BOUND_BATCH) {
            if (base + j0 < K) {                            // (wave-uniform)
                u32x4 v[BOUND_BATCH];
#pragma unroll
                for (int i = 0; i < BOUND_BATCH; ++i)
                    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, voff, __builtin_amdgcn_readlane(cur, j0 + i), 0);
#pragma unroll
                for (int i = 0; i < BOUND_BATCH; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[e] += v[i][e];
            }
        }
    }
    DBG_CLOCK(1, b == 0);
    const double inv = 1.0 / lv.cost_scale;
    double* __restrict__ bnd = lv.bounds + ((size_t)p * lv.ntheta + it) * nbt * nbq4;
    Best me{-INFINITY, INT_MAX, 0};
    if (active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = by * nbq4 + 4 * q + e;
            // every pose of the tile scores (-(cost sum) / scale + rv) + thetaWeight <= -L / scale + max(rv + thetaWeight),
            // L = (sum of the block minima >> 12) << 12; 1e-9 covers the different association of the roundings;
            // padding tiles (4 q + e >= nbt) come out as -inf (their tile_pmax is)
            const double U = (-(((double)sum[e] * 4096.0) * inv) + pmx[e]) + 1e-9;
            bnd[t] = U;
            Best cand{U, t, 0};
            if (4 * q + e < nbt && better(cand, me)) me = cand;
        }
    }
    me = wave_best_ordered(me);                             // tiles ascend with the lane
    if (lv.theta_umax && lane == 0) lv.theta_umax[(size_t)p * lv.ntheta + it] = me.i != INT_MAX ? me.v : -INFINITY;
    DBG_CLOCK(2, b == 0);
    // the tile with the largest bound, exactly: its best score is a lower bound of the cube's maximum
    const int seed = me.i;
    const int sby = seed / nbq4, sbx = seed - sby * nbq4;
    const int dy = 4 * sby + (lane >> 4);
    const bool leader = (lane & 15) == 0 && dy < nx;
    const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
    double prv[4], ptw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {                           // issued ahead of the tile's field loads
        const int qq = min(dy, nx - 1) * nx + min(4 * sbx + e, nx - 1);
        prv[e] = pr[qq]; ptw[e] = pr[npose + qq];
    }
    const size_t image = (size_t)lv.fmax * lv.fpitch;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.field + (size_t)p * image), (short)0, (int)(image * sizeof(uint32_t)), 0x00020000);
    unsigned long long acc[4];
    tile_exact<16>(lv, rsrc, cl, K, sby, sbx, pre, acc);
    double val = -INFINITY;
    if (leader) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * sbx + e < nx) {
                const double sc = (-((double)acc[e] * inv) + prv[e]) + ptw[e];
                if (!isnan(sc)) val = fmax(val, sc);
            }
    }
    val = fmax(val, __shfl_xor(val, 16));
    val = fmax(val, __shfl_xor(val, 32));
    DBG_CLOCK(3, b == 0);
    if (lane == 0 && val > -INFINITY) atomicMax(&lv.bnb_best[p], order_bits(val));
    DBG_CLOCK(4, b == 0);
}

__device__ __forceinline__ int wave64_min_i32(int v) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
// The seed of one (particle, theta): tile `seed` (the one with the largest bound) scored exactly by one wave; its best score
// raises lv.bnb_best[p].  (k_bound carries the same lines inline.)
__device__ __forceinline__ void bound_seed_exact(const Slam2dLevel& lv, const int p, const int sby, const int sbx, const int* __restrict__ cl,
                                                 const int K, const double inv) {
    const int lane = threadIdx.x & 63;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    int pre[16];
    tile_prefetch<16>(cl, K, pre);
    const int dy = 4 * sby + (lane >> 4);
    const bool leader = (lane & 15) == 0 && dy < nx;
    const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
    double prv[4], ptw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {                           // issued ahead of the tile's field loads
        const int qq = min(dy, nx - 1) * nx + min(4 * sbx + e, nx - 1);
        prv[e] = pr[qq]; ptw[e] = pr[npose + qq];
    }
    const size_t image = (size_t)lv.fmax * lv.fpitch;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.field + (size_t)p * image), (short)0, (int)(image * sizeof(uint32_t)), 0x00020000);
    unsigned long long acc[4];
    tile_exact<16>(lv, rsrc, cl, K, sby, sbx, pre, acc);
    double val = -INFINITY;
    if (leader) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * sbx + e < nx) {
                const double sc = (-((double)acc[e] * inv) + prv[e]) + ptw[e];
                if (!isnan(sc)) val = fmax(val, sc);
            }
    }
    val = fmax(val, __shfl_xor(val, 16));
    val = fmax(val, __shfl_xor(val, 32));
    if (lane == 0 && val > -INFINITY) atomicMax(&lv.bnb_best[p], order_bits(val));
}

// k_bound with the particle's bound image staged in LDS (the filled-machine form: DESIGN 4).  k_bound's loop is bound by the L1's
// tag lookups -- every wave-load touches ~20 lines (11 pose-tile rows of a 2-D window), 3 160 lines per (particle, theta) -- and
// at 128-256 particles per launch it runs AT that bound (profiles/r06_config2_p256_counters.txt: 0.70 of one line per clock and
// CU).  Here a block serves one particle (or 1 / bpp of its angles): its waves copy the particle's byte image of the bounds
// (Slam2dLevel.gmin2b: cost >> 24, written beside gmin2; gp x lp bytes, 42 KB at config 2, three blocks per CU) into LDS, then wave w
// bounds its angles one after the other: lane = one pose tile (NSET tiles per lane), one ds_read_u8 per cell and tile set --
// 2 LDS cycles per wave instruction instead of ~20 L1 cycles.  The bound is looser by < 2^-7 per cell (floor of 12 more bits;
// 1.2 at 158 cells against the margin of 30): a few more tiles survive, the results do not change (every surviving tile is
// scored exactly).  Lists are padded to a multiple of BL_BATCH cells with the cell at offset 0; what the padding added is
// taken off afterwards.
// Seeds: as k_bound, the best-bound tile of an angle is scored exactly -- unless its bound does not exceed lv.bnb_best[p] any
// more (a wave's second and third angles mostly find it raised by the first round's seeds): a seed's exact score cannot exceed
// its bound, so the FINAL bnb_best is the maximum over all seeds whatever is skipped, in whatever order (k_bound2's rule).
#define BL_BATCH 16
#ifndef BL_RLE_BATCH
#define BL_RLE_BATCH 8       // (runs of a 64-cell chunk: ~22 at config 5 -- batches of 16 waste a third of their gathers)
#endif
// RLE (long lists: ~1000 beams): neighbouring beams end in the same 4 x 4-cell block more often than not, so a wave's 64 cell offsets are
// run-length compressed first (ballot of the run heads, the (offset, run length) pairs compacted through 256 bytes of LDS per wave)
// and the gathers go over the runs: sum += entry x run length.
template <int NSET, bool RLE>
__global__ __launch_bounds__(1024) void k_bound_lds(Slam2dLevel lv, int P, int bpp, int tpb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char g2s[];                  // [gp][lp] (+ RLE: [waves][64] words)
    const int lp = lv.g2b_pitch;
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int p = (slot / bpp) * 8 + xcd, part = slot % bpp;
    if (p >= P) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int nx = 2 * lv.ncell + 1;
    const int nbt = (nx + 3) >> 2, nq = (nbt + 3) >> 2, nbq4 = nq << 2;
    const int gp = lv.tmax << 2;
    DBG_CLOCK(0, b == 0);
    {   // the image, as it lies in memory: 16-byte chunks, four loads in flight per thread
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(lv.gmin2b + (size_t)p * gp * lp);
        u32x4* dst = reinterpret_cast<u32x4*>(g2s);
        const int n16 = (gp * lp) >> 4, nt = blockDim.x;
        for (int i = tid; i < n16; i += 4 * nt) {
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i + k * nt < n16) v[k] = src[i + k * nt];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i + k * nt < n16) dst[i + k * nt] = v[k];
        }
    }
    DBG_CLOCK(1, b == 0);
    // this lane's tiles (byte offsets into the LDS image, slots of the bounds array)
    int lbase[NSET], tslot[NSET];
    bool valid[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        const int t = s * WAVE + lane;
        valid[s] = t < nbt * nbt;
        const int by = valid[s] ? t / nbt : 0, bx = valid[s] ? t - by * nbt : 0;
        lbase[s] = by * lp + bx;
        tslot[s] = (by << 8) | bx;
    }
    const double inv = 1.0 / lv.cost_scale;
    const double* __restrict__ pm = lv.tile_pmax + (size_t)p * nbt * nbq4;
    double pmx[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) pmx[s] = valid[s] ? pm[(tslot[s] >> 8) * nbq4 + (tslot[s] & 255)] : -INFINITY;
    const unsigned magic = (unsigned)((0x100000000ull + (unsigned)gp - 1u) / (unsigned)gp);      // e / gp = umulhi(e, magic), e < 2^32 / gp
    const int it0 = part * tpb, it_end = min(lv.ntheta, it0 + tpb);
    __syncthreads();
    DBG_CLOCK(2, b == 0);
    // angles from the middle of the block's range outwards: the first round (which finds lv.bnb_best[p] at -inf and scores every
    // seed) then holds the angles nearest the estimate's, where a tracked pose has its maximum
    const int nth = it_end - it0, mid = (nth - 1) >> 1;
    double seedU = -INFINITY;
    int seedT = INT_MAX, seedIt = 0, seedK = 0;
    for (int j = w; j < nth; j += nw) {
        const int it = it0 + mid + ((j & 1) ? ((j + 1) >> 1) : -((j + 1) >> 1));
        const int K = lv.kcount[p * lv.ntheta + it];
        const int* __restrict__ pcl = lv.pcells + ((size_t)p * lv.ntheta + it) * lv.kmax;
        unsigned sum[NSET];
#pragma unroll
        for (int s = 0; s < NSET; ++s) sum[s] = 0u;
        // a cell's byte offset into gmin2 ((Y0 gp + X0) 4) -> into the LDS image (Y0 lp + X0); beyond the list: offset 0
        auto conv = [&](const int k) -> int {
            if (k >= K) return 0;
            const unsigned e = (unsigned)pcl[k] >> 2, Y0 = __umulhi(e, magic);
            return (int)(Y0 * (unsigned)lp + (e - Y0 * (unsigned)gp));
        };
        int cv = conv(lane);
        for (int base = 0; base < K; base += WAVE) {
            const int cur = cv;
            if (base + WAVE < K) cv = conv(base + WAVE + lane);
            if constexpr (RLE) {
                volatile unsigned* rl = reinterpret_cast<volatile unsigned*>(g2s + (((size_t)gp * lp + 15) & ~(size_t)15)) + w * WAVE;
                const int n = min(WAVE, K - base);
                const int prev = __shfl_up(cur, 1);
                const bool head = lane < n && (lane == 0 || cur != prev);
                const unsigned long long hm = __ballot(head);
                const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
                const int nu = __popcll(hm);
                if (head) {
                    const unsigned long long later = hm & ~(below | (1ull << lane));
                    const int nxt = later ? __ffsll((long long)later) - 1 : n;
                    rl[__popcll(hm & below)] = (unsigned)cur | ((unsigned)(nxt - lane) << 16);     // (offsets < 64 KB: checked at the launch)
                }
                const unsigned comp = lane < nu ? rl[lane] : 0u;                                    // (one wave: its LDS operations stay in order)
#pragma unroll
                for (int j0 = 0; j0 < WAVE; j0 += BL_RLE_BATCH) {
                    if (j0 < nu) {                              // (wave-uniform)
                        unsigned char v[NSET][BL_RLE_BATCH];
                        unsigned cnt[BL_RLE_BATCH];
#pragma unroll
                        for (int i = 0; i < BL_RLE_BATCH; ++i) {
                            const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)comp, j0 + i);
                            cnt[i] = so >> 16;
#pragma unroll
                            for (int s = 0; s < NSET; ++s) v[s][i] = g2s[lbase[s] + (int)(so & 0xFFFFu)];
                        }
#pragma unroll
                        for (int i = 0; i < BL_RLE_BATCH; ++i)
#pragma unroll
                            for (int s = 0; s < NSET; ++s) sum[s] += (unsigned)v[s][i] * cnt[i];
                    }
                }
            } else {
#pragma unroll
            for (int j0 = 0; j0 < WAVE; j0 += BL_BATCH) {
                if (base + j0 < K) {                        // (wave-uniform)
                    unsigned char v[NSET][BL_BATCH];
#pragma unroll
                    for (int i = 0; i < BL_BATCH; ++i) {
                        const int so = __builtin_amdgcn_readlane(cur, j0 + i);
#pragma unroll
                        for (int s = 0; s < NSET; ++s) v[s][i] = g2s[lbase[s] + so];
                    }
#pragma unroll
                    for (int i = 0; i < BL_BATCH; ++i)
#pragma unroll
                        for (int s = 0; s < NSET; ++s) sum[s] += v[s][i];
                }
            }
            }
        }
        DBG_CLOCK(3, b == 0 && j == 0);
        const unsigned npad = RLE ? 0u : (unsigned)(((K + BL_BATCH - 1) / BL_BATCH) * BL_BATCH - K);
        double* __restrict__ bnd = lv.bounds + ((size_t)p * lv.ntheta + it) * nbt * nbq4;
        Best me{-INFINITY, INT_MAX, 0};
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            if (valid[s]) {
                const unsigned sm = sum[s] - npad * (unsigned)g2s[lbase[s]];
                // as k_bound: U >= every pose's score of the tile; L = (sum of the block minima >> 24) << 24
                const double U = (-(((double)sm * 16777216.0) * inv) + pmx[s]) + 1e-9;
                bnd[(tslot[s] >> 8) * nbq4 + (tslot[s] & 255)] = U;
                Best cand{U, tslot[s], 0};
                if (better(cand, me)) me = cand;
            }
        }
        if (lane < nbt)                                                     // padding tiles of every tile row: -inf
            for (int bx = nbt; bx < nbq4; ++bx) bnd[lane * nbq4 + bx] = -INFINITY;
        me = wave_best_fast(me);
        if (lv.theta_umax && lane == 0) lv.theta_umax[(size_t)p * lv.ntheta + it] = me.i != INT_MAX ? me.v : -INFINITY;
        DBG_CLOCK(4, b == 0 && j == 0);
        if (me.i == INT_MAX) continue;
        if (RLE) {
            // long lists: a seed costs as much as the angle's bounds (16 poses x ~1000 cells: 17.8 us against 16.0 at config 5), and the
            // looser byte bounds let few later seeds be skipped -- the wave scores ONE seed, that of its angle with the largest
            // bound (the particle's largest bound is some wave's largest: the seed that matters is among them)
            if (me.v > seedU || seedT == INT_MAX) { seedU = me.v; seedT = me.i; seedIt = it; seedK = K; }
            continue;
        }
        const unsigned long long best = __hip_atomic_load(&lv.bnb_best[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(me.v > unorder_bits(best))) continue;                         // (wave-uniform; bounds are never NaN: tile_pmax maps NaN to +inf)
        bound_seed_exact(lv, p, me.i >> 8, me.i & 255, lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax, K, inv);
        DBG_CLOCK(5, b == 0 && j == 0);
    }
    if (RLE && seedT != INT_MAX) {
        const unsigned long long best = __hip_atomic_load(&lv.bnb_best[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seedU > unorder_bits(best))
            bound_seed_exact(lv, p, seedT >> 8, seedT & 255, lv.cells + ((size_t)p * lv.ntheta + seedIt) * lv.kmax, seedK, inv);
    }
    DBG_CLOCK(14, b == 0);
}

// Two-level bounds for long cell lists (Slam2dLevel.bnb == 2; K ~ 1000 at 1081 beams, where k_bound sits at its
// texture-path bound of one load instruction per cell and theta).  Level 1: tiles of 8 x 8 poses, bounded through gmin3d
// (the 8 x 8 cell window of such a tile lies inside 3 x 3 aligned 4 x 4 blocks); a tile row is 2 quads, a cell 12-16 lanes,
// so ONE load instruction serves 4-5 cells.  Level 2 (k_bound2): the four 4 x 4-pose children of the few level-1 tiles that
// survive (0.3 per theta at config 5) get their gmin2 bound; every other 4 x 4 tile's bound is -inf.  k_exact_select is
// unchanged.  The seed: best level-1 tile -> its best child by level-2 bound -> exact.
// level-2 bounds of the four children of level-1 tile (R, Cx): lane = (child c = lane / 16, cell slice lane % 16); returns
// the child's bound in every lane of its row (-inf for a child outside the cube)
__device__ __forceinline__ double children_bounds(const Slam2dLevel& lv, const __amdgpu_buffer_rsrc_t rg2, const int* __restrict__ pcl,
                                                  const int K, const int R, const int Cx, const double* __restrict__ pm,
                                                  const int nbt, const int nbq4, const int gp, const double inv) {
    const int lane = threadIdx.x & 63, c = lane >> 4, sl = lane & 15;
    const int cy = 2 * R + (c >> 1), cx = 2 * Cx + (c & 1);
    const bool valid = cy < nbt && cx < nbt;
    const int lanepart = (cy * gp + cx) * 4;
    unsigned sum = 0u;
    for (int k = sl; k < K; k += 16 * 8) {
        int off[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) off[i] = (valid && k + 16 * i < K) ? lanepart + pcl[k + 16 * i] : 0x7ffffff0;
        unsigned v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b32(rg2, off[i], 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[i];
    }
    sum = row16_sum_u32(sum);
    return valid ? (-(((double)sum * 4096.0) * inv) + pm[cy * nbq4 + cx]) + 1e-9 : -INFINITY;
}

#define B1_DEPTH 16
__global__ __launch_bounds__(64) void k_bound1(Slam2dLevel lv, int P) {
    extern __shared__ __attribute__((aligned(16))) int b1_lds[];      // [64 * 4] partial sums, then [kmax] cell offsets
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int p = (slot / lv.ntheta) * 8 + xcd, it = slot % lv.ntheta;
    if (p >= P) return;
    const int lane = threadIdx.x;
    const int nx = 2 * lv.ncell + 1;
    const int nbt = (nx + 3) >> 2, nbq4 = ((nbt + 3) >> 2) << 2;
    const int nb1 = (nx + 7) >> 3, nq1 = (nb1 + 3) >> 2, L = nb1 * nq1, C = WAVE / L;
    const int gp = lv.tmax << 2, hp = gp >> 1;
    const int K = lv.kcount[p * lv.ntheta + it];
    const int* __restrict__ p3 = lv.p3cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    unsigned* red = reinterpret_cast<unsigned*>(b1_lds);
    int* cs = b1_lds + WAVE * 4;
    for (int k = lane; k < K; k += WAVE) cs[k] = p3[k];
    const __amdgpu_buffer_rsrc_t rg3 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.gmin3d + (size_t)p * gp * gp), (short)0, (int)((size_t)gp * gp * sizeof(uint32_t)), 0x00020000);
    const int g = lane / L, l2 = lane - g * L, r = l2 / nq1, q = l2 - r * nq1;
    const bool active = g < C;
    const int lanepart = (r * hp + 4 * q) * 4;
    unsigned sum[4] = {0u, 0u, 0u, 0u};
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += C * B1_DEPTH) {
        int off[B1_DEPTH];
#pragma unroll
        for (int i = 0; i < B1_DEPTH; ++i) {
            const int k = k0 + i * C + g;
            off[i] = (active && k < K) ? lanepart + cs[k] : 0x7ffffff0;
        }
        u32x4 v[B1_DEPTH];
#pragma unroll
        for (int i = 0; i < B1_DEPTH; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rg3, off[i], 0, 0);
#pragma unroll
        for (int i = 0; i < B1_DEPTH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] += v[i][e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[e * WAVE + lane] = sum[e];
    __syncthreads();
    const double inv = 1.0 / lv.cost_scale;
    const double* __restrict__ pm = lv.tile_pmax + (size_t)p * nbt * nbq4;
    double* __restrict__ b1 = lv.bounds1 + ((size_t)p * lv.ntheta + it) * 64;
    Best me{-INFINITY, INT_MAX, 0};
    if (lane < L) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c1 = 4 * q + e;
            if (c1 < 8) {
                unsigned tot = 0u;
                for (int g2 = 0; g2 < C; ++g2) tot += red[e * WAVE + g2 * L + lane];
                // largest prior of the tile's children (+inf if a child holds a NaN prior, -inf for a tile outside the cube)
                double pmx = -INFINITY;
                for (int a = 0; a < 2; ++a)
                    for (int bb = 0; bb < 2; ++bb)
                        if (2 * r + a < nbt && 2 * c1 + bb < nbt) pmx = fmax(pmx, pm[(2 * r + a) * nbq4 + 2 * c1 + bb]);
                const double U = (-(((double)tot * 4096.0) * inv) + pmx) + 1e-9;
                b1[r * 8 + c1] = U;
                Best cand{U, r * 8 + c1, 0};
                if (c1 < nb1 && better(cand, me)) me = cand;
            }
        }
    }
    if (lane >= L && lane < 64) {                           // level-1 tiles beyond the cube: never survive
        for (int t = lane - L; t < 64; t += 64 - L) {
            const int r1 = t >> 3, c1 = t & 7;
            if (r1 >= nb1 || c1 >= 4 * nq1) b1[t] = -INFINITY;
        }
    }
    me = wave_best_ordered(me);
    // candidate seed of the particle: its best finite level-1 bound over all theta (k_seed scores it exactly); the low
    // 14 bits of the bound make room for (theta, tile) -- this only picks WHERE the threshold is measured
    if (lane == 0 && me.v > -INFINITY && me.v < INFINITY)
        atomicMax(&lv.seed_key[p], (order_bits(me.v) & ~0x3FFFull) | (unsigned long long)((it << 6) | me.i));
}

// The threshold of a particle (bnb == 2): the level-1 tile with the best bound over all theta -> its best child by level-2
// bound -> that 4 x 4 tile exactly; the best of its 16 scores is a lower bound of the cube's maximum.  One block per
// particle, thread = cell (one seed per PARTICLE: the per-theta seeds of k_bound cost a third of its cache-line touches).
#define SEED_THREADS 1024
__global__ __launch_bounds__(SEED_THREADS) void k_seed(Slam2dLevel lv, int P) {
    __shared__ unsigned csum[4];
    __shared__ unsigned long long acc_s[16];
    __shared__ int best_s;
    const int p = blockIdx.x, tid = threadIdx.x;
    const unsigned long long key = lv.seed_key[p];
    if (!key) return;                                       // no finite bound anywhere: no threshold, every tile is scored
    const int it = (int)(key >> 6) & 0xFF, t1 = (int)key & 63, R = t1 >> 3, Cx = t1 & 7;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const int nbt = (nx + 3) >> 2, nbq4 = ((nbt + 3) >> 2) << 2;
    const int gp = lv.tmax << 2;
    const int K = lv.kcount[p * lv.ntheta + it];
    const int* __restrict__ pcl = lv.pcells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    const int* __restrict__ cl = lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    if (tid < 4) csum[tid] = 0u;
    if (tid < 16) acc_s[tid] = 0ull;
    __syncthreads();
    {
        const __amdgpu_buffer_rsrc_t rg2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(lv.gmin2 + (size_t)p * gp * gp), (short)0, (int)((size_t)gp * gp * sizeof(uint32_t)), 0x00020000);
        unsigned s4[4] = {0u, 0u, 0u, 0u};
        for (int k = tid; k < K; k += SEED_THREADS) {
            const int off = pcl[k];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cy = 2 * R + (c >> 1), cx = 2 * Cx + (c & 1);
                if (cy < nbt && cx < nbt) s4[c] += __builtin_amdgcn_raw_buffer_load_b32(rg2, (cy * gp + cx) * 4 + off, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned t = row16_sum_u32(s4[c]);
            if ((tid & 15) == 0 && t) atomicAdd(&csum[c], t);
        }
    }
    __syncthreads();
    const double inv = 1.0 / lv.cost_scale;
    if (tid == 0) {
        const double* __restrict__ pm = lv.tile_pmax + (size_t)p * nbt * nbq4;
        int best = -1;
        double ub = -INFINITY;
        for (int c = 0; c < 4; ++c) {
            const int cy = 2 * R + (c >> 1), cx = 2 * Cx + (c & 1);
            if (cy >= nbt || cx >= nbt) continue;
            const double u = (-(((double)csum[c] * 4096.0) * inv) + pm[cy * nbq4 + cx]) + 1e-9;
            if (best < 0 || u > ub) { ub = u; best = c; }
        }
        best_s = best < 0 ? 0 : best;
    }
    __syncthreads();
    const int sby = 2 * R + (best_s >> 1), sbx = 2 * Cx + (best_s & 1);
    {
        const size_t image = (size_t)lv.fmax * lv.fpitch;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(lv.field + (size_t)p * image), (short)0, (int)(image * sizeof(uint32_t)), 0x00020000);
        unsigned long long a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0ull;
        for (int k = tid; k < K; k += SEED_THREADS) {
            const int off = cl[k] * 4;
            u32x4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((4 * sby + r) * lv.fpitch + 4 * sbx) * 4 + off, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[r * 4 + e] += v[r][e];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned long long t = row16_sum_u64(a[i]);
            if ((tid & 15) == 0 && t) atomicAdd(&acc_s[i], t);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
        double val = -INFINITY;
        for (int r = 0; r < 4; ++r)
            for (int e = 0; e < 4; ++e) {
                const int dy = 4 * sby + r, dx = 4 * sbx + e;
                if (dy >= nx || dx >= nx) continue;
                const double sc = (-((double)acc_s[r * 4 + e] * inv) + pr[dy * nx + dx]) + pr[npose + dy * nx + dx];
                if (!isnan(sc)) val = fmax(val, sc);
            }
        if (val > -INFINITY) lv.bnb_best[p] = order_bits(val);
    }
}

// Level 2: the children of the level-1 tiles that reach the threshold get their gmin2 bounds (every other 4 x 4 tile: -inf);
// then the theta's best child is scored exactly when its bound still reaches the particle's best score, which tightens the
// threshold k_exact_select works with.  bnb_best rises while the kernel runs; the reads are racy on purpose: a tile or a
// seed that is skipped because of a fresher value has a bound below the FINAL best, so neither the final bnb_best (the
// maximum over all candidate seeds) nor the set of tiles k_exact_select keeps depends on the timing.
__global__ __launch_bounds__(64) void k_bound2(Slam2dLevel lv, int P) {
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int p = (slot / lv.ntheta) * 8 + xcd, it = slot % lv.ntheta;
    if (p >= P) return;
    const int lane = threadIdx.x;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const int nbt = (nx + 3) >> 2, nbq4 = ((nbt + 3) >> 2) << 2, nt = nbt * nbq4;
    const int gp = lv.tmax << 2;
    const double thr = unorder_bits(__hip_atomic_load(&lv.bnb_best[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - SLAM2D_BNB_MARGIN;
    double* __restrict__ bnd = lv.bounds + ((size_t)p * lv.ntheta + it) * nt;
    for (int t = lane; t < nt; t += WAVE) bnd[t] = -INFINITY;
    const double u1 = lv.bounds1[((size_t)p * lv.ntheta + it) * 64 + lane];
    unsigned long long todo = __ballot(u1 >= thr);
    if (!todo) return;
    const int K = lv.kcount[p * lv.ntheta + it];
    const int* __restrict__ pcl = lv.pcells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    const int* __restrict__ cl = lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
    const double* __restrict__ pm = lv.tile_pmax + (size_t)p * nbt * nbq4;
    const __amdgpu_buffer_rsrc_t rg2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.gmin2 + (size_t)p * gp * gp), (short)0, (int)((size_t)gp * gp * sizeof(uint32_t)), 0x00020000);
    const double inv = 1.0 / lv.cost_scale;
    int pre[8];
    tile_prefetch<8>(cl, K, pre);                           // for the seed below
    double ub = -INFINITY;
    int seed = -1;
    while (todo) {
        const int t1 = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int R = t1 >> 3, Cx = t1 & 7;
        const double u2 = children_bounds(lv, rg2, pcl, K, R, Cx, pm, nbt, nbq4, gp, inv);
        const int c = lane >> 4, cy = 2 * R + (c >> 1), cx = 2 * Cx + (c & 1);
        if ((lane & 15) == 0 && cy < nbt && cx < nbt) bnd[cy * nbq4 + cx] = u2;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {                    // (-inf for a child outside the cube)
            const double u = readlane_f64(u2, 16 * cc);
            if (u > ub) { ub = u; seed = (2 * R + (cc >> 1)) * nbq4 + 2 * Cx + (cc & 1); }
        }
    }
    if (seed < 0) return;
    if (ub < unorder_bits(__hip_atomic_load(&lv.bnb_best[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
    const int sby = seed / nbq4, sbx = seed - sby * nbq4;
    const int dy = 4 * sby + (lane >> 4);
    const bool leader = (lane & 15) == 0 && dy < nx;
    const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
    double prv[4], ptw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int qq = min(dy, nx - 1) * nx + min(4 * sbx + e, nx - 1);
        prv[e] = pr[qq]; ptw[e] = pr[npose + qq];
    }
    const size_t image = (size_t)lv.fmax * lv.fpitch;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.field + (size_t)p * image), (short)0, (int)(image * sizeof(uint32_t)), 0x00020000);
    unsigned long long acc[4];
    tile_exact<8>(lv, rsrc, cl, K, sby, sbx, pre, acc);
    double val = -INFINITY;
    if (leader) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * sbx + e < nx) {
                const double sc = (-((double)acc[e] * inv) + prv[e]) + ptw[e];
                if (!isnan(sc)) val = fmax(val, sc);
            }
    }
    val = fmax(val, __shfl_xor(val, 16));
    val = fmax(val, __shfl_xor(val, 32));
    if (lane == 0 && val > -INFINITY) atomicMax(&lv.bnb_best[p], order_bits(val));
}

// One block per particle: the tiles whose bound reaches bnb_best - SLAM2D_BNB_MARGIN, exactly, then the selection
// (np.argmax / np.random.choice / confidence, :133-143).  Everything serial in a single wave is slow here (an fp64
// exp is ~0.25 us when no other wave hides its latency, a vector-memory instruction costs the CU 16 clocks), so
// every stage is spread over the block and the tile loop issues as few memory instructions as it can:
//   scan     thread = consecutive (theta, tile) bounds; block-wide exclusive scan of the survivor counts -> an
//            ascending list of surviving tiles (= cube order of theta, then tile row, then tile column); meanwhile the
//            particle's cell lists are staged in LDS (when they fit)
//   tiles    one tile per wave, 16 waves at a time: lane = pose row x 1/16 of the cell list, 16-byte field loads,
//            DPP row sums; 16 lanes add the priors and store the scores (level->cube and LDS)
//   max      thread = scored pose -> block reduction -> the maximum M (np.argmax order: first NaN, else largest,
//            lowest index)
//   exp      thread = scored pose: exp(score - M) -> LDS; DPP row sums per tile; the first tile of every theta adds
//            up the theta's tiles
//   select   wave 0: cdf over theta, then over the pose rows of the chosen theta, then along the row
// More than XS_TILES surviving tiles (rare) are handled in passes; the row walk then re-reads the cube.
#define XS_THREADS 1024
#define XS_TILES 256
#define XS_MAX_PER 32
#define XS_CELLS 8192                // cell-list entries staged in LDS (ntheta * kmax: config 2 6480, reference 5400)
#define XS_SPLIT_MAX 8
#ifndef XS_SPLIT_STAGE_CELLS
#define XS_SPLIT_STAGE_CELLS 1       // split blocks stage all cell lists in LDS behind the scan (0: each tile's list from global memory)
#endif
#define XS_SPLIT_LIGHT 4             // blocks per particle that work when one list pass holds all surviving tiles
#define SLAM2D_BNB_MAX_THETA 256
// SPLIT: nsplit blocks per particle (all of them on the particle's XCD, for the field's sake).  Every block scans the
// particle's bounds (the same list comes out everywhere) and scores its share of the surviving tiles -- with one block
// per particle 64 of the 256 CUs worked and the particle with the most tiles (93 against a median of 27 at config 2) set
// the kernel's time.  The scores go to level->cube with write-through (sc1) stores; every wave drains its stores, the
// block takes a ticket from the particle's arrival counter (lv.sync: zero before the launch, put back to zero by the
// last arriver), and the block that draws the last ticket runs the rest -- maximum, exp, per-theta sums, selection -- over
// ALL the particle's tiles, reading the scores back with sc1 loads (MI355X_MICROARCH.md, "Inter-workgroup visibility":
// sc1 stores + a vmcnt(0) drain per storing wave + a relaxed agent-scope counter on the producer side, sc1 loads on the
// consumer side; nothing depends on where the blocks run).  The arithmetic of those stages and its order are the same
// as with one block, so the results are bit-identical whatever nsplit is.
__device__ __forceinline__ void store_score(double* dst, const double v, const bool through) {
    if (through) __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = v;
}
__device__ __forceinline__ double load_score(const double* src, const bool through) {
    if (through) return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *src;
}
template <bool SPLIT>
__global__ __launch_bounds__(XS_THREADS) void k_exact_select(Slam2dLevel lv, const double* __restrict__ est, int estride,
                                                             const double* __restrict__ uniform, Slam2dMatch* out, int P, int nsplit) {
    __shared__ int list_s[XS_TILES];                   // theta << 8 | tile, ascending
    __shared__ double sc_s[XS_TILES][16];              // a tile's scores (row-major 4 x 4, -inf = no pose), then exp(score - M)
    __shared__ double tsum_s[XS_TILES], tmax_s[XS_TILES];
    __shared__ int targ_s[XS_TILES], tnan_s[XS_TILES];
    __shared__ double S_s[SLAM2D_BNB_MAX_THETA];       // sum over the theta's scored poses of exp(score - M)
    __shared__ int kc_s[SLAM2D_BNB_MAX_THETA], j0_s[SLAM2D_BNB_MAX_THETA], jn_s[SLAM2D_BNB_MAX_THETA];
    __shared__ int cells_s[XS_CELLS];
    __shared__ int wtot_s[XS_THREADS / WAVE];
    __shared__ double wbv_s[XS_THREADS / WAVE];
    __shared__ int wbi_s[XS_THREADS / WAVE], wbn_s[XS_THREADS / WAVE];
    __shared__ double M_s[2];                          // [0] running maximum, [1] the one before this pass
    __shared__ int Mi_s[2];                            // its flat index, nan flag
    __shared__ int last_s;
    const int tid = threadIdx.x;
    int p = blockIdx.x, part = 0;
    if constexpr (SPLIT) {                             // block b runs on XCD b % 8: a particle's blocks share that XCD's L2
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        p = (slot / nsplit) * 8 + xcd; part = slot % nsplit;
        if (p >= P) return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nx = 2 * lv.ncell + 1, npose = nx * nx;
    const int nbt = (nx + 3) >> 2, nbq4 = ((nbt + 3) >> 2) << 2, nt = nbt * nbq4;
    const int ntot = lv.ntheta * nt;
    const double thr = unorder_bits(lv.bnb_best[p]) - SLAM2D_BNB_MARGIN;
    const double* __restrict__ bnd = lv.bounds + (size_t)p * ntot;
    // (what the selection tail needs from global memory is fetched now: a load there is a microsecond of one wave's serial chain)
    const double u_pre = uniform != nullptr ? uniform[p] : 0.0;
    const double th_pre = (tid & 63) < lv.ntheta ? lv.thetas[tid & 63] : 0.0;       // (lane i keeps angle i: the chosen one comes by v_readlane)
    const double ex = est[(size_t)p * estride], ey = est[(size_t)p * estride + 1], eth = est[(size_t)p * estride + 2];
    DBG_CLOCK(8, p == 0);
    // ---- scan ----
    const int per = (ntot + XS_THREADS - 1) / XS_THREADS;          // <= XS_MAX_PER (checked by the host)
    const int g0 = tid * per;
    unsigned keepbits = 0u;
    bool look = g0 < ntot;
    if (look && lv.theta_umax && lv.bnb == 1) {        // only the angles whose largest bound reaches the threshold can hold a surviving tile
        const double* __restrict__ um = lv.theta_umax + (size_t)p * lv.ntheta;
        look = false;
        for (int it = g0 / nt; it <= min(g0 + per - 1, ntot - 1) / nt; ++it) look |= um[it] >= thr;
    }
    if (look)
        for (int i = 0; i < per; ++i) {
            const int g = g0 + i;
            if (g < ntot && bnd[g] >= thr) keepbits |= 1u << i;    // (padding tiles hold -inf)
        }
    // (a block that scores a quarter of the tiles reads the few lists it needs from global memory instead of staging them all)
    const bool cells_in_lds = (!SPLIT || XS_SPLIT_STAGE_CELLS) && lv.ntheta * lv.kmax <= XS_CELLS;
    if (cells_in_lds) {
        const int* __restrict__ call = lv.cells + (size_t)p * lv.ntheta * lv.kmax;
        for (int i = tid; i < lv.ntheta * lv.kmax; i += XS_THREADS) cells_s[i] = call[i];
    }
    const int cnt = __popc(keepbits);
    const int incl = wave_scan_incl_i32(cnt);
    if (lane == WAVE - 1) wtot_s[wave] = incl;
    if (tid < lv.ntheta) { kc_s[tid] = lv.kcount[p * lv.ntheta + tid]; S_s[tid] = 0.0; }
    if (tid == 0) { M_s[0] = -INFINITY; Mi_s[0] = INT_MAX; Mi_s[1] = 0; }
    __syncthreads();
    int off = incl - cnt, n_all = 0;
    for (int w2 = 0; w2 < XS_THREADS / WAVE; ++w2) { const int c = wtot_s[w2]; if (w2 < wave) off += c; n_all += c; }
    DBG_CLOCK(9, p == 0);
    const size_t image = (size_t)lv.fmax * lv.fpitch;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(lv.field + (size_t)p * image), (short)0, (int)(image * sizeof(uint32_t)), 0x00020000);
    const double* __restrict__ pr = lv.prior + (size_t)p * 2 * npose;
    const double inv = 1.0 / lv.cost_scale;
    // the pass's list: surviving tiles [base, base + XS_TILES) in ascending (theta, tile) order
    auto build_list = [&](const int base) {
        int o = off;
        for (int i = 0; i < per; ++i)
            if ((keepbits >> i) & 1u) {
                if (o >= base && o < base + XS_TILES) { const int g = g0 + i; list_s[o - base] = ((g / nt) << 8) | (g % nt); }
                ++o;
            }
    };
    // exact scores of the pass's tiles j = first, first + stride, ... (one tile per wave at a time): lane = pose row x 1/16
    // of the cell list, 16-byte field loads, DPP row sums; 16 lanes add the priors and store the scores
    auto score_tiles = [&](const int n, const int first, const int stride) {
        for (int j = first; j < n; j += stride) {
            const int ent = list_s[j];
            const int it = ent >> 8, t = ent & 255;
            const int by = t / nbq4, bx = t - by * nbq4;
            const int K = kc_s[it];
            const int* __restrict__ cl = lv.cells + ((size_t)p * lv.ntheta + it) * lv.kmax;
            int pre[12];
            if (cells_in_lds) {
                const int* cs = cells_s + it * lv.kmax;
                const int sl = lane & 15;
#pragma unroll
                for (int i = 0; i < 12; ++i) pre[i] = sl + 16 * i < K ? cs[sl + 16 * i] * 4 : 0x7ffffff0;
            } else {
                tile_prefetch<12>(cl, K, pre);
            }
            // the 16 scoring lanes (16 r + e: pose row r, column e of the tile) fetch their pose's priors now
            const int r = lane >> 4, e = lane & 15;
            const int dy = 4 * by + r, dx = 4 * bx + e;
            const bool scorer = e < 4 && dy < nx && dx < nx;
            const int qq = dy * nx + dx;
            double prv = 0.0, ptw = 0.0;
            if (scorer) { prv = pr[qq]; ptw = pr[npose + qq]; }
            unsigned long long acc[4];
            tile_exact<12>(lv, rsrc, cl, K, by, bx, pre, acc);
            double sc = -INFINITY;
            if (e < 4) {
                const unsigned long long a = e == 0 ? acc[0] : e == 1 ? acc[1] : e == 2 ? acc[2] : acc[3];
                if (scorer) {
                    sc = (-((double)a * inv) + prv) + ptw;                                    // :131, as k_sweep
                    store_score(&lv.cube[((size_t)p * lv.ntheta + it) * npose + qq], sc, SPLIT);
                }
                if constexpr (!SPLIT) sc_s[j][r * 4 + e] = sc;
            }
            if constexpr (!SPLIT) {
                // the tile's maximum in np.argmax order (first NaN, else largest, lowest index): the scoring lanes 16 r + e
                // ascend with the flat pose index, so "lowest index" = lowest lane.  Quad butterflies, then the four rows.
                double mq = scorer && !isnan(sc) ? sc : -INFINITY;
                mq = fmax(mq, __longlong_as_double((long long)dpp_u64<0xB1>((unsigned long long)__double_as_longlong(mq))));
                mq = fmax(mq, __longlong_as_double((long long)dpp_u64<0x4E>((unsigned long long)__double_as_longlong(mq))));
                const double tmx = fmax(fmax(readlane_f64(mq, 0), readlane_f64(mq, 16)), fmax(readlane_f64(mq, 32), readlane_f64(mq, 48)));
                const unsigned long long nanm = __ballot(scorer && isnan(sc));
                const unsigned long long eqm = nanm ? nanm : __ballot(scorer && sc == tmx);
                if (lane == 0) {
                    int arg = INT_MAX;
                    if (eqm) {
                        const int l = __ffsll((long long)eqm) - 1;
                        arg = it * npose + (4 * by + (l >> 4)) * nx + 4 * bx + (l & 15);
                    }
                    tmax_s[j] = nanm ? NAN : tmx; targ_s[j] = arg; tnan_s[j] = nanm ? 1 : 0;
                }
            }
        }
    };
    if constexpr (SPLIT) {
        // Round 4: how many of the particle's blocks work depends on how many tiles survived.  The launch carries up to
        // XS_SPLIT_MAX blocks per particle; a particle with few surviving tiles (one list pass: the tracked case, 30-180 tiles)
        // keeps the four that measured best there, the others leave after their scan -- every block finds the same n_all, so
        // all agree on who stays and the ticket counts to that number.  A particle with many (a scan that does not fit its map:
        // hundreds to a thousand) uses all of them: k_exact_select 119 -> 92 us and 132 -> 92 us on the inputs of
        // bench.py's config2_displaced / config2_worst.  The scores and the order of every later stage do not depend on the
        // split, so results are the same bits either way.
        const int nsplit_all = nsplit;
        nsplit = n_all > XS_TILES ? nsplit_all : min(nsplit_all, XS_SPLIT_LIGHT);
        if (part >= nsplit) return;
        // ---- this block's share of the tiles, all passes; then the ticket ----
        for (int base = 0; base < n_all; base += XS_TILES) {           // (block-uniform)
            build_list(base);
            __syncthreads();
            score_tiles(min(XS_TILES, n_all - base), part + nsplit * wave, nsplit * (XS_THREADS / WAVE));
            __syncthreads();                                           // list_s is rebuilt by the next pass / the tail
        }
        // Publish / consume across blocks, the form MI355X_MICROARCH.md ("inter-workgroup visibility": valid forms) lists as "8-B agent
        // atomics both sides": the scores leave as relaxed agent-scope 8-byte atomic stores (global_store ... sc1: past this CU's L1,
        // the line dropped from the XCD's L2), every storing wave drains them (s_waitcnt vmcnt(0)), the block takes its ticket with an
        // agent-scope atomic, and the last block reads the scores back with relaxed agent-scope atomic loads (sc1: never served from
        // an L1).  No agent-scope fence on either side: a release (buffer_wbl2) costs 1.7-6.5 us per block and an acquire
        // (buffer_inv) 1.7 us (the guide's price list; round 2 measured this kernel at 134 us with them), and neither is needed when
        // both sides bypass the caches they would write back / invalidate.  What the C++ abstract machine is not told is kept in
        // order by the signal fences (compiler only) around the barrier.  tests/test_gpu_parity.py::test_split_exact_select_stress
        // runs 1 000 launches with every block count against the one-block path, bit for bit.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every storing wave drains its write-through stores
        __atomic_signal_fence(__ATOMIC_SEQ_CST);                       // (no instruction: the compiler may not move the score stores
        __syncthreads();                                               //  below the ticket, nor the read-back loads above it)
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(&lv.sync[p * SLAM2D_SYNC_WORDS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == (unsigned)(nsplit - 1);
            if (last) __hip_atomic_store(&lv.sync[p * SLAM2D_SYNC_WORDS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            last_s = last;
        }
        __syncthreads();
        if (!last_s) return;
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
    int n = 0;
    for (int base = 0; base < n_all; base += XS_TILES) {           // (block-uniform)
        n = min(XS_TILES, n_all - base);
        if (!(SPLIT && n_all <= XS_TILES)) build_list(base);       // (a single pass: the list of the scoring phase is still there)
        if (tid < lv.ntheta) jn_s[tid] = 0;
        __syncthreads();
        DBG_CLOCK(10, p == 0 && base == 0);
        // ---- tiles ----
        if constexpr (!SPLIT) {
            score_tiles(n, wave, XS_THREADS / WAVE);
        } else {
            // every tile of the pass back from the cube (sc1 loads: the stores were write-through), then the tiles' maxima
            for (int i0 = 0; i0 < n * 16; i0 += XS_THREADS) {
                const int idx = i0 + tid;
                if (idx < n * 16) {
                    const int ent = list_s[idx >> 4], it = ent >> 8, t = ent & 255;
                    const int by = t / nbq4, bx = t - by * nbq4;
                    const int dy = 4 * by + ((idx >> 2) & 3), dx = 4 * bx + (idx & 3);
                    double sc = -INFINITY;
                    if (dy < nx && dx < nx) sc = load_score(&lv.cube[((size_t)p * lv.ntheta + it) * npose + dy * nx + dx], true);
                    sc_s[idx >> 4][idx & 15] = sc;
                }
            }
            __syncthreads();
            if (tid < n) {                                  // np.argmax order inside the tile: first NaN, else largest, lowest index
                const int ent = list_s[tid], it = ent >> 8, t = ent & 255;
                const int by = t / nbq4, bx = t - by * nbq4;
                double tmx = -INFINITY;
                int arg = INT_MAX, nan = 0;
                for (int i = 0; i < 16; ++i) {
                    const int dy = 4 * by + (i >> 2), dx = 4 * bx + (i & 3);
                    if (dy >= nx || dx >= nx) continue;
                    const double sc = sc_s[tid][i];
                    const int q = it * npose + dy * nx + dx;
                    if (isnan(sc)) { if (!nan) { nan = 1; arg = q; } }
                    else if (!nan && (arg == INT_MAX || sc > tmx)) { tmx = sc; arg = q; }
                }
                tmax_s[tid] = nan ? NAN : tmx; targ_s[tid] = arg; tnan_s[tid] = nan;
            }
        }
        __syncthreads();
        DBG_CLOCK(11, p == 0 && base == 0);
        // ---- maximum ----
        {
            if (wave < XS_TILES / WAVE) {                // thread = tile
                Best bme{-INFINITY, INT_MAX, 0};
                if (tid < n && targ_s[tid] != INT_MAX) bme = Best{tmax_s[tid], targ_s[tid], tnan_s[tid]};
                bme = wave_best_fast(bme);
                if (lane == 0) { wbv_s[wave] = bme.v; wbi_s[wave] = bme.i; wbn_s[wave] = bme.nan; }
            }
            __syncthreads();
            if (tid == 0) {
                Best rbest{M_s[0], Mi_s[0], Mi_s[1]};
                M_s[1] = M_s[0];
                for (int w2 = 0; w2 < XS_TILES / WAVE; ++w2) {
                    Best c{wbv_s[w2], wbi_s[w2], wbn_s[w2]};
                    if (c.i != INT_MAX && better(c, rbest)) rbest = c;
                }
                M_s[0] = rbest.v; Mi_s[0] = rbest.i; Mi_s[1] = rbest.nan;
            }
            __syncthreads();
        }
        const double M = M_s[0];
        DBG_CLOCK(29, p == 0 && base == 0);
        // ---- exp ----
        if (base > 0 && tid < lv.ntheta) S_s[tid] *= exp(M_s[1] - M);       // earlier passes: rescale to the new maximum
        for (int i0 = 0; i0 < n * 16; i0 += XS_THREADS) {                   // (block-uniform trip count)
            const int idx = i0 + tid;
            double ev = 0.0;
            if (idx < n * 16) {
                const double sc = sc_s[idx >> 4][idx & 15];
                ev = sc == -INFINITY ? 0.0 : exp(sc - M);
                sc_s[idx >> 4][idx & 15] = ev;
            }
            const unsigned long long tb = row16_sum_f64(ev);
            if (idx < n * 16 && (idx & 15) == 0) tsum_s[idx >> 4] = __longlong_as_double((long long)tb);
        }
        __syncthreads();
        DBG_CLOCK(30, p == 0 && base == 0);
        if (tid < n) {                                  // the first tile of a theta adds up the theta's tiles, in list order
            const int th = list_s[tid] >> 8;
            if (tid == 0 || (list_s[tid - 1] >> 8) != th) {
                double a = 0.0;
                int k = tid;
                for (; k < n && (list_s[k] >> 8) == th; ++k) a += tsum_s[k];
                S_s[th] += a;
                j0_s[th] = tid; jn_s[th] = k - tid;
            }
        }
        __syncthreads();
        DBG_CLOCK(12, p == 0 && base == 0);
    }
    if (wave != 0) return;
    // ---- selection (wave 0) ----
    const double M = M_s[0];
    const int nW = lv.ntheta;
    const int tper = (nW + WAVE - 1) / WAVE;
    const int w0 = lane * tper, w1 = min(nW, w0 + tper);
    double mine = 0.0;
    for (int w = w0; w < w1; ++w) mine += S_s[w];
    const double cinc = wave_scan_incl_f64(mine);
    const double total = readlane_f64(cinc, WAVE - 1);
    int pick = Mi_s[0];
    if (uniform != nullptr && !isnan(total)) {
        // np.random.choice(n, 1, p): first index whose normalised cdf exceeds u (:137-138); the poses that were not
        // scored carry < 1e-8 of the mass
        const double target = u_pre * total;
        const unsigned long long ahead = __ballot(cinc > target);
        const int lsel = ahead ? __ffsll((long long)ahead) - 1 : WAVE - 1;
        double run = readlane_f64(cinc - mine, lsel);
        const int s0 = min(lsel * tper, nW - 1), s1 = max(s0 + 1, min(nW, lsel * tper + tper));
        int it = s1 - 1;
        for (int w = s0; w < s1; ++w) {                           // wave-uniform loop
            const double t = S_s[w];
            if (run + t > target || w == s1 - 1) { it = w; break; }
            run += t;
        }
        while (it > 0 && !(S_s[it] > 0.0)) --it;                 // rounding fallback landed on a theta without a scored pose
        const int by = lane >> 2, rr = lane & 3;
        const bool in_lds = n_all <= XS_TILES;
        const int ja = in_lds ? j0_s[it] : 0, jb = in_lds ? ja + jn_s[it] : 0;
        double lsum = 0.0;
        bool has = false;
        if (in_lds) {
            // the theta's tiles are in LDS: lane = pose row dy, its poses in dx order (tiles ascend with bx)
            if (lane < nx)
                for (int j = ja; j < jb; ++j)
                    if ((list_s[j] & 255) / nbq4 == by) {
                        has = true;
#pragma unroll
                        for (int e = 0; e < 4; ++e) lsum += sc_s[j][rr * 4 + e];
                    }
        } else {
            const double* c = lv.cube + ((size_t)p * lv.ntheta + it) * npose;
            const double* bt = bnd + (size_t)it * nt;
            if (lane < nx)
                for (int bx = 0; bx < nbt; ++bx)
                    if (bt[by * nbq4 + bx] >= thr) {
                        has = true;
                        for (int e = 0; e < 4 && 4 * bx + e < nx; ++e) lsum += exp(load_score(&c[lane * nx + 4 * bx + e], SPLIT) - M);
                    }
        }
        const double linc = wave_scan_incl_f64(lsum);
        const unsigned long long hit = __ballot(has && run + linc > target);
        const unsigned long long have = __ballot(has);
        if (have) {
            const int l2 = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)have);
            double r2 = run + readlane_f64(linc - lsum, l2);
            int found = -1, last = -1;
            if (lane == l2) {
                if (in_lds) {
                    for (int j = ja; j < jb && found < 0; ++j) {
                        const int t = list_s[j] & 255;
                        if (t / nbq4 == by) {
                            const int bx = t % nbq4;
                            for (int e = 0; e < 4 && 4 * bx + e < nx; ++e) {
                                last = it * npose + lane * nx + 4 * bx + e;
                                r2 += sc_s[j][rr * 4 + e];
                                if (r2 > target) { found = last; break; }
                            }
                        }
                    }
                } else {
                    const double* c = lv.cube + ((size_t)p * lv.ntheta + it) * npose;
                    const double* bt = bnd + (size_t)it * nt;
                    for (int bx = 0; bx < nbt && found < 0; ++bx)
                        if (bt[by * nbq4 + bx] >= thr)
                            for (int e = 0; e < 4 && 4 * bx + e < nx; ++e) {
                                last = it * npose + lane * nx + 4 * bx + e;
                                r2 += exp(load_score(&c[lane * nx + 4 * bx + e], SPLIT) - M);
                                if (r2 > target) { found = last; break; }
                            }
                }
                if (found < 0) found = last;                      // rounding fallback: the row's last scored pose
            }
            pick = __builtin_amdgcn_readlane(found, l2);
        }
    }
    const int it_sel = __builtin_amdgcn_readfirstlane(pick / npose);
    const double th_sel = lv.ntheta <= WAVE ? readlane_f64(th_pre, it_sel) : lv.thetas[it_sel];
    if (lane == 0) {
        Slam2dMatch m;
        const int it = pick / npose, rem = pick - it * npose;
        const int iy = rem / nx, ix = rem - iy * nx;
        m.x = ex + (double)(ix - lv.ncell) * lv.step;                               // :142-143
        m.y = ey + (double)(iy - lv.ncell) * lv.step;
        m.theta = eth + th_sel;
        m.confidence = exp(M) * total;                                              // :141
        m.log_confidence = M + log(total);
        m.best_score = M;
        m.pick = pick;
        m.argmax = Mi_s[0];
        out[p] = m;
        if (lv.arrive) __hip_atomic_fetch_add(lv.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (Slam2dScan.match_seq)
    }
    DBG_CLOCK(13, p == 0);
}

// ------------------------------------------------------------------------------------
// K4  weights                                        (Algorithm/FastSlam.py:30-48,135)
// ------------------------------------------------------------------------------------
// One 256-thread block.  `exchange`: the block runs beside the update blocks of the same launch (slam2d_scan_commit,
// slam2d_grid_update_weights), which may still raise bits: they are taken with an atomic exchange, so a bit raised
// after it stays in flags and is reported with the next scan instead of being lost.
// updateEstimatedPose of one particle (Algorithm/FastSlam.py:77-106): estimate = previous matched pose turned by the raw odometry's
// rotation; cos / sin of the expected moving direction (NaN: no direction known)
__device__ __forceinline__ void prior_one(const double* prev, const double* heading, const int p, const double raw_theta,
                                          const double prev_raw_theta, const int has_turn, const double raw_turn, double* est, double* psi_cs) {
    est[3 * p + 0] = prev[3 * p + 0];                                               // :79
    est[3 * p + 1] = prev[3 * p + 1];
    est[3 * p + 2] = prev[3 * p + 2] + raw_theta - prev_raw_theta;                  // :78
    double c = NAN, s = NAN;
    const double h = heading[p];
    if (has_turn && !isnan(h)) {                                                    // :89-95
        const double psi = h + raw_turn;
        c = cos(psi); s = sin(psi);
    }
    psi_cs[2 * p] = c; psi_cs[2 * p + 1] = s;
}
struct WeightsJob {
    double* logw; const double* logconf; int cstride; int N; double* w; double* stats; uint32_t* flags; uint32_t* flag_snapshot;
    // slam2d_scan_commit: the same block first does the scan's bookkeeping (k_post_match's work) for all particles
    const Slam2dMatch* fine; const Slam2dMatch* coarse; double* prev; double* heading; double* report;
    double* part;            // sharded filters: only the rank-local half (k_weights_local's work), the collective follows
    uint32_t abort_mask;     // slam2d_scan_commit: fault bits of the match that make the WHOLE launch a no-op (see there)
    const uint32_t* abort_flags; int abort_n;   // slam2d_groups_commit: the abort is decided over the fault bits of ALL groups (NULL: flags[0 .. N))
    // slam2d_scan_commit_next: the NEXT scan's pose prior (k_prior's work) behind this scan's bookkeeping, same thread per particle
    double* next_est; double* next_psi; double next_raw_theta, next_prev_raw_theta, next_raw_turn; int next_has_turn;
    // slam2d_groups_* with Slam2dScan.d_norm_sync (round 5): the groups' normaliser blocks merge among themselves on the device --
    // no stream, no event between the groups (normaliser_wait / normaliser_arrive)
    uint32_t* nsync = nullptr; int ngroups = 0, gidx = 0;
    const double* parts_all = nullptr; double* logw_all = nullptr; int n_all = 0; double total = 0.0; double* w_all = nullptr; double* stats_all = nullptr;
    // the scan's report pushed to the host by the block that finishes it (Slam2dScan.h_pack / h_seq, ABI 16): no copy, no event
    const double* pack_d = nullptr; double* pack_h = nullptr; int pack_n = 0; uint32_t* h_seq = nullptr; uint32_t seq = 0u;
    // the NEXT scan's ranges pulled from pinned host memory by the bookkeeping block, beside its prior (Slam2dScan.h_next_ranges)
    const double* pull_src = nullptr; double* pull_dst = nullptr; int pull_n = 0;
};
// Whoever finishes a scan for all groups -- the block that merged the normaliser, or the last group's block 0 of a voided scan --
// copies the scan's pack (report, weights, variance, fault-bit snapshot: device memory the groups' blocks have written and
// released) into the caller's pinned host buffer and publishes the scan's sequence number behind it, system scope: the host
// polls that word (slam2d_host_wait_seq) instead of waiting for a copy engine and an event (~15 us per scan).
__device__ __forceinline__ void publish_pack(const WeightsJob& wj) {
    if (!wj.h_seq) return;
    __syncthreads();
    for (int i = threadIdx.x; i < wj.pack_n; i += blockDim.x) wj.pack_h[i] = wj.pack_d[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(wj.h_seq, wj.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// pose / heading / log-weight bookkeeping of one particle after its match (Algorithm/FastSlam.py:110-117,134-135)
__device__ __forceinline__ void post_match_one(const Slam2dMatch* __restrict__ fine, const Slam2dMatch* __restrict__ coarse, const int p,
                                               double* prev, double* heading, double* logw, double* report) {
    const double x = fine[p].x, y = fine[p].y;
    const double mx = x - prev[3 * p], my = y - prev[3 * p + 1];                    // :110-111
    const double move = sqrt(mx * mx + my * my);
    double h = NAN;
    if (move != 0.0) h = my > 0.0 ? acos(mx / move) : -acos(mx / move);             // :113-117
    heading[p] = h;
    prev[3 * p] = x; prev[3 * p + 1] = y; prev[3 * p + 2] = fine[p].theta;          // :134
    logw[p] += coarse[p].log_confidence;                                            // :135
    if (report) {                                      // what the caller downloads once per scan
        report[5 * p] = x; report[5 * p + 1] = y; report[5 * p + 2] = fine[p].theta;
        report[5 * p + 3] = coarse[p].confidence; report[5 * p + 4] = coarse[p].log_confidence;       // :79 (coarse)
    }
}
__device__ __forceinline__ void weights_body(double* logw, const double* __restrict__ logconf, const int cstride, const int N,
                                             double* w, double* stats, uint32_t* flags, uint32_t* flag_snapshot, const bool exchange) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    if (flags)                                         // slam2d_scan_commit: the scan's fault bits move into the report
        for (int i = tid; i < N; i += 256) {
            if (exchange) flag_snapshot[i] = atomicExch(&flags[i], 0u);
            else { flag_snapshot[i] = flags[i]; flags[i] = 0u; }
        }
    double mx = -INFINITY;
    for (int i = tid; i < N; i += 256) {
        double v = logw[i] + (logconf ? logconf[(size_t)i * cstride] : 0.0);
        logw[i] = v;
        mx = fmax(mx, v);
    }
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmax(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    double s = 0.0;
    for (int i = tid; i < N; i += 256) s += exp(logw[i] - mx);
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double total = red[0];
    __syncthreads();
    const double lse = mx + log(total);
    double var = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double wi = exp(logw[i] - mx) / total;                                 // :47-48
        w[i] = wi;
        logw[i] = logw[i] - lse;
        const double d = wi - 1.0 / (double)N;                                       // :34
        var += d * d;
    }
    red[tid] = var;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) { stats[0] = red[0]; stats[1] = lse; }
}
// Sharded normaliser, rank-local half: log-weights += log-confidence, then this rank's
// [max log w, sum exp(lw - max), sum exp(2 (lw - max))].  The three doubles of every rank are
// exchanged by ONE all-gather (24 bytes per rank) and merged by k_weights_merge.
__device__ __forceinline__ void weights_local_body(double* logw, const double* __restrict__ logconf, const int cstride,
                                                   const int N, double* part) {
    __shared__ double red[256];
    __shared__ double red2[256];
    const int tid = threadIdx.x;
    double mx = -INFINITY;
    for (int i = tid; i < N; i += 256) {
        double v = logw[i] + (logconf ? logconf[(size_t)i * cstride] : 0.0);
        logw[i] = v;
        mx = fmax(mx, v);
    }
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmax(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double e = exp(logw[i] - mx);
        s1 += e;
        s2 += e * e;
    }
    red[tid] = s1;
    red2[tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; red2[tid] += red2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { part[0] = mx; part[1] = red[0]; part[2] = red2[0]; }
}
__global__ __launch_bounds__(256) void k_weights_local(double* logw, const double* __restrict__ logconf, int cstride,
                                                       int N, double* part) {
    weights_local_body(logw, logconf, cstride, N, part);
}

// The normaliser of G particle groups WITHOUT a stream of its own (round 5).  Every group's map-update launch has one block that
// folds the scan's confidences into the group's log-weights and leaves the group's partial [max, sum, sum of squares]
// (weights_local_body); the merge over the groups (k_weights_merge's arithmetic, partials in group order: the same bits) used to be a
// launch on a third stream behind an event of every group, and every group's next update waited for an event behind it: two event
// packets between a group's kernels per scan, 6-8 us each on the group's own chain (kernel trace: exact -> update 6.5 us,
// update -> next endpoints 7.5 us of a 126 us step).  Now the blocks settle it among themselves through three kinds of device
// words (Slam2dScan.d_norm_sync, zeroed once): [0] arrivals of this scan, [1] merges completed so far, [2 + g] merges group g's next
// normaliser block has to see.  The block that arrives last merges for all groups and publishes; a group's next block waits for
// that before it touches its log-weights.  Plain stores + release fence before every flag, one acquire fence behind every wait
// (MI355X_MICROARCH.md, "inter-workgroup visibility", the valid producer / consumer forms): placement-independent.  No deadlock:
// a waiting block was enqueued after everything it waits for (slam2d_groups_* issue a whole scan of every group per call).
// All waits are BOUNDED: a producer that never comes -- a dead rank behind the all-gather, a caller that broke the
// whole-scan-per-call order -- ends as the fatal SLAM2D_F_SYNC_TIMEOUT in the group's fault words (word 62 of the sync block
// carries it from the gate kernel), not as a hung GPU.  The bound is word 59 of the sync block in milliseconds (0: 30 s -- round 5's
// fixed 2 s turned ordinary lateness into a fault: a peer rank behind the all-gather that grows its maps, a second process or a
// profiler on the GPU); slam2d_host_wait_seq's bound on the host side should exceed it.
#define SLAM2D_SYNC_DEFAULT_MS 30000u
__device__ __forceinline__ unsigned long long sync_spin_ticks(const uint32_t* nsync) {
    const uint32_t ms = __hip_atomic_load(&nsync[59], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned long long)(ms ? ms : SLAM2D_SYNC_DEFAULT_MS) * 100000ull;                 // (100 MHz wall clock)
}
__device__ __forceinline__ void normaliser_wait(uint32_t* nsync, const int g, uint32_t* flags) {
    if (threadIdx.x == 0) {
        const uint32_t want = nsync[2 + g];
        const unsigned long long t0 = wall_clock64(), SYNC_SPIN_TICKS = sync_spin_ticks(nsync);
        bool late = false;
        while ((int)(__hip_atomic_load(&nsync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > SYNC_SPIN_TICKS) { late = true; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if ((late || __hip_atomic_load(&nsync[62], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && flags) atomicOr(&flags[0], SLAM2D_F_SYNC_TIMEOUT);
    }
    __syncthreads();
}
__device__ __forceinline__ void normaliser_arrive(const WeightsJob& wj) {
    __shared__ int last_s;
    __syncthreads();                                       // every wave's stores of this block (log-weights, the partial) are out
    if (threadIdx.x == 0) {
        wj.nsync[2 + wj.gidx] += 1u;                       // (this group's next block: after THIS scan's merge)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t ticket = __hip_atomic_fetch_add(&wj.nsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (sharded, wj.logw_all == NULL: nobody merges here -- the caller's gate kernel on the normaliser's stream counts the arrivals,
        // the all-gather and slam2d_weights_merge_publish follow there)
        const int last = wj.logw_all != nullptr && ticket == (uint32_t)(wj.ngroups - 1);
        if (last) {
            __hip_atomic_store(&wj.nsync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (nobody arrives for the next scan before the merge below is published)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                      // the other groups' partials and log-weights
        }
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    // k_weights_merge's arithmetic, over the partials in group order
    const double* parts = wj.parts_all;
    double gm = -INFINITY;
    for (int r = 0; r < wj.ngroups; ++r) gm = fmax(gm, parts[3 * r]);
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < wj.ngroups; ++r) {
        const double sc = exp(parts[3 * r] - gm);
        s1 += parts[3 * r + 1] * sc;
        s2 += parts[3 * r + 2] * sc * sc;
    }
    const double lse = gm + log(s1);
    for (int i = threadIdx.x; i < wj.n_all; i += 256) {
        const double lw = wj.logw_all[i];
        wj.w_all[i] = exp(lw - gm) / s1;
        wj.logw_all[i] = lw - lse;
    }
    if (threadIdx.x == 0) { wj.stats_all[0] = s2 / (s1 * s1) - 1.0 / wj.total; wj.stats_all[1] = lse; }
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t gen = __hip_atomic_load(&wj.nsync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&wj.nsync[1], gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    publish_pack(wj);                                      // (behind the word the groups wait for: the host is not on their path)
}
// A voided scan (k_grid_update's abort) has no normaliser: the groups' blocks 0 count themselves in word 60 instead and the last one
// publishes the report (fault-bit snapshot + coarse poses) to the host.
__device__ __forceinline__ void abort_arrive(const WeightsJob& wj) {
    if (!wj.h_seq || !wj.nsync) return;
    __shared__ int last_a;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t ticket = __hip_atomic_fetch_add(&wj.nsync[60], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == (uint32_t)(wj.ngroups - 1);
        if (last) {
            __hip_atomic_store(&wj.nsync[60], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        last_a = last;
    }
    __syncthreads();
    if (last_a) publish_pack(wj);
}

__global__ __launch_bounds__(256) void k_weights(double* logw, const double* __restrict__ logconf, int cstride, int N,
                                                 double* w, double* stats, uint32_t* flags, uint32_t* flag_snapshot) {
    weights_body(logw, logconf, cstride, N, w, stats, flags, flag_snapshot, false);
}

// ------------------------------------------------------------------------------------
// K3  occupancy-grid update                         (Utils/OccupancyGrid.py:127-152)
// ------------------------------------------------------------------------------------
// Beam-major update: the reference's own formulation (:134-152 walks the cells of each beam's spoke).
// Every spoke's cell list is ordered by radial band (SLAM2D_SPOKE_BAND cells of integer radius
// floor(r / unit) per band) and row-major inside a band, with the start of every band tabulated:
// a beam touches the bands up to the one holding range + w/2 -- work ~ touched cells (8 % of the
// W x W window on the Intel log) instead of the whole window -- and consecutive lanes fall on
// consecutive cells of a map row (the band is a short piece of the beam's wedge).  One wave per
// beam; a block takes UPDB_BEAMS adjacent beams of one particle (adjacent wedges share cache lines)
// and all blocks of a particle run on one XCD (block b -> XCD b % 8) so those lines meet in one L2.
// Each window cell belongs to exactly one spoke and each spoke to at most one beam: plain RMW.
// (int)rint(v / unit) without the fp64 division (~70 issue slots on gfx950, and the update needs two per
// cell): v * (1/unit) differs from v / unit by < 4e-16 relative, so the two round to the same integer
// unless the quotient is within 1e-6 of a half-integer -- there the exact division decides.
#define UPDB_BEAMS 4                 // = waves per block
#ifndef UPDB_UNROLL
#define UPDB_UNROLL 4
#endif
#ifndef UPDB_NO_LATTICE
#define UPDB_NO_LATTICE 0            // 1: always the per-cell fp64 index path (A/B builds)
#endif
#ifndef UPDB_MIN_WAVES
#define UPDB_MIN_WAVES 1
#endif
#ifndef UPDB_SKIP_R
#define UPDB_SKIP_R 1
#endif
__global__ __launch_bounds__(256, UPDB_MIN_WAVES) void k_grid_update(Slam2dLidar lid, const Slam2dMap* __restrict__ maps, int P,
                                                           const double* __restrict__ pose, int pstride,
                                                           const double* __restrict__ ranges,
                                                           const int32_t* __restrict__ beam_shift, uint32_t* flags,
                                                           int groups, WeightsJob wj) {
    if (wj.abort_mask) {
        // scan-level abort (slam2d_scan_commit): if the match left one of these fault bits for ANY particle -- a search window
        // outside its map: the host has to grow the map and run the scan again -- nothing of this launch may happen: no map
        // update, no bookkeeping, no weights.  The bits were set by earlier launches and are not cleared here, so every
        // block reads the same.
        bool bad = false;
        const uint32_t* af = wj.abort_flags ? wj.abort_flags : flags;
        const int an = wj.abort_flags ? wj.abort_n : wj.N;
        // (with the device-side gate: a gate that gave up has left SLAM2D_F_SYNC_TIMEOUT in a group's fault word -- the scan is
        // voided like one whose window left a map, its report still reaches the host, which raises at once instead of waiting
        // for a sequence number that would never come)
        const uint32_t am = wj.abort_mask | (wj.nsync ? SLAM2D_F_SYNC_TIMEOUT : 0u);
        for (int i = threadIdx.x; i < an; i += blockDim.x) bad |= (af[i] & am) != 0u;
        if (__syncthreads_or(bad)) {
            if (blockIdx.x == 0 && wj.flag_snapshot)
                for (int i = threadIdx.x; i < wj.N; i += blockDim.x) wj.flag_snapshot[i] = flags[i] | SLAM2D_F_SCAN_VOIDED;
            // ... except that the report receives the COARSE matched poses: the host grows the maps for the fine windows round
            // them (Utils/ScanMatcher_OGBased.py:27 at the fine level) before it issues the scan again
            if (blockIdx.x == 0 && wj.report && wj.coarse)
                for (int i = threadIdx.x; i < wj.N; i += blockDim.x) {
                    wj.report[5 * i] = wj.coarse[i].x; wj.report[5 * i + 1] = wj.coarse[i].y; wj.report[5 * i + 2] = wj.coarse[i].theta;
                }
            if (blockIdx.x == 0) abort_arrive(wj);
            if (blockIdx.x == 0 && wj.nsync && wj.part && !wj.logw_all && threadIdx.x == 0) {
                // a sharded rank (the groups only arrive, the merge follows the all-gather on the normaliser's stream): the gate, the
                // collective and the merge of this scan are enqueued already, on every rank.  The group leaves the VOID partial
                // (sum < 0: no sum of exponentials is) and counts itself in -- not in its own word 2 + g: the scan comes again --
                // so the gate passes, every rank's merge finds the marker, merges nothing and reports the scan as voided
                // (slam2d_weights_merge_publish_report); the ranks that committed keep their partials and gather again.
                wj.part[0] = -INFINITY; wj.part[1] = -1.0; wj.part[2] = 0.0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(&wj.nsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    DBG_CLOCK(58, wj.logw && blockIdx.x == 0);
    if (wj.logw && blockIdx.x == 0) {                      // one extra block: the normaliser, beside the update (one launch less;
        //                                                    block 0, so that it starts with the launch and not as its tail)
        if (wj.nsync) normaliser_wait(wj.nsync, wj.gidx, flags);  // (device-merged groups: the previous scan's merge has reached the log-weights)
        if (wj.fine) {                                     // ... after the scan's bookkeeping (the update blocks take their
            for (int i = threadIdx.x; i < wj.N; i += blockDim.x) {        // poses from the match buffer themselves)
                post_match_one(wj.fine, wj.coarse, i, wj.prev, wj.heading, wj.logw, wj.report);
                if (wj.next_est)                                            // (reads what this very thread has just written)
                    prior_one(wj.prev, wj.heading, i, wj.next_raw_theta, wj.next_prev_raw_theta, wj.next_has_turn, wj.next_raw_turn,
                              wj.next_est, wj.next_psi);
            }
            for (int i = threadIdx.x; i < wj.pull_n; i += blockDim.x) wj.pull_dst[i] = __builtin_nontemporal_load(wj.pull_src + i);
            __syncthreads();
        }
        if (wj.part) {
            if (wj.flags && wj.flag_snapshot)              // (slam2d_groups_commit: the scan's fault bits move into the report)
                for (int i = threadIdx.x; i < wj.N; i += blockDim.x) wj.flag_snapshot[i] = atomicExch(&wj.flags[i], 0u);
            weights_local_body(wj.logw, wj.logconf, wj.cstride, wj.N, wj.part);
            if (wj.nsync) normaliser_arrive(wj);
        }
        else weights_body(wj.logw, wj.logconf, wj.cstride, wj.N, wj.w, wj.stats, wj.flags, wj.flag_snapshot, true);
        DBG_CLOCK(59, true);
        return;
    }
    const int bidx = blockIdx.x - (wj.logw ? 1 : 0);       // (a particle's blocks still share blockIdx.x % 8, i.e. their XCD)
    DBG_CLOCK(60, bidx == 0);
    const int xcd = bidx & 7, q = bidx >> 3;
    const int p = (q / groups) * 8 + xcd, g = q % groups;
    if (p >= P) return;
    const int W = lid.lut_w, S = lid.num_spokes, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Slam2dMap m = maps[p];
    const double px = pose[(size_t)p * pstride], py = pose[(size_t)p * pstride + 1], th = pose[(size_t)p * pstride + 2];
    // spokesOffsetIdxByTheta = int(rint(theta / (2*pi) * numSpokes))  (:131)
    const int offset = (int)rint(th / (2 * 3.141592653589793) * (double)S);
    int first_spoke = (lid.spoke_start + offset) % S;            // spoke of beam 0 (:134), in [0, S)
    if (first_spoke < 0) first_spoke += S;
    const double inv_unit = 1.0 / lid.unit;
    uint32_t f = 0;
    // one wave per beam, UPDB_UNROLL chunks of 64 cells in flight: the walk is a chain of dependent loads
    // (list -> cell -> store), so the loads of several chunks are issued before the first use.  The map
    // index of a cell is computed per cell (rint_div: tabulating both axes per block costs more than that)
    {
        const int beam = g * UPDB_BEAMS + wave;
        if (beam >= lid.beams) return;
        const double rg = ranges[beam];
        const double lo = rg - lid.wall_half, hi = rg + lid.wall_half;
        const bool returned = rg < lid.max_range;
        const int spoke = (first_spoke + beam) % S;
        // floor(r / unit) is monotone in r: cells with r > lo sit in bands >= band(lo), cells with r < hi in
        // bands <= band(hi); a beam without return marks no free cells (:138) and starts at its wall band
        const int nb = lid.num_bands;
        const int* __restrict__ bp = lid.spoke_band + (size_t)spoke * (nb + 1);
        const int qlo = lo > 0.0 ? (int)fmin(floor(lo / lid.unit), 2.0e9) : 0;
        const int qhi = hi > 0.0 ? (int)fmin(floor(hi / lid.unit), 2.0e9) : -1;
        if (qhi < 0) return;
        const int b0 = returned ? 0 : min(qlo / SLAM2D_SPOKE_BAND, nb), b1 = min(qhi / SLAM2D_SPOKE_BAND + 1, nb);
        const int kbeg = bp[b0], kend = bp[max(b0, b1)];
        // cells of the bands wholly below band(lo) have floor(r / unit) < floor(lo / unit), hence r < lo: free by construction
        // (most of a returned beam's cells); their radii are not read (8 of a chunk's ~36 lines; UPDB_SKIP_R=0: read them all)
        const int klo = (UPDB_SKIP_R && lo > 0.0) ? bp[min(qlo / SLAM2D_SPOKE_BAND, nb)] : kbeg;
        const double* __restrict__ sr = lid.spoke_r;
        const uint32_t* __restrict__ sc = lid.spoke_cells;
        // stale indices of a beam during which the map grew on a low side (:144-152): shift of the beam's own growth,
        // of the later beams' growths, and the map shape at the time of the write (Python wraps a negative index
        // against THAT shape) -- LidarModel.grow_for_update
        int sx = 0, sy = 0, ax = 0, ay = 0, wc = m.cols, wr = m.rows;
        if (beam_shift) {
            const int32_t* bs = beam_shift + ((size_t)p * lid.beams + beam) * 6;
            sx = bs[0]; sy = bs[1]; ax = bs[2]; ay = bs[3]; wc = bs[4]; wr = bs[5];
        }
        const uint32_t ncells = (uint32_t)m.rows * (uint32_t)m.pitch;
        // The map index of a window cell is rint(((pose + xs[j]) - mapLim0) / unit) (:104-105,144-145), two fp64 chains per
        // cell.  When the window step IS the map unit (lidarMaxRange a whole number of cells: every configuration in use)
        // that is rint(A + j) with A = (pose - R - mapLim0) / unit: the fp64 evaluation differs from the real A + j by
        // < 1e-11 cells (three roundings at magnitude <= 1e4), so it rounds to j + rint(A) whenever A is farther than
        // 1e-6 from a half-integer -- decided once per particle; a particle that close to a rounding boundary takes the
        // per-cell path with its exact-division fallback.  (Config 5: 138 k waves x ~25 fp64 instructions per cell saved;
        // the kernel is instruction-bound there.)
        const double Ax = ((px + -lid.max_range) - m.lim_x0) * inv_unit, Ay = ((py + -lid.max_range) - m.lim_y0) * inv_unit;
        const double rAx = rint(Ax), rAy = rint(Ay);
        const bool lattice = !beam_shift && lid.lut_xs_step == lid.unit && fabs(Ax) < 1e8 && fabs(Ay) < 1e8 &&
                             fabs(fabs(Ax - rAx) - 0.5) > 1e-6 && fabs(fabs(Ay - rAy) - 0.5) > 1e-6 && !UPDB_NO_LATTICE;
        const int bx = (int)rAx, by = (int)rAy;
        for (int k0 = kbeg + lane; k0 < kend; k0 += 64 * UPDB_UNROLL) {
            // straight-line phases, every load of a phase issued before its first use (no branches in between)
            double r[UPDB_UNROLL], xj[UPDB_UNROLL], yi[UPDB_UNROLL];
            uint32_t cell[UPDB_UNROLL], inc[UPDB_UNROLL], c[UPDB_UNROLL], at[UPDB_UNROLL];
            int mxs[UPDB_UNROLL], mys[UPDB_UNROLL];
            if ((k0 - lane) + 64 * UPDB_UNROLL <= klo) {        // (wave-uniform)
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) {
                    r[u] = -INFINITY;
                    cell[u] = sc[k0 + u * 64];
                }
            } else {
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) {
                    const int k = min(k0 + u * 64, kend - 1);
                    r[u] = sr[k];
                    cell[u] = sc[k];
                }
            }
            if (lattice) {
                // the window's cells sit on the map's lattice: column index + the particle's offset, nothing else
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) {
                    mxs[u] = (int)(cell[u] & 0xffffu) + bx;
                    mys[u] = (int)(cell[u] >> 16) + by;
                }
            } else {
            // window coordinate of a column / row: np.linspace(-R, R, W)[j] = j * step + (-R), last element R
            // (lut_xs_step, checked against the table by the host), else the table itself
            if (lid.lut_xs_step != 0.0) {
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) {
                    const int cj = (int)(cell[u] & 0xffffu), ci = (int)(cell[u] >> 16);
                    xj[u] = cj == W - 1 ? lid.max_range : (double)cj * lid.lut_xs_step + -lid.max_range;
                    yi[u] = ci == W - 1 ? lid.max_range : (double)ci * lid.lut_xs_step + -lid.max_range;
                }
            } else {
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) {
                    xj[u] = lid.lut_xs[cell[u] & 0xffffu];
                    yi[u] = lid.lut_xs[cell[u] >> 16];
                }
            }
            bool slow = false;
#pragma unroll
            for (int u = 0; u < UPDB_UNROLL; ++u) {
                // convertRealXYToMapIdx(x + xAtSpokeDir, ...)  (:104-105,144-145) as rint_div, fast path only
                const double tx = ((px + xj[u]) - m.lim_x0) * inv_unit, ty = ((py + yi[u]) - m.lim_y0) * inv_unit;
                const double rx = rint(tx), ry = rint(ty);
                slow |= fabs(fabs(tx - rx) - 0.5) < 1e-6 || fabs(fabs(ty - ry) - 0.5) < 1e-6 || !(fabs(tx) < 1e9) || !(fabs(ty) < 1e9);
                mxs[u] = (int)rx; mys[u] = (int)ry;
            }
            if (__any(slow)) {                                     // a quotient next to a rounding boundary: exact division
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) {
                    mxs[u] = (int)rint(((px + xj[u]) - m.lim_x0) / lid.unit);
                    mys[u] = (int)rint(((py + yi[u]) - m.lim_y0) / lid.unit);
                }
            }
            }
#pragma unroll
            for (int u = 0; u < UPDB_UNROLL; ++u) {
                inc[u] = 0u;
                if (k0 + u * 64 < kend) {
                    if (returned && r[u] < lo) inc[u] = 1u;                      // :138-139,149
                    else if (r[u] > lo && r[u] < hi) inc[u] = 0x00020002u;       // :142-143,151-152
                }
                int mx = mxs[u] - sx, my = mys[u] - sy;
                if (beam_shift) {
                    mx -= ax; my -= ay;
                    if (mx < 0) mx += wc;
                    if (my < 0) my += wr;
                    mx += ax; my += ay;
                }
                if (inc[u] && (mx < 0 || mx >= m.cols || my < 0 || my >= m.rows)) { f |= SLAM2D_F_UPDATE_OUTSIDE_MAP; inc[u] = 0u; }
                mxs[u] = mx; mys[u] = my;
                at[u] = inc[u] ? (uint32_t)my * (uint32_t)m.pitch + (uint32_t)mx : 0u;     // cell 0: a harmless read
            }
            if (m.wide) {
                // 64-bit cells (visited << 32 | total): a map whose counts have outgrown 16 bits (the host promotes it before
                // that can happen; the reference's float64 counts never saturate, Utils/OccupancyGrid.py:148-152)
                unsigned long long* cells64 = reinterpret_cast<unsigned long long*>(m.cells);
                unsigned long long c64[UPDB_UNROLL];
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) c64[u] = cells64[min(at[u], ncells - 1u)];
#pragma unroll
                for (int u = 0; u < UPDB_UNROLL; ++u) {
                    if (!inc[u]) continue;
                    const unsigned long long add = inc[u] == 1u ? 1ull : 0x0000000200000002ull;
                    if (beam_shift) { atomicAdd(&cells64[at[u]], add); continue; }
                    const unsigned long long nc = c64[u] + add;
                    cells64[at[u]] = nc;
                    const bool was = 2ull * (c64[u] >> 32) > (c64[u] & 0xffffffffull), is = 2ull * (nc >> 32) > (nc & 0xffffffffull);
                    if (was != is) {
                        uint32_t* word = m.occ_bits + (size_t)mys[u] * m.bits_pitch + (mxs[u] >> 5);
                        if (is) atomicOr(word, 1u << (mxs[u] & 31)); else atomicAnd(word, ~(1u << (mxs[u] & 31)));
                    }
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < UPDB_UNROLL; ++u) c[u] = m.cells[min(at[u], ncells - 1u)];
#pragma unroll
            for (int u = 0; u < UPDB_UNROLL; ++u) {
                if (!inc[u]) continue;
                if ((c[u] & 0xffffu) + (inc[u] & 0xffffu) > 0xffffu) { f |= SLAM2D_F_COUNT_OVERFLOW; continue; }
                if (beam_shift) {
                    // stale-index writes can land on a cell of ANOTHER beam's spoke (the reference then adds both
                    // increments, Utils/OccupancyGrid.py:148-152): no plain read-modify-write here; the caller rebuilds
                    // the occupancy bits afterwards (slam2d_map_refresh_bits)
                    atomicAdd(&m.cells[at[u]], inc[u]);
                    continue;
                }
                const uint32_t nc = c[u] + inc[u];
                m.cells[at[u]] = nc;
                const bool was = 2u * (c[u] >> 16) > (c[u] & 0xffffu), is = 2u * (nc >> 16) > (nc & 0xffffu);
                if (was != is) {                                       // keep the occupancy bit in step
                    uint32_t* word = m.occ_bits + (size_t)mys[u] * m.bits_pitch + (mxs[u] >> 5);
                    if (is) atomicOr(word, 1u << (mxs[u] & 31)); else atomicAnd(word, ~(1u << (mxs[u] & 31)));
                }
            }
        }
    }
    DBG_CLOCK(62, bidx == 0);
    if (f) atomicOr(&flags[p], f);
}

// Sharded normaliser, merge half: every rank folds the gathered [world][3] partials in rank order
// (so the result does not depend on the network's reduction order), normalises its own particles
// and evaluates sum (w - 1/N)^2 = sum w^2 - 1/N over ALL N particles (Algorithm/FastSlam.py:32-35).
__global__ __launch_bounds__(256) void k_weights_merge(double* logw, int N, const double* __restrict__ parts, int world,
                                                       double total_particles, double* w, double* stats,
                                                       const uint32_t* __restrict__ abort_flags, int abort_n, uint32_t abort_mask,
                                                       uint32_t* nsync = nullptr, const double* pack_d = nullptr, double* pack_h = nullptr,
                                                       int pack_n = 0, uint32_t* h_seq = nullptr, uint32_t seq = 0u) {
    if (abort_mask) {                                      // slam2d_groups_commit: a voided scan (see k_grid_update) has no partials to merge
        bool bad = false;
        for (int i = threadIdx.x; i < abort_n; i += blockDim.x) bad |= (abort_flags[i] & abort_mask) != 0u;
        if (__syncthreads_or(bad)) return;
    }
    bool voided = false;                                   // (a group of some rank left the VOID partial: k_grid_update's sharded abort)
    for (int r = 0; r < world; ++r) voided |= parts[3 * r + 1] < 0.0;
    double gm = -INFINITY;
    for (int r = 0; r < world; ++r) gm = fmax(gm, parts[3 * r]);
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < world; ++r) {
        const double sc = exp(parts[3 * r] - gm);
        s1 += parts[3 * r + 1] * sc;
        s2 += parts[3 * r + 2] * sc * sc;
    }
    const double lse = gm + log(s1);
    if (!voided)
        for (int i = threadIdx.x; i < N; i += 256) {
            const double lw = logw[i];
            w[i] = exp(lw - gm) / s1;
            logw[i] = lw - lse;
        }
    if (threadIdx.x == 0) {
        if (voided) { stats[0] = NAN; stats[1] = -1.0; }   // the report's marker: variance NaN, log of the sum -1 (a peer's NaN weights give NaN, NaN)
        else { stats[0] = s2 / (s1 * s1) - 1.0 / total_particles; stats[1] = lse; }
    }
    if (nsync && !voided) {                                // (slam2d_weights_merge_publish: the groups' next normaliser blocks wait for this)
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const uint32_t gen = __hip_atomic_load(&nsync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nsync[1], gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (h_seq) {                                           // (slam2d_weights_merge_publish_report: the scan's pack to the host, as publish_pack)
        __syncthreads();
        for (int i = threadIdx.x; i < pack_n; i += blockDim.x) pack_h[i] = pack_d[i];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The sharded normaliser without events between the groups and the normaliser's stream (round 5): the groups' blocks arrive on
// d_norm_sync[0] (normaliser_arrive), this one-wave kernel on the normaliser's stream waits for all of them -- they were enqueued
// before it -- and takes the counter back to zero; the all-gather of the partials and the merge follow on that stream, and the merge
// publishes d_norm_sync[1], which the groups' next normaliser blocks wait for (normaliser_wait).
__global__ __launch_bounds__(64) void k_norm_gate(uint32_t* nsync, int G) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64(), SYNC_SPIN_TICKS = sync_spin_ticks(nsync);
        bool late = false;
        while (__hip_atomic_load(&nsync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)G) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > SYNC_SPIN_TICKS) {           // (sticky: every later normaliser block raises the fault bit)
                __hip_atomic_store(&nsync[62], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                late = true;
                break;
            }
        }
        // (a tripped gate leaves the arrivals where they are: late groups still count themselves in, and the fault is fatal anyway)
        if (!late) __hip_atomic_store(&nsync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (the partials: the collective behind this kernel reads them)
    }
}

// ------------------------------------------------------------------------------------
// K5  odometry prior / post-match bookkeeping          (Algorithm/FastSlam.py:77-120,131-135)
// ------------------------------------------------------------------------------------
__global__ void k_prior(const double* __restrict__ prev, double raw_theta, double prev_raw_theta, int has_turn,
                        double raw_turn, const double* __restrict__ heading, int P, double* est, double* psi_cs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    prior_one(prev, heading, p, raw_theta, prev_raw_theta, has_turn, raw_turn, est, psi_cs);
}

// The closed loop's prior of a particle group with the scan's ranges PULLED from the caller's pinned host buffer in the same launch
// (Slam2dScan.h_ranges, ABI 16) into the group's own device buffer, which every later kernel of the group's scan reads -- instead
// of a copy-engine transfer and an event in front of every group's match.  (The uniforms, one per particle and read once by the
// selecting kernel, are read from the pinned buffer where they are used.)  Normally neither is launched: the previous scan's commit
// has written this scan's prior and pulled its ranges (Slam2dScan.h_next_ranges).
__global__ __launch_bounds__(256) void k_prior_pull(const double* __restrict__ prev, double raw_theta, double prev_raw_theta, int has_turn,
                                                    double raw_turn, const double* __restrict__ heading, int P, double* est, double* psi_cs,
                                                    const double* h_ranges, int B, double* d_pull) {
    for (int i = threadIdx.x; i < B; i += blockDim.x) d_pull[i] = __builtin_nontemporal_load(h_ranges + i);
    for (int p = threadIdx.x; p < P; p += blockDim.x) prior_one(prev, heading, p, raw_theta, prev_raw_theta, has_turn, raw_turn, est, psi_cs);
}
// Abort decision across groups without events (Slam2dScan.match_seq, ABI 16): the wave that writes a particle's result at the scan's
// LAST level counts the particle in (Slam2dLevel.arrive = word 61 of the sync block: particle-matches finished since the words were
// zeroed); every group's commit starts with k_abort_gate, ONE wave that waits -- bounded -- until all particles of all groups of this
// scan's match call have been counted: the update launch behind it then reads every group's fault bits as it did behind the events.
// No deadlock, whatever hardware queues the groups' streams share: a commit call is issued after the match call of ALL groups has
// been issued completely, so everything a gate waits for sits in front of it in every queue; and only one wave per group waits.
// (A gate that counted its own group in -- one launch less per match -- deadlocked with eight groups on eight hardware queues, one
// of them the default stream's: two groups shared a queue and the first gate waited for the one queued behind it.)
__global__ __launch_bounds__(64) void k_abort_gate(uint32_t* nsync, uint32_t want, uint32_t* flags) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64(), SYNC_SPIN_TICKS = sync_spin_ticks(nsync);
        while ((int)(__hip_atomic_load(&nsync[61], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > SYNC_SPIN_TICKS) { atomicOr(&flags[0], SLAM2D_F_SYNC_TIMEOUT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}
__global__ __launch_bounds__(64) void k_match_arrive(uint32_t* nsync, uint32_t n) {       // (single-level matches: no level to carry the word)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&nsync[61], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_post_match(const Slam2dMatch* __restrict__ fine, const Slam2dMatch* __restrict__ coarse, int P,
                             double* prev, double* heading, double* logw, double* report) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) post_match_one(fine, coarse, p, prev, heading, logw, report);
}

// ------------------------------------------------------------------------------------
// resample state movement / fill
// ------------------------------------------------------------------------------------
__global__ void k_gather_maps(const Slam2dMap* __restrict__ src, const Slam2dMap* __restrict__ dst,
                              const int32_t* __restrict__ index) {
    const int p = blockIdx.y;
    const Slam2dMap s = src[index[p]], d = dst[p];
    const size_t n = (size_t)s.rows * s.pitch * (s.wide ? 2 : 1);          // 32-bit words (a wide map: two per cell)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d.cells[i] = s.cells[i];
}

// The picture of a map the reference draws (Algorithm/FastSlam.py:172-176): 1 - visited / total over a window, optionally
// flipped upside down (np.flipud); float64 like the reference's arrays and / or 8-bit grey.
__global__ void k_map_image(const Slam2dMap* __restrict__ maps, int p, int x0, int y0, int w, int h, int flipud,
                            double* out, uint8_t* out_u8) {
    const Slam2dMap m = maps[p];
    const long long n = (long long)w * h;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / w), c = (int)(i - (long long)r * w);
        const int my = flipud ? y0 + (h - 1 - r) : y0 + r;
        double vis, tot;
        if (m.wide) {
            const unsigned long long v = reinterpret_cast<const unsigned long long*>(m.cells)[(size_t)my * m.pitch + x0 + c];
            vis = (double)(v >> 32); tot = (double)(v & 0xffffffffull);
        } else {
            const uint32_t v = m.cells[(size_t)my * m.pitch + x0 + c];
            vis = (double)(v >> 16); tot = (double)(v & 0xffffu);
        }
        const double val = 1.0 - vis / tot;                                     // ogMap = visited / total; 1 - ogMap (:173,176)
        if (out) out[i] = val;
        if (out_u8) out_u8[i] = (uint8_t)rint(fmin(fmax(val, 0.0), 1.0) * 255.0);
    }
}

__global__ void k_fill(uint32_t* cells, long long n, uint32_t value) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        cells[i] = value;
}

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
template <int R>
static void launch_sweep(const Slam2dLevel& lv, int P, int chunks, hipStream_t s, int mode = 0, const double* sel_est = nullptr,
                         int sel_estride = 0, Slam2dMatch* sel_out = nullptr) {
    const int bpp = lv.ntheta * cdiv(chunks, mode == 2 ? SWEEP_REST_CHUNKS : (mode == 0 ? SWEEP_MAIN_CHUNKS : 1));   // blocks per particle
    const unsigned grid = cdiv(P, 8) * 8 * bpp;
    static const bool no_skip = [] { const char* e = getenv("SLAM2D_SWEEP_NOSKIP"); return e && atoi(e) == 1; }();
    // worth it for long cell lists (measured: 1081 beams -15 %, 180 beams +4 %: the vector prologue of the skip
    // loop costs more than 29 % fewer loads save when a wave has only ~43 cells)
    const bool skip = R == 1 && sweep_skips(lv) && !no_skip;
    if constexpr (R == 1) {
        if (mode == 1) { k_sweep<1, 1, false><<<grid, 256, 0, s>>>(lv, P, chunks, bpp); return; }
        if (skip) {
            if (mode == 0) k_sweep<1, 0, true><<<grid, 256, 0, s>>>(lv, P, chunks, bpp, sel_est, sel_estride, sel_out);
            else k_sweep<1, 2, true><<<grid, 256, 0, s>>>(lv, P, chunks, bpp);
            return;
        }
    }
    // deep (R == 1): SWEEP_DEPTH gathers in flight per wave; SLAM2D_SWEEP_DEEP = 0 never, 1 (default) where a particle has at most
    // 64 blocks (the launch is about one round of blocks: latency-bound), 2 always
    static const int deep_env = [] { const char* e = getenv("SLAM2D_SWEEP_DEEP"); return e ? atoi(e) : 1; }();
    const int deep = R == 1 && (deep_env == 2 || (deep_env == 1 && bpp <= 64)) ? 1 : 0;
    if (mode == 0) k_sweep<R, 0, false><<<grid, 256, 0, s>>>(lv, P, chunks, bpp, sel_est, sel_estride, sel_out, deep);
    else if (mode == 2) k_sweep<R, 2, false><<<grid, 256, 0, s>>>(lv, P, chunks, bpp, nullptr, 0, nullptr, deep);
}

extern "C" {

int slam2d_abi_version(void) { return SLAM2D_ABI_VERSION; }

int slam2d_sizeof(const char* name) {
    if (!name) return -1;
    if (!strcmp(name, "Slam2dMap")) return (int)sizeof(Slam2dMap);
    if (!strcmp(name, "Slam2dLidar")) return (int)sizeof(Slam2dLidar);
    if (!strcmp(name, "Slam2dFrame")) return (int)sizeof(Slam2dFrame);
    if (!strcmp(name, "Slam2dLevel")) return (int)sizeof(Slam2dLevel);
    if (!strcmp(name, "Slam2dMatch")) return (int)sizeof(Slam2dMatch);
    if (!strcmp(name, "Slam2dPartial")) return (int)sizeof(Slam2dPartial);
    if (!strcmp(name, "Slam2dGroup")) return (int)sizeof(Slam2dGroup);
    if (!strcmp(name, "Slam2dScan")) return (int)sizeof(Slam2dScan);
    return -1;
}

int slam2d_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

static int check_level(const Slam2dLidar* lidar, const Slam2dLevel* lv, int P) {
    if (!lidar || !lv || P <= 0) return SLAM2D_E_BADARG;
    if (lv->blur_radius < 0 || lv->blur_radius > SLAM2D_MAX_BLUR_RADIUS) return SLAM2D_E_TOOLARGE;
    if (lidar->beams < 2 || lidar->beams > SLAM2D_MAX_BEAMS) return SLAM2D_E_TOOLARGE;
    if (lv->fmax <= 0 || lv->fpitch < lv->fmax || lv->wmax <= 0 || lv->ncell < 0 || lv->ntheta <= 0) return SLAM2D_E_BADARG;
    if (!(lv->cost_scale > 0.0)) return SLAM2D_E_BADARG;
    if ((long long)lv->fmax * lv->fpitch >= (1ll << 29)) return SLAM2D_E_TOOLARGE;     // byte offsets into one field stay below 2^31
    return 0;
}

// ---- launch sequences shared by slam2d_field_build / slam2d_sweep / slam2d_match ----
static int check_field_args(const Slam2dLevel& lv, int P, bool lazy) {
    // (occ and tilemask are cleared by ONE memset when occ_gen == 0; with generation stamps nothing is cleared and a level may
    // be an offset view of a larger one -- slam2d_groups_* -- whose flags do not follow its image)
    if (!lv.occ || !lv.tilemask || (lv.occ_gen == 0 && lv.tilemask != lv.occ + (size_t)P * lv.fmax * lv.fpitch) || !lv.tilestate || !lv.tilemin || !lv.tilemax || !lv.tilelist || !lv.tilecount)
        return SLAM2D_E_BADARG;
    if (lazy && !lv.tileneed) return SLAM2D_E_BADARG;
    if (lv.tmax * lv.tmax > 28000) return SLAM2D_E_TOOLARGE;        // k_tile_triage: 32 passes, 5 bytes of LDS per tile (144 KB)
    if (lv.bnb == 3) {                                 // angle bounds: small cubes only, windows of <= 5 x 5 cells
        const int nx = 2 * lv.ncell + 1;
        if (!lazy || !lv.gmin || !lv.gmin2 || !lv.pcells || !lv.bounds || !lv.bnb_best || !lv.seed_key) return SLAM2D_E_BADARG;
        if (nx > 5 || nx * ((nx + 3) / 4) > 32 || lv.tmax * 16 != lv.fpitch || lv.ntheta >= (1 << 14)) return SLAM2D_E_BADARG;
    } else if (lv.bnb) {
        const int nx = 2 * lv.ncell + 1;
        if (!lazy || !lv.gmin || !lv.gmin2 || !lv.pcells || !lv.bounds || !lv.tile_pmax || !lv.bnb_best || !lv.prune_state)
            return SLAM2D_E_BADARG;
        if (lv.bnb == 2 && (!lv.gmin3d || !lv.p3cells || !lv.bounds1 || !lv.seed_key || (lv.tmax & 0) != 0)) return SLAM2D_E_BADARG;
        if (nx < 9 || nx > 64 || lv.tmax * 16 != lv.fpitch) return SLAM2D_E_BADARG;
        if (lv.ntheta > SLAM2D_BNB_MAX_THETA) return SLAM2D_E_TOOLARGE;
        if ((long long)lv.ntheta * ((nx + 3) / 4) * (((nx + 3) / 4 + 3) / 4 * 4) > (long long)XS_THREADS * XS_MAX_PER) return SLAM2D_E_TOOLARGE;
    }
    return 0;
}

// frame geometry, axis index vectors, cleared occupancy image / tile flags (/ needed-tile bitmap)
static int launch_frames(const Slam2dLidar& lid, const Slam2dLevel& lv, const Slam2dMap* d_maps, int P,
                         const double* d_centre, int centre_stride, uint32_t* d_flags, bool lazy, hipStream_t s,
                         const double* d_ranges = nullptr) {
    if (d_ranges && (!lv.beam_xy || centre_stride < 3)) d_ranges = nullptr;
    k_frame_axis<<<dim3(cdiv(lv.wmax, 256), P, 2), 256, 0, s>>>(lid, lv, d_maps, d_centre, centre_stride, d_flags, d_ranges);
    if (lv.occ_gen < 0 || lv.occ_gen > 255) return SLAM2D_E_BADARG;
    if (lv.occ_gen != 0) return 0;                     // generation stamps: nothing to clear
    return (int)hipMemsetAsync(lv.occ, 0, (size_t)P * lv.fmax * lv.fpitch + (size_t)P * flag_bytes(lv), s);
}

// occupied cells -> field image, tile triage (+ fill), blur + clamp, minimum check
static int launch_field(const Slam2dLevel& lv, const Slam2dMap* d_maps, int P, uint32_t* d_flags, bool lazy, hipStream_t s,
                        bool scattered = false, bool field_max_needed = true) {
    if (!scattered) {
        StageScope prof(SLAM2D_STAGE_SCATTER, s);
        k_occ_scatter<<<dim3(cdiv(cdiv(lv.wmax, 32) + 1, 64), cdiv(lv.wmax, SCATTER_ROWS), P), dim3(64, 4), (size_t)lv.wmax * sizeof(int32_t), s>>>(lv, d_maps);
    }
    const int ntile = lv.tmax * lv.tmax;
    // Without bounds (no gmin2 to derive), without the sweep's free-tile masks and without the prior pruning (which reads the
    // field's maximum) k_blur_check_redo has ONE duty left: the minimum check of a frame without a free tile -- the blur's last
    // block does it (k_blur_clamp, tail).  One launch per level less: 2 x 6 us per scan at the reference's defaults.
    // SLAM2D_FOLD_CHECK=0: the separate launch.
    static const bool fold = [] { const char* e = getenv("SLAM2D_FOLD_CHECK"); return !e || atoi(e) != 0; }();
    const bool folded = fold && lazy && !lv.bnb && !sweep_skips(lv) && !field_max_needed && lv.sync != nullptr;
    {
        const int kb = (lv.blur_radius + 7) >> FLAG_SHIFT;
        const int rw = (flag_pitch(lv) >> 4) + 1;
        const size_t lds = (size_t)((2 * ((((2 * lv.tmax + 2 * kb) * rw + 1) & ~1) + lv.tmax * (rw + 1)) + 15) & ~15) + ((ntile + 3) & ~3) + 4 * ((ntile + 31) / 32) + 2 * (size_t)ntile;
        // more than the default dynamic LDS limit: ask once per device (160 KB per CU on gfx950); a device that does not grant it
        // gets a clean error instead of a failed launch
        static size_t lds_allowed[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        size_t& allowed = lds_allowed[dev >= 0 && dev < 64 ? dev : 0];
        if (allowed == 0) allowed = 64 * 1024;
        if (lds > allowed) {
            const size_t want = 160 * 1024 - 512;
            if (lds > want || hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_triage), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess) {
                (void)hipGetLastError();
                return SLAM2D_E_TOOLARGE;
            }
            allowed = want;
        }
        k_tile_triage<<<P, TRIAGE_THREADS, lds, s>>>(lv, lazy ? 1 : 0);
    }
    {
        StageScope prof(SLAM2D_STAGE_BLUR, s);
        static const int blur_blocks = [] { const char* e = getenv("SLAM2D_BLUR_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : SLAM2D_BLUR_BLOCKS_PER_PARTICLE; }();
        static const bool xcd_pin = [] { const char* e = getenv("SLAM2D_BLUR_XCD"); return !e || atoi(e) != 0; }();
        const int bpp = xcd_pin ? min(ntile, blur_blocks) : 0;
        const dim3 bgrid = xcd_pin ? dim3((unsigned)cdiv(P, 8) * 8 * bpp) : dim3(min(ntile, blur_blocks), P);
        const int tail = folded ? 1 : 0;
        switch (lv.blur_radius) {
            case 2: k_blur_clamp<2><<<bgrid, BLUR_THREADS, 0, s>>>(lv, P, bpp, d_flags, tail); break;
            case 4: k_blur_clamp<4><<<bgrid, BLUR_THREADS, 0, s>>>(lv, P, bpp, d_flags, tail); break;
            case 8: k_blur_clamp<8><<<bgrid, BLUR_THREADS, 0, s>>>(lv, P, bpp, d_flags, tail); break;
            default: k_blur_clamp<0><<<bgrid, BLUR_THREADS, 0, s>>>(lv, P, bpp, d_flags, tail); break;
        }
    }
    if (folded) return 0;                              // the minimum check rode in the blur's launch: nothing else to do at this level
    // gmin2 only where a tile was written (SLAM2D_GMIN2_FULL=1: over the whole frame, as before round 3)
    static const bool full = [] { const char* e = getenv("SLAM2D_GMIN2_FULL"); return e && atoi(e) == 1; }();
    const int dirty = lazy && !full ? 1 : 0;
    static const int dblocks = [] { const char* e = getenv("SLAM2D_GMIN2_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 16; }();
    const dim3 cgrid(P, lv.bnb ? (dirty ? dblocks : 16) : 1);
    switch (lv.blur_radius) {
        case 2: k_blur_check_redo<2><<<cgrid, 256, 0, s>>>(lv, d_flags, dirty); break;
        case 4: k_blur_check_redo<4><<<cgrid, 256, 0, s>>>(lv, d_flags, dirty); break;
        case 8: k_blur_check_redo<8><<<cgrid, 256, 0, s>>>(lv, d_flags, dirty); break;
        default: k_blur_check_redo<0><<<cgrid, 256, 0, s>>>(lv, d_flags, dirty); break;
    }
    return 0;
}

// beam endpoints, unique cells per theta, priors (/ needed tiles)
static void launch_endpoints(const Slam2dLidar& lid, const Slam2dLevel& lv, int P, const double* d_est, int est_stride,
                             const double* d_ranges, double est_moving_dist, const double* d_psi_cs, uint32_t* d_flags,
                             bool mark, bool prune, hipStream_t s, bool beam_table = false,
                             const Slam2dMap* own_frame_maps = nullptr, bool with_scatter = false) {
    StageScope prof(SLAM2D_STAGE_ENDPOINTS, s);
    int n = 256;
    while (n < lid.beams) n <<= 1;
    int hsize = 512;
    while (hsize < lid.beams + (lid.beams >> 1)) hsize <<= 1;
    size_t ep_lds = (size_t)(2 * hsize + 32 + (mark ? 6 * lv.tmax * ((lv.tmax + 31) / 32) + (lv.tmax * lv.tmax + 31) / 32 : 0)) * sizeof(int);
    const int G = lv.ep_group > 0 ? lv.ep_group : 1;
    const int nt = lid.beams <= 192 ? 192 : 256;
    // round 4: only the tiles the poses read are marked, not the wider region the block minima of the bounds summarise (see
    // k_endpoints, mark == 2: 13 % fewer needed tiles, 6 % fewer blurred ones at config 2, the same surviving pose tiles;
    // SLAM2D_TIGHT_NEED=0 restores the wide marking)
    static const int markv = [] { const char* e = getenv("SLAM2D_TIGHT_NEED"); return e && atoi(e) == 0 ? 1 : 2; }();
    // with_scatter (needs own_frame_maps): the occupied-cell scatter as further blocks of this launch
    const int sbx = with_scatter ? cdiv(cdiv(lv.wmax, 32) + 1, 64) : 0, sby = with_scatter ? cdiv(lv.wmax, (nt / 64) * SCATTER_ROLE_NR) : 0;
    if (with_scatter) ep_lds = ep_lds > (size_t)lv.wmax * sizeof(int32_t) ? ep_lds : (size_t)lv.wmax * sizeof(int32_t);
    const dim3 grid(cdiv(lv.ntheta, G) + (own_frame_maps ? 2 : 1) + sbx * sby, P);
    if (nt == 192)
        k_endpoints<192><<<grid, 192, ep_lds, s>>>(lid, lv, d_est, est_stride, d_ranges, d_flags, est_moving_dist, lv.fine ? nullptr : d_psi_cs,
                                                   mark ? markv : 0, prune ? 1 : 0, beam_table && lv.beam_xy ? 1 : 0, own_frame_maps, sbx);
    else
        k_endpoints<256><<<grid, 256, ep_lds, s>>>(lid, lv, d_est, est_stride, d_ranges, d_flags, est_moving_dist, lv.fine ? nullptr : d_psi_cs,
                                                   mark ? markv : 0, prune ? 1 : 0, beam_table && lv.beam_xy ? 1 : 0, own_frame_maps, sbx);
}

// cube sweep + selection
// k_bound_lds where the level carries the byte image of the bounds (Slam2dLevel.gmin2b) and it fits the CU's LDS.
// SLAM2D_BOUND_LDS=0: never.  A particle's angles are split over up to SLAM2D_BOUND_LDS_SPLIT blocks (each stages the image)
// while the launch stays below SLAM2D_BOUND_LDS_BLOCKS blocks (128: measured at 16 / 64 / 128 / 256 particles per launch, four /
// two / one / one block per particle): small launches need the CUs, large ones pay for every extra staging pass and barrier.
static bool launch_bound_lds(const Slam2dLevel& lv, int P, hipStream_t s) {
    static const int mode = [] { const char* e = getenv("SLAM2D_BOUND_LDS"); return e ? atoi(e) : -1; }();
    static const int max_split = [] { const char* e = getenv("SLAM2D_BOUND_LDS_SPLIT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }();
    static const int want_blocks_env = [] { const char* e = getenv("SLAM2D_BOUND_LDS_BLOCKS"); return e ? atoi(e) : 0; }();
    static const int rle_mode = [] { const char* e = getenv("SLAM2D_BOUND_LDS_RLE"); return e ? atoi(e) : -1; }();
    if (mode == 0 || !lv.gmin2b) return false;
    const int nx = 2 * lv.ncell + 1, nbt = (nx + 3) >> 2;
    const int nset = cdiv(nbt * nbt, WAVE);
    const int gp = lv.tmax << 2, lp = lv.g2b_pitch;
    if (nset > 4 || lv.kmax > 2048 || lp < gp || (lp & 15)) return false;
    const size_t image = ((size_t)gp * lp + 15) & ~(size_t)15;
    const bool rle = (rle_mode < 0 ? lv.kmax >= SLAM2D_BEAM_TABLE_MIN : rle_mode != 0) && image + (size_t)nbt * lp < 65536;   // (16-bit offsets in the run words)
    // (long lists: an angle is 16 us of one wave -- 256 blocks: config 5's 64-particle launches, 139 angles, four blocks per particle)
    const int blocks = want_blocks_env > 0 ? want_blocks_env : (rle ? 256 : 128);
    const int bpp = max(1, min(min(max_split, lv.ntheta), blocks / max(P, 1)));
    const int tpb = cdiv(lv.ntheta, bpp);                       // angles per block
    const int rounds = cdiv(tpb, 16), nw = cdiv(tpb, rounds);
    const size_t lds = image + (rle ? (size_t)nw * WAVE * sizeof(unsigned) : 0);
    const size_t want = 160 * 1024 - 512;
    if (lds > want) return false;
    static size_t granted[64][8] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& allowed = granted[dev >= 0 && dev < 64 ? dev : 0][(nset - 1) * 2 + (rle ? 1 : 0)];
    if (allowed == 0) allowed = 64 * 1024;
    const void* fn[8] = {reinterpret_cast<const void*>(k_bound_lds<1, false>), reinterpret_cast<const void*>(k_bound_lds<1, true>),
                         reinterpret_cast<const void*>(k_bound_lds<2, false>), reinterpret_cast<const void*>(k_bound_lds<2, true>),
                         reinterpret_cast<const void*>(k_bound_lds<3, false>), reinterpret_cast<const void*>(k_bound_lds<3, true>),
                         reinterpret_cast<const void*>(k_bound_lds<4, false>), reinterpret_cast<const void*>(k_bound_lds<4, true>)};
    if (lds > allowed) {
        if (hipFuncSetAttribute(fn[(nset - 1) * 2 + (rle ? 1 : 0)], hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        allowed = want;
    }
    const unsigned grid = (unsigned)cdiv(P, 8) * 8 * bpp;
    switch ((nset - 1) * 2 + (rle ? 1 : 0)) {
        case 0: k_bound_lds<1, false><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
        case 1: k_bound_lds<1, true><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
        case 2: k_bound_lds<2, false><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
        case 3: k_bound_lds<2, true><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
        case 4: k_bound_lds<3, false><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
        case 5: k_bound_lds<3, true><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
        case 6: k_bound_lds<4, false><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
        default: k_bound_lds<4, true><<<grid, WAVE * nw, lds, s>>>(lv, P, bpp, tpb); break;
    }
    return true;
}

static int launch_scores(const Slam2dLevel& lv, int P, const double* d_est, int est_stride, const double* d_uniform,
                         Slam2dMatch* d_out, hipStream_t s, int ring_chunks = 0) {
    const int nx = 2 * lv.ncell + 1;
    const int nslot = nx * ((nx + 3) / 4);            // slots of 4 consecutive dx
    const int need = cdiv(nslot, WAVE);
    int bestR = 1;        // slots per lane; measured on MI355X (config 2): 1 -> 124 us, 2..4 -> 132-134 us
    if (const char* ov = getenv("SLAM2D_SWEEP_R")) {              // tuning knob
        const int R = atoi(ov);
        if (R >= 1 && R <= 4) bestR = R;
    }
    if (ring_chunks > 0) bestR = 1;
    const int chunks = cdiv(need, bestR);
    if (lv.ntheta * chunks > lv.npartial) return SLAM2D_E_BADARG;
    if (ring_chunks > 0) {
        // the prior's ring first (write_priors); the particles it does not settle are swept in full by the
        // second pair of launches, whose blocks return at once for everyone else
        {
            StageScope prof(SLAM2D_STAGE_SWEEP, s);
            launch_sweep<1>(lv, P, min(ring_chunks, chunks), s, 1);
        }
        {
            StageScope prof(SLAM2D_STAGE_SELECT, s);
            k_select<1><<<P, WAVE, 0, s>>>(lv, min(ring_chunks, chunks), 1, d_est, est_stride, d_uniform, d_out);
        }
    }
    const int mode = ring_chunks > 0 ? 2 : 0;
    static const bool no_small = [] { const char* e = getenv("SLAM2D_SWEEP_NOSMALL"); return e && atoi(e) == 1; }();
    if (mode == 0 && nslot <= 32 && chunks == 1 && !no_small) {           // small cube: one wave per (particle, theta) plane
        const unsigned grid = (unsigned)cdiv(P, 8) * 8 * lv.ntheta;
        const size_t lds = (size_t)WAVE * 4 * sizeof(unsigned long long) + (size_t)lv.kmax * sizeof(int);
        const int pruned = lv.bnb == 3 ? 1 : 0;
        if (pruned) {                                                     // angle bounds first, then one exact seed per particle
            StageScope prof(SLAM2D_STAGE_BOUND, s);
            k_abound<<<(unsigned)cdiv(P, 8) * 8 * cdiv(lv.ntheta, ABOUND_WAVES), WAVE * ABOUND_WAVES, 0, s>>>(lv, P);
            k_aseed<<<P, WAVE * ASEED_WAVES, (size_t)ASEED_WAVES * WAVE * 4 * sizeof(unsigned long long) + (size_t)lv.kmax * sizeof(int), s>>>(lv, P);
        }
        {
            StageScope prof(SLAM2D_STAGE_SWEEP, s);
            k_sweep_small<<<grid, WAVE, lds, s>>>(lv, P, pruned);
        }
        StageScope prof(SLAM2D_STAGE_SELECT, s);
        k_select<0><<<P, WAVE, 0, s>>>(lv, 1, 1, d_est, est_stride, d_uniform, d_out);
        return 0;
    }
    // a level matched by arg-max (no soft-max draw: the fine level, matchMax) needs nothing of the cube for its selection: the
    // sweep's last block of every particle does it from the partials (k_sweep, sel_out) and k_select is not launched
    // (SLAM2D_FUSE_SELECT=0: the separate launch)
    static const bool fuse_sel = [] { const char* e = getenv("SLAM2D_FUSE_SELECT"); return !e || atoi(e) != 0; }();
    const bool fused = fuse_sel && mode == 0 && d_uniform == nullptr && lv.sync != nullptr;
    {
        StageScope prof(SLAM2D_STAGE_SWEEP, s);
        switch (bestR) {
            case 1: launch_sweep<1>(lv, P, chunks, s, mode, fused ? d_est : nullptr, est_stride, fused ? d_out : nullptr); break;
            case 2: launch_sweep<2>(lv, P, chunks, s, mode, fused ? d_est : nullptr, est_stride, fused ? d_out : nullptr); break;
            case 3: launch_sweep<3>(lv, P, chunks, s, mode, fused ? d_est : nullptr, est_stride, fused ? d_out : nullptr); break;
            default: launch_sweep<4>(lv, P, chunks, s, mode, fused ? d_est : nullptr, est_stride, fused ? d_out : nullptr); break;
        }
    }
    if (fused) return 0;
    {
        StageScope prof(SLAM2D_STAGE_SELECT, s);
        if (mode == 0) k_select<0><<<P, WAVE, 0, s>>>(lv, chunks, bestR, d_est, est_stride, d_uniform, d_out);
        else k_select<2><<<P, WAVE, 0, s>>>(lv, chunks, bestR, d_est, est_stride, d_uniform, d_out);
    }
    return 0;
}

// Upper bound (a superset is harmless) of the number of sweep slots the prior's ring touches, for the grid of
// the ring pass; 0 = pruning not applicable.  The ring itself is listed on the device by write_priors.
static int ring_slot_bound(const Slam2dLevel& lv, double est_dist) {
    if (lv.fine || !lv.ring || !lv.prune_state || lv.ring_cap <= 0 || !(est_dist >= 0.0)) return 0;
    const int nx = 2 * lv.ncell + 1, nq = (nx + 3) / 4;
    const double slack = 1e-9 * (1.0 + est_dist + lv.step * nx);
    int n = 0;
    for (int iy = 0; iy < nx; ++iy)
        for (int sq = 0; sq < nq; ++sq) {
            bool in = false;
            for (int e = 0; e < 4 && 4 * sq + e < nx; ++e) {
                const double mx = (double)(4 * sq + e - lv.ncell) * lv.step, my = (double)(iy - lv.ncell) * lv.step;
                in = in || !(fabs(sqrt(mx * mx + my * my) - est_dist) > lv.max_move_dev + slack);
            }
            n += in ? 1 : 0;
        }
    return n <= lv.ring_cap ? n : 0;
}

// Blocks per particle of k_exact_select: enough to put every CU to work (256 CUs / P particles), at most 4; needs the
// arrival counters (Slam2dLevel.sync).  SLAM2D_XS_SPLIT overrides (1 = one block per particle).
static int exact_split(const Slam2dLevel& lv, int P) {
    static const int forced = [] { const char* e = getenv("SLAM2D_XS_SPLIT"); return e ? atoi(e) : 0; }();
    if (!lv.sync) return 1;
    int n = forced > 0 ? forced : 256 / (P > 0 ? P : 1);      // (the kernel lets only XS_SPLIT_LIGHT of them work on a particle with few tiles)
    if (n > XS_SPLIT_MAX) n = XS_SPLIT_MAX;
    return n < 1 ? 1 : n;
}

int slam2d_field_build(const Slam2dLidar* lidar, const Slam2dLevel* level, const Slam2dMap* d_maps, int32_t P,
                       const double* d_centre, int32_t centre_stride, uint32_t* d_flags, void* stream) {
    int rc = check_level(lidar, level, P);
    if (rc) return rc;
    if (!d_maps || !d_centre || !d_flags || centre_stride < 2) return SLAM2D_E_BADARG;
    { Slam2dLevel chk = *level; chk.bnb = 0; if ((rc = check_field_args(chk, P, false))) return rc; }
    hipStream_t s = (hipStream_t)stream;
    Slam2dLevel lv = *level;
    lv.bnb = 0;                                        // the full build has no pooled image
    if ((rc = launch_frames(*lidar, lv, d_maps, P, d_centre, centre_stride, d_flags, false, s))) return rc;
    if ((rc = launch_field(lv, d_maps, P, d_flags, false, s))) return rc;
    return launch_status();
}

int slam2d_sweep(const Slam2dLidar* lidar, const Slam2dLevel* level, int32_t P, const double* d_est,
                 int32_t est_stride, const double* d_ranges, double est_moving_dist, const double* d_psi_cs,
                 const double* d_uniform, Slam2dMatch* d_out, uint32_t* d_flags, void* stream) {
    int rc = check_level(lidar, level, P);
    if (rc) return rc;
    if (!d_est || !d_ranges || !d_out || !d_flags || est_stride < 3) return SLAM2D_E_BADARG;
    Slam2dLevel lv = *level;
    lv.bnb = 0;                                        // the whole cube, brute force
    if (lv.kmax < lidar->beams || !lv.partials) return SLAM2D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    launch_endpoints(*lidar, lv, P, d_est, est_stride, d_ranges, est_moving_dist, d_psi_cs, d_flags, false, false, s);
    if ((rc = launch_scores(lv, P, d_est, est_stride, d_uniform, d_out, s))) return rc;
    return launch_status();
}

int slam2d_match(const Slam2dLidar* lidar, const Slam2dLevel* level, const Slam2dMap* d_maps, int32_t P,
                 const double* d_est, int32_t est_stride, const double* d_ranges, double est_moving_dist,
                 const double* d_psi_cs, const double* d_uniform, Slam2dMatch* d_out, uint32_t* d_flags,
                 uint32_t options, void* stream) {
    int rc = check_level(lidar, level, P);
    if (rc) return rc;
    if (!d_maps || !d_est || !d_ranges || !d_out || !d_flags || est_stride < 3) return SLAM2D_E_BADARG;
    const Slam2dLevel& lv = *level;
    if (lv.kmax < lidar->beams || !lv.partials) return SLAM2D_E_BADARG;
    if ((rc = check_field_args(lv, P, true))) return rc;
    hipStream_t s = (hipStream_t)stream;
    int ring_chunks = 0;
    if (options & SLAM2D_MATCH_PRUNE_BY_PRIOR) {
        const int bound = ring_slot_bound(lv, est_moving_dist);
        const int nx = 2 * lv.ncell + 1, nslot = nx * ((nx + 3) / 4);
        if (bound > 0 && 2 * bound <= nslot) ring_chunks = cdiv(bound, WAVE);     // worth it only for a thin ring
    }
    // Below SLAM2D_BEAM_TABLE_MIN beams k_frame_axis is not launched at all: the endpoint kernel's per-particle block does
    // its work (one launch less per level); above, k_frame_axis also tabulates the beam endpoints once per particle.
    static const bool keep_frame_kernel = [] { const char* e = getenv("SLAM2D_FRAME_KERNEL"); return e && atoi(e) == 1; }();
    // ... from SLAM2D_FRAME_MIN_P particles per launch as well (round 6): the merged launch makes every angle block evaluate the
    // cos / sin of all beams and every scatter block the frame; when the launch fills the machine that work costs more than the
    // two launches it saves (config 2: k_endpoints 57 -> 27 + 26 us at 128 particles per launch, the step 0.309 -> 0.305 ms; at 256
    // per launch 0.580 -> 0.561; at 16 per launch the two launches cost 0.113 -> 0.123)
    static const int frame_min_p = [] { const char* e = getenv("SLAM2D_FRAME_MIN_P"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
    const bool framed = lidar->beams >= SLAM2D_BEAM_TABLE_MIN || keep_frame_kernel || lv.occ_gen == 0 || P >= frame_min_p;
    const Slam2dMap* own = framed ? nullptr : d_maps;
    // ... and then the occupied-cell scatter rides in the endpoint launch as well (SLAM2D_MERGE_SCATTER=0: its own launch)
    static const bool merge_scatter = [] { const char* e = getenv("SLAM2D_MERGE_SCATTER"); return !e || atoi(e) != 0; }();
    const bool merged = own && merge_scatter;
    if (lv.occ_gen < 0 || lv.occ_gen > 255) return SLAM2D_E_BADARG;
    if (lv.bnb && lv.bnb != 3) {
        // branch and bound over 4x4 pose tiles: tile bounds + seed tiles, surviving tiles + selection
        if (framed && (rc = launch_frames(*lidar, lv, d_maps, P, d_est, est_stride, d_flags, true, s, d_ranges))) return rc;
        launch_endpoints(*lidar, lv, P, d_est, est_stride, d_ranges, est_moving_dist, d_psi_cs, d_flags, true, false, s, framed, own, merged);
        if ((rc = launch_field(lv, d_maps, P, d_flags, true, s, merged))) return rc;
        const unsigned grid = (unsigned)cdiv(P, 8) * 8 * lv.ntheta;
        if (lv.bnb == 2) {
            StageScope prof(SLAM2D_STAGE_BOUND, s);
            k_bound1<<<grid, WAVE, (size_t)(WAVE * 4 + lv.kmax) * sizeof(int), s>>>(lv, P);
            k_seed<<<P, SEED_THREADS, 0, s>>>(lv, P);
            k_bound2<<<grid, WAVE, 0, s>>>(lv, P);
        } else {
            StageScope prof(SLAM2D_STAGE_BOUND, s);
            if (!launch_bound_lds(lv, P, s))
                k_bound<<<(unsigned)cdiv(P, 8) * 8 * cdiv(lv.ntheta, BOUND_GROUP), WAVE * BOUND_GROUP, 0, s>>>(lv, P);
        }
        {
            StageScope prof(SLAM2D_STAGE_EXACT, s);
            const int nsplit = exact_split(lv, P);
            if (nsplit > 1) k_exact_select<true><<<(unsigned)cdiv(P, 8) * 8 * nsplit, XS_THREADS, 0, s>>>(lv, d_est, est_stride, d_uniform, d_out, P, nsplit);
            else k_exact_select<false><<<P, XS_THREADS, 0, s>>>(lv, d_est, est_stride, d_uniform, d_out, P, 1);
        }
        return launch_status();
    }
    // the endpoints need only the frame, so they run first and tell the field build which tiles matter
    if (framed && (rc = launch_frames(*lidar, lv, d_maps, P, d_est, est_stride, d_flags, true, s, d_ranges))) return rc;
    launch_endpoints(*lidar, lv, P, d_est, est_stride, d_ranges, est_moving_dist, d_psi_cs, d_flags, true, ring_chunks > 0, s, framed, own, merged);
    if ((rc = launch_field(lv, d_maps, P, d_flags, true, s, merged, ring_chunks > 0))) return rc;      // (the ring pass reads the field's maximum)
    if ((rc = launch_scores(lv, P, d_est, est_stride, d_uniform, d_out, s, ring_chunks))) return rc;
    return launch_status();
}

static int launch_update(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const double* d_pose, int32_t pose_stride,
                         const double* d_ranges, const int32_t* d_beam_shift, uint32_t* d_flags, const WeightsJob& wj, void* stream) {
    if (!lidar || !d_maps || !d_pose || !d_ranges || !d_flags || P <= 0 || pose_stride < 3) return SLAM2D_E_BADARG;
    if (lidar->beams < 1 || lidar->beams > SLAM2D_MAX_BEAMS) return SLAM2D_E_TOOLARGE;
    if (!lidar->spoke_band || !lidar->spoke_cells || !lidar->spoke_r || lidar->num_bands < 1 || lidar->lut_w > 65535)
        return SLAM2D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int groups = cdiv(lidar->beams, UPDB_BEAMS);
    StageScope prof(SLAM2D_STAGE_UPDATE, s);
    k_grid_update<<<8 * cdiv(P, 8) * groups + (wj.logw ? 1 : 0), 64 * UPDB_BEAMS, 0, s>>>(*lidar, d_maps, P, d_pose, pose_stride, d_ranges,
                                                                                         d_beam_shift, d_flags, groups, wj);
    return launch_status();
}

int slam2d_grid_update(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const double* d_pose,
                       int32_t pose_stride, const double* d_ranges, const int32_t* d_beam_shift, uint32_t* d_flags,
                       void* stream) {
    return launch_update(lidar, d_maps, P, d_pose, pose_stride, d_ranges, d_beam_shift, d_flags, WeightsJob{}, stream);
}

int slam2d_grid_update_weights(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const double* d_pose,
                               int32_t pose_stride, const double* d_ranges, uint32_t* d_flags, double* d_logw,
                               const double* d_logconf, int32_t logconf_stride, double* d_w, double* d_stats, void* stream) {
    if (!d_logw || !d_w || !d_stats || (d_logconf && logconf_stride < 1)) return SLAM2D_E_BADARG;
    return launch_update(lidar, d_maps, P, d_pose, pose_stride, d_ranges, nullptr, d_flags,
                         WeightsJob{d_logw, d_logconf, logconf_stride, P, d_w, d_stats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, stream);
}

int slam2d_grid_update_weights_local(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const double* d_pose,
                                     int32_t pose_stride, const double* d_ranges, uint32_t* d_flags, double* d_logw,
                                     const double* d_logconf, int32_t logconf_stride, double* d_part, void* stream) {
    if (!d_logw || !d_part || (d_logconf && logconf_stride < 1)) return SLAM2D_E_BADARG;
    return launch_update(lidar, d_maps, P, d_pose, pose_stride, d_ranges, nullptr, d_flags,
                         WeightsJob{d_logw, d_logconf, logconf_stride, P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, d_part}, stream);
}

int slam2d_prior(const double* d_prev_pose, double raw_theta, double prev_raw_theta, int32_t has_turn,
                 double raw_turn, const double* d_heading, int32_t P, double* d_est, double* d_psi_cs,
                 void* stream) {
    if (!d_prev_pose || !d_heading || !d_est || !d_psi_cs || P <= 0) return SLAM2D_E_BADARG;
    k_prior<<<cdiv(P, 64), 64, 0, (hipStream_t)stream>>>(d_prev_pose, raw_theta, prev_raw_theta, has_turn, raw_turn,
                                                         d_heading, P, d_est, d_psi_cs);
    return launch_status();
}

int slam2d_post_match(const Slam2dMatch* d_fine, const Slam2dMatch* d_coarse, int32_t P, double* d_prev_pose,
                      double* d_heading, double* d_logw, double* d_report, void* stream) {
    if (!d_fine || !d_coarse || !d_prev_pose || !d_heading || !d_logw || P <= 0) return SLAM2D_E_BADARG;
    k_post_match<<<cdiv(P, 64), 64, 0, (hipStream_t)stream>>>(d_fine, d_coarse, P, d_prev_pose, d_heading, d_logw, d_report);
    return launch_status();
}

int slam2d_weights_normalize(double* d_logw, const double* d_logconf, int32_t logconf_stride, int32_t N, double* d_w,
                             double* d_stats, void* stream) {
    if (!d_logw || !d_w || !d_stats || N <= 0 || (d_logconf && logconf_stride < 1)) return SLAM2D_E_BADARG;
    k_weights<<<1, 256, 0, (hipStream_t)stream>>>(d_logw, d_logconf, logconf_stride, N, d_w, d_stats, nullptr, nullptr);
    return launch_status();
}

int slam2d_scan_match(const Slam2dLidar* lidar, const Slam2dLevel* coarse, const Slam2dLevel* fine, const Slam2dMap* d_maps,
                      int32_t P, const double* d_prev_pose, double raw_theta, double prev_raw_theta, int32_t has_turn,
                      double raw_turn, const double* d_heading, const double* d_ranges, double est_moving_dist,
                      const double* d_uniform, double* d_est, double* d_psi_cs, Slam2dMatch* d_coarse, Slam2dMatch* d_fine,
                      uint32_t* d_flags, uint32_t options, void* stream) {
    // (SLAM2D_MATCH_PRIOR_READY: d_est / d_psi_cs were written by the previous scan's slam2d_scan_commit_next)
    int rc = (options & SLAM2D_MATCH_PRIOR_READY) ? 0
             : slam2d_prior(d_prev_pose, raw_theta, prev_raw_theta, has_turn, raw_turn, d_heading, P, d_est, d_psi_cs, stream);
    if (rc) return rc;
    rc = slam2d_match(lidar, coarse, d_maps, P, d_est, 3, d_ranges, est_moving_dist, d_psi_cs, d_uniform, d_coarse, d_flags,
                      options & ~SLAM2D_MATCH_PRIOR_READY, stream);
    if (rc) return rc;
    // the fine level is centred on the coarse result and takes its arg-max (matchMax=True), priors off (:65-73)
    return slam2d_match(lidar, fine, d_maps, P, reinterpret_cast<const double*>(d_coarse), (int32_t)(sizeof(Slam2dMatch) / sizeof(double)),
                        d_ranges, est_moving_dist, nullptr, nullptr, d_fine, d_flags, 0u, stream);
}

int slam2d_scan_commit(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const Slam2dMatch* d_fine,
                       const Slam2dMatch* d_coarse, double* d_prev_pose, double* d_heading, double* d_logw, double* d_report,
                       const double* d_ranges, uint32_t* d_flags, double* d_w, double* d_stats, uint32_t* d_flag_snapshot,
                       uint32_t abort_mask, void* stream) {
    return slam2d_scan_commit_next(lidar, d_maps, P, d_fine, d_coarse, d_prev_pose, d_heading, d_logw, d_report, d_ranges, d_flags, d_w,
                                   d_stats, d_flag_snapshot, abort_mask, 0.0, 0.0, 0, 0.0, nullptr, nullptr, stream);
}

int slam2d_scan_commit_next(const Slam2dLidar* lidar, const Slam2dMap* d_maps, int32_t P, const Slam2dMatch* d_fine,
                            const Slam2dMatch* d_coarse, double* d_prev_pose, double* d_heading, double* d_logw, double* d_report,
                            const double* d_ranges, uint32_t* d_flags, double* d_w, double* d_stats, uint32_t* d_flag_snapshot,
                            uint32_t abort_mask, double next_raw_theta, double next_prev_raw_theta, int32_t next_has_turn,
                            double next_raw_turn, double* d_next_est, double* d_next_psi_cs, void* stream) {
    if ((d_next_est != nullptr) != (d_next_psi_cs != nullptr) || (d_next_est && !d_w)) return SLAM2D_E_BADARG;
    if (!d_w) {                                        // sharded filters run their own normaliser (a collective sits in it)
        const int rc = slam2d_post_match(d_fine, d_coarse, P, d_prev_pose, d_heading, d_logw, d_report, stream);
        if (rc) return rc;
        return slam2d_grid_update(lidar, d_maps, P, d_prev_pose, 3, d_ranges, nullptr, d_flags, stream);
    }
    if (!d_fine || !d_coarse || !d_prev_pose || !d_heading || !d_logw || !d_stats || !d_flag_snapshot) return SLAM2D_E_BADARG;
    // ONE launch: the update blocks read the matched poses from d_fine; one extra block does the bookkeeping
    // (k_post_match's work) and then the normaliser, which needs nothing the update writes
    static_assert(sizeof(Slam2dMatch) % sizeof(double) == 0, "Slam2dMatch is read as rows of doubles");
    return launch_update(lidar, d_maps, P, reinterpret_cast<const double*>(d_fine), (int)(sizeof(Slam2dMatch) / sizeof(double)), d_ranges,
                         nullptr, d_flags,
                         WeightsJob{d_logw, nullptr, 1, P, d_w, d_stats, d_flags, d_flag_snapshot, d_fine, d_coarse, d_prev_pose, d_heading,
                                    d_report, nullptr, abort_mask, nullptr, 0, d_next_est, d_next_psi_cs, next_raw_theta,
                                    next_prev_raw_theta, next_raw_turn, next_has_turn}, stream);
}

// ---- particle groups on several streams, one host call per scan (include/slam2d.h, "one scan for several particle GROUPS") ----
// The normaliser merged by the groups' own blocks (Slam2dScan.d_norm_sync): one rank (the sharded merge has an all-gather in front of
// it), no abort decision across groups (a voided scan's groups would not all arrive), the groups' partials in d_parts in group order.
// (round 5, ABI 16: ... or an abort decided behind k_match_arrive / k_abort_gate, Slam2dScan.match_seq != 0)
#define SLAM2D_SYNC_MAX_GROUPS 56       // words 2 .. 57 of the sync block: one per group; 59: the waits' bound; 60-62: tickets and flags
static inline bool abort_on_device(const Slam2dScan& sc) { return sc.d_norm_sync != nullptr && sc.match_seq != 0u; }
static inline bool device_merged(const Slam2dScan& sc) { return sc.d_norm_sync != nullptr && sc.merge && (!sc.abort_mask || sc.match_seq != 0u); }
// ... or only synchronised through those words (merge == 0, the sharded normaliser: slam2d_norm_gate, the collective and
// slam2d_weights_merge_publish follow on the caller's stream)
static inline bool device_synced(const Slam2dScan& sc) { return sc.d_norm_sync != nullptr && (!sc.abort_mask || sc.match_seq != 0u); }
static int groups_check(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* sc, bool commit) {
    if (!lidar || !groups || !sc || G <= 0 || G > 64 || (!sc->d_ranges && !sc->h_ranges)) return SLAM2D_E_BADARG;
    if (sc->match_seq && !sc->d_norm_sync) return SLAM2D_E_BADARG;
    if (sc->h_seq && (!device_merged(*sc) || !sc->h_pack || !sc->d_pack || sc->pack_doubles <= 0)) return SLAM2D_E_BADARG;
    for (int i = 0; i < G; ++i) {
        const Slam2dGroup& g = groups[i];
        if (!g.coarse || !g.d_maps || g.P <= 0 || !g.d_coarse || (g.fine && !g.d_fine) || !g.d_flags || (!g.ev_done && !sc->d_norm_sync)) return SLAM2D_E_BADARG;
        if (g.d_est ? g.est_stride < 3 : (!g.d_prev_pose || !g.d_heading || !g.d_est_out || !g.d_psi_out)) return SLAM2D_E_BADARG;
        if (commit && (!g.d_logw || !g.d_part)) return SLAM2D_E_BADARG;
        if (commit && sc->abort_mask && !g.ev_matched && !abort_on_device(*sc)) return SLAM2D_E_BADARG;
        if (sc->h_ranges && (g.d_est || !g.d_pull)) return SLAM2D_E_BADARG;          // (the pull rides in the closed loop's prior launch)
        if (commit && sc->h_next_ranges && (!sc->h_ranges || !g.d_pull_next || !device_synced(*sc))) return SLAM2D_E_BADARG;
    }
    if (commit) {
        const bool dm = device_merged(*sc);
        if (!dm && !sc->norm_stream && sc->merge) return SLAM2D_E_BADARG;
        if (sc->merge && (!sc->d_logw_all || !sc->d_parts || !sc->d_w || !sc->d_stats || (!dm && !sc->ev_merged) || sc->n_local <= 0 || sc->n_parts <= 0 ||
                          sc->total_particles < sc->n_local)) return SLAM2D_E_BADARG;
        if (dm && sc->n_parts != G) return SLAM2D_E_BADARG;                  // (one partial per group, in group order)
        if (sc->d_norm_sync && (G > SLAM2D_SYNC_MAX_GROUPS || (sc->abort_mask && !sc->match_seq))) return SLAM2D_E_BADARG;  // (2 + G words, 59-62 taken; an abort
        //                                                             decision across groups needs the events or k_match_arrive / k_abort_gate)
        if (sc->abort_mask && (!sc->d_abort_flags || sc->n_abort_flags <= 0)) return SLAM2D_E_BADARG;
    }
    return 0;
}

static int group_match(const Slam2dLidar* lidar, const Slam2dGroup& g, const Slam2dScan& sc, const int G, const bool step = false) {
    hipStream_t s = (hipStream_t)g.stream;
    int rc = 0;
    if (sc.ev_inputs && (rc = (int)hipStreamWaitEvent(s, (hipEvent_t)sc.ev_inputs, 0))) return rc;
    const double* est = g.d_est;
    const double* psi = g.d_psi_cs;
    int stride = g.est_stride;
    const double* ranges = sc.d_ranges;
    const double* uniform = g.d_uniform;
    if (!est) {                                        // closed loop: the pose prior from the previous matched poses
        if (sc.h_ranges) {                             // ... and the scan's ranges pulled from pinned host memory in the same launch
            if (!(sc.options & SLAM2D_MATCH_PRIOR_READY)) {             // (else: the previous commit did both, Slam2dScan.h_next_ranges)
                k_prior_pull<<<1, 256, 0, s>>>(g.d_prev_pose, sc.raw_theta, sc.prev_raw_theta, sc.has_turn, sc.raw_turn, g.d_heading, g.P,
                                               g.d_est_out, g.d_psi_out, sc.h_ranges, lidar->beams, g.d_pull);
                if ((rc = launch_status())) return rc;
            }
            ranges = g.d_pull; uniform = g.h_uniform;
        } else if ((rc = slam2d_prior(g.d_prev_pose, sc.raw_theta, sc.prev_raw_theta, sc.has_turn, sc.raw_turn, g.d_heading, g.P, g.d_est_out,
                                      g.d_psi_out, g.stream))) return rc;
        est = g.d_est_out; psi = g.d_psi_out; stride = 3;
    }
    if ((rc = slam2d_match(lidar, g.coarse, g.d_maps, g.P, est, stride, ranges, sc.est_moving_dist, psi, uniform, g.d_coarse,
                           g.d_flags, sc.options & ~SLAM2D_MATCH_PRIOR_READY, g.stream))) return rc;
    const bool count_in = abort_on_device(sc) && !step && G > 1;       // (one group: its stream orders its commit behind its match)
    if (g.fine) {
        Slam2dLevel last = *g.fine;                    // (the level whose selecting waves count their particles in for the commit's gate)
        if (count_in) last.arrive = sc.d_norm_sync + 61;
        if ((rc = slam2d_match(lidar, &last, g.d_maps, g.P, reinterpret_cast<const double*>(g.d_coarse),
                               (int32_t)(sizeof(Slam2dMatch) / sizeof(double)), ranges, sc.est_moving_dist, nullptr, nullptr,
                               g.d_fine, g.d_flags, 0u, g.stream))) return rc;
    } else if (count_in) {
        k_match_arrive<<<1, 64, 0, s>>>(sc.d_norm_sync, (uint32_t)g.P);
        if ((rc = launch_status())) return rc;
    }
    if (abort_on_device(sc) && !step) return rc;
    // (ev_matched serves a LATER commit call's abort decision; slam2d_groups_step has none, and an event packet between two kernels
    // of a group costs its chain 3 us)
    if (g.ev_matched && !step) rc = (int)hipEventRecord((hipEvent_t)g.ev_matched, s);
    return rc;
}

static int group_commit(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, int i, const Slam2dScan& sc) {
    const Slam2dGroup& g = groups[i];
    hipStream_t s = (hipStream_t)g.stream;
    int rc = 0;
    if (sc.abort_mask && abort_on_device(sc)) {        // the abort is decided over every group's fault bits: wait for every group's match
        if (G > 1) {
            k_abort_gate<<<1, 64, 0, s>>>(sc.d_norm_sync, (uint32_t)sc.n_abort_flags * sc.match_seq, g.d_flags);
            if ((rc = launch_status())) return rc;
        }
    } else if (sc.abort_mask)
        for (int j = 0; j < G; ++j)
            if (j != i && (rc = (int)hipStreamWaitEvent(s, (hipEvent_t)groups[j].ev_matched, 0))) return rc;
    // the previous scan's merge works on the log-weights this launch rewrites: an event behind the merge launch -- or, with
    // Slam2dScan.d_norm_sync, nothing on the stream: the normaliser blocks settle it on the device (normaliser_wait / _arrive)
    const bool device_merge = device_merged(sc), device_sync = device_synced(sc);
    if (!device_sync && sc.wait_merged && sc.ev_merged && (rc = (int)hipStreamWaitEvent(s, (hipEvent_t)sc.ev_merged, 0))) return rc;
    const Slam2dMatch* fin = g.fine ? g.d_fine : g.d_coarse;
    const int md = (int)(sizeof(Slam2dMatch) / sizeof(double));
    WeightsJob wj = g.d_est                            // open loop: weight *= coarse confidence, update at the matched pose
        ? WeightsJob{g.d_logw, &g.d_coarse->log_confidence, md, g.P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, g.d_part, 0u, nullptr, 0}
        // closed loop: slam2d_scan_commit's launch with the normaliser's local half
        : WeightsJob{g.d_logw, nullptr, 1, g.P, nullptr, nullptr, g.d_flags, g.d_flag_snapshot, fin, g.d_coarse, g.d_prev_pose,
                     g.d_heading, g.d_report, g.d_part, sc.abort_mask, sc.d_abort_flags, sc.n_abort_flags};
    if (device_sync) { wj.nsync = sc.d_norm_sync; wj.ngroups = G; wj.gidx = i; }
    if (device_merge) {
        wj.parts_all = sc.d_parts; wj.logw_all = sc.d_logw_all; wj.n_all = sc.n_local; wj.total = (double)sc.total_particles;
        wj.w_all = sc.d_w; wj.stats_all = sc.d_stats;
        if (sc.h_seq) { wj.pack_d = sc.d_pack; wj.pack_h = sc.h_pack; wj.pack_n = sc.pack_doubles; wj.h_seq = sc.h_seq; wj.seq = sc.report_seq; }
    }
    if (device_sync && sc.h_next_ranges && !g.d_est) {  // closed loop: the next scan's prior and ranges ride in this launch's block 0 (needs the
        //                                                 matched poses only: a sharded rank's commit, which does not merge, carries them too)
        wj.next_est = g.d_est_out; wj.next_psi = g.d_psi_out; wj.next_raw_theta = sc.next_raw_theta;
        wj.next_prev_raw_theta = sc.next_prev_raw_theta; wj.next_raw_turn = sc.next_raw_turn; wj.next_has_turn = sc.next_has_turn;
        wj.pull_src = sc.h_next_ranges; wj.pull_dst = g.d_pull_next; wj.pull_n = lidar->beams;
    }
    rc = launch_update(lidar, g.d_maps, g.P, reinterpret_cast<const double*>(fin), md, sc.h_ranges ? g.d_pull : sc.d_ranges, nullptr, g.d_flags, wj, g.stream);
    if (rc || (device_sync && !g.ev_done)) return rc;
    return (int)hipEventRecord((hipEvent_t)g.ev_done, s);
}

static int groups_merge(const Slam2dGroup* groups, int32_t G, const Slam2dScan& sc) {
    if (device_merged(sc)) return 0;                   // the last group's normaliser block has merged (or will)
    if (device_synced(sc)) return 0;                   // (merge == 0: the caller's slam2d_norm_gate orders its stream behind the groups)
    hipStream_t ns = (hipStream_t)sc.norm_stream;
    int rc = 0;
    for (int i = 0; i < G; ++i)
        if ((rc = (int)hipStreamWaitEvent(ns, (hipEvent_t)groups[i].ev_done, 0))) return rc;
    if (!sc.merge) return 0;                           // sharded: the caller's all-gather and slam2d_weights_merge follow on norm_stream
    k_weights_merge<<<1, 256, 0, ns>>>(sc.d_logw_all, sc.n_local, sc.d_parts, sc.n_parts, (double)sc.total_particles, sc.d_w, sc.d_stats,
                                       sc.d_abort_flags, sc.n_abort_flags, sc.d_abort_flags ? sc.abort_mask : 0u);
    if ((rc = launch_status())) return rc;
    return (int)hipEventRecord((hipEvent_t)sc.ev_merged, ns);
}

// One host thread per group beyond the first: a HIP launch costs the calling thread ~4-5 us, a group's scan 7-15 launches, so
// issuing G groups from one thread takes G x 35-70 us -- as long as the device needs for the whole scan from two groups up
// (round 4: 0.076-0.109 ms of a 0.133 ms step with two groups, host-bound with four; the closed loop, twice the launches per
// scan, ran SLOWER in two groups than in one).  The workers spin briefly between scans (a condition variable's wake-up costs as
// much as the job) and sleep when no scan has come for a while.  SLAM2D_GROUP_THREADS=0: everything from the calling thread.
enum { JOB_MATCH = 1, JOB_COMMIT = 2, JOB_STEP = 3 };
static int run_group_job(int kind, const Slam2dLidar* lidar, const Slam2dGroup* groups, int G, int i, const Slam2dScan& sc) {
    int rc = 0;
    if (kind & JOB_MATCH) rc = group_match(lidar, groups[i], sc, G, kind == JOB_STEP);
    if (!rc && (kind & JOB_COMMIT)) rc = group_commit(lidar, groups, G, i, sc);
    return rc;
}
// Host cores this process may really use: the scheduler affinity capped by the cgroup CPU quota (v2 cpu.max, v1 cfs quota) --
// what bench.effective_cores() reports.  A container often sees every core of the host and may use a few.
static int effective_cores() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c >= 1 && c < n) n = c; }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0}; double period = 0.0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) { const int c = (int)(atof(q) / period); if (c < n) n = c < 1 ? 1 : c; }
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long quota = -1, period = 0;
        if (fscanf(g, "%ld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = 0; fclose(h); }
        if (quota > 0 && period > 0) { const int c = (int)(quota / period); if (c < n) n = c < 1 ? 1 : c; }
    }
    return n;
}
// How the groups of a slam2d_groups_* call are issued on this host (decided once).  One worker thread per group beyond the first
// halves the issue time of a scan -- when the threads have cores to run on: a worker polls for ~1-2 ms before it sleeps and the
// caller polls for its workers, i.e. TWO spinning threads per process, and a node runs one process per GPU.  With fewer than 3
// cores per local rank (8 ranks on a 16-core quota: the driver's box) the spinners would take the cores of Python, the HIP
// runtime's signal threads and RCCL's proxy threads: the groups are then issued from the calling thread.  In between (< 6) the
// threads stay but wait politely (short polls, sched_yield).
//   SLAM2D_GROUP_THREADS = 0 / 1 forces the choice; SLAM2D_LOCAL_RANKS (default: LOCAL_WORLD_SIZE, torchrun's, else 1) says how
//   many such processes share the host.
struct GroupPolicy { bool threads; bool polite; int cores, ranks; };
static const GroupPolicy& group_policy() {
    static const GroupPolicy pol = [] {
        GroupPolicy g;
        g.cores = effective_cores();
        const char* r = getenv("SLAM2D_LOCAL_RANKS");
        if (!r || atoi(r) < 1) r = getenv("LOCAL_WORLD_SIZE");
        g.ranks = r && atoi(r) >= 1 ? atoi(r) : 1;
        const int per = g.cores / g.ranks;
        g.threads = per >= 3;
        g.polite = per < 6;
        if (const char* e = getenv("SLAM2D_GROUP_THREADS")) g.threads = atoi(e) != 0;
        return g;
    }();
    return pol;
}

struct GroupWorker {
    std::atomic<int> state{0};           // 0 idle, 1 job posted, 2 done
    std::atomic<bool> sleeping{false};
    std::mutex m;
    std::condition_variable cv;
    int kind = 0, G = 0, i = 0, dev = 0, rc = 0;
    const Slam2dLidar* lidar = nullptr;
    const Slam2dGroup* groups = nullptr;
    const Slam2dScan* sc = nullptr;
    void loop() {
        int cur_dev = -1;
        const int poll = group_policy().polite ? 2000 : 200000;              // ~1-2 ms of polling (polite: ~15 us), then sleep
        for (;;) {
            int spins = 0;
            while (state.load(std::memory_order_acquire) != 1) {
                if (++spins < poll) {
                    // (a spinning thread that the scheduler has put on the caller's CPU must not hold it for a time slice: in the first
                    // bench process of two sessions the four-group closed loop ran 0.28 s per leg instead of 0.20, the one-group legs
                    // beside it did not.  sched_yield returns at once when nobody else wants the CPU)
                    if ((spins & 127) == 127) sched_yield(); else __builtin_ia32_pause();
                    continue;
                }
                std::unique_lock<std::mutex> lk(m);
                sleeping.store(true);
                cv.wait(lk, [&] { return state.load(std::memory_order_acquire) == 1; });
                sleeping.store(false);
            }
            if (dev != cur_dev) { (void)hipSetDevice(dev); cur_dev = dev; }
            rc = run_group_job(kind, lidar, groups, G, i, *sc);
            state.store(2, std::memory_order_release);
        }
    }
};
static GroupWorker* group_worker(int k) {
    static std::mutex mk;
    static GroupWorker* pool[64] = {};
    std::lock_guard<std::mutex> lk(mk);
    if (!pool[k]) {
        pool[k] = new GroupWorker;                       // (never freed: the thread sleeps until the process ends)
        std::thread(&GroupWorker::loop, pool[k]).detach();
    }
    return pool[k];
}
// One call at a time: the workers are a process-wide pool indexed by group number (ctypes releases the GIL, so two Python threads
// could otherwise post into the same worker).  A call may be left RUNNING (slam2d_groups_match_begin: the caller goes on while the
// workers issue the launches -- every group on a worker then, the descriptors copied so that the caller may rewrite its own); the
// next slam2d_groups_* call, or slam2d_groups_join, waits for it first and returns its error.
static std::mutex g_one_call;
static struct { int G = 0; int rc = 0; Slam2dScan scan; Slam2dGroup groups[64]; } g_begun;
static int join_locked() {
    const GroupPolicy& pol = group_policy();
    int rc = g_begun.rc;
    for (int i = 0; i < g_begun.G; ++i) {
        GroupWorker* w = group_worker(i);
        int spins = 0;
        while (w->state.load(std::memory_order_acquire) != 2) {
            if ((++spins & 127) == 127 || (pol.polite && spins > 256)) sched_yield(); else __builtin_ia32_pause();
        }
        if (!rc) rc = w->rc;
        w->state.store(0, std::memory_order_release);
    }
    g_begun.G = 0; g_begun.rc = 0;
    return rc;
}
static void post(GroupWorker* g, int kind, const Slam2dLidar* lidar, const Slam2dGroup* groups, int G, int i, const Slam2dScan* scan, int dev) {
    g->kind = kind; g->lidar = lidar; g->groups = groups; g->G = G; g->i = i; g->sc = scan; g->dev = dev;
    g->state.store(1, std::memory_order_seq_cst);
    if (g->sleeping.load()) { { std::lock_guard<std::mutex> lk(g->m); } g->cv.notify_one(); }
}
static int run_groups(int kind, const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan, const bool leave_running = false) {
    std::lock_guard<std::mutex> serial(g_one_call);
    int rc = join_locked();
    if (rc) return rc;
    const GroupPolicy& pol = group_policy();
    if (!pol.threads || (G < 2 && !leave_running)) {
        for (int i = 0; i < G && !rc; ++i) rc = run_group_job(kind, lidar, groups, G, i, *scan);
        return rc;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (leave_running) {
        g_begun.scan = *scan;
        for (int i = 0; i < G; ++i) g_begun.groups[i] = groups[i];
        for (int i = 0; i < G; ++i) post(group_worker(i), kind, lidar, g_begun.groups, G, i, &g_begun.scan, dev);
        g_begun.G = G;
        return 0;
    }
    for (int i = 1; i < G; ++i) post(group_worker(i), kind, lidar, groups, G, i, scan, dev);
    rc = run_group_job(kind, lidar, groups, G, 0, *scan);
    for (int i = 1; i < G; ++i) {
        GroupWorker* w = group_worker(i);
        int spins = 0;
        while (w->state.load(std::memory_order_acquire) != 2) {
            if ((++spins & 127) == 127 || (pol.polite && spins > 256)) sched_yield(); else __builtin_ia32_pause();
        }
        if (!rc) rc = w->rc;
        w->state.store(0, std::memory_order_release);
    }
    return rc;
}

int slam2d_groups_match_begin(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan) {
    const int rc = groups_check(lidar, groups, G, scan, false);
    return rc ? rc : run_groups(JOB_MATCH, lidar, groups, G, scan, true);
}

int slam2d_groups_join(void) {
    std::lock_guard<std::mutex> serial(g_one_call);
    return join_locked();
}

int slam2d_host_wait_seq(const uint32_t* h_seq, uint32_t want, double timeout_s) {
    if (!h_seq) return SLAM2D_E_BADARG;
    const auto t0 = std::chrono::steady_clock::now();
    const GroupPolicy& pol = group_policy();
    const bool polite = pol.polite || !pol.threads;        // (few cores per rank: the runtime's and RCCL's threads need them too)
    for (unsigned spins = 0;; ++spins) {
        if ((int32_t)(__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) - want) >= 0) return 0;
        if ((spins & (polite ? 63u : 255u)) == (polite ? 63u : 255u)) sched_yield(); else __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return SLAM2D_E_TIMEOUT;
    }
}

/* how slam2d_groups_* issue their groups on this host: out[0] = effective cores, out[1] = local ranks assumed, out[2] = 1 if a
 * worker thread per group is used, out[3] = 1 if the threads wait politely */
int slam2d_group_policy(int32_t* out4) {
    if (!out4) return SLAM2D_E_BADARG;
    const GroupPolicy& g = group_policy();
    out4[0] = g.cores; out4[1] = g.ranks; out4[2] = g.threads ? 1 : 0; out4[3] = g.polite ? 1 : 0;
    return 0;
}

int slam2d_groups_match(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan) {
    const int rc = groups_check(lidar, groups, G, scan, false);
    return rc ? rc : run_groups(JOB_MATCH, lidar, groups, G, scan);
}

int slam2d_groups_commit(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan) {
    int rc = groups_check(lidar, groups, G, scan, true);
    if (!rc) rc = run_groups(JOB_COMMIT, lidar, groups, G, scan);     // (every ev_matched was recorded by the match call before this one)
    return rc ? rc : groups_merge(groups, G, *scan);
}

int slam2d_groups_step(const Slam2dLidar* lidar, const Slam2dGroup* groups, int32_t G, const Slam2dScan* scan) {
    int rc = groups_check(lidar, groups, G, scan, true);
    if (!rc && scan->abort_mask) return SLAM2D_E_BADARG;      // (an abort needs every match before any commit: use the two calls)
    if (!rc) rc = run_groups(JOB_STEP, lidar, groups, G, scan);
    return rc ? rc : groups_merge(groups, G, *scan);
}

int slam2d_weights_local(double* d_logw, const double* d_logconf, int32_t logconf_stride, int32_t N, double* d_part,
                         void* stream) {
    if (!d_logw || !d_part || N <= 0 || (d_logconf && logconf_stride < 1)) return SLAM2D_E_BADARG;
    k_weights_local<<<1, 256, 0, (hipStream_t)stream>>>(d_logw, d_logconf, logconf_stride, N, d_part);
    return launch_status();
}

int slam2d_weights_merge(double* d_logw, int32_t N, const double* d_parts, int32_t world, int64_t total_particles,
                         double* d_w, double* d_stats, void* stream) {
    if (!d_logw || !d_parts || !d_w || !d_stats || N <= 0 || world <= 0 || total_particles < N) return SLAM2D_E_BADARG;
    k_weights_merge<<<1, 256, 0, (hipStream_t)stream>>>(d_logw, N, d_parts, world, (double)total_particles, d_w, d_stats, nullptr, 0, 0u);
    return launch_status();
}

int slam2d_norm_gate(uint32_t* d_norm_sync, int32_t G, void* stream) {
    if (!d_norm_sync || G <= 0 || G > SLAM2D_SYNC_MAX_GROUPS) return SLAM2D_E_BADARG;
    k_norm_gate<<<1, 64, 0, (hipStream_t)stream>>>(d_norm_sync, G);
    return launch_status();
}

int slam2d_weights_merge_publish(double* d_logw, int32_t N, const double* d_parts, int32_t world, int64_t total_particles,
                                 double* d_w, double* d_stats, uint32_t* d_norm_sync, void* stream) {
    if (!d_logw || !d_parts || !d_w || !d_stats || !d_norm_sync || N <= 0 || world <= 0 || total_particles < N) return SLAM2D_E_BADARG;
    k_weights_merge<<<1, 256, 0, (hipStream_t)stream>>>(d_logw, N, d_parts, world, (double)total_particles, d_w, d_stats, nullptr, 0, 0u, d_norm_sync);
    return launch_status();
}

int slam2d_weights_merge_publish_report(double* d_logw, int32_t N, const double* d_parts, int32_t world, int64_t total_particles,
                                        double* d_w, double* d_stats, uint32_t* d_norm_sync, const double* d_pack, double* h_pack,
                                        int32_t pack_doubles, uint32_t* h_seq, uint32_t report_seq, void* stream) {
    if (!d_logw || !d_parts || !d_w || !d_stats || !d_norm_sync || N <= 0 || world <= 0 || total_particles < N) return SLAM2D_E_BADARG;
    if (!d_pack || !h_pack || !h_seq || pack_doubles <= 0) return SLAM2D_E_BADARG;
    k_weights_merge<<<1, 256, 0, (hipStream_t)stream>>>(d_logw, N, d_parts, world, (double)total_particles, d_w, d_stats, nullptr, 0, 0u, d_norm_sync,
                                                        d_pack, h_pack, pack_doubles, h_seq, report_seq);
    return launch_status();
}

int slam2d_gather_maps(const Slam2dMap* d_src, const Slam2dMap* d_dst, const int32_t* d_index, int32_t P,
                       int64_t cells_per_map, void* stream) {
    if (!d_src || !d_dst || !d_index || P <= 0 || cells_per_map <= 0) return SLAM2D_E_BADARG;
    const int gx = (int)((cells_per_map + 256 * 8 - 1) / (256 * 8));
    k_gather_maps<<<dim3(gx < 1 ? 1 : gx, P), 256, 0, (hipStream_t)stream>>>(d_src, d_dst, d_index);
    return launch_status();
}

int slam2d_map_refresh_bits(const Slam2dMap* d_maps, const int32_t* d_index, int32_t n, void* stream) {
    if (!d_maps || n <= 0) return SLAM2D_E_BADARG;
    k_refresh_bits<<<dim3(1024, n), 256, 0, (hipStream_t)stream>>>(d_maps, d_index);
    return launch_status();
}

int slam2d_map_image(const Slam2dMap* d_maps, int32_t p, int32_t x0, int32_t x1, int32_t y0, int32_t y1, int32_t flipud,
                     double* d_out, uint8_t* d_out_u8, void* stream) {
    if (!d_maps || p < 0 || x0 < 0 || y0 < 0 || x1 <= x0 || y1 <= y0 || (!d_out && !d_out_u8)) return SLAM2D_E_BADARG;
    const long long n = (long long)(x1 - x0) * (y1 - y0);
    long long blocks = (n + 256 * 4 - 1) / (256 * 4);
    if (blocks > 65535) blocks = 65535;
    k_map_image<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_maps, p, x0, y0, x1 - x0, y1 - y0, flipud, d_out, d_out_u8);
    return launch_status();
}

int slam2d_map_fill(uint32_t* d_cells, int64_t n, uint32_t value, void* stream) {
    if (!d_cells || n <= 0) return SLAM2D_E_BADARG;
    long long blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 65535) blocks = 65535;
    k_fill<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(d_cells, (long long)n, value);
    return launch_status();
}

// expandOccupancyGrid (Utils/OccupancyGrid.py:59-100) on the device, once per growth SEQUENCE: the new count array in one pass --
// old content at its shifted place (np.insert / np.append of fresh columns / rows: :70-71, :81-82), SLAM2D_INIT_CELL everywhere
// else, the pitch padding included -- and the new occupancy bits from the cells as they are written (k_refresh_bits' test).
// One wave = 64 consecutive cells of a row: a 256-byte read where the old map covers them, a 256-byte write, one ballot.
__global__ __launch_bounds__(256) void k_map_grow(const Slam2dMap o, const Slam2dMap n, const int d_row, const int d_col) {
    const int lane = threadIdx.x & 63;
    const int groups_per_row = (n.pitch + 63) >> 6;
    const long long ngroups = (long long)n.rows * groups_per_row;
    for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < ngroups; g += (long long)gridDim.x * 4) {
        const int row = (int)(g / groups_per_row), c0 = (int)(g - (long long)row * groups_per_row) << 6;
        const int col = c0 + lane;
        const int orow = row - d_row, ocol = col - d_col;
        const bool inside = orow >= 0 && orow < o.rows && ocol >= 0 && ocol < o.cols && col < n.cols;
        bool occ = false;
        if (n.wide) {
            unsigned long long v = ((unsigned long long)(SLAM2D_INIT_CELL >> 16) << 32) | (SLAM2D_INIT_CELL & 0xffffu);
            if (inside) v = reinterpret_cast<const unsigned long long*>(o.cells)[(size_t)orow * o.pitch + ocol];
            if (col < n.pitch) reinterpret_cast<unsigned long long*>(n.cells)[(size_t)row * n.pitch + col] = v;
            occ = col < n.cols && 2ull * (v >> 32) > (v & 0xffffffffull);
        } else {
            uint32_t v = SLAM2D_INIT_CELL;
            if (inside) v = o.cells[(size_t)orow * o.pitch + ocol];
            if (col < n.pitch) n.cells[(size_t)row * n.pitch + col] = v;
            occ = col < n.cols && 2u * (v >> 16) > (v & 0xffffu);                   // :29-31
        }
        const unsigned long long mask = __ballot(occ);
        if (lane < 2 && (c0 >> 5) + lane < n.bits_pitch)
            n.occ_bits[(size_t)row * n.bits_pitch + (c0 >> 5) + lane] = (uint32_t)(mask >> (32 * lane));
    }
}

int slam2d_map_grow(const Slam2dMap* old_map, const Slam2dMap* new_map, int32_t d_row, int32_t d_col, void* stream) {
    if (!old_map || !new_map || !old_map->cells || !new_map->cells || !new_map->occ_bits || d_row < 0 || d_col < 0) return SLAM2D_E_BADARG;
    if (old_map->wide != new_map->wide || new_map->rows < old_map->rows + d_row || new_map->cols < old_map->cols + d_col ||
        new_map->pitch < new_map->cols || new_map->bits_pitch * 32 < new_map->cols) return SLAM2D_E_BADARG;
    const long long ngroups = (long long)new_map->rows * ((new_map->pitch + 63) >> 6);
    long long blocks = (ngroups + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    k_map_grow<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(*old_map, *new_map, d_row, d_col);
    return launch_status();
}

// ---- stage profiling ----
int slam2d_prof_enable(uint32_t stage_mask, int32_t capacity) {
    if (capacity <= 0) return SLAM2D_E_BADARG;
    for (int st = 0; st < SLAM2D_STAGE_COUNT; ++st) {
        StageProf& p = g_prof[st];
        if (!((stage_mask >> st) & 1u)) continue;
        if (p.capacity < capacity) {
            for (int i = 0; i < p.capacity; ++i) { (void)hipEventDestroy(p.start[i]); (void)hipEventDestroy(p.stop[i]); }
            delete[] p.start; delete[] p.stop;
            p.start = new hipEvent_t[capacity]; p.stop = new hipEvent_t[capacity];
            for (int i = 0; i < capacity; ++i) {
                hipError_t e = hipEventCreate(&p.start[i]);
                if (e == hipSuccess) e = hipEventCreate(&p.stop[i]);
                if (e != hipSuccess) { p.capacity = 0; return (int)e; }
            }
            p.capacity = capacity;
        }
        p.used = 0;
        p.seen = 0;
    }
    g_prof_mask = stage_mask;
    return 0;
}

int slam2d_prof_every(int32_t every) {
    if (every < 1) return SLAM2D_E_BADARG;
    g_prof_every = every;
    return 0;
}

int slam2d_prof_collect(int32_t stage, double* total_ms, int32_t* launches) {
    if (stage < 0 || stage >= SLAM2D_STAGE_COUNT || !total_ms || !launches) return SLAM2D_E_BADARG;
    StageProf& p = g_prof[stage];
    double tot = 0.0;
    for (int i = 0; i < p.used; ++i) {
        hipError_t e = hipEventSynchronize(p.stop[i]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, p.start[i], p.stop[i]);
        if (e != hipSuccess) return (int)e;
        tot += ms;
    }
    *total_ms = tot;
    *launches = p.used;
    p.used = 0;
    return 0;
}

void slam2d_prof_disable(void) { g_prof_mask = 0; }

// cos / sin exactly as k_endpoints evaluates them for the beam angles (ocml fp64): lets the tests measure the
// distance to NumPy's libm on the same angles (deviation (i))
__global__ void k_sincos(const double* __restrict__ a, int n, double* c, double* s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { c[i] = cos(a[i]); s[i] = sin(a[i]); }
}
int slam2d_device_sincos(const double* d_angles, int32_t n, double* d_cos, double* d_sin, void* stream) {
    if (!d_angles || !d_cos || !d_sin || n <= 0) return SLAM2D_E_BADARG;
    k_sincos<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(d_angles, n, d_cos, d_sin);
    return launch_status();
}

#ifdef SLAM2D_DEBUG_CLOCK
int slam2d_debug_clock(long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg_clock), sizeof(long long) * 64);
}
#endif

// ---- ordering between streams (no timing): events a host driver uses to chain particle groups on several streams ----
void* slam2d_event_create(void) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void*)e;
}
void slam2d_event_destroy(void* event) { if (event) (void)hipEventDestroy((hipEvent_t)event); }
int slam2d_event_record(void* event, void* stream) { return event ? (int)hipEventRecord((hipEvent_t)event, (hipStream_t)stream) : SLAM2D_E_BADARG; }
int slam2d_stream_wait_event(void* stream, void* event) { return event ? (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) : SLAM2D_E_BADARG; }

// ---- a batch of streams for particle groups ----
// Which HARDWARE queue a HIP stream sits on decides whether two groups overlap or take turns: the runtime keeps at most
// GPU_MAX_HW_QUEUES of them and hands a new stream the least-referenced one.  Streams created one after the other, in one batch and
// each used once at once, land on distinct queues (up to that limit); streams picked one by one out of a framework's round-robin pool
// at different times do not (round 5: the closed loop in four groups ran 0.21 s with the first five pooled streams of a process and
// 0.42-0.50 s with the pool's later ones; the bench's probe legs 0.12 against 0.25 ms per scan).
__global__ void k_touch() {}
int slam2d_streams_create(void** out, int32_t n) {
    if (!out || n <= 0 || n > 64) return SLAM2D_E_BADARG;
    for (int i = 0; i < n; ++i) {
        hipStream_t st;
        const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) { for (int j = 0; j < i; ++j) (void)hipStreamDestroy((hipStream_t)out[j]); return (int)e; }
        out[i] = (void*)st;
        k_touch<<<1, 64, 0, st>>>();                   // (first use = queue assignment, in creation order)
    }
    for (int i = 0; i < n; ++i) (void)hipStreamSynchronize((hipStream_t)out[i]);
    return launch_status();
}
void slam2d_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }

// ---- plain event timer ----
struct Timer { hipEvent_t a, b; };
void* slam2d_timer_create(void) {
    Timer* t = new Timer;
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) { delete t; return nullptr; }
    return t;
}
void slam2d_timer_destroy(void* timer) {
    if (!timer) return;
    Timer* t = (Timer*)timer;
    (void)hipEventDestroy(t->a); (void)hipEventDestroy(t->b);
    delete t;
}
int slam2d_timer_start(void* timer, void* stream) { return timer ? (int)hipEventRecord(((Timer*)timer)->a, (hipStream_t)stream) : SLAM2D_E_BADARG; }
int slam2d_timer_stop(void* timer, void* stream) { return timer ? (int)hipEventRecord(((Timer*)timer)->b, (hipStream_t)stream) : SLAM2D_E_BADARG; }
int slam2d_timer_elapsed_ms(void* timer, float* ms) {
    if (!timer || !ms) return SLAM2D_E_BADARG;
    Timer* t = (Timer*)timer;
    hipError_t e = hipEventSynchronize(t->b);
    if (e != hipSuccess) return (int)e;
    return (int)hipEventElapsedTime(ms, t->a, t->b);
}

}  // extern "C"
