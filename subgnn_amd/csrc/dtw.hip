// Structure similarity 1/(1+fastdtw) (a11): pyramids, the general and the register-resident DP kernels.
#include "common.h"
#include <type_traits>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// a11  1 / (1 + fastdtw(x, y, radius=1, dist=calc_dist))  (reference SubGNN/gamma.py:51-59)
//
// One lane per (component, anchor) pair; fp64 DP.  fastdtw's recursion is unrolled bottom-up:
//   * a pre-kernel builds the halved series ("pyramid") of every x row and every y row ONCE
//     (each x row meets every anchor, each anchor every x row), x transposed so that the lanes
//     of a wavefront -- consecutive components, same anchor -- read it coalesced, y read
//     wave-uniformly;
//   * per pair, the coarsest level (either length < 3) runs a full-window DTW, and each finer
//     level derives its window from the coarser warp path.  Because a warp path is monotone, the
//     published expand_window (dilate by radius 1, project to the fine grid, keep one contiguous
//     run per row starting no earlier than the previous row's) reduces to per-row bounds
//       lo_i = max(0, 2*(first_col[max(ci-1,0)] - 1)),  hi_i = min(ly-1, 2*(last_col[min(ci+1,lxc-1)] + 1) + 1)
//     with ci = i/2 and first/last_col the coarse path's column range per coarse row, so only
//     those two small arrays travel between levels;
//   * the DP keeps two rolling rows (the diagonal predecessor stays in a register) and a 2-bit
//     predecessor code per window cell for the backtrack.
// Per-lane state (~1.4 KB at 20 x 50) lives in a caller workspace, element-interleaved across
// lanes so that lanes in lockstep touch consecutive addresses; the resident thread count is kept
// small enough for that scratch to stay in the Infinity Cache (the first version's 3.5 KB x 131k
// lanes spilled to HBM and waited on it 78 % of the time).  VALU / latency-bound (one fp64
// divide per cell); not an HBM kernel.
// ---------------------------------------------------------------------------------------------
#define DTW_THREADS 256
#ifndef DTW_BLOCKS
#define DTW_BLOCKS (256 * 2)
#endif
#define DTW_NT ((int64_t)DTW_THREADS * DTW_BLOCKS)
#ifndef DTW_REG_BLOCKS
#define DTW_REG_BLOCKS (256 * 32)     // register variant: its scratch is LDS; many more workgroups than fit at once, so the tail of the launch is short (4096 / 8192: 4.65 / 4.51 ms)
#endif
#define DTW_REG_NT ((int64_t)DTW_THREADS * DTW_REG_BLOCKS)
#define DTW_MAX_LEVELS 16

struct DtwLayout {
    int64_t MX, MY;
    int64_t XL, YL;                // pyramid lengths per sequence (sum of M >> k)
    int64_t n_dbl;                 // per lane: prev(MY) cur(MY)
    int64_t n_i32;                 // per lane: rowstart(MX) lohi(MX) firstlast[2](MX each)
    int64_t n_dir;                 // per lane: ceil(MX*MY/16) words of 2-bit codes
    int64_t xoff[DTW_MAX_LEVELS], yoff[DTW_MAX_LEVELS];
};

static inline DtwLayout dtw_layout(int64_t MX, int64_t MY) {
    DtwLayout L;
    L.MX = MX; L.MY = MY;
    int64_t xo = 0, yo = 0;
    for (int k = 0; k < DTW_MAX_LEVELS; ++k) {
        L.xoff[k] = xo; L.yoff[k] = yo;
        xo += (MX >> k) > 0 ? (MX >> k) : 0;
        yo += (MY >> k) > 0 ? (MY >> k) : 0;
    }
    L.XL = xo; L.YL = yo;
    L.n_dbl = 2 * MY;
    L.n_i32 = 4 * MX;
    L.n_dir = (MX * MY + 15) / 16;
    return L;
}

static inline int64_t dtw_align8(int64_t b) { return (b + 7) / 8 * 8; }

// scratch both kernels carve out of the caller's workspace: the general kernel's per-lane state, or -- register variant with an
// anchor series too long for its predecessor words to sit in LDS -- one 32-bit word per column of the backtracked levels and lane
static inline bool dtw_words_in_lds(int64_t max_y_len) { return (max_y_len >> 1) * DTW_THREADS * 4 <= 48 * 1024; }
static inline int64_t dtw_scratch_bytes(const DtwLayout& L) {
    const int64_t lane = L.n_dbl * 8 + dtw_align8(L.n_i32 * 4) + dtw_align8(L.n_dir * 4);
    int64_t scratch = DTW_NT * lane;
    const int64_t reg = dtw_words_in_lds(L.MY) ? 0 : dtw_align8(DTW_REG_NT * (L.YL - L.MY) * 4);
    return reg > scratch ? reg : scratch;
}

extern "C" int64_t sgnn_dtw_workspace_bytes(int64_t n_x, int64_t max_x_len, int64_t n_y, int64_t max_y_len) {
    if (max_x_len < 1) max_x_len = 1;
    if (max_y_len < 1) max_y_len = 1;
    const DtwLayout L = dtw_layout(max_x_len, max_y_len);
    return dtw_scratch_bytes(L)
         + 2 * (n_x * L.XL * 8 + n_y * L.YL * 8) + dtw_align8(n_x * 4) + dtw_align8(n_y * 4);   // value + reciprocal pyramids
}

// pyramid of one series per thread.  transposed != 0: element e of sequence s at out[e * n + s].
// rec (same layout) receives 1 / (value + 1), correctly rounded: the register kernel's cost
// function divides by multiplying with it (see dtw_cost_rcp).
// order (nullable): position s of the output holds sequence order[s] -- the register kernel walks the
// x rows in the caller's processing order, and with the pyramids laid out in that order the lanes of
// a wavefront read consecutive addresses instead of gathering 64 cache lines per load.
__global__ void dtw_pyramid_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ val, int64_t n,
                                   int64_t M, int64_t PL, int transposed, double* __restrict__ out,
                                   double* __restrict__ rec, int32_t* __restrict__ len_out,
                                   const int32_t* __restrict__ order)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = order ? order[s] : s;
        const int64_t b = ptr[src];
        int len = (int)(ptr[src + 1] - b);
        len_out[s] = len;
#define PYI(e) (transposed ? (int64_t)(e) * n + s : s * PL + (e))
#define PY(e) out[PYI(e)]
        for (int i = 0; i < len; ++i) { const double v = (double)val[b + i]; PY(i) = v; rec[PYI(i)] = 1.0 / (v + 1.0); }
        int64_t off = 0;
        int k = 0;
        while (len >= 2 && k + 1 < DTW_MAX_LEVELS) {
            const int64_t noff = off + (M >> k);
            const int nlen = len / 2;
            for (int i = 0; i < nlen; ++i) {
                const double v = (PY(off + 2 * i) + PY(off + 2 * i + 1)) / 2.0;
                PY(noff + i) = v;
                rec[PYI(noff + i)] = 1.0 / (v + 1.0);
            }
            off = noff; len = nlen; ++k;
        }
#undef PY
#undef PYI
    }
}

// The same pyramid with one WAVEFRONT per series (values staged in LDS, a level's elements spread over the lanes): for
// the few long series of the anchor-patch side (210 series of 50 values on the benchmark) one thread per series is a
// serial chain of ~100 fp64 divisions and dependent global round trips -- 80 us per call, twice per pass.
__global__ __launch_bounds__(256) void dtw_pyramid_wave_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ val,
                                                               int64_t n, int64_t M, int64_t PL, int transposed,
                                                               double* __restrict__ out, double* __restrict__ rec,
                                                               int32_t* __restrict__ len_out, const int32_t* __restrict__ order)
{
    extern __shared__ double pyr_sh[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = blockIdx.x * 4ll + wave;
    if (s >= n) return;                                   // (whole wavefronts leave; no workgroup barrier below)
    double* buf = pyr_sh + (int64_t)wave * 2 * M;
    const int64_t src = order ? order[s] : s;
    const int64_t b = ptr[src];
    int len = (int)(ptr[src + 1] - b);
    if (lane == 0) len_out[s] = len;
#define PYI(e) (transposed ? (int64_t)(e) * n + s : s * PL + (e))
    for (int i = lane; i < len; i += 64) {
        const double v = (double)val[b + i];
        buf[i] = v;
        out[PYI(i)] = v;
        rec[PYI(i)] = 1.0 / (v + 1.0);
    }
    int64_t off = 0;
    int k = 0;
    while (len >= 2 && k + 1 < DTW_MAX_LEVELS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t noff = off + (M >> k);
        const int nlen = len / 2;
        for (int i = lane; i < nlen; i += 64) {
            const double v = (buf[off + 2 * i] + buf[off + 2 * i + 1]) / 2.0;
            buf[noff + i] = v;
            out[PYI(noff + i)] = v;
            rec[PYI(noff + i)] = 1.0 / (v + 1.0);
        }
        off = noff; len = nlen; ++k;
    }
#undef PYI
}

// series per launch below which (and lengths up to which) the wavefront-per-series form is used
#define DTW_PYR_WAVE_MAX_SERIES 8192
#define DTW_PYR_WAVE_MAX_LEN 1024

static void dtw_launch_pyramid(const int64_t* ptr, const int32_t* val, int64_t n, int64_t M, int64_t PL, int transposed,
                               double* out, double* rec, int32_t* len_out, const int32_t* order, hipStream_t st)
{
    if (n <= DTW_PYR_WAVE_MAX_SERIES && M <= DTW_PYR_WAVE_MAX_LEN)
        hipLaunchKernelGGL(dtw_pyramid_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)(4 * 2 * M * sizeof(double)), st,
                           ptr, val, n, M, PL, transposed, out, rec, len_out, order);
    else
        hipLaunchKernelGGL(dtw_pyramid_kernel, dim3(sgnn_grid_for(n, 256)), dim3(256), 0, st, ptr, val, n, M, PL, transposed, out,
                           rec, len_out, order);
}

__device__ static inline double dtw_cost(double a, double b) {            // gamma.py:51-52
    const double mx = a > b ? a : b, mn = a > b ? b : a;
    return (mx + 1.0) / (mn + 1.0) - 1.0;
}

// The same cost from a1 = a + 1, b1 = b + 1 and their correctly rounded reciprocals ra, rb, without
// a divide instruction sequence: q0 = RN(mx * r), rem = mx - q0 * mn (exact in an fma),
// q = RN(q0 + rem * r) is the correctly rounded quotient mx / mn when r = RN(1 / mn) (Markstein's
// division step; it can only fail for divisors whose significand is all ones, and mn is a small
// dyadic rational here).  tests/test_oracle_integer.py::test_reciprocal_division_is_exact runs the
// identity exhaustively over the integer range and on 10^7 random dyadic pairs on the CPU.
__device__ __forceinline__ double dtw_cost_rcp(double a1, double ra, double b1, double rb) {
    // max / min with ONE division step (round 5; rounds 2-4 formed both quotients and took the larger: 8 fp64 instructions, this
    // is 7, and every one of them issues at half rate on gfx950): mx = max(a1, b1), mn = min(a1, b1), and the correctly rounded
    // reciprocal of mn is max(ra, rb) -- rounding is monotone, so a1 <= b1 implies RN(1 / a1) >= RN(1 / b1).  The quotient is
    // then the same correctly rounded mx / mn >= 1 the larger of the two quotients was (the direction the CPU test covers).
    const double mx = fmax(a1, b1), mn = fmin(a1, b1), r = fmax(ra, rb);
    const double q0 = __dmul_rn(mx, r);
    const double q = __fma_rn(__fma_rn(-q0, mn, mx), r, q0);
    return __dadd_rn(q, -1.0);
}

// A cost for a cell outside the lane's window: only the HIGH word is replaced (one v_cndmask instead of the two a
// 64-bit select of INF takes), giving a finite value >= 2^1023 whatever the low word holds.  Such a cell then carries
// min(...) + BIG = BIG or INF: it loses every later comparison against a reachable cell, exactly like INF (no product
// or difference is ever taken of these values, so no NaN can arise).
__device__ __forceinline__ double dtw_mask_cost(bool in, double dt) {
    return __hiloint2double(in ? __double2hiint(dt) : 0x7fe00000, __double2loint(dt));
}

__global__ __launch_bounds__(DTW_THREADS) void dtw_similarity_kernel(
    const double* __restrict__ xpyr, const int32_t* __restrict__ xlen, int64_t n_x,
    const double* __restrict__ ypyr, const int32_t* __restrict__ ylen, int64_t n_y,
    int tie_order, float* __restrict__ out, double* __restrict__ wd, int32_t* __restrict__ wi,
    uint32_t* __restrict__ wb, DtwLayout L)
{
    const int64_t NT = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
#define WD(k) wd[(int64_t)(k) * NT + tid]
#define WI(k) wi[(int64_t)(k) * NT + tid]
#define WB(k) wb[(int64_t)(k) * NT + tid]
#define ROWSTART(i) WI(i)
#define LOHI(i) WI(L.MX + (i))
#define FL(h, i) WI((2 + (h)) * L.MX + (i))
    // predecessor codes: 0 = (i-1,j), 1 = (i,j-1), 2 = (i-1,j-1); evaluation order per tie_order
    // tie_order 0 / 1: first minimum over the three sums in that order; 2: the predecessor costs compared with <=
    // (diagonal, then (i-1,j), then (i,j-1)) before the distance is added (oracle/fastdtw_restate.py)
    const int o0 = tie_order == 0 ? 0 : 2, o1 = tie_order == 0 ? 1 : 0, o2 = tie_order == 0 ? 2 : 1;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const int64_t total = n_x * n_y;
    for (int64_t pair = tid; pair < total; pair += NT) {
        // consecutive lanes: consecutive components, same anchor
        const int64_t a = pair / n_x, r = pair % n_x;
        const int lx0 = xlen[r], ly0 = ylen[a];
        if (lx0 == 0 || ly0 == 0) { out[r * n_y + a] = 0.f; continue; }     // padded row: PAD (SubGNN.py:831)
        const double* __restrict__ yp = ypyr + a * L.YL;
        int n_levels = 1;
        {
            int lx = lx0, ly = ly0;
            while (lx >= 3 && ly >= 3) { lx >>= 1; ly >>= 1; ++n_levels; }
        }
        double result = 0.0;
        for (int lev = n_levels - 1; lev >= 0; --lev) {
            const int lx = lx0 >> lev, ly = ly0 >> lev;
            const int64_t xo = L.xoff[lev], yo = L.yoff[lev];
            const int hc = lev & 1, hp = (lev + 1) & 1;        // ping-pong halves of first/last
            if (lev == n_levels - 1) {
                int cells = 0;
                for (int i = 0; i < lx; ++i) { LOHI(i) = (ly - 1) << 16; ROWSTART(i) = cells; cells += ly; }
            } else {
                const int lxc = lx0 >> (lev + 1);
                int prev_lo = 0, cells = 0;
                for (int i = 0; i < lx; ++i) {
                    const int ci = i >> 1;
                    const int ca = ci - 1 < 0 ? 0 : (ci - 1 > lxc - 1 ? lxc - 1 : ci - 1);
                    const int cb = ci + 1 > lxc - 1 ? lxc - 1 : ci + 1;
                    int lo = 2 * ((FL(hp, ca) & 0xffff) - 1);
                    int hi = 2 * ((FL(hp, cb) >> 16) + 1) + 1;
                    if (lo < prev_lo) lo = prev_lo;
                    if (lo < 0) lo = 0;
                    if (hi > ly - 1) hi = ly - 1;
                    if (hi < lo) { lo = 1; hi = 0; }           // empty row marker
                    LOHI(i) = (hi << 16) | lo;
                    ROWSTART(i) = cells;
                    if (hi >= lo) { cells += hi - lo + 1; prev_lo = lo; }
                }
            }
            // DP over the window, row-major (any topological order gives the same cells)
            int64_t prow = 0, crow = L.MY;
            int plo = 0, phi = -1;
            uint32_t acc = 0;
            int cell = 0;
            for (int i = 0; i < lx; ++i) {
                const int lohi = LOHI(i);
                const int lo = lohi & 0xffff, hi = lohi >> 16;
                const double xi = xpyr[(xo + i) * n_x + r];
                double left = INF;
                double diag = INF;
                if (i == 0) { if (lo == 0) diag = 0.0; }
                else if (lo - 1 >= plo && lo - 1 <= phi) diag = WD(prow + lo - 1);
                for (int j = lo; j <= hi; ++j) {
                    const double dt = dtw_cost(xi, yp[yo + j]);
                    double up = INF;
                    if (i > 0 && j >= plo && j <= phi) up = WD(prow + j);
                    double c[3];
                    c[0] = up + dt; c[1] = left + dt; c[2] = diag + dt;
                    int best = o0;
                    double bc = c[o0];
                    if (tie_order == 2) {
                        best = (diag <= up && diag <= left) ? 2 : (up <= left ? 0 : 1);
                        bc = c[best];
                    } else {
                        if (c[o1] < bc) { bc = c[o1]; best = o1; }
                        if (c[o2] < bc) { bc = c[o2]; best = o2; }
                    }
                    WD(crow + j) = bc;
                    acc |= (uint32_t)best << ((cell & 15) * 2);
                    if ((cell & 15) == 15) { WB(cell >> 4) = acc; acc = 0; }
                    ++cell;
                    left = bc;
                    diag = up;                                  // (i-1, j) is the diagonal of (i, j+1)
                }
                if (hi >= lo) { const int64_t t = prow; prow = crow; crow = t; plo = lo; phi = hi; }
            }
            if (cell & 15) WB(cell >> 4) = acc;
            result = WD(prow + (ly - 1));
            if (lev == 0) break;
            // backtrack: column range of the path per row of THIS level
            for (int q = 0; q < lx; ++q) FL(hc, q) = ((-1) << 16) | 0xffff;     // last = -1, first = 65535
            int i = lx - 1, j = ly - 1;
            while (i >= 0 && j >= 0) {
                const int fl = FL(hc, i);
                int first = fl & 0xffff, last = fl >> 16;
                if (last < j) last = j;
                if (first > j) first = j;
                FL(hc, i) = (last << 16) | first;
                const int lohi = LOHI(i);
                const int lo = lohi & 0xffff, hi = lohi >> 16;
                if (j < lo || j > hi) break;                    // cannot happen for a finite path
                const int c = ROWSTART(i) + (j - lo);
                const int d = (WB(c >> 4) >> ((c & 15) * 2)) & 3;
                if (d == 0) --i; else if (d == 1) --j; else { --i; --j; }
            }
        }
        out[r * n_y + a] = (float)(1.0 / (result + 1.0));
    }
#undef WD
#undef WI
#undef WB
#undef ROWSTART
#undef LOHI
#undef FL
}

// ---- register-resident variant for components of at most DTW_R entries ---------------------------
// One wavefront = 64 pairs that share the ANCHOR (y is wave-uniform) and hold 64 consecutive x rows of
// the caller's processing order, one pair per lane.  The DP runs column-major (any topological order
// fills identical cells and makes identical predecessor choices): the row values of the previous
// column live in registers and are updated in place while the column index j walks the anchor series,
// so the DP state never leaves the register file.  The row loop is fully unrolled (static register
// indexing).
//
// Which (row, column) cells a wavefront evaluates is decided with SCALAR control (round 3; rounds 1-2
// tested a 4-row block per lane and swept 554 cells of the finest level where a pair's own window has
// 348 and the union over the 64 lanes 433-483):
//   * fastdtw's window is the same for the fine rows 2p and 2p+1 (both come from coarse row p), so the
//     unit is the ROW PAIR: one window test per pair and column;
//   * per level the 64 lanes' windows are united per row pair (packed 16-bit max over the wavefront with
//     DPP row shifts / broadcasts: hull [min lo, max hi]) and turned into a per-COLUMN range of row pairs
//     [ra_j, rb_j] (lane c works out column c's range; the column loop fetches it with v_readlane);
//   * a column enters the unrolled row chain at pair ra_j through a scalar branch tree and leaves it
//     after pair rb_j: exactly the hull's cells are evaluated, lanes whose own window does not hold
//     the cell get a cost >= 2^1023 (dtw_mask_cost).  A row is evaluated over one contiguous column
//     interval; its register is INF before that interval and stale after it -- the only later reader
//     of a stale register would be the diagonal of the entry row of a later column, which is taken
//     from the register only when the row above the entry pair was evaluated in the previous column
//     (scalar flag), and is INF otherwise.  Rows >= lx of the last pair compute values nobody reads.
// One 32-bit word of 2-bit predecessor codes per column is the only per-cell state written to memory
// (LDS, or the global scratch for long anchor series) on the levels that are backtracked (they have at
// most DTW_R / 2 = 16 rows); the backtrack reads it back and keeps the per-row column range of the
// path in LDS for the next finer level.  The finest level -- more than half of all cells -- is never
// backtracked: it neither tracks nor stores predecessors, and takes one add instead of three.
#define DTW_R 32
#ifndef DTW_OLD_COARSE
#define DTW_OLD_COARSE 0          // 1: the coarse levels of rounds 1-3 (one column array, 2-bit codes at fixed positions) for every instantiation
#endif
#ifndef DTW_NO_ROW_MAJOR
#define DTW_NO_ROW_MAJOR 0        // 1: round 4's coarse levels (a word of predecessor bits per column in LDS, cell-by-cell back-trace) also for levels of <= 32 columns
#endif
#ifndef DTW_COARSE_ONE_SIZE
#define DTW_COARSE_ONE_SIZE 0     // 1: every coarse level runs the RMAX / 2-row instantiation (rounds 4-5)
#endif
#ifndef DTW_MINB12
#define DTW_MINB12 3            // resident 256-thread blocks per CU the 12-row kernel is compiled for
#endif
#ifndef DTW_MINB32
#define DTW_MINB32 2
#endif
#ifndef DTW_MINB20
#define DTW_MINB20 3            // 168 registers: 3 wavefronts per SIMD (5.46 -> 4.65 ms on the benchmark's external side)
#endif

#ifdef DTW_PROBE_COUNT
// Measurement build (tools/dtw_budget.py): what the register kernel executes per level, for the per-level instruction budget.
// [level][0] wavefront-levels run, [1] cells evaluated by the wavefront (the lanes' union: per column 2 x pairs touched),
// [2] cells of the lanes' own windows (summed over lanes), [3] columns swept, [4] (column, row pair) visits, [5] back-trace
// steps (summed over lanes), [6] lanes with a pair at this level
__device__ unsigned long long g_dtw_counts[8][8];
extern "C" int sgnn_dtw_probe_counts(unsigned long long* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dtw_counts), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[64]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_dtw_counts), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
__device__ __forceinline__ unsigned long long dtw_probe_wave_sum(unsigned long long v)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
#define DTW_COUNT(LEV, WHAT, V) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dtw_counts[(LEV) & 7][WHAT], (unsigned long long)(V)); } while (0)
#else
#define DTW_COUNT(LEV, WHAT, V) do { } while (0)
#endif

typedef unsigned short dtw_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dtw_pkmax(uint32_t a, uint32_t b) {          // v_pk_max_u16
    const dtw_us2 m = __builtin_elementwise_max(__builtin_bit_cast(dtw_us2, a), __builtin_bit_cast(dtw_us2, b));
    return __builtin_bit_cast(uint32_t, m);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dtw_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// both 16-bit halves maximised over the 64 lanes (all lanes must be executing); result wave-uniform
__device__ __forceinline__ uint32_t dtw_wave_pkmax(uint32_t v) {
    v = dtw_pkmax(v, dtw_dpp<0xb1, 0xf>(v));            // quad_perm [1,0,3,2]
    v = dtw_pkmax(v, dtw_dpp<0x4e, 0xf>(v));            // quad_perm [2,3,0,1]
    v = dtw_pkmax(v, dtw_dpp<0x124, 0xf>(v));           // row_ror:4
    v = dtw_pkmax(v, dtw_dpp<0x128, 0xf>(v));           // row_ror:8   -> every lane holds its row's maximum
    v = dtw_pkmax(v, dtw_dpp<0x142, 0xa>(v));           // row_bcast:15 into rows 1 and 3
    v = dtw_pkmax(v, dtw_dpp<0x143, 0xc>(v));           // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// One level of the register-resident DP for a wavefront, unrolled over RR (even) rows = RR / 2 row pairs.
// fl: this lane's column of the workgroup's LDS table (stride DTW_THREADS words) holding the coarser
// path's first | last << 16 column per row.  act: the lane has a pair and this level exists for it;
// inactive lanes run along with empty windows.  ly / lyc are wave-uniform (one anchor per wavefront).
// Predecessor codes of a non-finest level go to wl (LDS) when WLDS, else to the global scratch wq.
template <int RR, int TIE, bool WLDS, bool FINEST>
__device__ __forceinline__ double dtw_wave_level(
    int32_t* __restrict__ fl, const double* __restrict__ xcol, const double* __restrict__ xrcol, int64_t n_x,
    const double* __restrict__ ycol, const double* __restrict__ yrcol, bool act,
    int lx, int ly, int lxc, int lyc, bool coarsest, uint32_t* __restrict__ wl, uint32_t* __restrict__ wq, int64_t NT, int lev = 0)
{
    static_assert(RR % 2 == 0 && RR <= 32, "rows come in pairs");
    constexpr int P = RR / 2;
#define FLQ(q) fl[(q) * DTW_THREADS]
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const int32_t EMPTY = 1;                                  // lo = 1, hi = 0
    // ---- this lane's window per row pair (rows 2p and 2p+1 share coarse row p) ----
    int32_t lohi[P];
    if (coarsest) {
#pragma unroll
        for (int p = 0; p < P; ++p) lohi[p] = (act && 2 * p < lx) ? ((ly - 1) << 16) : EMPTY;
    } else {
        int prev_lo = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int ca = p - 1 < 0 ? 0 : p - 1;                            // <= lxc - 1 for every real row
            const int cb = p + 1;                                            // rows past the coarse path's end
            const int firstc = FLQ(ca) & 0xffff;                             // take its last column
            const int lastc = (cb < lxc) ? (FLQ(cb < P ? cb : P - 1) >> 16) : (lyc - 1);
            int lo = 2 * (firstc - 1);
            int hi = 2 * (lastc + 1) + 1;
            if (lo < prev_lo) lo = prev_lo;
            if (lo < 0) lo = 0;
            if (hi > ly - 1) hi = ly - 1;
            int32_t v = (hi << 16) | lo;
            if (hi < lo || 2 * p >= lx || !act) v = EMPTY; else prev_lo = lo;
            lohi[p] = v;
        }
    }
    // ---- hull of the 64 lanes' windows per row pair: (0x7fff - lo) << 16 | (hi + 1), both halves maximised ----
    uint32_t hull[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t lo = lohi[p] & 0xffff, hi = (uint32_t)lohi[p] >> 16;
        hull[p] = dtw_wave_pkmax(lohi[p] == EMPTY ? 0u : (((0x7fffu - lo) << 16) | (hi + 1)));
    }
#ifdef DTW_PROBE_COUNT
    {
        unsigned long long own = 0;
        for (int p = 0; p < P; ++p)
            if (lohi[p] != EMPTY) own += (unsigned long long)(((uint32_t)lohi[p] >> 16) - (lohi[p] & 0xffff) + 1) * (2 * p + 1 < lx ? 2 : 1);
        own = dtw_probe_wave_sum(own);
        const unsigned long long lanes = dtw_probe_wave_sum(act ? 1ull : 0ull);
        DTW_COUNT(lev, 0, 1); DTW_COUNT(lev, 2, own); DTW_COUNT(lev, 6, lanes);
    }
#endif
    double xp1[RR], xr[RR], col[RR];
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int64_t ic = i < lx ? i : 0;                                   // rows past the series repeat row 0: nobody reads them,
        xp1[i] = xcol[ic * n_x] + 1.0;                                       // and all loads of the level are in flight together
        xr[i] = xrcol[ic * n_x];
        col[i] = INF;
    }
    const int lane = threadIdx.x & 63;
    uint32_t prev_am = 0;                                                    // the previous column's row-pair mask
    for (int jc = 0; jc < ly; jc += 64) {
        // lane c: the row pairs column jc + c touches in the hull, as a bit mask (pairs [ra, rb1)) | ra << 16
        uint32_t tab;
        {
            const uint32_t c = (uint32_t)(jc + lane);
            uint32_t ra = P, rb1 = 0;
#pragma unroll
            for (int p = P - 1; p >= 0; --p) ra = ((hull[p] & 0xffffu) > c) ? (uint32_t)p : ra;              // hi >= c
#pragma unroll
            for (int p = 0; p < P; ++p) rb1 = (hull[p] != 0u && (0x7fffu - (hull[p] >> 16)) <= c) ? (uint32_t)(p + 1) : rb1;
            tab = rb1 > ra ? ((((1u << rb1) - 1u) & ~((1u << ra) - 1u)) | (ra << 16)) : 0u;
        }
        const int jend = jc + 64 < ly ? jc + 64 : ly;
        double y_next = ycol[jc], yr_next = yrcol[jc];
        for (int j = jc; j < jend; ++j) {
            const double yp1 = y_next + 1.0, yr = yr_next;
            if (j + 1 < ly) { y_next = ycol[j + 1]; yr_next = yrcol[j + 1]; }   // in flight during this column
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tab, j - jc);
            const uint32_t am = t & 0xffffu;                                 // 0: no lane holds this column (cannot happen for real series)
            const int ra = (int)(t >> 16);
            DTW_COUNT(lev, 1, 2 * __builtin_popcount(am)); DTW_COUNT(lev, 3, 1); DTW_COUNT(lev, 4, __builtin_popcount(am));
            // the row above the entry pair was evaluated in the previous column: its register is D[2 ra - 1][j - 1]
            const bool carry = ra > 0 && ((prev_am >> (ra - 1)) & 1u);
            prev_am = am;
            uint32_t word = 0;
            // The row chain, written so that no value has to be copied between rows or at the chain's entry points:
            //  * a cell's result goes INTO the row's register, and everything later cells need of the OLD value is
            //    formed first -- on the finest level pm = min(D[i][j-1], D[i+1][j-1]) (the next row's left and diagonal
            //    predecessors; min is exact, its grouping does not matter), elsewhere cd = D[i][j-1] + cost(i+1, j) (the
            //    next row's diagonal candidate inside a pair; the tie rules need the three sums apart) and one copy dg of
            //    the pair's last old value for the next pair;
            //  * "up" is the register of the row above; the entry pair, whose row above is not part of this column, runs
            //    its own copy of the pair's first cell with up = INF and joins the chain behind it.
            double pm = INF, cd = INF, dg = INF, da = INF;
            bool in_ = false;
            double dt1_ = 0.0;
#define DTW_IN(K) { const int lo_ = lohi[(K) < P ? (K) : 0] & 0xffff, hi_ = lohi[(K) < P ? (K) : 0] >> 16; in_ = j >= lo_ && j <= hi_; }
#define DTW_COST(I) dtw_mask_cost(in_, dtw_cost_rcp(xp1[(I) < RR ? (I) : 0], xr[(I) < RR ? (I) : 0], yp1, yr))
            // first cell of pair K (row 2K); UPV = the value above it
#define DTW_CELL_A(K, UPV)                                                                                           \
            {                                                                                                        \
                constexpr int a_ = 2 * (K) < RR ? 2 * (K) : 0, b_ = 2 * (K) + 1 < RR ? 2 * (K) + 1 : 0;              \
                const double dt0_ = DTW_COST(2 * (K));                                                               \
                dt1_ = DTW_COST(2 * (K) + 1);                                                                        \
                if constexpr (FINEST) {                                                                              \
                    /* only the value is needed: rounding is monotone, so the smallest of the three rounded sums */  \
                    /* is the rounded sum of the smallest candidate -- one add instead of three */                    \
                    const double m_ = fmin(pm, (UPV));                                                               \
                    pm = fmin(col[a_], col[b_]);                                                                     \
                    col[a_] = m_ + dt0_;                                                                             \
                } else {                                                                                             \
                    const double c_up = (UPV) + dt0_, c_left = col[a_] + dt0_, c_diag = dg + dt0_;                   \
                    const double m1_ = fmin((UPV), col[a_]);                                                         \
                    const double mv = TIE == 2 ? fmin(m1_, dg) + dt0_                                                \
                                               : fmin(fmin(c_up, c_left), c_diag);   /* two v_min_f64 (no NaNs here) */ \
                    /* predecessor = the first candidate, in the tie order, that attains the minimum; a cell */      \
                    /* outside the lane's window is never read back, whatever code it gets */                        \
                    uint32_t best;                                                                                   \
                    if (TIE == 0) best = c_up == mv ? 0u : (c_left == mv ? 1u : 2u);        /* (i-1,j), (i,j-1), (i-1,j-1) */ \
                    else if (TIE == 1) best = c_diag == mv ? 2u : (c_up == mv ? 0u : 1u);   /* (i-1,j-1), (i-1,j), (i,j-1) */ \
                    else best = dg <= m1_ ? 2u : ((UPV) <= col[a_] ? 0u : 1u);           /* on the predecessor costs, <= */ \
                    word |= best << (2 * a_);                                                                        \
                    if (TIE == 2) da = col[a_]; else cd = col[a_] + dt1_;    /* (i, j-1) is the diagonal of (i+1, j) */ \
                    col[a_] = mv;                                                                                    \
                }                                                                                                    \
            }
            // second cell of pair K (row 2K+1): the row above is the pair's first
#define DTW_CELL_B(K)                                                                                                \
            {                                                                                                        \
                constexpr int a_ = 2 * (K) < RR ? 2 * (K) : 0, b_ = 2 * (K) + 1 < RR ? 2 * (K) + 1 : 0;              \
                constexpr int c_ = 2 * (K) + 2 < RR ? 2 * (K) + 2 : b_;                                              \
                if constexpr (FINEST) {                                                                              \
                    const double m_ = fmin(pm, col[a_]);                                                             \
                    pm = fmin(col[b_], col[c_]);                                                                     \
                    col[b_] = m_ + dt1_;                                                                             \
                } else {                                                                                             \
                    const double c_up = col[a_] + dt1_, c_left = col[b_] + dt1_, c_diag = TIE == 2 ? da + dt1_ : cd; \
                    const double m1_ = fmin(col[a_], col[b_]);                                                       \
                    const double mv = TIE == 2 ? fmin(m1_, da) + dt1_ : fmin(fmin(c_up, c_left), c_diag);            \
                    uint32_t best;                                                                                   \
                    if (TIE == 0) best = c_up == mv ? 0u : (c_left == mv ? 1u : 2u);                                 \
                    else if (TIE == 1) best = c_diag == mv ? 2u : (c_up == mv ? 0u : 1u);                            \
                    else best = da <= m1_ ? 2u : (col[a_] <= col[b_] ? 0u : 1u);                                     \
                    word |= best << (2 * b_);                                                                        \
                    dg = col[b_];                                            /* the next pair's diagonal */           \
                    col[b_] = mv;                                                                                    \
                }                                                                                                    \
            }
            // one structured, wave-uniform region per row pair (scalar bit test), skipped unless the column touches the
            // pair.  At the entry pair the row above is not part of this column (nor of any later one: the ranges only
            // move down): its register hands over the diagonal if it was evaluated in the previous column and is then
            // set to INF for good, which makes it the "up" the first cell needs -- the chain itself has no special case.
#define DTW_PAIR(K)                                                                                                  \
            if ((K) < P && (am & (1u << (K)))) {                                                                     \
                constexpr int u_ = (2 * (K) - 1) >= 0 && (2 * (K) - 1) < RR ? (2 * (K) - 1) : 0;                     \
                DTW_IN(K)                                                                                            \
                if ((K) == 0) {                                                                                      \
                    const double diag = (j == 0) ? 0.0 : INF;                /* virtual origin D[-1][-1] = 0 */       \
                    if constexpr (FINEST) pm = fmin(diag, col[0]); else dg = diag;                                   \
                    DTW_CELL_A(K, INF)                                                                               \
                } else {                                                                                             \
                    if (ra == (K)) {                                                                                 \
                        const double diag = carry ? col[u_] : INF;                                                   \
                        if constexpr (FINEST) pm = fmin(diag, col[2 * (K) < RR ? 2 * (K) : 0]); else dg = diag;      \
                        col[u_] = INF;                                                                               \
                        asm volatile("" ::: "memory");                       /* keeps this a branch: as selects it costs every pair 9 instructions */ \
                    }                                                                                                \
                    DTW_CELL_A(K, col[u_])                                                                           \
                }                                                                                                    \
                DTW_CELL_B(K)                                                                                        \
            }
            DTW_PAIR(0) DTW_PAIR(1) DTW_PAIR(2) DTW_PAIR(3) DTW_PAIR(4) DTW_PAIR(5) DTW_PAIR(6) DTW_PAIR(7)
            DTW_PAIR(8) DTW_PAIR(9) DTW_PAIR(10) DTW_PAIR(11) DTW_PAIR(12) DTW_PAIR(13) DTW_PAIR(14) DTW_PAIR(15)
#undef DTW_IN
#undef DTW_CELL_A
#undef DTW_CELL_B
#undef DTW_COST
#undef DTW_PAIR
            if (!FINEST) {                                                   // the finest level is never backtracked
                if (WLDS) wl[j * DTW_THREADS] = word; else wq[(int64_t)j * NT] = word;
            }
        }
    }
    double result = 0.0;
#pragma unroll
    for (int i = 0; i < RR; ++i) if (i == lx - 1) result = col[i];
    if (FINEST || !act) return result;
#ifdef DTW_PROBE_NO_BACKTRACK
    return result;
#endif
    // backtrack through the predecessor codes; record the path's column range per row.  The path enters a row at its
    // LAST column and leaves it at its first, and it visits every row: one LDS write per row when the path leaves it (no
    // initialisation, no read-modify-write per step), and the column's word of codes is fetched when the column changes
    int i = lx - 1, j = ly - 1;
    int last = j;
    uint32_t word = WLDS ? wl[j * DTW_THREADS] : wq[(int64_t)j * NT];
#ifdef DTW_PROBE_COUNT
    unsigned long long bt_steps = 0;
#endif
    while (i >= 0 && j >= 0) {
#ifdef DTW_PROBE_COUNT
        ++bt_steps;
#endif
        const int d = (int)((word >> (2 * i)) & 3);
        if (d != 1) {                                                        // the path leaves row i here, at column j
            FLQ(i) = (last << 16) | j;
            --i;
        }
        if (d != 0) {
            --j;
            if (j >= 0) word = WLDS ? wl[j * DTW_THREADS] : wq[(int64_t)j * NT];
        }
        if (d != 1) last = j;                                                // ... and enters the row above at column j (or j - 1)
    }
    // (a path that runs off the first column inside a row -- only possible through a window's edge -- leaves that row open)
    if (i >= 0 && j < 0) FLQ(i) = (last << 16) | 0;
#ifdef DTW_PROBE_COUNT
    atomicAdd(&g_dtw_counts[lev & 7][5], bt_steps);
#endif
    return result;
#undef FLQ
}

// ---- coarse levels, round 4: two column arrays + predecessor bits shifted in through the carry ----------------------------
// The levels that are backtracked paid ~23 vector instructions per cell where the finest level pays 12 (ISA count): per row pair
// three v_mov_b64 (a cell's old value had to be copied out of the row's register before the new one went in: it is the next
// row's diagonal) and per cell two v_cndmask + an OR (+ v_mov of the shifted constants) to turn two compare masks into a 2-bit
// code at the row's fixed position of the column's word.  Here
//   * the DP keeps TWO column arrays, read (column j - 1) and written (column j) alternately -- a cell reads up = cur[i - 1],
//     left = prev[i], diagonal = prev[i - 1] and writes cur[i]: no copies; the column loop is unrolled by two so that the
//     arrays keep static registers.  A row is evaluated over one contiguous column interval and both of its registers are +inf
//     before it, so prev[] of a row that joins the sweep in this column is +inf by itself; the entry pair's row above (not
//     part of this column, nor of any later one) is made +inf in cur[] and -- unless it was evaluated in the previous
//     column -- in prev[];
//   * the two compare masks of a cell (wave-wide lane masks in scalar registers) are shifted into the lane's word through the
//     carry input of v_addc_co_u32 (word = 2 word + bit): two instructions per cell.  The bits of the rows a column evaluated
//     follow each other in evaluation order, the last row in the lowest two bits; the column's last row number (wave-uniform)
//     sits in bits 27..31, so the back-trace finds row i's bits at 2 (last_row - i).  Needs 2 RR <= 27: the levels of the
//     12- and 20-row instantiations (6 / 10 rows); the 32-row one keeps the fixed-position words.
// Bits per tie rule (first, second): 0: (c_up == min, c_left == min); 1: (c_diag == min, c_up == min); 2: (diag <= up && diag
// <= left, up <= left) on the predecessor costs -- decoded only along the path.
__device__ __forceinline__ uint32_t dtw_shift_in(uint32_t word, uint64_t lane_mask)
{
#if defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__)
    // (VOP3 v_addc_co_u32 with a 64-bit scalar carry-in: the gfx9 family's wave64 encoding)
    uint32_t out;
    asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(out) : "v"(word), "s"(lane_mask) : "vcc");
    return out;
#elif defined(__HIP_DEVICE_COMPILE__)
#error "dtw_shift_in: hand-written gfx9 wave64 instruction -- build with -DDTW_OLD_COARSE=1 for another target"
#else
    return 2u * word + (uint32_t)(lane_mask & 1u);              // host pass of the single-source compile: never executed
#endif
}

template <int RR, int TIE, bool WLDS>
__device__ __forceinline__ void dtw_wave_level_pp(
    int32_t* __restrict__ fl, const double* __restrict__ xcol, const double* __restrict__ xrcol, int64_t n_x,
    const double* __restrict__ ycol, const double* __restrict__ yrcol, bool act,
    int lx, int ly, int lxc, int lyc, bool coarsest, uint32_t* __restrict__ wl, uint32_t* __restrict__ wq, int64_t NT, int lev = 0)
{
    static_assert(RR % 2 == 0 && 2 * RR <= 27, "two bits per row below the row number");
    constexpr int P = RR / 2;
#define FLQ(q) fl[(q) * DTW_THREADS]
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const int32_t EMPTY = 1;
    int32_t lohi[P];
    if (coarsest) {
#pragma unroll
        for (int p = 0; p < P; ++p) lohi[p] = (act && 2 * p < lx) ? ((ly - 1) << 16) : EMPTY;
    } else {
        int prev_lo = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int ca = p - 1 < 0 ? 0 : p - 1;
            const int cb = p + 1;
            const int firstc = FLQ(ca) & 0xffff;
            const int lastc = (cb < lxc) ? (FLQ(cb < P ? cb : P - 1) >> 16) : (lyc - 1);
            int lo = 2 * (firstc - 1);
            int hi = 2 * (lastc + 1) + 1;
            if (lo < prev_lo) lo = prev_lo;
            if (lo < 0) lo = 0;
            if (hi > ly - 1) hi = ly - 1;
            int32_t v = (hi << 16) | lo;
            if (hi < lo || 2 * p >= lx || !act) v = EMPTY; else prev_lo = lo;
            lohi[p] = v;
        }
    }
    uint32_t hull[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t lo = lohi[p] & 0xffff, hi = (uint32_t)lohi[p] >> 16;
        hull[p] = dtw_wave_pkmax(lohi[p] == EMPTY ? 0u : (((0x7fffu - lo) << 16) | (hi + 1)));
    }
#ifdef DTW_PROBE_COUNT
    {
        unsigned long long own = 0;
        for (int p = 0; p < P; ++p)
            if (lohi[p] != EMPTY) own += (unsigned long long)(((uint32_t)lohi[p] >> 16) - (lohi[p] & 0xffff) + 1) * (2 * p + 1 < lx ? 2 : 1);
        own = dtw_probe_wave_sum(own);
        const unsigned long long lanes = dtw_probe_wave_sum(act ? 1ull : 0ull);
        DTW_COUNT(lev, 0, 1); DTW_COUNT(lev, 2, own); DTW_COUNT(lev, 6, lanes);
    }
#endif
    double xp1[RR], xr[RR], cA[RR], cB[RR];
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int64_t ic = i < lx ? i : 0;
        xp1[i] = xcol[ic * n_x] + 1.0;
        xr[i] = xrcol[ic * n_x];
        cA[i] = INF;
        cB[i] = INF;
    }
    const int lane = threadIdx.x & 63;
    uint32_t prev_am = 0;
    for (int jc = 0; jc < ly; jc += 64) {
        uint32_t tab;
        {
            const uint32_t c = (uint32_t)(jc + lane);
            uint32_t ra = P, rb1 = 0;
#pragma unroll
            for (int p = P - 1; p >= 0; --p) ra = ((hull[p] & 0xffffu) > c) ? (uint32_t)p : ra;
#pragma unroll
            for (int p = 0; p < P; ++p) rb1 = (hull[p] != 0u && (0x7fffu - (hull[p] >> 16)) <= c) ? (uint32_t)(p + 1) : rb1;
            // pairs [ra, rb1) as a mask | ra << 16 | the column's last row 2 rb1 - 1, already at its place in the word (bits 27..31)
            tab = rb1 > ra ? ((((1u << rb1) - 1u) & ~((1u << ra) - 1u)) | (ra << 16) | ((2u * rb1 - 1u) << 27)) : 0u;
        }
        const int jend = jc + 64 < ly ? jc + 64 : ly;
        double y_next = ycol[jc], yr_next = yrcol[jc];
#define DTWP_IN(K) { const int lo_ = lohi[(K) < P ? (K) : 0] & 0xffff, hi_ = lohi[(K) < P ? (K) : 0] >> 16; in_ = J_ >= lo_ && J_ <= hi_; }
#define DTWP_COST(I) dtw_mask_cost(in_, dtw_cost_rcp(xp1[(I) < RR ? (I) : 0], xr[(I) < RR ? (I) : 0], yp1, yr))
        // one cell: UPV / LEFT / DIAG are the three predecessors' values, OUT the register the cell's value goes into
#define DTWP_CELL(UPV, LEFT, DIAG, DT, OUT)                                                                          \
            {                                                                                                        \
                const double c_up = (UPV) + (DT), c_left = (LEFT) + (DT), c_diag = (DIAG) + (DT);                    \
                /* rule 2 picks on the predecessors' values: only the cell's value is needed of the sums, and rounding */ \
                /* is monotone -- the smallest rounded sum is the rounded sum of the smallest predecessor: 1 add, not 3 */ \
                /* (and "diagonal <= up and diagonal <= left" is ONE compare against min(up, left), which the value needs anyway) */ \
                const double m1_ = fmin((UPV), (LEFT));                                                              \
                const double mv = TIE == 2 ? fmin(m1_, (DIAG)) + (DT) : fmin(fmin(c_up, c_left), c_diag);            \
                if (TIE == 0) { word = dtw_shift_in(word, __ballot(c_up == mv)); word = dtw_shift_in(word, __ballot(c_left == mv)); } \
                else if (TIE == 1) { word = dtw_shift_in(word, __ballot(c_diag == mv)); word = dtw_shift_in(word, __ballot(c_up == mv)); } \
                else { word = dtw_shift_in(word, __ballot((DIAG) <= m1_));                                           \
                       word = dtw_shift_in(word, __ballot((UPV) <= (LEFT))); }                                      \
                (OUT) = mv;                                                                                          \
            }
#define DTWP_PAIR(K, PREV, CUR)                                                                                      \
            if ((K) < P && (am & (1u << (K)))) {                                                                     \
                constexpr int a_ = 2 * (K) < RR ? 2 * (K) : 0, b_ = 2 * (K) + 1 < RR ? 2 * (K) + 1 : 0;              \
                constexpr int u_ = (2 * (K) - 1) >= 0 && (2 * (K) - 1) < RR ? (2 * (K) - 1) : 0;                     \
                bool in_;                                                                                            \
                DTWP_IN(K)                                                                                           \
                const double dt0_ = DTWP_COST(2 * (K));                                                              \
                const double dt1_ = DTWP_COST(2 * (K) + 1);                                                          \
                if ((K) == 0) {                                                                                      \
                    const double diag0_ = (J_ == 0) ? 0.0 : INF;             /* virtual origin D[-1][-1] = 0 */       \
                    DTWP_CELL(INF, PREV[a_], diag0_, dt0_, CUR[a_])                                                  \
                } else {                                                                                             \
                    if (ra == (K)) {                                         /* the row above is not part of this column */ \
                        CUR[u_] = INF;                                                                               \
                        if (!carry) PREV[u_] = INF;                                                                  \
                        asm volatile("" ::: "memory");                       /* keeps this a branch */               \
                    }                                                                                                \
                    DTWP_CELL(CUR[u_], PREV[a_], PREV[u_], dt0_, CUR[a_])                                            \
                }                                                                                                    \
                DTWP_CELL(CUR[a_], PREV[b_], PREV[a_], dt1_, CUR[b_])                                                \
            }
#define DTWP_COLUMN(JJ, PREV, CUR)                                                                                   \
            {                                                                                                        \
                const int J_ = (JJ);                                                                                 \
                const double yp1 = y_next + 1.0, yr = yr_next;                                                       \
                if (J_ + 1 < ly) { y_next = ycol[J_ + 1]; yr_next = yrcol[J_ + 1]; }                                 \
                const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tab, J_ - jc);                           \
                const uint32_t am = t & 0xffffu;                                                                     \
                const int ra = (int)((t >> 16) & 0x7ffu);                                                            \
                DTW_COUNT(lev, 1, 2 * __builtin_popcount(am)); DTW_COUNT(lev, 3, 1); DTW_COUNT(lev, 4, __builtin_popcount(am)); \
                const bool carry = ra > 0 && ((prev_am >> (ra - 1)) & 1u);                                           \
                prev_am = am;                                                                                        \
                uint32_t word = 0;                                                                                   \
                DTWP_PAIR(0, PREV, CUR) DTWP_PAIR(1, PREV, CUR) DTWP_PAIR(2, PREV, CUR) DTWP_PAIR(3, PREV, CUR)      \
                DTWP_PAIR(4, PREV, CUR) DTWP_PAIR(5, PREV, CUR)                                                      \
                word |= t & 0xf8000000u;                                     /* the column's last row */             \
                if (WLDS) wl[J_ * DTW_THREADS] = word; else wq[(int64_t)J_ * NT] = word;                             \
            }
        int j = jc;
        for (; j + 1 < jend; j += 2) {
            DTWP_COLUMN(j, cB, cA)
            DTWP_COLUMN(j + 1, cA, cB)
        }
        if (j < jend) DTWP_COLUMN(j, cB, cA)
#undef DTWP_IN
#undef DTWP_COST
#undef DTWP_CELL
#undef DTWP_PAIR
#undef DTWP_COLUMN
    }
    if (!act) return;
#ifdef DTW_PROBE_NO_BACKTRACK
    return;
#endif
    int i = lx - 1, j = ly - 1;
    int last = j;
    uint32_t word = WLDS ? wl[j * DTW_THREADS] : wq[(int64_t)j * NT];
#ifdef DTW_PROBE_COUNT
    unsigned long long bt_steps = 0;
#endif
    while (i >= 0 && j >= 0) {
#ifdef DTW_PROBE_COUNT
        ++bt_steps;
#endif
        const int sh = 2 * ((int)(word >> 27) - i);
        const uint32_t b = (word >> sh) & 3u;                                // first bit << 1 | second bit
        int d;
        if (TIE == 0) d = (b & 2u) ? 0 : ((b & 1u) ? 1 : 2);
        else d = (b & 2u) ? 2 : ((b & 1u) ? 0 : 1);
        if (d != 1) {
            FLQ(i) = (last << 16) | j;
            --i;
        }
        if (d != 0) {
            --j;
            if (j >= 0) word = WLDS ? wl[j * DTW_THREADS] : wq[(int64_t)j * NT];
        }
        if (d != 1) last = j;
    }
    if (i >= 0 && j < 0) FLQ(i) = (last << 16) | 0;
#ifdef DTW_PROBE_COUNT
    atomicAdd(&g_dtw_counts[lev & 7][5], bt_steps);
#endif
#undef FLQ
}

// ---- coarse levels, round 5: predecessor bits kept ROW-major in registers, the back-trace one step per ROW ------------------
// Round 4's levels wrote one word of predecessor bits per column to LDS and walked the path back cell by cell: ~45 dependent
// steps per pair over the three coarse levels of the benchmark (26 + 13 + 6), each a chain of ~20 vector instructions around an
// LDS read whose address depends on the step before -- a per-lane serial loop inside a kernel whose every other part runs in
// lockstep.  Here
//   * each ROW keeps two 32-bit words (the two compare masks of its cells, one bit per evaluated column, shifted in through the
//     carry exactly as before -- still two instructions per cell, no per-column word, no LDS store).  A row is evaluated over
//     one contiguous column interval -- its pair's hull [ulo, uhi], wave-uniform -- so column j's bit sits at position uhi - j.
//     Needs a level of at most 32 columns (anchor series of up to 65 entries); longer ones keep dtw_wave_level_pp;
//   * the warp path visits every row, from the last to the first, and inside a row it can only move LEFT: the cells it crosses
//     in row i are a run of "left" codes that starts at the column it entered the row.  With the row's bits in a register that
//     run is a count of trailing ones -- shift, complement, find-first-set -- and the code at the run's end says whether the
//     path goes up or diagonally.  The back-trace is a loop over ROWS, unrolled (all lanes are in the same row at the same
//     time; only the column differs), ~12 vector instructions per row and no memory access but the row's (first, last) write.
// Same predecessor rule, same path, same windows for the finer level: bit-identical to the cell-by-cell walk.
template <int RR, int TIE>
__device__ __forceinline__ void dtw_wave_level_rm(
    int32_t* __restrict__ fl, const double* __restrict__ xcol, const double* __restrict__ xrcol, int64_t n_x,
    const double* __restrict__ ycol, const double* __restrict__ yrcol, bool act,
    int lx, int ly, int lxc, int lyc, bool coarsest, int lev = 0)
{
    static_assert(RR % 2 == 0 && RR <= 16, "rows come in pairs; two words of bits per row");
    constexpr int P = RR / 2;
#define FLQ(q) fl[(q) * DTW_THREADS]
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const int32_t EMPTY = 1;
    int32_t lohi[P];
    if (coarsest) {
#pragma unroll
        for (int p = 0; p < P; ++p) lohi[p] = (act && 2 * p < lx) ? ((ly - 1) << 16) : EMPTY;
    } else {
        int prev_lo = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int ca = p - 1 < 0 ? 0 : p - 1;
            const int cb = p + 1;
            const int firstc = FLQ(ca) & 0xffff;
            const int lastc = (cb < lxc) ? (FLQ(cb < P ? cb : P - 1) >> 16) : (lyc - 1);
            int lo = 2 * (firstc - 1);
            int hi = 2 * (lastc + 1) + 1;
            if (lo < prev_lo) lo = prev_lo;
            if (lo < 0) lo = 0;
            if (hi > ly - 1) hi = ly - 1;
            int32_t v = (hi << 16) | lo;
            if (hi < lo || 2 * p >= lx || !act) v = EMPTY; else prev_lo = lo;
            lohi[p] = v;
        }
    }
    uint32_t hull[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t lo = lohi[p] & 0xffff, hi = (uint32_t)lohi[p] >> 16;
        hull[p] = dtw_wave_pkmax(lohi[p] == EMPTY ? 0u : (((0x7fffu - lo) << 16) | (hi + 1)));
    }
#ifdef DTW_PROBE_COUNT
    {
        unsigned long long own = 0;
        for (int p = 0; p < P; ++p)
            if (lohi[p] != EMPTY) own += (unsigned long long)(((uint32_t)lohi[p] >> 16) - (lohi[p] & 0xffff) + 1) * (2 * p + 1 < lx ? 2 : 1);
        own = dtw_probe_wave_sum(own);
        const unsigned long long lanes = dtw_probe_wave_sum(act ? 1ull : 0ull);
        DTW_COUNT(lev, 0, 1); DTW_COUNT(lev, 2, own); DTW_COUNT(lev, 6, lanes);
    }
#endif
    double xp1[RR], xr[RR], cA[RR], cB[RR];
    uint32_t mA[RR], mB[RR];                                                 // the rows' bits: first / second compare mask of every cell
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int64_t ic = i < lx ? i : 0;
        xp1[i] = xcol[ic * n_x] + 1.0;
        xr[i] = xrcol[ic * n_x];
        cA[i] = INF;
        cB[i] = INF;
        mA[i] = 0u;
        mB[i] = 0u;
    }
    const int lane = threadIdx.x & 63;
    uint32_t prev_am = 0;
    {                                                                        // (at most 32 columns: one table)
        uint32_t tab;
        {
            const uint32_t c = (uint32_t)lane;
            uint32_t ra = P, rb1 = 0;
#pragma unroll
            for (int p = P - 1; p >= 0; --p) ra = ((hull[p] & 0xffffu) > c) ? (uint32_t)p : ra;
#pragma unroll
            for (int p = 0; p < P; ++p) rb1 = (hull[p] != 0u && (0x7fffu - (hull[p] >> 16)) <= c) ? (uint32_t)(p + 1) : rb1;
            tab = rb1 > ra ? ((((1u << rb1) - 1u) & ~((1u << ra) - 1u)) | (ra << 16)) : 0u;
        }
        double y_next = ycol[0], yr_next = yrcol[0];
#define DTWR_IN(K) { const int lo_ = lohi[(K) < P ? (K) : 0] & 0xffff, hi_ = lohi[(K) < P ? (K) : 0] >> 16; in_ = J_ >= lo_ && J_ <= hi_; }
#define DTWR_COST(I) dtw_mask_cost(in_, dtw_cost_rcp(xp1[(I) < RR ? (I) : 0], xr[(I) < RR ? (I) : 0], yp1, yr))
        // one cell of row ROW: UPV / LEFT / DIAG are the three predecessors' values, OUT the register the cell's value goes into
#define DTWR_CELL(UPV, LEFT, DIAG, DT, OUT, ROW)                                                                     \
            {                                                                                                        \
                const double c_up = (UPV) + (DT), c_left = (LEFT) + (DT), c_diag = (DIAG) + (DT);                    \
                /* rule 2 picks on the predecessors' values: only the cell's value is needed of the sums, and rounding */ \
                /* is monotone -- the smallest rounded sum is the rounded sum of the smallest predecessor: 1 add, not 3 */ \
                /* (and "diagonal <= up and diagonal <= left" is ONE compare against min(up, left), which the value needs anyway) */ \
                const double m1_ = fmin((UPV), (LEFT));                                                              \
                const double mv = TIE == 2 ? fmin(m1_, (DIAG)) + (DT) : fmin(fmin(c_up, c_left), c_diag);            \
                if (TIE == 0) { mA[ROW] = dtw_shift_in(mA[ROW], __ballot(c_up == mv)); mB[ROW] = dtw_shift_in(mB[ROW], __ballot(c_left == mv)); } \
                else if (TIE == 1) { mA[ROW] = dtw_shift_in(mA[ROW], __ballot(c_diag == mv)); mB[ROW] = dtw_shift_in(mB[ROW], __ballot(c_up == mv)); } \
                else { mA[ROW] = dtw_shift_in(mA[ROW], __ballot((DIAG) <= m1_));                                     \
                       mB[ROW] = dtw_shift_in(mB[ROW], __ballot((UPV) <= (LEFT))); }                                 \
                (OUT) = mv;                                                                                          \
            }
#define DTWR_PAIR(K, PREV, CUR)                                                                                      \
            if ((K) < P && (am & (1u << (K)))) {                                                                     \
                constexpr int a_ = 2 * (K) < RR ? 2 * (K) : 0, b_ = 2 * (K) + 1 < RR ? 2 * (K) + 1 : 0;              \
                constexpr int u_ = (2 * (K) - 1) >= 0 && (2 * (K) - 1) < RR ? (2 * (K) - 1) : 0;                     \
                bool in_;                                                                                            \
                DTWR_IN(K)                                                                                           \
                const double dt0_ = DTWR_COST(2 * (K));                                                              \
                const double dt1_ = DTWR_COST(2 * (K) + 1);                                                          \
                if ((K) == 0) {                                                                                      \
                    const double diag0_ = (J_ == 0) ? 0.0 : INF;             /* virtual origin D[-1][-1] = 0 */       \
                    DTWR_CELL(INF, PREV[a_], diag0_, dt0_, CUR[a_], a_)                                              \
                } else {                                                                                             \
                    if (ra == (K)) {                                         /* the row above is not part of this column */ \
                        CUR[u_] = INF;                                                                               \
                        if (!carry) PREV[u_] = INF;                                                                  \
                        asm volatile("" ::: "memory");                       /* keeps this a branch */               \
                    }                                                                                                \
                    DTWR_CELL(CUR[u_], PREV[a_], PREV[u_], dt0_, CUR[a_], a_)                                        \
                }                                                                                                    \
                DTWR_CELL(CUR[a_], PREV[b_], PREV[a_], dt1_, CUR[b_], b_)                                            \
            }
#define DTWR_COLUMN(JJ, PREV, CUR)                                                                                   \
            {                                                                                                        \
                const int J_ = (JJ);                                                                                 \
                const double yp1 = y_next + 1.0, yr = yr_next;                                                       \
                if (J_ + 1 < ly) { y_next = ycol[J_ + 1]; yr_next = yrcol[J_ + 1]; }                                 \
                const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tab, J_);                                \
                const uint32_t am = t & 0xffffu;                                                                     \
                const int ra = (int)((t >> 16) & 0x7ffu);                                                            \
                DTW_COUNT(lev, 1, 2 * __builtin_popcount(am)); DTW_COUNT(lev, 3, 1); DTW_COUNT(lev, 4, __builtin_popcount(am)); \
                const bool carry = ra > 0 && ((prev_am >> (ra - 1)) & 1u);                                           \
                prev_am = am;                                                                                        \
                DTWR_PAIR(0, PREV, CUR) DTWR_PAIR(1, PREV, CUR) DTWR_PAIR(2, PREV, CUR) DTWR_PAIR(3, PREV, CUR)      \
                DTWR_PAIR(4, PREV, CUR) DTWR_PAIR(5, PREV, CUR) DTWR_PAIR(6, PREV, CUR) DTWR_PAIR(7, PREV, CUR)      \
            }
        int j = 0;
        for (; j + 1 < ly; j += 2) {
            DTWR_COLUMN(j, cB, cA)
            DTWR_COLUMN(j + 1, cA, cB)
        }
        if (j < ly) DTWR_COLUMN(j, cB, cA)
#undef DTWR_IN
#undef DTWR_COST
#undef DTWR_CELL
#undef DTWR_PAIR
#undef DTWR_COLUMN
    }
#ifdef DTW_PROBE_NO_BACKTRACK
    return;
#endif
    // ---- back-trace, one step per row ------------------------------------------------------------------------------------
    int j = act ? ly - 1 : -1;                                               // the column the path enters the current row at
#pragma unroll
    for (int i = RR - 1; i >= 0; --i) {
        const uint32_t h = hull[i >> 1];
        const int uhi = (int)(h & 0xffffu) - 1, ulo = 0x7fff - (int)(h >> 16);       // the row's evaluated columns (wave-uniform)
        if (i < lx && j >= 0) {
            const int pos = uhi - j;                                         // column j's bit
            const uint32_t valid = (uhi - ulo + 1) >= 32 ? ~0u : ((1u << (uhi - ulo + 1)) - 1u);
            const uint32_t A = mA[i], B = mB[i];
            // the cells whose predecessor is (i, j - 1): rule 0: not up, left; rules 1 / 2: neither first nor second
            const uint32_t is_left = (TIE == 0 ? (~A & B) : (~A & ~B)) & valid;
            const uint32_t tl = (pos >= 0 && pos < 32) ? (is_left >> pos) : 0u;
            int run = __ffs((int)~tl) - 1;                                   // trailing ones (tl has a zero: bit 31 - pos at the latest ... see valid)
            if (run < 0) run = 32;
            if (run > j) run = j;                                            // (a run into column 0 ends there)
            const int first = j - run;
            FLQ(i) = (j << 16) | first;
            const int p2 = pos + run;
            const uint32_t a2 = (p2 >= 0 && p2 < 32) ? ((A >> p2) & 1u) : 0u, b2 = (p2 >= 0 && p2 < 32) ? ((B >> p2) & 1u) : 0u;
            // at the run's end: up, diagonal -- or still left, when the run was cut at column 0: the path ends in this row
            const bool up = TIE == 0 ? (a2 != 0u) : (a2 == 0u && b2 != 0u);
            const bool left_still = TIE == 0 ? (a2 == 0u && b2 != 0u) : (a2 == 0u && b2 == 0u);
            j = left_still ? -1 : (up ? first : first - 1);
        }
    }
#undef FLQ
}

// RMAX = rows the instantiation can hold (12 / 20 / 32): the register budget -- and with it the
// number of resident wavefronts that hide the fp64 dependency chains -- follows the longest
// component of the call, not the longest the kernel family supports.  The levels that are backtracked
// have at most RMAX / 2 rows: one instantiation of the level for them, one for the finest.
template <int RMAX, int TIE, int MINB, bool WLDS>
__global__ __launch_bounds__(DTW_THREADS, MINB) void dtw_similarity_reg_kernel(
    const double* __restrict__ xpyr, const int32_t* __restrict__ xlen, int64_t n_x,
    const double* __restrict__ ypyr, const int32_t* __restrict__ ylen, int64_t n_y,
    float* __restrict__ out, uint32_t* __restrict__ wq, DtwLayout L, const int32_t* __restrict__ x_order,
    const int64_t* __restrict__ x_live)
{
    constexpr int RH = ((RMAX / 2) + 1) & ~1;                 // rows of the coarser levels, even
    __shared__ int32_t s_fl[RH * DTW_THREADS];
    extern __shared__ uint32_t s_words[];                    // WLDS: (max_y_len / 2) x DTW_THREADS predecessor words
    int32_t* fl = s_fl + threadIdx.x;
    uint32_t* wl = s_words + threadIdx.x;
    const int64_t NT = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // x_live = {first, count}: only these positions of the processing order hold non-empty rows (the caller
    // sorted the empty ones to the front and zeroed their output rows)
    const int64_t first_live = x_live ? x_live[0] : 0;
    const int64_t n_live = x_live ? x_live[1] : n_x;
    const int64_t chunks = (n_live + 63) / 64;                // a task = one anchor x 64 consecutive positions
    const int64_t n_tasks = chunks * n_y;
    const int64_t n_waves = NT / 64;
    const int64_t wave0 = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    for (int64_t task = wave0; task < n_tasks; task += n_waves) {
        const int64_t a = task / chunks;
        const int64_t p0 = first_live + (task - a * chunks) * 64 + lane;   // position in the processing order:
        const bool have = p0 < first_live + n_live;                        // the pyramids are laid out by position
        const int64_t pos = have ? p0 : first_live;
        const int64_t r = x_order ? x_order[pos] : pos;
        const int ly0 = ylen[a];
        const int lx0 = have ? xlen[pos] : 0;
        const bool valid = have && lx0 > 0 && ly0 > 0;
        int n_levels = 0;
        if (valid) {
            n_levels = 1;
            int lx = lx0, ly = ly0;
            while (lx >= 3 && ly >= 3) { lx >>= 1; ly >>= 1; ++n_levels; }
        }
        const int max_levels = (int)(dtw_wave_pkmax((uint32_t)n_levels) & 0xffff);
        const double* __restrict__ yp = ypyr + a * L.YL;
        double result = 0.0;
        for (int lev = max_levels - 1; lev >= 0; --lev) {
            const bool act = lev < n_levels;
            const int lx = act ? lx0 >> lev : 0, ly = ly0 >> lev;
            const int lxc = lx0 >> (lev + 1), lyc = ly0 >> (lev + 1);
            const double* xcol = xpyr + L.xoff[lev] * n_x + pos;
            const double* xrcol = xcol + L.XL * n_x;                        // reciprocal pyramids follow the values
            const double* ycol = yp + L.yoff[lev];
            const double* yrcol = ycol + L.YL * n_y;
            uint32_t* w = wq + (L.yoff[lev] - L.MY) * NT + tid;         // levels >= 1 only (not dereferenced on level 0)
            const bool coarsest = lev == n_levels - 1;
#ifdef DTW_PROBE_NO_FINEST                                       /* measurement only: wrong results */
            if (lev == 0) continue;
#endif
#ifdef DTW_PROBE_NO_COARSE
            if (lev != 0) continue;
#endif
            if (lev == 0)
                result = dtw_wave_level<RMAX, TIE, WLDS, true>(fl, xcol, xrcol, n_x, ycol, yrcol, act, lx, ly, lxc, lyc,
                                                               coarsest, wl, w, NT, lev);
            else if constexpr (2 * RH <= 27 && !DTW_OLD_COARSE) {
                // (wave-uniform: one anchor per wavefront.  Only in the instantiation whose words sit in LDS -- anchor series of up to
                // 96 entries: the other one serves longer series, whose first coarse level has more than 32 columns anyway)
                if (WLDS && ly <= 32 && !DTW_NO_ROW_MAJOR) {
                    // Level lev of a series of <= RMAX entries has <= RMAX >> lev rows: the second coarse level and the ones
                    // below it get instantiations of their own size (round 6).  Everything per row of the level function is
                    // unrolled over its row count -- loading the series, clearing two column arrays and two bit rows, the
                    // window and hull words, a back-trace step -- and with RH rows for every coarse level the 2-row and the
                    // 5-row level of a 20-entry series paid ~500 vector instructions each for rows they do not have.
                    constexpr int R2 = (((RMAX >> 2) + 1) & ~1) < 2 ? 2 : (((RMAX >> 2) + 1) & ~1);
                    constexpr int R3 = (((RMAX >> 3) + 1) & ~1) < 2 ? 2 : (((RMAX >> 3) + 1) & ~1);
                    if (DTW_COARSE_ONE_SIZE || lev == 1)
                        dtw_wave_level_rm<RH, TIE>(fl, xcol, xrcol, n_x, ycol, yrcol, act, lx, ly, lxc, lyc, coarsest, lev);
                    else if (lev == 2)
                        dtw_wave_level_rm<R2, TIE>(fl, xcol, xrcol, n_x, ycol, yrcol, act, lx, ly, lxc, lyc, coarsest, lev);
                    else
                        dtw_wave_level_rm<R3, TIE>(fl, xcol, xrcol, n_x, ycol, yrcol, act, lx, ly, lxc, lyc, coarsest, lev);
                }
                else
                    dtw_wave_level_pp<RH, TIE, WLDS>(fl, xcol, xrcol, n_x, ycol, yrcol, act, lx, ly, lxc, lyc, coarsest, wl, w, NT, lev);
            }
            else
                dtw_wave_level<RH, TIE, WLDS, false>(fl, xcol, xrcol, n_x, ycol, yrcol, act, lx, ly, lxc, lyc, coarsest,
                                                     wl, w, NT, lev);
        }
        if (have) out[r * n_y + a] = valid ? (float)(1.0 / (result + 1.0)) : 0.f;
    }
}

// Processing-order key of the x rows of a DTW call: (length, the row's TWICE-HALVED series -- means of four
// consecutive entries, what fastdtw's second coarsening level sees -- on a log scale, 4 steps per octave (round 5; 8 before), up to six of
// them, first to last).  A pair's finest-level window follows from its coarse warp paths, and those from the coarse
// series: rows whose coarse series agree sweep the same windows, so the lanes of a wavefront (consecutive rows of the
// order, same anchor) stay in step.  Replay of 32 wavefronts of the benchmark's external side through the oracle:
// cells evaluated per pair on the finest level 558 with round 1's key (length, four quantiles of the raw row), 500
// with this one (a lane's own window: 347); kernel 7.1 -> see DESIGN.  Results do not depend on the order.
__global__ void dtw_order_keys_kernel(const int64_t* __restrict__ x_ptr, const int32_t* __restrict__ x_val, int64_t n_x,
                                      int64_t* __restrict__ keys)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_x) return;
    const int64_t b = x_ptr[i], len = x_ptr[i + 1] - b;
    int64_t key = (len > 0xFFF ? (int64_t)0xFFF : len) << 48;
#ifndef DTW_KEY_STEPS
#define DTW_KEY_STEPS 4.f       // log-scale steps per octave.  Round 5, cells a wavefront evaluates on the finest level (tools/dtw_budget.py; a
#endif                          // lane's own window: 346.2) and the external side: 2 / 4 / 8 / 16 steps: 435.9 / 425.6 / 438.6 / 463.1 cells, 3.90 / 3.88 / 3.97 / 4.08 ms;
                                // components in reverse (largest degrees first) 581 cells, middle-out 480: the ascending order stays
#ifndef DTW_KEY_ORDER
#define DTW_KEY_ORDER 0
#endif
    auto q8 = [](float v) { const int q = (int)lrintf(DTW_KEY_STEPS * log2f(1.f + (v < 0.f ? 0.f : v))); return (int64_t)(q > 255 ? 255 : q); };
    const int64_t n2 = len / 4;
    if (n2 == 0) {
        for (int64_t f = 0; f < len; ++f) key |= q8((float)x_val[b + f]) << (40 - 8 * f);
    } else {
        const int64_t nf = n2 < 6 ? n2 : 6;
        for (int64_t f = 0; f < nf; ++f) {
            const int64_t g = b + 4 * ((f * n2) / nf);
            const float v = 0.25f * ((float)x_val[g] + (float)x_val[g + 1] + (float)x_val[g + 2] + (float)x_val[g + 3]);
            int64_t slot = f;                                        // significance of component f (0 = most significant)
            if (DTW_KEY_ORDER == 1) slot = nf - 1 - f;               // the largest degrees first
            else if (DTW_KEY_ORDER == 2) { const int64_t mid = nf / 2; const int64_t dlt = f - mid; slot = dlt == 0 ? 0 : (dlt > 0 ? 2 * dlt - 1 : -2 * dlt); if (slot >= nf) slot = nf - 1; }   // middle out
            key |= q8(v) << (40 - 8 * slot);
        }
    }
    keys[i] = key;
}

extern "C" int sgnn_dtw_order_keys(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t* out_keys, void* stream)
{
    if (!x_ptr || !x_val || !out_keys || n_x < 0) return SGNN_ERR_BAD_ARG;
    if (n_x == 0) return SGNN_OK;
    hipLaunchKernelGGL(dtw_order_keys_kernel, dim3((unsigned)((n_x + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_ptr,
                       x_val, n_x, out_keys);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

static int dtw_run(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t max_x_len,
                   const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int64_t max_y_len,
                   int tie_order, int kernel, const int32_t* x_order, const int64_t* x_live, float* out, void* workspace,
                   int64_t workspace_bytes, void* stream)
{
    if (!x_ptr || !x_val || !y_ptr || !y_val || !out || !workspace || n_x < 0 || n_y < 0) return SGNN_ERR_BAD_ARG;
    if (tie_order < 0 || tie_order > 2 || kernel < 0 || kernel > 1) return SGNN_ERR_BAD_ARG;
    if (max_x_len < 1) max_x_len = 1;
    if (max_y_len < 1) max_y_len = 1;
    if (max_x_len > 32767 || max_y_len > 32767) return SGNN_ERR_SET_TOO_LARGE;
    if (workspace_bytes < sgnn_dtw_workspace_bytes(n_x, max_x_len, n_y, max_y_len)) return SGNN_ERR_BAD_ARG;
    if (n_x * n_y == 0) return SGNN_OK;
    const DtwLayout L = dtw_layout(max_x_len, max_y_len);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    const int64_t scratch = dtw_scratch_bytes(L);
    double* wd = (double*)w;
    int32_t* wi = (int32_t*)(w + DTW_NT * L.n_dbl * 8);
    uint32_t* wb = (uint32_t*)(w + DTW_NT * (L.n_dbl * 8 + dtw_align8(L.n_i32 * 4)));
    uint32_t* wq = (uint32_t*)w;           w += scratch;
    double* xpyr = (double*)w;             w += 2 * n_x * L.XL * 8;          // values, then reciprocals of value + 1
    double* ypyr = (double*)w;             w += 2 * n_y * L.YL * 8;
    int32_t* xlen = (int32_t*)w;           w += dtw_align8(n_x * 4);
    int32_t* ylen = (int32_t*)w;
    const bool use_reg = max_x_len <= DTW_R && kernel == 0;
    dtw_launch_pyramid(x_ptr, x_val, n_x, max_x_len, L.XL, 1, xpyr, xpyr + n_x * L.XL, xlen,
                       use_reg ? x_order : (const int32_t*)nullptr, st);
    SGNN_CHECK_LAUNCH();
    dtw_launch_pyramid(y_ptr, y_val, n_y, max_y_len, L.YL, 0, ypyr, ypyr + n_y * L.YL, ylen, (const int32_t*)nullptr, st);
    SGNN_CHECK_LAUNCH();
    if (use_reg) {
        // predecessor words of the coarse levels in LDS when (max_y_len / 2) words per lane fit
        const int64_t words = max_y_len >> 1;
        const bool wlds = dtw_words_in_lds(max_y_len);
        // SGNN_DTW_LDS_PAD (bytes, measurement only: tools/dtw_overlap_probe.py): extra dynamic LDS per workgroup, which lowers the
        // number of resident workgroups per CU (>= 20 KB: two instead of three) and leaves vector registers to other streams' kernels
        // -- compiled in only with -DDTW_PROBE_LDS_PAD (the production launch reads no environment variable), clamped to what is left
        // of a workgroup's 64 KB
        size_t dyn = wlds ? (size_t)((words > 0 ? words : 1) * DTW_THREADS * 4) : 0;
#ifdef DTW_PROBE_LDS_PAD
        static const long lds_pad = getenv("SGNN_DTW_LDS_PAD") ? atol(getenv("SGNN_DTW_LDS_PAD")) : 0;
        if (lds_pad > 0) dyn += (size_t)(lds_pad < (long)(48 * 1024 - (long)dyn) ? lds_pad : (48 * 1024 > (long)dyn ? 48 * 1024 - (long)dyn : 0));
#endif
#define DTW_LAUNCH2(RMAX, TIE, MINB, WL) \
        hipLaunchKernelGGL((dtw_similarity_reg_kernel<RMAX, TIE, MINB, WL>), dim3(DTW_REG_BLOCKS), dim3(DTW_THREADS), dyn, st, \
                           xpyr, xlen, n_x, ypyr, ylen, n_y, out, wq, L, x_order, x_live)
#define DTW_LAUNCH(RMAX, TIE, MINB) do { if (wlds) DTW_LAUNCH2(RMAX, TIE, MINB, true); else DTW_LAUNCH2(RMAX, TIE, MINB, false); } while (0)
#define DTW_LAUNCH_T(RMAX, MINB) do { if (tie_order == 0) DTW_LAUNCH(RMAX, 0, MINB); else if (tie_order == 1) DTW_LAUNCH(RMAX, 1, MINB); else DTW_LAUNCH(RMAX, 2, MINB); } while (0)
        if (max_x_len <= 12) DTW_LAUNCH_T(12, DTW_MINB12);
        else if (max_x_len <= 20) DTW_LAUNCH_T(20, DTW_MINB20);
        else DTW_LAUNCH_T(32, DTW_MINB32);
#undef DTW_LAUNCH_T
#undef DTW_LAUNCH2
#undef DTW_LAUNCH
    } else {
        hipLaunchKernelGGL(dtw_similarity_kernel, dim3(DTW_BLOCKS), dim3(DTW_THREADS), 0, st, xpyr, xlen, n_x, ypyr,
                           ylen, n_y, tie_order, out, wd, wi, wb, L);
    }
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_dtw_similarity(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t max_x_len,
                                   const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int64_t max_y_len,
                                   int tie_order, int kernel, const int32_t* x_order, float* out, void* workspace,
                                   int64_t workspace_bytes, void* stream)
{
    return dtw_run(x_ptr, x_val, n_x, max_x_len, y_ptr, y_val, n_y, max_y_len, tie_order, kernel, x_order, nullptr, out, workspace,
                   workspace_bytes, stream);
}

extern "C" int sgnn_dtw_similarity_live(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t max_x_len,
                                        const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int64_t max_y_len,
                                        int tie_order, int kernel, const int32_t* x_order, const int64_t* x_live_range,
                                        float* out, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (x_live_range && !x_order) return SGNN_ERR_BAD_ARG;
    return dtw_run(x_ptr, x_val, n_x, max_x_len, y_ptr, y_val, n_y, max_y_len, tie_order, kernel, x_order, x_live_range, out,
                   workspace, workspace_bytes, stream);
}

SGNN_DEFINE_WARM(dtw)
