// Deterministic row scatter-add: table[key[e], :] += c1[e] * G[row[e], :] + c2[e] * v[:]  for edges e
// given in an order sorted (stably) by target key.  Replaces the float atomics of the embedding-table
// gradient (a18: SubGNN/SubGNN.py:1163-1164 through autograd of a12 / a13 / a15): with atomics the
// 137.6 M adds of the benchmark's neighbourhood-border layer ran at the ~6 G cache-line operations per
// second the memory side sustains for device-scope float atomics, in an order that changes from run to
// run.  Here every table row has exactly one writer and a fixed summation order:
//
//   kernel A  one wavefront per run of 64 consecutive sorted positions, lane = column.  The run's
//             keys / edge numbers / coefficients are loaded one per lane and read back lane by lane
//             (scalar), the source rows are read as whole contiguous rows (256 B for D = 64), and a
//             register accumulates until the key changes.  A segment that lies inside the run is added
//             to the table with a plain read-modify-write; the first / last segment of a run that
//             continues from / into the neighbouring run goes to a carry slot instead;
//   kernel B  one wavefront per chain of carry slots (a target with thousands of edges -- a hub that
//             sits in 15 k subgraphs -- spans hundreds of runs): the partials are summed in run order
//             and added to the table.
//
// Traffic: the sorted order (8 B per edge), the coefficient gathers, one source row per edge (L2 hits:
// a source row is shared by all edges of its component) and one table row per DISTINCT target.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#define SC_RUN 64
#define SC_RUN_SMALL 16        // run length of batch-sized launches: a wavefront walks its run edge by edge, so 64-edge runs make a launch of a
                               // few thousand edges a chain of 64 dependent steps on a handful of wavefronts (HPO-METAB stand-in, batch of 64:
                               // 21 launches of 31 us per step); 16-edge runs give the same launch four times the wavefronts
#define SC_SMALL_EDGES 65536   // launches up to this many edges take the short runs
#define SC_AHEAD 8             // carry slots in flight per wavefront in the chain kernel
#define SC_MAXC 4              // columns per lane: D <= 256

struct ScArgs {
    const int32_t* order;      // E: edge numbers, sorted by target key (stable)
    const int32_t* key;        // E: key of order[p], ascending; key 0 (PAD) contributes nothing
    int64_t E;
    const int32_t* edge_row;   // nullable: source row of edge e; NULL: e / edges_per_row
    int64_t edges_per_row;
    const float* G;            // (rows, D), nullable (then only the c2 * v term exists)
    int64_t D;
    const float* c1;           // nullable per-edge multiplier of the source row (NULL = 1)
    const float* c2;           // nullable per-edge multiplier of v
    const float* v;            // (D), with c2
    const int32_t* arg;        // nullable (rows, D): column d of edge e counts only if arg[row, d] == key
    float* table;
    int32_t* carry_key;        // 2 per run: head, tail
    int32_t* carry_flag;       // 2 per run: bit 0 = present, bit 1 (head only) = the segment runs on into the next run
    float* carry_val;          // 2 per run x D
    // MULTI (several scatters into one table as ONE sorted list, sgnn_scatter_add_rows_multi): per edge the ADDRESS of its
    // source row (0 = none) and of its v (0 = none); c1 / c2 are then per-edge arrays over the concatenated edges
    const unsigned long long* src_rows;
    const unsigned long long* v_rows;
};

// AHEAD source rows are requested together, MAXC columns per lane.  For D <= 64: (64, 1) when the launch has fewer
// wavefronts than the chip holds -- the whole run's rows are in registers before the first add and nothing in the
// walk waits on memory (with fewer rows in flight the wait for the next group's rows also waits for the table adds
// issued in between: the memory counter is in-order) -- and (8, 1) for large launches, where 8 resident wavefronts
// per SIMD hide those waits better than 3 with 159 registers each (benchmark, 2.1 M border edges: 293 us with 64 rows
// in flight, 218 with 8; what remains is the random 256-byte read-modify-write of ~0.9 M distinct table rows in HBM).
// (16, 1) with the argmax ids, (16, 2) for D <= 128, (8, 4) up to 256.
__device__ __forceinline__ unsigned long long sc_readlane64(unsigned long long v, int lane)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

template <int AHEAD, int MAXC, int RUN = SC_RUN, bool MULTI = false>
__global__ __launch_bounds__(256) void scatter_runs_kernel(ScArgs a)
{
    static_assert(RUN <= 64 && AHEAD <= RUN, "one sorted position per lane");
    const int lane = threadIdx.x & 63;
    const int64_t run = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t p0 = run * RUN;
    if (p0 >= a.E) return;
    const int cnt = (int)(a.E - p0 < RUN ? a.E - p0 : RUN);
    const int64_t D = a.D;
    // one sorted position per lane
    int32_t k_l = 0, row_l = 0;
    float c1_l = 0.f, c2_l = 0.f;
    unsigned long long src_l = 0ull, v_l = 0ull;
    if (lane < cnt) {
        k_l = a.key[p0 + lane];
        const int64_t e = a.order[p0 + lane];
        if (MULTI) { src_l = a.src_rows[e]; v_l = a.v_rows[e]; }
        else row_l = a.edge_row ? a.edge_row[e] : (int32_t)(e / a.edges_per_row);
        c1_l = a.c1 ? a.c1[e] : 1.f;
        c2_l = a.c2 ? a.c2[e] : 0.f;
    }
    const int32_t prev_key = p0 > 0 ? a.key[p0 - 1] : -1;
    const int32_t next_key = p0 + cnt < a.E ? a.key[p0 + cnt] : -1;
    float vv[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) vv[c] = (!MULTI && a.v && lane + 64 * c < D) ? a.v[lane + 64 * c] : 0.f;
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    int32_t cur = __builtin_amdgcn_readlane(k_l, 0);
    // arg mode (max aggregator): a member listed twice in its set counts once, as the argmax records one id per
    // column -- the stable sort keeps the entries of one source row adjacent inside a segment, so "same row as
    // the previous entry of this key" identifies the repeats (also across the run boundary)
    int64_t last_row = -1;
    if (!MULTI && a.arg && p0 > 0 && prev_key == cur) {
        const int64_t pe = a.order[p0 - 1];
        last_row = a.edge_row ? a.edge_row[pe] : pe / a.edges_per_row;
    }
    bool first_seg = true;
    int32_t head_flag = 0, tail_flag = 0;

    auto flush = [&](int32_t key, bool is_last) {
        // where does the finished segment go?
        const bool from_prev = first_seg && key == prev_key;
        const bool into_next = is_last && key == next_key;
        if (key != 0) {
            if (from_prev) {                                   // head carry (maybe running on)
                head_flag = 1 | (into_next ? 2 : 0);
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (lane + 64 * c < D) a.carry_val[(2 * run) * D + lane + 64 * c] = acc[c];
                if (lane == 0) a.carry_key[2 * run] = key;
            } else if (into_next) {                            // tail carry: starts a chain
                tail_flag = 1;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (lane + 64 * c < D) a.carry_val[(2 * run + 1) * D + lane + 64 * c] = acc[c];
                if (lane == 0) a.carry_key[2 * run + 1] = key;
            } else {                                           // the segment is ours alone
                // one writer per table row in this launch, so a returnless float add is as deterministic as a
                // read-modify-write -- and the wavefront does not wait for the row to come back (a run of the
                // benchmark's border edges ends ~30 segments: 30 dependent round trips before)
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (lane + 64 * c < D) unsafeAtomicAdd(&a.table[(int64_t)key * D + lane + 64 * c], acc[c]);
            }
        }
        first_seg = false;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    };

    // The run is walked AHEAD edges at a time: their source rows (one coalesced row read each, an L2 hit) are
    // requested together, then added in order.  Reading them one by one made the run a chain of 64 dependent
    // cache round trips per wavefront (0.27 ms for the benchmark's 2.1 M border edges).
    for (int j0 = 0; j0 < cnt; j0 += AHEAD) {
        float g[AHEAD][MAXC];
        float wv[MULTI ? AHEAD : 1][MAXC];
        int32_t am[AHEAD][MAXC];
        // (keys ascend: the PAD edges -- masked or zero-weight ones -- fill whole runs at the front; no rows are read for them)
        if (MULTI) {
            if (__builtin_amdgcn_readlane(k_l, (j0 + AHEAD - 1) < cnt ? (j0 + AHEAD - 1) : cnt - 1) != 0) {
#pragma unroll
                for (int u = 0; u < AHEAD; ++u) {
                    const float* sp = reinterpret_cast<const float*>(sc_readlane64(src_l, (j0 + u) & 63));   // lanes past cnt hold 0
                    const float* vp = reinterpret_cast<const float*>(sc_readlane64(v_l, (j0 + u) & 63));
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int64_t d = lane + 64 * c;
                        g[u][c] = (sp && d < D) ? sp[d] : 0.f;
                        wv[u][c] = (vp && d < D) ? vp[d] : 0.f;
                    }
                }
            }
        } else
        if (a.G && __builtin_amdgcn_readlane(k_l, (j0 + AHEAD - 1) < cnt ? (j0 + AHEAD - 1) : cnt - 1) != 0) {
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) {
                const int64_t row = __builtin_amdgcn_readlane(row_l, (j0 + u) & 63);   // lanes past cnt hold row 0: a valid row
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int64_t d = lane + 64 * c;
                    g[u][c] = d < D ? a.G[row * D + d] : 0.f;
                    am[u][c] = (a.arg && d < D) ? a.arg[row * D + d] : 0;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            const int j = j0 + u;
            if (j >= cnt) break;
            const int32_t k = __builtin_amdgcn_readlane(k_l, j);
            if (k != cur) { flush(cur, false); cur = k; last_row = -1; }
            if (k == 0) continue;
            const int64_t row = __builtin_amdgcn_readlane(row_l, j);
            if (!MULTI && a.arg) { if (row == last_row) continue; last_row = row; }
            const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c1_l), j));
            const float c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c2_l), j));
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (lane + 64 * c < D) {
                    if (MULTI) {
                        acc[c] += c2 * wv[u][c] + c1 * g[u][c];
                    } else {
                        float val = c2 * vv[c];
                        if (a.G) {
                            if (a.arg) { if (am[u][c] == k) val += c1 * g[u][c]; }
                            else val += c1 * g[u][c];
                        }
                        acc[c] += val;
                    }
                }
            }
        }
    }
    flush(cur, true);
    if (lane == 0) { a.carry_flag[2 * run] = head_flag; a.carry_flag[2 * run + 1] = tail_flag; }
}

__global__ __launch_bounds__(256) void scatter_chains_kernel(ScArgs a, int64_t n_runs)
{
    const int lane = threadIdx.x & 63;
    const int64_t run = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (run >= n_runs) return;
    if (!(a.carry_flag[2 * run + 1] & 1)) return;              // no chain starts here
    const int64_t D = a.D;
    const int32_t key = a.carry_key[2 * run + 1];
    float acc[SC_MAXC];
#pragma unroll
    for (int c = 0; c < SC_MAXC; ++c) acc[c] = lane + 64 * c < D ? a.carry_val[(2 * run + 1) * D + lane + 64 * c] : 0.f;
    // SC_AHEAD links at a time: flags and partials of the next runs are requested together and added in run order
    // (a hub that sits in 15 k components spans ~230 runs: 460 dependent round trips one by one)
    bool more = true;
    for (int64_t j0 = run + 1; more && j0 < n_runs; j0 += SC_AHEAD) {
        int32_t f[SC_AHEAD];
        float val[SC_AHEAD][SC_MAXC];
#pragma unroll
        for (int u = 0; u < SC_AHEAD; ++u) {
            const int64_t j = j0 + u < n_runs ? j0 + u : n_runs - 1;
            f[u] = a.carry_flag[2 * j];
#pragma unroll
            for (int c = 0; c < SC_MAXC; ++c) val[u][c] = lane + 64 * c < D ? a.carry_val[(2 * j) * D + lane + 64 * c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SC_AHEAD; ++u) {
            if (more && j0 + u < n_runs && (f[u] & 1)) {       // (a tail implies a head next door)
#pragma unroll
                for (int c = 0; c < SC_MAXC; ++c) acc[c] += val[u][c];
                if (!(f[u] & 2)) more = false;
            } else {
                more = false;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < SC_MAXC; ++c)
        if (lane + 64 * c < D) a.table[(int64_t)key * D + lane + 64 * c] += acc[c];
}

static inline int sc_run_len(int64_t n_edges) { return n_edges <= SC_SMALL_EDGES ? SC_RUN_SMALL : SC_RUN; }

extern "C" int64_t sgnn_scatter_add_rows_workspace_bytes(int64_t n_edges, int64_t D)
{
    const int run = sc_run_len(n_edges);
    const int64_t n_runs = (n_edges + run - 1) / run;
    return n_runs * 2 * (D * 4 + 8) + 64;
}

// ---- stable sort of the edges by target key -------------------------------------------------------
// rocPRIM's radix sort of (key, position) pairs over the bits the keys can have (20 for a million-node
// table: 3 passes instead of the 4 a full 32-bit sort takes), positions as int32.
static inline int sc_key_bits(int64_t max_key) { int b = 1; while (b < 31 && (1ll << b) <= max_key) ++b; return b; }

static hipError_t sc_sort(void* temp, size_t& temp_bytes, const int32_t* keys, int32_t* keys_out, int32_t* order_out,
                          int64_t n, int bits, hipStream_t st)
{
    // (rocPRIM's default: a merge sort up to 2^20 keys -- a block sort + two launches per doubling --, the one-sweep radix sort
    // beyond.  Forcing one-sweep from 8 193 keys on was measured: 15 k-100 k keys, the batch-sized steps' lists, sort SLOWER that
    // way (DENSITY stand-in step 0.86 -> 1.22 ms, PPI-BP 1.35 -> 1.45) and the 269 k keys of a 6 250-subgraph shard no faster.)
    return rocprim::radix_sort_pairs(temp, temp_bytes, reinterpret_cast<const uint32_t*>(keys), reinterpret_cast<uint32_t*>(keys_out),
                                     rocprim::counting_iterator<int32_t>(0), order_out, (size_t)n, 0u, (unsigned)bits, st);
}

// Batch-sized lists (a step of 64 subgraphs sorts 1-5 k edges 6-18 times): rocPRIM's device sort is a block sort + 2-7 merge
// launches of ~4.7 us each whatever the size.  Here ONE workgroup of 1024 lanes radix-sorts up to SC_ONE_WG (key, position) pairs
// (rocprim::block_radix_sort: 8 bits per pass over the bits the keys can have, stable): one launch.  (A bitonic network over
// (key | position) words in LDS was tried first: 91 barrier stages, ~20 us per sort -- slower than the launches it replaced.)
#define SC_ONE_WG 8192
template <int IPT>
__global__ __launch_bounds__(1024) void sort_one_wg_kernel(const int32_t* __restrict__ keys, int n, int bits,
                                                           int32_t* __restrict__ key_sorted, int32_t* __restrict__ order)
{
    using Sort = rocprim::block_radix_sort<uint32_t, 1024, IPT, int32_t>;
    __shared__ typename Sort::storage_type storage;
    uint32_t k[IPT];
    int32_t v[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int i = threadIdx.x * IPT + u;                   // blocked arrangement: the sort is stable in this order
        k[u] = i < n ? (uint32_t)keys[i] : 0xFFFFFFFFu;        // padding ties with the largest key at worst, and comes after it
        v[u] = i;
    }
    Sort().sort(k, v, storage, 0u, (unsigned)bits);
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int i = threadIdx.x * IPT + u;
        if (i < n) { key_sorted[i] = (int32_t)k[u]; order[i] = v[u]; }
    }
}

static bool sc_sort_one_wg(const int32_t* keys, int32_t* keys_out, int32_t* order_out, int64_t n, int bits, hipStream_t st)
{
    if (n > SC_ONE_WG) return false;
    const int m = (int)n;
    // (the sort's time grows with the items per lane: 6.5 us at 1, 26 us at 8 -- the step of a 4-layer model sorts 4 352 keys 8 times)
#define SC_ONE_WG_CASE(IPT) if (m <= 1024 * IPT) { hipLaunchKernelGGL(sort_one_wg_kernel<IPT>, dim3(1), dim3(1024), 0, st, keys, m, bits, keys_out, order_out); return true; }
    SC_ONE_WG_CASE(1) SC_ONE_WG_CASE(2) SC_ONE_WG_CASE(3) SC_ONE_WG_CASE(4) SC_ONE_WG_CASE(5) SC_ONE_WG_CASE(6) SC_ONE_WG_CASE(8)
#undef SC_ONE_WG_CASE
    return true;
}

extern "C" int64_t sgnn_sort_edges_by_key_workspace_bytes(int64_t n_edges, int64_t max_key)
{
    if (n_edges <= 0) return 0;
    size_t bytes = 0;
    if (sc_sort(nullptr, bytes, nullptr, nullptr, nullptr, n_edges, sc_key_bits(max_key), nullptr) != hipSuccess) return -1;
    return (int64_t)bytes + 256;
}

extern "C" int sgnn_sort_edges_by_key(const int32_t* keys, int64_t n_edges, int64_t max_key, int32_t* key_sorted,
                                      int32_t* order, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (n_edges < 0 || max_key < 0 || max_key >= (1ll << 31)) return SGNN_ERR_BAD_ARG;
    if (n_edges == 0) return SGNN_OK;
    if (!keys || !key_sorted || !order || !workspace) return SGNN_ERR_BAD_ARG;
    if (n_edges >= (1ll << 31)) return SGNN_ERR_SET_TOO_LARGE;
    const int bits = sc_key_bits(max_key);
    if (sc_sort_one_wg(keys, key_sorted, order, n_edges, bits, (hipStream_t)stream)) { SGNN_CHECK_LAUNCH(); return SGNN_OK; }
    size_t need = 0;
    if (sc_sort(nullptr, need, nullptr, nullptr, nullptr, n_edges, bits, nullptr) != hipSuccess) return SGNN_ERR_LAUNCH;
    if ((int64_t)need > workspace_bytes) return SGNN_ERR_BAD_ARG;
    const hipError_t e = sc_sort(workspace, need, keys, key_sorted, order, n_edges, bits, (hipStream_t)stream);
    if (e != hipSuccess) { sgnn_set_last_error(e); return SGNN_ERR_LAUNCH; }
    return SGNN_OK;
}

extern "C" int sgnn_scatter_add_rows_sorted(const int32_t* order, const int32_t* key_sorted, int64_t n_edges,
                                            const int32_t* edge_row, int64_t edges_per_row,
                                            const float* G, int64_t D, const float* c1, const float* c2, const float* v,
                                            const int32_t* arg, float* table,
                                            void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!order || !key_sorted || !table || n_edges < 0 || D <= 0 || (!G && !c2) || (c2 && !v) || (arg && !G)) return SGNN_ERR_BAD_ARG;
    if (!edge_row && edges_per_row < 1) return SGNN_ERR_BAD_ARG;
    if (D > 64 * SC_MAXC) return SGNN_ERR_UNSUPPORTED_D;
    if (n_edges >= (1ll << 31)) return SGNN_ERR_SET_TOO_LARGE;
    if (n_edges == 0) return SGNN_OK;
    if (!workspace || workspace_bytes < sgnn_scatter_add_rows_workspace_bytes(n_edges, D)) return SGNN_ERR_BAD_ARG;
    const int run_len = sc_run_len(n_edges);
    const int64_t n_runs = (n_edges + run_len - 1) / run_len;
    ScArgs a;
    a.order = order; a.key = key_sorted; a.E = n_edges; a.edge_row = edge_row; a.edges_per_row = edges_per_row;
    a.G = G; a.D = D; a.c1 = c1; a.c2 = c2; a.v = v; a.arg = arg; a.table = table;
    a.src_rows = nullptr; a.v_rows = nullptr;
    a.carry_val = (float*)workspace;
    a.carry_key = (int32_t*)(a.carry_val + n_runs * 2 * D);
    a.carry_flag = a.carry_key + n_runs * 2;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n_runs + 3) / 4);
    if (run_len == SC_RUN_SMALL) {
        // batch-sized launch: whole short runs in flight (16 rows requested together), one instantiation per column count
        if (D <= 64) hipLaunchKernelGGL((scatter_runs_kernel<SC_RUN_SMALL, 1, SC_RUN_SMALL>), dim3(grid), dim3(256), 0, st, a);
        else if (D <= 128) hipLaunchKernelGGL((scatter_runs_kernel<SC_RUN_SMALL, 2, SC_RUN_SMALL>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((scatter_runs_kernel<8, 4, SC_RUN_SMALL>), dim3(grid), dim3(256), 0, st, a);
    }
    else if (D <= 64 && !arg && n_runs <= 8192) hipLaunchKernelGGL((scatter_runs_kernel<64, 1>), dim3(grid), dim3(256), 0, st, a);
    else if (D <= 64 && !arg) hipLaunchKernelGGL((scatter_runs_kernel<8, 1>), dim3(grid), dim3(256), 0, st, a);
    else if (D <= 64) hipLaunchKernelGGL((scatter_runs_kernel<16, 1>), dim3(grid), dim3(256), 0, st, a);   // (+ the argmax ids: 2 registers per row)
    else if (D <= 128) hipLaunchKernelGGL((scatter_runs_kernel<16, 2>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((scatter_runs_kernel<8, 4>), dim3(grid), dim3(256), 0, st, a);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(scatter_chains_kernel, dim3(grid), dim3(256), 0, st, a, n_runs);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---- several scatters into one table as ONE sorted list ---------------------------------------------------------------------
// A batch-sized training step scatters 6-18 short edge lists (1-30 k edges each: one per layer body, the component embeddings,
// the shared anchors' lookups) into the same table gradient: 6-18 sorts + 12-36 scatter launches, each a few microseconds of
// work behind 4-13 us of launch and sort latency.  Here the lists are concatenated (one pack launch writes, per edge, its key,
// the ADDRESS of its source row and of its v, and its two coefficients), sorted once and scattered once.  Each table row still
// has one writer and a fixed order of addition (by list, then by position in the list: the sort is stable).
#define SC_MAX_JOBS 40
struct ScPackJob {
    const int32_t* keys; const int32_t* edge_row; const float* G; const float* c1; const float* c2; const float* v;
    long long edges_per_row, n, off;
    int blk;
};
struct ScPack { ScPackJob j[SC_MAX_JOBS]; int blk_end; int count; };

__global__ __launch_bounds__(256) void scatter_pack_kernel(const ScPack P, int64_t D, int32_t* __restrict__ key_all,
                                                           unsigned long long* __restrict__ src_rows,
                                                           unsigned long long* __restrict__ v_rows, float* __restrict__ c1_all,
                                                           float* __restrict__ c2_all)
{
    const int b = blockIdx.x;
    int lo = 0, hi = P.count - 1;                    // the block's list: largest t with blk[t] <= b
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (P.j[mid].blk <= b) lo = mid; else hi = mid - 1; }
    const ScPackJob& J = P.j[lo];
    const int64_t i = (int64_t)(b - J.blk) * 256 + threadIdx.x;
    if (i >= J.n) return;
    const int64_t e = J.off + i;
    const int64_t row = J.edge_row ? J.edge_row[i] : i / J.edges_per_row;
    key_all[e] = J.keys[i];
    src_rows[e] = J.G ? (unsigned long long)(uintptr_t)(J.G + row * D) : 0ull;
    const float c2 = J.c2 ? J.c2[i] : 0.f;
    v_rows[e] = (J.v && J.c2) ? (unsigned long long)(uintptr_t)J.v : 0ull;
    c1_all[e] = J.c1 ? J.c1[i] : 1.f;
    c2_all[e] = c2;
}

static inline int64_t sc_align(int64_t x) { return (x + 255) & ~255ll; }

extern "C" int64_t sgnn_scatter_add_rows_multi_workspace_bytes(int64_t total_edges, int64_t D, int64_t max_key)
{
    if (total_edges <= 0) return 0;
    // keys, sorted keys, order (4 B each), source / v addresses (8 B each), two coefficients (4 B each), the sort's and the scatter's own
    return sc_align(total_edges * 4) * 5 + sc_align(total_edges * 8) * 2 + sc_align(sgnn_sort_edges_by_key_workspace_bytes(total_edges, max_key))
           + sc_align(sgnn_scatter_add_rows_workspace_bytes(total_edges, D)) + 256;
}

extern "C" int sgnn_scatter_add_rows_multi(int64_t n_lists, const int32_t* const* keys, const int64_t* n_edges,
                                           const int32_t* const* edge_row, const int64_t* edges_per_row, const float* const* G,
                                           const float* const* c1, const float* const* c2, const float* const* v, int64_t D,
                                           int64_t max_key, float* table, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (n_lists < 0 || D <= 0 || !table || (n_lists && (!keys || !n_edges || !edge_row || !edges_per_row || !G || !c1 || !c2 || !v)))
        return SGNN_ERR_BAD_ARG;
    if (D > 64 * SC_MAXC) return SGNN_ERR_UNSUPPORTED_D;
    int64_t total = 0;
    for (int64_t k = 0; k < n_lists; ++k) {
        if (n_edges[k] < 0 || (n_edges[k] && !keys[k]) || (!G[k] && !c2[k]) || (c2[k] && !v[k]) || (!edge_row[k] && edges_per_row[k] < 1))
            return SGNN_ERR_BAD_ARG;
        total += n_edges[k];
    }
    if (total >= (1ll << 31)) return SGNN_ERR_SET_TOO_LARGE;
    if (total == 0) return SGNN_OK;
    if (!workspace || workspace_bytes < sgnn_scatter_add_rows_multi_workspace_bytes(total, D, max_key)) return SGNN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    int32_t* key_all = (int32_t*)w; w += sc_align(total * 4);
    int32_t* key_sorted = (int32_t*)w; w += sc_align(total * 4);
    int32_t* order = (int32_t*)w; w += sc_align(total * 4);
    float* c1_all = (float*)w; w += sc_align(total * 4);
    float* c2_all = (float*)w; w += sc_align(total * 4);
    unsigned long long* src_rows = (unsigned long long*)w; w += sc_align(total * 8);
    unsigned long long* v_rows = (unsigned long long*)w; w += sc_align(total * 8);
    void* sort_ws = w;
    const int64_t sort_bytes = sc_align(sgnn_sort_edges_by_key_workspace_bytes(total, max_key));
    w += sort_bytes;
    void* scatter_ws = w;
    // 1. pack
    int64_t off = 0;
    for (int64_t from = 0; from < n_lists; from += SC_MAX_JOBS) {
        const int64_t to = from + SC_MAX_JOBS < n_lists ? from + SC_MAX_JOBS : n_lists;
        ScPack P;
        int blocks = 0;
        P.count = 0;
        for (int64_t k = from; k < to; ++k) {
            if (n_edges[k] == 0) continue;
            ScPackJob& J = P.j[P.count++];
            J.keys = keys[k]; J.edge_row = edge_row[k]; J.G = G[k]; J.c1 = c1[k]; J.c2 = c2[k]; J.v = v[k];
            J.edges_per_row = edges_per_row[k]; J.n = n_edges[k]; J.off = off; J.blk = blocks;
            blocks += (int)((n_edges[k] + 255) / 256);
            off += n_edges[k];
        }
        P.blk_end = blocks;
        if (!P.count) continue;
        hipLaunchKernelGGL(scatter_pack_kernel, dim3(blocks), dim3(256), 0, st, P, D, key_all, src_rows, v_rows, c1_all, c2_all);
        SGNN_CHECK_LAUNCH();
    }
    // 2. one stable sort of all keys
    int rc = sgnn_sort_edges_by_key(key_all, total, max_key, key_sorted, order, sort_ws, sort_bytes, stream);
    if (rc != SGNN_OK) return rc;
    // 3. one scatter
    const int run_len = sc_run_len(total);
    const int64_t n_runs = (total + run_len - 1) / run_len;
    ScArgs a;
    a.order = order; a.key = key_sorted; a.E = total; a.edge_row = nullptr; a.edges_per_row = 1;
    a.G = nullptr; a.D = D; a.c1 = c1_all; a.c2 = c2_all; a.v = nullptr; a.arg = nullptr; a.table = table;
    a.src_rows = src_rows; a.v_rows = v_rows;
    a.carry_val = (float*)scatter_ws;
    a.carry_key = (int32_t*)(a.carry_val + n_runs * 2 * D);
    a.carry_flag = a.carry_key + n_runs * 2;
    const unsigned grid = (unsigned)((n_runs + 3) / 4);
    if (run_len == SC_RUN_SMALL) {
        if (D <= 64) hipLaunchKernelGGL((scatter_runs_kernel<SC_RUN_SMALL, 1, SC_RUN_SMALL, true>), dim3(grid), dim3(256), 0, st, a);
        else if (D <= 128) hipLaunchKernelGGL((scatter_runs_kernel<SC_RUN_SMALL, 2, SC_RUN_SMALL, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((scatter_runs_kernel<8, 4, SC_RUN_SMALL, true>), dim3(grid), dim3(256), 0, st, a);
    }
    else if (D <= 64) hipLaunchKernelGGL((scatter_runs_kernel<16, 1, SC_RUN, true>), dim3(grid), dim3(256), 0, st, a);
    else if (D <= 128) hipLaunchKernelGGL((scatter_runs_kernel<16, 2, SC_RUN, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((scatter_runs_kernel<8, 4, SC_RUN, true>), dim3(grid), dim3(256), 0, st, a);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(scatter_chains_kernel, dim3(grid), dim3(256), 0, st, a, n_runs);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(scatter)
