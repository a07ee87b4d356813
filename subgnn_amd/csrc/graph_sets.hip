// Integer graph kernels on ragged node sets: connected components of induced subgraphs (a7),
// k-hop border of a component (a8), in-border nodes of an anchor patch (a3).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// a7  connected components (reference SubGNN/SubGNN.py:589-592)
// One wavefront per subgraph.  Positions 0..n-1 of the subgraph are the union-find elements
// (parents in LDS); every ordered pair (i<j) is tested for adjacency by binary search in the
// shorter of the two sorted neighbour lists, and adjacent / identical nodes are united by a
// lock-free hook of the larger root under the smaller one, so a component's root is its
// smallest position.  Integer-only; HBM traffic is the two rowptr pairs + O(log deg) probes.
// ---------------------------------------------------------------------------------------------
#define CC_MAX 2048

__device__ static inline int cc_find(volatile int32_t* parent, int x) {
    int p = parent[x];
    while (p != x) { x = p; p = parent[x]; }
    return x;
}

// NMAX = 64: subgraphs of at most 64 nodes (the common case) keep 512 B of LDS per wavefront, so
// the CU holds its full complement of wavefronts; NMAX = CC_MAX handles the rest (and only those).
template <int NMAX>
__global__ __launch_bounds__(64) void cc_labels_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col_sorted,
    const int64_t* __restrict__ sub_ptr, const int32_t* __restrict__ sub_nodes, int64_t n_sub,
    int32_t* __restrict__ out_label)
{
    __shared__ int32_t s_id[NMAX];
    __shared__ int32_t s_parent[NMAX];
    __shared__ int64_t s_r0[NMAX];                          // the members' row starts and degrees: read once, not per pair
    __shared__ int32_t s_deg[NMAX];
    const int lane = threadIdx.x;
    for (int64_t s = blockIdx.x; s < n_sub; s += gridDim.x) {
        const int64_t beg = sub_ptr[s];
        const int n = (int)(sub_ptr[s + 1] - beg);
        if (n <= 0) continue;
        if (NMAX == 64 ? n > 64 : n <= 64) continue;        // the other instantiation owns it
        if (n > CC_MAX) {                                   // flagged by the host wrapper too
            for (int i = lane; i < n; i += 64) out_label[beg + i] = -1;
            continue;
        }
        for (int i = lane; i < n; i += 64) {
            const int32_t v = sub_nodes[beg + i];
            const int64_t a0 = rowptr[v];
            s_id[i] = v; s_parent[i] = i; s_r0[i] = a0; s_deg[i] = (int32_t)(rowptr[v + 1] - a0);
        }
        __syncthreads();
        // the n (n - 1) / 2 pairs i < j, 64 per round (j-major: pair p = j (j - 1) / 2 + i); the loop ends as soon as n - 1
        // unions have succeeded -- the subgraph is one component and no further pair can change a label (a BFS subgraph
        // of 20 nodes is connected after about half of its 190 pairs)
        const int64_t npairs = (int64_t)n * (n - 1) / 2;
        int unions = 0;
        for (int64_t p0 = 0; p0 < npairs && unions < n - 1; p0 += 64) {
            const int64_t p = p0 + lane;
            bool merged = false;
            if (p < npairs) {
                int j = (int)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
                while ((int64_t)j * (j - 1) / 2 > p) --j;
                while ((int64_t)(j + 1) * j / 2 <= p) ++j;
                const int i = (int)(p - (int64_t)j * (j - 1) / 2);
                const int32_t a = s_id[i], b = s_id[j];
                bool linked = (a == b);
                if (!linked) {
                    if (s_deg[i] <= s_deg[j]) linked = sgnn_sorted_contains(col_sorted + s_r0[i], s_deg[i], b);
                    else linked = sgnn_sorted_contains(col_sorted + s_r0[j], s_deg[j], a);
                }
                if (linked) {
                    int x = i, y = j;
                    while (true) {
                        x = cc_find(s_parent, x);
                        y = cc_find(s_parent, y);
                        if (x == y) break;
                        if (x < y) { const int t = x; x = y; y = t; }       // hook x (larger) under y
                        const int32_t old = atomicCAS(&s_parent[x], x, y);
                        if (old == x) { merged = true; break; }
                    }
                }
            }
            unions += __popcll(__ballot(merged));           // every successful hook removes one component
        }
        __syncthreads();
        for (int i = lane; i < n; i += 64) out_label[beg + i] = cc_find(s_parent, i);
        __syncthreads();
    }
}

extern "C" int sgnn_cc_labels(const int64_t* rowptr, const int32_t* col_sorted, int64_t nnz,
                              const int64_t* sub_ptr, const int32_t* sub_nodes, int64_t n_subgraphs,
                              int64_t max_sub_len, int32_t* out_label, void* stream)
{
    if (!rowptr || !col_sorted || !sub_ptr || !sub_nodes || !out_label || n_subgraphs < 0) return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (n_subgraphs == 0) return SGNN_OK;
    const int grid = (int)(n_subgraphs < 256 * 32 ? n_subgraphs : 256 * 32);
    hipLaunchKernelGGL(cc_labels_kernel<64>, dim3(grid), dim3(64), 0, (hipStream_t)stream, rowptr, col_sorted,
                       sub_ptr, sub_nodes, n_subgraphs, out_label);
    if (max_sub_len <= 0 || max_sub_len > 64)
        hipLaunchKernelGGL(cc_labels_kernel<CC_MAX>, dim3(grid < 2048 ? grid : 2048), dim3(64), 0, (hipStream_t)stream,
                           rowptr, col_sorted, sub_ptr, sub_nodes, n_subgraphs, out_label);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// a7 (second half)  labels -> the padded (S, C, L) component tensor of SubGNN.initialize_cc_ids
// (SubGNN/SubGNN.py:575-607) in canonical order: components by the position of their first node,
// nodes in subgraph order, duplicates of a node dropped (first position kept).
// One wavefront per subgraph, two launches of the same code: a statistics pass (number of
// components, longest component -> the host takes the maxima for C and L) and the write pass.
// Rank of a component = roots at smaller positions (ballot + popcount prefix); position inside a
// component = kept nodes of the same label before me (64 broadcast steps per 64-node chunk plus a
// running per-label counter in LDS).  Subgraphs of up to 64 nodes need no hash; longer ones find
// duplicates through an LDS hash keyed by node id that keeps the smallest position.
// ---------------------------------------------------------------------------------------------
#define CCK_HASH 4096

template <bool BIG, bool WRITE>
__global__ __launch_bounds__(64) void cc_compact_kernel(
    const int64_t* __restrict__ sub_ptr, const int32_t* __restrict__ sub_nodes, const int32_t* __restrict__ labels,
    int64_t n_sub, int64_t C, int64_t L, int32_t* __restrict__ out_ncc, int32_t* __restrict__ out_maxlen,
    int64_t* __restrict__ out)
{
    constexpr int NMAX = BIG ? CC_MAX : 64;
    __shared__ int32_t s_rank[NMAX], s_cnt[NMAX];
    __shared__ int32_t s_hk[BIG ? CCK_HASH : 1], s_hv[BIG ? CCK_HASH : 1];
    const int lane = threadIdx.x;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int64_t s = blockIdx.x; s < n_sub; s += gridDim.x) {
        const int64_t beg = sub_ptr[s];
        const int n = (int)(sub_ptr[s + 1] - beg);
        if (n <= 0 || n > NMAX || (BIG && n <= 64)) {
            if (!WRITE && n <= 0) { if (lane == 0) { out_ncc[s] = 0; out_maxlen[s] = 0; } }
            continue;                                           // the other instantiation owns it
        }
        for (int i = lane; i < n; i += 64) s_cnt[i] = 0;
        if (BIG) {
            for (int i = lane; i < CCK_HASH; i += 64) { s_hk[i] = 0; s_hv[i] = 0x7fffffff; }
            __syncthreads();
            for (int i = lane; i < n; i += 64) {
                const int32_t v = sub_nodes[beg + i];
                uint32_t h = sgnn_hash32((uint32_t)v) >> 20;
                while (true) {
                    const int32_t old = atomicCAS(&s_hk[h], 0, v);
                    if (old == 0 || old == v) { atomicMin(&s_hv[h], i); break; }
                    h = (h + 1) & (CCK_HASH - 1);
                }
            }
        }
        __syncthreads();
        int running = 0, maxlen = 0;
        // roots and their ranks
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            int32_t v = 0, lab = -1;
            if (i < n) { v = sub_nodes[beg + i]; lab = labels[beg + i]; }
            const bool root = (i < n) && lab == i;              // a root is the first position of its node
            const uint64_t m = __ballot(root);
            if (root) s_rank[i] = running + __popcll(m & lt);
            running += __popcll(m);
        }
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int i = c0 + lane;
            int32_t v = 0, lab = -1;
            if (i < n) { v = sub_nodes[beg + i]; lab = labels[beg + i]; }
            bool keep = i < n;
            if (BIG) {
                if (keep) {
                    uint32_t h = sgnn_hash32((uint32_t)v) >> 20;
                    while (s_hk[h] != v) h = (h + 1) & (CCK_HASH - 1);
                    keep = (s_hv[h] == i);
                }
            } else {
                for (int l = 0; l < n; ++l) { const int32_t vl = __shfl(v, l); if (l < lane && vl == v) keep = false; }
            }
            const int32_t my = keep ? lab : -1 - lane;          // distinct sentinels for dropped lanes
            int before = 0;
            bool later = false;
            const int lim = (n - c0) < 64 ? (n - c0) : 64;
            for (int l = 0; l < lim; ++l) {
                const int32_t o = __shfl(my, l);
                before += (o == my && l < lane) ? 1 : 0;
                later = later || (o == my && l > lane);
            }
            int within = 0;
            if (keep) within = s_cnt[lab] + before;
            if (keep && !later) s_cnt[lab] = within + 1;         // last of its component in this chunk
            if (keep) {
                maxlen = within + 1 > maxlen ? within + 1 : maxlen;
                if (WRITE) out[((int64_t)s * C + s_rank[lab]) * L + within] = (int64_t)v;
            }
        }
        if (!WRITE) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(maxlen, d); maxlen = o > maxlen ? o : maxlen; }
            if (lane == 0) { out_ncc[s] = running; out_maxlen[s] = maxlen; }
        }
        __syncthreads();
    }
}

static int cc_compact_launch(bool write, const int64_t* sub_ptr, const int32_t* sub_nodes, const int32_t* labels,
                             int64_t n_sub, int64_t max_sub_len, int64_t C, int64_t L, int32_t* out_ncc,
                             int32_t* out_maxlen, int64_t* out, void* stream)
{
    if (!sub_ptr || !sub_nodes || !labels || n_sub < 0) return SGNN_ERR_BAD_ARG;
    if (n_sub == 0) return SGNN_OK;                             // (subgraphs of more than CC_MAX nodes: sgnn_cc_compact_huge)
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(n_sub < 256 * 64 ? n_sub : 256 * 64);
    if (write) {
        hipLaunchKernelGGL((cc_compact_kernel<false, true>), dim3(grid), dim3(64), 0, st, sub_ptr, sub_nodes, labels, n_sub,
                           C, L, out_ncc, out_maxlen, out);
        if (max_sub_len > 64 || max_sub_len <= 0)
            hipLaunchKernelGGL((cc_compact_kernel<true, true>), dim3(grid < 2048 ? grid : 2048), dim3(64), 0, st, sub_ptr,
                               sub_nodes, labels, n_sub, C, L, out_ncc, out_maxlen, out);
    } else {
        hipLaunchKernelGGL((cc_compact_kernel<false, false>), dim3(grid), dim3(64), 0, st, sub_ptr, sub_nodes, labels,
                           n_sub, C, L, out_ncc, out_maxlen, out);
        if (max_sub_len > 64 || max_sub_len <= 0)
            hipLaunchKernelGGL((cc_compact_kernel<true, false>), dim3(grid < 2048 ? grid : 2048), dim3(64), 0, st, sub_ptr,
                               sub_nodes, labels, n_sub, C, L, out_ncc, out_maxlen, out);
    }
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_cc_compact_stats(const int64_t* sub_ptr, const int32_t* sub_nodes, const int32_t* labels,
                                     int64_t n_subgraphs, int64_t max_sub_len, int32_t* out_n_components,
                                     int32_t* out_longest, void* stream)
{
    if (!out_n_components || !out_longest) return SGNN_ERR_BAD_ARG;
    return cc_compact_launch(false, sub_ptr, sub_nodes, labels, n_subgraphs, max_sub_len, 0, 0, out_n_components,
                             out_longest, nullptr, stream);
}

extern "C" int sgnn_cc_compact(const int64_t* sub_ptr, const int32_t* sub_nodes, const int32_t* labels,
                               int64_t n_subgraphs, int64_t max_sub_len, int64_t C, int64_t L, int64_t* out, void* stream)
{
    if (!out || C < 1 || L < 1) return SGNN_ERR_BAD_ARG;
    return cc_compact_launch(true, sub_ptr, sub_nodes, labels, n_subgraphs, max_sub_len, C, L, nullptr, nullptr, out,
                             stream);
}

// ---------------------------------------------------------------------------------------------
// a7 for subgraphs of MORE than CC_MAX nodes (round 3: rounds 1-2 refused them; the reference pads to any size,
// SubGNN/SubGNN.py:575-607).  The LDS tables of the kernels above do not hold such a set and their all-pairs test is
// quadratic, so these take another route, one 256-thread workgroup per subgraph with its state in a caller workspace:
// node id -> smallest position in an open-addressing table (2-4 slots per node), union-find parents in global memory,
// and the induced edges found by streaming the members' neighbour lists against the table (work ~ sum of degrees).
// Compaction: component rank = roots at smaller positions (a running workgroup scan), position inside a component =
// members of the same component before me, counted chunk by chunk of 256 positions in subgraph order.
// Layout of the workspace for a call over `total` nodes (sub_ptr[n_sub]): hkey[4 total] | hpos[4 total] | parent[total] |
// rank[total] | cnt[total]  (int32 each); subgraph s uses the slices at its own offset, so no two workgroups meet.
// ---------------------------------------------------------------------------------------------
#define CCH_THREADS 256

extern "C" int64_t sgnn_cc_huge_workspace_bytes(int64_t total_nodes)
{
    return (total_nodes < 0 ? 0 : total_nodes) * 11 * 4 + 64;
}

struct CchRegion { int32_t* hk; int32_t* hv; int32_t* par; int32_t* rank; int32_t* cnt; uint32_t H; };

__device__ static inline CchRegion cch_region(int32_t* ws, int64_t total, int64_t beg, int n)
{
    CchRegion r;
    r.hk = ws + 4 * beg;
    r.hv = ws + 4 * total + 4 * beg;
    r.par = ws + 8 * total + beg;
    r.rank = ws + 9 * total + beg;
    r.cnt = ws + 10 * total + beg;
    uint32_t H = 1;
    while (H < 2u * (uint32_t)n) H <<= 1;                      // <= 4 n
    r.H = H;
    return r;
}

__device__ static inline void cch_build_table(const CchRegion& r, const int32_t* __restrict__ nodes, int n)
{
    for (uint32_t i = threadIdx.x; i < r.H; i += CCH_THREADS) { r.hk[i] = 0; r.hv[i] = 0x7fffffff; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += CCH_THREADS) {
        const int32_t v = nodes[i];
        uint32_t h = sgnn_hash32((uint32_t)v) & (r.H - 1);
        while (true) {
            const int32_t old = atomicCAS(&r.hk[h], 0, v);
            if (old == 0 || old == v) { atomicMin(&r.hv[h], i); break; }
            h = (h + 1) & (r.H - 1);
        }
    }
    __syncthreads();
}

__device__ static inline int cch_lookup(const CchRegion& r, int32_t v)     // smallest position of id v in the set, -1 = absent
{
    uint32_t h = sgnn_hash32((uint32_t)v) & (r.H - 1);
    while (true) {
        const int32_t k = r.hk[h];
        if (k == v) return r.hv[h];
        if (k == 0) return -1;
        h = (h + 1) & (r.H - 1);
    }
}

__device__ static inline void cch_union(int32_t* par, int x, int y)
{
    while (true) {
        x = cc_find(par, x);
        y = cc_find(par, y);
        if (x == y) return;
        if (x < y) { const int t = x; x = y; y = t; }           // hook the larger root under the smaller: root = smallest position
        if (atomicCAS(&par[x], x, y) == x) return;
    }
}

__global__ __launch_bounds__(CCH_THREADS) void cc_huge_labels_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int64_t* __restrict__ sub_ptr,
    const int32_t* __restrict__ sub_nodes, int64_t n_sub, int32_t* __restrict__ out_label, int32_t* __restrict__ ws)
{
    const int64_t total = sub_ptr[n_sub];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t s = blockIdx.x; s < n_sub; s += gridDim.x) {
        const int64_t beg = sub_ptr[s];
        const int n = (int)(sub_ptr[s + 1] - beg);
        if (n <= CC_MAX) continue;                              // sgnn_cc_labels owns those
        const CchRegion r = cch_region(ws, total, beg, n);
        const int32_t* nodes = sub_nodes + beg;
        for (int i = threadIdx.x; i < n; i += CCH_THREADS) r.par[i] = i;
        cch_build_table(r, nodes, n);
        // a member's list by one wavefront: every neighbour that is in the set joins the member's component;
        // a repeated id joins its first occurrence
        for (int i = wave; i < n; i += CCH_THREADS / 64) {
            const int32_t v = nodes[i];
            const int first = cch_lookup(r, v);
            if (first != i) { if (lane == 0) cch_union(r.par, i, first); continue; }
            const int64_t a = rowptr[v], b = rowptr[v + 1];
            for (int64_t e = a + lane; e < b; e += 64) {
                const int j = cch_lookup(r, col[e]);
                if (j >= 0 && j != i) cch_union(r.par, i, j);
            }
        }
        __threadfence_block();
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += CCH_THREADS) out_label[beg + i] = cc_find(r.par, i);
        __syncthreads();
    }
}

template <bool WRITE>
__global__ __launch_bounds__(CCH_THREADS) void cc_huge_compact_kernel(
    const int64_t* __restrict__ sub_ptr, const int32_t* __restrict__ sub_nodes, const int32_t* __restrict__ labels,
    int64_t n_sub, int64_t C, int64_t L, int32_t* __restrict__ out_ncc, int32_t* __restrict__ out_maxlen,
    int64_t* __restrict__ out, int32_t* __restrict__ ws)
{
    __shared__ int32_t s_scan[CCH_THREADS];
    __shared__ int32_t s_my[CCH_THREADS];
    __shared__ int32_t s_red[CCH_THREADS];
    const int64_t total = sub_ptr[n_sub];
    const int tid = threadIdx.x;
    for (int64_t s = blockIdx.x; s < n_sub; s += gridDim.x) {
        const int64_t beg = sub_ptr[s];
        const int n = (int)(sub_ptr[s + 1] - beg);
        if (n <= CC_MAX) continue;
        const CchRegion r = cch_region(ws, total, beg, n);
        const int32_t* nodes = sub_nodes + beg;
        const int32_t* lab = labels + beg;
        for (int i = tid; i < n; i += CCH_THREADS) r.cnt[i] = 0;
        cch_build_table(r, nodes, n);
        // ranks of the roots: a running scan over the positions, 256 at a time
        int running = 0;
        for (int c0 = 0; c0 < n; c0 += CCH_THREADS) {
            const int i = c0 + tid;
            const int root = (i < n && lab[i] == i) ? 1 : 0;
            s_scan[tid] = root;
            __syncthreads();
            for (int d = 1; d < CCH_THREADS; d <<= 1) {
                const int t = tid >= d ? s_scan[tid - d] : 0;
                __syncthreads();
                s_scan[tid] += t;
                __syncthreads();
            }
            if (root) r.rank[i] = running + s_scan[tid] - 1;
            running += s_scan[CCH_THREADS - 1];
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        // positions inside the components, in subgraph order
        int maxlen = 0;
        for (int c0 = 0; c0 < n; c0 += CCH_THREADS) {
            const int i = c0 + tid;
            int32_t v = 0, lb = -1;
            bool keep = i < n;
            if (keep) { v = nodes[i]; lb = lab[i]; keep = cch_lookup(r, v) == i; }       // repeated ids: first occurrence only
            s_my[tid] = keep ? lb : -1 - tid;                   // distinct sentinels for dropped positions
            __syncthreads();
            int before = 0;
            bool later = false;
            const int lim = (n - c0) < CCH_THREADS ? (n - c0) : CCH_THREADS;
            if (keep) {
                for (int l = 0; l < lim; ++l) {
                    const int32_t o = s_my[l];
                    before += (o == lb && l < tid) ? 1 : 0;
                    later = later || (o == lb && l > tid);
                }
            }
            int within = 0;
            if (keep) within = r.cnt[lb] + before;
            __syncthreads();
            if (keep && !later) r.cnt[lb] = within + 1;          // the component's last member of this chunk
            if (keep) {
                maxlen = within + 1 > maxlen ? within + 1 : maxlen;
                if (WRITE) out[((int64_t)s * C + r.rank[lb]) * L + within] = (int64_t)v;
            }
            __threadfence_block();
            __syncthreads();
        }
        if (!WRITE) {
            s_red[tid] = maxlen;
            __syncthreads();
            for (int d = CCH_THREADS / 2; d >= 1; d >>= 1) {
                if (tid < d) s_red[tid] = s_red[tid] > s_red[tid + d] ? s_red[tid] : s_red[tid + d];
                __syncthreads();
            }
            if (tid == 0) { out_ncc[s] = running; out_maxlen[s] = s_red[0]; }
        }
        __syncthreads();
    }
}

extern "C" int sgnn_cc_labels_huge(const int64_t* rowptr, const int32_t* col, int64_t nnz, const int64_t* sub_ptr,
                                   const int32_t* sub_nodes, int64_t n_subgraphs, int64_t total_nodes, int32_t* out_label,
                                   void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!rowptr || !col || !sub_ptr || !sub_nodes || !out_label || !workspace || n_subgraphs < 0 || total_nodes < 0)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (total_nodes >= (1ll << 28)) return SGNN_ERR_SET_TOO_LARGE;          // 4 x total must index with 32 bits
    if (workspace_bytes < sgnn_cc_huge_workspace_bytes(total_nodes)) return SGNN_ERR_BAD_ARG;
    if (n_subgraphs == 0) return SGNN_OK;
    hipLaunchKernelGGL(cc_huge_labels_kernel, dim3((int)(n_subgraphs < 1024 ? n_subgraphs : 1024)), dim3(CCH_THREADS), 0,
                       (hipStream_t)stream, rowptr, col, sub_ptr, sub_nodes, n_subgraphs, out_label, (int32_t*)workspace);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_cc_compact_huge(const int64_t* sub_ptr, const int32_t* sub_nodes, const int32_t* labels,
                                    int64_t n_subgraphs, int64_t total_nodes, int write, int64_t C, int64_t L,
                                    int32_t* out_n_components, int32_t* out_longest, int64_t* out, void* workspace,
                                    int64_t workspace_bytes, void* stream)
{
    if (!sub_ptr || !sub_nodes || !labels || !workspace || n_subgraphs < 0 || total_nodes < 0) return SGNN_ERR_BAD_ARG;
    if (write ? (!out || C < 1 || L < 1) : (!out_n_components || !out_longest)) return SGNN_ERR_BAD_ARG;
    if (total_nodes >= (1ll << 28)) return SGNN_ERR_SET_TOO_LARGE;
    if (workspace_bytes < sgnn_cc_huge_workspace_bytes(total_nodes)) return SGNN_ERR_BAD_ARG;
    if (n_subgraphs == 0) return SGNN_OK;
    const int grid = (int)(n_subgraphs < 1024 ? n_subgraphs : 1024);
    if (write)
        hipLaunchKernelGGL((cc_huge_compact_kernel<true>), dim3(grid), dim3(CCH_THREADS), 0, (hipStream_t)stream, sub_ptr,
                           sub_nodes, labels, n_subgraphs, C, L, out_n_components, out_longest, out, (int32_t*)workspace);
    else
        hipLaunchKernelGGL((cc_huge_compact_kernel<false>), dim3(grid), dim3(CCH_THREADS), 0, (hipStream_t)stream, sub_ptr,
                           sub_nodes, labels, n_subgraphs, C, L, out_n_components, out_longest, out, (int32_t*)workspace);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// a8  k-hop border (reference SubGNN/subgraph_utils.py:146-176), optionally fused with the
// neighbourhood-border anchor draw (a4, reference anchor_patch_samplers.py:184-194).
// One workgroup per component, level-synchronous BFS.  The visited bitmap over node ids lives in
// LDS when it fits (max_id + 1 <= ~1.27 M bits in the CU's 160 KB: one 1024-thread workgroup per
// CU, LDS atomics) and otherwise in the caller's workspace (256-thread workgroups, L2 atomics);
// the BFS queue is always in the workspace.  A frontier node is taken by one wavefront whose
// lanes stream its neighbour list (coalesced), claim unseen neighbours with atomicOr on the
// bitmap word and append them to the queue.  Every bit set is un-set on exit so the next
// component handled by the workgroup starts clean.
// Fused sampling (neighbourhood-anchor law, common.h): a slot needs "the k-th smallest border id".
// Once the members' bits are un-set the bitmap IS the border in ascending order, so the workgroup
// popcounts it once (each thread a contiguous run of words, odd stride = no LDS bank conflicts),
// prefix-sums the runs, and every slot is answered by one thread with a binary search over the
// run prefixes plus a walk through one run -- the border is never materialised or sorted, and
// a slot costs O(log) instead of one hash per border node.
// ---------------------------------------------------------------------------------------------
#define KB_THREADS_G 256
#define KB_THREADS_L 1024
#define KB_MAX_WG_G 1024
#define KB_MAX_WG_L 256
#define KB_LDS_BYTES (150 * 1024)      // bitmap; the rest of the 160 KB holds the rank tables of the fused draw
#ifndef KB_BIG
#define KB_BIG 1024                     // list length from which a frontier node's list is streamed on its own
#endif
#ifndef KB_INFLIGHT
#define KB_INFLIGHT 2                    // 64-edge chunks a wavefront has in flight during the expansion
#endif
#define KB_SEL_CHUNK 256                // slots answered per pass (selected ids staged in LDS for the hop lookup)

static inline int64_t kb_words(int64_t max_id) { return (max_id + 32) / 32; }
static inline bool kb_fits_lds(int64_t max_id) { return kb_words(max_id) * 4 <= KB_LDS_BYTES; }
static inline int64_t kb_n_wg(int64_t n_sets, bool lds) {
    const int64_t cap = lds ? KB_MAX_WG_L : KB_MAX_WG_G;
    return n_sets < cap ? (n_sets < 1 ? 1 : n_sets) : cap;
}

// ---------------------------------------------------------------------------------------------
// canonical order inside every set (ascending ids; equal ids keep their relative order): a rank
// sort per set.  A group of G lanes owns a set: the ids are staged in LDS, every element counts
// the elements that sort before it (broadcast LDS reads) and is written to that slot.  Sets are
// the components / border sets of single subgraphs (tens of ids; up to SORT_SETS_MAX with G = 64,
// O(n^2 / 64) per wave) -- one launch instead of a device-wide sort of (set, id) keys.
// ---------------------------------------------------------------------------------------------
#define SORT_SETS_MAX 1024
template <int G>
__global__ __launch_bounds__(256) void sort_sets_kernel(const int64_t* __restrict__ set_ptr,
                                                        const int32_t* __restrict__ nodes, int64_t n_sets, int cap,
                                                        int32_t* __restrict__ out_nodes, int32_t* __restrict__ out_pos)
{
    extern __shared__ int32_t s_keys[];
    const int sub = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    int32_t* keys = s_keys + (int64_t)grp * cap;
    const int64_t n_groups = (int64_t)gridDim.x * (256 / G);
    // the trip count is made uniform over the wave (groups of one wave sit on different sets)
    const int64_t first = (int64_t)blockIdx.x * (256 / G) + grp;
    const int64_t first_of_wave = (int64_t)blockIdx.x * (256 / G) + (threadIdx.x / 64) * (64 / G);
    for (int64_t base = first_of_wave, s = first; base < n_sets; base += n_groups, s += n_groups) {
        const bool live = s < n_sets;
        const int64_t b = live ? set_ptr[s] : 0;
        const int len = live ? (int)(set_ptr[s + 1] - b) : 0;
        for (int i = sub; i < len; i += G) keys[i] = nodes[b + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = sub; i < len; i += G) {
            const int32_t k = keys[i];
            int rank = 0;
            for (int j = 0; j < len; ++j) {
                const int32_t o = keys[j];
                rank += (o < k || (o == k && j < i)) ? 1 : 0;
            }
            out_nodes[b + rank] = k;
            if (out_pos) out_pos[b + rank] = (int32_t)(b + i);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- padded id rows -> ragged sets (PAD / masked entries stripped, order kept) ---------------------------------------------
// Every stage takes its node sets as CSR-style (ptr, nodes); the reference-shaped tensors are padded (rows, L) int64
// matrices (component ids S.py:575-607, structure patches aps:210-243, node views aps:131).  Two passes around the caller's
// prefix sum: counts per row, then the kept entries written behind the row's offset.  (As torch ops this was thirteen
// launches per conversion -- a triangular GEMM for the ranks, where, scatter ... -- four conversions per pass.)
__global__ __launch_bounds__(256) void pack_rows_count_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ mask,
                                                              int64_t n, int64_t L, int64_t* __restrict__ counts)
{
    const int64_t r = blockIdx.x * 256ll + threadIdx.x;
    if (r >= n) return;
    int64_t c = 0;
    for (int64_t j = 0; j < L; ++j) c += mask ? (mask[r * L + j] != 0) : (ids[r * L + j] != 0);
    counts[r] = c;
}

__global__ __launch_bounds__(256) void pack_rows_write_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ mask,
                                                              int64_t n, int64_t L, const int64_t* __restrict__ ptr,
                                                              int32_t* __restrict__ nodes)
{
    const int64_t r = blockIdx.x * 256ll + threadIdx.x;
    if (r >= n) return;
    int64_t o = ptr[r];
    for (int64_t j = 0; j < L; ++j) {
        const int64_t v = ids[r * L + j];
        if (mask ? (mask[r * L + j] != 0) : (v != 0)) nodes[o++] = (int32_t)v;
    }
}

// keep[r, i] = ids[r, i] is not PAD and no earlier entry of row r holds the same id: the node view of a patch (the unique nodes
// of a walk in first-occurrence order, anchor_patch_samplers.py:131-138) as a mask for sgnn_pack_rows_*
__global__ __launch_bounds__(256) void first_occurrence_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t L,
                                                               uint8_t* __restrict__ keep)
{
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= n * L) return;
    const int64_t r = t / L, i = t - r * L;
    const int64_t v = ids[t];
    bool k = v != 0;
    for (int64_t j = 0; k && j < i; ++j) k = ids[r * L + j] != v;
    keep[t] = k ? 1 : 0;
}

extern "C" int sgnn_first_occurrence_mask(const int64_t* ids, int64_t n_rows, int64_t row_len, uint8_t* keep, void* stream)
{
    if (!ids || !keep || n_rows < 0 || row_len < 0) return SGNN_ERR_BAD_ARG;
    const int64_t total = n_rows * row_len;
    if (total == 0) return SGNN_OK;
    if ((total + 255) / 256 > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(first_occurrence_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, n_rows,
                       row_len, keep);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// the flagged entries of every ragged set, order kept (a patch's in-border nodes among its view nodes,
// subgraph_utils.py:126-144): counts, then -- around the caller's prefix sum -- the packed write
__global__ __launch_bounds__(256) void filter_sets_kernel(const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes,
                                                          const uint8_t* __restrict__ flags, int64_t n_sets,
                                                          int64_t* __restrict__ counts, const int64_t* __restrict__ out_ptr,
                                                          int32_t* __restrict__ out_nodes)
{
    const int64_t s = blockIdx.x * 256ll + threadIdx.x;
    if (s >= n_sets) return;
    const int64_t b = set_ptr[s], e = set_ptr[s + 1];
    if (counts) {
        int64_t c = 0;
        for (int64_t i = b; i < e; ++i) c += flags[i] != 0;
        counts[s] = c;
    } else {
        int64_t o = out_ptr[s];
        for (int64_t i = b; i < e; ++i)
            if (flags[i]) out_nodes[o++] = set_nodes[i];
    }
}

extern "C" int sgnn_filter_sets(const int64_t* set_ptr, const int32_t* set_nodes, const uint8_t* flags, int64_t n_sets,
                                int64_t* counts, const int64_t* out_ptr, int32_t* out_nodes, void* stream)
{
    if (!set_ptr || !set_nodes || !flags || n_sets < 0 || (!counts && (!out_ptr || !out_nodes))) return SGNN_ERR_BAD_ARG;
    if (n_sets == 0) return SGNN_OK;
    hipLaunchKernelGGL(filter_sets_kernel, dim3((unsigned)((n_sets + 255) / 256)), dim3(256), 0, (hipStream_t)stream, set_ptr, set_nodes,
                       flags, n_sets, counts, out_ptr, out_nodes);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---- the same packing in ONE launch for small inputs (round 4) -------------------------------------------------------
// A few hundred structure patches are packed three times per pass (node views, in-border sets, degree-sequence sets): as
// count + prefix sum + write that is 5-6 launches each (two fills, a count, one or two scan kernels, a write) for microseconds
// of work -- at shard size the pass is bound by the host's launches.  One 1024-thread workgroup does all of it: every thread
// owns up to SGNN_PACK_FUSED_RPT consecutive rows; the entries and their keep flags are staged in LDS (see below), the block
// scans the per-thread totals, rows are written from LDS and the arena's tail is zeroed.
// mode 0: keep non-PAD entries; 1: keep where mask != 0; 2: keep non-PAD entries that no earlier entry of the row repeats
// (the node view of a patch: first_occurrence_kernel + pack in one).
#define SGNN_PACK_FUSED_THREADS 1024
#define SGNN_PACK_FUSED_RPT 8
#define SGNN_PACK_FUSED_MAX_ROWS (SGNN_PACK_FUSED_THREADS * SGNN_PACK_FUSED_RPT)

__device__ __forceinline__ int64_t pack_block_excl_scan(int64_t c, int64_t* s_wave, int64_t* total)
{
    // exclusive prefix of c over the 1024 threads: wave scans (shuffles) + the 16 wave totals through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int64_t before = 0, all = 0;
    for (int w = 0; w < SGNN_PACK_FUSED_THREADS / 64; ++w) {
        const int64_t t = s_wave[w];
        if (w < wave) before += t;
        all += t;
    }
    *total = all;
    return before + inc - c;
}

// LDS staging (round 4, second version: the first one let a thread walk its rows out of global memory -- 1250 dependent
// compares per 50-entry row for the first-occurrence mode: 100+ us on one workgroup, slower than the launches it replaced):
// all entries are loaded coalesced into LDS (ids as int32, one keep byte each), the keep flags are computed one THREAD PER
// ENTRY against the LDS copy (no early exit: independent reads), then the row owners count, the block scans, and the rows
// are written from LDS.  Entries are limited by the LDS (SGNN_PACK_FUSED_MAX_ENTRIES).
#define SGNN_PACK_FUSED_MAX_ENTRIES 24576

__global__ __launch_bounds__(SGNN_PACK_FUSED_THREADS) void pack_rows_fused_kernel(
    const int64_t* __restrict__ ids, const uint8_t* __restrict__ mask, int mode, int64_t n, int64_t L,
    int64_t* __restrict__ ptr, int32_t* __restrict__ nodes)
{
    extern __shared__ int32_t s_pack[];                      // n * L ids, then n * L keep bytes
    __shared__ int64_t s_wave[SGNN_PACK_FUSED_THREADS / 64];
    const int total_e = (int)(n * L);
    int32_t* s_ids = s_pack;
    uint8_t* s_keep = reinterpret_cast<uint8_t*>(s_pack + total_e);
    for (int t = threadIdx.x; t < total_e; t += SGNN_PACK_FUSED_THREADS) s_ids[t] = (int32_t)ids[t];
    __syncthreads();
    const int Li = (int)L;
    for (int t = threadIdx.x; t < total_e; t += SGNN_PACK_FUSED_THREADS) {
        const int32_t v = s_ids[t];
        bool k;
        if (mode == 1) k = mask[t] != 0;
        else {
            k = v != 0;
            if (mode == 2 && k) {
                const int r = t / Li, i = t - r * Li;
                bool dup = false;
                for (int q = 0; q < i; ++q) dup |= s_ids[r * Li + q] == v;
                k = !dup;
            }
        }
        s_keep[t] = k ? 1 : 0;
    }
    __syncthreads();
    const int64_t rpt = (n + SGNN_PACK_FUSED_THREADS - 1) / SGNN_PACK_FUSED_THREADS;      // <= SGNN_PACK_FUSED_RPT (host-checked)
    const int64_t r0 = threadIdx.x * rpt;
    int64_t c = 0;
    for (int64_t k = 0; k < rpt; ++k) {
        const int64_t r = r0 + k;
        if (r < n)
            for (int j = 0; j < Li; ++j) c += s_keep[r * Li + j];
    }
    int64_t total;
    int64_t o = pack_block_excl_scan(c, s_wave, &total);
    for (int64_t k = 0; k < rpt; ++k) {
        const int64_t r = r0 + k;
        if (r < n) {
            ptr[r] = o;
            for (int j = 0; j < Li; ++j)
                if (s_keep[r * Li + j]) nodes[o++] = s_ids[r * Li + j];
        }
    }
    if (threadIdx.x == 0) ptr[n] = total;
    for (int64_t i = total + threadIdx.x; i <= n * L; i += SGNN_PACK_FUSED_THREADS) nodes[i] = 0;     // the arena's tail (and spare slot)
}

__global__ __launch_bounds__(SGNN_PACK_FUSED_THREADS) void filter_sets_fused_kernel(
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, const uint8_t* __restrict__ flags, int64_t n,
    int64_t arena, int64_t* __restrict__ out_ptr, int32_t* __restrict__ out_nodes)
{
    extern __shared__ int32_t s_pack[];                      // the sets' entries, then their flag bytes
    __shared__ int64_t s_wave[SGNN_PACK_FUSED_THREADS / 64];
    const int64_t base = set_ptr[0];
    const int total_e = (int)(set_ptr[n] - base);            // <= SGNN_PACK_FUSED_MAX_ENTRIES (host: the arena's size bounds it)
    int32_t* s_ids = s_pack;
    uint8_t* s_keep = reinterpret_cast<uint8_t*>(s_pack + total_e);
    for (int t = threadIdx.x; t < total_e; t += SGNN_PACK_FUSED_THREADS) { s_ids[t] = set_nodes[base + t]; s_keep[t] = flags[base + t] != 0; }
    __syncthreads();
    const int64_t rpt = (n + SGNN_PACK_FUSED_THREADS - 1) / SGNN_PACK_FUSED_THREADS;
    const int64_t r0 = threadIdx.x * rpt;
    int64_t c = 0;
    for (int64_t k = 0; k < rpt; ++k) {
        const int64_t r = r0 + k;
        if (r < n)
            for (int64_t i = set_ptr[r] - base; i < set_ptr[r + 1] - base; ++i) c += s_keep[i];
    }
    int64_t total;
    int64_t o = pack_block_excl_scan(c, s_wave, &total);
    for (int64_t k = 0; k < rpt; ++k) {
        const int64_t r = r0 + k;
        if (r < n) {
            out_ptr[r] = o;
            for (int64_t i = set_ptr[r] - base; i < set_ptr[r + 1] - base; ++i)
                if (s_keep[i]) out_nodes[o++] = s_ids[i];
        }
    }
    if (threadIdx.x == 0) out_ptr[n] = total;
    for (int64_t i = total + threadIdx.x; i < arena; i += SGNN_PACK_FUSED_THREADS) out_nodes[i] = 0;
}

// Entries one fused launch can stage on the CURRENT device: 5 bytes of LDS each, capped by the compile-time figure (sized for
// gfx950's 160 KB).  Per device, not per process: a part with less LDS advertises less and its callers take the multi-launch path.
static int64_t pack_fused_entries_limit()
{
    static int64_t lim[64];                                  // 0 = not asked yet (a racing first call writes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SGNN_PACK_FUSED_MAX_ENTRIES;   // no device here: the compile-time figure
    if (lim[dev] == 0) {
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds <= 0) lds = 64 * 1024;
        const int64_t e = ((int64_t)lds - 16) / 5;
        lim[dev] = e < SGNN_PACK_FUSED_MAX_ENTRIES ? (e > 0 ? e : 1) : SGNN_PACK_FUSED_MAX_ENTRIES;
    }
    return lim[dev];
}
// the opt-in to more than 64 KB of dynamic LDS, once per kernel AND device; its failure is reported, not discarded
static int pack_fused_opt_in(const void* kernel, bool* done /* [64] */, size_t bytes)
{
    if (bytes <= 64 * 1024) return SGNN_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SGNN_ERR_LAUNCH;
    if (!done[dev]) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(pack_fused_entries_limit() * 5 + 16)) != hipSuccess) {
            (void)hipGetLastError();
            return SGNN_ERR_SET_TOO_LARGE;                   // the caller's multi-launch path serves this input
        }
        done[dev] = true;
    }
    return SGNN_OK;
}

extern "C" int64_t sgnn_pack_fused_max_rows(void) { return SGNN_PACK_FUSED_MAX_ROWS; }
extern "C" int64_t sgnn_pack_fused_max_entries(void) { return pack_fused_entries_limit(); }

extern "C" int sgnn_pack_rows_fused(const int64_t* ids, const uint8_t* mask, int mode, int64_t n_rows, int64_t row_len,
                                    int64_t* ptr, int32_t* nodes, void* stream)
{
    if (!ids || !ptr || !nodes || n_rows < 1 || row_len < 1 || mode < 0 || mode > 2 || (mode == 1 && !mask)) return SGNN_ERR_BAD_ARG;
    if (n_rows > SGNN_PACK_FUSED_MAX_ROWS || n_rows * row_len > pack_fused_entries_limit()) return SGNN_ERR_SET_TOO_LARGE;
    static bool attr_set[64];
    const int rc = pack_fused_opt_in((const void*)pack_rows_fused_kernel, attr_set, (size_t)(n_rows * row_len * 5 + 16));
    if (rc != SGNN_OK) return rc;
    hipLaunchKernelGGL(pack_rows_fused_kernel, dim3(1), dim3(SGNN_PACK_FUSED_THREADS), (size_t)(n_rows * row_len * 5 + 16), (hipStream_t)stream,
                       ids, mask, mode, n_rows, row_len, ptr, nodes);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_filter_sets_fused(const int64_t* set_ptr, const int32_t* set_nodes, const uint8_t* flags, int64_t n_sets,
                                      int64_t arena_entries, int64_t* out_ptr, int32_t* out_nodes, void* stream)
{
    if (!set_ptr || !set_nodes || !flags || !out_ptr || !out_nodes || n_sets < 1 || arena_entries < 1) return SGNN_ERR_BAD_ARG;
    // (the sets' entries are staged in LDS: the caller's arena -- at least the sets' total -- bounds them)
    if (n_sets > SGNN_PACK_FUSED_MAX_ROWS || arena_entries > pack_fused_entries_limit()) return SGNN_ERR_SET_TOO_LARGE;
    static bool attr_set[64];
    const int rc = pack_fused_opt_in((const void*)filter_sets_fused_kernel, attr_set, (size_t)(arena_entries * 5 + 16));
    if (rc != SGNN_OK) return rc;
    hipLaunchKernelGGL(filter_sets_fused_kernel, dim3(1), dim3(SGNN_PACK_FUSED_THREADS), (size_t)(arena_entries * 5 + 16), (hipStream_t)stream,
                       set_ptr, set_nodes, flags, n_sets, arena_entries, out_ptr, out_nodes);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_pack_rows_count(const int64_t* ids, const uint8_t* mask, int64_t n_rows, int64_t row_len, int64_t* counts,
                                    void* stream)
{
    if (!ids || !counts || n_rows < 0 || row_len < 0) return SGNN_ERR_BAD_ARG;
    if (n_rows == 0) return SGNN_OK;
    hipLaunchKernelGGL(pack_rows_count_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, mask, n_rows,
                       row_len, counts);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_pack_rows_write(const int64_t* ids, const uint8_t* mask, int64_t n_rows, int64_t row_len, const int64_t* ptr,
                                    int32_t* nodes, void* stream)
{
    if (!ids || !ptr || !nodes || n_rows < 0 || row_len < 0) return SGNN_ERR_BAD_ARG;
    if (n_rows == 0) return SGNN_OK;
    hipLaunchKernelGGL(pack_rows_write_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, mask, n_rows,
                       row_len, ptr, nodes);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_sort_sets(const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int64_t max_set_size,
                              int32_t* out_nodes, int32_t* out_pos, void* stream)
{
    if (!set_ptr || !set_nodes || !out_nodes || n_sets < 0 || max_set_size < 0) return SGNN_ERR_BAD_ARG;
    if (max_set_size > SORT_SETS_MAX) return SGNN_ERR_SET_TOO_LARGE;
    if (n_sets == 0 || max_set_size == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (max_set_size <= 16) {
        hipLaunchKernelGGL(sort_sets_kernel<16>, dim3(sgnn_grid_for(n_sets, 16)), dim3(256), 16 * 16 * 4, st, set_ptr,
                           set_nodes, n_sets, 16, out_nodes, out_pos);
    } else if (max_set_size <= 32) {
        hipLaunchKernelGGL(sort_sets_kernel<32>, dim3(sgnn_grid_for(n_sets, 8)), dim3(256), 8 * 32 * 4, st, set_ptr,
                           set_nodes, n_sets, 32, out_nodes, out_pos);
    } else {
        const int cap = (int)max_set_size;
        hipLaunchKernelGGL(sort_sets_kernel<64>, dim3(sgnn_grid_for(n_sets, 4)), dim3(256), (size_t)4 * cap * 4, st,
                           set_ptr, set_nodes, n_sets, cap, out_nodes, out_pos);
    }
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_khop_border_bitmap_fits_lds(int64_t max_id) { return kb_fits_lds(max_id) ? 1 : 0; }

extern "C" int64_t sgnn_khop_border_workspace_bytes(int64_t max_id, int64_t n_sets, int bitmap_in_lds) {
    // per workgroup: a BFS queue, plus the visited bitmap when it does not live in LDS
    const bool lds = bitmap_in_lds != 0;
    const int64_t per_wg = (lds ? 0 : kb_words(max_id) * 4) + (max_id + 1) * 4;
    return per_wg * kb_n_wg(n_sets, lds) + 16;                  // + the device-wide set counter
}

struct KbSample {             // fused neighbourhood-border anchor draw (all NULL/0 = off)
    int64_t n_slots;
    uint64_t h0;
    int64_t* anchor;          // (n_sets, n_slots) chosen border node id (0 for an empty border)
    uint8_t* hop;             // (n_sets, n_slots) its hop level
    uint8_t* allneg;          // (n_sets, n_slots) 1 if "every variate negative" (PAD wins if the row is padded)
    int64_t item_base;        // tape item of (set s, slot i) = (item_base + s) * n_slots + i
};

// The global-memory bitmap is written with L2 atomics only; a plain load could be served from a
// line the CU's vector L1 cached during the previous set's rank query, so read it at L2 too.
template <bool LDS_BM>
__device__ __forceinline__ uint32_t kb_word(const uint32_t* bm, int64_t w)
{
    if (LDS_BM) return bm[w];
    return __hip_atomic_load(bm + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rank tables + draw over the bitmap bm[0..words) that holds exactly the cnt border nodes of set s
#ifndef KB_TAKE
#define KB_TAKE 4                // sets a workgroup takes per trip to the device-wide counter (1 / 2 / 4: 2.50 / 2.37 / 2.35 ms)
#endif
#ifndef KB_STATIC_DISPATCH
#define KB_STATIC_DISPATCH 0     // 1: sets dealt round-robin in dispatch order instead of taken from a device-wide counter
#endif

template <bool LDS_BM, int THREADS>
__device__ __forceinline__ void kb_select(const KbSample& smp, int64_t s, const uint32_t* bm, int64_t words,
                                          const int32_t* __restrict__ q, int cnt, int hops, const int32_t* s_lvl,
                                          int32_t* s_pref, int32_t* s_wtot, int32_t* s_sel, int tid)
{
    constexpr int NW = THREADS / 64;
    const int lane = tid & 63, wave = tid >> 6;
    // phase 1: popcount of this thread's run of words, exclusive prefix over the workgroup
    const int64_t run = ((words + THREADS - 1) / THREADS) | 1;             // odd: conflict-free LDS strides
    const int64_t w0 = (int64_t)tid * run;
    const int64_t w1 = w0 + run < words ? w0 + run : words;
    int c = 0;
    for (int64_t w = w0; w < w1; ++w) c += __popc(kb_word<LDS_BM>(bm, w));
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    if (lane == 63) s_wtot[wave] = inc;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int w = 0; w < NW; ++w) { const int t = s_wtot[w]; s_wtot[w] = acc; acc += t; } }
    __syncthreads();
    s_pref[tid] = s_wtot[wave] + inc - c;
    __syncthreads();
    // phase 2: one thread per slot
    for (int64_t c0 = 0; c0 < smp.n_slots; c0 += KB_SEL_CHUNK) {
        const int64_t slot = c0 + tid;
        if (tid < KB_SEL_CHUNK && slot < smp.n_slots) {
            const int64_t o = s * smp.n_slots + slot;
            int32_t id = 0;
            uint8_t an = 1;
            if (cnt > 0) {
                const uint64_t h1 = sgnn_tape_h1(smp.h0, (uint64_t)(o + smp.item_base * smp.n_slots));
                an = sgnn_nanchor_allneg(h1, (uint32_t)cnt) ? 1 : 0;
                int rem = (int)sgnn_nanchor_index(h1, (uint32_t)cnt);
                int lo = 0, hi = THREADS - 1;                                // largest T with s_pref[T] <= rem
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pref[mid] <= rem) lo = mid; else hi = mid - 1; }
                rem -= s_pref[lo];
                int64_t w = (int64_t)lo * run;
                uint32_t word = kb_word<LDS_BM>(bm, w);
                int pc = __popc(word);
                while (rem >= pc) { rem -= pc; word = kb_word<LDS_BM>(bm, ++w); pc = __popc(word); }
                for (int t = 0; t < rem; ++t) word &= word - 1;             // drop the rem lowest set bits
                id = (int32_t)(w * 32 + (__ffs((int)word) - 1));
            }
            smp.anchor[o] = (int64_t)id;
            smp.allneg[o] = an;
            smp.hop[o] = (uint8_t)(cnt > 0 ? 1 : 0);
            if (hops > 1) s_sel[tid] = id;
        }
        if (hops > 1) {
            // hop level of the chosen nodes: their position in the (unsorted) BFS queue tells it
            __syncthreads();
            const int ns = (int)(smp.n_slots - c0 < KB_SEL_CHUNK ? smp.n_slots - c0 : KB_SEL_CHUNK);
            for (int i = s_lvl[1] + tid; i < cnt; i += THREADS) {            // hop-1 nodes keep the default
                const int32_t v = q[i];
                int h = 2;
                while (h < hops && i >= s_lvl[h]) ++h;
                for (int u = 0; u < ns; ++u) if (s_sel[u] == v) smp.hop[s * smp.n_slots + c0 + u] = (uint8_t)h;
            }
            __syncthreads();
        }
    }
}

template <bool LDS_BM, int THREADS>
__global__ __launch_bounds__(THREADS) void khop_border_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t max_id,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    int k, int ego_mode,
    int64_t* __restrict__ out_count, const int64_t* __restrict__ out_ptr,
    int32_t* __restrict__ out_nodes, uint8_t* __restrict__ out_hop,
    uint32_t* __restrict__ bitmaps, int32_t* __restrict__ queues, int64_t words, KbSample smp, int queue_in_output,
    unsigned long long* __restrict__ next_set, const int32_t* __restrict__ set_order)
{
    extern __shared__ uint32_t s_bm[];
    __shared__ int32_t s_qn;
    __shared__ int32_t s_lvl[260];
    __shared__ int32_t s_pref[THREADS];
    __shared__ int32_t s_wtot[THREADS / 64];
    __shared__ int32_t s_sel[KB_SEL_CHUNK];
    __shared__ int32_t s_tincl[64];               // frontier tile: inclusive degree scan, row starts
    __shared__ uint32_t s_tr0[64];
    __shared__ int32_t s_tdeg[64];
    __shared__ uint32_t s_big[2];                 // members of the tile whose list is streamed on its own
    uint32_t* bm = LDS_BM ? s_bm : bitmaps + (int64_t)blockIdx.x * words;
    int32_t* q = queue_in_output ? nullptr : queues + (int64_t)blockIdx.x * (max_id + 1);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = THREADS / 64;
    const int hops = ego_mode ? 1 : k;
    // the queue feeds the next hop's frontier, the materialised output and the bit clean-up; a
    // one-hop count / fused draw on the LDS bitmap needs none of them (the bitmap is wiped whole)
    const bool need_queue = !(LDS_BM && hops == 1 && out_nodes == nullptr);
    if (LDS_BM) {
        for (int64_t i = tid; i < words; i += THREADS) s_bm[i] = 0;
        __syncthreads();
    }
    // Sets are taken from a device-wide counter (their cost varies by orders of magnitude with the
    // members' degrees: a static round-robin leaves CUs idle behind the unlucky ones), in the caller's
    // dispatch order when given (heaviest first).
    __shared__ long long s_next;
#if KB_STATIC_DISPATCH
    for (int64_t si = blockIdx.x; si < n_sets; si += gridDim.x) {
        __syncthreads();
#else
    int64_t si_next = 0, si_end = 0;                          // KB_TAKE consecutive sets per trip to the counter
    while (true) {
        if (si_next >= si_end) {                              // uniform over the workgroup
            if (tid == 0) s_next = (long long)atomicAdd(next_set, (unsigned long long)KB_TAKE);
            __syncthreads();
            si_next = s_next;
            si_end = si_next + KB_TAKE;
        }
        const int64_t si = si_next++;
        if (si >= n_sets) break;
#endif
        const int64_t s = set_order ? set_order[si] : si;
        const int64_t beg = set_ptr[s];
        const int n = (int)(set_ptr[s + 1] - beg);
        if (queue_in_output) q = out_nodes + out_ptr[s];        // the caller's slice IS the queue
        if (tid == 0) { s_qn = 0; s_lvl[0] = 0; }              // s_lvl[h] = queue length after hop h
        // members' bits: a set of at most 64 nodes is one frontier tile, whose builder (wavefront 0)
        // sets them on the way -- one load round and one barrier less per set
        const bool members_in_tile = n <= 64;
        if (!members_in_tile) {
            for (int i = tid; i < n; i += THREADS) {
                const int32_t v = set_nodes[beg + i];
                atomicOr(&bm[v >> 5], 1u << (v & 31));
            }
            __syncthreads();
        }
        for (int h = 1; h <= hops; ++h) {
            const int f0 = (h == 1) ? 0 : s_lvl[h - 2];      // frontier of hop h = nodes found at hop h-1
            const int f1 = (h == 1) ? n : s_lvl[h - 1];
            // The frontier is taken 64 nodes at a time and their neighbour lists are handled as ONE
            // flat edge range dealt to the wavefronts in 64-edge chunks (a hub's list is shared by
            // all wavefronts instead of serialising one of them), KB_INFLIGHT chunks in flight per wavefront.
            for (int t0 = f0; t0 < f1; t0 += 64) {
                if (wave == 0) {
                    int32_t deg = 0;
                    uint32_t r0 = 0;
                    if (t0 + lane < f1) {
                        const int32_t v = (h == 1) ? set_nodes[beg + t0 + lane] : q[t0 + lane];
                        if (h == 1 && members_in_tile) atomicOr(&bm[v >> 5], 1u << (v & 31));
                        const int64_t a = rowptr[v], b = rowptr[v + 1];
                        r0 = (uint32_t)a;
                        deg = (int32_t)(b - a);
                    }
                    // lists of at least KB_BIG entries are streamed one at a time by all wavefronts (no
                    // search per chunk); the short ones form the flat range
                    const uint64_t bigm = __ballot(deg >= KB_BIG);
                    int32_t incl = deg >= KB_BIG ? 0 : deg;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
                    s_tincl[lane] = incl;
                    s_tr0[lane] = r0;
                    s_tdeg[lane] = deg;
                    if (lane == 0) { s_big[0] = (uint32_t)bigm; s_big[1] = (uint32_t)(bigm >> 32); }
                }
                __syncthreads();
                int wave_new = 0;
                // claim the unseen neighbours of all chunks of a group first, then append them with ONE
                // reservation on the queue counter per wavefront and group (the counter is a single LDS
                // word shared by 16 wavefronts); a count / draw that needs no queue counts in a register
#define KB_PROCESS_GROUP()                                                                                     \
                do {                                                                                           \
                    uint64_t mk[KB_INFLIGHT];                                                                  \
                    int32_t cc[KB_INFLIGHT];                                                                   \
                    int n_new = 0;                                                                             \
                    _Pragma("unroll") for (int u = 0; u < KB_INFLIGHT; ++u) {                                  \
                        bool fresh = false;                                                                    \
                        cc[u] = ego_mode ? c[u] - 1 : c[u];                                                    \
                        if (valid[u]) {                                                                        \
                            const uint32_t bit = 1u << (cc[u] & 31);                                           \
                            const uint32_t old = atomicOr(&bm[cc[u] >> 5], bit);                               \
                            fresh = !(old & bit);                                                              \
                        }                                                                                      \
                        mk[u] = __ballot(fresh);                                                               \
                        n_new += (int)__popcll(mk[u]);                                                         \
                    }                                                                                          \
                    if (n_new) {                                                                               \
                        if (need_queue) {                                                                      \
                            int base = 0;                                                                      \
                            if (lane == 0) base = atomicAdd(&s_qn, n_new);                                     \
                            base = __shfl(base, 0);                                                            \
                            _Pragma("unroll") for (int u = 0; u < KB_INFLIGHT; ++u) {                          \
                                if ((mk[u] >> lane) & 1ull) q[base + __popcll(mk[u] & ((1ull << lane) - 1ull))] = cc[u]; \
                                base += (int)__popcll(mk[u]);                                                  \
                            }                                                                                  \
                        } else {                                                                               \
                            wave_new += n_new;                                                                 \
                        }                                                                                      \
                    }                                                                                          \
                } while (0)
#ifndef KB_DEBUG_SKIP_EDGES
                // (a) the long lists, one after the other, every wavefront a share of each
                uint64_t bigm = ((uint64_t)s_big[1] << 32) | s_big[0];
                while (bigm) {
                    const int m = __ffsll((unsigned long long)bigm) - 1;
                    bigm &= bigm - 1;
                    const uint32_t m_r0 = s_tr0[m];
                    const int32_t m_deg = s_tdeg[m];
                    for (int32_t cb = wave * 64; cb < m_deg; cb += KB_INFLIGHT * NW * 64) {
                        int32_t c[KB_INFLIGHT];
                        bool valid[KB_INFLIGHT];
#pragma unroll
                        for (int u = 0; u < KB_INFLIGHT; ++u) {
                            const int32_t t = cb + u * NW * 64 + lane;
                            valid[u] = t < m_deg;
                            c[u] = valid[u] ? col[m_r0 + (uint32_t)t] : 0;
                        }
                        KB_PROCESS_GROUP();
                    }
                }
                // (b) the short lists as one flat edge range
                const int32_t incl = s_tincl[lane];
                const uint32_t r0 = s_tr0[lane];
                int32_t excl = __shfl_up(incl, 1);
                if (lane == 0) excl = 0;
                const int32_t total = __shfl(incl, 63);
                for (int32_t cb = wave * 64; cb < total; cb += KB_INFLIGHT * NW * 64) {
                    int32_t c[KB_INFLIGHT];
                    bool valid[KB_INFLIGHT];
#pragma unroll
                    for (int u = 0; u < KB_INFLIGHT; ++u) {
                        const int32_t t = cb + u * NW * 64 + lane;
                        valid[u] = t < total;
                        int lo = 0, hi = 63;                      // smallest m with incl[m] > t
#pragma unroll
                        for (int it = 0; it < 6; ++it) {
                            const int mid = (lo + hi) >> 1;
                            const int32_t x = __shfl(incl, mid);
                            if (x > t) hi = mid; else lo = mid + 1;
                        }
                        const int m = lo & 63;
                        const int32_t m_excl = __shfl(excl, m);
                        const uint32_t m_r0 = __shfl(r0, m);
                        c[u] = valid[u] ? col[m_r0 + (uint32_t)(t - m_excl)] : 0;
                    }
                    KB_PROCESS_GROUP();
                }
#endif
#undef KB_PROCESS_GROUP
                if (!need_queue && wave_new) {                     // wave-uniform
                    if (lane == 0) atomicAdd(&s_qn, wave_new);
                }
                __syncthreads();                                  // the tile tables are rewritten next
            }
            if (tid == 0) s_lvl[h] = s_qn;
            __syncthreads();
        }
        const int cnt = s_qn;
        if (out_count && tid == 0) out_count[s] = cnt;
        if (out_nodes != nullptr && !queue_in_output) {
            const int64_t o = out_ptr[s];
            for (int i = tid; i < cnt; i += THREADS) {
                out_nodes[o + i] = q[i];
                if (out_hop) {
                    int h = 1;
                    while (h < hops && i >= s_lvl[h]) ++h;
                    out_hop[o + i] = (uint8_t)h;
                }
            }
        }
        // un-set the members' bits: what is left in the bitmap is exactly the border
        for (int i = tid; i < n; i += THREADS) {
            const int32_t v = set_nodes[beg + i];
            atomicAnd(&bm[v >> 5], ~(1u << (v & 31)));
        }
#ifndef KB_DEBUG_SKIP_SELECT
        if (smp.n_slots > 0) {
            __syncthreads();
            kb_select<LDS_BM, THREADS>(smp, s, bm, words, q, cnt, hops, s_lvl, s_pref, s_wtot, s_sel, tid);
            __syncthreads();
        }
#endif
        // un-set every border bit so the next component handled by the workgroup starts clean
#ifdef KB_DEBUG_SKIP_WIPE
        if (false) {
#else
        if (LDS_BM && (!need_queue || (int64_t)cnt * 8 > words)) {
#endif
            // cheaper than re-reading the queue: wipe the LDS bitmap with 16-byte stores
            int4* bm4 = reinterpret_cast<int4*>(s_bm);
            for (int64_t i = tid; i < (words + 3) / 4; i += THREADS) bm4[i] = make_int4(0, 0, 0, 0);
        } else {
            for (int i = tid; i < cnt; i += THREADS) {
                const int32_t c = q[i];
                atomicAnd(&bm[c >> 5], ~(1u << (c & 31)));
            }
        }
        __syncthreads();
    }
}

static int kb_launch(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int k, int ego_dict_mode,
                     int64_t* out_count, const int64_t* out_ptr, int32_t* out_nodes, uint8_t* out_hop,
                     void* workspace, int64_t workspace_bytes, int bitmap_in_lds, KbSample smp, void* stream,
                     int queue_in_output = 0, const int32_t* set_order = nullptr)
{
    if (!rowptr || !col || !set_ptr || !set_nodes || !workspace || n_sets < 0 || k < 1 || k > 255)
        return SGNN_ERR_BAD_ARG;
    if (out_nodes != nullptr && out_ptr == nullptr) return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (bitmap_in_lds && !kb_fits_lds(max_id)) return SGNN_ERR_BAD_ARG;
    if (workspace_bytes < sgnn_khop_border_workspace_bytes(max_id, n_sets, bitmap_in_lds)) return SGNN_ERR_BAD_ARG;
    if (n_sets == 0) return SGNN_OK;
    const int64_t words = kb_words(max_id);
    const int64_t nwg = kb_n_wg(n_sets, bitmap_in_lds != 0);
    uint32_t* bitmaps = (uint32_t*)workspace;                     // empty region for the LDS variant
    int32_t* queues = (int32_t*)(bitmaps + (bitmap_in_lds ? 0 : words * nwg));
    hipStream_t st = (hipStream_t)stream;
    // the set counter sits behind the per-workgroup regions, 8-byte aligned
    const int64_t body = ((bitmap_in_lds ? 0 : words * 4) + (max_id + 1) * 4) * nwg;
    unsigned long long* next_set = (unsigned long long*)((char*)workspace + ((body + 7) & ~(int64_t)7));
    { const hipError_t me = hipMemsetAsync(next_set, 0, 8, st); if (me != hipSuccess) { sgnn_set_last_error(me); return SGNN_ERR_LAUNCH; } }
    if (bitmap_in_lds) {
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute((const void*)khop_border_kernel<true, KB_THREADS_L>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, KB_LDS_BYTES);
            attr_set = true;
        }
        hipLaunchKernelGGL((khop_border_kernel<true, KB_THREADS_L>), dim3((int)nwg), dim3(KB_THREADS_L),
                           (size_t)(((words + 3) / 4) * 16), st, rowptr, col, max_id, set_ptr, set_nodes, n_sets, k, ego_dict_mode,
                           out_count, out_ptr, out_nodes, out_hop, bitmaps, queues, words, smp, queue_in_output, next_set,
                           set_order);
    } else {
        hipLaunchKernelGGL((khop_border_kernel<false, KB_THREADS_G>), dim3((int)nwg), dim3(KB_THREADS_G), 0, st,
                           rowptr, col, max_id, set_ptr, set_nodes, n_sets, k, ego_dict_mode, out_count, out_ptr,
                           out_nodes, out_hop, bitmaps, queues, words, smp, queue_in_output, next_set, set_order);
    }
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_khop_border(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                                const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                int k, int ego_dict_mode,
                                int64_t* out_count, const int64_t* out_ptr, int32_t* out_nodes, uint8_t* out_hop,
                                void* workspace, int64_t workspace_bytes, int bitmap_in_lds, void* stream)
{
    if (out_nodes == nullptr && out_count == nullptr) return SGNN_ERR_BAD_ARG;
    KbSample none = {0, 0, nullptr, nullptr, nullptr};
    return kb_launch(rowptr, col, nnz, max_id, set_ptr, set_nodes, n_sets, k, ego_dict_mode, out_count, out_ptr,
                     out_nodes, out_hop, workspace, workspace_bytes, bitmap_in_lds, none, stream);
}

/* One-pass variant: the caller reserves, for every set, a slice of `arena` that is guaranteed to hold
 * its border (slice s starts at arena_off[s]); the BFS uses the slice as its queue, so the border is
 * materialised by the BFS itself -- no count pass, no copy.  For k = 1 the bound
 * sum_{v in set} deg(v) always suffices. */
extern "C" int sgnn_khop_border_arena(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                                      const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int k,
                                      const int64_t* arena_off, int32_t* arena, int64_t* out_count,
                                      void* workspace, int64_t workspace_bytes, int bitmap_in_lds, void* stream)
{
    if (!arena_off || !arena || !out_count) return SGNN_ERR_BAD_ARG;
    KbSample none = {0, 0, nullptr, nullptr, nullptr};
    return kb_launch(rowptr, col, nnz, max_id, set_ptr, set_nodes, n_sets, k, 0, out_count, arena_off, arena, nullptr,
                     workspace, workspace_bytes, bitmap_in_lds, none, stream, 1);
}

// ---------------------------------------------------------------------------------------------
// One-hop border + fused neighbourhood-border anchor draw, specialised (k = 1, no materialised border):
// the shape of the benchmark and of every reference configuration with ego_graphs.txt present.
// What differs from the general kernel above:
//   * the expansion only ORs bits -- LDS atomics WITHOUT return value, no ballots, no queue: the border
//     size is the popcount total the rank query computes anyway.  Nothing in the edge loop waits for
//     LDS, so its global loads stay in flight;
//   * every wavefront loads the members and their row pointers itself (20 cached loads) instead of
//     waiting at a barrier for a tile table built by wavefront 0; every list is shared by all 16
//     wavefronts in 64-entry chunks (no flat-range search);
//   * a slot of the draw is answered by a group of 16 lanes instead of one thread: two LDS reads find the
//     run (16 coarse prefixes, then 64 fine ones as 4 per lane), one round per 16 words scans the run --
//     ~10 dependent steps where a single thread's binary search + walk took ~70;
//   * id ranges beyond the LDS bitmap are processed in slices of the range (SLICED; rows ascending,
//     each member's slice of its list found by binary search): pass A counts the border per slice,
//     pass B rebuilds each slice and answers the slots whose rank falls into it -- twice the edge
//     traffic, against the ~9x of keeping the bitmap in L2.
// ---------------------------------------------------------------------------------------------
#define K1_THREADS 1024         // 16 wavefronts: the rank tables below are laid out for exactly this (64 groups of 16 lanes, one DPP row of wavefront totals)
#define K1_MAX_SLICES 32
#ifndef K1_INFLIGHT
#define K1_INFLIGHT 4           // 64-entry chunk loads a wavefront issues before it sets any bit
#endif
#ifndef K1_RANK_UNROLL
#define K1_RANK_UNROLL 8
#endif
#ifndef K1_LONG
#define K1_LONG 512             // lists of at least this many entries are shared by all wavefronts of the workgroup
#endif
#ifndef K1_TAKE
#define K1_TAKE 8               // sets per trip to the device-wide counter (all but the first of a trip are prefetched)
#endif

#ifdef K1_DEBUG_TIMING
__device__ unsigned long long k1_dbg[16];
#define K1_T(i) do { if (tid == 0) { const unsigned long long now_ = clock64(); atomicAdd(&k1_dbg[i], now_ - t_last); t_last = now_; } } while (0)
extern "C" int sgnn_debug_k1_timing(unsigned long long* out16, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(k1_dbg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(k1_dbg), 16 * 8);
}
#else
#define K1_T(i) do { } while (0)
#endif

__device__ __forceinline__ void k1_lds_barrier()
{
    // orders LDS only: the global loads queued for the next set stay in flight across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int64_t sgnn_readlane64(int64_t v, int l)       // l wave-uniform
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int k1_nth_set_bit(uint32_t word, int n)           // position of the n-th (0-based) set bit
{
    int pos = 0;
#pragma unroll
    for (int shift = 16; shift >= 1; shift >>= 1) {
        const int c = __popc((word >> pos) & ((1u << shift) - 1u));
        if (n >= c) { n -= c; pos += shift; }
    }
    return pos;
}

template <bool SLICED>
__global__ __launch_bounds__(K1_THREADS) void khop1_sample_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t max_id,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    int64_t* __restrict__ out_count, KbSample smp, int64_t slice_ids, int n_slices,
    unsigned long long* __restrict__ next_set, const int32_t* __restrict__ set_order)
{
    extern __shared__ uint32_t s_bm[];                       // bitmap of one slice of the id range
    __shared__ __attribute__((aligned(16))) int32_t s_pref[K1_THREADS];
    __shared__ int32_t s_wtot[K1_THREADS / 64 + 1];
    __shared__ int32_t s_cum[K1_MAX_SLICES + 1];
    __shared__ long long s_next;
    __shared__ uint32_t s_u0[K1_THREADS / 16], s_u1[K1_THREADS / 16];     // the first 64 slots' two tape draws of the set in hand
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);          // the same number, known to be uniform
    constexpr int NW = K1_THREADS / 64;
    const int grp = tid >> 4, gl = tid & 15, gshift = (lane >> 4) * 16;     // 64 groups of 16 lanes
    const int words_alloc = (int)((((slice_ids + 31) / 32 + K1_THREADS - 1) / K1_THREADS) | 1) * K1_THREADS;   // host: k1_alloc_bytes
    {
        int4* bm4 = reinterpret_cast<int4*>(s_bm);
        for (int i = tid; i < words_alloc / 4; i += K1_THREADS) bm4[i] = make_int4(0, 0, 0, 0);
    }
    __syncthreads();
    int64_t si_next = 0, si_end = 0;
#ifdef K1_DEBUG_TIMING
    unsigned long long t_last = clock64();
#endif
    bool have_pf = false;
    uint32_t pf_r0 = 0, pf_r1 = 0;                           // the prefetched row's start and end, RAW: see step 2 below
    int32_t nx_v = 0;                                        // the NEXT set's first 64 members (prefetch step 1); read back as v_pf one set later
    int64_t trip0 = 0, trip_ptr = 0;
    while (true) {
        if (si_next >= si_end) {
            if (tid == 0) s_next = (long long)atomicAdd(next_set, (unsigned long long)K1_TAKE);
            __syncthreads();
            si_next = s_next;
            si_end = si_next + K1_TAKE;
            __syncthreads();
            // the trip's slice of set_ptr in one load (lane j: set si_next + j), read back lane by lane below
            trip0 = si_next;
            trip_ptr = 0;
            if (set_order == nullptr && lane <= K1_TAKE && si_next + lane <= n_sets) trip_ptr = set_ptr[si_next + lane];
        }
        K1_T(0);                                                   // dispatch
        const int64_t si = si_next++;
        if (si >= n_sets) break;
        const int64_t s = set_order ? set_order[si] : si;
        const int tj = __builtin_amdgcn_readfirstlane((int)(si - trip0));     // position within the trip (wave-uniform)
        // (the first 64 members' row pointers may have been fetched while the previous set was being processed)
        const int64_t beg = set_order ? set_ptr[s] : sgnn_readlane64(trip_ptr, tj);
        const int n = (int)((set_order ? set_ptr[s + 1] : sgnn_readlane64(trip_ptr, tj + 1)) - beg);
        const uint32_t r0_pf = pf_r0, r1_pf = pf_r1;
        const bool use_pf = have_pf;
        have_pf = false;
        // the next set of this trip, if there is one: its members are loaded now and their row pointers after
        // the expansion -- each step is issued long after the previous one returned, and the LDS-only barriers
        // in between leave the loads in flight: the set_ptr -> members -> row pointers -> lists chain of
        // dependent global latencies (~4 x 1 us per set on one resident workgroup) is paid once per trip
        // The slots' tape draws (four 64-bit mixes each) do not depend on anything the set computes: ONE wavefront hashes
        // them now, under the latency of the first loads, and the barriers of the expansion publish them -- hashed by
        // every 16-lane group at draw time they were 11 wavefronts x ~130 vector instructions per set, a quarter of
        // the kernel's vector work.  (The previous set's draw has finished: two barriers lie behind it.)
        if (wave_s == NW - 1 && lane < smp.n_slots) {
            const uint64_t h1_ = sgnn_tape_h1(smp.h0, (uint64_t)(s * smp.n_slots + lane + smp.item_base * smp.n_slots));
            s_u0[lane] = (uint32_t)(sgnn_tape_draw(h1_, 0) >> 32);
            s_u1[lane] = (uint32_t)(sgnn_tape_draw(h1_, 1) >> 32);
        }
        const bool can_pf = !SLICED && set_order == nullptr && si_next < si_end && si_next < n_sets;
        int64_t nx_beg = 0, nx_end = 0;
        if (can_pf) {
            nx_beg = sgnn_readlane64(trip_ptr, tj + 1);
            nx_end = sgnn_readlane64(trip_ptr, tj + 2);
        }
        const int32_t v_pf = nx_v;                                 // this set's first 64 members, if prefetched (copied HERE, a set after the load: no wait)
        int32_t v_tile0 = 0;
        int cnt = 0;
        for (int pass = 0; pass < (SLICED ? 2 : 1); ++pass) {
            for (int sl = 0; sl < n_slices; ++sl) {
                const int64_t lo_id = (int64_t)sl * slice_ids;
                const int64_t hi_id = lo_id + slice_ids < max_id + 1 ? lo_id + slice_ids : max_id + 1;
                const int64_t words = (hi_id - lo_id + 31) / 32;
                // ---- expansion: OR the bits of every member's neighbours (of this slice) ----------------
                // Every wavefront holds the tile's row starts and degrees (lane m = member m) and the 64-entry
                // chunks of all lists are numbered through: chunk q belongs to wavefront q % 16, which finds
                // its member by a scalar search over the in-register chunk scan and keeps K1_INFLIGHT chunk
                // loads in flight before any bit is set (a list per iteration would serialise 20 latencies).
                for (int t0 = 0; t0 < n; t0 += 64) {
                    uint32_t r0 = 0;
                    int32_t deg = 0;
                    if (t0 == 0 && use_pf) { r0 = r0_pf; deg = (int32_t)(r1_pf - r0_pf); v_tile0 = v_pf; }
                    else if (t0 + lane < n) {
                        const int32_t v = set_nodes[beg + t0 + lane];
                        if (t0 == 0) v_tile0 = v;
                        int64_t a = rowptr[v], b = rowptr[v + 1];
                        if (SLICED) {                              // rows ascending: the list's part inside [lo_id, hi_id)
                            int64_t l = a, h = b;
                            while (l < h) { const int64_t m = (l + h) >> 1; if ((int64_t)col[m] < lo_id) l = m + 1; else h = m; }
                            const int64_t first = l;
                            h = b;
                            while (l < h) { const int64_t m = (l + h) >> 1; if ((int64_t)col[m] < hi_id) l = m + 1; else h = m; }
                            a = first; b = l;
                        }
                        r0 = (uint32_t)a;
                        deg = (int32_t)(b - a);
                    }
                    K1_T(6);                                       // (debug) tile ready
                    // step 1 of the prefetch, issued only now: a load issued before the prefetched row pointers
                    // above are consumed would have to complete first (the counter of outstanding loads cannot
                    // tell them apart across the loop's back edge)
                    if (t0 == 0 && can_pf && lane < nx_end - nx_beg) nx_v = set_nodes[nx_beg + lane];
                    // Who reads what, without any search: a list of >= K1_LONG entries is shared by all wavefronts
                    // (wavefront w takes the 64-entry chunks w, w + 16, ...), a shorter one belongs whole to wavefront
                    // (member index) % 16.  Everything that selects a list is a ballot bit or a v_readlane -- rounds 1-2
                    // numbered all chunks through and every wavefront walked the whole member list in scalar registers
                    // to find its own: ~500 scalar instructions per wavefront and set.
#define K1_OR_BIT(C) do { if ((C) >= 0) { const int32_t x_ = (C) - (int32_t)lo_id; K1_OR_AT(x_); } } while (0)
#ifdef K1_DEBUG_NO_OR
#define K1_OR_AT(X) do { if ((X) == 0x7fffffff) s_bm[0] = 1; } while (0)
#else
#define K1_OR_AT(X) __hip_atomic_fetch_or(&s_bm[(X) >> 5], 1u << ((X) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif
#ifdef K1_DEBUG_NO_LOAD
#define K1_COL(I) (int32_t)(((uint32_t)(I)) * 2654435761u % (uint32_t)max_id)
#else
#define K1_COL(I) col[(I)]
#endif
#ifdef K1_DEBUG_SKIP_CHUNKS
                    if (false)
#endif
                    {
                        uint64_t longs = __ballot(deg >= K1_LONG);
                        while (longs) {                                               // scalar loop over the long lists
                            const int m = __ffsll((long long)longs) - 1;
                            longs &= longs - 1;
                            const int32_t m_deg = __builtin_amdgcn_readlane(deg, m);
                            const uint32_t m_r0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, m);
                            for (int32_t tb = wave_s * 64; tb < m_deg; tb += K1_INFLIGHT * NW * 64) {
                                int32_t c[K1_INFLIGHT];
#pragma unroll
                                for (int u = 0; u < K1_INFLIGHT; ++u) {
                                    const int32_t t = tb + u * NW * 64 + lane;
                                    c[u] = t < m_deg ? K1_COL(m_r0 + (uint32_t)t) : -1;
                                }
#pragma unroll
                                for (int u = 0; u < K1_INFLIGHT; ++u) K1_OR_BIT(c[u]);
                            }
                        }
                        // (skipping the slots past a list's end with scalar tests, or unrolling the short lists exactly,
                        // was slower: 1.48 -> 1.52-1.60 ms -- the branches cost more than the predicated-off instructions)
                        for (int m = wave_s; m < 64 && t0 + m < n; m += NW) {          // this wavefront's short lists
                            const int32_t m_deg = __builtin_amdgcn_readlane(deg, m);
                            if (m_deg >= K1_LONG) continue;
                            const uint32_t m_r0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, m);
                            for (int32_t tb = 0; tb < m_deg; tb += K1_INFLIGHT * 64) {
                                int32_t c[K1_INFLIGHT];
#pragma unroll
                                for (int u = 0; u < K1_INFLIGHT; ++u) {
                                    const int32_t t = tb + u * 64 + lane;
                                    c[u] = t < m_deg ? K1_COL(m_r0 + (uint32_t)t) : -1;
                                }
#pragma unroll
                                for (int u = 0; u < K1_INFLIGHT; ++u) K1_OR_BIT(c[u]);
                            }
                        }
                    }
#undef K1_OR_BIT
#undef K1_OR_AT
#undef K1_COL
                }
                K1_T(7);                                               // (debug) own chunks done
                if (n == 0 && can_pf && lane < nx_end - nx_beg) nx_v = set_nodes[nx_beg + lane];    // (no tile ran step 1)
                if (can_pf) {                                          // step 2 of the prefetch: the next set's row pointers
                    // (the loaded words go into pf_r0 / pf_r1 untouched: forming the degree here made the compiler wait for the
                    // load on the spot -- the whole prefetch latency sat in front of the expansion's barrier)
                    // Only the LOW words of the two row pointers are loaded (nnz < 2^31), each into its own register: a 16-byte
                    // load left two dead registers in its destination, the allocator put a live value into one of them, and that copy
                    // waited for the load as well.
                    pf_r0 = 0; pf_r1 = 0;
                    if (lane < nx_end - nx_beg) {
                        const uint32_t* __restrict__ rp32 = reinterpret_cast<const uint32_t*>(rowptr) + 2 * (int64_t)nx_v;
                        pf_r0 = rp32[0];
                        pf_r1 = rp32[2];
                    }
                    have_pf = true;
                }
                k1_lds_barrier();
                K1_T(1);                                           // expansion
                // ---- the members themselves are not border ---------------------------------------------
                // (two separate paths: with the load of the members beyond the first tile in the same loop, the wait for it sat on
                // the common path -- and, the counter of outstanding loads being in-order, waited for the prefetch above as well)
                if (tid < 64 && tid < n) {                                           // wavefront 0 still holds the first tile
                    const int64_t v = (int64_t)v_tile0;
                    if (v >= lo_id && v < hi_id) {
                        const int32_t x = (int32_t)(v - lo_id);
                        __hip_atomic_fetch_and(&s_bm[x >> 5], ~(1u << (x & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (n > 64) {
                    for (int i = 64 + tid; i < n; i += K1_THREADS) {
                        const int64_t v = (int64_t)set_nodes[beg + i];
                        if (v >= lo_id && v < hi_id) {
                            const int32_t x = (int32_t)(v - lo_id);
                            __hip_atomic_fetch_and(&s_bm[x >> 5], ~(1u << (x & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                k1_lds_barrier();
                K1_T(2);                                           // members un-set
                // ---- rank table: popcount of each thread's run of words, exclusive prefix ----------------
                const int run = (int)(((words + K1_THREADS - 1) / K1_THREADS) | 1);     // odd: conflict-free strides
                int c = 0;
                {
                    // 32-bit indices and no bounds tests (the allocation is run x K1_THREADS words, zero beyond `words`):
                    // whole groups of eight independent reads, then the LAST eight words of the run, of which the ones the
                    // groups have already counted are masked by loop-invariant scalars -- a read + a v_bcnt per word (the
                    // first version selected every index and every count with scalar compares: ~10 instructions per word)
                    const uint32_t* __restrict__ mine_ = s_bm + tid * run;
#ifdef K1_DEBUG_SKIP_RANK
                    if (false)
#endif
                    if (run >= 8) {
                        for (int g = 0; g < (run >> 3); ++g) {
                            uint32_t x[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) x[u] = mine_[8 * g + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) c += __popc(x[u]);
                        }
                        const int first_new = 8 - (run & 7);                  // of the last eight words, those from here on are new
                        const uint32_t* __restrict__ tail_ = mine_ + run - 8;
                        uint32_t x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = tail_[u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) c += u >= first_new ? __popc(x[u]) : 0;
                    } else {
                        for (int u = 0; u < run; ++u) c += __popc(mine_[u]);
                    }
                }
                // (the whole per-set chain below is latency, not throughput: one workgroup per CU, every step waits for
                // the one before -- scans run on the DPP path, every wavefront scans the 16 wavefront totals itself
                // instead of waiting for thread 0 to walk them, and no step goes through an LDS mailbox)
                // s_pref holds WAVEFRONT-LOCAL exclusive prefixes and s_wtot the 16 wavefront totals: one barrier publishes
                // both, and every 16-lane group scans the totals itself (a DPP row scan) where a second barrier used to
                // separate "totals known" from "global prefixes written"
                const int inc = sgnn_wave_incl_scan(c);
                s_pref[tid] = inc - c;
                if (lane == 63) s_wtot[wave] = inc;
                k1_lds_barrier();
                const int wt = s_wtot[gl];                                        // NW = 16 totals, replicated in every row
                const int wincl = sgnn_row_incl_scan(wt);
                const int wexcl = wincl - wt;                                     // lane gl: ranks before wavefront gl's runs
                const int total = __builtin_amdgcn_readlane(wincl, NW - 1);
                if (SLICED && pass == 0) {
                    if (tid == 0) { if (sl == 0) s_cum[0] = 0; s_cum[sl + 1] = (sl == 0 ? 0 : s_cum[sl]) + total; }
                    k1_lds_barrier();                                             // (pass 1 reads s_cum)
                }
                K1_T(3);                                           // rank table
                if (!(SLICED && pass == 0)) {
                    cnt = SLICED ? s_cum[n_slices] : total;
                    const int cum_lo = SLICED ? s_cum[sl] : 0;
                    if (sl == 0 && tid == 0 && out_count) out_count[s] = cnt;
                    // ---- the draw: 64 slots per round, a group of 16 lanes (one DPP row) each -----------------
                    // Every lane of a group hashes its slot's tape draws itself (three 64-bit mixes: cheaper than a
                    // trip through LDS and a barrier), the searches' prefix sums run on the DPP path inside the row,
                    // and the lane that holds the answer writes it.
#ifdef K1_DEBUG_SKIP_DRAW
                    if (false)
#endif
                    for (int64_t c0 = 0; c0 < smp.n_slots; c0 += K1_THREADS / 16) {
                        if (c0 + wave_s * 4 >= smp.n_slots) continue;                // none of this wavefront's four slots exists
                        const int64_t slot = c0 + grp;
                        const bool active = slot < smp.n_slots;
                        const int64_t o = s * smp.n_slots + (active ? slot : 0);
                        int rem = -1;
                        uint8_t an = 1;
                        if (active && cnt > 0) {
                            uint32_t u0, u1;
                            if (c0 == 0) { u0 = s_u0[grp]; u1 = s_u1[grp]; }                  // hashed at the set's start
                            else {                                                            // more than 64 slots: the later rounds hash here
                                const uint64_t h1_ = sgnn_tape_h1(smp.h0, (uint64_t)(s * smp.n_slots + slot + smp.item_base * smp.n_slots));
                                u0 = (uint32_t)(sgnn_tape_draw(h1_, 0) >> 32);
                                u1 = (uint32_t)(sgnn_tape_draw(h1_, 1) >> 32);
                            }
                            // sgnn_nanchor_index / sgnn_nanchor_allneg (common.h) on the two draws
                            rem = (int32_t)(((uint64_t)u0 * (uint64_t)(uint32_t)cnt) >> 32);
                            an = cnt > 32 ? 0 : ((cnt == 32 ? u1 == 0u : (u1 >> (32 - cnt)) == 0u) ? 1 : 0);
                        }
                        rem -= cum_lo;
                        const bool mine = active && cnt > 0 && rem >= 0 && rem < total;       // the rank lies in this slice
                        if (!mine) rem = 0;
                        // coarse: the wavefront whose runs hold the rank (16 exclusive prefixes, one per lane of the group)
                        const uint32_t b1 = (uint32_t)(__ballot(wexcl <= rem) >> gshift) & 0xffffu;
                        const int blk = 31 - __clz((int)(b1 | 1u));
                        rem -= __shfl(wexcl, blk, 16);                                        // ranks inside that wavefront's runs
                        // fine: the wavefront's 64 local prefixes, 4 per lane
                        const int4 e = *reinterpret_cast<const int4*>(&s_pref[blk * 64 + gl * 4]);
                        int n_le = __popc((uint32_t)(__ballot(e.x <= rem) >> gshift) & 0xffffu);
                        n_le += __popc((uint32_t)(__ballot(e.y <= rem) >> gshift) & 0xffffu);
                        n_le += __popc((uint32_t)(__ballot(e.z <= rem) >> gshift) & 0xffffu);
                        n_le += __popc((uint32_t)(__ballot(e.w <= rem) >> gshift) & 0xffffu);
                        const int T = blk * 64 + (n_le > 0 ? n_le - 1 : 0);
                        // (the prefix of run T is one of the four values this group has just read: lane (T % 64) / 4 holds it)
                        rem -= s_pref[T];
                        // the run's words, 16 per round in word order
                        const int rw0 = T * run, rw1 = rw0 + run;
                        bool found = false;
                        for (int wb = 0; wb < run; wb += 16) {                                // run is the same for every group
                            const int w = rw0 + wb + gl;
                            const uint32_t word = w < rw1 ? s_bm[w] : 0u;
                            const int pc = __popc(word);
                            const int incl = sgnn_row_incl_scan(pc);
                            const uint32_t over = (uint32_t)(__ballot(incl > rem) >> gshift) & 0xffffu;
                            // the row's total = its lane 15: the four rows' totals read into scalars, each lane takes its row's
                            const int t0_ = __builtin_amdgcn_readlane(incl, 15), t1_ = __builtin_amdgcn_readlane(incl, 31);
                            const int t2_ = __builtin_amdgcn_readlane(incl, 47), t3_ = __builtin_amdgcn_readlane(incl, 63);
                            const int tot16 = gshift == 0 ? t0_ : (gshift == 16 ? t1_ : (gshift == 32 ? t2_ : t3_));
                            if (!found && over) {
                                const int L = __ffs((int)over) - 1;
                                if (mine && gl == L) {                                        // this lane's word holds the bit: it answers
                                    smp.anchor[o] = lo_id + (int64_t)w * 32 + k1_nth_set_bit(word, rem - (incl - pc));
                                    smp.allneg[o] = an;
                                    smp.hop[o] = 1;
                                }
                                found = true;
                            }
                            if (!found) rem -= tot16;
                        }
                        if (active && cnt == 0 && sl == 0 && gl == 0) { smp.anchor[o] = 0; smp.allneg[o] = 1; smp.hop[o] = 0; }
                    }
                }
                // ---- wipe for the next slice / set ------------------------------------------------------
                k1_lds_barrier();
                K1_T(4);                                           // draw
#ifndef K1_DEBUG_SKIP_WIPE
                {
                    int4* bm4 = reinterpret_cast<int4*>(s_bm);
                    const int n4 = (int)((words + 3) / 4);
                    for (int i = tid; i < n4; i += K1_THREADS) bm4[i] = make_int4(0, 0, 0, 0);
                }
#endif
                k1_lds_barrier();
                K1_T(5);                                           // wipe
            }
        }
    }
}

// The rank pass gives every thread a run of `run` words (odd: conflict-free strides) and reads them without bounds tests,
// so the bitmap is allocated as run x K1_THREADS words (the padding is never set and stays zero).
static inline int64_t k1_run(int64_t words) { return ((words + K1_THREADS - 1) / K1_THREADS) | 1; }
static inline int64_t k1_alloc_bytes(int64_t slice_ids) { return k1_run((slice_ids + 31) / 32) * K1_THREADS * 4; }
#define K1_LDS_MAX (155 * 1024)          // dynamic LDS the kernel may take (4.2 KB of static tables on top: 160 KB per CU)

static int k1_plan(int64_t max_id, int64_t lds_budget, int64_t* slice_ids, int* n_slices)
{
    const int64_t ids = max_id + 1;
    int64_t cap = (lds_budget / 16) * 16 * 8;                            // ids one slice can hold
    const int64_t run_max = ((K1_LDS_MAX / (K1_THREADS * 4)) - 1) | 1;   // largest odd run whose padded bitmap fits
    if (cap > run_max * K1_THREADS * 32) cap = run_max * K1_THREADS * 32;
    if (cap < 128) return 0;
    int64_t ns = (ids + cap - 1) / cap;
    if (ns > K1_MAX_SLICES) return 0;
    int64_t per = ((ids + ns - 1) / ns + 127) / 128 * 128;
    if (per > cap) per = cap;
    *slice_ids = per;
    *n_slices = (int)((ids + per - 1) / per);
    return *n_slices <= K1_MAX_SLICES;
}

static bool k1_applies(int64_t max_id, int k, int rows_sorted, int bitmap_in_lds, int64_t* slice_ids, int* n_slices)
{
    if (k != 1 || !bitmap_in_lds) return false;
    const int64_t budget = bitmap_in_lds > 1 ? (bitmap_in_lds < KB_LDS_BYTES ? bitmap_in_lds : KB_LDS_BYTES) : KB_LDS_BYTES;
    return k1_plan(max_id, budget, slice_ids, n_slices) && (*n_slices == 1 || rows_sorted);
}

extern "C" int64_t sgnn_khop_border_sample_workspace_bytes(int64_t max_id, int64_t n_sets, int k, int rows_sorted,
                                                           int bitmap_in_lds)
{
    int64_t slice_ids = 0;
    int n_slices = 0;
    if (k1_applies(max_id, k, rows_sorted, bitmap_in_lds, &slice_ids, &n_slices)) return 16;     // the set counter
    return sgnn_khop_border_workspace_bytes(max_id, n_sets, bitmap_in_lds && kb_fits_lds(max_id) ? 1 : 0);
}

// The reference's PAD rule applied to the drawn border anchors (anchor_patch_samplers.py:189-191: the padded border matrix
// holds 0 in its PAD columns, so PAD wins a slot when every real variate is negative AND the row has a PAD column, i.e. its
// border is smaller than the matrix's width = the largest border) and the hop level as the slot's similarity (0 on PAD):
// one pass instead of eleven element-wise launches.  width: DEVICE scalar (the caller may have MAX-reduced it over ranks).
__global__ __launch_bounds__(256) void khop_sample_finish_kernel(int64_t* __restrict__ anchor, const uint8_t* __restrict__ hop,
                                                                 const uint8_t* __restrict__ allneg, const int64_t* __restrict__ counts,
                                                                 const int64_t* __restrict__ width, int64_t n_sets, int64_t n_slots,
                                                                 float* __restrict__ sims)
{
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= n_sets * n_slots) return;
    const int64_t r = t / n_slots;
    int64_t a = anchor[t];
    if (allneg[t] != 0 && counts[r] < width[0]) { a = 0; anchor[t] = 0; }
    sims[t] = a == 0 ? 0.f : (float)hop[t];
}

extern "C" int sgnn_khop_sample_finish(int64_t* anchor, const uint8_t* hop, const uint8_t* allneg, const int64_t* counts,
                                       const int64_t* width, int64_t n_sets, int64_t n_slots, float* sims, void* stream)
{
    if (!anchor || !hop || !allneg || !counts || !width || !sims || n_sets < 0 || n_slots < 0) return SGNN_ERR_BAD_ARG;
    const int64_t total = n_sets * n_slots;
    if (total == 0) return SGNN_OK;
    if ((total + 255) / 256 > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(khop_sample_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, anchor, hop,
                       allneg, counts, width, n_sets, n_slots, sims);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_khop_border_sample(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz,
                                       int64_t max_id,
                                       const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, int k,
                                       int64_t n_slots, uint64_t seed, uint64_t stream_id, int64_t item_base,
                                       int64_t* out_anchor, uint8_t* out_hop, uint8_t* out_allneg, int64_t* out_count,
                                       const int32_t* set_order,
                                       void* workspace, int64_t workspace_bytes, int bitmap_in_lds, void* stream)
{
    if (!out_anchor || !out_hop || !out_allneg || !out_count || n_slots < 1 || item_base < 0) return SGNN_ERR_BAD_ARG;
    KbSample smp = {n_slots, sgnn_tape_h0(seed, stream_id), out_anchor, out_hop, out_allneg, item_base};
    {
        // the specialised one-hop kernel; bitmap_in_lds > 1 = LDS bytes the bitmap may take (tests: forces slicing)
        int64_t slice_ids = 0;
        int n_slices = 0;
        if (k1_applies(max_id, k, col_sorted != nullptr, bitmap_in_lds, &slice_ids, &n_slices)) {
            if (!rowptr || !col || !set_ptr || !set_nodes || !workspace || n_sets < 0) return SGNN_ERR_BAD_ARG;
            if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
            if (workspace_bytes < 16) return SGNN_ERR_BAD_ARG;
            if (n_sets == 0) return SGNN_OK;
            hipStream_t st = (hipStream_t)stream;
            // the set counter: the last 8-byte aligned 8 bytes of the workspace
            unsigned long long* next_set = (unsigned long long*)((char*)workspace + ((workspace_bytes - 8) & ~(int64_t)7));
            { const hipError_t me = hipMemsetAsync(next_set, 0, 8, st); if (me != hipSuccess) { sgnn_set_last_error(me); return SGNN_ERR_LAUNCH; } }
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)khop1_sample_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, K1_LDS_MAX);
                (void)hipFuncSetAttribute((const void*)khop1_sample_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, K1_LDS_MAX);
                attr_set = true;
            }
            const int64_t nwg = kb_n_wg(n_sets, true);
            const size_t lds = (size_t)k1_alloc_bytes(slice_ids);
            if (n_slices == 1)
                hipLaunchKernelGGL(khop1_sample_kernel<false>, dim3((int)nwg), dim3(K1_THREADS), lds, st, rowptr, col, max_id,
                                   set_ptr, set_nodes, n_sets, out_count, smp, slice_ids, 1, next_set, set_order);
            else
                hipLaunchKernelGGL(khop1_sample_kernel<true>, dim3((int)nwg), dim3(K1_THREADS), lds, st, rowptr, col_sorted, max_id,
                                   set_ptr, set_nodes, n_sets, out_count, smp, slice_ids, n_slices, next_set, set_order);
            SGNN_CHECK_LAUNCH();
            return SGNN_OK;
        }
    }
    bitmap_in_lds = bitmap_in_lds && kb_fits_lds(max_id) ? 1 : 0;      // general kernel; beyond the LDS bitmap: bitmap in L2
    return kb_launch(rowptr, col, nnz, max_id, set_ptr, set_nodes, n_sets, k, 0, out_count, nullptr, nullptr, nullptr,
                     workspace, workspace_bytes, bitmap_in_lds, smp, stream, 0, set_order);
}

// ---------------------------------------------------------------------------------------------
// a3  in-border nodes of a patch (reference SubGNN/subgraph_utils.py:126-144, with the id-1 /
// node-order indexing quirk: id x is read as the node at position x-1 of G.nodes()).
// One workgroup per patch; patch ids in an LDS hash; one thread per member walks "its" list.
// ---------------------------------------------------------------------------------------------
#define PB_HASH_BITS 12
#define PB_HASH (1 << PB_HASH_BITS)
#define PB_MAX 2048

__global__ __launch_bounds__(256) void patch_in_border_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ node_order, const int32_t* __restrict__ node_pos,
    const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ patch_nodes, int64_t n_patches,
    uint8_t* __restrict__ out_flag)
{
    __shared__ int32_t hash[PB_HASH];
    const int tid = threadIdx.x;
    for (int64_t p = blockIdx.x; p < n_patches; p += gridDim.x) {
        const int64_t beg = patch_ptr[p];
        const int n = (int)(patch_ptr[p + 1] - beg);
        if (n <= 0) continue;
        for (int i = tid; i < PB_HASH; i += 256) hash[i] = 0;
        __syncthreads();
        if (n <= PB_MAX) {
            for (int i = tid; i < n; i += 256) {
                const int32_t v = patch_nodes[beg + i];
                uint32_t h = sgnn_hash32((uint32_t)v) >> (32 - PB_HASH_BITS);
                while (true) {
                    const int32_t old = atomicCAS(&hash[h], 0, v);
                    if (old == 0 || old == v) break;
                    h = (h + 1) & (PB_HASH - 1);
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            if (n > PB_MAX) { out_flag[beg + i] = 255; continue; }     // unsupported size: poisoned
            const int32_t x = patch_nodes[beg + i];
            const int32_t px = node_order[x - 1];
            const int64_t r0 = rowptr[px], r1 = rowptr[px + 1];
            uint8_t flag = 0;
            for (int64_t e = r0; e < r1 && !flag; ++e) {
                const int32_t y = node_pos[col[e]] + 1;
                uint32_t h = sgnn_hash32((uint32_t)y) >> (32 - PB_HASH_BITS);
                bool member = false;
                while (true) {
                    const int32_t kk = hash[h];
                    if (kk == y) { member = true; break; }
                    if (kk == 0) break;
                    h = (h + 1) & (PB_HASH - 1);
                }
                if (!member) flag = 1;
            }
            out_flag[beg + i] = flag;
        }
        __syncthreads();
    }
}

extern "C" int sgnn_patch_in_border(const int64_t* rowptr, const int32_t* col, int64_t nnz,
                                    const int32_t* node_order, const int32_t* node_pos, int64_t n_nodes,
                                    const int64_t* patch_ptr, const int32_t* patch_nodes, int64_t n_patches,
                                    uint8_t* out_flag, void* stream)
{
    if (!rowptr || !col || !node_order || !node_pos || !patch_ptr || !patch_nodes || !out_flag || n_patches < 0)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    (void)n_nodes;
    if (n_patches == 0) return SGNN_OK;
    const int grid = (int)(n_patches < 256 * 8 ? n_patches : 256 * 8);
    hipLaunchKernelGGL(patch_in_border_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                       node_order, node_pos, patch_ptr, patch_nodes, n_patches, out_flag);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// Patches of more than PB_MAX nodes (ego-graph patches around hubs): the membership table lives in the caller's workspace
// (4 int32 slots per patch node at the patch's own offset) instead of LDS; the kernel above poisons those patches' flags
// with 255 and this one overwrites them.  workspace: sgnn_patch_in_border_huge_workspace_bytes(total patch nodes).
extern "C" int64_t sgnn_patch_in_border_huge_workspace_bytes(int64_t total_nodes)
{
    return (total_nodes < 0 ? 0 : total_nodes) * 4 * 4 + 64;
}

__global__ __launch_bounds__(256) void patch_in_border_huge_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ node_order, const int32_t* __restrict__ node_pos,
    const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ patch_nodes, int64_t n_patches,
    uint8_t* __restrict__ out_flag, int32_t* __restrict__ ws)
{
    const int tid = threadIdx.x;
    for (int64_t p = blockIdx.x; p < n_patches; p += gridDim.x) {
        const int64_t beg = patch_ptr[p];
        const int n = (int)(patch_ptr[p + 1] - beg);
        if (n <= PB_MAX) continue;                              // sgnn_patch_in_border owns those
        int32_t* hash = ws + 4 * beg;
        uint32_t H = 1;
        while (H < 2u * (uint32_t)n) H <<= 1;
        for (uint32_t i = tid; i < H; i += 256) hash[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const int32_t v = patch_nodes[beg + i];
            uint32_t h = sgnn_hash32((uint32_t)v) & (H - 1);
            while (true) {
                const int32_t old = atomicCAS(&hash[h], 0, v);
                if (old == 0 || old == v) break;
                h = (h + 1) & (H - 1);
            }
        }
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const int32_t x = patch_nodes[beg + i];
            const int32_t px = node_order[x - 1];               // the reference's id - 1 / node-order quirk (su:139)
            const int64_t r0 = rowptr[px], r1 = rowptr[px + 1];
            uint8_t flag = 0;
            for (int64_t e = r0; e < r1 && !flag; ++e) {
                const int32_t y = node_pos[col[e]] + 1;
                uint32_t h = sgnn_hash32((uint32_t)y) & (H - 1);
                bool member = false;
                while (true) {
                    const int32_t kk = hash[h];
                    if (kk == y) { member = true; break; }
                    if (kk == 0) break;
                    h = (h + 1) & (H - 1);
                }
                if (!member) flag = 1;
            }
            out_flag[beg + i] = flag;
        }
        __syncthreads();
    }
}

extern "C" int sgnn_patch_in_border_huge(const int64_t* rowptr, const int32_t* col, int64_t nnz,
                                         const int32_t* node_order, const int32_t* node_pos, int64_t n_nodes,
                                         const int64_t* patch_ptr, const int32_t* patch_nodes, int64_t n_patches,
                                         int64_t total_nodes, uint8_t* out_flag, void* workspace, int64_t workspace_bytes,
                                         void* stream)
{
    if (!rowptr || !col || !node_order || !node_pos || !patch_ptr || !patch_nodes || !out_flag || !workspace || n_patches < 0 ||
        total_nodes < 0)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (total_nodes >= (1ll << 28)) return SGNN_ERR_SET_TOO_LARGE;
    if (workspace_bytes < sgnn_patch_in_border_huge_workspace_bytes(total_nodes)) return SGNN_ERR_BAD_ARG;
    (void)n_nodes;
    if (n_patches == 0) return SGNN_OK;
    hipLaunchKernelGGL(patch_in_border_huge_kernel, dim3((int)(n_patches < 1024 ? n_patches : 1024)), dim3(256), 0,
                       (hipStream_t)stream, rowptr, col, node_order, node_pos, patch_ptr, patch_nodes, n_patches, out_flag,
                       (int32_t*)workspace);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}


SGNN_DEFINE_WARM(graph_sets)
