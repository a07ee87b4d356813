// Anchor-patch samplers driven by the counter-based draw tape (a1-a6).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// a4  neighbourhood anchors (reference SubGNN/anchor_patch_samplers.py:163-198)
// One wavefront per row; for each slot the lanes scan the row's columns, hash (row, slot, id)
// into a signed 53-bit key (PAD columns hold key 0), and a butterfly reduction picks the
// maximum key, the smallest column winning ties (torch.argmax returns the first maximum).
// VALU-bound (two 64-bit multiplies-mix rounds per element), no memory traffic beyond the ids.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_anchors_padded_kernel(
    const int64_t* __restrict__ ids, int64_t n_rows, int64_t L, int64_t n_slots,
    uint64_t h0, int64_t* __restrict__ out)
{
    const int lane = threadIdx.x;
    for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int64_t* row = ids + r * L;
        for (int64_t i = 0; i < n_slots; ++i) {
            const uint64_t h1 = sgnn_tape_h1(h0, (uint64_t)(r * n_slots + i));
            int64_t best = INT64_MIN;
            int32_t bcol = INT32_MAX, bid = 0;
            for (int64_t c = lane; c < L; c += 64) {
                const int64_t v = row[c];
                const int64_t key = (v == 0) ? 0 : sgnn_symmetric_key(h1, (uint64_t)v);
                if (key > best) { best = key; bcol = (int32_t)c; bid = (int32_t)v; }
            }
            sgnn_argmax_reduce(best, bcol, bid);
            if (lane == 0) out[r * n_slots + i] = (L > 0) ? (int64_t)bid : 0;
        }
    }
}

#define SA_SC 8               // anchor slots hashed per pass over a row
#define SA_WAVES 4            // rows in flight per 256-thread workgroup

__global__ __launch_bounds__(64 * SA_WAVES) void sample_anchors_ragged_kernel(
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    const uint8_t* __restrict__ row_has_pad, int64_t n_slots, uint64_t h0, int64_t* __restrict__ out,
    int64_t* __restrict__ out_pos)
{
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * SA_WAVES + (threadIdx.x >> 6); r < n_sets; r += (int64_t)gridDim.x * SA_WAVES) {
        const int64_t beg = set_ptr[r];
        const int64_t n = set_ptr[r + 1] - beg;
        const bool has_pad = row_has_pad ? (row_has_pad[r] != 0) : true;
        for (int64_t s0 = 0; s0 < n_slots; s0 += SA_SC) {
            uint64_t h1[SA_SC];
            int64_t best[SA_SC];
            int32_t bcol[SA_SC], bid[SA_SC];
#pragma unroll
            for (int u = 0; u < SA_SC; ++u) {
                h1[u] = sgnn_tape_h1(h0, (uint64_t)(r * n_slots + s0 + u));
                best[u] = INT64_MIN; bcol[u] = INT32_MAX; bid[u] = 0;
            }
            for (int64_t c = lane; c < n; c += 64) {       // one read of the row per SA_SC slots
                const int32_t v = set_nodes[beg + c];
#pragma unroll
                for (int u = 0; u < SA_SC; ++u) {
                    const int64_t key = (v == 0) ? 0 : sgnn_symmetric_key(h1[u], (uint64_t)v);
                    if (key > best[u]) { best[u] = key; bcol[u] = (int32_t)c; bid[u] = v; }
                }
            }
#pragma unroll
            for (int u = 0; u < SA_SC; ++u) {
                sgnn_argmax_reduce(best[u], bcol[u], bid[u]);
                // the PAD columns of the padded row sit after the real ones with key 0
                if (n == 0 || (has_pad && best[u] < 0)) { bid[u] = 0; bcol[u] = -1; }
                if (lane == 0 && s0 + u < n_slots) {
                    out[r * n_slots + s0 + u] = (int64_t)bid[u];
                    if (out_pos) out_pos[r * n_slots + s0 + u] = (bid[u] == 0) ? -1 : beg + bcol[u];
                }
            }
        }
    }
}

extern "C" int sgnn_sample_anchors_padded(const int64_t* ids, int64_t n_rows, int64_t L, int64_t n_slots,
                                          uint64_t seed, uint64_t stream_id, int64_t* out, void* stream)
{
    if (!ids || !out || n_rows < 0 || L < 0 || n_slots < 0) return SGNN_ERR_BAD_ARG;
    if (n_rows == 0 || n_slots == 0) return SGNN_OK;
    const int grid = (int)(n_rows < 256 * 32 ? n_rows : 256 * 32);
    hipLaunchKernelGGL(sample_anchors_padded_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, ids, n_rows, L,
                       n_slots, sgnn_tape_h0(seed, stream_id), out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_sample_anchors_ragged(const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                          const uint8_t* row_has_pad, int64_t n_slots,
                                          uint64_t seed, uint64_t stream_id, int64_t* out, int64_t* out_pos,
                                          void* stream)
{
    if (!set_ptr || !set_nodes || !out || n_sets < 0 || n_slots < 0) return SGNN_ERR_BAD_ARG;
    if (n_sets == 0 || n_slots == 0) return SGNN_OK;
    const int64_t want = (n_sets + SA_WAVES - 1) / SA_WAVES;
    const int grid = (int)(want < 256 * 8 ? want : 256 * 8);
    hipLaunchKernelGGL(sample_anchors_ragged_kernel, dim3(grid), dim3(64 * SA_WAVES), 0, (hipStream_t)stream, set_ptr,
                       set_nodes, n_sets, row_has_pad, n_slots, sgnn_tape_h0(seed, stream_id), out, out_pos);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// a5/a6  np.random.choice(seq, n, replace=True) (reference anchor_patch_samplers.py:206,208,326)
// ---------------------------------------------------------------------------------------------
__global__ void choice_ragged_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ seq,
                                     int64_t n_items, int64_t n_draws, uint64_t h0, int64_t* __restrict__ out)
{
    const int64_t total = n_items * n_draws;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n_draws, j = t % n_draws;
        const int64_t beg = ptr[r];
        const int64_t n = ptr[r + 1] - beg;
        int64_t v = 0;
        if (n > 0) v = seq[beg + sgnn_choice_index(sgnn_tape_h1(h0, (uint64_t)r), (uint64_t)j, (uint32_t)n)];
        out[t] = v;
    }
}

extern "C" int sgnn_choice_ragged(const int64_t* ptr, const int32_t* seq, int64_t n_items, int64_t n_draws,
                                  uint64_t seed, uint64_t stream_id, int64_t* out, void* stream)
{
    if (!ptr || !seq || !out || n_items < 0 || n_draws < 0) return SGNN_ERR_BAD_ARG;
    if (n_items * n_draws == 0) return SGNN_OK;
    hipLaunchKernelGGL(choice_ragged_kernel, dim3(sgnn_grid_for(n_items * n_draws, 256)), dim3(256), 0,
                       (hipStream_t)stream, ptr, seq, n_items, n_draws, sgnn_tape_h0(seed, stream_id), out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// a1-a3  triangular random walks (reference SubGNN/anchor_patch_samplers.py:20-158, 210-243)
// One WAVEFRONT per walk.  A walk is a serial dependent chain (each step needs the previous
// node), so walks run in parallel across wavefronts, and inside a step the 64 lanes share the
// work that dominates on hub nodes: the neighbour list of the current node is streamed in
// networkx order 64 entries at a time (coalesced), every lane classifies its entry (valid for
// the mode?  adjacent to the previous node? -- binary search in that node's sorted list), and
// wavefront ballots + popcounts give the triangle / non-triangle counts (pass 1) and the
// position of the drawn candidate (pass 2).  The walk state (tape counter, prev, curr) is
// wave-uniform.  Latency-bound integer work; the walks are few (patches x walks).
// ---------------------------------------------------------------------------------------------
struct WalkCtx {
    const int64_t* rowptr;
    const int32_t* col;
    const int32_t* col_sorted;
    const int32_t* patch;     // patch node view (mode 1/2), unique ids
    int32_t n_patch;
    const int32_t* inb;       // in-border nodes (mode 2)
    int32_t n_inb;
    int mode;
};

__device__ static inline bool walk_in_list(const int32_t* a, int32_t n, int32_t v) {
    for (int32_t i = 0; i < n; ++i)
        if (a[i] == v) return true;
    return false;
}

__device__ static inline bool walk_valid(const WalkCtx& c, int32_t v) {
    if (c.mode == 0) return true;
    const bool member = walk_in_list(c.patch, c.n_patch, v);
    if (c.mode == 1) return member;
    return !member || walk_in_list(c.inb, c.n_inb, v);       // aps:143
}

// pass 1: counts of valid neighbours of v that are / are not adjacent to prev (prev = 0: no test)
__device__ static inline void walk_count(const WalkCtx& c, int32_t v, int32_t prev, int lane, int32_t& nt, int32_t& nn) {
    const int64_t r0 = c.rowptr[v], r1 = c.rowptr[v + 1];
    int64_t p0 = 0;
    int32_t pdeg = 0;
    if (prev) { p0 = c.rowptr[prev]; pdeg = (int32_t)(c.rowptr[prev + 1] - p0); }
    nt = 0; nn = 0;
    for (int64_t base = r0; base < r1; base += 64) {
        const int64_t e = base + lane;
        bool ok = false, tri = false;
        if (e < r1) {
            const int32_t w = c.col[e];
            ok = walk_valid(c, w);
            if (ok && prev) tri = sgnn_sorted_contains(c.col_sorted + p0, pdeg, w);
        }
        nt += __popcll(__ballot(ok && tri));
        nn += __popcll(__ballot(ok && !tri));
    }
}

// pass 2: the pick-th (0-based, adjacency order) valid neighbour of v in the wanted class
__device__ static inline int32_t walk_pick(const WalkCtx& c, int32_t v, int32_t prev, bool want_tri, int32_t pick, int lane) {
    const int64_t r0 = c.rowptr[v], r1 = c.rowptr[v + 1];
    int64_t p0 = 0;
    int32_t pdeg = 0;
    if (prev) { p0 = c.rowptr[prev]; pdeg = (int32_t)(c.rowptr[prev + 1] - p0); }
    int32_t seen = 0;
    for (int64_t base = r0; base < r1; base += 64) {
        const int64_t e = base + lane;
        bool hit = false;
        int32_t w = 0;
        if (e < r1) {
            w = c.col[e];
            if (walk_valid(c, w)) {
                const bool tri = prev ? sgnn_sorted_contains(c.col_sorted + p0, pdeg, w) : false;
                hit = (tri == want_tri);
            }
        }
        const uint64_t m = __ballot(hit);
        const int32_t cnt = __popcll(m);
        if (seen + cnt > pick) {
            const int32_t rank = __popcll(m & ((1ull << lane) - 1ull));
            const uint64_t sel = __ballot(hit && (seen + rank == pick));
            const int src = __ffsll((unsigned long long)sel) - 1;
            return __shfl(w, src);
        }
        seen += cnt;
    }
    return 0;
}

__global__ __launch_bounds__(64) void triangular_walks_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ col_sorted,
    const int32_t* __restrict__ node_order, int64_t n_nodes,
    const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ patch_nodes,
    const int64_t* __restrict__ inb_ptr, const int32_t* __restrict__ inb_nodes,
    int mode, int64_t n_items, int64_t walks_per_patch, int64_t walk_len, double beta,
    uint64_t h0, int64_t* __restrict__ out)
{
    const int lane = threadIdx.x;
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        int64_t* o = out + item * walk_len;
        for (int64_t t = lane; t < walk_len; t += 64) o[t] = 0;
        WalkCtx c;
        c.rowptr = rowptr; c.col = col; c.col_sorted = col_sorted; c.mode = mode;
        c.patch = nullptr; c.n_patch = 0; c.inb = nullptr; c.n_inb = 0;
        const uint64_t h1 = sgnn_tape_h1(h0, (uint64_t)item);
        uint64_t j = 0;
        int32_t prev;
        if (mode == 0) {
            prev = node_order[sgnn_choice_index(h1, j++, (uint32_t)n_nodes)];          // aps:70
        } else {
            const int64_t p = item / walks_per_patch;
            c.patch = patch_nodes + patch_ptr[p];
            c.n_patch = (int32_t)(patch_ptr[p + 1] - patch_ptr[p]);
            if (c.n_patch == 0) continue;                                              // aps:134-135
            if (mode == 1) {
                prev = c.patch[sgnn_choice_index(h1, j++, (uint32_t)c.n_patch)];      // aps:70
            } else {
                c.inb = inb_nodes + inb_ptr[p];
                c.n_inb = (int32_t)(inb_ptr[p + 1] - inb_ptr[p]);
                if (c.n_inb == 0) continue;        // reference raises ValueError here (aps:78)
                prev = c.inb[sgnn_choice_index(h1, j++, (uint32_t)c.n_inb)];          // aps:78
            }
        }
        if (walk_len < 1) continue;
        __syncthreads();                            // zero fill above precedes the lane-0 writes
        if (lane == 0) o[0] = prev;
        int32_t nt, nn;
        walk_count(c, prev, 0, lane, nt, nn);                                            // aps:72,79
        if (nn == 0 || walk_len < 2) continue;                                          // aps:83-84
        int32_t curr = walk_pick(c, prev, 0, false, (int32_t)sgnn_choice_index(h1, j++, (uint32_t)nn), lane);   // aps:74,80
        if (lane == 0) o[1] = curr;
        for (int64_t step = 2; step < walk_len; ++step) {
            walk_count(c, curr, prev, lane, nt, nn);                                     // aps:35-45
            if (nt + nn == 0) break;                                                     // aps:94
            bool want_tri;
            if (nt == 0) want_tri = false;                                               // aps:97-98
            else if (nn == 0) want_tri = true;                                           // aps:99-100
            else want_tri = (sgnn_uniform01(h1, j++) <= beta);                           // aps:102
            const int32_t pick = (int32_t)sgnn_choice_index(h1, j++, (uint32_t)(want_tri ? nt : nn));
            const int32_t nxt = walk_pick(c, curr, prev, want_tri, pick, lane);
            prev = curr;
            curr = nxt;
            if (lane == 0) o[step] = nxt;
        }
    }
}

extern "C" int sgnn_triangular_walks(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz,
                                     const int32_t* node_order, int64_t n_nodes,
                                     const int64_t* patch_ptr, const int32_t* patch_nodes,
                                     const int64_t* inb_ptr, const int32_t* inb_nodes,
                                     int mode, int64_t n_items, int64_t walks_per_patch, int64_t walk_len, double beta,
                                     uint64_t seed, uint64_t stream_id, int64_t* out, void* stream)
{
    if (!rowptr || !col || !col_sorted || !out || n_items < 0 || walk_len < 0 || mode < 0 || mode > 2)
        return SGNN_ERR_BAD_ARG;
    if (mode == 0 && (!node_order || n_nodes <= 0)) return SGNN_ERR_BAD_ARG;
    if (mode >= 1 && (!patch_ptr || !patch_nodes || walks_per_patch <= 0)) return SGNN_ERR_BAD_ARG;
    if (mode == 2 && (!inb_ptr || !inb_nodes)) return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (n_items == 0 || walk_len == 0) return SGNN_OK;
    hipLaunchKernelGGL(triangular_walks_kernel, dim3((int)(n_items < 256 * 32 ? n_items : 256 * 32)), dim3(64), 0, (hipStream_t)stream,
                       rowptr, col, col_sorted, node_order, n_nodes, patch_ptr, patch_nodes, inb_ptr, inb_nodes,
                       mode, n_items, walks_per_patch, walk_len, beta, sgnn_tape_h0(seed, stream_id), out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}
