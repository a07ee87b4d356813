// Anchor-patch samplers driven by the counter-based draw tape (a1-a6).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// a4  neighbourhood anchors (reference SubGNN/anchor_patch_samplers.py:163-198) under the tape's
// neighbourhood-anchor law (common.h): per (row, slot) one index draw into the row's non-PAD
// entries in ascending order, one "all variates negative" draw for the PAD rule.  The rows are
// taken in canonical form -- ascending, PADs last (the host wrapper sorts; the fused border kernel
// answers the same rank query from its visited bitmap without ever sorting) -- so a slot is O(1).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_anchors_padded_kernel(
    const int64_t* __restrict__ ids, int64_t n_rows, int64_t L, int64_t n_slots,
    uint64_t h0, int64_t* __restrict__ out)
{
    const int64_t total = n_rows * n_slots;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n_slots;
        const int64_t* row = ids + r * L;
        int64_t lo = 0, hi = L;                               // n = first PAD column of the canonical row
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (row[mid] != 0) lo = mid + 1; else hi = mid; }
        const int64_t n = lo;
        int64_t v = 0;
        if (n > 0) {
            const uint64_t h1 = sgnn_tape_h1(h0, (uint64_t)t);
            if (!(n < L && sgnn_nanchor_allneg(h1, (uint32_t)n))) v = row[sgnn_nanchor_index(h1, (uint32_t)n)];
        }
        out[t] = v;
    }
}

__global__ __launch_bounds__(256) void sample_anchors_ragged_kernel(
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    const uint8_t* __restrict__ row_has_pad, int64_t n_slots, uint64_t h0, int64_t item_base, int64_t* __restrict__ out)
{
    const int64_t total = n_sets * n_slots;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n_slots;
        const int64_t beg = set_ptr[r];
        const int64_t n = set_ptr[r + 1] - beg;
        const bool has_pad = row_has_pad ? (row_has_pad[r] != 0) : true;
        int64_t v = 0;
        if (n > 0) {
            const uint64_t h1 = sgnn_tape_h1(h0, (uint64_t)(t + item_base * n_slots));
            if (!(has_pad && sgnn_nanchor_allneg(h1, (uint32_t)n))) v = set_nodes[beg + sgnn_nanchor_index(h1, (uint32_t)n)];
        }
        out[t] = v;
    }
}

extern "C" int sgnn_sample_anchors_padded(const int64_t* ids, int64_t n_rows, int64_t L, int64_t n_slots,
                                          uint64_t seed, uint64_t stream_id, int64_t* out, void* stream)
{
    if (!ids || !out || n_rows < 0 || L < 0 || n_slots < 0) return SGNN_ERR_BAD_ARG;
    if (L >= (1ll << 32)) return SGNN_ERR_SET_TOO_LARGE;
    if (n_rows == 0 || n_slots == 0) return SGNN_OK;
    hipLaunchKernelGGL(sample_anchors_padded_kernel, dim3(sgnn_grid_for(n_rows * n_slots, 256)), dim3(256), 0,
                       (hipStream_t)stream, ids, n_rows, L, n_slots, sgnn_tape_h0(seed, stream_id), out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_sample_anchors_ragged(const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                          const uint8_t* row_has_pad, int64_t n_slots,
                                          uint64_t seed, uint64_t stream_id, int64_t item_base, int64_t* out, void* stream)
{
    if (!set_ptr || !set_nodes || !out || n_sets < 0 || n_slots < 0 || item_base < 0) return SGNN_ERR_BAD_ARG;
    if (n_sets == 0 || n_slots == 0) return SGNN_OK;
    hipLaunchKernelGGL(sample_anchors_ragged_kernel, dim3(sgnn_grid_for(n_sets * n_slots, 256)), dim3(256), 0,
                       (hipStream_t)stream, set_ptr, set_nodes, n_sets, row_has_pad, n_slots,
                       sgnn_tape_h0(seed, stream_id), item_base, out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// a5/a6  np.random.choice(seq, n, replace=True) (reference anchor_patch_samplers.py:206,208,326)
// ---------------------------------------------------------------------------------------------
__global__ void choice_ragged_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ seq,
                                     int64_t n_items, int64_t n_draws, uint64_t h0, int64_t item_base,
                                     int64_t* __restrict__ out)
{
    const int64_t total = n_items * n_draws;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n_draws, j = t % n_draws;
        const int64_t beg = ptr[r];
        const int64_t n = ptr[r + 1] - beg;
        int64_t v = 0;
        if (n > 0) v = seq[beg + sgnn_choice_index(sgnn_tape_h1(h0, (uint64_t)(r + item_base)), (uint64_t)j, (uint32_t)n)];
        out[t] = v;
    }
}

extern "C" int sgnn_choice_ragged(const int64_t* ptr, const int32_t* seq, int64_t n_items, int64_t n_draws,
                                  uint64_t seed, uint64_t stream_id, int64_t item_base, int64_t* out, void* stream)
{
    if (!ptr || !seq || !out || n_items < 0 || n_draws < 0 || item_base < 0) return SGNN_ERR_BAD_ARG;
    if (n_items * n_draws == 0) return SGNN_OK;
    hipLaunchKernelGGL(choice_ragged_kernel, dim3(sgnn_grid_for(n_items * n_draws, 256)), dim3(256), 0,
                       (hipStream_t)stream, ptr, seq, n_items, n_draws, sgnn_tape_h0(seed, stream_id), item_base, out);
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
#define WK_HASH_BITS 8
#define WK_HASH (1 << WK_HASH_BITS)
#define WK_HASH_MAX 96          // patches / in-border sets up to this size use the LDS hash (else linear scan)
#define WK_CHUNKS 1024          // classification masks cached for neighbour lists up to 64 * WK_CHUNKS entries

struct WalkCtx {
    const int64_t* rowptr;
    const int32_t* col;
    const int32_t* col_sorted;
    const int32_t* patch;     // patch node view (mode 1/2), unique ids
    int32_t n_patch;
    const int32_t* inb;       // in-border nodes (mode 2)
    int32_t n_inb;
    int mode;
    const int32_t* hpatch;    // LDS hash of the patch (nullptr: linear scan)
    const int32_t* hinb;      // LDS hash of the in-border set
    int pp, pi;               // wave-uniform probe counts of the two hashes
};

__device__ static inline bool walk_in_list(const int32_t* a, int32_t n, int32_t v) {
    for (int32_t i = 0; i < n; ++i)
        if (a[i] == v) return true;
    return false;
}

__device__ static inline bool walk_in_hash(const int32_t* h, int P, int32_t v) {
    const uint32_t b = sgnn_hash32((uint32_t)v) >> (32 - WK_HASH_BITS);
    int hit = 0;
    for (int p = 0; p < P; ++p) hit |= (h[(b + p) & (WK_HASH - 1)] == v);
    return hit != 0;
}

// wave-cooperative build of a membership hash; returns the longest probe chain (wave-uniform)
__device__ static inline int walk_build_hash(int32_t* h, const int32_t* a, int32_t n, int lane) {
#pragma unroll
    for (int q = 0; q < WK_HASH / 64; ++q) h[lane + 64 * q] = 0;
    __syncthreads();
    int chain = 0;
    for (int i = lane; i < n; i += 64) {
        const int32_t v = a[i];
        uint32_t b = sgnn_hash32((uint32_t)v) >> (32 - WK_HASH_BITS);
        int c = 0;
        while (true) {
            ++c;
            const int32_t old = atomicCAS(&h[b], 0, v);
            if (old == 0 || old == v) break;
            b = (b + 1) & (WK_HASH - 1);
        }
        chain = c > chain ? c : chain;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(chain, d); chain = o > chain ? o : chain; }
    __syncthreads();
    return chain;
}

__device__ static inline bool walk_valid(const WalkCtx& c, int32_t v) {
    if (c.mode == 0) return true;
    const bool member = c.hpatch ? walk_in_hash(c.hpatch, c.pp, v) : walk_in_list(c.patch, c.n_patch, v);
    if (c.mode == 1) return member;
    if (!member) return true;                                                        // aps:143
    return c.hinb ? walk_in_hash(c.hinb, c.pi, v) : walk_in_list(c.inb, c.n_inb, v);
}

// pass 1: classify the valid neighbours of v (adjacent to prev or not; prev = 0: no test), count both
// classes and keep the per-chunk ballots in LDS for pass 2 (lists longer than the cache: not kept)
__device__ static inline void walk_count(const WalkCtx& c, int32_t v, int32_t prev, int lane, int32_t& nt, int32_t& nn,
                                         uint64_t* s_tri, uint64_t* s_non) {
    const int64_t r0 = c.rowptr[v], r1 = c.rowptr[v + 1];
    int64_t p0 = 0;
    int32_t pdeg = 0;
    if (prev) { p0 = c.rowptr[prev]; pdeg = (int32_t)(c.rowptr[prev + 1] - p0); }
    nt = 0; nn = 0;
    int chunk = 0;
    for (int64_t base = r0; base < r1; base += 64, ++chunk) {
        const int64_t e = base + lane;
        bool ok = false, tri = false;
        if (e < r1) {
            const int32_t w = c.col[e];
            ok = walk_valid(c, w);
            if (ok && prev) tri = sgnn_sorted_contains(c.col_sorted + p0, pdeg, w);
        }
        const uint64_t mt = __ballot(ok && tri), mn = __ballot(ok && !tri);
        if (chunk < WK_CHUNKS && lane == 0) { s_tri[chunk] = mt; s_non[chunk] = mn; }
        nt += __popcll(mt);
        nn += __popcll(mn);
    }
}

// pass 2: the pick-th (0-based, adjacency order) valid neighbour of v in the wanted class, from the
// cached ballots (one load of col[]), or by re-classifying when the list exceeded the cache
__device__ static inline int32_t walk_pick(const WalkCtx& c, int32_t v, int32_t prev, bool want_tri, int32_t pick, int lane,
                                           const uint64_t* s_tri, const uint64_t* s_non) {
    const int64_t r0 = c.rowptr[v], r1 = c.rowptr[v + 1];
    const int64_t n_chunks = (r1 - r0 + 63) / 64;
    int32_t seen = 0;
    if (n_chunks <= WK_CHUNKS) {
        const uint64_t* s = want_tri ? s_tri : s_non;
        for (int64_t ch = 0; ch < n_chunks; ++ch) {
            uint64_t m = s[ch];
            const int32_t cnt = __popcll(m);
            if (seen + cnt > pick) {
                int k = pick - seen;                          // k-th set bit of m
                while (k-- > 0) m &= m - 1;
                return c.col[r0 + ch * 64 + (__ffsll((unsigned long long)m) - 1)];
            }
            seen += cnt;
        }
        return 0;
    }
    int64_t p0 = 0;
    int32_t pdeg = 0;
    if (prev) { p0 = c.rowptr[prev]; pdeg = (int32_t)(c.rowptr[prev + 1] - p0); }
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
    uint64_t h0, int64_t item_base, int64_t* __restrict__ out)
{
    __shared__ uint64_t s_tri[WK_CHUNKS], s_non[WK_CHUNKS];
    __shared__ int32_t s_hp[WK_HASH], s_hi[WK_HASH];
    const int lane = threadIdx.x;
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        int64_t* o = out + item * walk_len;
        for (int64_t t = lane; t < walk_len; t += 64) o[t] = 0;
        WalkCtx c;
        c.rowptr = rowptr; c.col = col; c.col_sorted = col_sorted; c.mode = mode;
        c.patch = nullptr; c.n_patch = 0; c.inb = nullptr; c.n_inb = 0;
        c.hpatch = nullptr; c.hinb = nullptr; c.pp = 0; c.pi = 0;
        const uint64_t h1 = sgnn_tape_h1(h0, (uint64_t)(item_base + item));   // the walk's GLOBAL number: a dealt share draws what the whole launch would
        uint64_t j = 0;
        int32_t prev;
        __syncthreads();                            // previous item's LDS tables are no longer read
        if (mode == 0) {
            prev = node_order[sgnn_choice_index(h1, j++, (uint32_t)n_nodes)];          // aps:70
        } else {
            const int64_t p = item / walks_per_patch;
            c.patch = patch_nodes + patch_ptr[p];
            c.n_patch = (int32_t)(patch_ptr[p + 1] - patch_ptr[p]);
            if (c.n_patch == 0) continue;                                              // aps:134-135
            if (c.n_patch <= WK_HASH_MAX) { c.pp = walk_build_hash(s_hp, c.patch, c.n_patch, lane); c.hpatch = s_hp; }
            if (mode == 1) {
                prev = c.patch[sgnn_choice_index(h1, j++, (uint32_t)c.n_patch)];      // aps:70
            } else {
                c.inb = inb_nodes + inb_ptr[p];
                c.n_inb = (int32_t)(inb_ptr[p + 1] - inb_ptr[p]);
                if (c.n_inb == 0) continue;        // reference raises ValueError here (aps:78)
                if (c.n_inb <= WK_HASH_MAX) { c.pi = walk_build_hash(s_hi, c.inb, c.n_inb, lane); c.hinb = s_hi; }
                prev = c.inb[sgnn_choice_index(h1, j++, (uint32_t)c.n_inb)];          // aps:78
            }
        }
        if (walk_len < 1) continue;
        __syncthreads();                            // zero fill above precedes the lane-0 writes
        if (lane == 0) o[0] = prev;
        int32_t nt, nn;
        walk_count(c, prev, 0, lane, nt, nn, s_tri, s_non);                              // aps:72,79
        if (nn == 0 || walk_len < 2) continue;                                          // aps:83-84
        __syncthreads();
        int32_t curr = walk_pick(c, prev, 0, false, (int32_t)sgnn_choice_index(h1, j++, (uint32_t)nn), lane, s_tri, s_non);   // aps:74,80
        if (lane == 0) o[1] = curr;
        for (int64_t step = 2; step < walk_len; ++step) {
            __syncthreads();                        // pass 2 of the previous step has read the masks
            walk_count(c, curr, prev, lane, nt, nn, s_tri, s_non);                       // aps:35-45
            if (nt + nn == 0) break;                                                     // aps:94
            bool want_tri;
            if (nt == 0) want_tri = false;                                               // aps:97-98
            else if (nn == 0) want_tri = true;                                           // aps:99-100
            else want_tri = (sgnn_uniform01(h1, j++) <= beta);                           // aps:102
            const int32_t pick = (int32_t)sgnn_choice_index(h1, j++, (uint32_t)(want_tri ? nt : nn));
            __syncthreads();
            const int32_t nxt = walk_pick(c, curr, prev, want_tri, pick, lane, s_tri, s_non);
            prev = curr;
            curr = nxt;
            if (lane == 0) o[step] = nxt;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Workgroup-per-walk variant (graphs whose id range fits an LDS bitmap, ~1.1 M ids).
// The wavefront-per-walk kernel above spends a step on a hub in (a) streaming the hub's list 64
// entries at a time on ONE wavefront and (b) a binary search in the previous node's sorted list
// per entry -- ~13 dependent global loads per 64 entries.  Here the 16 wavefronts of a 1024-thread
// workgroup share the list (chunks of 64 dealt round-robin), and "adjacent to prev?" is one bit
// test in an LDS bitmap that holds N(prev): when the walk moves on, the bits of the old N(prev)
// are un-set and those of the new one set by two more cooperative, coalesced streams.  The walks
// are few (patches x walks per patch), so giving each a whole CU costs nothing.
// Same tape, same draws, same results as the wavefront kernel (tests compare both).
// ---------------------------------------------------------------------------------------------
#define WKB_THREADS 1024
#define WKB_LDS_BYTES (136 * 1024)

__device__ static inline void wkb_bits(uint32_t* bm, const int32_t* __restrict__ list, int32_t n, bool set, int tid) {
    for (int32_t i = tid; i < n; i += WKB_THREADS) {
        const int32_t w = list[i];
        if (set) atomicOr(&bm[w >> 5], 1u << (w & 31)); else atomicAnd(&bm[w >> 5], ~(1u << (w & 31)));
    }
}

// workgroup-cooperative build of a membership hash; returns the longest probe chain (uniform)
__device__ static inline int wkb_build_hash(int32_t* h, const int32_t* a, int32_t n, int tid, int32_t* s_red) {
    for (int i = tid; i < WK_HASH; i += WKB_THREADS) h[i] = 0;
    if (tid == 0) *s_red = 0;
    __syncthreads();
    int chain = 0;
    for (int i = tid; i < n; i += WKB_THREADS) {
        const int32_t v = a[i];
        uint32_t b = sgnn_hash32((uint32_t)v) >> (32 - WK_HASH_BITS);
        int c = 0;
        while (true) {
            ++c;
            const int32_t old = atomicCAS(&h[b], 0, v);
            if (old == 0 || old == v) break;
            b = (b + 1) & (WK_HASH - 1);
        }
        chain = c > chain ? c : chain;
    }
    if (chain) atomicMax(s_red, chain);
    __syncthreads();
    return *s_red;
}

// pass 1 on the whole workgroup: classify the valid neighbours of v, keep the per-chunk ballots
__device__ static inline void wkb_count(const WalkCtx& c, int32_t v, bool has_prev, const uint32_t* bm, int tid,
                                        int32_t& nt, int32_t& nn, uint64_t* s_tri, uint64_t* s_non, int32_t* s_cnt) {
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = c.rowptr[v], r1 = c.rowptr[v + 1];
    const int64_t n_chunks = (r1 - r0 + 63) / 64;
    if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    int32_t lt = 0, ln = 0;
    for (int64_t ch = wave; ch < n_chunks; ch += WKB_THREADS / 64) {
        const int64_t e = r0 + ch * 64 + lane;
        bool ok = false, tri = false;
        if (e < r1) {
            const int32_t w = c.col[e];
            ok = walk_valid(c, w);
            if (ok && has_prev) tri = (bm[w >> 5] >> (w & 31)) & 1u;
        }
        const uint64_t mt = __ballot(ok && tri), mn = __ballot(ok && !tri);
        if (ch < WK_CHUNKS && lane == 0) { s_tri[ch] = mt; s_non[ch] = mn; }
        lt += __popcll(mt);
        ln += __popcll(mn);
    }
    if (lane == 0 && (lt | ln)) { atomicAdd(&s_cnt[0], lt); atomicAdd(&s_cnt[1], ln); }
    __syncthreads();
    nt = s_cnt[0];
    nn = s_cnt[1];
}

// pass 2: the pick-th valid neighbour of v in the wanted class (adjacency order) -> *s_next
__device__ static inline int32_t wkb_pick(const WalkCtx& c, int32_t v, bool has_prev, const uint32_t* bm, bool want_tri,
                                          int32_t pick, int tid, const uint64_t* s_tri, const uint64_t* s_non, int32_t* s_next) {
    const int lane = tid & 63;
    const int64_t r0 = c.rowptr[v], r1 = c.rowptr[v + 1];
    const int64_t n_chunks = (r1 - r0 + 63) / 64;
    if (tid < 64) {
        if (n_chunks <= WK_CHUNKS) {
            // lane l owns chunks [l*per, (l+1)*per): popcount them, scan over the wave, the owner of
            // the crossing walks its chunks
            const uint64_t* s = want_tri ? s_tri : s_non;
            const int per = (int)((n_chunks + 63) / 64);
            const int c0 = lane * per, c1 = (c0 + per < n_chunks) ? c0 + per : (int)n_chunks;
            int32_t mine = 0;
            for (int ch = c0; ch < c1; ++ch) mine += __popcll(s[ch]);
            int32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            const int32_t excl = incl - mine;
            if (pick >= excl && pick < incl) {
                int32_t seen = excl;
                for (int ch = c0; ch < c1; ++ch) {
                    uint64_t m = s[ch];
                    const int32_t cnt = __popcll(m);
                    if (seen + cnt > pick) {
                        int k = pick - seen;
                        while (k-- > 0) m &= m - 1;
                        *s_next = c.col[r0 + (int64_t)ch * 64 + (__ffsll((unsigned long long)m) - 1)];
                        break;
                    }
                    seen += cnt;
                }
            }
        } else {
            // list longer than the ballot cache: wavefront 0 re-classifies it in order
            int32_t seen = 0, res = 0;
            bool done = false;
            for (int64_t base = r0; base < r1 && !done; base += 64) {
                const int64_t e = base + lane;
                bool hit = false;
                int32_t w = 0;
                if (e < r1) {
                    w = c.col[e];
                    if (walk_valid(c, w)) {
                        const bool tri = has_prev ? (((bm[w >> 5] >> (w & 31)) & 1u) != 0) : false;
                        hit = (tri == want_tri);
                    }
                }
                const uint64_t m = __ballot(hit);
                const int32_t cnt = __popcll(m);
                if (seen + cnt > pick) {
                    const int32_t rank = __popcll(m & ((1ull << lane) - 1ull));
                    const uint64_t sel = __ballot(hit && (seen + rank == pick));
                    res = __shfl(w, __ffsll((unsigned long long)sel) - 1);
                    done = true;
                }
                seen += cnt;
            }
            if (lane == 0) *s_next = res;
        }
    }
    __syncthreads();
    return *s_next;
}

__global__ __launch_bounds__(WKB_THREADS) void triangular_walks_wg_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ col_sorted,
    const int32_t* __restrict__ node_order, int64_t n_nodes,
    const int64_t* __restrict__ patch_ptr, const int32_t* __restrict__ patch_nodes,
    const int64_t* __restrict__ inb_ptr, const int32_t* __restrict__ inb_nodes,
    int mode_all, int64_t n_items, int64_t walks_per_patch, int64_t walk_len, double beta,
    uint64_t h0_all, int64_t item_base, int64_t* __restrict__ out, int64_t words, uint64_t h0_border)
{
    extern __shared__ uint32_t s_adj[];                    // bitmap over node ids: N(prev)
    __shared__ uint64_t s_tri[WK_CHUNKS], s_non[WK_CHUNKS];
    __shared__ int32_t s_hp[WK_HASH], s_hi[WK_HASH];
    __shared__ int32_t s_cnt[2], s_next, s_red;
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < words; i += WKB_THREADS) s_adj[i] = 0;
    __syncthreads();
    for (int64_t item_all = blockIdx.x; item_all < n_items; item_all += gridDim.x) {
        int64_t* o = out + item_all * walk_len;
        for (int64_t t = tid; t < walk_len; t += WKB_THREADS) o[t] = 0;
        // mode 3: the internal walks (items [0, n / 2): mode 1) and the border walks (items [n / 2, n): mode 2) of the same
        // patches in ONE launch -- each side draws from its own tape stream under its own walk numbers, as two launches would
        int mode = mode_all;
        int64_t item = item_all;
        uint64_t h0 = h0_all;
        if (mode_all == 3) {
            const int64_t half = n_items >> 1;
            if (item_all >= half) { mode = 2; item = item_all - half; h0 = h0_border; } else mode = 1;
        }
        WalkCtx c;
        c.rowptr = rowptr; c.col = col; c.col_sorted = col_sorted; c.mode = mode;
        c.patch = nullptr; c.n_patch = 0; c.inb = nullptr; c.n_inb = 0;
        c.hpatch = nullptr; c.hinb = nullptr; c.pp = 0; c.pi = 0;
        const uint64_t h1 = sgnn_tape_h1(h0, (uint64_t)(item_base + item));   // the walk's GLOBAL number: a dealt share draws what the whole launch would
        uint64_t j = 0;
        int32_t prev;
        __syncthreads();                            // previous item's LDS tables are no longer read
        if (mode == 0) {
            prev = node_order[sgnn_choice_index(h1, j++, (uint32_t)n_nodes)];          // aps:70
        } else {
            const int64_t p = item / walks_per_patch;
            c.patch = patch_nodes + patch_ptr[p];
            c.n_patch = (int32_t)(patch_ptr[p + 1] - patch_ptr[p]);
            if (c.n_patch == 0) continue;                                              // aps:134-135
            if (c.n_patch <= WK_HASH_MAX) { c.pp = wkb_build_hash(s_hp, c.patch, c.n_patch, tid, &s_red); c.hpatch = s_hp; }
            if (mode == 1) {
                prev = c.patch[sgnn_choice_index(h1, j++, (uint32_t)c.n_patch)];      // aps:70
            } else {
                c.inb = inb_nodes + inb_ptr[p];
                c.n_inb = (int32_t)(inb_ptr[p + 1] - inb_ptr[p]);
                if (c.n_inb == 0) continue;        // reference raises ValueError here (aps:78)
                if (c.n_inb <= WK_HASH_MAX) { c.pi = wkb_build_hash(s_hi, c.inb, c.n_inb, tid, &s_red); c.hinb = s_hi; }
                prev = c.inb[sgnn_choice_index(h1, j++, (uint32_t)c.n_inb)];          // aps:78
            }
        }
        if (walk_len < 1) continue;
        __syncthreads();                            // zero fill above precedes the thread-0 writes
        if (tid == 0) o[0] = prev;
        int32_t nt, nn;
        wkb_count(c, prev, false, s_adj, tid, nt, nn, s_tri, s_non, s_cnt);             // aps:72,79
        if (nn == 0 || walk_len < 2) continue;                                          // aps:83-84
        int32_t curr = wkb_pick(c, prev, false, s_adj, false, (int32_t)sgnn_choice_index(h1, j++, (uint32_t)nn), tid,
                                s_tri, s_non, &s_next);                                  // aps:74,80
        if (tid == 0) o[1] = curr;
        // the bitmap holds N(prev) from here on
        wkb_bits(s_adj, col + rowptr[prev], (int32_t)(rowptr[prev + 1] - rowptr[prev]), true, tid);
        __syncthreads();
        for (int64_t step = 2; step < walk_len; ++step) {
            wkb_count(c, curr, true, s_adj, tid, nt, nn, s_tri, s_non, s_cnt);          // aps:35-45
            if (nt + nn == 0) break;                                                     // aps:94
            bool want_tri;
            if (nt == 0) want_tri = false;                                               // aps:97-98
            else if (nn == 0) want_tri = true;                                           // aps:99-100
            else want_tri = (sgnn_uniform01(h1, j++) <= beta);                           // aps:102
            const int32_t pick = (int32_t)sgnn_choice_index(h1, j++, (uint32_t)(want_tri ? nt : nn));
            const int32_t nxt = wkb_pick(c, curr, true, s_adj, want_tri, pick, tid, s_tri, s_non, &s_next);
            if (tid == 0) o[step] = nxt;
            // N(prev) out, N(curr) in: curr becomes prev
            wkb_bits(s_adj, col + rowptr[prev], (int32_t)(rowptr[prev + 1] - rowptr[prev]), false, tid);
            __syncthreads();
            if (step + 1 < walk_len) wkb_bits(s_adj, col + rowptr[curr], (int32_t)(rowptr[curr + 1] - rowptr[curr]), true, tid);
            __syncthreads();
            prev = (step + 1 < walk_len) ? curr : 0;
            curr = nxt;
        }
        // leave the bitmap clean for the next item of this workgroup
        if (prev) wkb_bits(s_adj, col + rowptr[prev], (int32_t)(rowptr[prev + 1] - rowptr[prev]), false, tid);
        __syncthreads();
    }
}

extern "C" int sgnn_triangular_walks(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz,
                                     const int32_t* node_order, int64_t n_nodes,
                                     const int64_t* patch_ptr, const int32_t* patch_nodes,
                                     const int64_t* inb_ptr, const int32_t* inb_nodes,
                                     int mode, int64_t n_items, int64_t walks_per_patch, int64_t walk_len, double beta,
                                     uint64_t seed, uint64_t stream_id, int64_t item_base, int64_t max_id, int kernel, int64_t* out,
                                     void* stream)
{
    if (!rowptr || !col || !col_sorted || !out || n_items < 0 || walk_len < 0 || mode < 0 || mode > 2 || kernel < 0 || kernel > 1 || item_base < 0)
        return SGNN_ERR_BAD_ARG;
    if (mode == 0 && (!node_order || n_nodes <= 0)) return SGNN_ERR_BAD_ARG;
    if (mode >= 1 && (!patch_ptr || !patch_nodes || walks_per_patch <= 0)) return SGNN_ERR_BAD_ARG;
    if (mode == 2 && (!inb_ptr || !inb_nodes)) return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (n_items == 0 || walk_len == 0) return SGNN_OK;
    const int64_t words = (max_id + 32) / 32;
    if (max_id > 0 && words * 4 <= WKB_LDS_BYTES && kernel == 0) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)triangular_walks_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      WKB_LDS_BYTES);
            attr_set = true;
        }
        hipLaunchKernelGGL(triangular_walks_wg_kernel, dim3((int)(n_items < 2048 ? n_items : 2048)), dim3(WKB_THREADS),
                           (size_t)(words * 4), (hipStream_t)stream, rowptr, col, col_sorted, node_order, n_nodes, patch_ptr,
                           patch_nodes, inb_ptr, inb_nodes, mode, n_items, walks_per_patch, walk_len, beta,
                           sgnn_tape_h0(seed, stream_id), item_base, out, words, (uint64_t)0);
        SGNN_CHECK_LAUNCH();
        return SGNN_OK;
    }
    hipLaunchKernelGGL(triangular_walks_kernel, dim3((int)(n_items < 256 * 32 ? n_items : 256 * 32)), dim3(64), 0, (hipStream_t)stream,
                       rowptr, col, col_sorted, node_order, n_nodes, patch_ptr, patch_nodes, inb_ptr, inb_nodes,
                       mode, n_items, walks_per_patch, walk_len, beta, sgnn_tape_h0(seed, stream_id), item_base, out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// Internal AND border walks over the same patches (aps:118-158 called twice by the reference: inside = True / False) in one
// launch where the graph's id bitmap fits LDS: out (2, n_items, walk_len), [0] = internal (tape stream stream_id_int), [1] =
// border (stream_id_bor).  Returns SGNN_ERR_SET_TOO_LARGE when it does not fit: the caller then makes the two launches.
extern "C" int sgnn_triangular_walks_both(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz,
                                          const int64_t* patch_ptr, const int32_t* patch_nodes, const int64_t* inb_ptr,
                                          const int32_t* inb_nodes, int64_t n_items, int64_t walks_per_patch, int64_t walk_len,
                                          double beta, uint64_t seed, uint64_t stream_id_int, uint64_t stream_id_bor,
                                          int64_t item_base, int64_t max_id, int64_t* out, void* stream)
{
    if (!rowptr || !col || !col_sorted || !out || !patch_ptr || !patch_nodes || !inb_ptr || !inb_nodes || n_items < 0 || walk_len < 0
        || walks_per_patch <= 0 || item_base < 0)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (n_items == 0 || walk_len == 0) return SGNN_OK;
    const int64_t words = (max_id + 32) / 32;
    if (max_id <= 0 || words * 4 > WKB_LDS_BYTES) return SGNN_ERR_SET_TOO_LARGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)triangular_walks_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WKB_LDS_BYTES);
        attr_set = true;
    }
    const int64_t both = 2 * n_items;
    hipLaunchKernelGGL(triangular_walks_wg_kernel, dim3((int)(both < 2048 ? both : 2048)), dim3(WKB_THREADS), (size_t)(words * 4),
                       (hipStream_t)stream, rowptr, col, col_sorted, (const int32_t*)nullptr, (int64_t)0, patch_ptr, patch_nodes, inb_ptr,
                       inb_nodes, 3, both, walks_per_patch, walk_len, beta, sgnn_tape_h0(seed, stream_id_int), item_base, out, words,
                       sgnn_tape_h0(seed, stream_id_bor));
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(samplers)
