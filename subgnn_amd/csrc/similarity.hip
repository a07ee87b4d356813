// Similarity kernels: shortest-path similarities (a9, dense-parity and sparse forms) and the
// structure similarity 1/(1+fastdtw) (a11).
#include "common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// a9  dense-parity form (reference SubGNN/SubGNN.py:752-781): column-wise min over the APSP rows
// of a component.  HBM-bound streaming: 8*|cc|*N bytes in, 4*N out per component; lanes walk the
// columns (coalesced 512 B per wave-instruction), the member loop re-uses nothing.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_similarity_dense_kernel(
    const double* __restrict__ apsp, int64_t n_cols,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes,
    float* __restrict__ out)
{
    const int64_t r = blockIdx.y;
    const int64_t beg = set_ptr[r];
    const int n = (int)(set_ptr[r + 1] - beg);
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n_cols; c += (int64_t)gridDim.x * blockDim.x) {
        float res = 0.f;
        if (n > 0) {
            double m = apsp[(int64_t)(set_nodes[beg] - 1) * n_cols + c];
            for (int i = 1; i < n; ++i) {
                const double v = apsp[(int64_t)(set_nodes[beg + i] - 1) * n_cols + c];
                m = v < m ? v : m;
            }
            res = (float)m;
        }
        out[r * n_cols + c] = res;
    }
}

extern "C" int sgnn_sp_similarity_dense(const double* apsp, int64_t n_cols,
                                        const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                        float* out, void* stream)
{
    if (!apsp || !set_ptr || !set_nodes || !out || n_cols <= 0 || n_sets < 0) return SGNN_ERR_BAD_ARG;
    if (n_sets == 0) return SGNN_OK;
    if (n_sets > 65535 * 32768ll) return SGNN_ERR_BAD_ARG;
    int gx = (int)((n_cols + 255) / 256);
    if (gx > 64) gx = 64;
    // grid.y is limited to 65535: fold larger set counts into several launches
    for (int64_t off = 0; off < n_sets; off += 65535) {
        const int64_t cnt = (n_sets - off) < 65535 ? (n_sets - off) : 65535;
        hipLaunchKernelGGL(sp_similarity_dense_kernel, dim3(gx, (int)cnt), dim3(256), 0, (hipStream_t)stream,
                           apsp, n_cols, set_ptr + off, set_nodes, out + off * n_cols);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// a9  sparse form: bit-parallel multi-source BFS.  Each node carries one 64-bit word per group
// of 64 sources (seen / frontier / next).  A level is two grid-wide kernels:
//   expand : direction-optimising.  While the frontier is small it PUSHES: every node with a
//            non-zero frontier word ORs it into next[] of its neighbours (16-lane groups stream
//            a neighbour list, coalesced; 64 BFS's per edge visit; atomics).  Once the frontier's
//            edge volume passes 1/alpha of all edges it PULLS: every node that still misses a
//            source ORs the frontier words of its neighbours into its own next[] -- no atomics,
//            nodes that have seen every source are skipped and a list is abandoned as soon as
//            nothing is missing any more (on a scale-free graph almost every node is complete
//            after 3-4 levels).  The choice is made on the device from a counter the previous
//            commit left behind, so the host still enqueues all levels without synchronising.
//   commit : new = next & ~seen; seen |= new; frontier = new; dist[src][node] = level for the
//            new bits; flags[level] says whether any bit was new (later levels exit early);
//            fvol[level] = sum over new frontier words of the node's degree.
// ---------------------------------------------------------------------------------------------
static int g_bfs_alpha = 32;     // pull when frontier word-edges * alpha > nnz * n_words; 0 = never pull

extern "C" int sgnn_bfs_hops_tuning(int alpha)
{
    if (alpha < 0) return SGNN_ERR_BAD_ARG;
    g_bfs_alpha = alpha;
    return SGNN_OK;
}

__global__ void msbfs_init_kernel(const int32_t* __restrict__ sources, int64_t n_sources, int64_t n_words,
                                  int64_t n_ids, uint64_t* __restrict__ seen, uint64_t* __restrict__ frontier,
                                  uint64_t* __restrict__ next, uint8_t* __restrict__ dist, int32_t* __restrict__ flags,
                                  unsigned long long* __restrict__ fvol, uint32_t* __restrict__ fbits,
                                  int max_hops)  // dist layout-agnostic: filled as a flat array
{
    const int64_t gtid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = gtid; i < 2 * ((n_ids + 31) / 32); i += gsz) fbits[i] = 0;   // frontier nodes | complete nodes
    for (int64_t i = gtid; i < n_ids * n_words; i += gsz) { seen[i] = 0; frontier[i] = 0; next[i] = 0; }
    if (dist) for (int64_t i = gtid; i < n_sources * n_ids; i += gsz) dist[i] = 255;
    for (int64_t i = gtid; i <= max_hops; i += gsz) { flags[i] = (i == 0) ? 1 : 0; fvol[i] = 0; }
}

__global__ void msbfs_seed_kernel(const int32_t* __restrict__ sources, int64_t n_sources, int64_t n_words,
                                  int64_t n_ids, uint64_t* __restrict__ seen, uint64_t* __restrict__ frontier,
                                  uint8_t* __restrict__ dist, int64_t ss, int64_t sv, uint32_t* __restrict__ fbits)
{
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= n_sources) return;
    const int32_t v = sources[s];
    atomicOr(&fbits[v >> 5], 1u << (v & 31));                 // level-0 frontier nodes
    const uint64_t bit = 1ull << (s & 63);
    atomicOr((unsigned long long*)&seen[(int64_t)v * n_words + (s >> 6)], (unsigned long long)bit);
    atomicOr((unsigned long long*)&frontier[(int64_t)v * n_words + (s >> 6)], (unsigned long long)bit);
    if (dist) dist[s * ss + v * sv] = 0;
}

#define MSBFS_WCHUNK 4          // source words a pull pass keeps in registers
#define MSBFS_HUB_DEGREE 128     // push: lists this long go to the whole workgroup
#define MSBFS_HUB_SLOTS 128
#define MSBFS_PULL_HUB_DEGREE 512

__device__ __forceinline__ uint64_t msbfs_group_or(uint64_t x)
{
    // OR over the 16 lanes of a group (xor-butterfly: every lane ends with the full value)
    for (int off = 8; off >= 1; off >>= 1) {
        const uint32_t lo = __shfl_xor((int)(uint32_t)x, off, 64);
        const uint32_t hi = __shfl_xor((int)(uint32_t)(x >> 32), off, 64);
        x |= ((uint64_t)hi << 32) | lo;
    }
    return x;
}

__global__ __launch_bounds__(256) void msbfs_expand_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_ids, int64_t n_words,
    int64_t n_sources, const uint64_t* __restrict__ seen, const uint64_t* __restrict__ frontier,
    uint64_t* __restrict__ next, const int32_t* __restrict__ flags, const unsigned long long* __restrict__ fvol,
    unsigned long long pull_above, int level, const uint32_t* __restrict__ fnode, unsigned long long sparse_below,
    const uint32_t* __restrict__ fdone)
{
    if (flags[level - 1] == 0) return;                       // previous level found nothing
    const int sub = threadIdx.x & 15;
    // node ids are dealt out to the workgroups round-robin (group g of workgroup b takes b + G*(g + 16 i)):
    // consecutive ids -- the oldest, highest-degree nodes of a preferential-attachment graph sit next to
    // each other at the low ids -- land in different workgroups
    const int64_t group = blockIdx.x + (int64_t)gridDim.x * (threadIdx.x >> 4);
    const int64_t n_groups = (int64_t)gridDim.x * (blockDim.x >> 4);
    const bool pull = level > 1 && pull_above != ~0ull && fvol[level - 1] > pull_above;
    const bool sparse_frontier = fvol[level - 1] < sparse_below;
    __shared__ int32_t s_hub[MSBFS_HUB_SLOTS];
    __shared__ int s_nhub;
    __shared__ unsigned long long s_acc[MSBFS_WCHUNK];
    if (threadIdx.x == 0) s_nhub = 0;
    __syncthreads();
    if (!pull) {
        // A frontier node's list is streamed by its 16-lane group -- except long lists (hubs: a BA graph
        // of 1M nodes has lists of 10k+ entries, and hubs are on the frontier from level 1 on), which one
        // group would walk for a millisecond while the rest of the chip is done: those are parked in LDS
        // and streamed by the whole workgroup afterwards.
        for (int64_t v = group; v < n_ids; v += n_groups) {
            uint64_t any = 0;
            for (int64_t w = 0; w < n_words; ++w) any |= frontier[v * n_words + w];
            if (any == 0) continue;
            // one pass over the neighbour list for all source words: col[] is read once, and the
            // n_words seen/next words of a neighbour are contiguous
            const int64_t r0 = rowptr[v], r1 = rowptr[v + 1];
            if (r1 - r0 >= MSBFS_HUB_DEGREE) {
                int slot = 0;
                if (sub == 0) slot = atomicAdd(&s_nhub, 1);
                slot = __shfl(slot, (threadIdx.x & 63) & ~15, 64);
                if (slot < MSBFS_HUB_SLOTS) {
                    if (sub == 0) s_hub[slot] = (int32_t)v;
                    continue;
                }
            }
            for (int64_t e = r0 + sub; e < r1; e += 16) {
                const int64_t u = col[e];
                for (int64_t w = 0; w < n_words; ++w) {
                    const uint64_t f = frontier[v * n_words + w];
                    const uint64_t m = f & ~seen[u * n_words + w];
                    if (m) atomicOr((unsigned long long*)&next[u * n_words + w], (unsigned long long)m);
                }
            }
        }
        __syncthreads();
        const int n_hub = s_nhub < MSBFS_HUB_SLOTS ? s_nhub : MSBFS_HUB_SLOTS;
        for (int h = 0; h < n_hub; ++h) {
            const int64_t v = s_hub[h];
            const int64_t r0 = rowptr[v], r1 = rowptr[v + 1];
            for (int64_t e = r0 + threadIdx.x; e < r1; e += blockDim.x) {
                const int64_t u = col[e];
                for (int64_t w = 0; w < n_words; ++w) {
                    const uint64_t f = frontier[v * n_words + w];
                    const uint64_t m = f & ~seen[u * n_words + w];
                    if (m) atomicOr((unsigned long long*)&next[u * n_words + w], (unsigned long long)m);
                }
            }
        }
        return;
    }
    for (int64_t v = group; v < n_ids; v += n_groups) {
        // one bit per node: every source has reached it (kept by commit) -- from level 3-4 on that is
        // almost every node, and the test replaces the row-pointer and seen-word loads of the skip path
        if ((fdone[v >> 5] >> (v & 31)) & 1u) continue;
        const int64_t r0 = rowptr[v], r1 = rowptr[v + 1];
        for (int64_t w0 = 0; w0 < n_words; w0 += MSBFS_WCHUNK) {
            uint64_t need[MSBFS_WCHUNK], acc[MSBFS_WCHUNK];
            uint64_t missing = 0;
#pragma unroll
            for (int k = 0; k < MSBFS_WCHUNK; ++k) {
                const int64_t w = w0 + k;
                uint64_t valid = 0;
                if (w < n_words) {
                    const int64_t left = n_sources - w * 64;
                    valid = left >= 64 ? ~0ull : ((1ull << left) - 1);
                    valid &= ~seen[v * n_words + w];
                }
                need[k] = valid;
                acc[k] = 0;
                missing |= valid;
            }
            if (missing == 0) continue;                      // this node has every source of the chunk
            if (r1 - r0 >= MSBFS_PULL_HUB_DEGREE) {          // long list: the whole workgroup, below
                int slot = 0;
                if (sub == 0) slot = atomicAdd(&s_nhub, 1);
                slot = __shfl(slot, (threadIdx.x & 63) & ~15, 64);
                if (slot < MSBFS_HUB_SLOTS) {
                    if (sub == 0) s_hub[slot] = (int32_t)v;
                    break;                                   // all chunks of this node are done there
                }
            }
            int since = 0;
            for (int64_t e = r0 + sub; e < r1 + sub; e += 16) {          // uniform trip count per group
                if (e < r1) {
                    const int64_t u = col[e];
                    // one bit per node says whether it is on the frontier at all: a 125 KB table that
                    // stays in L2, consulted before the 24-byte gather from the 24 MB word array -- while the
                    // frontier is sparse (once most nodes are on it the test only adds a load)
                    if (!sparse_frontier || ((fnode[u >> 5] >> (u & 31)) & 1u)) {
#pragma unroll
                        for (int k = 0; k < MSBFS_WCHUNK; ++k)
                            if (need[k]) acc[k] |= frontier[u * n_words + w0 + k];  // completed words are not read
                    }
                }
                if (++since == 8) {                          // every 128 neighbours: anything still missing?
                    since = 0;
                    uint64_t left = 0;
#pragma unroll
                    for (int k = 0; k < MSBFS_WCHUNK; ++k) { acc[k] = msbfs_group_or(acc[k]); left |= need[k] & ~acc[k]; }
                    if (left == 0) break;
                }
            }
#pragma unroll
            for (int k = 0; k < MSBFS_WCHUNK; ++k) {
                acc[k] = msbfs_group_or(acc[k]);
                if (sub == k && w0 + k < n_words) {
                    const uint64_t m = acc[k] & need[k];
                    if (m) next[v * n_words + w0 + k] = m;   // next[] is all zero before a pull level
                }
            }
        }
    }
    // parked long lists: 256 lanes per list, the words OR-ed through LDS
    __syncthreads();
    const int n_hub = s_nhub < MSBFS_HUB_SLOTS ? s_nhub : MSBFS_HUB_SLOTS;
    for (int h = 0; h < n_hub; ++h) {
        const int64_t v = s_hub[h];
        const int64_t r0 = rowptr[v], r1 = rowptr[v + 1];
        for (int64_t w0 = 0; w0 < n_words; w0 += MSBFS_WCHUNK) {
            uint64_t need[MSBFS_WCHUNK], acc[MSBFS_WCHUNK];
            uint64_t missing = 0;
#pragma unroll
            for (int k = 0; k < MSBFS_WCHUNK; ++k) {
                const int64_t w = w0 + k;
                uint64_t valid = 0;
                if (w < n_words) {
                    const int64_t left = n_sources - w * 64;
                    valid = left >= 64 ? ~0ull : ((1ull << left) - 1);
                    valid &= ~seen[v * n_words + w];
                }
                need[k] = valid;
                acc[k] = 0;
                missing |= valid;
            }
            if (missing == 0) continue;                      // uniform over the workgroup
            if (threadIdx.x < MSBFS_WCHUNK) s_acc[threadIdx.x] = 0;
            __syncthreads();
            for (int64_t e = r0 + threadIdx.x; e < r1; e += blockDim.x) {
                const int64_t u = col[e];
                if (!sparse_frontier || ((fnode[u >> 5] >> (u & 31)) & 1u)) {
#pragma unroll
                    for (int k = 0; k < MSBFS_WCHUNK; ++k)
                        if (need[k]) acc[k] |= frontier[u * n_words + w0 + k];
                }
            }
#pragma unroll
            for (int k = 0; k < MSBFS_WCHUNK; ++k)
                if (acc[k]) atomicOr(&s_acc[k], (unsigned long long)acc[k]);
            __syncthreads();
            if (threadIdx.x < MSBFS_WCHUNK && w0 + threadIdx.x < n_words) {
                const uint64_t m = s_acc[threadIdx.x] & need[threadIdx.x];
                if (m) next[v * n_words + w0 + threadIdx.x] = m;
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void msbfs_commit_kernel(
    const int64_t* __restrict__ rowptr, int64_t n_ids, int64_t n_words, int64_t n_sources, uint64_t* __restrict__ seen,
    uint64_t* __restrict__ frontier, uint64_t* __restrict__ next, uint8_t* __restrict__ dist, int32_t* __restrict__ flags,
    unsigned long long* __restrict__ fvol, int level, int64_t ss, int64_t sv, uint32_t* __restrict__ fcur,
    uint32_t* __restrict__ fdone)
{
    if (flags[level - 1] == 0) return;
    // One lane per node, a wave per 64 consecutive nodes: the ballot of "some word of my node is new"
    // IS the two 32-bit words of the frontier-node bitmap for those nodes -- plain stores, every word of
    // fcur rewritten each level (nobody reads it while this kernel runs: expand of this level is done).
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t fwords = (n_ids + 31) / 32;
    bool any = false;
    unsigned long long vol = 0;
    for (int64_t base = wave * 64; base < n_ids; base += n_waves * 64) {
        const int64_t v = base + lane;
        int n_new = 0;
        bool done = v < n_ids;
        if (v < n_ids) {
            for (int64_t w = 0; w < n_words; ++w) {
                const int64_t i = v * n_words + w;
                const uint64_t nx = next[i];
                const uint64_t sn = seen[i];
                const uint64_t nw = nx & ~sn;
                const int64_t left = n_sources - w * 64;
                done = done && ((sn | nw) == (left >= 64 ? ~0ull : ((1ull << left) - 1)));
                if (nx) next[i] = 0;
                frontier[i] = nw;
                if (nw) {
                    ++n_new;
                    seen[i] = sn | nw;
                    if (dist) {
                        uint64_t bits = nw;
                        while (bits) {
                            const int b = __ffsll((unsigned long long)bits) - 1;
                            bits &= bits - 1;
                            const int64_t s = w * 64 + b;
                            if (s < n_sources) dist[s * ss + v * sv] = (uint8_t)level;
                        }
                    }
                }
            }
            if (n_new) {
                any = true;
                vol += (unsigned long long)n_new * (unsigned long long)(rowptr[v + 1] - rowptr[v]);
            }
        }
        const unsigned long long mask = __ballot(n_new != 0);
        const unsigned long long dmask = __ballot(done);
        if (lane == 0) {
            fcur[base >> 5] = (uint32_t)mask;
            fdone[base >> 5] = (uint32_t)dmask;
            if ((base >> 5) + 1 < fwords) {
                fcur[(base >> 5) + 1] = (uint32_t)(mask >> 32);
                fdone[(base >> 5) + 1] = (uint32_t)(dmask >> 32);
            }
        }
    }
    // one pair of atomics per workgroup: every wave adding to the same two words serialises at the
    // memory side (16k waves -> ~150 us of a 200 us launch)
    __shared__ unsigned long long s_vol[4];
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = __shfl_xor((int)(uint32_t)vol, off, 64);
        const uint32_t hi = __shfl_xor((int)(uint32_t)(vol >> 32), off, 64);
        vol += ((unsigned long long)hi << 32) | lo;
    }
    const bool wave_any = __any(any);
    if (lane == 0) s_vol[threadIdx.x >> 6] = wave_any ? (vol | (1ull << 63)) : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0, flag = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { t += s_vol[k] & ~(1ull << 63); flag |= s_vol[k] >> 63; }
        if (flag) { atomicOr(&flags[level], 1); atomicAdd(&fvol[level], t); }
    }
}

extern "C" int64_t sgnn_bfs_hops_workspace_bytes(int64_t max_id, int64_t n_sources, int max_hops) {
    const int64_t n_words = (n_sources + 63) / 64;
    return 3 * (max_id + 1) * n_words * 8 + ((int64_t)max_hops + 2) * 8 + ((int64_t)max_hops + 2) * 4 +
           8 + 2 * ((max_id + 32) / 32) * 4;                      // + two frontier-node bitmaps
}

// Per level, for every set: which sources reached one of its members for the first time?  The new
// frontier words of the members are OR-ed, masked with what the set has not seen yet, and the level
// is written for those sources: out[set, source] = min over members of the hop distance, without a
// (sources x nodes) hop table.  One thread per (set, word).
__global__ __launch_bounds__(256) void msbfs_set_reduce_kernel(
    const uint64_t* __restrict__ frontier, int64_t n_words, int64_t n_sources,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    uint64_t* __restrict__ set_seen, float* __restrict__ out, const int32_t* __restrict__ flags, int level)
{
    if (level > 0 && flags[level] == 0) return;              // this level found nothing
    const int64_t total = n_sets * n_words;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n_words, w = t % n_words;
        uint64_t acc = 0;
        for (int64_t i = set_ptr[r]; i < set_ptr[r + 1]; ++i) acc |= frontier[(int64_t)set_nodes[i] * n_words + w];
        const uint64_t fresh = acc & ~set_seen[t];
        if (fresh) {
            set_seen[t] |= fresh;
            uint64_t bits = fresh;
            while (bits) {
                const int b = __ffsll((unsigned long long)bits) - 1;
                bits &= bits - 1;
                const int64_t s = w * 64 + b;
                if (s < n_sources) out[r * n_sources + s] = (float)level;
            }
        }
    }
}

// The reference's matrix holds 0 for unreachable pairs, and its row-min runs over those zeros: a
// source that never reaches SOME member of a set gives 0 for the whole set (SubGNN.py:772).  After
// the last level: AND the members' seen words, zero the sources missing from it.
__global__ __launch_bounds__(256) void msbfs_set_finalize_kernel(
    const uint64_t* __restrict__ seen, int64_t n_words, int64_t n_sources,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets, float* __restrict__ out)
{
    const int64_t total = n_sets * n_words;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n_words, w = t % n_words;
        uint64_t all = ~0ull;
        for (int64_t i = set_ptr[r]; i < set_ptr[r + 1]; ++i) all &= seen[(int64_t)set_nodes[i] * n_words + w];
        uint64_t missing = ~all;
        while (missing) {
            const int b = __ffsll((unsigned long long)missing) - 1;
            missing &= missing - 1;
            const int64_t s = w * 64 + b;
            if (s < n_sources) out[r * n_sources + s] = 0.f;
        }
    }
}

// status[0] = the last level that found anything, status[1] = 1 if the LAST enqueued level still found something
// (the search may be incomplete: the caller enqueued too few levels)
__global__ void msbfs_status_kernel(const int32_t* __restrict__ flags, int max_hops, int32_t* __restrict__ status)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int last = 0;
        for (int l = 1; l <= max_hops; ++l) if (flags[l]) last = l;
        status[0] = last;
        status[1] = flags[max_hops] != 0;
    }
}

static int msbfs_run(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                     const int32_t* sources, int64_t n_sources, int max_hops, int node_major, uint8_t* dist,
                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, float* set_out,
                     void* workspace, hipStream_t st, int32_t* status = nullptr)
{
    const int64_t n_ids = max_id + 1;
    const int64_t n_words = (n_sources + 63) / 64;
    const int64_t ss = node_major ? 1 : n_ids, sv = node_major ? n_sources : 1;
    uint64_t* seen = (uint64_t*)workspace;
    uint64_t* frontier = seen + n_ids * n_words;
    uint64_t* next = frontier + n_ids * n_words;
    unsigned long long* fvol = (unsigned long long*)(next + n_ids * n_words);
    int32_t* flags = (int32_t*)(fvol + max_hops + 2);
    uint32_t* fbits = (uint32_t*)(((uintptr_t)(flags + max_hops + 2) + 7) & ~(uintptr_t)7);
    const int64_t fwords = (n_ids + 31) / 32;
    uint64_t* set_seen = (uint64_t*)(((uintptr_t)(fbits + 2 * fwords) + 7) & ~(uintptr_t)7);
    const unsigned long long pull_above =
        g_bfs_alpha > 0 ? (unsigned long long)((nnz * n_words) / g_bfs_alpha) : ~0ull;
    const int big = sgnn_grid_for(n_ids * ((dist && n_sources > n_words) ? n_sources : n_words), 256);
    hipLaunchKernelGGL(msbfs_init_kernel, dim3(big), dim3(256), 0, st, sources, n_sources, n_words, n_ids, seen,
                       frontier, next, dist, flags, fvol, fbits, max_hops);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(msbfs_seed_kernel, dim3((int)((n_sources + 255) / 256)), dim3(256), 0, st, sources, n_sources,
                       n_words, n_ids, seen, frontier, dist, ss, sv, fbits);
    SGNN_CHECK_LAUNCH();
    const int g_sets = set_out ? sgnn_grid_for(n_sets * n_words, 256) : 0;
    if (set_out) {
        hipError_t me = hipMemsetAsync(set_seen, 0, (size_t)(n_sets * n_words * 8), st);
        if (me == hipSuccess) me = hipMemsetAsync(set_out, 0, (size_t)(n_sets * n_sources * 4), st);  // unreachable pairs hold 0
        if (me != hipSuccess) { sgnn_set_last_error(me); return SGNN_ERR_LAUNCH; }
        hipLaunchKernelGGL(msbfs_set_reduce_kernel, dim3(g_sets), dim3(256), 0, st, frontier, n_words, n_sources, set_ptr,
                           set_nodes, n_sets, set_seen, set_out, flags, 0);
        SGNN_CHECK_LAUNCH();
    }
    const int g_expand = sgnn_grid_for(n_ids * 16, 256);
    const int g_commit = sgnn_grid_for(n_ids, 256, 256 * 4);
    for (int level = 1; level <= max_hops; ++level) {
        hipLaunchKernelGGL(msbfs_expand_kernel, dim3(g_expand), dim3(256), 0, st, rowptr, col, n_ids, n_words, n_sources,
                           seen, frontier, next, flags, fvol, pull_above, level, fbits,
                           (unsigned long long)((nnz * n_words) / 4), fbits + fwords);
        SGNN_CHECK_LAUNCH();
        hipLaunchKernelGGL(msbfs_commit_kernel, dim3(g_commit), dim3(256), 0, st, rowptr, n_ids, n_words, n_sources, seen,
                           frontier, next, dist, flags, fvol, level, ss, sv, fbits, fbits + fwords);
        SGNN_CHECK_LAUNCH();
        if (set_out) {
            hipLaunchKernelGGL(msbfs_set_reduce_kernel, dim3(g_sets), dim3(256), 0, st, frontier, n_words, n_sources,
                               set_ptr, set_nodes, n_sets, set_seen, set_out, flags, level);
            SGNN_CHECK_LAUNCH();
        }
    }
    if (set_out) {
        hipLaunchKernelGGL(msbfs_set_finalize_kernel, dim3(g_sets), dim3(256), 0, st, seen, n_words, n_sources, set_ptr,
                           set_nodes, n_sets, set_out);
        SGNN_CHECK_LAUNCH();
    }
    if (status) {
        hipLaunchKernelGGL(msbfs_status_kernel, dim3(1), dim3(64), 0, st, flags, max_hops, status);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

extern "C" int sgnn_bfs_hops(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                             const int32_t* sources, int64_t n_sources, int max_hops, int node_major,
                             uint8_t* dist, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!rowptr || !col || !sources || !dist || !workspace || n_sources < 0 || max_hops < 1 || max_hops > 254)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (workspace_bytes < sgnn_bfs_hops_workspace_bytes(max_id, n_sources, max_hops)) return SGNN_ERR_BAD_ARG;
    if (n_sources == 0) return SGNN_OK;
    return msbfs_run(rowptr, col, nnz, max_id, sources, n_sources, max_hops, node_major, dist, nullptr, nullptr, 0,
                     nullptr, workspace, (hipStream_t)stream);
}

extern "C" int64_t sgnn_bfs_min_hops_workspace_bytes(int64_t max_id, int64_t n_sources, int max_hops, int64_t n_sets) {
    const int64_t n_words = (n_sources + 63) / 64;
    return sgnn_bfs_hops_workspace_bytes(max_id, n_sources, max_hops) + 16 + n_sets * n_words * 8;
}

extern "C" int sgnn_bfs_min_hops_to_sets(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                                         const int32_t* sources, int64_t n_sources, int max_hops,
                                         const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                         float* out, int32_t* out_status, void* workspace, int64_t workspace_bytes,
                                         void* stream)
{
    if (!rowptr || !col || !sources || !set_ptr || !set_nodes || !out || !workspace || n_sources < 0 || n_sets < 0 ||
        max_hops < 1 || max_hops > 254)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (workspace_bytes < sgnn_bfs_min_hops_workspace_bytes(max_id, n_sources, max_hops, n_sets)) return SGNN_ERR_BAD_ARG;
    if (n_sources == 0 || n_sets == 0) return SGNN_OK;
    return msbfs_run(rowptr, col, nnz, max_id, sources, n_sources, max_hops, 0, nullptr, set_ptr, set_nodes, n_sets, out,
                     workspace, (hipStream_t)stream, out_status);
}

__global__ void min_hops_to_sets_kernel(const uint8_t* __restrict__ dist, int64_t n_sources, int64_t n_ids,
                                        const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes,
                                        int64_t n_sets, float* __restrict__ out, int64_t ss, int64_t sv)
{
    const int64_t total = n_sets * n_sources;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n_sources, a = t % n_sources;
        const int64_t beg = set_ptr[r];
        const int n = (int)(set_ptr[r + 1] - beg);
        int m = 0;
        for (int i = 0; i < n; ++i) {
            int d = dist[a * ss + (int64_t)set_nodes[beg + i] * sv];
            if (d == 255) d = 0;                               // unreachable pairs hold 0 in the matrix
            m = (i == 0 || d < m) ? d : m;
        }
        out[t] = (float)m;
    }
}

extern "C" int sgnn_min_hops_to_sets(const uint8_t* dist, int64_t n_sources, int64_t max_id, int node_major,
                                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                     float* out, void* stream)
{
    if (!dist || !set_ptr || !set_nodes || !out || n_sources < 0 || n_sets < 0) return SGNN_ERR_BAD_ARG;
    if (n_sets * n_sources == 0) return SGNN_OK;
    hipLaunchKernelGGL(min_hops_to_sets_kernel, dim3(sgnn_grid_for(n_sets * n_sources, 256)), dim3(256), 0,
                       (hipStream_t)stream, dist, n_sources, max_id + 1, set_ptr, set_nodes, n_sets, out,
                       node_major ? 1 : max_id + 1, node_major ? n_sources : 1);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

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
#define DTW_REG_BLOCKS (256 * 16)     // register variant: its scratch is LDS; 4x more workgroups than fit at once, so the tail of the launch is short (1024 / 2048 / 4096 / 8192: 8.35 / 8.02 / 7.81 / 7.78 ms)
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

extern "C" int64_t sgnn_dtw_workspace_bytes(int64_t n_x, int64_t max_x_len, int64_t n_y, int64_t max_y_len) {
    if (max_x_len < 1) max_x_len = 1;
    if (max_y_len < 1) max_y_len = 1;
    const DtwLayout L = dtw_layout(max_x_len, max_y_len);
    const int64_t lane = L.n_dbl * 8 + dtw_align8(L.n_i32 * 4) + dtw_align8(L.n_dir * 4);
    int64_t scratch = DTW_NT * lane;
    if (DTW_REG_NT * L.YL * 8 > scratch) scratch = DTW_REG_NT * L.YL * 8;   // register variant: one word per column and level
    return scratch
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
    // Both quotients, the larger one is max / min: rounding is monotone, so RN(a1 / b1) >= 1 >= RN(b1 / a1) when
    // a1 >= b1 -- the division step only has to be exact for the quotient that is >= 1 (the direction the CPU test
    // covers); the other one only has to stay <= 1, and b1 / a1 <= 1 - 2^-24 for these operands.  7 instructions
    // instead of compare + two 64-bit selects + max + min + the division step.
    const double qa0 = __dmul_rn(a1, rb), qb0 = __dmul_rn(b1, ra);
    const double qa = __fma_rn(__fma_rn(-qa0, b1, a1), rb, qa0);
    const double qb = __fma_rn(__fma_rn(-qb0, a1, b1), ra, qb0);
    return __dadd_rn(fmax(qa, qb), -1.0);
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
// The DP runs column-major (any topological order fills identical cells and makes identical
// predecessor choices): the DTW_R row values of the previous column live in registers and are
// updated in place while the column index j walks the anchor series, so the DP state never
// leaves the register file.  The row loop is fully unrolled (static register indexing); rows
// outside a lane's window are skipped by the exec mask, and a (row, column) slot no lane of the
// wavefront needs is skipped altogether.  One 64-bit word of 2-bit predecessor codes per column
// is the only per-cell state written to memory (write-once, coalesced); the backtrack reads it
// back and keeps the per-row column range of the path in registers for the next finer level.
#define DTW_R 32
#ifndef DTW_MINB12
#define DTW_MINB12 3            // resident 256-thread blocks per CU the 12-row kernel is compiled for
#endif
#ifndef DTW_UNIFORM_BLOCKS
#define DTW_UNIFORM_BLOCKS 0            // scalar block tests from wave-union ranges: measured no gain (8.86 vs 8.81 ms)
#endif
#ifndef DTW_BRANCHLESS_ROWS
#define DTW_BRANCHLESS_ROWS 1
#endif
#ifndef DTW_MINB32
#define DTW_MINB32 1
#endif
#ifndef DTW_MINB20
#define DTW_MINB20 2
#endif

// one level of the register-resident DP, unrolled over RR <= DTW_R rows (the coarse levels and short
// components take the narrow instantiations, so the unrolled row loop does not sweep empty rows)
// fl: this lane's column of the workgroup's LDS table (stride DTW_THREADS words) holding the coarser
// path's first | last << 16 column per row.  Predecessor codes of a non-finest level go to wl (LDS,
// 32-bit words: such a level has at most 16 rows) when WLDS, else to the global scratch wq.
// FINEST: the level whose distance is the result -- it is never backtracked, so it neither tracks
// nor stores predecessor codes (it holds more than half of all cells).
#ifndef DTW_BLK
#define DTW_BLK 4          // rows per block of the column sweep (divides 12, 20 and 32)
#endif
template <int RMAX, int RR, int TIE, bool WLDS, bool FINEST>
__device__ __forceinline__ double dtw_reg_level(
    int32_t* __restrict__ fl, const double* __restrict__ xcol, const double* __restrict__ xrcol, int64_t n_x,
    const double* __restrict__ ycol, const double* __restrict__ yrcol,
    int lx, int ly, int lxc, int lyc, bool coarsest, int32_t* ublk, uint32_t* __restrict__ wl,
    uint64_t* __restrict__ wq, int64_t NT)
{
    constexpr bool finest = FINEST;
    typedef typename std::conditional<(RR <= 16), uint32_t, uint64_t>::type word_t;   // 2 bits per row
#define FLQ(q) fl[(q) * DTW_THREADS]
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const int32_t EMPTY = 1;                                  // lo = 1, hi = 0
    int32_t lohi[RR];
    if (coarsest) {
#pragma unroll
        for (int i = 0; i < RR; ++i) lohi[i] = i < lx ? ((ly - 1) << 16) : EMPTY;
    } else {
        int prev_lo = 0;
#pragma unroll
        for (int i = 0; i < RR; ++i) {
            const int ci = i >> 1;
            const int ca = ci - 1 < 0 ? 0 : ci - 1;                          // <= lxc - 1 for every real row
            const int cb = ci + 1;                                           // static; rows past the coarse
            const int firstc = FLQ(ca) & 0xffff;                             // path end take its last column
            const int lastc = (cb < lxc) ? (FLQ(cb < RMAX ? cb : RMAX - 1) >> 16) : (lyc - 1);
            int lo = 2 * (firstc - 1);
            int hi = 2 * (lastc + 1) + 1;
            if (lo < prev_lo) lo = prev_lo;
            if (lo < 0) lo = 0;
            if (hi > ly - 1) hi = ly - 1;
            int32_t v = (hi << 16) | lo;
            if (hi < lo || i >= lx) v = EMPTY; else prev_lo = lo;
            lohi[i] = v;
        }
    }
    // rows in blocks of DTW_BLK: a block is swept for the columns [min lo, max hi + 1] of its rows (the
    // extra column lets every cell of the block fall back to INF once), and skipped with one test
    // elsewhere -- most (row, column) pairs lie outside the radius-1 window
    int32_t blk[RR / DTW_BLK];
#pragma unroll
    for (int b = 0; b < RR / DTW_BLK; ++b) {
        int lo = 0x7fff, hi = -1;
#pragma unroll
        for (int q = 0; q < DTW_BLK; ++q) {
            const int l = lohi[DTW_BLK * b + q] & 0xffff, h = lohi[DTW_BLK * b + q] >> 16;
            if (h >= l) { lo = l < lo ? l : lo; hi = h > hi ? h : hi; }
        }
        blk[b] = hi < 0 ? EMPTY : (((hi + 1) << 16) | lo);
    }
#if DTW_UNIFORM_BLOCKS
    // The block tests of the column loop become scalar: the lanes of a wavefront work on similar
    // series, so the union of their block ranges (LDS min / max over the active lanes, once per level,
    // read back into scalar registers) is what the wavefront executes anyway -- testing it with
    // s_cmp / s_cbranch costs the vector pipeline nothing, where the per-lane test cost ~5 vector
    // instructions per block and column.  Rows inside a live block select by their own window.
    int ublo[RR / DTW_BLK], ubhi[RR / DTW_BLK];
    {
        int32_t* su = ublk + (threadIdx.x >> 6) * 16;
        const int l16 = threadIdx.x & 15;
        if ((threadIdx.x & 63) < 16) su[l16] = (l16 & 1) ? -1 : 0x7fff;          // even: lo, odd: hi
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < RR / DTW_BLK; ++b)
            if (blk[b] != EMPTY) { atomicMin(&su[2 * b], blk[b] & 0xffff); atomicMax(&su[2 * b + 1], blk[b] >> 16); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < RR / DTW_BLK; ++b) {
            ublo[b] = __builtin_amdgcn_readfirstlane(su[2 * b]);
            ubhi[b] = __builtin_amdgcn_readfirstlane(su[2 * b + 1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#endif
    double xp1[RR], xr[RR], col[RR];
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        xp1[i] = i < lx ? xcol[(int64_t)i * n_x] + 1.0 : 1.0;
        xr[i] = i < lx ? xrcol[(int64_t)i * n_x] : 1.0;
        col[i] = INF;
    }
    double y_next = ycol[0], yr_next = yrcol[0];
    for (int j = 0; j < ly; ++j) {
        const double yp1 = y_next + 1.0, yr = yr_next;
        if (j + 1 < ly) { y_next = ycol[j + 1]; yr_next = yrcol[j + 1]; }       // in flight during this column
        word_t word = 0;
        double up = INF;
        double diag = (j == 0) ? 0.0 : INF;                                  // virtual origin D[0][0] = 0
#pragma unroll
        for (int b = 0; b < RR / DTW_BLK; ++b) {
#if DTW_UNIFORM_BLOCKS
            if (j >= ublo[b] && j <= ubhi[b]) {
#else
            if (j >= (blk[b] & 0xffff) && j <= (blk[b] >> 16)) {
#endif
#pragma unroll
                for (int q = 0; q < DTW_BLK; ++q) {
                    const int i = DTW_BLK * b + q;
                    const int lo = lohi[i] & 0xffff, hi = lohi[i] >> 16;
                    const double old = col[i];
#if DTW_BRANCHLESS_ROWS
                    // every row of a live block is evaluated and the result selected by the row's own
                    // window test: no per-row branch (a branch stalls the wavefront's issue; the block
                    // test above already removed most out-of-window rows)
                    const bool in = j >= lo && j <= hi;
                    const double dt = dtw_mask_cost(in, dtw_cost_rcp(xp1[i], xr[i], yp1, yr));
                    double mv, c_up = 0.0, c_left = 0.0, c_diag = 0.0;
                    if (FINEST) {
                        // only the value is needed: rounding is monotone, so the smallest of the three rounded
                        // sums is the rounded sum of the smallest candidate -- one add instead of three
                        mv = fmin(fmin(up, old), diag) + dt;
                    } else {
                        c_up = up + dt; c_left = old + dt; c_diag = diag + dt;
                        mv = fmin(fmin(c_up, c_left), c_diag);               // two v_min_f64 (no NaNs here)
                    }
                    const double nv = mv;                                    // (outside the window: >= 2^1023, see dtw_mask_cost)
                    if (!FINEST) {
                        // predecessor = the first candidate, in the tie order, that attains the minimum
                        int best;
                        if (TIE == 0) best = c_up == mv ? 0 : (c_left == mv ? 1 : 2);        // (i-1,j), (i,j-1), (i-1,j-1)
                        else if (TIE == 1) best = c_diag == mv ? 2 : (c_up == mv ? 0 : 1);    // (i-1,j-1), (i-1,j), (i,j-1)
                        else best = (diag <= up && diag <= old) ? 2 : (up <= old ? 0 : 1);    // on the predecessor costs, <=
                        word |= in ? ((word_t)best << (2 * i)) : (word_t)0;
                    }
#else
                    double nv = INF;
                    if (j >= lo && j <= hi) {
                        const double dt = dtw_cost_rcp(xp1[i], xr[i], yp1, yr);
                        const double c_up = up + dt, c_left = old + dt, c_diag = diag + dt;
                        nv = fmin(fmin(c_up, c_left), c_diag);               // two v_min_f64 (no NaNs here)
                        if (!FINEST) {
                            // predecessor = the first candidate, in the tie order, that attains the minimum
                            int best;
                            if (TIE == 0) best = c_up == nv ? 0 : (c_left == nv ? 1 : 2);    // (i-1,j), (i,j-1), (i-1,j-1)
                            else if (TIE == 1) best = c_diag == nv ? 2 : (c_up == nv ? 0 : 1);   // (i-1,j-1), (i-1,j), (i,j-1)
                            else best = (diag <= up && diag <= old) ? 2 : (up <= old ? 0 : 1);
                            word |= (word_t)best << (2 * i);
                        }
                    }
#endif
                    col[i] = nv;
                    diag = old;                                              // (i, j-1) is the diagonal of (i+1, j)
                    up = nv;
                }
            } else {                                                         // every cell of the block is INF
                up = INF;                                                    // in this column and the previous one
                diag = INF;
            }
        }
        if (!finest) {                                                       // the finest level is never backtracked
            if (WLDS) wl[j * DTW_THREADS] = (uint32_t)word; else wq[(int64_t)j * NT] = (uint64_t)word;
        }
    }
    double result = 0.0;
#pragma unroll
    for (int i = 0; i < RR; ++i) if (i == lx - 1) result = col[i];
    if (finest) return result;
    // backtrack through the predecessor codes; record the path's column range per row
#pragma unroll
    for (int q = 0; q < RR; ++q) FLQ(q) = 0xffff;                            // first = 65535, last = 0
    int i = lx - 1, j = ly - 1;
    while (i >= 0 && j >= 0) {
        const int v = FLQ(i);
        int f = v & 0xffff, l = v >> 16;
        f = j < f ? j : f;
        l = j > l ? j : l;
        FLQ(i) = (l << 16) | f;
        const uint64_t word = WLDS ? (uint64_t)wl[j * DTW_THREADS] : wq[(int64_t)j * NT];
        const int d = (int)((word >> (2 * i)) & 3);
        if (d == 0) --i; else if (d == 1) --j; else { --i; --j; }
    }
    return result;
#undef FLQ
}

// RMAX = rows the instantiation can hold (12 / 20 / 32): the register budget -- and with it the
// number of resident wavefronts that hide the fp64 dependency chains -- follows the longest
// component of the call, not the longest the kernel family supports.
template <int RMAX, int TIE, int MINB, bool WLDS>
__global__ __launch_bounds__(DTW_THREADS, MINB) void dtw_similarity_reg_kernel(
    const double* __restrict__ xpyr, const int32_t* __restrict__ xlen, int64_t n_x,
    const double* __restrict__ ypyr, const int32_t* __restrict__ ylen, int64_t n_y,
    float* __restrict__ out, uint64_t* __restrict__ wq, DtwLayout L, const int32_t* __restrict__ x_order,
    const int64_t* __restrict__ x_live)
{
    __shared__ int32_t s_fl[RMAX * DTW_THREADS];
    __shared__ int32_t s_ublk[(DTW_THREADS / 64) * 16];                   // per wavefront: union block ranges
    extern __shared__ uint32_t s_words[];                    // WLDS: (max_y_len / 2) x DTW_THREADS predecessor words
    int32_t* fl = s_fl + threadIdx.x;
    uint32_t* wl = s_words + threadIdx.x;
    const int64_t NT = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // x_live = {first, count}: only these positions of the processing order hold non-empty rows (the caller
    // sorted the empty ones to the front and zeroed their output rows) -- the pairs are then dealt out over
    // the live rows only; with thousands of empty rows in front the lanes' shares were very uneven
    const int64_t first_live = x_live ? x_live[0] : 0;
    const int64_t n_live = x_live ? x_live[1] : n_x;
    const int64_t total = n_live * n_y;
    for (int64_t pair = tid; pair < total; pair += NT) {
        // consecutive lanes: same anchor, consecutive components of the caller's processing order
        // (similar series side by side keep the lanes' windows aligned)
        const int64_t a = pair / n_live;
        const int64_t pos = first_live + pair % n_live;                    // position in the processing order:
        const int64_t r = x_order ? x_order[pos] : pos;                    // the pyramids are laid out by position
        const int lx0 = xlen[pos], ly0 = ylen[a];
        if (lx0 == 0 || ly0 == 0) { out[r * n_y + a] = 0.f; continue; }
        const double* __restrict__ yp = ypyr + a * L.YL;
        int n_levels = 1;
        {
            int lx = lx0, ly = ly0;
            while (lx >= 3 && ly >= 3) { lx >>= 1; ly >>= 1; ++n_levels; }
        }
        double result = 0.0;
        for (int lev = n_levels - 1; lev >= 0; --lev) {
            const int lx = lx0 >> lev, ly = ly0 >> lev;
            const int lxc = lx0 >> (lev + 1), lyc = ly0 >> (lev + 1);
            const double* xcol = xpyr + L.xoff[lev] * n_x + pos;
            const double* xrcol = xcol + L.XL * n_x;                        // reciprocal pyramids follow the values
            const double* ycol = yp + L.yoff[lev];
            const double* yrcol = ycol + L.YL * n_y;
            uint64_t* w = wq + L.yoff[lev] * NT + tid;
            const bool coarsest = lev == n_levels - 1, finest = lev == 0;
#define DTW_LEVEL_F(RR, F) result = dtw_reg_level<RMAX, (RR) <= RMAX ? (RR) : RMAX, TIE, WLDS, F>(fl, xcol, xrcol, n_x, ycol, yrcol, lx, ly, lxc, lyc, coarsest, s_ublk, wl, w, NT)
#define DTW_LEVEL(RR) do { if (finest) DTW_LEVEL_F(RR, true); else DTW_LEVEL_F(RR, false); } while (0)
            if (lx <= 4) DTW_LEVEL(4);                       // narrow instantiations: the unrolled row
            else if (lx <= 8) DTW_LEVEL(8);                  // loop sweeps at most 3 empty rows
            else if (lx <= 12) DTW_LEVEL(12);
            else if (RMAX > 12 && lx <= 16) DTW_LEVEL(16);
            else if (RMAX > 12 && lx <= 20) DTW_LEVEL(20);
            else if (RMAX > 20 && lx <= 24) DTW_LEVEL(24);
            else if (RMAX > 20 && lx <= 28) DTW_LEVEL(28);
            else if (RMAX > 20) DTW_LEVEL(32);
#undef DTW_LEVEL
#undef DTW_LEVEL_F
        }
        out[r * n_y + a] = (float)(1.0 / (result + 1.0));
    }
}

// Processing-order key of the x rows of a DTW call: (length, the row's TWICE-HALVED series -- means of four
// consecutive entries, what fastdtw's second coarsening level sees -- on a log scale, 8 steps per octave, up to six of
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
    auto q8 = [](float v) { const int q = (int)lrintf(8.f * log2f(1.f + (v < 0.f ? 0.f : v))); return (int64_t)(q > 255 ? 255 : q); };
    const int64_t n2 = len / 4;
    if (n2 == 0) {
        for (int64_t f = 0; f < len; ++f) key |= q8((float)x_val[b + f]) << (40 - 8 * f);
    } else {
        const int64_t nf = n2 < 6 ? n2 : 6;
        for (int64_t f = 0; f < nf; ++f) {
            const int64_t g = b + 4 * ((f * n2) / nf);
            const float v = 0.25f * ((float)x_val[g] + (float)x_val[g + 1] + (float)x_val[g + 2] + (float)x_val[g + 3]);
            key |= q8(v) << (40 - 8 * f);
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

static int g_dtw_force_general = 0;
/* test hook: 1 = always take the general (workspace-resident) kernel, 0 = pick by size */
extern "C" int sgnn_dtw_force_general(int on) { const int old = g_dtw_force_general; g_dtw_force_general = on; return old; }

static int dtw_run(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t max_x_len,
                   const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int64_t max_y_len,
                   int tie_order, const int32_t* x_order, const int64_t* x_live, float* out, void* workspace,
                   int64_t workspace_bytes, void* stream)
{
    if (!x_ptr || !x_val || !y_ptr || !y_val || !out || !workspace || n_x < 0 || n_y < 0) return SGNN_ERR_BAD_ARG;
    if (tie_order < 0 || tie_order > 2) return SGNN_ERR_BAD_ARG;
    if (max_x_len < 1) max_x_len = 1;
    if (max_y_len < 1) max_y_len = 1;
    if (max_x_len > 32767 || max_y_len > 32767) return SGNN_ERR_SET_TOO_LARGE;
    if (workspace_bytes < sgnn_dtw_workspace_bytes(n_x, max_x_len, n_y, max_y_len)) return SGNN_ERR_BAD_ARG;
    if (n_x * n_y == 0) return SGNN_OK;
    const DtwLayout L = dtw_layout(max_x_len, max_y_len);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    const int64_t lane = L.n_dbl * 8 + dtw_align8(L.n_i32 * 4) + dtw_align8(L.n_dir * 4);
    int64_t scratch = DTW_NT * lane;
    if (DTW_REG_NT * L.YL * 8 > scratch) scratch = DTW_REG_NT * L.YL * 8;
    double* wd = (double*)w;
    int32_t* wi = (int32_t*)(w + DTW_NT * L.n_dbl * 8);
    uint32_t* wb = (uint32_t*)(w + DTW_NT * (L.n_dbl * 8 + dtw_align8(L.n_i32 * 4)));
    uint64_t* wq = (uint64_t*)w;           w += scratch;
    double* xpyr = (double*)w;             w += 2 * n_x * L.XL * 8;          // values, then reciprocals of value + 1
    double* ypyr = (double*)w;             w += 2 * n_y * L.YL * 8;
    int32_t* xlen = (int32_t*)w;           w += dtw_align8(n_x * 4);
    int32_t* ylen = (int32_t*)w;
    const bool use_reg = max_x_len <= DTW_R && !g_dtw_force_general;
    hipLaunchKernelGGL(dtw_pyramid_kernel, dim3(sgnn_grid_for(n_x, 256)), dim3(256), 0, st, x_ptr, x_val, n_x,
                       max_x_len, L.XL, 1, xpyr, xpyr + n_x * L.XL, xlen, use_reg ? x_order : (const int32_t*)nullptr);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(dtw_pyramid_kernel, dim3(sgnn_grid_for(n_y, 256)), dim3(256), 0, st, y_ptr, y_val, n_y,
                       max_y_len, L.YL, 0, ypyr, ypyr + n_y * L.YL, ylen, (const int32_t*)nullptr);
    SGNN_CHECK_LAUNCH();
    if (use_reg) {
        // predecessor words of the coarse levels in LDS when (max_y_len / 2) words per lane fit
        const int64_t words = max_y_len >> 1;
        const bool wlds = words * DTW_THREADS * 4 <= 48 * 1024;
        const size_t dyn = wlds ? (size_t)((words > 0 ? words : 1) * DTW_THREADS * 4) : 0;
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
                                   int tie_order, const int32_t* x_order, float* out, void* workspace,
                                   int64_t workspace_bytes, void* stream)
{
    return dtw_run(x_ptr, x_val, n_x, max_x_len, y_ptr, y_val, n_y, max_y_len, tie_order, x_order, nullptr, out, workspace,
                   workspace_bytes, stream);
}

extern "C" int sgnn_dtw_similarity_live(const int64_t* x_ptr, const int32_t* x_val, int64_t n_x, int64_t max_x_len,
                                        const int64_t* y_ptr, const int32_t* y_val, int64_t n_y, int64_t max_y_len,
                                        int tie_order, const int32_t* x_order, const int64_t* x_live_range, float* out,
                                        void* workspace, int64_t workspace_bytes, void* stream)
{
    if (x_live_range && !x_order) return SGNN_ERR_BAD_ARG;
    return dtw_run(x_ptr, x_val, n_x, max_x_len, y_ptr, y_val, n_y, max_y_len, tie_order, x_order, x_live_range, out,
                   workspace, workspace_bytes, stream);
}
