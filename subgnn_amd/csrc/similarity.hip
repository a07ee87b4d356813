// Similarity kernels: shortest-path similarities (a9, dense-parity and sparse forms); the structure
// similarity 1/(1+fastdtw) (a11) lives in dtw.hip.
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
// Row stride (in 64-bit words) of the per-node arrays seen / frontier / next: three words (129-192 sources, the benchmark's
// 183) are padded to four -- a pull level gathers one row per neighbour, and a 24-byte row straddles two 32-byte sectors
// every other time (1.5 sectors per gather on average), a 32-byte row is always one
static inline int64_t msbfs_row_stride(int64_t n_words) { return n_words == 3 ? 4 : n_words; }

#define MSBFS_DEFAULT_ALPHA 32     // pull when frontier word-edges * alpha > nnz * n_words; 0 = never pull

__global__ void msbfs_init_kernel(const int32_t* __restrict__ sources, int64_t n_sources, int64_t n_words,
                                  int64_t n_ids, uint64_t* __restrict__ seen, uint64_t* __restrict__ frontier,
                                  uint64_t* __restrict__ next, uint8_t* __restrict__ dist, int32_t* __restrict__ flags,
                                  unsigned long long* __restrict__ fvol, uint32_t* __restrict__ fbits,
                                  int max_hops, int64_t rs,  // dist layout-agnostic: filled as a flat array
                                  uint64_t* __restrict__ set_seen, int64_t n_set_words, float* __restrict__ set_out, int64_t n_set_out)
{
    const int64_t gtid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = gtid; i < 2 * ((n_ids + 31) / 32); i += gsz) fbits[i] = 0;   // frontier nodes | complete nodes
    for (int64_t i = gtid; i < n_ids * rs; i += gsz) { seen[i] = 0; frontier[i] = 0; next[i] = 0; }
    // (the fused set reduction's state too: four memset launches less per search)
    for (int64_t i = gtid; i < n_set_words; i += gsz) set_seen[i] = 0;
    for (int64_t i = gtid; i < n_set_out; i += gsz) set_out[i] = 0.f;              // unreachable pairs hold 0
    if (dist) for (int64_t i = gtid; i < n_sources * n_ids; i += gsz) dist[i] = 255;
    for (int64_t i = gtid; i <= max_hops; i += gsz) { flags[i] = (i == 0) ? 1 : 0; fvol[i] = 0; }
}

__global__ void msbfs_seed_kernel(const int32_t* __restrict__ sources, int64_t n_sources, int64_t n_words,
                                  int64_t n_ids, uint64_t* __restrict__ seen, uint64_t* __restrict__ frontier,
                                  uint8_t* __restrict__ dist, int64_t ss, int64_t sv, uint32_t* __restrict__ fbits, int64_t rs)
{
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= n_sources) return;
    const int32_t v = sources[s];
    atomicOr(&fbits[v >> 5], 1u << (v & 31));                 // level-0 frontier nodes
    const uint64_t bit = 1ull << (s & 63);
    atomicOr((unsigned long long*)&seen[(int64_t)v * rs + (s >> 6)], (unsigned long long)bit);
    atomicOr((unsigned long long*)&frontier[(int64_t)v * rs + (s >> 6)], (unsigned long long)bit);
    if (dist) dist[s * ss + v * sv] = 0;
}

#define MSBFS_WCHUNK 4          // source words a pull pass keeps in registers
#define MSBFS_HUB_DEGREE 128     // push: lists this long go to the whole workgroup
#define MSBFS_HUB_SLOTS 128
#define MSBFS_PULL_HUB_DEGREE 512

__device__ __forceinline__ uint32_t msbfs_row_or32(uint32_t v)
{
    // OR over the 16 lanes of a DPP row by rotations: every lane ends with the full value (no LDS crossbar trips:
    // the xor-butterfly on __shfl_xor was 8 ds_bpermute per 64-bit word and call)
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);     // row_ror:8
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);     // row_ror:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);     // row_ror:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);     // row_ror:1
    return v;
}

__device__ __forceinline__ uint64_t msbfs_group_or(uint64_t x)
{
    // OR over the 16 lanes of a group (= one DPP row; all lanes of the wavefront must be executing)
    return ((uint64_t)msbfs_row_or32((uint32_t)(x >> 32)) << 32) | msbfs_row_or32((uint32_t)x);
}

// Per level, for every set: which sources reached one of its members for the first time?  The new
// frontier words of the members are OR-ed, masked with what the set has not seen yet, and the level
// is written for those sources: out[set, source] = min over members of the hop distance, without a
// (sources x nodes) hop table.  One thread per (set, word).  Round 4: not a launch of its own any more -- level L's
// reduction is the prologue of level L + 1's expand launch (both only READ the frontier words level L's commit wrote; the
// reduction writes set_seen / out, which no expand touches) and the last level's runs inside the finalize launch, thread for
// thread in front of the finalisation of the same (set, word): 3 launches per level -> 2.
__device__ __forceinline__ void msbfs_set_reduce_item(
    const uint64_t* __restrict__ frontier, int64_t n_words, int64_t n_sources,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes,
    uint64_t* __restrict__ set_seen, float* __restrict__ out, int level, int64_t rs, int64_t t)
{
    const int64_t r = t / n_words, w = t % n_words;
    uint64_t acc = 0;
    const int64_t left_ = n_sources - w * 64;
    const uint64_t full_ = left_ >= 64 ? ~0ull : ((1ull << left_) - 1);
    const uint64_t have_ = set_seen[t];
    if ((have_ & full_) == full_) return;                    // every source has reached the set: nothing left to record
    for (int64_t i = set_ptr[r]; i < set_ptr[r + 1]; ++i) acc |= frontier[(int64_t)set_nodes[i] * rs + w];
    const uint64_t fresh = acc & ~have_;
    if (fresh) {
        set_seen[t] = have_ | fresh;
        uint64_t bits = fresh;
        while (bits) {
            const int b = __ffsll((unsigned long long)bits) - 1;
            bits &= bits - 1;
            const int64_t s = w * 64 + b;
            if (s < n_sources) out[r * n_sources + s] = (float)level;
        }
    }
}

struct MsbfsSets {                 // the fused set reduction's operands (n_sets = 0: none)
    const int64_t* set_ptr;
    const int32_t* set_nodes;
    int64_t n_sets;
    uint64_t* set_seen;
    float* out;
};

__global__ __launch_bounds__(256) void msbfs_expand_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_ids, int64_t n_words,
    int64_t n_sources, const uint64_t* __restrict__ seen, const uint64_t* __restrict__ frontier,
    uint64_t* __restrict__ next, const int32_t* __restrict__ flags, const unsigned long long* __restrict__ fvol,
    unsigned long long pull_above, int level, const uint32_t* __restrict__ fnode, unsigned long long sparse_below,
    const uint32_t* __restrict__ fdone, int64_t rs, MsbfsSets sets)
{
    if (level > 1 && flags[level - 1] == 0) return;          // previous level found nothing (nothing to reduce either)
    if (sets.n_sets > 0) {                                   // the previous level's set reduction (level 0: the seeds)
        const int64_t total = sets.n_sets * n_words;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
            msbfs_set_reduce_item(frontier, n_words, n_sources, sets.set_ptr, sets.set_nodes, sets.set_seen, sets.out, level - 1, rs, t);
    }
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
            for (int64_t w = 0; w < n_words; ++w) any |= frontier[v * rs + w];
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
                    const uint64_t f = frontier[v * rs + w];
                    const uint64_t m = f & ~seen[u * rs + w];
                    if (m) atomicOr((unsigned long long*)&next[u * rs + w], (unsigned long long)m);
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
                    const uint64_t f = frontier[v * rs + w];
                    const uint64_t m = f & ~seen[u * rs + w];
                    if (m) atomicOr((unsigned long long*)&next[u * rs + w], (unsigned long long)m);
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
                    valid &= ~seen[v * rs + w];
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
                        if (rs == 4) {                       // a padded row = one 32-byte sector: two 16-byte loads, no per-word tests
                            const ulonglong2 f01 = *reinterpret_cast<const ulonglong2*>(&frontier[u * 4]);
                            const ulonglong2 f23 = *reinterpret_cast<const ulonglong2*>(&frontier[u * 4 + 2]);
                            acc[0] |= f01.x; acc[1] |= f01.y; acc[2] |= f23.x; acc[3] |= f23.y;
                        } else {
#pragma unroll
                            for (int k = 0; k < MSBFS_WCHUNK; ++k)
                                if (need[k]) acc[k] |= frontier[u * rs + w0 + k];  // completed words are not read
                        }
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
                    if (m) next[v * rs + w0 + k] = m;   // next[] is all zero before a pull level
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
                    valid &= ~seen[v * rs + w];
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
                    if (rs == 4) {
                        const ulonglong2 f01 = *reinterpret_cast<const ulonglong2*>(&frontier[u * 4]);
                        const ulonglong2 f23 = *reinterpret_cast<const ulonglong2*>(&frontier[u * 4 + 2]);
                        acc[0] |= f01.x; acc[1] |= f01.y; acc[2] |= f23.x; acc[3] |= f23.y;
                    } else {
#pragma unroll
                        for (int k = 0; k < MSBFS_WCHUNK; ++k)
                            if (need[k]) acc[k] |= frontier[u * rs + w0 + k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < MSBFS_WCHUNK; ++k)
                if (acc[k]) atomicOr(&s_acc[k], (unsigned long long)acc[k]);
            __syncthreads();
            if (threadIdx.x < MSBFS_WCHUNK && w0 + threadIdx.x < n_words) {
                const uint64_t m = s_acc[threadIdx.x] & need[threadIdx.x];
                if (m) next[v * rs + w0 + threadIdx.x] = m;
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void msbfs_commit_kernel(
    const int64_t* __restrict__ rowptr, int64_t n_ids, int64_t n_words, int64_t n_sources, uint64_t* __restrict__ seen,
    uint64_t* __restrict__ frontier, uint64_t* __restrict__ next, uint8_t* __restrict__ dist, int32_t* __restrict__ flags,
    unsigned long long* __restrict__ fvol, int level, int64_t ss, int64_t sv, uint32_t* __restrict__ fcur,
    uint32_t* __restrict__ fdone, int64_t rs)
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
        if (v < n_ids && rs == 4 && !dist) {
            // a padded row is one 32-byte sector: the three arrays are read and written with 16-byte accesses (one 8-byte
            // word per lane and array left the loads a quarter-filled: 50 us per level for 130 MB)
            const ulonglong2 n01 = *reinterpret_cast<const ulonglong2*>(&next[v * 4]);
            const ulonglong2 n23 = *reinterpret_cast<const ulonglong2*>(&next[v * 4 + 2]);
            const ulonglong2 s01 = *reinterpret_cast<const ulonglong2*>(&seen[v * 4]);
            const ulonglong2 s23 = *reinterpret_cast<const ulonglong2*>(&seen[v * 4 + 2]);
            const uint64_t nx[4] = {n01.x, n01.y, n23.x, n23.y}, sn[4] = {s01.x, s01.y, s23.x, s23.y};
            uint64_t nw[4];
            uint64_t any_nx = 0, any_nw = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                nw[w] = w < n_words ? (nx[w] & ~sn[w]) : 0;
                any_nx |= nx[w];
                any_nw |= nw[w];
                if (w < n_words) {
                    const int64_t left = n_sources - w * 64;
                    done = done && ((sn[w] | nw[w]) == (left >= 64 ? ~0ull : ((1ull << left) - 1)));
                    if (nw[w]) ++n_new;
                }
            }
            if (any_nx) {
                *reinterpret_cast<ulonglong2*>(&next[v * 4]) = make_ulonglong2(0, 0);
                *reinterpret_cast<ulonglong2*>(&next[v * 4 + 2]) = make_ulonglong2(0, 0);
            }
            *reinterpret_cast<ulonglong2*>(&frontier[v * 4]) = make_ulonglong2(nw[0], nw[1]);
            *reinterpret_cast<ulonglong2*>(&frontier[v * 4 + 2]) = make_ulonglong2(nw[2], nw[3]);
            if (any_nw) {
                *reinterpret_cast<ulonglong2*>(&seen[v * 4]) = make_ulonglong2(sn[0] | nw[0], sn[1] | nw[1]);
                *reinterpret_cast<ulonglong2*>(&seen[v * 4 + 2]) = make_ulonglong2(sn[2] | nw[2], sn[3] | nw[3]);
            }
            if (n_new) {
                any = true;
                vol += (unsigned long long)n_new * (unsigned long long)(rowptr[v + 1] - rowptr[v]);
            }
        } else if (v < n_ids) {
            for (int64_t w = 0; w < n_words; ++w) {
                const int64_t i = v * rs + w;
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
    return 3 * (max_id + 1) * msbfs_row_stride(n_words) * 8 + ((int64_t)max_hops + 2) * 8 + ((int64_t)max_hops + 2) * 4 +
           8 + 2 * ((max_id + 32) / 32) * 4;                      // + two frontier-node bitmaps
}

// The reference's matrix holds 0 for unreachable pairs, and its row-min runs over those zeros: a
// source that never reaches SOME member of a set gives 0 for the whole set (SubGNN.py:772).  After
// the last level: AND the members' seen words, zero the sources missing from it.
__global__ __launch_bounds__(256) void msbfs_set_finalize_kernel(
    const uint64_t* __restrict__ seen, int64_t n_words, int64_t n_sources,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets, float* __restrict__ out, int64_t rs,
    const uint64_t* __restrict__ frontier, uint64_t* __restrict__ set_seen, const int32_t* __restrict__ flags, int last_level,
    int32_t* __restrict__ status)
{
    if (status && blockIdx.x == 0 && threadIdx.x == 0) {     // (msbfs_status_kernel's work: one launch less per search)
        int last = 0;
        for (int l = 1; l <= last_level; ++l) if (flags[l]) last = l;
        status[0] = last;
        status[1] = flags[last_level] != 0;
    }
    const int64_t total = n_sets * n_words;
    const bool reduce_last = flags[last_level] != 0;         // the last enqueued level found something: its reduction is still due
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        if (reduce_last)
            msbfs_set_reduce_item(frontier, n_words, n_sources, set_ptr, set_nodes, set_seen, out, last_level, rs, t);
        const int64_t r = t / n_words, w = t % n_words;
        uint64_t all = ~0ull;
        for (int64_t i = set_ptr[r]; i < set_ptr[r + 1]; ++i) all &= seen[(int64_t)set_nodes[i] * rs + w];
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
                     void* workspace, hipStream_t st, int pull_alpha, int32_t* status = nullptr)
{
    const int g_bfs_alpha = pull_alpha < 0 ? MSBFS_DEFAULT_ALPHA : pull_alpha;
    const int64_t n_ids = max_id + 1;
    const int64_t n_words = (n_sources + 63) / 64;
    const int64_t ss = node_major ? 1 : n_ids, sv = node_major ? n_sources : 1;
    const int64_t rs = msbfs_row_stride(n_words);
    uint64_t* seen = (uint64_t*)workspace;
    uint64_t* frontier = seen + n_ids * rs;
    uint64_t* next = frontier + n_ids * rs;
    unsigned long long* fvol = (unsigned long long*)(next + n_ids * rs);
    int32_t* flags = (int32_t*)(fvol + max_hops + 2);
    uint32_t* fbits = (uint32_t*)(((uintptr_t)(flags + max_hops + 2) + 7) & ~(uintptr_t)7);
    const int64_t fwords = (n_ids + 31) / 32;
    uint64_t* set_seen = (uint64_t*)(((uintptr_t)(fbits + 2 * fwords) + 7) & ~(uintptr_t)7);
    const unsigned long long pull_above =
        g_bfs_alpha > 0 ? (unsigned long long)((nnz * n_words) / g_bfs_alpha) : ~0ull;
    const int big = sgnn_grid_for(n_ids * ((dist && n_sources > rs) ? n_sources : rs), 256);
    hipLaunchKernelGGL(msbfs_init_kernel, dim3(big), dim3(256), 0, st, sources, n_sources, n_words, n_ids, seen,
                       frontier, next, dist, flags, fvol, fbits, max_hops, rs, set_seen, set_out ? n_sets * n_words : 0,
                       set_out, set_out ? n_sets * n_sources : 0);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(msbfs_seed_kernel, dim3((int)((n_sources + 255) / 256)), dim3(256), 0, st, sources, n_sources,
                       n_words, n_ids, seen, frontier, dist, ss, sv, fbits, rs);
    SGNN_CHECK_LAUNCH();
    const int g_sets = set_out ? sgnn_grid_for(n_sets * n_words, 256) : 0;
    MsbfsSets sets;
    sets.set_ptr = set_ptr; sets.set_nodes = set_nodes; sets.n_sets = set_out ? n_sets : 0; sets.set_seen = set_seen; sets.out = set_out;
    const int g_expand = sgnn_grid_for(n_ids * 16, 256);
    const int g_commit = sgnn_grid_for(n_ids, 256, 256 * 4);
    for (int level = 1; level <= max_hops; ++level) {
        hipLaunchKernelGGL(msbfs_expand_kernel, dim3(g_expand), dim3(256), 0, st, rowptr, col, n_ids, n_words, n_sources,
                           seen, frontier, next, flags, fvol, pull_above, level, fbits,
                           (unsigned long long)((nnz * n_words) / 4), fbits + fwords, rs, sets);
        SGNN_CHECK_LAUNCH();
        hipLaunchKernelGGL(msbfs_commit_kernel, dim3(g_commit), dim3(256), 0, st, rowptr, n_ids, n_words, n_sources, seen,
                           frontier, next, dist, flags, fvol, level, ss, sv, fbits, fbits + fwords, rs);
        SGNN_CHECK_LAUNCH();
    }
    if (set_out) {
        hipLaunchKernelGGL(msbfs_set_finalize_kernel, dim3(g_sets), dim3(256), 0, st, seen, n_words, n_sources, set_ptr,
                           set_nodes, n_sets, set_out, rs, frontier, set_seen, flags, max_hops, status);
        SGNN_CHECK_LAUNCH();
    }
    if (status && !set_out) {
        hipLaunchKernelGGL(msbfs_status_kernel, dim3(1), dim3(64), 0, st, flags, max_hops, status);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

extern "C" int sgnn_bfs_hops(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                             const int32_t* sources, int64_t n_sources, int max_hops, int node_major, int pull_alpha,
                             uint8_t* dist, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!rowptr || !col || !sources || !dist || !workspace || n_sources < 0 || max_hops < 1 || max_hops > 254)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (workspace_bytes < sgnn_bfs_hops_workspace_bytes(max_id, n_sources, max_hops)) return SGNN_ERR_BAD_ARG;
    if (n_sources == 0) return SGNN_OK;
    return msbfs_run(rowptr, col, nnz, max_id, sources, n_sources, max_hops, node_major, dist, nullptr, nullptr, 0,
                     nullptr, workspace, (hipStream_t)stream, pull_alpha);
}

extern "C" int64_t sgnn_bfs_min_hops_workspace_bytes(int64_t max_id, int64_t n_sources, int max_hops, int64_t n_sets) {
    const int64_t n_words = (n_sources + 63) / 64;
    return sgnn_bfs_hops_workspace_bytes(max_id, n_sources, max_hops) + 16 + n_sets * n_words * 8;
}

extern "C" int sgnn_bfs_min_hops_to_sets(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                                         const int32_t* sources, int64_t n_sources, int max_hops, int pull_alpha,
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
                     workspace, (hipStream_t)stream, pull_alpha, out_status);
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
