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
// Round 5: a PULL level is ONE launch and the search stays in pull mode once it has entered it.
//   * what a pulling node needs from a neighbour u is "has source s reached u before this level" -- seen[u], not the
//     frontier: for a node v that s has not reached, a neighbour with dist(u, s) <= level - 2 cannot exist (v would have been
//     reached a level earlier), so seen and frontier give the same answer.  The level therefore reads ONE version of the
//     seen rows and writes the next version into another buffer (seen' = seen | new): no next[] accumulation, no commit pass
//     over three 32 MB arrays per level (30-34 us each on the benchmark), no zeroing.
//   * the three per-node arrays (seen / next / frontier) rotate as versions: pull number k reads B[k % 3], writes B[(k+1) % 3];
//     B[(k+2) % 3] -- the version before -- is what the fused set reduction of the NEXT launch subtracts to get the bits that
//     were new (it runs as that launch's prologue, while the launch already overwrites the oldest version: two buffers would
//     race).  Rows of nodes that every source has reached are never rewritten (the done bitmap skips them): they go stale in
//     two of the three buffers, which nobody notices -- all neighbours of a complete node are complete one level later and
//     skipped too, the set reduction masks with what a set has already recorded, and the finalisation ORs the three versions
//     (every version is a subset of the truth and the newest write of a row is exact).
//   * push levels walk the frontier-node BITMAP (125 KB) to find their nodes instead of reading every node's frontier row
//     (32 MB): level 1 of the benchmark's search went from 55 us to the time of its 183 lists.
//   * host side: a commit launch is enqueued only for the first ``push_levels`` levels (the caller's hint, from the status of
//     an earlier search: [2] = the level that switched to pull); beyond them the device pulls whatever the frontier's size.
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
    for (int64_t i = gtid; i < 2 * ((n_ids + 31) / 32) + (n_ids + 3) / 4; i += gsz) fbits[i] = 0;   // frontier-node bits | ever-on-the-frontier bits | one "complete" BYTE per node
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
    atomicOr(&fbits[((n_ids + 31) / 32) + (v >> 5)], 1u << (v & 31));   // ... which have a non-empty seen row from now on
    const uint64_t bit = 1ull << (s & 63);
    atomicOr((unsigned long long*)&seen[(int64_t)v * rs + (s >> 6)], (unsigned long long)bit);
    atomicOr((unsigned long long*)&frontier[(int64_t)v * rs + (s >> 6)], (unsigned long long)bit);
    if (dist) dist[s * ss + v * sv] = 0;
}

#define MSBFS_WCHUNK 4          // source words a pull pass keeps in registers
#define MSBFS_HUB_DEGREE 128     // push: lists this long go to the whole workgroup
#define MSBFS_HUB_SLOTS 128
#define MSBFS_PULL_HUB_DEGREE 512
#ifndef MSBFS_PULL_CHECK
#define MSBFS_PULL_CHECK 2        // pull: rounds of 16 neighbours between two tests of "is anything still missing" (round 4: 8)
#endif

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
    uint64_t* __restrict__ set_seen, float* __restrict__ out, int level, int64_t rs, int64_t t,
    const uint64_t* __restrict__ older = nullptr)
{
    // ``older`` == nullptr: ``frontier`` holds the level's new bits (a push level's commit wrote them); else ``frontier`` is the
    // seen version the level wrote and ``older`` the one it read: the new bits are their difference
    const int64_t r = t / n_words, w = t % n_words;
    uint64_t acc = 0;
    const int64_t left_ = n_sources - w * 64;
    const uint64_t full_ = left_ >= 64 ? ~0ull : ((1ull << left_) - 1);
    const uint64_t have_ = set_seen[t];
    if ((have_ & full_) == full_) return;                    // every source has reached the set: nothing left to record
    if (older) {
        for (int64_t i = set_ptr[r]; i < set_ptr[r + 1]; ++i) {
            const int64_t o = (int64_t)set_nodes[i] * rs + w;
            acc |= frontier[o] & ~older[o];
        }
    } else
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

struct MsbfsBufs { uint64_t* b[3]; };      // seen, next, frontier: in pull mode three versions of the seen rows (see above)

// Which form level ``level`` takes, the same answer in every launch that asks: pull(l) for l >= 2 is sticky --
// pull(l - 1), or the frontier volume level l - 1 left behind exceeds the threshold, or l lies beyond the levels the
// host enqueued a commit launch for.  pulls_before = number of pull levels in front of ``level`` (the version rotation's k).
__device__ __forceinline__ void msbfs_mode(const unsigned long long* __restrict__ fvol, unsigned long long pull_above,
                                           int push_levels, int level, bool& pull, bool& prev_pull, int& pulls_before)
{
    bool p = false, pp = false;
    int k = 0;
    for (int l = 2; l <= level; ++l) {
        const bool pl = p || l > push_levels || (pull_above != ~0ull && fvol[l - 1] > pull_above);
        if (l < level) k += pl ? 1 : 0;
        pp = p;
        p = pl;
    }
    pull = p; prev_pull = (level >= 2) ? pp : false; pulls_before = k;
}

__global__ __launch_bounds__(256) void msbfs_level_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_ids, int64_t n_words,
    int64_t n_sources, MsbfsBufs B, int32_t* __restrict__ flags, const unsigned long long* __restrict__ fvol,
    unsigned long long pull_above, int push_levels, int level, const uint32_t* __restrict__ fnode,
    const uint32_t* __restrict__ fany, uint8_t* __restrict__ fdone, int64_t rs, MsbfsSets sets, uint8_t* __restrict__ dist,
    int64_t ss, int64_t sv)
{
    if (level > 1 && flags[level - 1] == 0) return;          // previous level found nothing (nothing to reduce either)
    bool pull, prev_pull;
    int k;
    msbfs_mode(fvol, pull_above, push_levels, level, pull, prev_pull, k);
    if (sets.n_sets > 0) {                                   // the previous level's set reduction (level 0: the seeds)
        const int64_t total = sets.n_sets * n_words;
        // the previous level pushed (or seeded): its commit left the new bits in B[2]; it pulled: it wrote version B[k % 3]
        // from B[(k + 2) % 3] -- neither is written by this launch (a pull writes B[(k + 1) % 3], a push B[1] with k = 0)
        const uint64_t* cur = prev_pull ? B.b[k % 3] : B.b[2];
        const uint64_t* old = prev_pull ? B.b[(k + 2) % 3] : nullptr;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
            msbfs_set_reduce_item(cur, n_words, n_sources, sets.set_ptr, sets.set_nodes, sets.set_seen, sets.out, level - 1, rs, t, old);
    }
    const int sub = threadIdx.x & 15;
    // node ids are dealt out to the workgroups round-robin (group g of workgroup b takes b + G*(g + 16 i)):
    // consecutive ids -- the oldest, highest-degree nodes of a preferential-attachment graph sit next to
    // each other at the low ids -- land in different workgroups
    const int64_t group = blockIdx.x + (int64_t)gridDim.x * (threadIdx.x >> 4);
    const int64_t n_groups = (int64_t)gridDim.x * (blockDim.x >> 4);
    __shared__ int32_t s_hub[MSBFS_HUB_SLOTS];
    __shared__ int s_nhub;
    __shared__ unsigned long long s_acc[MSBFS_WCHUNK];
    if (threadIdx.x == 0) s_nhub = 0;
    __syncthreads();
    if (!pull) {
        const uint64_t* __restrict__ seen = B.b[0];
        uint64_t* __restrict__ next = B.b[1];
        const uint64_t* __restrict__ frontier = B.b[2];
        // A frontier node's list is streamed by its 16-lane group -- except long lists (hubs: a BA graph
        // of 1M nodes has lists of 10k+ entries, and hubs are on the frontier from level 1 on), which one
        // group would walk for a millisecond while the rest of the chip is done: those are parked in LDS
        // and streamed by the whole workgroup afterwards.  Frontier nodes are found in the frontier-node bitmap
        // (one bit per node, written by the seed / commit launch), not by reading every node's row.
        for (int64_t v = group; v < n_ids; v += n_groups) {
            if (!((fnode[v >> 5] >> (v & 31)) & 1u)) continue;
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
    // ---- pull: version k -> version k + 1 of the seen rows, in one launch ----------------------------------------------
    const uint64_t* __restrict__ in = B.b[k % 3];
    uint64_t* __restrict__ out = B.b[(k + 1) % 3];
    bool found = false;                                      // this thread recorded a new bit
    // the first pull level behind push levels: the commits kept one bit per node "has ever been on a frontier" = its seen row is
    // not empty (125 KB, L2-resident), consulted before the 32-byte gather -- on the benchmark's level 3 about half of the
    // neighbours still have an empty row.  Later pull levels do not maintain the bits (and hardly any row is empty by then).
    const bool filter = (k == 0);
    for (int64_t v = group; v < n_ids; v += n_groups) {
        // one bit per node: every source has reached it -- from level 3-4 on that is almost every node; such a row is
        // not read, not rewritten (it goes stale in the other versions: harmless, see the header)
        if (fdone[v]) continue;
        const int64_t r0 = rowptr[v], r1 = rowptr[v + 1];
        bool all_done = true, parked = false;
        for (int64_t w0 = 0; w0 < rs && !parked; w0 += MSBFS_WCHUNK) {
            uint64_t have[MSBFS_WCHUNK], need[MSBFS_WCHUNK], acc[MSBFS_WCHUNK];
            uint64_t missing = 0;
            if (rs == 4) {
                const ulonglong2 s01 = *reinterpret_cast<const ulonglong2*>(&in[v * 4]);
                const ulonglong2 s23 = *reinterpret_cast<const ulonglong2*>(&in[v * 4 + 2]);
                have[0] = s01.x; have[1] = s01.y; have[2] = s23.x; have[3] = s23.y;
            } else {
#pragma unroll
                for (int q = 0; q < MSBFS_WCHUNK; ++q) have[q] = (w0 + q < rs) ? in[v * rs + w0 + q] : 0;
            }
#pragma unroll
            for (int q = 0; q < MSBFS_WCHUNK; ++q) {
                const int64_t w = w0 + q;
                uint64_t valid = 0;
                if (w < n_words) {
                    const int64_t left = n_sources - w * 64;
                    valid = left >= 64 ? ~0ull : ((1ull << left) - 1);
                }
                need[q] = valid & ~have[q];
                acc[q] = 0;
                missing |= need[q];
            }
            if (missing != 0 && r1 - r0 >= MSBFS_PULL_HUB_DEGREE) {          // long list: the whole workgroup, below
                int slot = 0;
                if (sub == 0) slot = atomicAdd(&s_nhub, 1);
                slot = __shfl(slot, (threadIdx.x & 63) & ~15, 64);
                if (slot < MSBFS_HUB_SLOTS) {
                    if (sub == 0) s_hub[slot] = (int32_t)v;
                    parked = true;                           // all chunks of this node are done there
                    break;
                }
            }
            if (missing != 0) {
                int since = 0;
                for (int64_t e = r0 + sub; e < r1 + sub; e += 16) {          // uniform trip count per group
                    if (e < r1) {
                        const int64_t u = col[e];
                        if (!filter || ((fany[u >> 5] >> (u & 31)) & 1u)) {
                            if (rs == 4) {                   // a padded row = one 32-byte sector: two 16-byte loads, no per-word tests
                                const ulonglong2 f01 = *reinterpret_cast<const ulonglong2*>(&in[u * 4]);
                                const ulonglong2 f23 = *reinterpret_cast<const ulonglong2*>(&in[u * 4 + 2]);
                                acc[0] |= f01.x; acc[1] |= f01.y; acc[2] |= f23.x; acc[3] |= f23.y;
                            } else {
#pragma unroll
                                for (int q = 0; q < MSBFS_WCHUNK; ++q)
                                    if (need[q]) acc[q] |= in[u * rs + w0 + q];  // completed words are not read
                            }
                        }
                    }
                    if (++since == MSBFS_PULL_CHECK) {       // every MSBFS_PULL_CHECK x 16 neighbours: anything still missing?
                        since = 0;
                        uint64_t left = 0;
#pragma unroll
                        for (int q = 0; q < MSBFS_WCHUNK; ++q) { acc[q] = msbfs_group_or(acc[q]); left |= need[q] & ~acc[q]; }
                        if (left == 0) break;
                    }
                }
            }
            uint64_t fresh[MSBFS_WCHUNK];
#pragma unroll
            for (int q = 0; q < MSBFS_WCHUNK; ++q) {
                acc[q] = missing != 0 ? msbfs_group_or(acc[q]) : 0;
                fresh[q] = acc[q] & need[q];
                if (need[q] & ~fresh[q]) all_done = false;
            }
            // the next version of the row, every word of it (the target buffer holds an older version)
            if (sub == 0) {
                if (rs == 4) {
                    *reinterpret_cast<ulonglong2*>(&out[v * 4]) = make_ulonglong2(have[0] | fresh[0], have[1] | fresh[1]);
                    *reinterpret_cast<ulonglong2*>(&out[v * 4 + 2]) = make_ulonglong2(have[2] | fresh[2], have[3] | fresh[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < MSBFS_WCHUNK; ++q)
                        if (w0 + q < rs) out[v * rs + w0 + q] = have[q] | fresh[q];
                }
            }
#pragma unroll
            for (int q = 0; q < MSBFS_WCHUNK; ++q) {
                if (sub == q && fresh[q]) {
                    found = true;
                    if (dist) {
                        uint64_t bits = fresh[q];
                        while (bits) {
                            const int b = __ffsll((unsigned long long)bits) - 1;
                            bits &= bits - 1;
                            const int64_t sidx = (w0 + q) * 64 + b;
                            if (sidx < n_sources) dist[sidx * ss + v * sv] = (uint8_t)level;
                        }
                    }
                }
            }
        }
        // (a byte per node, plain store: as bits in shared words this was an atomic OR per completing node -- 830k of them on the
        // level that completes most nodes, from workgroups on all XCDs into 1000 cache lines: 2.5x the level's time)
        if (!parked && all_done && sub == 0) fdone[v] = 1;
    }
    // parked long lists: 256 lanes per list, the words OR-ed through LDS
    __syncthreads();
    const int n_hub = s_nhub < MSBFS_HUB_SLOTS ? s_nhub : MSBFS_HUB_SLOTS;
    for (int h = 0; h < n_hub; ++h) {
        const int64_t v = s_hub[h];
        const int64_t r0 = rowptr[v], r1 = rowptr[v + 1];
        bool all_done = true;
        for (int64_t w0 = 0; w0 < rs; w0 += MSBFS_WCHUNK) {
            uint64_t have[MSBFS_WCHUNK], need[MSBFS_WCHUNK], acc[MSBFS_WCHUNK];
            uint64_t missing = 0;
#pragma unroll
            for (int q = 0; q < MSBFS_WCHUNK; ++q) {
                const int64_t w = w0 + q;
                uint64_t valid = 0;
                have[q] = w < rs ? in[v * rs + w] : 0;
                if (w < n_words) {
                    const int64_t left = n_sources - w * 64;
                    valid = left >= 64 ? ~0ull : ((1ull << left) - 1);
                }
                need[q] = valid & ~have[q];
                acc[q] = 0;
                missing |= need[q];
            }
            if (threadIdx.x < MSBFS_WCHUNK) s_acc[threadIdx.x] = 0;
            __syncthreads();
            if (missing != 0) {                              // uniform over the workgroup
                for (int64_t e = r0 + threadIdx.x; e < r1; e += blockDim.x) {
                    const int64_t u = col[e];
                    if (filter && !((fany[u >> 5] >> (u & 31)) & 1u)) continue;
                    if (rs == 4) {
                        const ulonglong2 f01 = *reinterpret_cast<const ulonglong2*>(&in[u * 4]);
                        const ulonglong2 f23 = *reinterpret_cast<const ulonglong2*>(&in[u * 4 + 2]);
                        acc[0] |= f01.x; acc[1] |= f01.y; acc[2] |= f23.x; acc[3] |= f23.y;
                    } else {
#pragma unroll
                        for (int q = 0; q < MSBFS_WCHUNK; ++q)
                            if (need[q]) acc[q] |= in[u * rs + w0 + q];
                    }
                }
#pragma unroll
                for (int q = 0; q < MSBFS_WCHUNK; ++q)
                    if (acc[q]) atomicOr(&s_acc[q], (unsigned long long)acc[q]);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MSBFS_WCHUNK; ++q) {          // (static register indices: thread q finishes word q)
                if ((int)threadIdx.x == q && w0 + q < rs) {
                    const uint64_t m = s_acc[q] & need[q];
                    out[v * rs + w0 + q] = have[q] | m;
                    if (m) {
                        found = true;
                        if (dist) {
                            uint64_t bits = m;
                            while (bits) {
                                const int b = __ffsll((unsigned long long)bits) - 1;
                                bits &= bits - 1;
                                const int64_t sidx = (w0 + q) * 64 + b;
                                if (sidx < n_sources) dist[sidx * ss + v * sv] = (uint8_t)level;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < MSBFS_WCHUNK; ++q)
                if (need[q] & ~s_acc[q]) all_done = false;   // (every thread reads the same words)
            __syncthreads();
        }
        if (all_done && threadIdx.x == 0) fdone[v] = 1;
    }
    if (__syncthreads_or(found ? 1 : 0) && threadIdx.x == 0) atomicOr(&flags[level], 1);
}

__global__ __launch_bounds__(256) void msbfs_commit_kernel(
    const int64_t* __restrict__ rowptr, int64_t n_ids, int64_t n_words, int64_t n_sources, uint64_t* __restrict__ seen,
    uint64_t* __restrict__ frontier, uint64_t* __restrict__ next, uint8_t* __restrict__ dist, int32_t* __restrict__ flags,
    unsigned long long* __restrict__ fvol, int level, int64_t ss, int64_t sv, uint32_t* __restrict__ fcur,
    uint8_t* __restrict__ fdone, int64_t rs, unsigned long long pull_above, int push_levels)
{
    if (flags[level - 1] == 0) return;
    {
        bool pull, prev_pull;
        int k;
        msbfs_mode(fvol, pull_above, push_levels, level, pull, prev_pull, k);
        if (pull) return;                                    // a pull level commits itself (version rotation): nothing to do
    }
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
        if (v < n_ids) fdone[v] = done ? 1 : 0;                 // (64 consecutive bytes per wavefront)
        if (lane == 0) {                                        // (this wavefront owns the two words: plain read-modify-write)
            uint32_t* __restrict__ fany = fcur + fwords;
            fcur[base >> 5] = (uint32_t)mask;
            if ((uint32_t)mask) fany[base >> 5] |= (uint32_t)mask;
            if ((base >> 5) + 1 < fwords) {
                fcur[(base >> 5) + 1] = (uint32_t)(mask >> 32);
                if ((uint32_t)(mask >> 32)) fany[(base >> 5) + 1] |= (uint32_t)(mask >> 32);
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
           8 + 2 * ((max_id + 32) / 32) * 4 + ((max_id + 4) / 4) * 4 + 8;   // + the frontier-node bitmaps (now | ever) + one "complete" byte per node
}

// status[0] = the last level that found anything, [1] = 1 if the LAST enqueued level still found something (too few levels
// enqueued), [2] = the first level that pulled among the levels that ran (0: none did) -- what a caller that repeats the
// search hands back as ``push_levels`` (+ a margin), [3] = 0
__device__ static inline void msbfs_write_status(const int32_t* __restrict__ flags, const unsigned long long* __restrict__ fvol,
                                                 unsigned long long pull_above, int push_levels, int last_level,
                                                 int32_t* __restrict__ status)
{
    int last = 0;
    for (int l = 1; l <= last_level; ++l) if (flags[l]) last = l;
    status[0] = last;
    status[1] = flags[last_level] != 0;
    int first_pull = 0;
    for (int l = 2; l <= last + 1 && l <= last_level && first_pull == 0; ++l) {
        bool pull, prev_pull;
        int k;
        msbfs_mode(fvol, pull_above, push_levels, l, pull, prev_pull, k);
        if (pull) first_pull = l;
    }
    status[2] = first_pull;
    status[3] = 0;
}

// The reference's matrix holds 0 for unreachable pairs, and its row-min runs over those zeros: a
// source that never reaches SOME member of a set gives 0 for the whole set (SubGNN.py:772).  After
// the last level: AND the members' seen words, zero the sources missing from it.
__global__ __launch_bounds__(256) void msbfs_set_finalize_kernel(
    const uint64_t* __restrict__ seen, int64_t n_words, int64_t n_sources,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets, float* __restrict__ out, int64_t rs,
    const uint64_t* __restrict__ frontier, uint64_t* __restrict__ set_seen, const int32_t* __restrict__ flags, int last_level,
    int32_t* __restrict__ status, const uint64_t* __restrict__ nextbuf, const unsigned long long* __restrict__ fvol,
    unsigned long long pull_above, int push_levels, const uint8_t* __restrict__ fdone)
{
    if (status && blockIdx.x == 0 && threadIdx.x == 0) {     // (msbfs_status_kernel's work: one launch less per search)
        msbfs_write_status(flags, fvol, pull_above, push_levels, last_level, status);
    }
    const int64_t total = n_sets * n_words;
    const bool reduce_last = flags[last_level] != 0;         // the last enqueued level found something: its reduction is still due
    bool pull, prev_pull;
    int k;
    msbfs_mode(fvol, pull_above, push_levels, last_level, pull, prev_pull, k);
    const uint64_t* B[3] = {seen, nextbuf, frontier};
    // pull levels that ran: the newest version of the seen rows is B[n_written % 3] (no pull: the commits kept B[0] current)
    int n_written = 0;
    for (int l = 2; l <= last_level && flags[l - 1] != 0; ++l) {
        bool pl, ppl;
        int kk;
        msbfs_mode(fvol, pull_above, push_levels, l, pl, ppl, kk);
        n_written += pl ? 1 : 0;
    }
    // the last level's new bits: its commit left them in the frontier rows (push), or they are the difference of the version
    // it wrote and the one it read (pull number k: B[(k + 1) % 3] from B[k % 3])
    const uint64_t* lcur = pull ? B[(k + 1) % 3] : frontier;
    const uint64_t* lold = pull ? B[k % 3] : nullptr;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        if (reduce_last)
            msbfs_set_reduce_item(lcur, n_words, n_sources, set_ptr, set_nodes, set_seen, out, last_level, rs, t, lold);
        const int64_t r = t / n_words, w = t % n_words;
        uint64_t all = ~0ull;
        // a member every source has reached (its "complete" byte: set by the commit or the pull level that saw it) restricts
        // nothing; any other member's row is exact in the version written last (complete rows are the only ones that go stale)
        for (int64_t i = set_ptr[r]; i < set_ptr[r + 1]; ++i) {
            const int64_t v = set_nodes[i];
            if (!fdone[v]) all &= B[n_written % 3][v * rs + w];
        }
        uint64_t missing = ~all;
        while (missing) {
            const int b = __ffsll((unsigned long long)missing) - 1;
            missing &= missing - 1;
            const int64_t s = w * 64 + b;
            if (s < n_sources) out[r * n_sources + s] = 0.f;
        }
    }
}

__global__ void msbfs_status_kernel(const int32_t* __restrict__ flags, int max_hops, int32_t* __restrict__ status,
                                    const unsigned long long* __restrict__ fvol, unsigned long long pull_above, int push_levels)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) msbfs_write_status(flags, fvol, pull_above, push_levels, max_hops, status);
}

static int msbfs_run(const int64_t* rowptr, const int32_t* col, int64_t nnz, int64_t max_id,
                     const int32_t* sources, int64_t n_sources, int max_hops, int node_major, uint8_t* dist,
                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets, float* set_out,
                     void* workspace, hipStream_t st, int pull_alpha, int32_t* status = nullptr, int push_levels = -1)
{
    // push_levels: levels that may still push (a commit launch is enqueued for each); beyond them the search pulls.  < 0, or
    // alpha = 0 (never pull: the caller's choice): every level
    if (push_levels < 0 || pull_alpha == 0 || push_levels > max_hops) push_levels = max_hops;
    if (push_levels < 1) push_levels = 1;                    // (level 1 always pushes: the seeds' lists)
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
    uint8_t* fdone = (uint8_t*)(fbits + 2 * fwords);             // n_ids bytes, rounded up to whole words
    uint64_t* set_seen = (uint64_t*)(((uintptr_t)(fbits + 2 * fwords + (n_ids + 3) / 4) + 7) & ~(uintptr_t)7);
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
    MsbfsBufs bufs;
    bufs.b[0] = seen; bufs.b[1] = next; bufs.b[2] = frontier;
    for (int level = 1; level <= max_hops; ++level) {
        hipLaunchKernelGGL(msbfs_level_kernel, dim3(g_expand), dim3(256), 0, st, rowptr, col, n_ids, n_words, n_sources,
                           bufs, flags, fvol, pull_above, push_levels, level, fbits, fbits + fwords, fdone, rs, sets, dist, ss, sv);
        SGNN_CHECK_LAUNCH();
        if (level <= push_levels) {                          // (a level beyond them pulls: it commits itself)
            hipLaunchKernelGGL(msbfs_commit_kernel, dim3(g_commit), dim3(256), 0, st, rowptr, n_ids, n_words, n_sources, seen,
                               frontier, next, dist, flags, fvol, level, ss, sv, fbits, fdone, rs, pull_above, push_levels);
            SGNN_CHECK_LAUNCH();
        }
    }
    if (set_out) {
        hipLaunchKernelGGL(msbfs_set_finalize_kernel, dim3(g_sets), dim3(256), 0, st, seen, n_words, n_sources, set_ptr,
                           set_nodes, n_sets, set_out, rs, frontier, set_seen, flags, max_hops, status, next, fvol, pull_above,
                           push_levels, fdone);
        SGNN_CHECK_LAUNCH();
    }
    if (status && !set_out) {
        hipLaunchKernelGGL(msbfs_status_kernel, dim3(1), dim3(64), 0, st, flags, max_hops, status, fvol, pull_above, push_levels);
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
                                         const int32_t* sources, int64_t n_sources, int max_hops, int pull_alpha, int push_levels,
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
                     workspace, (hipStream_t)stream, pull_alpha, out_status, push_levels);
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

SGNN_DEFINE_WARM(similarity)
