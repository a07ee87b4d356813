// Structure-channel CSR gather: internal / external degree sequences of node sets.
// Replaces gamma.get_degree_sequence (reference SubGNN/gamma.py:21-49).
//
// HBM-bound: per set S the algorithmic traffic is  sum_{v in S} (16 + 4 deg(v)) + 8 |S|  bytes
// (rowptr pair, neighbour list, id in, two degrees out).  Design:
//   * one 64-lane wavefront per set (|S| <= 64; the common case: CCs and walk patches);
//     lane i owns member i: loads its id and rowptr pair, inserts the id into a 1024-slot table
//     in LDS (the set's membership structure; collision-free under one of four multipliers, so a
//     lookup is a single slot read);
//   * members with >= 64 neighbours (which carry most of the bytes on scale-free graphs) are
//     streamed one list at a time with eight coalesced 256 B loads in flight and a wave
//     sum at the end; the neighbour lists of the other members are streamed as ONE flat range
//     of sum(deg) items: lane l locates its (member, offset) by a 6-step binary search over
//     the wave-resident inclusive degree scan (ds_bpermute), loads col[] (consecutive lanes hit
//     consecutive addresses inside a list -> coalesced 256 B runs even across list ends),
//     probes the LDS hash, and the hits are reduced per member with wavefront ballot +
//     popcount over each member's bit range -- no atomics, no divergence on list length;
//   * per-set ascending order by an in-register rank sort (n <= 64 compare rounds of readlane).
//   * sets of 65..2048 entries take a 256-thread workgroup variant (LDS scan, LDS integer
//     atomics, LDS bitonic sort).
#include "common.h"

#define DS_HASH_BITS 10
#define DS_HASH (1 << DS_HASH_BITS)
#define DS_TRIES 4

#ifndef DS_BIG
#define DS_BIG 64
#endif
#ifndef DS_FLAT_LOADS
#define DS_FLAT_LOADS 2         // 64-entry loads in flight per step of the flat range of the short lists (phase B)
#endif
#ifndef DS_SEARCH
#define DS_SEARCH 512           // lists at least this long are searched (sorted rows given) instead of streamed (256 / 512 / 1024 / 2048 with three lists per round: 0.215 / 0.210 / 0.213 / 0.229 ms; all streamed: 0.372)
#endif
#ifndef DS_WIDE
#define DS_WIDE 0               // > 0: lists of at least 256 x DS_WIDE entries are read 16 bytes per lane, DS_WIDE loads in flight
#endif
#ifndef DS_INFLIGHT
#define DS_INFLIGHT 32          // 256-byte loads a wavefront keeps in flight while it streams a long list (4 / 8 / 16 / 32: 0.414 / 0.396 / 0.388 / 0.372 ms)
#endif
//           // members with at least this many neighbours are streamed on their own

// Membership structure of a set: a 1024-slot table in LDS, one per wavefront.  The slot of an id
// is the top 10 bits of a 24 x 24-bit product (v_mul_u32_u24 issues at full rate; the 32-bit
// multiply at a quarter of it -- and this hash is evaluated once per streamed neighbour).  The
// insert phase tries up to four multipliers until the <= 64 members land in distinct slots
// (83 % per try for 20 members), so a lookup is ONE slot read with no probe loop; if every try
// collides the table falls back to linear probing with a wave-uniform probe count P (no
// data-dependent exit: divergent loops cost scalar instructions, and the CU's single scalar unit
// was the measured bottleneck of the first version of this kernel).
__device__ static const uint32_t ds_mult[DS_TRIES] = {0x9E3779u | 1u, 0x85EBCBu, 0xC2B2AFu, 0x27D4EBu};

__device__ __forceinline__ uint32_t ds_slot4(int32_t u, uint32_t k24) {
    // byte offset of the slot: bits [31:22] of the product, already scaled by 4
    return (__umul24((uint32_t)u & 0xFFFFFFu, k24) >> (32 - DS_HASH_BITS - 2)) & ((DS_HASH - 1) << 2);
}
template <bool P1>
__device__ __forceinline__ int ds_probe(const int32_t* hash, int32_t u, uint32_t k24, int P) {
    const uint32_t o = ds_slot4(u, k24);
    const char* base = reinterpret_cast<const char*>(hash);
    if (P1) return *reinterpret_cast<const int32_t*>(base + o) == u;
    int hit = 0;
    for (int p = 0; p < P; ++p) hit |= (*reinterpret_cast<const int32_t*>(base + ((o + 4 * p) & ((DS_HASH - 1) << 2))) == u);
    return hit;
}

__device__ static inline int32_t ds_wave_sum(int32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

#ifndef DS_WAVES
#define DS_WAVES 1             // wavefronts (= sets) per workgroup; 1 measured best (finest dispatch granularity)
#endif

// hits of every member's neighbour list in the set's table -> cnt (per lane = per member); SELF:
// count the self-loop entries of each list here (the caller has no per-node self-loop table)
// INFL: 256-byte loads in flight while a long list is streamed (DS_INFLIGHT; the SEARCH instantiations never stream a list of
// DS_SEARCH = 512 entries or more, so their full-block loop would be dead weight in the register budget: they take 8)
template <bool P1, bool SELF, int INFL = DS_INFLIGHT>
__device__ __forceinline__ void ds_count(const int32_t* __restrict__ col, const int32_t* __restrict__ col_sorted,
                                         const int32_t* hash, uint32_t k24, int P, int lane, int n, int32_t v,
                                         bool dup, int32_t deg, uint32_t r0, int32_t& cnt, int32_t& selfc,
                                         const uint32_t* __restrict__ hub_bits = nullptr, int64_t hub_words = 0, int32_t hidx = -1)
{
    uint64_t big = __ballot(deg >= DS_BIG);
    // ---- phase A0: very long lists, when the caller has the rows in ascending order: the set is looked up
    // IN the list instead of the list in the set.  Every member (one lane each) binary-searches its id in the
    // list -- log2(deg) dependent 4-byte loads for the whole set, where streaming a 30k-entry hub list costs
    // 470 wave loads and as many table probes.  (Simple graph: an id occurs at most once in a list.)
    if (col_sorted != nullptr || hub_bits != nullptr) {
        // (without the sorted rows only the lists that HAVE a bitmap are taken out of the streaming phases)
        uint64_t huge = __ballot(deg >= DS_SEARCH && (col_sorted != nullptr || hidx >= 0));
        big &= ~huge;
        // G = 64 / n lists are searched at a time: lane l works for list slot l / n as member l % n (a search
        // is a chain of dependent loads; a set of 20 members keeps three chains going in its 64 lanes)
        const int G = 64 / n;
        const int slot = lane / n, idx = lane - slot * n;
        const int32_t my_v = __shfl(v, idx);
        const bool my_dup = __shfl((int)dup, idx) != 0;
        while (huge) {
            int m_of_slot = -1;                                   // the list this lane's slot searches
            int taken = 0;
            uint64_t rest = huge;
            for (int gsl = 0; gsl < G && rest; ++gsl) {           // uniform: G and huge are
                const int m = __ffsll((unsigned long long)rest) - 1;
                rest &= rest - 1;
                if (slot == gsl) m_of_slot = m;
                ++taken;
            }
            huge = rest;
            const bool act = slot < taken;
            const int ms = act ? m_of_slot : 0;
            const int32_t m_deg = act ? __shfl(deg, ms) : 0;
            // A list whose membership BITMAP the caller has (round 6: sgnn_degree_sequence_hub_bitmaps; one bit per node id
            // for every list of >= DS_SEARCH entries, built once per graph) answers "is my_v in the list" with ONE load of one
            // word -- where the binary search is ~15 dependent loads, each of them a separate 64-byte request per lane (three
            // hub lists per 20-node set: 900 such requests per set, a third of the launch's time in the address coalescer).
            const int32_t m_hidx = __shfl(hidx, ms);
            const bool by_bits = act && hub_bits != nullptr && m_hidx >= 0;
            bool found_bits = false;
            // (the bitmaps are laid out BY NODE: row my_v holds one bit per hub list, so the lookups of one member against all the
            // hubs of its set fall into one line -- 20 lines per 20-node set instead of 60 with one row per hub: 0.72 -> see DESIGN GB)
            if (by_bits) found_bits = ((hub_bits[(int64_t)my_v * hub_words + ((uint32_t)m_hidx >> 5)] >> ((uint32_t)m_hidx & 31u)) & 1u) != 0u;
            const bool by_search = act && !by_bits;              // (col_sorted is given whenever a list without a bitmap is parked here)
            const int32_t* __restrict__ list = col_sorted + __shfl(r0, ms);
            int32_t lo = 0, hi = m_deg;                           // lower bound of my_v in list[0, m_deg)
            int steps = by_search ? 32 - __clz(m_deg) : 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(steps, d); steps = o > steps ? o : steps; }
            for (int it = 0; it < steps; ++it) {
                const int32_t mid = (lo + hi) >> 1;
                const int32_t x = (by_search && lo < hi) ? list[mid] : 0;
                const bool right = by_search && lo < hi && x < my_v;
                lo = right ? mid + 1 : lo;
                hi = (lo < hi && !right) ? mid : hi;
            }
            const bool found = by_bits ? found_bits : (by_search && lo < m_deg && list[lo < m_deg ? lo : 0] == my_v);
            const uint64_t fm = __ballot(found && !my_dup);       // an id listed twice in the set counts once
            const uint64_t fself = __ballot(found);
            // lane m of the set reads the count of the slot that searched ITS list (the first lane of every
            // slot knows which list the slot took)
            int my_slot = -1;
            for (int gsl = 0; gsl < taken; ++gsl) {
                const int mm = __shfl(m_of_slot, gsl * n);
                if (mm == lane) my_slot = gsl;
            }
            if (my_slot >= 0) {
                const uint64_t range = (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << (my_slot * n);
                cnt = (int32_t)__popcll(fm & range);
                if (SELF) selfc = (int32_t)((fself >> (my_slot * n + lane)) & 1ull);
            }
        }
    }
    // ---- phase A: members with >= 64 neighbours, one list at a time, coalesced 256 B loads in flight ----
    while (big) {
        const int m = __ffsll((unsigned long long)big) - 1;
        big &= big - 1;
        const int32_t m_v = __shfl(v, m), m_deg = __shfl(deg, m);
        const int32_t* __restrict__ list = col + __shfl(r0, m);
        // hits are counted per 64-entry chunk with ballot + scalar popcount: the running totals live in
        // scalar registers and no wave reduction is needed at the end of a list
        int32_t tot = 0, st = 0;
        int32_t base = 0;
#if DS_WIDE
        // very long lists: 16 bytes per lane (1 KB per wave instruction) from the first 16-byte boundary on;
        // the <= 3 entries before it and the remainder go through the 4-byte forms below
        if (m_deg >= 64 * 4 * DS_WIDE + 3) {
            const int32_t head = (int32_t)((4u - (__shfl(r0, m) & 3u)) & 3u);
            if (head) {
                const int32_t u0 = lane < head ? list[lane] : -1;
                tot += (int32_t)__popcll(__ballot(ds_probe<P1>(hash, u0, k24, P) != 0));
                if (SELF) st += (int32_t)__popcll(__ballot(u0 == m_v));
                base = head;
            }
            const int4* __restrict__ list4 = reinterpret_cast<const int4*>(list + base);
            const int32_t n4 = (m_deg - base) >> 2;
            int32_t i4 = 0;
            for (; i4 + 64 * DS_WIDE <= n4; i4 += 64 * DS_WIDE) {
                int4 w[DS_WIDE];
#pragma unroll
                for (int q = 0; q < DS_WIDE; ++q) w[q] = list4[i4 + q * 64 + lane];
#pragma unroll
                for (int q = 0; q < DS_WIDE; ++q) {
                    tot += (int32_t)__popcll(__ballot(ds_probe<P1>(hash, w[q].x, k24, P) != 0));
                    tot += (int32_t)__popcll(__ballot(ds_probe<P1>(hash, w[q].y, k24, P) != 0));
                    tot += (int32_t)__popcll(__ballot(ds_probe<P1>(hash, w[q].z, k24, P) != 0));
                    tot += (int32_t)__popcll(__ballot(ds_probe<P1>(hash, w[q].w, k24, P) != 0));
                    if (SELF) {
                        st += (int32_t)__popcll(__ballot(w[q].x == m_v)) + (int32_t)__popcll(__ballot(w[q].y == m_v));
                        st += (int32_t)__popcll(__ballot(w[q].z == m_v)) + (int32_t)__popcll(__ballot(w[q].w == m_v));
                    }
                }
            }
            base += i4 * 4;
        }
#endif
        for (; base + 64 * INFL <= m_deg; base += 64 * INFL) {   // full blocks: INFL x 256 B loads in flight
            int32_t u[INFL];
#pragma unroll
            for (int q = 0; q < INFL; ++q) u[q] = list[base + q * 64 + lane];
#pragma unroll
            for (int q = 0; q < INFL; ++q) {
                tot += (int32_t)__popcll(__ballot(ds_probe<P1>(hash, u[q], k24, P) != 0));
                if (SELF) st += (int32_t)__popcll(__ballot(u[q] == m_v));
            }
        }
        for (; base < m_deg; base += 512) {              // tail (< 64 x DS_INFLIGHT entries) in blocks of 8 clamped loads:
            const int32_t last = m_deg - 1;              // out-of-range lanes get the never-stored key -1
            int32_t u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int32_t i = base + q * 64 + lane;
                u[q] = list[i < last ? i : last];
                u[q] = i <= last ? u[q] : -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (base + q * 64 < m_deg) {                                                   // wave-uniform
                    tot += (int32_t)__popcll(__ballot(ds_probe<P1>(hash, u[q], k24, P) != 0));
                    if (SELF) st += (int32_t)__popcll(__ballot(u[q] == m_v));
                }
        }
        if (lane == m) { cnt = tot; if (SELF) selfc = st; }
    }
    // ---- phase B: the remaining lists as one flat range, 128 entries per step ------------
    const int32_t sdeg = deg >= DS_BIG ? 0 : deg;
    int32_t incl = sdeg;                                 // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int32_t total = __shfl(incl, 63);
    const int32_t excl = incl - sdeg;
    for (int32_t base = 0; base < total; base += 64 * DS_FLAT_LOADS) {
        int32_t u[DS_FLAT_LOADS], mv[DS_FLAT_LOADS];
#pragma unroll
        for (int q = 0; q < DS_FLAT_LOADS; ++q) {
            const int32_t t = base + q * 64 + lane;
            int lo = 0, hi = 63;                         // smallest m with incl[m] > t
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi) >> 1;
                const int32_t x = __shfl(incl, mid);
                if (x > t) hi = mid; else lo = mid + 1;
            }
            const int m = lo & 63;
            const int32_t m_excl = __shfl(excl, m);
            const uint32_t m_r0 = __shfl(r0, m);
            mv[q] = __shfl(v, m);
            u[q] = t < total ? col[m_r0 + (uint32_t)(t - m_excl)] : -1;
        }
#pragma unroll
        for (int q = 0; q < DS_FLAT_LOADS; ++q) {
            if (base + q * 64 >= total) break;                                   // (wave-uniform)
            const uint64_t mh = __ballot(ds_probe<P1>(hash, u[q], k24, P) != 0);
            const int32_t b0 = base + q * 64;
            int32_t lo_i = excl - b0, hi_i = incl - b0;  // this lane's member range in the step
            lo_i = lo_i < 0 ? 0 : (lo_i > 64 ? 64 : lo_i);
            hi_i = hi_i < 0 ? 0 : (hi_i > 64 ? 64 : hi_i);
            const uint64_t below_hi = hi_i >= 64 ? ~0ull : ((1ull << hi_i) - 1ull);
            const uint64_t below_lo = lo_i >= 64 ? ~0ull : ((1ull << lo_i) - 1ull);
            const uint64_t rm = below_hi & ~below_lo;
            cnt += __popcll(mh & rm);
            if (SELF) selfc += __popcll(__ballot(u[q] == mv[q]) & rm);
        }
    }
}

// FEW: instantiation taken by launches over a handful of sets (the anchor patches: a few hundred);
// same code -- it only keeps those microsecond launches apart from the shard-sized ones in profiles.
#ifndef DS_GRID_CAP
#define DS_GRID_CAP (1 << 20)    // workgroups of the wave kernel: one set each up to this many, grid-stride beyond (50k / 25k / 16k / 8k / 5k workgroups for 50k sets: 0.395 / 0.406 / 0.423 / 0.467 / 0.537 ms -- finer is better)
#endif
#ifndef DS_MIN_WAVES
#define DS_MIN_WAVES 5         // wavefronts per SIMD the register allocation aims for (86 VGPRs -> 5); 4 / 5 / 6 / 7 / 8 measured 0.392 / 0.391 / 0.396 / 0.398 / 0.401 ms: not occupancy-bound
#endif
#ifndef DS_MIN_WAVES_SEARCH
#define DS_MIN_WAVES_SEARCH 5  // the SEARCH instantiations: 85 VGPRs with 8 loads in flight (INFL below).  With DS_INFLIGHT = 32 in flight they
#endif                         // needed 102: under this 5-wave budget (96) round 4's build spilled 4 registers to scratch -- 20 B per lane, 59 MB
                               // written per launch for 8 MB of output -- and un-spilled at 4 waves the launch was SLOWER (0.243 ms back to back
                               // against 0.212 spilled: it is latency-bound, occupancy matters more than 5 scratch dwords)
// SEARCH: instantiation given the row-sorted CSR (long lists searched) -- the same code with col_sorted == nullptr
// would do, but profiles should tell the two forms of the launch apart.
template <bool SORTED, bool FEW = false, bool SEARCH = false>
__global__ __launch_bounds__(64 * DS_WAVES) __attribute__((amdgpu_waves_per_eu(SEARCH ? DS_MIN_WAVES_SEARCH : DS_MIN_WAVES))) void degseq_wave_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ full_degree, const uint8_t* __restrict__ self_loops,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    int32_t* __restrict__ out_int, int32_t* __restrict__ out_ext, const int32_t* __restrict__ set_order,
    const int32_t* __restrict__ col_sorted, const int32_t* __restrict__ hub_index, const uint32_t* __restrict__ hub_bits,
    int64_t hub_words, const int4* __restrict__ node_info, int info_degree)
{
    // one hash table per wavefront; a wavefront's LDS operations execute in issue order, so the
    // waves of a workgroup never need a workgroup barrier (they work on different sets)
    __shared__ int32_t hash_all[DS_WAVES][DS_HASH];
    constexpr int INFL = SEARCH ? 8 : DS_INFLIGHT;
    const int lane = threadIdx.x & 63;
    int32_t* hash = hash_all[threadIdx.x >> 6];
    for (int64_t si = (int64_t)blockIdx.x * DS_WAVES + (threadIdx.x >> 6); si < n_sets; si += (int64_t)gridDim.x * DS_WAVES) {
        // set_order: the caller's dispatch order (heaviest sets first keeps the tail of the launch short)
        const int64_t s = set_order ? set_order[si] : si;
        const int64_t beg = set_ptr[s];
        const int n = (int)(set_ptr[s + 1] - beg);
        if (n <= 0) continue;                               // wave-uniform
        if (n > 64) {
            // the caller promised sets of at most 64 entries (max_set_size) and this one is larger: its outputs are
            // poisoned, never left as they were -- a wrong bound shows up as INT32_MIN, not as stale memory
            for (int i = lane; i < n; i += 64) {
                out_int[beg + i] = INT32_MIN;
                if (out_ext) out_ext[beg + i] = INT32_MIN;
            }
            continue;
        }
        int32_t v = 0, deg = 0, hidx = -1, info_self = 0, info_full = 0;
        uint32_t r0 = 0;
        const bool have_info = SEARCH && node_info != nullptr;           // (wave-uniform)
        if (lane < n) {
            v = set_nodes[beg + lane];
            if (have_info) {
                // ONE 16-byte record per member (round 6: row start, degree, hub number | self loops << 24, full degree) instead of
                // a line each out of rowptr, hub_index, self_loops and full_degree: 4 random requests per member were 80 per set
                const int4 ni = node_info[v];
                r0 = (uint32_t)ni.x;
                deg = ni.y;
                hidx = ((uint32_t)ni.z & 0xffffffu) == 0xffffffu ? -1 : (int32_t)((uint32_t)ni.z & 0xffffffu);
                info_self = (int32_t)((uint32_t)ni.z >> 24);
                info_full = ni.w;
            } else {
                const int64_t a = rowptr[v], b = rowptr[v + 1];
                if (SEARCH && hub_index != nullptr) hidx = hub_index[v]; // (>= 0: the list has a membership bitmap)
                r0 = (uint32_t)a;
                deg = (int32_t)(b - a);
            }
        }
        // ---- build the table: first multiplier under which no two members share a slot --------
        uint32_t k24 = ds_mult[0];
        int P = 1;
        bool dup = false;
        for (int t = 0; t < DS_TRIES; ++t) {
            k24 = ds_mult[t];
#pragma unroll
            for (int q = 0; q < DS_HASH / 256; ++q)          // 16 B per lane per store
                reinterpret_cast<int4*>(hash)[lane + 64 * q] = make_int4(0, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int chain = 0;
            dup = false;
            if (lane < n) {
                uint32_t h = ds_slot4(v, k24) >> 2;
                const bool last_try = (t == DS_TRIES - 1);
                while (true) {
                    ++chain;
                    const int32_t old = atomicCAS(&hash[h], 0, v);
                    if (old == 0 || old == v) { dup = (old == v); break; }    // dup: another lane holds the same id
                    if (!last_try) { chain = 2; break; }     // collision: this multiplier is rejected
                    h = (h + 1) & (DS_HASH - 1);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(chain, d); chain = o > chain ? o : chain; }
            P = chain;                                       // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (P <= 1) break;
        }
        int32_t cnt = 0, selfc = 0;
        if (self_loops != nullptr || have_info) {
            if (lane < n) selfc = have_info ? info_self : self_loops[v];
            if (P <= 1) ds_count<true, false, INFL>(col, SEARCH ? col_sorted : nullptr, hash, k24, P, lane, n, v, dup, deg, r0, cnt, selfc,
                                                   SEARCH ? hub_bits : nullptr, hub_words, hidx);
            else ds_count<false, false, INFL>(col, SEARCH ? col_sorted : nullptr, hash, k24, P, lane, n, v, dup, deg, r0, cnt, selfc,
                                                   SEARCH ? hub_bits : nullptr, hub_words, hidx);
        } else {
            if (P <= 1) ds_count<true, true, INFL>(col, SEARCH ? col_sorted : nullptr, hash, k24, P, lane, n, v, dup, deg, r0, cnt, selfc,
                                                   SEARCH ? hub_bits : nullptr, hub_words, hidx);
            else ds_count<false, true, INFL>(col, SEARCH ? col_sorted : nullptr, hash, k24, P, lane, n, v, dup, deg, r0, cnt, selfc,
                                                   SEARCH ? hub_bits : nullptr, hub_words, hidx);
        }
        cnt += selfc;                                        // a self loop counts twice (networkx)
        int32_t full = deg + selfc;
        if (have_info) { if (info_degree) full = info_full; }
        else if (full_degree != nullptr && lane < n) full = full_degree[v];
        const int32_t internal = cnt;
        const int32_t external = full - cnt;
        if (!SORTED) {
            if (lane < n) {
                out_int[beg + lane] = internal;
                if (out_ext) out_ext[beg + lane] = external;
            }
        } else {
            int ri = 0, re = 0;                              // stable rank among the n entries
            for (int j = 0; j < n; ++j) {
                const int32_t xj = __shfl(internal, j);
                const int32_t ej = __shfl(external, j);
                ri += (xj < internal) || (xj == internal && j < lane);
                re += (ej < external) || (ej == external && j < lane);
            }
            if (lane < n) {
                out_int[beg + ri] = internal;
                if (out_ext) out_ext[beg + re] = external;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- workgroup variant for 64 < |S| <= DSB_MAX ------------------------------------------------
#define DSB_MAX 2048
#define DSB_HASH_BITS 12
#define DSB_HASH (1 << DSB_HASH_BITS)
#define DSB_THREADS 256

__device__ static inline void dsb_bitonic_sort(int32_t* a, int n_pow2, int tid) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += DSB_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int32_t x = a[i], y = a[ixj];
                    const bool up = ((i & k) == 0);
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

template <bool SORTED>
__global__ __launch_bounds__(DSB_THREADS) void degseq_block_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ full_degree,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    int32_t* __restrict__ out_int, int32_t* __restrict__ out_ext)
{
    __shared__ int32_t hash[DSB_HASH];
    __shared__ int32_t s_v[DSB_MAX];
    __shared__ uint32_t s_r0[DSB_MAX];
    __shared__ int32_t s_incl[DSB_MAX];
    __shared__ int32_t s_cnt[DSB_MAX];
    __shared__ int32_t s_self[DSB_MAX];
    __shared__ int32_t s_part[DSB_THREADS];
    const int tid = threadIdx.x;
    for (int64_t s = blockIdx.x; s < n_sets; s += gridDim.x) {
        const int64_t beg = set_ptr[s];
        const int n = (int)(set_ptr[s + 1] - beg);
        if (n <= 64 || n > DSB_MAX) continue;               // block-uniform
        for (int i = tid; i < DSB_HASH; i += DSB_THREADS) hash[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += DSB_THREADS) {
            const int32_t v = set_nodes[beg + i];
            const int64_t a = rowptr[v], b = rowptr[v + 1];
            s_v[i] = v;
            s_r0[i] = (uint32_t)a;
            s_incl[i] = (int32_t)(b - a);
            s_cnt[i] = 0;
            s_self[i] = 0;
            uint32_t h = sgnn_hash32((uint32_t)v) >> (32 - DSB_HASH_BITS);
            while (true) {
                const int32_t old = atomicCAS(&hash[h], 0, v);
                if (old == 0 || old == v) break;
                h = (h + 1) & (DSB_HASH - 1);
            }
        }
        __syncthreads();
        // inclusive scan of s_incl[0..n): per-thread chunks of 8, then a scan of the 256 partials
        const int per = (n + DSB_THREADS - 1) / DSB_THREADS;
        const int c0 = tid * per, c1 = (c0 + per < n) ? c0 + per : n;
        int32_t acc = 0;
        for (int i = c0; i < c1; ++i) { acc += s_incl[i]; s_incl[i] = acc; }
        s_part[tid] = acc;
        __syncthreads();
        for (int d = 1; d < DSB_THREADS; d <<= 1) {
            int32_t t = 0;
            if (tid >= d) t = s_part[tid - d];
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        const int32_t offset = (tid == 0) ? 0 : s_part[tid - 1];
        for (int i = c0; i < c1; ++i) s_incl[i] += offset;
        __syncthreads();
        const int32_t total = s_incl[n - 1];
        for (int32_t t = tid; t < total; t += DSB_THREADS) {
            int lo = 0, hi = n - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_incl[mid] > t) hi = mid; else lo = mid + 1;
            }
            const int m = lo;
            const int32_t m_excl = (m == 0) ? 0 : s_incl[m - 1];
            const int32_t u = col[s_r0[m] + (uint32_t)(t - m_excl)];
            uint32_t h = sgnn_hash32((uint32_t)u) >> (32 - DSB_HASH_BITS);
            bool hit = false;
            while (true) {
                const int32_t k = hash[h];
                if (k == u) { hit = true; break; }
                if (k == 0) break;
                h = (h + 1) & (DSB_HASH - 1);
            }
            if (u == s_v[m]) { atomicAdd(&s_cnt[m], 2); atomicAdd(&s_self[m], 1); }
            else if (hit) atomicAdd(&s_cnt[m], 1);
        }
        __syncthreads();
        // s_cnt = internal; reuse s_self for external
        for (int i = tid; i < n; i += DSB_THREADS) {
            const int32_t deg = s_incl[i] - (i == 0 ? 0 : s_incl[i - 1]);
            const int32_t full = full_degree ? full_degree[s_v[i]] : deg + s_self[i];
            s_r0[i] = (uint32_t)(full - s_cnt[i]);
        }
        __syncthreads();
        int32_t* s_ext = reinterpret_cast<int32_t*>(s_r0);
        if (SORTED) {
            int np2 = 128;
            while (np2 < n) np2 <<= 1;
            for (int i = n + tid; i < np2; i += DSB_THREADS) { s_cnt[i] = INT32_MAX; s_ext[i] = INT32_MAX; }
            __syncthreads();
            dsb_bitonic_sort(s_cnt, np2, tid);
            if (out_ext) dsb_bitonic_sort(s_ext, np2, tid);
        }
        for (int i = tid; i < n; i += DSB_THREADS) {
            out_int[beg + i] = s_cnt[i];
            if (out_ext) out_ext[beg + i] = s_ext[i];
        }
        __syncthreads();
    }
}

static int ds_run(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted, int64_t nnz,
                  const int32_t* full_degree, const uint8_t* self_loops,
                  const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                  int64_t max_set_size, int sorted,
                  int32_t* out_internal, int32_t* out_external, const int32_t* set_order,
                  void* stream, const int32_t* hub_index = nullptr, const uint32_t* hub_bits = nullptr, int64_t hub_words = 0,
                  const int32_t* node_info = nullptr, int info_degree = 0)
{
    if (!rowptr || !col || !set_ptr || !set_nodes || !out_internal || n_sets < 0 || max_set_size <= 0)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (n_sets == 0) return SGNN_OK;                            // (sets of more than DSB_MAX entries: sgnn_degree_sequence_huge)
    hipStream_t st = (hipStream_t)stream;
    // 256-thread workgroups = 4 independent wavefronts, one set per wavefront: the hardware
    // dispatcher hands out workgroups as CUs free up, which balances the very uneven per-set
    // cost (sum of member degrees) better than a static grid-stride assignment
    const int64_t want = (n_sets + DS_WAVES - 1) / DS_WAVES;
    const int grid = (int)(want < DS_GRID_CAP ? want : DS_GRID_CAP);
    const bool few = n_sets <= 4096;
#define DS_LAUNCH2(S, F, X) hipLaunchKernelGGL((degseq_wave_kernel<S, F, X>), dim3(grid), dim3(64 * DS_WAVES), 0, st, rowptr, col, \
                                           full_degree, self_loops, set_ptr, set_nodes, n_sets, out_internal, out_external, set_order, col_sorted, \
                                           hub_index, hub_bits, hub_words, reinterpret_cast<const int4*>(node_info), info_degree)
#define DS_LAUNCH(S, F) do { if (col_sorted || hub_bits) DS_LAUNCH2(S, F, true); else DS_LAUNCH2(S, F, false); } while (0)
    if (sorted) { if (few) DS_LAUNCH(true, true); else DS_LAUNCH(true, false); }
    else { if (few) DS_LAUNCH(false, true); else DS_LAUNCH(false, false); }
#undef DS_LAUNCH
#undef DS_LAUNCH2
    SGNN_CHECK_LAUNCH();
    if (max_set_size > 64) {
        const int gridb = (int)(n_sets < 256 * 4 ? n_sets : 256 * 4);
        if (sorted)
            hipLaunchKernelGGL(degseq_block_kernel<true>, dim3(gridb), dim3(DSB_THREADS), 0, st, rowptr, col,
                               full_degree, set_ptr, set_nodes, n_sets, out_internal, out_external);
        else
            hipLaunchKernelGGL(degseq_block_kernel<false>, dim3(gridb), dim3(DSB_THREADS), 0, st, rowptr, col,
                               full_degree, set_ptr, set_nodes, n_sets, out_internal, out_external);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

extern "C" int sgnn_degree_sequence(const int64_t* rowptr, const int32_t* col, int64_t nnz,
                                    const int32_t* full_degree, const uint8_t* self_loops,
                                    const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                    int64_t max_set_size, int sorted,
                                    int32_t* out_internal, int32_t* out_external, const int32_t* set_order,
                                    void* stream)
{
    return ds_run(rowptr, col, nullptr, nnz, full_degree, self_loops, set_ptr, set_nodes, n_sets, max_set_size, sorted,
                  out_internal, out_external, set_order, stream);
}

extern "C" int sgnn_degree_sequence_sorted_rows(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted,
                                                int64_t nnz, const int32_t* full_degree, const uint8_t* self_loops,
                                                const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                                int64_t max_set_size, int sorted, int32_t* out_internal,
                                                int32_t* out_external, const int32_t* set_order, void* stream)
{
    return ds_run(rowptr, col, col_sorted, nnz, full_degree, self_loops, set_ptr, set_nodes, n_sets, max_set_size, sorted,
                  out_internal, out_external, set_order, stream);
}

// The same with membership bitmaps for the long lists (round 6): hub_index[v] >= 0 numbers the lists that have one, -1 = no
// bitmap; hub_bits holds one row of hub_words 32-bit words PER NODE ID x, bit hub_index[v] of row x = "x is in v's list"
// (by node, so that one member's lookups against all the hubs of its set share a line); node_info (may be NULL): one 16-byte
// record per node id -- {row start, degree, hub number (0xffffff: none) | self-loop entries << 24, full degree} -- read INSTEAD
// of rowptr / hub_index / self_loops / full_degree (info_degree: take the record's full degree, else degree + self loops); a list of at least
// sgnn_degree_sequence_search_threshold() entries with a bitmap is neither streamed nor searched -- every member of the set reads
// its one bit.  Lists of that length WITHOUT a bitmap are searched when col_sorted is given, streamed otherwise.  Same results.
extern "C" int sgnn_degree_sequence_hub_bitmaps(const int64_t* rowptr, const int32_t* col, const int32_t* col_sorted,
                                                int64_t nnz, const int32_t* full_degree, const uint8_t* self_loops,
                                                const int32_t* hub_index, const uint32_t* hub_bits, int64_t hub_words,
                                                const int32_t* node_info, int info_degree,
                                                const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                                int64_t max_set_size, int sorted, int32_t* out_internal,
                                                int32_t* out_external, const int32_t* set_order, void* stream)
{
    if (!hub_index || !hub_bits || hub_words <= 0) return SGNN_ERR_BAD_ARG;
    if (node_info && (((uintptr_t)node_info) & 15)) return SGNN_ERR_BAD_ARG;
    return ds_run(rowptr, col, col_sorted, nnz, full_degree, self_loops, set_ptr, set_nodes, n_sets, max_set_size, sorted,
                  out_internal, out_external, set_order, stream, hub_index, hub_bits, hub_words, node_info, info_degree);
}

extern "C" int64_t sgnn_degree_sequence_search_threshold(void) { return DS_SEARCH; }

// Sets of more than DSB_MAX entries (components of subgraphs with thousands of nodes; rounds 1-2 refused them): the calls
// above leave them alone and this one fills in their degrees UNSORTED -- the membership table lives in the caller's
// workspace (4 int32 slots per entry at the set's own offset), a member's list is streamed by one wavefront, hits counted by
// ballot.  Same counting rules as above (a self loop counts twice; a repeated member gets its own count).  Sorting such a
// set is the caller's (any segment sort).  workspace: sgnn_degree_sequence_huge_workspace_bytes(set_ptr[n_sets]).
extern "C" int64_t sgnn_degree_sequence_huge_workspace_bytes(int64_t total_entries)
{
    return (total_entries < 0 ? 0 : total_entries) * 4 * 4 + 64;
}

__global__ __launch_bounds__(256) void degseq_huge_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ full_degree,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    int32_t* __restrict__ out_int, int32_t* __restrict__ out_ext, int32_t* __restrict__ ws)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int64_t s = blockIdx.x; s < n_sets; s += gridDim.x) {
        const int64_t beg = set_ptr[s];
        const int n = (int)(set_ptr[s + 1] - beg);
        if (n <= DSB_MAX) continue;
        int32_t* hash = ws + 4 * beg;
        uint32_t H = 1;
        while (H < 2u * (uint32_t)n) H <<= 1;
        for (uint32_t i = tid; i < H; i += 256) hash[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const int32_t v = set_nodes[beg + i];
            uint32_t h = sgnn_hash32((uint32_t)v) & (H - 1);
            while (true) {
                const int32_t old = atomicCAS(&hash[h], 0, v);
                if (old == 0 || old == v) break;
                h = (h + 1) & (H - 1);
            }
        }
        __threadfence_block();
        __syncthreads();
        for (int i = wave; i < n; i += 4) {
            const int32_t v = set_nodes[beg + i];
            const int64_t a = rowptr[v], b = rowptr[v + 1];
            int hits = 0, self = 0;
            for (int64_t e = a + lane; e < b + lane; e += 64) {          // uniform trip count per wavefront
                bool hit = false, loop = false;
                if (e < b) {
                    const int32_t u = col[e];
                    loop = (u == v);
                    if (!loop) {
                        uint32_t h = sgnn_hash32((uint32_t)u) & (H - 1);
                        while (true) {
                            const int32_t k = hash[h];
                            if (k == u) { hit = true; break; }
                            if (k == 0) break;
                            h = (h + 1) & (H - 1);
                        }
                    }
                }
                hits += __popcll(__ballot(hit));
                self += __popcll(__ballot(loop));
            }
            if (lane == 0) {
                const int32_t internal = hits + 2 * self;
                const int32_t full = full_degree ? full_degree[v] : (int32_t)(b - a) + self;
                out_int[beg + i] = internal;
                if (out_ext) out_ext[beg + i] = full - internal;
            }
        }
        __syncthreads();
    }
}

extern "C" int sgnn_degree_sequence_huge(const int64_t* rowptr, const int32_t* col, int64_t nnz, const int32_t* full_degree,
                                         const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                         int64_t total_entries, int32_t* out_internal, int32_t* out_external,
                                         void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!rowptr || !col || !set_ptr || !set_nodes || !out_internal || !workspace || n_sets < 0 || total_entries < 0)
        return SGNN_ERR_BAD_ARG;
    if (nnz >= (1ll << 31)) return SGNN_ERR_NNZ_TOO_LARGE;
    if (total_entries >= (1ll << 28)) return SGNN_ERR_SET_TOO_LARGE;
    if (workspace_bytes < sgnn_degree_sequence_huge_workspace_bytes(total_entries)) return SGNN_ERR_BAD_ARG;
    if (n_sets == 0) return SGNN_OK;
    hipLaunchKernelGGL(degseq_huge_kernel, dim3((int)(n_sets < 1024 ? n_sets : 1024)), dim3(256), 0, (hipStream_t)stream, rowptr,
                       col, full_degree, set_ptr, set_nodes, n_sets, out_internal, out_external, (int32_t*)workspace);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}


SGNN_DEFINE_WARM(degree_sequence)
