// Shared device/host helpers for libsubgnn_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/subgnn_hip.h"

#define SGNN_WAVE 64

void sgnn_set_last_error(hipError_t e);

#define SGNN_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t _e = hipGetLastError();                   \
        if (_e != hipSuccess) {                              \
            sgnn_set_last_error(_e);                         \
            return SGNN_ERR_LAUNCH;                          \
        }                                                    \
    } while (0)

static inline int sgnn_grid_for(int64_t work_items, int items_per_block, int max_blocks = 256 * 16) {
    int64_t b = (work_items + items_per_block - 1) / items_per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// ---- draw tape (twin of oracle/tape.py and subgnn_amd/tape.py) --------------------------
#define SGNN_K_STREAM 0x9E3779B97F4A7C15ull
#define SGNN_K_ITEM   0xD1B54A32D192ED03ull
#define SGNN_K_DRAW   0x8CB92BA72F3D8DD7ull

__host__ __device__ static inline uint64_t sgnn_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// h0 depends on (seed, stream) only: hoist it out of loops
__host__ __device__ static inline uint64_t sgnn_tape_h0(uint64_t seed, uint64_t stream) {
    return sgnn_mix64(seed ^ (stream * SGNN_K_STREAM));
}
__host__ __device__ static inline uint64_t sgnn_tape_h1(uint64_t h0, uint64_t item) {
    return sgnn_mix64(h0 + item * SGNN_K_ITEM);
}
__host__ __device__ static inline uint64_t sgnn_tape_draw(uint64_t h1, uint64_t j) {
    return sgnn_mix64(h1 + j * SGNN_K_DRAW);
}
__host__ __device__ static inline uint32_t sgnn_choice_index(uint64_t h1, uint64_t j, uint32_t n) {
    uint32_t u = (uint32_t)(sgnn_tape_draw(h1, j) >> 32);
    return (uint32_t)(((uint64_t)u * (uint64_t)n) >> 32);
}
__host__ __device__ static inline double sgnn_uniform01(uint64_t h1, uint64_t j) {
    uint32_t u = (uint32_t)(sgnn_tape_draw(h1, j) >> 32);
    return (double)u * (1.0 / 4294967296.0);
}
// 32-bit avalanche hash (two multiplies; 64-bit multiplies run at a fraction of this rate)
__host__ __device__ static inline uint32_t sgnn_lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x21F0AAADu; x ^= x >> 15; x *= 0x735A2D97u; x ^= x >> 15;
    return x;
}
// signed 53-bit key; the reference-side 'randn' value is key * 2^-52.  h1 = per-(row, slot) state
// from the 64-bit tape chain, j = node id: hi word = lowbias32(j ^ lo32(h1)) (signed), low 21 bits
// from lowbias32(j ^ hi32(h1)).
__host__ __device__ static inline int64_t sgnn_symmetric_key(uint64_t h1, uint64_t j) {
    const uint32_t hi = sgnn_lowbias32((uint32_t)j ^ (uint32_t)h1);
    const uint32_t lo = sgnn_lowbias32((uint32_t)j ^ (uint32_t)(h1 >> 32));
    return (int64_t)(int32_t)hi * (int64_t)(1 << 21) + (int64_t)(lo >> 11);
}

// The key orders lexicographically by (hi word signed, low 21 bits), so a running argmax only needs
// the second hash when the first one ties or beats the current best -- which happens O(log n) times
// per lane over n entries.  (id 0 = PAD-as-member holds key 0.)
__host__ __device__ static inline int32_t sgnn_key_hi(uint64_t h1, uint32_t v) {
    return v == 0 ? 0 : (int32_t)sgnn_lowbias32(v ^ (uint32_t)h1);
}
__host__ __device__ static inline uint32_t sgnn_key_lo(uint64_t h1, uint32_t v) {
    return v == 0 ? 0u : (sgnn_lowbias32(v ^ (uint32_t)(h1 >> 32)) >> 11);
}
__host__ __device__ static inline int64_t sgnn_key_join(int32_t hi, uint32_t lo) {
    return (int64_t)hi * (int64_t)(1 << 21) + (int64_t)lo;
}
// running argmax update for one candidate (col c, id v); best_lo is valid whenever best_hi came from
// a real candidate; ties on the full key keep the earlier column
#define SGNN_KEY_UPDATE(h1, v, c, best_hi, best_lo, bcol, bid)                                   \
    do {                                                                                         \
        const int32_t _hi = sgnn_key_hi((h1), (uint32_t)(v));                                    \
        const bool _cand = _hi >= (best_hi);                                                     \
        if (__ballot(_cand)) {              /* wave-uniform branch: no exec-mask bookkeeping */  \
            const uint32_t _lo = sgnn_key_lo((h1), (uint32_t)(v));                               \
            const bool _take = _cand && (_hi > (best_hi) || _lo > (best_lo) || (bcol) == INT32_MAX); \
            (best_hi) = _take ? _hi : (best_hi);                                                 \
            (best_lo) = _take ? _lo : (best_lo);                                                 \
            (bcol) = _take ? (int32_t)(c) : (bcol);                                              \
            (bid) = _take ? (int32_t)(v) : (bid);                                                \
        }                                                                                        \
    } while (0)

// ---- small device helpers ---------------------------------------------------------------
__device__ static inline uint32_t sgnn_hash32(uint32_t x) { return x * 2654435761u; }

// binary search: is `key` in the ascending array a[0..n)?
__device__ static inline bool sgnn_sorted_contains(const int32_t* __restrict__ a, int32_t n, int32_t key) {
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        int32_t v = a[mid];
        if (v < key) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == key;
}

// wave-wide argmax of (key, smallest column on ties); every lane ends with the winner
__device__ static inline void sgnn_argmax_reduce(int64_t& key, int32_t& colv, int32_t& idv) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int64_t k2 = __shfl_xor(key, d);
        const int32_t c2 = __shfl_xor(colv, d);
        const int32_t i2 = __shfl_xor(idv, d);
        if (k2 > key || (k2 == key && c2 < colv)) { key = k2; colv = c2; idv = i2; }
    }
}
