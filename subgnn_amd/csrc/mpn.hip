// One anchor -> component message-passing layer: gather, weight by similarity, aggregate per
// component, position/structure read-out (a13 + a15), forward and backward.
// Replaces SG_MPN.forward/propagate/message/generate_pos_struc_embeddings
// (reference SubGNN/subgraph_mpn.py:105-174,227-231) fused with the anchor-embedding gather of
// get_anchor_patches/embed_anchor_patch (reference SubGNN/anchor_patch_samplers.py:333-411).
//
// HBM-bound gather-scale-reduce.  Per real component with A anchors (fp32, width D):
//   GATHER  A*(4D + 8 + 4) + 4D + 4A  bytes  (anchor row, id, similarity; agg and read-out out)
//   SHARED  4A + 4D + 4A per component, + 4AD once (the shared anchor matrix stays in L2)
//   DENSE   A*(4D + 1 + 4) + 4D + 4A         (the reference's materialised (B,C,A,D) tensor)
// Thread = (component row, 16-byte column slice).  The D/4 lanes of a row read one anchor row as
// consecutive float4 (one 256 B segment for D = 64), accumulate w*x in registers, and reduce the
// read-out dot product <wp, x> across the row's lanes by butterfly shuffles.  No atomics in the
// forward pass; the backward pass scatters anchor-row gradients with float atomics whose
// wave-instruction footprint is whole contiguous rows (the full-rate shape on gfx950).
#include "common.h"

// grad of the read-out entry idx as the backward kernels consume it: through the fused relu when the forward wrote relu(z)
// (a.z_act = that output; SGNN_MPN_RELU_Z) -- what a separate threshold launch per layer did before
__device__ static inline float mpn_gz(const sgnn_mpn_args& a, const float* __restrict__ grad_z, int64_t idx) {
    if (!grad_z) return 0.f;
    const float g = grad_z[idx];
    return (a.z_act && !(a.z_act[idx] > 0.f)) ? 0.f : g;
}

__device__ static inline float group_sum(float v, int lanes) {
    for (int d = lanes >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

#define MPN_U 4
// T: element type of the GATHER table (float, or __half for an fp16-stored table; fp32 accumulate)
// (bx, gx, by, gy: the workgroup's place in its body's own grid -- blockIdx / gridDim for a single launch, a body's share of a
// many-bodies launch otherwise)
template <int SRC, typename T = float>
__device__ __forceinline__ void mpn_fwd_body(const sgnn_mpn_args& a, float* __restrict__ agg, float* __restrict__ z, int64_t D4,
                                             int64_t bx, int64_t gx, int64_t by, int64_t gy)
{
    const int64_t total = a.R * D4;
    const float4* x4 = reinterpret_cast<const float4*>(a.x);
    const T* xt = reinterpret_cast<const T*>(a.x);
    const int lanes = (int)D4;
    const float bp = a.bp[0];
    const bool relu_z = (a.flags & SGNN_MPN_RELU_Z) != 0;
    // batch-sized calls (a few hundred component rows) do not fill the chip with one lane group per row and
    // walk their anchors one dependent load after the other: the anchors are then split over grid.y
    const int64_t a_per = (a.A + gy - 1) / gy;
    const int64_t a0 = by * a_per;
    const int64_t a1 = a0 + a_per < a.A ? a0 + a_per : a.A;
    for (int64_t t = bx * (int64_t)blockDim.x + threadIdx.x; t < total; t += gx * blockDim.x) {
        const int64_t r = t / D4, dv = t % D4;
        const bool row_real = a.row_mask ? (a.row_mask[r] != 0) : true;
        const float4 wp = reinterpret_cast<const float4*>(a.wp)[dv];
        const int64_t idrow = (a.id_div > 1 ? r / a.id_div : r) * a.A;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // MPN_U anchors at a time: their ids, then their weights, then their rows are requested together and only then consumed --
        // one anchor after the other (id -> weight -> row -> the read-out's store) a batch-sized call was a chain of 3 A dependent
        // round trips per lane (the stores to z keep the compiler from hoisting the next anchor's loads itself)
        for (int64_t ai0 = a0; ai0 < a1; ai0 += MPN_U) {
            int64_t id[MPN_U];
            bool edge[MPN_U];
            float w[MPN_U];
            float4 x[MPN_U];
#pragma unroll
            for (int u = 0; u < MPN_U; ++u) {
                const int64_t ai = ai0 + u;
                id[u] = 1;
                edge[u] = false;
                if (ai < a1) {
                    if (SRC == SGNN_SRC_DENSE) {
                        edge[u] = a.edge_mask[r * a.A + ai] != 0;
                        if (a.ids) id[u] = a.ids[idrow + ai];
                    } else if (SRC == SGNN_SRC_GATHER) {
                        id[u] = a.ids[idrow + ai];
                        edge[u] = (id[u] != 0) && row_real;
                    } else {
                        if (a.ids) id[u] = a.ids[ai];
                        edge[u] = row_real && (id[u] != 0);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < MPN_U; ++u) {
                w[u] = 0.f;
                if (edge[u]) {                                    // uniform over the row's lanes
                    const int64_t ai = ai0 + u;
                    const int64_t col = a.sim_col ? a.sim_col[ai] : (a.sims_per_edge ? ai : id[u] - 1);
                    w[u] = a.sims[r * a.sims_ld + col];
                }
            }
#pragma unroll
            for (int u = 0; u < MPN_U; ++u) {
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (edge[u]) {
                    const int64_t ai = ai0 + u;
                    if (SRC == SGNN_SRC_DENSE) x[u] = x4[(r * a.A + ai) * D4 + dv];
                    else if (SRC == SGNN_SRC_GATHER) x[u] = sgnn_load4<T>(xt, id[u], D4, dv);
                    else x[u] = x4[ai * D4 + dv];
                }
            }
#pragma unroll
            for (int u = 0; u < MPN_U; ++u) {
                const int64_t ai = ai0 + u;
                if (ai >= a1) break;
                float zval = bp;
                if (edge[u]) {
                    acc.x += w[u] * x[u].x; acc.y += w[u] * x[u].y; acc.z += w[u] * x[u].z; acc.w += w[u] * x[u].w;
                    const float dot = group_sum(wp.x * x[u].x + wp.y * x[u].y + wp.z * x[u].z + wp.w * x[u].w, lanes);
                    zval = w[u] * dot + bp;
                }
                if (dv == 0 && z) z[r * a.A + ai] = relu_z ? fmaxf(zval, 0.f) : zval;
            }
        }
        // anchor chunks of one row: each writes its partial aggregate to its own (R, D) slice, the caller adds them up
        reinterpret_cast<float4*>(agg)[by * total + t] = acc;
    }
}

template <int SRC, typename T = float>
__global__ __launch_bounds__(256) void mpn_fwd_kernel(sgnn_mpn_args a, float* __restrict__ agg, float* __restrict__ z,
                                                      int64_t D4)
{
    mpn_fwd_body<SRC, T>(a, agg, z, D4, blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}

// The layer bodies of ONE message-passing layer (up to three channels x two sides of a batch-sized step: they read the layer
// below only) in one launch: blockIdx.z = body; a body's own grid is (gx[k], chunks[k]) inside the launch's (max, max).
#define MPN_MAX_BODIES 8
struct MpnMany {
    sgnn_mpn_args a[MPN_MAX_BODIES];
    float* agg[MPN_MAX_BODIES];
    float* z[MPN_MAX_BODIES];
    int gx[MPN_MAX_BODIES], gy[MPN_MAX_BODIES];
};

__global__ __launch_bounds__(256) void mpn_fwd_many_kernel(const MpnMany M)
{
    const int k = blockIdx.z;
    if ((int)blockIdx.x >= M.gx[k] || (int)blockIdx.y >= M.gy[k]) return;
    const sgnn_mpn_args& a = M.a[k];
    const int64_t D4 = a.D / 4;
    if (a.src == SGNN_SRC_DENSE) mpn_fwd_body<SGNN_SRC_DENSE>(a, M.agg[k], M.z[k], D4, blockIdx.x, M.gx[k], blockIdx.y, M.gy[k]);
    else if (a.src == SGNN_SRC_GATHER && a.x_f16) mpn_fwd_body<SGNN_SRC_GATHER, __half>(a, M.agg[k], M.z[k], D4, blockIdx.x, M.gx[k], blockIdx.y, M.gy[k]);
    else if (a.src == SGNN_SRC_GATHER) mpn_fwd_body<SGNN_SRC_GATHER>(a, M.agg[k], M.z[k], D4, blockIdx.x, M.gx[k], blockIdx.y, M.gy[k]);
    else mpn_fwd_body<SGNN_SRC_SHARED>(a, M.agg[k], M.z[k], D4, blockIdx.x, M.gx[k], blockIdx.y, M.gy[k]);
}

// backward for DENSE (grad_x written) and GATHER (grad_x accumulated with atomics)
template <int SRC>
__global__ __launch_bounds__(256) void mpn_bwd_kernel(sgnn_mpn_args a, const float* __restrict__ grad_agg,
                                                      const float* __restrict__ grad_z, float* __restrict__ grad_x,
                                                      float* __restrict__ grad_wp, int64_t D4)
{
    __shared__ float s_gwp[1024];
    const int64_t D = D4 * 4;
    if (grad_wp) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) s_gwp[i] = 0.f;
        __syncthreads();
    }
    const int64_t total = a.R * D4;
    const float4* x4 = reinterpret_cast<const float4*>(a.x);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / D4, dv = t % D4;
        const bool row_real = a.row_mask ? (a.row_mask[r] != 0) : true;
        const float4 wp = reinterpret_cast<const float4*>(a.wp)[dv];
        float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grad_agg) ga = reinterpret_cast<const float4*>(grad_agg)[t];
        const int64_t idrow = (a.id_div > 1 ? r / a.id_div : r) * a.A;
        float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t ai = 0; ai < a.A; ++ai) {
            int64_t id = 1;
            bool edge;
            if (SRC == SGNN_SRC_DENSE) {
                edge = a.edge_mask[r * a.A + ai] != 0;
                if (a.ids) id = a.ids[idrow + ai];
            } else {
                id = a.ids[idrow + ai];
                edge = (id != 0) && row_real;
            }
            float4 dx = make_float4(0.f, 0.f, 0.f, 0.f);
            if (edge) {
                const int64_t col = a.sim_col ? a.sim_col[ai] : (a.sims_per_edge ? ai : id - 1);
                const float w = a.sims[r * a.sims_ld + col];
                const float gz = mpn_gz(a, grad_z, r * a.A + ai);
                dx.x = w * (ga.x + gz * wp.x); dx.y = w * (ga.y + gz * wp.y);
                dx.z = w * (ga.z + gz * wp.z); dx.w = w * (ga.w + gz * wp.w);
                if (grad_wp && gz != 0.f) {
                    const float4 x = (SRC == SGNN_SRC_DENSE) ? x4[(r * a.A + ai) * D4 + dv] : x4[id * D4 + dv];
                    const float s = gz * w;
                    gw.x += s * x.x; gw.y += s * x.y; gw.z += s * x.z; gw.w += s * x.w;
                }
                if (SRC == SGNN_SRC_GATHER && grad_x) {
                    float* dst = grad_x + id * D + dv * 4;
                    atomicAdd(dst + 0, dx.x); atomicAdd(dst + 1, dx.y); atomicAdd(dst + 2, dx.z); atomicAdd(dst + 3, dx.w);
                }
            }
            if (SRC == SGNN_SRC_DENSE && grad_x) reinterpret_cast<float4*>(grad_x)[(r * a.A + ai) * D4 + dv] = dx;
        }
        if (grad_wp) {
            if (a.flags & SGNN_MPN_WP_PARTIAL) {
                reinterpret_cast<float4*>(grad_wp)[t] = gw;                  // (R, D) partials, summed by the caller
            } else {
                atomicAdd(&s_gwp[dv * 4 + 0], gw.x); atomicAdd(&s_gwp[dv * 4 + 1], gw.y);
                atomicAdd(&s_gwp[dv * 4 + 2], gw.z); atomicAdd(&s_gwp[dv * 4 + 3], gw.w);
            }
        }
    }
    if (grad_wp && !(a.flags & SGNN_MPN_WP_PARTIAL)) {
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += blockDim.x) atomicAdd(&grad_wp[i], s_gwp[i]);
    }
}

// backward for GATHER with one lane per COLUMN: the lanes of a row are consecutive floats, so every
// float-atomic wave-instruction covers whole contiguous row segments (256 B for D = 64) -- the
// full-rate shape of global_atomic_add_f32 on gfx950; the float4-per-lane mapping of the forward
// pass would issue four strided 16-lane fragments per row instead.  Edges whose weight is exactly
// 0 (every N-internal edge: the similarity of a node inside the component is 0) add nothing and
// are skipped.
__global__ __launch_bounds__(256) void mpn_bwd_gather_kernel(sgnn_mpn_args a, const float* __restrict__ grad_agg,
                                                             const float* __restrict__ grad_z,
                                                             float* __restrict__ grad_x, float* __restrict__ grad_wp)
{
    __shared__ float s_gwp[1024];
    const int64_t D = a.D;
    if (grad_wp) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) s_gwp[i] = 0.f;
        __syncthreads();
    }
    const int64_t total = a.R * D;
    const int64_t a_per = (a.A + gridDim.y - 1) / gridDim.y;  // batch-sized calls split the anchors over grid.y
    const int64_t a0 = blockIdx.y * a_per;
    const int64_t a1 = a0 + a_per < a.A ? a0 + a_per : a.A;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / D, d = t % D;
        const bool row_real = a.row_mask ? (a.row_mask[r] != 0) : true;
        if (!row_real) continue;
        const float wp = a.wp[d];
        const float ga = grad_agg ? grad_agg[t] : 0.f;
        const int64_t idrow = (a.id_div > 1 ? r / a.id_div : r) * a.A;
        float gw = 0.f;
        for (int64_t ai = a0; ai < a1; ++ai) {
            const int64_t id = a.ids[idrow + ai];
            if (id == 0) continue;
            const int64_t col = a.sim_col ? a.sim_col[ai] : (a.sims_per_edge ? ai : id - 1);
            const float w = a.sims[r * a.sims_ld + col];
            if (w == 0.f) continue;
            const float gz = mpn_gz(a, grad_z, r * a.A + ai);
            if (grad_x) atomicAdd(grad_x + id * D + d, w * (ga + gz * wp));
            if (grad_wp && gz != 0.f)
                gw += gz * w * (a.x_f16 ? __half2float(reinterpret_cast<const __half*>(a.x)[id * D + d]) : a.x[id * D + d]);
        }
        if (grad_wp) atomicAdd(&s_gwp[d], gw);
    }
    if (grad_wp) {
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += blockDim.x) atomicAdd(&grad_wp[i], s_gwp[i]);
    }
}

// backward for SHARED anchors: dX[a,:] = sum_r edge * w[r,a] * (g_agg[r,:] + g_z[r,a] * wp).
// One workgroup per (tile of rows, chunk of 256 (anchor, column slice) items); the tile's g_agg
// rows are re-read per anchor from L1/L2; one atomic row-add per (tile, anchor).  The host picks
// the tile height: 64 rows when there are enough rows to fill the chip that way (large shards),
// down to 4 rows for a batch of a few hundred component rows, where the item chunks become the
// second grid dimension instead of a loop.
#define MPN_SH_TILE 64
__global__ __launch_bounds__(256) void mpn_bwd_shared_kernel(sgnn_mpn_args a, const float* __restrict__ grad_agg,
                                                             const float* __restrict__ grad_z,
                                                             float* __restrict__ grad_x, float* __restrict__ grad_wp,
                                                             int64_t D4, int64_t tile_rows)
{
    __shared__ float s_gwp[1024];
    const int64_t D = D4 * 4;
    if (grad_wp) {
        for (int i = threadIdx.x; i < D; i += blockDim.x) s_gwp[i] = 0.f;
        __syncthreads();
    }
    const float4* x4 = reinterpret_cast<const float4*>(a.x);
    const int64_t n_tiles = (a.R + tile_rows - 1) / tile_rows;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * tile_rows;
        const int64_t r1 = (r0 + tile_rows < a.R) ? r0 + tile_rows : a.R;
        for (int64_t item = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; item < a.A * D4;
             item += (int64_t)gridDim.y * blockDim.x) {
            const int64_t ai = item / D4, dv = item % D4;
            const int64_t id = a.ids ? a.ids[ai] : 1;
            if (id == 0) continue;
            const int64_t col = a.sim_col ? a.sim_col[ai] : (a.sims_per_edge ? ai : id - 1);
            const float4 wp = reinterpret_cast<const float4*>(a.wp)[dv];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            float sgz = 0.f;
            for (int64_t r = r0; r < r1; ++r) {
                if (a.row_mask && !a.row_mask[r]) continue;
                const float w = a.sims[r * a.sims_ld + col];
                const float gz = mpn_gz(a, grad_z, r * a.A + ai);
                float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
                if (grad_agg) ga = reinterpret_cast<const float4*>(grad_agg)[r * D4 + dv];
                acc.x += w * (ga.x + gz * wp.x); acc.y += w * (ga.y + gz * wp.y);
                acc.z += w * (ga.z + gz * wp.z); acc.w += w * (ga.w + gz * wp.w);
                sgz += w * gz;
            }
            if (grad_x) {
                float* dst = grad_x + ai * D + dv * 4;
                atomicAdd(dst + 0, acc.x); atomicAdd(dst + 1, acc.y); atomicAdd(dst + 2, acc.z); atomicAdd(dst + 3, acc.w);
            }
            if (grad_wp) {
                const float4 x = x4[ai * D4 + dv];
                atomicAdd(&s_gwp[dv * 4 + 0], sgz * x.x); atomicAdd(&s_gwp[dv * 4 + 1], sgz * x.y);
                atomicAdd(&s_gwp[dv * 4 + 2], sgz * x.z); atomicAdd(&s_gwp[dv * 4 + 3], sgz * x.w);
            }
        }
    }
    if (grad_wp) {
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += blockDim.x) atomicAdd(&grad_wp[i], s_gwp[i]);
    }
}

// ---- deterministic backward for SHARED anchors -----------------------------------------------------------
// The same contraction as mpn_bwd_shared_kernel, without atomics: every (row tile, anchor, column slice) item writes
// its partial row to part_x[tile][anchor][:] and the tile's sum of w * g_z per anchor to part_s[tile][anchor]; a second
// kernel adds the tiles in order (dX) and forms grad_wp[d] = sum_a (sum_tiles part_s[.][a]) * X[a][d] in anchor order.
// Batch-sized calls only (the shard-sized ones are library GEMMs): the partials are n_tiles x A x D floats.
__global__ __launch_bounds__(256) void mpn_bwd_shared_det_kernel(sgnn_mpn_args a, const float* __restrict__ grad_agg,
                                                                 const float* __restrict__ grad_z,
                                                                 float* __restrict__ part_x, float* __restrict__ part_s,
                                                                 float* __restrict__ part_b, int64_t D4, int64_t tile_rows)
{
    const int64_t D = D4 * 4;
    const int64_t tile = blockIdx.x;
    const int64_t r0 = tile * tile_rows;
    const int64_t r1 = (r0 + tile_rows < a.R) ? r0 + tile_rows : a.R;
    for (int64_t item = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; item < a.A * D4;
         item += (int64_t)gridDim.y * blockDim.x) {
        const int64_t ai = item / D4, dv = item % D4;
        const int64_t id = a.ids ? a.ids[ai] : 1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float sgz = 0.f;
        if (id != 0) {
            const int64_t col = a.sim_col ? a.sim_col[ai] : (a.sims_per_edge ? ai : id - 1);
            const float4 wp = reinterpret_cast<const float4*>(a.wp)[dv];
            for (int64_t r = r0; r < r1; ++r) {
                if (a.row_mask && !a.row_mask[r]) continue;
                const float w = a.sims[r * a.sims_ld + col];
                const float gz = mpn_gz(a, grad_z, r * a.A + ai);
                float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
                if (grad_agg) ga = reinterpret_cast<const float4*>(grad_agg)[r * D4 + dv];
                acc.x += w * (ga.x + gz * wp.x); acc.y += w * (ga.y + gz * wp.y);
                acc.z += w * (ga.z + gz * wp.z); acc.w += w * (ga.w + gz * wp.w);
                sgz += w * gz;
            }
        }
        reinterpret_cast<float4*>(part_x)[(tile * a.A + ai) * D4 + dv] = acc;
        if (dv == 0) {
            part_s[tile * a.A + ai] = sgz;
            if (part_b) {                 // grad_bp's share: the read-out bias reaches every entry, masked ones included
                float sb = 0.f;
                for (int64_t r = r0; r < r1; ++r) sb += mpn_gz(a, grad_z, r * a.A + ai);
                part_b[tile * a.A + ai] = sb;
            }
        }
    }
}

__global__ __launch_bounds__(256) void mpn_bwd_shared_reduce_kernel(sgnn_mpn_args a, const float* __restrict__ part_x,
                                                                    const float* __restrict__ part_s, int64_t n_tiles,
                                                                    float* __restrict__ grad_x, float* __restrict__ grad_wp,
                                                                    const float* __restrict__ part_b, float* __restrict__ grad_bp)
{
    if (grad_bp && blockIdx.x == gridDim.x - 1) {
        // one wavefront's worth of work: per lane a strided share of the (tile, anchor) partials, then the lanes in order
        __shared__ float s_b[256];
        float v = 0.f;
        for (int64_t k = threadIdx.x; k < n_tiles * a.A; k += blockDim.x) v += part_b[k];
        s_b[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int k = 0; k < 256; ++k) t += s_b[k];
            grad_bp[0] = t;
        }
    }
    const int64_t AD = a.A * a.D;
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (grad_x && t < AD) {
        float s = 0.f;
#pragma unroll 8
        for (int64_t tile = 0; tile < n_tiles; ++tile) s += part_x[tile * AD + t];      // (loads independent, adds in tile order)
        grad_x[t] = s;
    }
    if (grad_wp && blockIdx.x == 0) {
        // per anchor, the tiles' sums of w * g_z in tile order (one anchor per thread, kept in the workspace's first
        // row of part_s -- nobody reads that row any more), then grad_wp[d] = sum over the anchors, in anchor order
        float* sg = const_cast<float*>(part_s);
        for (int64_t ai = threadIdx.x; ai < a.A; ai += blockDim.x) {
            float v = 0.f;
#pragma unroll 8
            for (int64_t tile = 0; tile < n_tiles; ++tile) v += part_s[tile * a.A + ai];
            sg[ai] = v;
        }
        __syncthreads();
        for (int64_t d = threadIdx.x; d < a.D; d += blockDim.x) {
            float g = 0.f;
#pragma unroll 8
            for (int64_t ai = 0; ai < a.A; ++ai) g += sg[ai] * a.x[ai * a.D + d];
            grad_wp[d] = g;
        }
    }
}

// ---- deterministic backward of the GATHER source (embedding-table gradient without atomics) ---------------
// Per edge (component row r, anchor slot ai) of a GATHER layer: the target table row and the two
// coefficients of its contribution  dE[id, :] += w * g_agg[r, :] + (w * g_z[r, ai]) * wp  -- the input of
// sgnn_scatter_add_rows_sorted (scatter.hip) once the keys are sorted.  Masked edges (PAD anchor, padded
// component row, weight exactly 0) get key 0.
__device__ __forceinline__ void mpn_bwd_edges_body(const sgnn_mpn_args& a, const float* __restrict__ grad_z,
                                                   int32_t* __restrict__ keys, float* __restrict__ c1, float* __restrict__ c2,
                                                   int64_t bx, int64_t gx)
{
    const int64_t total = a.R * a.A;
    for (int64_t e = bx * (int64_t)blockDim.x + threadIdx.x; e < total; e += gx * blockDim.x) {
        const int64_t r = e / a.A, ai = e % a.A;
        const bool row_real = a.row_mask ? (a.row_mask[r] != 0) : true;
        const int64_t id = a.ids[(a.id_div > 1 ? r / a.id_div : r) * a.A + ai];
        float w = 0.f;
        if (row_real && id != 0) {
            const int64_t col = a.sim_col ? a.sim_col[ai] : (a.sims_per_edge ? ai : id - 1);
            w = a.sims[r * a.sims_ld + col];
        }
        keys[e] = w != 0.f ? (int32_t)id : 0;
        c1[e] = w;
        if (c2) c2[e] = w * mpn_gz(a, grad_z, e);
    }
}

__global__ __launch_bounds__(256) void mpn_bwd_edges_kernel(sgnn_mpn_args a, const float* __restrict__ grad_z,
                                                            int32_t* __restrict__ keys, float* __restrict__ c1,
                                                            float* __restrict__ c2)
{
    mpn_bwd_edges_body(a, grad_z, keys, c1, c2, blockIdx.x, gridDim.x);
}

// the edge lists of several GATHER bodies in one launch (their consumer is the step's combined table-gradient scatter, which
// runs when the table's gradient is handed over: the lists can wait for each other until then)
struct MpnEdgesMany {
    sgnn_mpn_args a[MPN_MAX_BODIES];
    const float* gz[MPN_MAX_BODIES];
    int32_t* keys[MPN_MAX_BODIES];
    float* c1[MPN_MAX_BODIES];
    float* c2[MPN_MAX_BODIES];
    int gx[MPN_MAX_BODIES];
};

__global__ __launch_bounds__(256) void mpn_bwd_edges_many_kernel(const MpnEdgesMany M)
{
    const int k = blockIdx.y;
    if ((int)blockIdx.x >= M.gx[k]) return;
    mpn_bwd_edges_body(M.a[k], M.gz[k], M.keys[k], M.c1[k], M.c2[k], blockIdx.x, M.gx[k]);
}

// grad_wp of a GATHER layer as per-row partial sums (no atomics): partial[r, d] = sum_ai g_z[r, ai] * w * x[id, d];
// the caller sums the rows (a fixed reduction tree).
__global__ __launch_bounds__(256) void mpn_bwd_wp_partial_kernel(sgnn_mpn_args a, const float* __restrict__ grad_z,
                                                                 float* __restrict__ partial, int64_t ld)
{
    // ld == D + 1: column D of a row is the sum of the row's (gated) read-out gradients -- grad_bp's per-row partial: the
    // read-out bias reaches EVERY entry (masked edges and padded rows read out bp itself)
    const int64_t D = a.D, total = a.R * ld;
    // work items [0, R D): the weight partials; [R D, R ld): one per row for the bias column -- kept apart from the others so that
    // only the last wavefronts run that branch (interleaved as column D of each row it doubled the launch's time)
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        if (t >= a.R * D) {
            const int64_t r = t - a.R * D;
            float gb = 0.f;
            for (int64_t ai = 0; ai < a.A; ++ai) gb += mpn_gz(a, grad_z, r * a.A + ai);
            partial[r * ld + D] = gb;
            continue;
        }
        const int64_t r = t / D, d = t % D;
        float gw = 0.f;
        if (!a.row_mask || a.row_mask[r]) {
            const int64_t idrow = (a.id_div > 1 ? r / a.id_div : r) * a.A;
            // four anchors at a time: ids and read-out gradients, then weights, then table elements are requested together
            // (one anchor after the other a row of 34-90 anchors was 4 A dependent round trips: 17 us for 128 rows)
            for (int64_t ai0 = 0; ai0 < a.A; ai0 += MPN_U) {
                int64_t id[MPN_U];
                float gz[MPN_U], w[MPN_U], xv[MPN_U];
#pragma unroll
                for (int u = 0; u < MPN_U; ++u) {
                    const int64_t ai = ai0 + u;
                    id[u] = ai < a.A ? a.ids[idrow + ai] : 0;
                    gz[u] = ai < a.A ? mpn_gz(a, grad_z, r * a.A + ai) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < MPN_U; ++u) {
                    const int64_t ai = ai0 + u;
                    w[u] = 0.f;
                    if (id[u] != 0 && gz[u] != 0.f) {
                        const int64_t col = a.sim_col ? a.sim_col[ai] : (a.sims_per_edge ? ai : id[u] - 1);
                        w[u] = a.sims[r * a.sims_ld + col];
                    }
                }
#pragma unroll
                for (int u = 0; u < MPN_U; ++u) {
                    xv[u] = 0.f;
                    if (w[u] != 0.f)
                        xv[u] = a.x_f16 ? __half2float(reinterpret_cast<const __half*>(a.x)[id[u] * D + d]) : a.x[id[u] * D + d];
                }
#pragma unroll
                for (int u = 0; u < MPN_U; ++u)
                    if (w[u] != 0.f) gw += gz[u] * w[u] * xv[u];
            }
        }
        partial[r * ld + d] = gw;
    }
}

static int mpn_check(const sgnn_mpn_args* a)
{
    if (!a || a->R < 0 || a->A < 0 || a->D <= 0 || !a->x || !a->sims || !a->wp || !a->bp) return SGNN_ERR_BAD_ARG;
    if (a->D % 4 != 0) return SGNN_ERR_UNSUPPORTED_D;
    const int64_t D4 = a->D / 4;
    if (D4 > 64 || (D4 & (D4 - 1)) != 0 || a->D > 1024) return SGNN_ERR_UNSUPPORTED_D;   // lanes per row: 1,2,4..64
    if (a->src == SGNN_SRC_DENSE && !a->edge_mask) return SGNN_ERR_BAD_ARG;
    if (a->src == SGNN_SRC_GATHER && !a->ids) return SGNN_ERR_BAD_ARG;
    if (a->src < 0 || a->src > 2) return SGNN_ERR_BAD_ARG;
    if (a->x_f16 && a->src != SGNN_SRC_GATHER) return SGNN_ERR_BAD_ARG;
    if (!a->sim_col && !a->sims_per_edge && !a->ids) return SGNN_ERR_BAD_ARG;
    if (a->id_div < 1) return SGNN_ERR_BAD_ARG;
    return SGNN_OK;
}

static int mpn_fwd_chunks(const sgnn_mpn_args* args)
{
    const int gx = sgnn_grid_for(args->R * (args->D / 4), 256);
    int chunks = 1;
    if (gx < 512 && args->A >= 16) {                          // too few rows to fill 256 CUs: split the anchors
        chunks = (1024 + gx - 1) / gx;
        const int64_t most = (args->A + 7) / 8;               // at least 8 anchors per chunk
        if (chunks > most) chunks = (int)most;
    }
    return chunks < 1 ? 1 : chunks;
}

extern "C" int sgnn_mpn_fwd_chunks(const sgnn_mpn_args* args)
{
    if (mpn_check(args) != SGNN_OK || args->R == 0) return 1;
    return mpn_fwd_chunks(args);
}

extern "C" int sgnn_mpn_fwd(const sgnn_mpn_args* args, float* agg, float* z, void* stream)
{
    const int rc = mpn_check(args);
    if (rc != SGNN_OK) return rc;
    if (!agg) return SGNN_ERR_BAD_ARG;
    if (args->R == 0) return SGNN_OK;
    const int64_t D4 = args->D / 4;
    const int gx = sgnn_grid_for(args->R * D4, 256);
    hipStream_t st = (hipStream_t)stream;
    const int chunks = mpn_fwd_chunks(args);
    const dim3 grid(gx, chunks);
    if (args->src == SGNN_SRC_DENSE)
        hipLaunchKernelGGL(mpn_fwd_kernel<SGNN_SRC_DENSE>, grid, dim3(256), 0, st, *args, agg, z, D4);
    else if (args->src == SGNN_SRC_GATHER && args->x_f16)
        hipLaunchKernelGGL((mpn_fwd_kernel<SGNN_SRC_GATHER, __half>), grid, dim3(256), 0, st, *args, agg, z, D4);
    else if (args->src == SGNN_SRC_GATHER)
        hipLaunchKernelGGL(mpn_fwd_kernel<SGNN_SRC_GATHER>, grid, dim3(256), 0, st, *args, agg, z, D4);
    else
        hipLaunchKernelGGL(mpn_fwd_kernel<SGNN_SRC_SHARED>, grid, dim3(256), 0, st, *args, agg, z, D4);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_mpn_bwd(const sgnn_mpn_args* args, const float* grad_agg, const float* grad_z,
                            float* grad_x, float* grad_wp, void* stream)
{
    const int rc = mpn_check(args);
    if (rc != SGNN_OK) return rc;
    if (args->R == 0 || args->A == 0) return SGNN_OK;
    const int64_t D4 = args->D / 4;
    hipStream_t st = (hipStream_t)stream;
    if (args->src == SGNN_SRC_SHARED) {
        int64_t tile_rows = MPN_SH_TILE, chunks = 1;
        int64_t n_tiles = (args->R + tile_rows - 1) / tile_rows;
        if (n_tiles < 1024) {                       // too few rows to fill 256 CUs with 64-row tiles
            chunks = (args->A * D4 + 255) / 256;
            const int64_t want = (1024 + chunks - 1) / chunks;
            tile_rows = (args->R + want - 1) / want;
            tile_rows = tile_rows < 4 ? 4 : (tile_rows > MPN_SH_TILE ? MPN_SH_TILE : tile_rows);
            n_tiles = (args->R + tile_rows - 1) / tile_rows;
        }
        const int grid = (int)(n_tiles < 4096 ? n_tiles : 4096);
        hipLaunchKernelGGL(mpn_bwd_shared_kernel, dim3(grid, (unsigned)chunks), dim3(256), 0, st, *args, grad_agg, grad_z,
                           grad_x, grad_wp, D4, tile_rows);
    } else {
        const int grid = sgnn_grid_for(args->R * D4, 256, 2048);
        if (args->src == SGNN_SRC_DENSE)
            hipLaunchKernelGGL(mpn_bwd_kernel<SGNN_SRC_DENSE>, dim3(grid), dim3(256), 0, st, *args, grad_agg, grad_z,
                               grad_x, grad_wp, D4);
        else
        {
            const int gx = sgnn_grid_for(args->R * args->D, 256, 8192);
            int chunks = 1;
            if (gx < 512 && args->A >= 16) {
                chunks = (1024 + gx - 1) / gx;
                const int64_t most = (args->A + 7) / 8;
                if (chunks > most) chunks = (int)most;
            }
            hipLaunchKernelGGL(mpn_bwd_gather_kernel, dim3(gx, chunks), dim3(256), 0, st, *args, grad_agg, grad_z, grad_x,
                               grad_wp);
        }
    }
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_mpn_bwd_edges(const sgnn_mpn_args* args, const float* grad_z, int32_t* out_keys, float* out_c1,
                                  float* out_c2, void* stream)
{
    const int rc = mpn_check(args);
    if (rc != SGNN_OK) return rc;
    if (args->src != SGNN_SRC_GATHER || !out_keys || !out_c1) return SGNN_ERR_BAD_ARG;
    if (args->R * args->A == 0) return SGNN_OK;
    hipLaunchKernelGGL(mpn_bwd_edges_kernel, dim3(sgnn_grid_for(args->R * args->A, 256, 8192)), dim3(256), 0,
                       (hipStream_t)stream, *args, grad_z, out_keys, out_c1, out_c2);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_mpn_bwd_edges_many(int64_t n, const sgnn_mpn_args* args, const float* const* grad_z, int32_t* const* out_keys,
                                       float* const* out_c1, float* const* out_c2, void* stream)
{
    if (n < 1 || n > MPN_MAX_BODIES || !args || !grad_z || !out_keys || !out_c1 || !out_c2) return SGNN_ERR_BAD_ARG;
    MpnEdgesMany M;
    int mx = 0;
    for (int k = 0; k < n; ++k) {
        const int rc = mpn_check(&args[k]);
        if (rc != SGNN_OK) return rc;
        if (args[k].src != SGNN_SRC_GATHER || !out_keys[k] || !out_c1[k] || args[k].R * args[k].A <= 0) return SGNN_ERR_BAD_ARG;
        M.a[k] = args[k]; M.gz[k] = grad_z[k]; M.keys[k] = out_keys[k]; M.c1[k] = out_c1[k]; M.c2[k] = out_c2[k];
        M.gx[k] = sgnn_grid_for(args[k].R * args[k].A, 256, 8192);
        if (M.gx[k] > mx) mx = M.gx[k];
    }
    hipLaunchKernelGGL(mpn_bwd_edges_many_kernel, dim3(mx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, M);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_mpn_bwd_wp_partial(const sgnn_mpn_args* args, const float* grad_z, float* partial, int64_t partial_ld,
                                       void* stream)
{
    const int rc = mpn_check(args);
    if (rc != SGNN_OK) return rc;
    if (args->src != SGNN_SRC_GATHER || !grad_z || !partial) return SGNN_ERR_BAD_ARG;
    if (partial_ld != args->D && partial_ld != args->D + 1) return SGNN_ERR_BAD_ARG;
    if (args->R == 0) return SGNN_OK;
    hipLaunchKernelGGL(mpn_bwd_wp_partial_kernel, dim3(sgnn_grid_for(args->R * partial_ld, 256, 8192)), dim3(256), 0,
                       (hipStream_t)stream, *args, grad_z, partial, partial_ld);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

static void mpn_shared_det_tiling(int64_t R, int64_t A, int64_t D4, int64_t* tile_rows, int64_t* n_tiles, int64_t* chunks)
{
    *chunks = (A * D4 + 255) / 256;
    const int64_t want = (512 + *chunks - 1) / *chunks;                 // ~512 workgroups: two per CU; fewer tiles = a shorter reduction
    int64_t tr = (R + want - 1) / want;
    tr = tr < 4 ? 4 : (tr > MPN_SH_TILE ? MPN_SH_TILE : tr);
    *tile_rows = tr;
    *n_tiles = (R + tr - 1) / tr;
}

extern "C" int64_t sgnn_mpn_bwd_shared_det_workspace_bytes(int64_t R, int64_t A, int64_t D)
{
    if (R <= 0 || A <= 0 || D <= 0) return 0;
    int64_t tr, nt, ch;
    mpn_shared_det_tiling(R, A, D / 4, &tr, &nt, &ch);
    return nt * A * (D + 2) * 4 + 64;
}

extern "C" int sgnn_mpn_bwd_shared_det(const sgnn_mpn_args* args, const float* grad_agg, const float* grad_z,
                                       float* grad_x, float* grad_wp, float* grad_bp, void* workspace, int64_t workspace_bytes,
                                       void* stream)
{
    const int rc = mpn_check(args);
    if (rc != SGNN_OK) return rc;
    if (args->src != SGNN_SRC_SHARED) return SGNN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (args->R == 0 || args->A == 0) return SGNN_OK;
    if (!workspace || workspace_bytes < sgnn_mpn_bwd_shared_det_workspace_bytes(args->R, args->A, args->D)) return SGNN_ERR_BAD_ARG;
    const int64_t D4 = args->D / 4;
    int64_t tile_rows, n_tiles, chunks;
    mpn_shared_det_tiling(args->R, args->A, D4, &tile_rows, &n_tiles, &chunks);
    float* part_x = (float*)workspace;
    float* part_s = part_x + n_tiles * args->A * args->D;
    float* part_b = grad_bp ? part_s + n_tiles * args->A : nullptr;
    hipLaunchKernelGGL(mpn_bwd_shared_det_kernel, dim3((unsigned)n_tiles, (unsigned)chunks), dim3(256), 0, st, *args, grad_agg,
                       grad_z, part_x, part_s, part_b, D4, tile_rows);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(mpn_bwd_shared_reduce_kernel, dim3((unsigned)((args->A * args->D + 255) / 256)), dim3(256), 0, st, *args,
                       part_x, part_s, n_tiles, grad_x, grad_wp, part_b, grad_bp);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}


extern "C" int64_t sgnn_mpn_fwd_many_max_bodies(void) { return MPN_MAX_BODIES; }

/* args: HOST array of n sgnn_mpn_args; agg[k] (sgnn_mpn_fwd_chunks(&args[k]), R_k, D_k) and z[k] (R_k, A_k) as sgnn_mpn_fwd writes
 * them (z[k] nullable).  Bodies with R = 0 or A = 0 are the caller's to handle. */
extern "C" int sgnn_mpn_fwd_many(int64_t n, const sgnn_mpn_args* args, float* const* agg, float* const* z, void* stream)
{
    if (n < 1 || n > MPN_MAX_BODIES || !args || !agg || !z) return SGNN_ERR_BAD_ARG;
    MpnMany M;
    int mx = 0, my = 0;
    for (int k = 0; k < n; ++k) {
        const int rc = mpn_check(&args[k]);
        if (rc != SGNN_OK) return rc;
        if (!agg[k] || args[k].R <= 0 || args[k].A <= 0) return SGNN_ERR_BAD_ARG;
        M.a[k] = args[k]; M.agg[k] = agg[k]; M.z[k] = z[k];
        M.gx[k] = sgnn_grid_for(args[k].R * (args[k].D / 4), 256);
        M.gy[k] = mpn_fwd_chunks(&args[k]);
        if (M.gx[k] > mx) mx = M.gx[k];
        if (M.gy[k] > my) my = M.gy[k];
    }
    hipLaunchKernelGGL(mpn_fwd_many_kernel, dim3(mx, my, (unsigned)n), dim3(256), 0, (hipStream_t)stream, M);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(mpn)
