// The tail of the forward pass, fused: the position / structure read-outs of a layer over SHARED anchors and the
// assembly of the subgraph embedding (reference SubGNN/subgraph_mpn.py:122-131 -- position read-out of the messages,
// relu -- and SubGNN/SubGNN.py:286-312 -- concatenation of the channel outputs, masked sum over a subgraph's
// components).
//
// For a layer whose anchors are shared by all component rows (position-border anchors, structure patches) the message
// of edge (row r, anchor a) is W[r,a] * X[a,:] and its read-out is W[r,a] * (X[a,:] . wp) + bp: with s = X wp (A values,
// a tiny product left to the caller) the whole (R, A) read-out is an element-wise function of the similarity rows.  What
// the model consumes of it, when the layer is the channel's last one, is only
//     out[b, a] = sum over the real components c of subgraph b of relu(W[b,c,a] * s[a] + bp)
// which lands in a column slot of the (B, H) subgraph embedding.  The library form materialised W (column select, two
// mask multiplies), the read-out (addcmul), its relu, the concatenation of all channel outputs into (B, C, H) and the
// masked sum: 7 passes over 50k x 183 floats and 2 over 50k x 516 on the benchmark; here one pass reads the similarity
// rows and writes the slot.  Backward: d s[a] = sum_r g[b(r), a] [z > 0] W[r, a], d bp = sum_{r,a} g [z > 0]: per
// row-block partials, then one workgroup adds them in block order -- no atomics, bit-reproducible.
#include "common.h"

#define RO_ROWS_PER_BLOCK 128

// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void readout_sum_fwd_kernel(const float* __restrict__ sims, int64_t ld,
                                                              const int64_t* __restrict__ sim_col, const float* __restrict__ s,
                                                              const float* __restrict__ bp, const uint8_t* __restrict__ row_mask,
                                                              int64_t B, int64_t C, int64_t A, float* __restrict__ out, int64_t out_ld)
{
    const int64_t total = B * A;
    const float bb = bp[0];
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / A, a = t - b * A;
        const float sa = s[a];
        const int64_t col = sim_col ? sim_col[a] : a;
        float acc = 0.f;
        for (int64_t c = 0; c < C; ++c) {
            const int64_t r = b * C + c;
            if (row_mask && !row_mask[r]) continue;
            const float w = sims ? sims[r * ld + col] : 0.f;
            acc += fmaxf(fmaf(w, sa, bb), 0.f);
        }
        out[b * out_ld + a] = acc;
    }
}

// one workgroup per block of RO_ROWS_PER_BLOCK component rows; thread = (row lane tr, column lane ta); the row lanes'
// sums are added in lane order through LDS.  partial layout [2][A][nblk] (d s, then d bp), block index fastest.
template <int TA>
__global__ __launch_bounds__(256) void readout_sum_bwd_partial_kernel(const float* __restrict__ g, int64_t g_ld,
                                                                      const float* __restrict__ sims, int64_t ld,
                                                                      const int64_t* __restrict__ sim_col, const float* __restrict__ s,
                                                                      const float* __restrict__ bp, const uint8_t* __restrict__ row_mask,
                                                                      int64_t R, int64_t C, int64_t A, int64_t nblk,
                                                                      float* __restrict__ partial)
{
    constexpr int TR = 256 / TA;
    __shared__ float sh_s[TR][TA], sh_b[TR][TA];
    const int ta = threadIdx.x % TA, tr = threadIdx.x / TA;
    const int64_t blk = blockIdx.x;
    const int64_t r0 = blk * RO_ROWS_PER_BLOCK, r1 = (r0 + RO_ROWS_PER_BLOCK < R) ? r0 + RO_ROWS_PER_BLOCK : R;
    const float bb = bp[0];
    for (int64_t a0 = 0; a0 < A; a0 += TA) {
        const int64_t a = a0 + ta;
        float acc_s = 0.f, acc_b = 0.f;
        if (a < A) {
            const float sa = s[a];
            const int64_t col = sim_col ? sim_col[a] : a;
            for (int64_t r = r0 + tr; r < r1; r += TR) {
                if (row_mask && !row_mask[r]) continue;
                const float w = sims ? sims[r * ld + col] : 0.f;
                if (fmaf(w, sa, bb) > 0.f) {
                    const float gz = g[(r / C) * g_ld + a];
                    acc_s = fmaf(gz, w, acc_s);
                    acc_b += gz;
                }
            }
        }
        sh_s[tr][ta] = acc_s;
        sh_b[tr][ta] = acc_b;
        __syncthreads();
        if (tr == 0 && a < A) {
            float vs = sh_s[0][ta], vb = sh_b[0][ta];
#pragma unroll
            for (int k = 1; k < TR; ++k) { vs += sh_s[k][ta]; vb += sh_b[k][ta]; }
            partial[a * nblk + blk] = vs;
            partial[(A + a) * nblk + blk] = vb;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float ro_wave_sum(float v) {          // fixed tree: the same order every run
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one workgroup of 16 wavefronts: wavefront w owns columns w, w + 16, ...; a lane adds every 64th block partial, the
// wavefront's lanes are added by a fixed butterfly; d bp = the column sums added in column order per wavefront, then the
// 16 wavefront sums in order.
__global__ __launch_bounds__(1024) void readout_sum_bwd_finish_kernel(const float* __restrict__ partial, int64_t A, int64_t nblk,
                                                                      float* __restrict__ grad_s, float* __restrict__ grad_bp)
{
    __shared__ float sh[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float bsum = 0.f;
    for (int64_t a = wave; a < A; a += 16) {
        float vs = 0.f, vb = 0.f;
        const float* ps = partial + a * nblk;
        const float* pb = partial + (A + a) * nblk;
        for (int64_t k = lane; k < nblk; k += 64) { vs += ps[k]; vb += pb[k]; }
        vs = ro_wave_sum(vs);
        vb = ro_wave_sum(vb);
        if (lane == 0 && grad_s) grad_s[a] = vs;
        bsum += vb;
    }
    if (lane == 0) sh[wave] = bsum;
    __syncthreads();
    if (threadIdx.x == 0 && grad_bp) {
        float v = sh[0];
        for (int k = 1; k < 16; ++k) v += sh[k];
        grad_bp[0] = v;
    }
}

extern "C" int sgnn_readout_sum_fwd(const float* sims, int64_t sims_ld, const int64_t* sim_col, const float* s, const float* bp,
                                    const uint8_t* row_mask, int64_t B, int64_t C, int64_t A, float* out, int64_t out_ld,
                                    void* stream)
{
    if (!s || !bp || !out || B < 0 || C < 0 || A < 0 || out_ld < A || (sims && sims_ld < 1)) return SGNN_ERR_BAD_ARG;
    if (B * A == 0) return SGNN_OK;
    hipLaunchKernelGGL(readout_sum_fwd_kernel, dim3(sgnn_grid_for(B * A, 256)), dim3(256), 0, (hipStream_t)stream, sims, sims_ld,
                       sim_col, s, bp, row_mask, B, C, A, out, out_ld);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

static inline int64_t ro_blocks(int64_t R) { return (R + RO_ROWS_PER_BLOCK - 1) / RO_ROWS_PER_BLOCK; }

extern "C" int64_t sgnn_readout_sum_bwd_workspace_bytes(int64_t B, int64_t C, int64_t A)
{
    if (B < 0 || C < 0 || A < 0) return -1;
    return 2 * A * ro_blocks(B * C) * (int64_t)sizeof(float) + 16;
}

extern "C" int sgnn_readout_sum_bwd(const float* grad_out, int64_t grad_ld, const float* sims, int64_t sims_ld,
                                    const int64_t* sim_col, const float* s, const float* bp, const uint8_t* row_mask, int64_t B,
                                    int64_t C, int64_t A, float* grad_s, float* grad_bp, void* workspace, int64_t workspace_bytes,
                                    void* stream)
{
    if (!grad_out || !s || !bp || B < 0 || C < 0 || A < 0 || grad_ld < A || (sims && sims_ld < 1)) return SGNN_ERR_BAD_ARG;
    if (!grad_s && !grad_bp) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t R = B * C;
    if (R * A == 0) {
        if (grad_s && A) { if (hipMemsetAsync(grad_s, 0, A * sizeof(float), st) != hipSuccess) return SGNN_ERR_LAUNCH; }
        if (grad_bp) { if (hipMemsetAsync(grad_bp, 0, sizeof(float), st) != hipSuccess) return SGNN_ERR_LAUNCH; }
        return SGNN_OK;
    }
    if (!workspace || workspace_bytes < sgnn_readout_sum_bwd_workspace_bytes(B, C, A)) return SGNN_ERR_BAD_ARG;
    float* partial = (float*)workspace;
    const int64_t nblk = ro_blocks(R);
#define RO_LAUNCH(TA) hipLaunchKernelGGL(readout_sum_bwd_partial_kernel<TA>, dim3((unsigned)nblk), dim3(256), 0, st, grad_out, \
                                         grad_ld, sims, sims_ld, sim_col, s, bp, row_mask, R, C, A, nblk, partial)
    if (A <= 32) RO_LAUNCH(32);
    else if (A <= 64) RO_LAUNCH(64);
    else if (A <= 128) RO_LAUNCH(128);
    else RO_LAUNCH(256);
#undef RO_LAUNCH
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(readout_sum_bwd_finish_kernel, dim3(1), dim3(1024), 0, st, partial, A, nblk, grad_s, grad_bp);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// masked sum of one channel output (B, C, W) into a column slot of the (B, H) subgraph embedding, and its backward out of
// a slot of the embedding's gradient (SubGNN/subgraph_utils.py:213-237 applied per concatenated piece: no (B, C, H) tensor)
template <typename V>
__global__ __launch_bounds__(256) void masked_sum_slot_fwd_kernel(const V* __restrict__ x, const uint8_t* __restrict__ mask, int64_t B,
                                                                  int64_t C, int64_t Wv, V* __restrict__ out, int64_t out_ldv)
{
    const int64_t total = B * Wv;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / Wv, h = t - b * Wv;
        V acc;
        if constexpr (sizeof(V) == 16) acc = make_float4(0.f, 0.f, 0.f, 0.f); else acc = 0.f;
        for (int64_t c = 0; c < C; ++c)
            if (mask[b * C + c]) {
                const V v = x[(b * C + c) * Wv + h];
                if constexpr (sizeof(V) == 16) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; } else acc += v;
            }
        out[b * out_ldv + h] = acc;
    }
}

template <typename V>
__global__ __launch_bounds__(256) void masked_sum_slot_bwd_kernel(const V* __restrict__ g, int64_t g_ldv, const uint8_t* __restrict__ mask,
                                                                  int64_t B, int64_t C, int64_t Wv, V* __restrict__ gx)
{
    const int64_t total = B * C * Wv;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bc = t / Wv, h = t - bc * Wv;
        V z;
        if constexpr (sizeof(V) == 16) z = make_float4(0.f, 0.f, 0.f, 0.f); else z = 0.f;
        gx[t] = mask[bc] ? g[(bc / C) * g_ldv + h] : z;
    }
}

static inline bool ro_vec4_ok(const void* a, const void* b, int64_t W, int64_t ld) {
    return W % 4 == 0 && ld % 4 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

extern "C" int sgnn_masked_sum_slot_fwd(const float* x, const uint8_t* mask, int64_t B, int64_t C, int64_t W, float* out,
                                        int64_t out_ld, void* stream)
{
    if (!x || !mask || !out || B < 0 || C < 0 || W < 0 || out_ld < W) return SGNN_ERR_BAD_ARG;
    if (B * W == 0) return SGNN_OK;
    if (ro_vec4_ok(x, out, W, out_ld))
        hipLaunchKernelGGL(masked_sum_slot_fwd_kernel<float4>, dim3(sgnn_grid_for(B * (W / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)x, mask, B, C, W / 4, (float4*)out, out_ld / 4);
    else
        hipLaunchKernelGGL(masked_sum_slot_fwd_kernel<float>, dim3(sgnn_grid_for(B * W, 256)), dim3(256), 0, (hipStream_t)stream, x, mask,
                           B, C, W, out, out_ld);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_masked_sum_slot_bwd(const float* grad_out, int64_t grad_ld, const uint8_t* mask, int64_t B, int64_t C, int64_t W,
                                        float* grad_x, void* stream)
{
    if (!grad_out || !mask || !grad_x || B < 0 || C < 0 || W < 0 || grad_ld < W) return SGNN_ERR_BAD_ARG;
    if (B * C * W == 0) return SGNN_OK;
    if (ro_vec4_ok(grad_out, grad_x, W, grad_ld))
        hipLaunchKernelGGL(masked_sum_slot_bwd_kernel<float4>, dim3(sgnn_grid_for(B * C * (W / 4), 256)), dim3(256), 0,
                           (hipStream_t)stream, (const float4*)grad_out, grad_ld / 4, mask, B, C, W / 4, (float4*)grad_x);
    else
        hipLaunchKernelGGL(masked_sum_slot_bwd_kernel<float>, dim3(sgnn_grid_for(B * C * W, 256)), dim3(256), 0, (hipStream_t)stream,
                           grad_out, grad_ld, mask, B, C, W, grad_x);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}
