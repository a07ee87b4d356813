// Additive-attention scores of the optional ff_attn read-out over a subgraph's components.
// Replaces attention.AdditiveAttention._forward_internal (reference SubGNN/attention.py:130-139,
// used at SubGNN/SubGNN.py:298-301):   score[r] = sum_j v_j * tanh( (q W)[b(r), j] + (X U)[r, j] ).
//
// The one dense contraction of the hot path, X (R, H) x U (H, H) with H = hid_dim (420-615): it
// runs on the matrix cores.  One wavefront owns a 32-row tile of X (staged once in LDS with an odd
// row stride, so the 32 rows of a fragment column hit 32 different banks) and sweeps the H columns
// in 32-wide tiles with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain,
// so no precision is traded for the matrix pipe).  The epilogue -- + qW, tanh, x v_j, sum over
// columns -- is applied to the accumulator fragment in registers (C/D map: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), so the (R, H) intermediate never exists.
#include "common.h"

typedef float sgnn_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void attn_scores_kernel(
    const float* __restrict__ X, const float* __restrict__ U, const float* __restrict__ cq,
    const float* __restrict__ v, int64_t R, int64_t H, int64_t rows_per_batch, float* __restrict__ out)
{
    extern __shared__ float s_x[];
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int64_t ldx = H | 1;                                  // odd stride: conflict-free column reads
    const int64_t n_tiles = (R + 31) / 32;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * 32;
        for (int64_t idx = lane; idx < 32 * H; idx += 64) {     // coalesced staging of the row tile
            const int64_t row = idx / H, k = idx % H;
            s_x[row * ldx + k] = (r0 + row < R) ? X[(r0 + row) * H + k] : 0.f;
        }
        __syncthreads();
        float part[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) part[g] = 0.f;
        for (int64_t j0 = 0; j0 < H; j0 += 32) {
            sgnn_f32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.f;
            const int64_t j = j0 + l31;
            for (int64_t k = 0; k < H; k += 2) {
                const int64_t kk = k + half;                    // A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
                const float a = (kk < H) ? s_x[l31 * ldx + kk] : 0.f;
                const float b = (kk < H && j < H) ? U[kk * H + j] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (j < H) {
                const float vj = v[j];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int64_t r = r0 + (g & 3) + 8 * (g >> 2) + 4 * half;
                    const float c = (r < R) ? cq[(r / rows_per_batch) * H + j] : 0.f;
                    part[g] += vj * tanhf(acc[g] + c);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) part[g] += __shfl_xor(part[g], d);   // over the 32 columns of a half
        }
        if (l31 == 0) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int64_t r = r0 + (g & 3) + 8 * (g >> 2) + 4 * half;
                if (r < R) out[r] = part[g];
            }
        }
        __syncthreads();
    }
}

extern "C" int sgnn_attn_scores_fwd(const float* X, const float* U, const float* qW, const float* v,
                                    int64_t R, int64_t H, int64_t rows_per_batch, float* out, void* stream)
{
    if (!X || !U || !qW || !v || !out || R < 0 || H <= 0 || rows_per_batch <= 0) return SGNN_ERR_BAD_ARG;
    const size_t lds = (size_t)(32 * (H | 1) * 4);
    if (lds > 152 * 1024) return SGNN_ERR_UNSUPPORTED_D;
    if (R == 0) return SGNN_OK;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)attn_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        attr_set = true;
    }
    const int64_t n_tiles = (R + 31) / 32;
    const int grid = (int)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
    hipLaunchKernelGGL(attn_scores_kernel, dim3(grid), dim3(64), lds, (hipStream_t)stream, X, U, qW, v, R, H,
                       rows_per_batch, out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}
