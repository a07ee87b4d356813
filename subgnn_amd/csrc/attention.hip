// Additive-attention scores of the optional ff_attn read-out over a subgraph's components.
// Replaces attention.AdditiveAttention._forward_internal (reference SubGNN/attention.py:130-139,
// used at SubGNN/SubGNN.py:298-301):   score[r] = sum_j v_j * tanh( (q W)[b(r), j] + (X U)[r, j] ).
//
// The one dense contraction of the hot path, X (R, H) x U (H, H) with H = hid_dim (420-615).  Two forms:
// exact f32 (library GEMM + the fused epilogue below) and half operands on the matrix cores in one
// hand-written kernel (v_mfma_f32_32x32x16_f16 -- the gfx950 shape, twice the K per instruction of CDNA3's 32x32x8 --
// the epilogue applied to the accumulator fragment in registers: C/D map col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); A / B operand map: lane (r = lane & 31, h = lane >> 5) holds
// A[r][8 h + j] and B[8 h + j][r], j = 0..7).
#include "common.h"

typedef float sgnn_f32x16 __attribute__((ext_vector_type(16)));

// Exact (f32) form: the contraction X U is a plain dense GEMM and goes to the library (rocBLAS through the
// caller: 0.4 ms for 50k x 615 x 615, where a hand-written v_mfma_f32_32x32x2_f32 kernel with one wavefront per
// workgroup and the B operand fetched from global memory took 9.2 ms); what is fused here is everything
// after it -- + qW, tanh, x v, sum over the columns -- so that the (R, H) activation is read once and no
// intermediate of the epilogue is materialised.  One wavefront per row, lanes over the columns.
__global__ __launch_bounds__(256) void attn_epilogue_kernel(
    const float* __restrict__ XU, const float* __restrict__ cq, const float* __restrict__ v,
    int64_t R, int64_t H, int64_t rows_per_batch, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (int64_t)gridDim.x * 4) {
        const float* x = XU + r * H;
        const float* c = cq + (r / rows_per_batch) * H;
        float acc = 0.f;
        for (int64_t j = lane; j < H; j += 64) acc += v[j] * tanhf(x[j] + c[j]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) out[r] = acc;
    }
}

extern "C" int sgnn_attn_scores_epilogue(const float* XU, const float* qW, const float* v, int64_t R, int64_t H,
                                         int64_t rows_per_batch, float* out, void* stream)
{
    if (!XU || !qW || !v || !out || R < 0 || H <= 0 || rows_per_batch <= 0) return SGNN_ERR_BAD_ARG;
    if (R == 0) return SGNN_OK;
    hipLaunchKernelGGL(attn_epilogue_kernel, dim3(sgnn_grid_for(R, 4, 256 * 16)), dim3(256), 0, (hipStream_t)stream, XU, qW, v,
                       R, H, rows_per_batch, out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// The same scores with half operands on the matrix cores (BASELINE.json configs[4]: "fp16 embeddings with
// MFMA attention scores"; hparams['embedding_dtype'] = 'fp16' + ff_attn): v_mfma_f32_32x32x16_f16, fp32
// accumulate.  X and U stay fp32 in HBM (they are activations / a parameter); they are rounded to IEEE
// half on their way into registers / LDS.
//   * workgroup = 4 wavefronts = 4 row tiles of 32 (128 rows); a wavefront keeps ITS row tile as A
//     fragments in registers for the whole sweep (KS = ceil(H / 16) half8 per lane: 160 VGPRs at H = 640,
//     one wavefront per SIMD), loaded once;
//   * U is rounded to half and transposed ONCE per call (attn_u_half_kernel, into the caller's workspace); the H
//     columns are then swept in panels of 32: a panel is one contiguous block of that copy, staged per
//     workgroup into LDS with 16-byte copies (k contiguous: a B fragment is one 16-byte LDS read; row stride
//     = 8 x odd halves: 16-byte aligned, and the 16 lanes a ds_read_b128 serves per LDS cycle -- rows {0-3, 12-15,
//     20-27}, ... -- start on 16 different 4-bank groups: conflict-free), shared by the four wavefronts
//     and double-buffered through registers so that the fetch of the next panel runs under the MFMAs;
//   * epilogue on the accumulator fragment as in the f32 kernel (+ qW, tanh, x v, sum over the columns).
// H <= 16 * ATT_KS_MAX; larger hid_dims keep the f32 kernel.
// ---------------------------------------------------------------------------------------------
typedef _Float16 sgnn_f16x8 __attribute__((ext_vector_type(8)));
#define ATT_KS_MAX 40

// U (H, H) f32 -> Ut (Hp, ldk) half, TRANSPOSED (Ut[j][k] = U[k][j]) and zero padded to ldk = 16 KS + 8 columns and
// Hp = 32 ceil(H / 32) rows: a 32-column panel of U is then 32 consecutive rows of Ut, i.e. one contiguous block
// that goes into LDS with 16-byte copies, already in the layout the B fragments are read in.
__global__ __launch_bounds__(256) void attn_u_half_kernel(const float* __restrict__ U, int64_t H, int64_t Hp, int ldk,
                                                          _Float16* __restrict__ Ut)
{
    const int64_t total = Hp * ldk;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = t / ldk, k = t % ldk;
        Ut[t] = (_Float16)((j < H && k < H) ? U[k * H + j] : 0.f);
    }
}

template <int KS>
__global__ __launch_bounds__(256, 1) void attn_scores_f16_kernel(
    const float* __restrict__ X, const _Float16* __restrict__ Ut, const float* __restrict__ cq,
    const float* __restrict__ v, int64_t R, int64_t H, int64_t rows_per_batch, float* __restrict__ out)
{
    constexpr int LDK = 16 * KS + 8;                           // halves per row: 8 x odd -> 16-byte aligned fragments,
    constexpr int PANEL16 = 32 * LDK * 2 / 16;                 // conflict-free ds_read_b128; a panel in 16-byte units
    constexpr int PER_T = (PANEL16 + 255) / 256;
    extern __shared__ _Float16 s_u[];                          // 32 x LDK halves: the column panel, k contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    // A fragments of this wavefront's 32 rows: lane (i = lane & 31, kg = lane >> 5) holds X[r0 + i][16 ks + 8 kg .. + 7].
    // The four row tiles pass through the LDS buffer one after the other: the whole workgroup reads a tile with
    // coalesced loads (consecutive threads = consecutive columns of a row), rounds it to half into the same
    // k-contiguous layout the panels use, and the owning wavefront picks its fragments up with 16-byte reads.
    sgnn_f16x8 a[KS];
    for (int w = 0; w < 4; ++w) {
        const int64_t t0 = ((int64_t)blockIdx.x * 4 + w) * 32;
        __syncthreads();
        for (int64_t idx = tid; idx < 32 * (int64_t)(16 * KS); idx += 256) {
            const int64_t i = idx / (16 * KS), k = idx % (16 * KS);
            const float x = (t0 + i < R && k < H) ? X[(t0 + i) * H + k] : 0.f;
            s_u[i * LDK + k] = (_Float16)x;
        }
        __syncthreads();
        if (w == wave) {
            const _Float16* arow = s_u + l31 * LDK + 8 * half;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const sgnn_f16x8*>(arow + 16 * ks);
        }
    }
    float part[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) part[g] = 0.f;
    // panels: the next one is fetched into registers (16-byte loads of the pre-transposed half copy) while the matrix
    // cores work on the current one
    uint4 nxt[PER_T];
    const uint4* Ut16 = reinterpret_cast<const uint4*>(Ut);
#pragma unroll
    for (int u = 0; u < PER_T; ++u) { const int q = tid + 256 * u; nxt[u] = q < PANEL16 ? Ut16[q] : make_uint4(0, 0, 0, 0); }
    for (int64_t j0 = 0; j0 < H; j0 += 32) {
        __syncthreads();                                       // the previous panel (or the last row tile) is no longer read
        uint4* s16 = reinterpret_cast<uint4*>(s_u);
#pragma unroll
        for (int u = 0; u < PER_T; ++u) { const int q = tid + 256 * u; if (q < PANEL16) s16[q] = nxt[u]; }
        __syncthreads();
        if (j0 + 32 < H) {
            const uint4* src = Ut16 + ((j0 + 32) / 32) * (int64_t)PANEL16;
#pragma unroll
            for (int u = 0; u < PER_T; ++u) { const int q = tid + 256 * u; nxt[u] = q < PANEL16 ? src[q] : make_uint4(0, 0, 0, 0); }
        }
        sgnn_f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        const _Float16* bcol = s_u + l31 * LDK + 8 * half;      // B[k = 16 ks + 8 kg + q][j = lane & 31]
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const sgnn_f16x8 b = *reinterpret_cast<const sgnn_f16x8*>(bcol + 16 * ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b, acc, 0, 0, 0);
        }
        const int64_t j = j0 + l31;
        if (j < H) {
            const float vj = v[j];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int64_t r = r0 + (g & 3) + 8 * (g >> 2) + 4 * half;
                const float c = (r < R) ? cq[(r / rows_per_batch) * H + j] : 0.f;
                // tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp / rcp (relative error ~1e-6: far inside the
                // rounding of the half operands); tanhf's ~40-instruction expansion cost as much as the MFMAs
                const float e = __expf(2.f * (acc[g] + c));
                part[g] += vj * (1.f - 2.f * __frcp_rn(e + 1.f));
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) part[g] += __shfl_xor(part[g], d);
    }
    if (l31 == 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int64_t r = r0 + (g & 3) + 8 * (g >> 2) + 4 * half;
            if (r < R) out[r] = part[g];
        }
    }
}

static int attn_f16_ks(int64_t H) { const int ks = (int)((H + 15) / 16); return ks <= 8 ? 8 : (ks <= 28 ? 28 : ATT_KS_MAX); }

extern "C" int64_t sgnn_attn_scores_f16_workspace_bytes(int64_t H)
{
    if (H <= 0 || H > 16 * ATT_KS_MAX) return 0;
    const int64_t ldk = 16 * attn_f16_ks(H) + 8, Hp = (H + 31) / 32 * 32;
    return Hp * ldk * 2 + 64;
}

extern "C" int sgnn_attn_scores_fwd_f16(const float* X, const float* U, const float* qW, const float* v,
                                        int64_t R, int64_t H, int64_t rows_per_batch, float* out,
                                        void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!X || !U || !qW || !v || !out || R < 0 || H <= 0 || rows_per_batch <= 0) return SGNN_ERR_BAD_ARG;
    if (H > 16 * ATT_KS_MAX) return SGNN_ERR_UNSUPPORTED_D;
    if (!workspace || workspace_bytes < sgnn_attn_scores_f16_workspace_bytes(H)) return SGNN_ERR_BAD_ARG;
    if (R == 0) return SGNN_OK;
    const int KS = attn_f16_ks(H);
    const int ldk = 16 * KS + 8;
    const int64_t Hp = (H + 31) / 32 * 32;
    hipStream_t st = (hipStream_t)stream;
    _Float16* Ut = (_Float16*)workspace;
    hipLaunchKernelGGL(attn_u_half_kernel, dim3(sgnn_grid_for(Hp * ldk, 256)), dim3(256), 0, st, U, H, Hp, ldk, Ut);
    SGNN_CHECK_LAUNCH();
    const size_t lds = (size_t)32 * ldk * 2;
    const unsigned grid = (unsigned)((R + 127) / 128);
    if (KS == 8)
        hipLaunchKernelGGL(attn_scores_f16_kernel<8>, dim3(grid), dim3(256), lds, st, X, Ut, qW, v, R, H, rows_per_batch, out);
    else if (KS == 28)
        hipLaunchKernelGGL(attn_scores_f16_kernel<28>, dim3(grid), dim3(256), lds, st, X, Ut, qW, v, R, H, rows_per_batch, out);
    else
        hipLaunchKernelGGL(attn_scores_f16_kernel<ATT_KS_MAX>, dim3(grid), dim3(256), lds, st, X, Ut, qW, v, R, H, rows_per_batch, out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(attention)
