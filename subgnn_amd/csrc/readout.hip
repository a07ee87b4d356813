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

#define RO_ROWS_PER_BLOCK 64
#define RO_TA 64                       // column lanes of a workgroup (one wavefront reads 256 consecutive bytes of a row)
#define RO_TR 4                        // row lanes

// ------------------------------------------------------------------------------------------------------------------
// a thread owns four (subgraph, anchor) elements a quarter of the grid apart: their loads are independent and in flight
// together (one element per thread was latency-bound: 45 us for 50k x 183, a wavefront's life being one load round trip)
__global__ __launch_bounds__(256) void readout_sum_fwd_kernel(const float* __restrict__ sims, int64_t ld,
                                                              const int64_t* __restrict__ sim_col, const float* __restrict__ s,
                                                              const float* __restrict__ bp, const uint8_t* __restrict__ row_mask,
                                                              int64_t B, int32_t C, int32_t A, float* __restrict__ out, int64_t out_ld)
{
    const int64_t total = B * A, T = (int64_t)gridDim.x * 256;
    const int64_t t0 = blockIdx.x * 256ll + threadIdx.x;
    const float bb = bp[0];
    int64_t b[4], col[4];
    int32_t a[4];
    float sa[4], acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t t = t0 + j * T;
        const bool on = t < total;
        b[j] = on ? t / A : -1;
        a[j] = on ? (int32_t)(t - b[j] * A) : 0;
        sa[j] = s[a[j]];
        col[j] = sim_col ? sim_col[a[j]] : a[j];
        acc[j] = 0.f;
    }
    // four components at a time: 16 independent loads in flight (a batch of subgraphs with 7-50 components each walked them one
    // round trip after the other: the dense similarity slab is a cache miss per element)
    for (int32_t c0 = 0; c0 < C; c0 += 4) {
        float w[4][4];
        bool live[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = b[j] * C + c0 + u;
                live[u][j] = b[j] >= 0 && c0 + u < C && (!row_mask || row_mask[r]);
                w[u][j] = (live[u][j] && sims) ? sims[r * ld + col[j]] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live[u][j]) acc[j] += fmaxf(fmaf(w[u][j], sa[j], bb), 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (b[j] >= 0) out[b[j] * out_ld + a[j]] = acc[j];
}

// one workgroup per block of RO_ROWS_PER_BLOCK component rows, thread = (row lane tr, column lane ta), columns walked in
// chunks of RO_TA; a thread's rows are loaded four at a time (independent loads in flight: the loop is latency-bound
// otherwise -- 168 us for 50k x 183 with one dependent row per iteration).  d s: the row lanes' sums are added in lane
// order through LDS -> partial_s [A][nblk]; d bp: all of the workgroup's terms added in a fixed order -> partial_b [nblk].
__global__ __launch_bounds__(256) void readout_sum_bwd_partial_kernel(const float* __restrict__ g, int64_t g_ld,
                                                                      const float* __restrict__ sims, int64_t ld,
                                                                      const int64_t* __restrict__ sim_col, const float* __restrict__ s,
                                                                      const float* __restrict__ bp, const uint8_t* __restrict__ row_mask,
                                                                      int64_t R, int32_t C, int32_t A, int64_t nblk,
                                                                      float* __restrict__ partial_s, float* __restrict__ partial_b)
{
    __shared__ float sh_s[RO_TR][RO_TA];
    __shared__ float sh_b[256];
    const int ta = threadIdx.x % RO_TA, tr = threadIdx.x / RO_TA;
    const int64_t blk = blockIdx.x;
    const int64_t r0 = blk * RO_ROWS_PER_BLOCK;
    const int nrows = (int)((r0 + RO_ROWS_PER_BLOCK < R ? r0 + RO_ROWS_PER_BLOCK : R) - r0);
    const float bb = bp[0];
    constexpr int PER = RO_ROWS_PER_BLOCK / RO_TR;                     // rows per thread: tr, tr + TR, ...
    // the rows' mask bits and subgraph numbers do not depend on the column chunk
    uint32_t live = 0;
    int32_t sub[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tr + k * RO_TR;
        if (i < nrows && (!row_mask || row_mask[r0 + i])) live |= 1u << k;
        sub[k] = (int32_t)((uint32_t)(r0 + i) / (uint32_t)C);
    }
    float acc_b = 0.f;
    for (int32_t a0 = 0; a0 < A; a0 += RO_TA) {
        const int32_t a = a0 + ta;
        float acc_s = 0.f;
        if (a < A) {
            const float sa = s[a];
            const int64_t col = sim_col ? sim_col[a] : a;
#pragma unroll
            for (int k0 = 0; k0 < PER; k0 += 4) {
                float w[4], gz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t r = r0 + tr + (k0 + j) * RO_TR;
                    const bool on = (live >> (k0 + j)) & 1;
                    w[j] = (on && sims) ? sims[r * ld + col] : 0.f;
                    gz[j] = on ? g[(int64_t)sub[k0 + j] * g_ld + a] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((live >> (k0 + j)) & 1) && fmaf(w[j], sa, bb) > 0.f) {
                        acc_s = fmaf(gz[j], w[j], acc_s);
                        acc_b += gz[j];
                    }
            }
        }
        sh_s[tr][ta] = acc_s;
        __syncthreads();
        if (tr == 0 && a < A) {
            float vs = sh_s[0][ta];
#pragma unroll
            for (int k = 1; k < RO_TR; ++k) vs += sh_s[k][ta];
            partial_s[(int64_t)a * nblk + blk] = vs;
        }
        __syncthreads();
    }
    sh_b[threadIdx.x] = acc_b;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = (sh_b[threadIdx.x] + sh_b[threadIdx.x + 64]) + (sh_b[threadIdx.x + 128] + sh_b[threadIdx.x + 192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);          // fixed tree: the same order every run
        if (threadIdx.x == 0) partial_b[blk] = v;
    }
}

__device__ __forceinline__ float ro_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wavefront per column (4 per workgroup): a lane adds every 64th block partial, the lanes are added by a fixed
// butterfly.  The workgroup after the last column group does the same for d bp.
__global__ __launch_bounds__(256) void readout_sum_bwd_finish_kernel(const float* __restrict__ partial_s,
                                                                     const float* __restrict__ partial_b, int32_t A, int64_t nblk,
                                                                     float* __restrict__ grad_s, float* __restrict__ grad_bp)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t groups = (A + 3) / 4;
    const float* p;
    float* dst;
    if ((int32_t)blockIdx.x < groups) {
        const int32_t a = blockIdx.x * 4 + wave;
        if (a >= A || !grad_s) return;
        p = partial_s + (int64_t)a * nblk;
        dst = grad_s + a;
    } else {
        if (wave != 0 || !grad_bp) return;
        p = partial_b;
        dst = grad_bp;
    }
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;                      // four loads in flight; combined in a fixed order
    int64_t k = lane;
    for (; k + 192 < nblk; k += 256) { v0 += p[k]; v1 += p[k + 64]; v2 += p[k + 128]; v3 += p[k + 192]; }
    for (; k < nblk; k += 64) v0 += p[k];
    const float v = ro_wave_sum((v0 + v1) + (v2 + v3));
    if (lane == 0) *dst = v;
}

extern "C" int sgnn_readout_sum_fwd(const float* sims, int64_t sims_ld, const int64_t* sim_col, const float* s, const float* bp,
                                    const uint8_t* row_mask, int64_t B, int64_t C, int64_t A, float* out, int64_t out_ld,
                                    void* stream)
{
    if (!s || !bp || !out || B < 0 || C < 0 || A < 0 || out_ld < A || (sims && sims_ld < 1)) return SGNN_ERR_BAD_ARG;
    if (A > 0x7fffffff || C > 0x7fffffff || (B * A + 255) / 256 > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    if (B * A == 0) return SGNN_OK;
    hipLaunchKernelGGL(readout_sum_fwd_kernel, dim3((unsigned)((B * A + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, sims,
                       sims_ld, sim_col, s, bp, row_mask, B, (int32_t)C, (int32_t)A, out, out_ld);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

static inline int64_t ro_blocks(int64_t R) { return (R + RO_ROWS_PER_BLOCK - 1) / RO_ROWS_PER_BLOCK; }

extern "C" int64_t sgnn_readout_sum_bwd_workspace_bytes(int64_t B, int64_t C, int64_t A)
{
    if (B < 0 || C < 0 || A < 0) return -1;
    return (A + 1) * ro_blocks(B * C) * (int64_t)sizeof(float) + 16;
}

extern "C" int sgnn_readout_sum_bwd(const float* grad_out, int64_t grad_ld, const float* sims, int64_t sims_ld,
                                    const int64_t* sim_col, const float* s, const float* bp, const uint8_t* row_mask, int64_t B,
                                    int64_t C, int64_t A, float* grad_s, float* grad_bp, void* workspace, int64_t workspace_bytes,
                                    void* stream)
{
    if (!grad_out || !s || !bp || B < 0 || C < 0 || A < 0 || grad_ld < A || (sims && sims_ld < 1)) return SGNN_ERR_BAD_ARG;
    if (A > 0x7fffffff || C > 0x7fffffff || B * C > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    if (!grad_s && !grad_bp) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t R = B * C;
    if (R * A == 0) {
        if (grad_s && A) { if (hipMemsetAsync(grad_s, 0, A * sizeof(float), st) != hipSuccess) return SGNN_ERR_LAUNCH; }
        if (grad_bp) { if (hipMemsetAsync(grad_bp, 0, sizeof(float), st) != hipSuccess) return SGNN_ERR_LAUNCH; }
        return SGNN_OK;
    }
    if (!workspace || workspace_bytes < sgnn_readout_sum_bwd_workspace_bytes(B, C, A)) return SGNN_ERR_BAD_ARG;
    const int64_t nblk = ro_blocks(R);
    if (nblk > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    float* partial_s = (float*)workspace;
    float* partial_b = partial_s + A * nblk;
    hipLaunchKernelGGL(readout_sum_bwd_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, st, grad_out, grad_ld, sims, sims_ld,
                       sim_col, s, bp, row_mask, R, (int32_t)C, (int32_t)A, nblk, partial_s, partial_b);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(readout_sum_bwd_finish_kernel, dim3((unsigned)((A + 3) / 4 + 1)), dim3(256), 0, st, partial_s, partial_b,
                       (int32_t)A, nblk, grad_s, grad_bp);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---- every read-out piece of a step in one launch each way ------------------------------------------------------------------
// A pass has one read-out piece per channel side whose last layer runs over SHARED anchors or all-zero similarities (benchmark:
// P internal, P border, S internal, S border).  Each was: s = X wp (a matrix-vector launch), its PAD mask (three element-wise
// launches), the slot launch above; backward: partials + finish per piece, then the backward of s = X wp (two more products and
// two multiplies) -- 11 launches forward and 15 backward per pass.  Here: the scores of all pieces in one launch
// (readout_scores_kernel), all slots in one (blockIdx.y = piece), the backward partials in one, and one finish launch that
// also turns d s into d X = d s wp^T (the wavefront that owns column a writes row a) and, in the piece's last workgroup (a
// ticket), d wp = X^T d s and d bp.  Every sum keeps the fixed order of the single-piece kernels.
#define RO_MAX_PIECES 8
struct RoPieces {
    const float* sims[RO_MAX_PIECES]; long long ld[RO_MAX_PIECES]; const int64_t* sim_col[RO_MAX_PIECES];
    const float* X[RO_MAX_PIECES];            // (A, D) anchor embeddings; null: all scores zero (all-zero similarities)
    const float* wp[RO_MAX_PIECES]; const float* bp[RO_MAX_PIECES];
    const int64_t* ids[RO_MAX_PIECES];        // nullable (A): anchor id 0 = PAD -> score 0
    const uint8_t* row_mask[RO_MAX_PIECES];
    float* s[RO_MAX_PIECES];                  // (A) scores: written by the scores launch, read by the others
    int A[RO_MAX_PIECES]; int off[RO_MAX_PIECES];
    float* partial_s[RO_MAX_PIECES]; float* partial_b[RO_MAX_PIECES];
    float* gX[RO_MAX_PIECES]; float* gwp[RO_MAX_PIECES]; float* gbp[RO_MAX_PIECES]; float* gs[RO_MAX_PIECES];
    unsigned* ticket;                         // RO_MAX_PIECES counters, zero before the launch, left zero
    int D; int count;
};

__global__ __launch_bounds__(256) void readout_scores_kernel(const RoPieces P)
{
    const int p = blockIdx.x, A = P.A[p], D = P.D;
    const float* __restrict__ X = P.X[p];
    const float* __restrict__ wp = P.wp[p];
    const int64_t* __restrict__ ids = P.ids[p];
    for (int a = threadIdx.x; a < A; a += 256) {
        double v = 0.0;                          // (a few hundred dot products of D terms per step: accumulated in double, rounded once)
        if (X && !(ids && ids[a] == 0)) {
            const float* __restrict__ x = X + (int64_t)a * D;
            for (int d = 0; d < D; ++d) v = fma((double)x[d], (double)wp[d], v);
        }
        P.s[p][a] = (float)v;
    }
}

__global__ __launch_bounds__(256) void readout_sum_fwd_many_kernel(const RoPieces P, int64_t B, int32_t C, float* __restrict__ out,
                                                                   int64_t out_ld)
{
    const int p = blockIdx.y, A = P.A[p];
    const int64_t total = B * A;
    if ((int64_t)blockIdx.x * 1024 >= total) return;
    const int64_t T = ((total + 1023) / 1024) * 256;
    const int64_t t0 = blockIdx.x * 256ll + threadIdx.x;
    const float* __restrict__ sims = P.sims[p];
    const int64_t* __restrict__ sim_col = P.sim_col[p];
    const uint8_t* __restrict__ row_mask = P.row_mask[p];
    const float* __restrict__ s = P.s[p];
    const int64_t ld = P.ld[p];
    const float bb = P.bp[p][0];
    int64_t b[4], col[4];
    int32_t a[4];
    float sa[4], acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t t = t0 + j * T;
        const bool on = t < total;
        b[j] = on ? t / A : -1;
        a[j] = on ? (int32_t)(t - b[j] * A) : 0;
        sa[j] = s[a[j]];
        col[j] = sim_col ? sim_col[a[j]] : a[j];
        acc[j] = 0.f;
    }
    for (int32_t c0 = 0; c0 < C; c0 += 4) {
        float w[4][4];
        bool live[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = b[j] * C + c0 + u;
                live[u][j] = b[j] >= 0 && c0 + u < C && (!row_mask || row_mask[r]);
                w[u][j] = (live[u][j] && sims) ? sims[r * ld + col[j]] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live[u][j]) acc[j] += fmaxf(fmaf(w[u][j], sa[j], bb), 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (b[j] >= 0) out[b[j] * out_ld + P.off[p] + a[j]] = acc[j];
}

__global__ __launch_bounds__(256) void readout_sum_bwd_partial_many_kernel(const RoPieces P, const float* __restrict__ g, int64_t g_ld,
                                                                           int64_t R, int32_t C, int64_t nblk)
{
    __shared__ float sh_s[RO_TR][RO_TA];
    __shared__ float sh_b[256];
    const int p = blockIdx.y, A = P.A[p];
    const float* __restrict__ sims = P.sims[p];
    const int64_t* __restrict__ sim_col = P.sim_col[p];
    const uint8_t* __restrict__ row_mask = P.row_mask[p];
    const float* __restrict__ s = P.s[p];
    const int64_t ld = P.ld[p];
    g += P.off[p];
    float* __restrict__ partial_s = P.partial_s[p];
    float* __restrict__ partial_b = P.partial_b[p];
    const int ta = threadIdx.x % RO_TA, tr = threadIdx.x / RO_TA;
    const int64_t blk = blockIdx.x;
    const int64_t r0 = blk * RO_ROWS_PER_BLOCK;
    const int nrows = (int)((r0 + RO_ROWS_PER_BLOCK < R ? r0 + RO_ROWS_PER_BLOCK : R) - r0);
    const float bb = P.bp[p][0];
    constexpr int PER = RO_ROWS_PER_BLOCK / RO_TR;
    uint32_t live = 0;
    int32_t sub[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tr + k * RO_TR;
        if (i < nrows && (!row_mask || row_mask[r0 + i])) live |= 1u << k;
        sub[k] = (int32_t)((uint32_t)(r0 + i) / (uint32_t)C);
    }
    float acc_b = 0.f;
    for (int32_t a0 = 0; a0 < A; a0 += RO_TA) {
        const int32_t a = a0 + ta;
        float acc_s = 0.f;
        if (a < A) {
            const float sa = s[a];
            const int64_t col = sim_col ? sim_col[a] : a;
#pragma unroll
            for (int k0 = 0; k0 < PER; k0 += 4) {
                float w[4], gz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t r = r0 + tr + (k0 + j) * RO_TR;
                    const bool on = (live >> (k0 + j)) & 1;
                    w[j] = (on && sims) ? sims[r * ld + col] : 0.f;
                    gz[j] = on ? g[(int64_t)sub[k0 + j] * g_ld + a] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((live >> (k0 + j)) & 1) && fmaf(w[j], sa, bb) > 0.f) {
                        acc_s = fmaf(gz[j], w[j], acc_s);
                        acc_b += gz[j];
                    }
            }
        }
        sh_s[tr][ta] = acc_s;
        __syncthreads();
        if (tr == 0 && a < A) {
            float vs = sh_s[0][ta];
#pragma unroll
            for (int k = 1; k < RO_TR; ++k) vs += sh_s[k][ta];
            partial_s[(int64_t)a * nblk + blk] = vs;
        }
        __syncthreads();
    }
    sh_b[threadIdx.x] = acc_b;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = (sh_b[threadIdx.x] + sh_b[threadIdx.x + 64]) + (sh_b[threadIdx.x + 128] + sh_b[threadIdx.x + 192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) partial_b[blk] = v;
    }
}

// workgroups [0, groups) of a piece: one wavefront per column a -> d s[a] (block partials added as in readout_sum_bwd_finish_kernel)
// and row a of d X; workgroup `groups`: d bp; the LAST workgroup of the piece to finish: d wp[d] = sum_a d s[a] X[a, d], a in order.
__global__ __launch_bounds__(256) void readout_bwd_finish_many_kernel(const RoPieces P, int64_t nblk, int max_groups)
{
    __shared__ int s_last;
    const int p = blockIdx.y, A = P.A[p], D = P.D;
    const int groups = (A + 3) / 4;
    if ((int)blockIdx.x > groups) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* src = nullptr;
    int a = -1;
    if ((int)blockIdx.x < groups) {
        a = blockIdx.x * 4 + wave;
        if (a < A) src = P.partial_s[p] + (int64_t)a * nblk; else a = -1;
    } else if (wave == 0) src = P.partial_b[p];
    if (src) {
        // (the block partials of a column are added in double: a column of 50k x C rows is ~800 partials whose sum is a gradient
        // the LSTM's weight gradients are contracted from -- rounded once here instead of once per partial)
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        int64_t k = lane;
        for (; k + 192 < nblk; k += 256) { v0 += src[k]; v1 += src[k + 64]; v2 += src[k + 128]; v3 += src[k + 192]; }
        for (; k < nblk; k += 64) v0 += src[k];
        double vd = (v0 + v1) + (v2 + v3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vd += __shfl_xor(vd, o, 64);
        const float v = (float)vd;
        if (a >= 0) {
            const bool pad = P.ids[p] && P.ids[p][a] == 0;
            const float gs = pad ? 0.f : v;
            if (lane == 0 && P.gs[p]) P.gs[p][a] = gs;
            if (P.gX[p] && P.wp[p])
                for (int d = lane; d < D; d += 64) P.gX[p][(int64_t)a * D + d] = gs * P.wp[p][d];
        } else if (lane == 0 && P.gbp[p]) P.gbp[p][0] = v;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(P.ticket + p, 1u) == (unsigned)groups) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (P.gwp[p] && P.X[p] && P.gs[p]) {
        // d s staged in LDS (its volatile reads were one dependent round trip per anchor: 70 us for 183 anchors), anchors in chunks
        __shared__ float s_gs[1024];
        __shared__ float s_part[256];
        const volatile float* gs = P.gs[p];
        const float* __restrict__ X = P.X[p];
        if (D <= 128 && 256 % D == 0) {
            // the anchors in 256 / D contiguous ranges, one per group of D threads; the ranges' sums added in range order
            const int nq = 256 / D, q = threadIdx.x / D, d = threadIdx.x - q * D;
            double v = 0.0;
            for (int a0 = 0; a0 < A; a0 += 1024) {
                const int na = A - a0 < 1024 ? A - a0 : 1024;
                __syncthreads();
                for (int aa = threadIdx.x; aa < na; aa += 256) s_gs[aa] = gs[a0 + aa];
                __syncthreads();
                const int per = (na + nq - 1) / nq, lo = q * per, hi = lo + per < na ? lo + per : na;
#pragma unroll 8
                for (int aa = lo; aa < hi; ++aa) v = fma((double)s_gs[aa], (double)X[(int64_t)(a0 + aa) * D + d], v);
            }
            s_part[threadIdx.x] = (float)v;
            __syncthreads();
            if (q == 0) {
                float sum = s_part[d];
                for (int k = 1; k < nq; ++k) sum += s_part[k * D + d];
                P.gwp[p][d] = sum;
            }
        } else {
            float v[4] = {0.f, 0.f, 0.f, 0.f};                       // columns d, d + 256, ... of this thread (D <= 1024 here)
            for (int a0 = 0; a0 < A; a0 += 1024) {
                const int na = A - a0 < 1024 ? A - a0 : 1024;
                __syncthreads();
                for (int aa = threadIdx.x; aa < na; aa += 256) s_gs[aa] = gs[a0 + aa];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int d = threadIdx.x + 256 * q;
                    if (d < D) {
#pragma unroll 8
                        for (int aa = 0; aa < na; ++aa) v[q] = fmaf(s_gs[aa], X[(int64_t)(a0 + aa) * D + d], v[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int d = threadIdx.x + 256 * q; if (d < D) P.gwp[p][d] = v[q]; }
        }
    }
    if (threadIdx.x == 0) P.ticket[p] = 0u;
}

// host arrays of DEVICE pointers, one entry per piece (n <= sgnn_readout_many_max()); D: width of the anchor embeddings
extern "C" int64_t sgnn_readout_many_max(void) { return RO_MAX_PIECES; }

static int ro_fill_pieces(RoPieces& P, int64_t n, const float* const* sims, const int64_t* sims_ld, const int64_t* const* sim_col,
                          const float* const* X, const float* const* wp, const float* const* bp, const int64_t* const* ids,
                          const uint8_t* const* row_mask, float* const* s, const int64_t* A, const int64_t* off, int64_t D, int64_t H)
{
    if (n < 1 || n > RO_MAX_PIECES || !sims || !sims_ld || !sim_col || !X || !wp || !bp || !ids || !row_mask || !s || !A || !off) return -1;
    if (D < 1 || D > 1024) return -1;
    P.count = (int)n; P.D = (int)D; P.ticket = nullptr;
    for (int k = 0; k < n; ++k) {
        if (!bp[k] || !s[k] || A[k] < 1 || A[k] > (1 << 24) || off[k] < 0 || off[k] + A[k] > H) return -1;
        if (sims[k] && sims_ld[k] < 1) return -1;
        if (X[k] && !wp[k]) return -1;
        P.sims[k] = sims[k]; P.ld[k] = sims_ld[k]; P.sim_col[k] = sim_col[k]; P.X[k] = X[k]; P.wp[k] = wp[k]; P.bp[k] = bp[k];
        P.ids[k] = ids[k]; P.row_mask[k] = row_mask[k]; P.s[k] = s[k]; P.A[k] = (int)A[k]; P.off[k] = (int)off[k];
        P.partial_s[k] = P.partial_b[k] = P.gX[k] = P.gwp[k] = P.gbp[k] = P.gs[k] = nullptr;
    }
    for (int k = (int)n; k < RO_MAX_PIECES; ++k) {
        P.sims[k] = P.X[k] = P.wp[k] = P.bp[k] = nullptr; P.sim_col[k] = P.ids[k] = nullptr; P.row_mask[k] = nullptr; P.s[k] = nullptr;
        P.ld[k] = 0; P.A[k] = 0; P.off[k] = 0;
        P.partial_s[k] = P.partial_b[k] = P.gX[k] = P.gwp[k] = P.gbp[k] = P.gs[k] = nullptr;
    }
    return 0;
}

extern "C" int sgnn_readout_many_fwd(int64_t n, const float* const* sims, const int64_t* sims_ld, const int64_t* const* sim_col,
                                     const float* const* X, const float* const* wp, const float* const* bp,
                                     const int64_t* const* ids, const uint8_t* const* row_mask, float* const* s, const int64_t* A,
                                     const int64_t* off, int64_t D, int64_t B, int64_t C, float* out, int64_t out_ld, void* stream)
{
    if (!out || B < 0 || C < 0 || C > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    RoPieces P;
    if (ro_fill_pieces(P, n, sims, sims_ld, sim_col, X, wp, bp, ids, row_mask, s, A, off, D, out_ld) != 0) return SGNN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(readout_scores_kernel, dim3((unsigned)n), dim3(256), 0, st, P);
    SGNN_CHECK_LAUNCH();
    int64_t most = 0;
    for (int k = 0; k < n; ++k) most = B * A[k] > most ? B * A[k] : most;
    if (most == 0) return SGNN_OK;
    if ((most + 1023) / 1024 > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(readout_sum_fwd_many_kernel, dim3((unsigned)((most + 1023) / 1024), (unsigned)n), dim3(256), 0, st, P, B,
                       (int32_t)C, out, out_ld);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

/* workspace floats per piece: (A + 1) * blocks + A (d s); tickets: RO_MAX_PIECES uint32, zero before the first call */
extern "C" int64_t sgnn_readout_many_bwd_workspace_bytes(int64_t n, const int64_t* A, int64_t B, int64_t C)
{
    if (n < 1 || n > RO_MAX_PIECES || !A || B < 0 || C < 0) return -1;
    int64_t fl = 0;
    for (int k = 0; k < n; ++k) fl += (A[k] + 1) * ro_blocks(B * C) + A[k];
    return fl * 4 + 64;
}

extern "C" int sgnn_readout_many_bwd(int64_t n, const float* grad_out, int64_t grad_ld, const float* const* sims, const int64_t* sims_ld,
                                     const int64_t* const* sim_col, const float* const* X, const float* const* wp,
                                     const float* const* bp, const int64_t* const* ids, const uint8_t* const* row_mask,
                                     float* const* s, const int64_t* A, const int64_t* off, int64_t D, int64_t B, int64_t C,
                                     float* const* grad_X, float* const* grad_wp, float* const* grad_bp, void* workspace,
                                     int64_t workspace_bytes, unsigned* tickets, void* stream)
{
    if (!grad_out || !grad_X || !grad_wp || !grad_bp || !tickets || B < 1 || C < 1 || C > 0x7fffffff || B * C > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    RoPieces P;
    if (ro_fill_pieces(P, n, sims, sims_ld, sim_col, X, wp, bp, ids, row_mask, s, A, off, D, grad_ld) != 0) return SGNN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < sgnn_readout_many_bwd_workspace_bytes(n, A, B, C)) return SGNN_ERR_BAD_ARG;
    const int64_t nblk = ro_blocks(B * C);
    float* w = (float*)workspace;
    int max_groups = 0;
    for (int k = 0; k < n; ++k) {
        P.partial_s[k] = w; w += A[k] * nblk;
        P.partial_b[k] = w; w += nblk;
        P.gs[k] = w; w += A[k];
        P.gX[k] = grad_X[k]; P.gwp[k] = grad_wp[k]; P.gbp[k] = grad_bp[k];
        const int gr = (int)((A[k] + 3) / 4);
        max_groups = gr > max_groups ? gr : max_groups;
    }
    P.ticket = tickets;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(readout_sum_bwd_partial_many_kernel, dim3((unsigned)nblk, (unsigned)n), dim3(256), 0, st, P, grad_out, grad_ld,
                       B * C, (int32_t)C, nblk);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(readout_bwd_finish_many_kernel, dim3((unsigned)(max_groups + 1), (unsigned)n), dim3(256), 0, st, P, nblk, max_groups);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Column sums of a tall matrix x (R, A) -- the bias gradient of a Linear over R rows (autograd of SubGNN/SubGNN.py:304-312 and
// of the read-out weights): per-256-row-block partials with four loads in flight per thread, then the block partials of a
// column added in block order by one wavefront (readout_sum_bwd_finish_kernel).  torch's reduction over the leading
// dimension of a tall, narrow matrix took three launches (a row-block view, two sums) + an add for the tail.
// rows per block: 256 for a very tall matrix, fewer when that would leave the chip short of workgroups (>= ~1000 blocks)
static inline int64_t cs_rows_per_block(int64_t R) {
    int64_t rpb = (R + 1023) / 1024;
    rpb = (rpb + RO_TR - 1) / RO_TR * RO_TR;
    return rpb < 16 ? 16 : (rpb > 256 ? 256 : rpb);
}
__global__ __launch_bounds__(256) void column_sum_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int32_t A,
                                                                 int64_t nblk, int64_t rpb, float* __restrict__ partial)
{
    __shared__ float sh[RO_TR][RO_TA];
    const int ta = threadIdx.x % RO_TA, tr = threadIdx.x / RO_TA;
    const int64_t blk = blockIdx.x;
    const int64_t r0 = blk * rpb;
    const int64_t r1 = r0 + rpb < R ? r0 + rpb : R;
    for (int32_t a0 = 0; a0 < A; a0 += RO_TA) {
        const int32_t a = a0 + ta;
        float acc = 0.f;
        if (a < A) {
            int64_t r = r0 + tr;
            for (; r + 3 * RO_TR < r1; r += 4 * RO_TR) {
                const float v0 = x[r * ld + a], v1 = x[(r + RO_TR) * ld + a], v2 = x[(r + 2 * RO_TR) * ld + a],
                            v3 = x[(r + 3 * RO_TR) * ld + a];
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; r < r1; r += RO_TR) acc += x[r * ld + a];
        }
        sh[tr][ta] = acc;
        __syncthreads();
        if (tr == 0 && a < A) {
            float v = sh[0][ta];
#pragma unroll
            for (int k = 1; k < RO_TR; ++k) v += sh[k][ta];
            partial[(int64_t)a * nblk + blk] = v;
        }
        __syncthreads();
    }
}

extern "C" int64_t sgnn_column_sum_workspace_bytes(int64_t R, int64_t A)
{
    if (R < 0 || A < 0) return -1;
    const int64_t rpb = cs_rows_per_block(R);
    return A * ((R + rpb - 1) / rpb) * (int64_t)sizeof(float) + 16;
}

extern "C" int sgnn_column_sum(const float* x, int64_t ld, int64_t R, int64_t A, float* out, void* workspace, int64_t workspace_bytes,
                               void* stream)
{
    if (!x || !out || R < 0 || A < 0 || ld < A || A > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (A == 0) return SGNN_OK;
    if (R == 0) return hipMemsetAsync(out, 0, A * sizeof(float), st) == hipSuccess ? SGNN_OK : SGNN_ERR_LAUNCH;
    if (!workspace || workspace_bytes < sgnn_column_sum_workspace_bytes(R, A)) return SGNN_ERR_BAD_ARG;
    const int64_t rpb = cs_rows_per_block(R);
    const int64_t nblk = (R + rpb - 1) / rpb;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(column_sum_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, st, x, ld, R, (int32_t)A, nblk, rpb, partial);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(readout_sum_bwd_finish_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, st, partial, (const float*)nullptr,
                       (int32_t)A, nblk, out, (float*)nullptr);
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
        // (a select between a loaded float4 and a zero one went through a 32-byte scratch slot: the value is loaded or zeroed
        // element by element instead)
        V v;
        if constexpr (sizeof(V) == 16) v = make_float4(0.f, 0.f, 0.f, 0.f); else v = 0.f;
        if (mask[bc]) v = g[(bc / C) * g_ldv + h];
        gx[t] = v;
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

// ---- every tensor piece of the embedding in one launch -----------------------------------------------------------------------
// A batch-sized step of a 4-layer model sums 22 pieces of 64 x C x (34..128) floats: 22 launches of 2.5 us each way.  The pieces
// travel as kernel arguments; a work item is one (subgraph, column of the concatenation of the listed pieces).
#define RO_MAX_SLOTS 96
struct RoSlots {
    const float* x[RO_MAX_SLOTS];         // fwd: piece (B, C, w);   bwd: gradient out (B, C, w), may be null (not wanted)
    int w[RO_MAX_SLOTS];
    int off[RO_MAX_SLOTS];                // the piece's first column in the (B, H) embedding
    int col0[RO_MAX_SLOTS + 1];           // its first column in the concatenation of the LISTED pieces
    int count;
};

__device__ __forceinline__ int ro_find(const RoSlots& S, int j)
{
    int lo = 0, hi = S.count - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (S.col0[mid] <= j) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ __launch_bounds__(256) void masked_sum_slots_fwd_kernel(const RoSlots S, const uint8_t* __restrict__ mask, int64_t B, int64_t C,
                                                                   float* __restrict__ out, int64_t H)
{
    const int64_t W = S.col0[S.count], total = B * W;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / W;
        const int j = (int)(t - b * W), k = ro_find(S, j), jj = j - S.col0[k], w = S.w[k];
        const float* __restrict__ x = S.x[k];
        float acc = 0.f;
        for (int64_t c = 0; c < C; ++c)
            if (mask[b * C + c]) acc += x[(b * C + c) * w + jj];
        out[b * H + S.off[k] + jj] = acc;
    }
}

__global__ __launch_bounds__(256) void masked_sum_slots_bwd_kernel(const RoSlots S, const float* __restrict__ g, int64_t H,
                                                                   const uint8_t* __restrict__ mask, int64_t B, int64_t C)
{
    const int64_t W = S.col0[S.count], total = B * C * W;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bc = t / W;
        const int j = (int)(t - bc * W), k = ro_find(S, j), jj = j - S.col0[k];
        float* __restrict__ gx = const_cast<float*>(S.x[k]);
        if (gx) gx[bc * S.w[k] + jj] = mask[bc] ? g[(bc / C) * H + S.off[k] + jj] : 0.f;
    }
}

static int ro_fill(RoSlots& S, const float* const* ptrs, const int64_t* widths, const int64_t* offsets, int64_t from, int64_t to,
                   int64_t H, bool need_ptr)
{
    int col = 0;
    S.count = (int)(to - from);
    for (int64_t i = from; i < to; ++i) {
        const int k = (int)(i - from);
        if (widths[i] < 0 || offsets[i] < 0 || offsets[i] + widths[i] > H || widths[i] > (1 << 24)) return -1;
        if (need_ptr && widths[i] > 0 && !ptrs[i]) return -1;
        S.x[k] = ptrs[i]; S.w[k] = (int)widths[i]; S.off[k] = (int)offsets[i]; S.col0[k] = col;
        col += (int)widths[i];
    }
    S.col0[S.count] = col;
    return col;
}

extern "C" int sgnn_masked_sum_slots_fwd(const float* const* xs, const int64_t* widths, const int64_t* offsets, int64_t n_pieces,
                                         const uint8_t* mask, int64_t B, int64_t C, float* out, int64_t out_ld, void* stream)
{
    if (n_pieces < 0 || B < 0 || C < 0 || (n_pieces && (!xs || !widths || !offsets || !mask || !out))) return SGNN_ERR_BAD_ARG;
    for (int64_t from = 0; from < n_pieces; from += RO_MAX_SLOTS) {
        const int64_t to = from + RO_MAX_SLOTS < n_pieces ? from + RO_MAX_SLOTS : n_pieces;
        RoSlots S;
        const int W = ro_fill(S, xs, widths, offsets, from, to, out_ld, true);
        if (W < 0) return SGNN_ERR_BAD_ARG;
        if (B * W == 0) continue;
        hipLaunchKernelGGL(masked_sum_slots_fwd_kernel, dim3(sgnn_grid_for(B * W, 256)), dim3(256), 0, (hipStream_t)stream, S, mask, B, C,
                           out, out_ld);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

extern "C" int sgnn_masked_sum_slots_bwd(const float* grad_out, int64_t grad_ld, const uint8_t* mask, int64_t B, int64_t C,
                                         float* const* grad_xs, const int64_t* widths, const int64_t* offsets, int64_t n_pieces,
                                         void* stream)
{
    if (n_pieces < 0 || B < 0 || C < 0 || (n_pieces && (!grad_xs || !widths || !offsets || !mask || !grad_out))) return SGNN_ERR_BAD_ARG;
    for (int64_t from = 0; from < n_pieces; from += RO_MAX_SLOTS) {
        const int64_t to = from + RO_MAX_SLOTS < n_pieces ? from + RO_MAX_SLOTS : n_pieces;
        RoSlots S;
        const int W = ro_fill(S, (const float* const*)grad_xs, widths, offsets, from, to, grad_ld, false);
        if (W < 0) return SGNN_ERR_BAD_ARG;
        if (B * C * W == 0) continue;
        hipLaunchKernelGGL(masked_sum_slots_bwd_kernel, dim3(sgnn_grid_for(B * C * W, 256)), dim3(256), 0, (hipStream_t)stream, S, grad_out,
                           grad_ld, mask, B, C);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

SGNN_DEFINE_WARM(readout)
