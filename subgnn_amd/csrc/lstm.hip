// One bidirectional LSTM layer over short sequences, forward and backward, one launch each.
// Replaces the nn.LSTM(bidirectional=True) of the structure channel's walk aggregator (reference
// SubGNN/SubGNN.py:60-88, called from SubGNN/anchor_patch_samplers.py:413-433 on (patches x walks,
// walk_len, D) inputs: a few hundred sequences of 10-20 steps, hidden size D).  The vendor library
// runs such a layer as one small GEMM plus one point-wise launch per time step and direction --
// ~80 launches forward and ~80 backward, each a few microseconds of launch latency around
// nanoseconds of arithmetic; here the recurrence stays inside one kernel.
//
// Layout.  A workgroup owns a tile of BT sequences of one direction and has 4H lanes: lane j
// holds row j of [W_ih | W_hh] (gate order i, f, g, o as in torch) in registers for the whole
// sequence.  Per step: [x_t | h_{t-1}] of the tile is staged in LDS, every lane forms its gate
// pre-activation for the BT sequences (broadcast LDS reads, K = I + H fma per sequence), the
// pre-activations cross to the lanes that own (sequence, hidden unit) pairs through LDS, and those
// apply the non-linearities and keep c in registers.  Gates and cell states are saved for the
// backward pass, which walks the steps in reverse: gate gradients in the owner lanes, then dx and
// the recurrent dh as a (BT x 4H) x (4H x K) product whose weight columns sit in registers (lane
// (k, half) holds half of column k).  The forward pass also leaves [x_t | h_{t-1}] in memory and the
// backward pass the gate gradients: the weight gradient is their plain product over all
// (sequence, step) rows, one library GEMM for the caller (SubGNN's batches are a few hundred
// sequences: a per-tile accumulation with atomics was 3x the cost of the recurrence itself).
// The tile height BT (2, 4 or 8 sequences) is chosen by the host so that a small batch still
// spreads over the chip: the recurrence is a latency chain, a workgroup is one wavefront per SIMD.
#include "common.h"

__device__ __forceinline__ float lstm_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

template <int H, int I, int BT>
__global__ __launch_bounds__(4 * H) void lstm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wcat, const float* __restrict__ bias, int64_t B, int64_t T,
    float* __restrict__ y, float* __restrict__ gates, float* __restrict__ cst, float* __restrict__ xh)
{
    constexpr int G = 4 * H, K = I + H;
    constexpr int OWN = (BT * H + G - 1) / G;              // (sequence, unit) pairs per owner lane
    __shared__ __attribute__((aligned(16))) float s_in[BT][K];            // [x_t | h_{t-1}]
    __shared__ float s_pre[BT][G];
    const int j = threadIdx.x, d = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * BT;
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = wcat[((int64_t)d * G + j) * K + k];
    const float bj = bias[d * G + j];
    float c_own[OWN];
#pragma unroll
    for (int q = 0; q < OWN; ++q) c_own[q] = 0.f;
    for (int idx = j; idx < BT * H; idx += G) s_in[idx / H][I + idx % H] = 0.f;
    for (int64_t step = 0; step < T; ++step) {
        const int64_t t = d ? T - 1 - step : step;
        for (int idx = j; idx < BT * I; idx += G) {
            const int b = idx / I, k = idx % I;
            s_in[b][k] = (b0 + b < B) ? x[((b0 + b) * T + t) * I + k] : 0.f;
        }
        __syncthreads();
        for (int idx = j; idx < BT * K; idx += G) {         // kept for the weight gradient
            const int b = idx / K, k = idx % K;
            if (b0 + b < B) xh[(((int64_t)d * B + b0 + b) * T + t) * K + k] = s_in[b][k];
        }
        float acc[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b] = bj;
#pragma unroll
        for (int k = 0; k < K; k += 4) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(&s_in[b][k]);
                acc[b] = fmaf(w[k], v.x, acc[b]);
                acc[b] = fmaf(w[k + 1], v.y, acc[b]);
                acc[b] = fmaf(w[k + 2], v.z, acc[b]);
                acc[b] = fmaf(w[k + 3], v.w, acc[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) s_pre[b][j] = acc[b];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int p = j + q * G;
            if (p < BT * H) {
                const int b = p / H, u = p % H;
                const float gi = lstm_sigmoid(s_pre[b][u]), gf = lstm_sigmoid(s_pre[b][H + u]);
                const float gg = tanhf(s_pre[b][2 * H + u]), go = lstm_sigmoid(s_pre[b][3 * H + u]);
                const float c = gf * c_own[q] + gi * gg;
                const float h = go * tanhf(c);
                c_own[q] = c;
                s_in[b][I + u] = h;                          // h_{t-1} of the next step (the x part is rewritten above)
                if (b0 + b < B) {
                    float* gp = gates + (((int64_t)d * B + b0 + b) * T + t) * G;
                    gp[u] = gi; gp[H + u] = gf; gp[2 * H + u] = gg; gp[3 * H + u] = go;
                    cst[(((int64_t)d * B + b0 + b) * T + t) * H + u] = c;
                    y[((b0 + b) * T + t) * (2 * H) + d * H + u] = h;
                }
            }
        }
        // the next step's staging writes s_in[.][0..I) only, its dot products start after the barrier
        // that follows -- by then every owner lane has stored its h
    }
}

template <int H, int I, int BT>
__global__ __launch_bounds__(4 * H) void lstm_bwd_kernel(
    const float* __restrict__ wcat, const float* __restrict__ gates, const float* __restrict__ cst,
    const float* __restrict__ dy, int64_t B, int64_t T, float* __restrict__ dx, float* __restrict__ dpre)
{
    constexpr int G = 4 * H, K = I + H;
    constexpr int OWN = (BT * H + G - 1) / G;
    constexpr int NJ = (G % K == 0) ? G / K : 1;           // lanes per output column of the dx/dh product
    constexpr int ROWS = G / NJ;                            // gate rows one lane sums over
    constexpr bool REGW = ROWS <= 128;                      // its share of the weight column fits in registers
    __shared__ __attribute__((aligned(16))) float s_dpre_t[G][BT];        // gate gradients [row][sequence]: broadcast reads over rows
    __shared__ float s_dh[BT][H];                                          // recurrent dh from the step after
    __shared__ float s_part[BT][K];
    const int j = threadIdx.x, d = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * BT;
    const int kcol = j % K, part = j / K;
    const float* wp = wcat + (int64_t)d * G * K + kcol + (int64_t)(part < NJ ? part : 0) * ROWS * K;
    float wreg[REGW ? ROWS : 1];
    if (REGW) {
#pragma unroll
        for (int r = 0; r < (REGW ? ROWS : 1); ++r) wreg[r] = wp[(int64_t)r * K];
    }
    float dc_own[OWN];
#pragma unroll
    for (int q = 0; q < OWN; ++q) dc_own[q] = 0.f;
    for (int idx = j; idx < BT * H; idx += G) s_dh[idx / H][idx % H] = 0.f;
    for (int64_t step = 0; step < T; ++step) {
        const int64_t sf = T - 1 - step;                    // forward step being undone
        const int64_t t = d ? T - 1 - sf : sf;
        const int64_t tp = d ? t + 1 : t - 1;               // time index of the forward step before it
        __syncthreads();                                    // s_dh / s_part / s_dpre_t of the previous step are done with
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int p = j + q * G;
            if (p < BT * H) {
                const int b = p / H, u = p % H;
                float pi = 0.f, pf = 0.f, pg = 0.f, po = 0.f;
                if (b0 + b < B) {
                    const int64_t row = ((int64_t)d * B + b0 + b) * T + t;
                    const float* gp = gates + row * G;
                    const float gi = gp[u], gf = gp[H + u], gg = gp[2 * H + u], go = gp[3 * H + u];
                    const float c = cst[row * H + u];
                    const float cprev = sf > 0 ? cst[(((int64_t)d * B + b0 + b) * T + tp) * H + u] : 0.f;
                    const float dh = dy[((b0 + b) * T + t) * (2 * H) + d * H + u] + s_dh[b][u];
                    const float tc = tanhf(c);
                    const float dc = dh * go * (1.f - tc * tc) + dc_own[q];
                    dc_own[q] = dc * gf;
                    pi = dc * gg * gi * (1.f - gi);
                    pf = dc * cprev * gf * (1.f - gf);
                    pg = dc * gi * (1.f - gg * gg);
                    po = dh * tc * go * (1.f - go);
                    float* dp = dpre + row * G;
                    dp[u] = pi; dp[H + u] = pf; dp[2 * H + u] = pg; dp[3 * H + u] = po;
                }
                s_dpre_t[u][b] = pi; s_dpre_t[H + u][b] = pf; s_dpre_t[2 * H + u][b] = pg; s_dpre_t[3 * H + u][b] = po;
            }
        }
        __syncthreads();
        // [dx_t | dh_{t-1}][b, k] = sum_j dpre[b, j] * W[j, k]: lane (k, part) sums its share of the rows
        {
            float acc[BT];
#pragma unroll
            for (int b = 0; b < BT; ++b) acc[b] = 0.f;
            if (part < NJ) {
                const int j0 = part * ROWS;
                if (REGW) {
#pragma unroll
                    for (int r = 0; r < (REGW ? ROWS : 1); ++r) {
#pragma unroll
                        for (int b = 0; b < BT; ++b) acc[b] = fmaf(s_dpre_t[j0 + r][b], wreg[r], acc[b]);
                    }
                } else {
#pragma unroll 1
                    for (int r0 = 0; r0 < ROWS; r0 += 16) {  // 16 weight loads in flight, then their products
                        float wv[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) wv[i] = wp[(int64_t)(r0 + i) * K];
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
#pragma unroll
                            for (int b = 0; b < BT; ++b) acc[b] = fmaf(s_dpre_t[j0 + r0 + i][b], wv[i], acc[b]);
                        }
                    }
                }
            }
            if (NJ > 1) {
                if (part == 1) {
#pragma unroll
                    for (int b = 0; b < BT; ++b) s_part[b][kcol] = acc[b];
                }
                __syncthreads();
                if (part == 0) {
#pragma unroll
                    for (int b = 0; b < BT; ++b) acc[b] += s_part[b][kcol];
                }
            }
            if (part == 0) {
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    if (kcol < I) {
                        if (b0 + b < B) atomicAdd(&dx[((b0 + b) * T + t) * I + kcol], acc[b]);   // + the other direction's share
                    } else {
                        s_dh[b][kcol - I] = acc[b];
                    }
                }
            }
        }
    }
}

template <int H, int I, int BT>
static int lstm_launch_fwd(const float* x, const float* wcat, const float* bias, int64_t B, int64_t T, float* y,
                           float* gates, float* cst, float* xh, hipStream_t st)
{
    hipLaunchKernelGGL((lstm_fwd_kernel<H, I, BT>), dim3((unsigned)((B + BT - 1) / BT), 2), dim3(4 * H), 0, st, x, wcat,
                       bias, B, T, y, gates, cst, xh);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

template <int H, int I, int BT>
static int lstm_launch_bwd(const float* wcat, const float* gates, const float* cst, const float* dy, int64_t B, int64_t T,
                           float* dx, float* dpre, hipStream_t st)
{
    hipLaunchKernelGGL((lstm_bwd_kernel<H, I, BT>), dim3((unsigned)((B + BT - 1) / BT), 2), dim3(4 * H), 0, st, wcat,
                       gates, cst, dy, B, T, dx, dpre);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// tile height: 8 sequences per workgroup once that fills the chip twice over, fewer for small batches
static int lstm_tile(int64_t B) { return B >= 2048 ? 8 : (B >= 768 ? 4 : 2); }

#define LSTM_DISPATCH(CALL)                                                                  \
    do {                                                                                     \
        const int bt = lstm_tile(B);                                                         \
        if (hidden_size == 64 && input_size == 64) {                                         \
            if (bt == 8) return CALL(64, 64, 8); if (bt == 4) return CALL(64, 64, 4); return CALL(64, 64, 2);      \
        } else if (hidden_size == 64) {                                                      \
            if (bt == 8) return CALL(64, 128, 8); if (bt == 4) return CALL(64, 128, 4); return CALL(64, 128, 2);   \
        } else if (input_size == 32) {                                                       \
            if (bt == 8) return CALL(32, 32, 8); if (bt == 4) return CALL(32, 32, 4); return CALL(32, 32, 2);      \
        } else {                                                                             \
            if (bt == 8) return CALL(32, 64, 8); if (bt == 4) return CALL(32, 64, 4); return CALL(32, 64, 2);      \
        }                                                                                    \
    } while (0)

extern "C" int sgnn_lstm_supported(int64_t input_size, int64_t hidden_size)
{
    return ((hidden_size == 64 || hidden_size == 32) && (input_size == hidden_size || input_size == 2 * hidden_size)) ? 1 : 0;
}

extern "C" int sgnn_lstm_fwd(const float* x, const float* wcat, const float* bias, int64_t B, int64_t T,
                             int64_t input_size, int64_t hidden_size, float* y, float* gates, float* cell, float* xh,
                             void* stream)
{
    if (!x || !wcat || !bias || !y || !gates || !cell || !xh || B < 0 || T < 0) return SGNN_ERR_BAD_ARG;
    if (!sgnn_lstm_supported(input_size, hidden_size)) return SGNN_ERR_UNSUPPORTED_D;
    if (B == 0 || T == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
#define LSTM_FWD(HH, II, BB) lstm_launch_fwd<HH, II, BB>(x, wcat, bias, B, T, y, gates, cell, xh, st)
    LSTM_DISPATCH(LSTM_FWD);
#undef LSTM_FWD
}

extern "C" int sgnn_lstm_bwd(const float* wcat, const float* gates, const float* cell, const float* dy, int64_t B,
                             int64_t T, int64_t input_size, int64_t hidden_size, float* dx, float* dgates, void* stream)
{
    if (!wcat || !gates || !cell || !dy || !dx || !dgates || B < 0 || T < 0) return SGNN_ERR_BAD_ARG;
    if (!sgnn_lstm_supported(input_size, hidden_size)) return SGNN_ERR_UNSUPPORTED_D;
    if (B == 0 || T == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
#define LSTM_BWD(HH, II, BB) lstm_launch_bwd<HH, II, BB>(wcat, gates, cell, dy, B, T, dx, dgates, st)
    LSTM_DISPATCH(LSTM_BWD);
#undef LSTM_BWD
}
