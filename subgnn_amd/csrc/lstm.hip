// One bidirectional LSTM layer over short sequences, forward and backward, one launch each.
// Replaces the nn.LSTM(bidirectional=True) of the structure channel's walk aggregator (reference
// SubGNN/SubGNN.py:60-88, called from SubGNN/anchor_patch_samplers.py:413-433 on (patches x walks,
// walk_len, D) inputs: a few hundred sequences of 10-20 steps, hidden size D).  The vendor library
// runs such a layer as one small GEMM plus one point-wise launch per time step and direction --
// ~80 launches forward and ~80 backward, each a few microseconds of launch latency around
// nanoseconds of arithmetic; here the recurrence stays inside one kernel.
//
// Layout.  A workgroup owns a tile of LSTM_BT sequences of one direction and has 4H lanes: lane j
// holds row j of [W_ih | W_hh] (gate order i, f, g, o as in torch) in registers for the whole
// sequence.  Per step: [x_t | h_{t-1}] of the tile is staged in LDS, every lane forms its gate
// pre-activation for the LSTM_BT sequences (broadcast LDS reads, K = I + H fma per sequence), the
// pre-activations cross to the lanes that own (sequence, hidden unit) pairs through LDS, and those
// apply the non-linearities and keep c in registers.  Gates and cell states are saved for the
// backward pass, two launches: the recurrence walks the steps in reverse -- gate gradients in the
// owner lanes, dx and the recurrent dh as a (LSTM_BT x 4H) x (4H x K) product with the weight columns
// streamed from L2 -- and leaves the gate gradients in memory; the weight gradient is then
// accumulated in the registers of lane j (row j, all K columns) over all steps of the tile and added
// to the global gradient once.  (One kernel holding both the K accumulators and the product's
// operands spills to scratch.)
#include "common.h"

#define LSTM_BT 8

__device__ __forceinline__ float lstm_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

template <int H, int I>
__global__ __launch_bounds__(4 * H) void lstm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wcat, const float* __restrict__ bias, int64_t B, int64_t T,
    float* __restrict__ y, float* __restrict__ gates, float* __restrict__ cst)
{
    constexpr int G = 4 * H, K = I + H;
    __shared__ __attribute__((aligned(16))) float s_in[LSTM_BT][K];      // [x_t | h_{t-1}]
    __shared__ float s_pre[LSTM_BT][G];
    const int j = threadIdx.x, d = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * LSTM_BT;
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = wcat[((int64_t)d * G + j) * K + k];
    const float bj = bias[d * G + j];
    const int u = j % H, bq = j / H;                       // owner of (bq, u) and (bq + 4, u)
    float c_own[2] = {0.f, 0.f};
    for (int idx = j; idx < LSTM_BT * H; idx += G) s_in[idx / H][I + idx % H] = 0.f;
    for (int64_t step = 0; step < T; ++step) {
        const int64_t t = d ? T - 1 - step : step;
        for (int idx = j; idx < LSTM_BT * I; idx += G) {
            const int b = idx / I, k = idx % I;
            s_in[b][k] = (b0 + b < B) ? x[((b0 + b) * T + t) * I + k] : 0.f;
        }
        __syncthreads();
        float acc[LSTM_BT];
#pragma unroll
        for (int b = 0; b < LSTM_BT; ++b) acc[b] = bj;
#pragma unroll
        for (int k = 0; k < K; k += 4) {
#pragma unroll
            for (int b = 0; b < LSTM_BT; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(&s_in[b][k]);
                acc[b] = fmaf(w[k], v.x, acc[b]);
                acc[b] = fmaf(w[k + 1], v.y, acc[b]);
                acc[b] = fmaf(w[k + 2], v.z, acc[b]);
                acc[b] = fmaf(w[k + 3], v.w, acc[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < LSTM_BT; ++b) s_pre[b][j] = acc[b];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = bq + 4 * q;
            const float gi = lstm_sigmoid(s_pre[b][u]), gf = lstm_sigmoid(s_pre[b][H + u]);
            const float gg = tanhf(s_pre[b][2 * H + u]), go = lstm_sigmoid(s_pre[b][3 * H + u]);
            const float c = gf * c_own[q] + gi * gg;
            const float h = go * tanhf(c);
            c_own[q] = c;
            s_in[b][I + u] = h;                              // h_{t-1} of the next step (the x part is rewritten above)
            if (b0 + b < B) {
                float* gp = gates + (((int64_t)d * B + b0 + b) * T + t) * G;
                gp[u] = gi; gp[H + u] = gf; gp[2 * H + u] = gg; gp[3 * H + u] = go;
                cst[(((int64_t)d * B + b0 + b) * T + t) * H + u] = c;
                y[((b0 + b) * T + t) * (2 * H) + d * H + u] = h;
            }
        }
        // the next step's staging writes s_in[.][0..I) only, its dot products start after the barrier
        // that follows -- by then every owner lane has stored its h
    }
}

template <int H, int I>
__global__ __launch_bounds__(4 * H) void lstm_bwd_kernel(
    const float* __restrict__ wcat, const float* __restrict__ gates, const float* __restrict__ cst,
    const float* __restrict__ dy, int64_t B, int64_t T, float* __restrict__ dx, float* __restrict__ dpre)
{
    constexpr int G = 4 * H, K = I + H;
    constexpr int NJ = (G % K == 0) ? G / K : 1;           // lanes per output column of the dx/dh product
    __shared__ __attribute__((aligned(16))) float s_dpre_t[G][LSTM_BT];  // gate gradients [row][sequence]: broadcast reads over rows
    __shared__ float s_dh[LSTM_BT][H];                                    // recurrent dh from the step after
    __shared__ float s_part[LSTM_BT][K];
    const int j = threadIdx.x, d = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * LSTM_BT;
    const int u = j % H, bq = j / H;
    float dc_own[2] = {0.f, 0.f};
    for (int idx = j; idx < LSTM_BT * H; idx += G) s_dh[idx / H][idx % H] = 0.f;
    for (int64_t step = 0; step < T; ++step) {
        const int64_t sf = T - 1 - step;                    // forward step being undone
        const int64_t t = d ? T - 1 - sf : sf;
        const int64_t tp = d ? t + 1 : t - 1;               // time index of the forward step before it
        __syncthreads();                                    // s_dh / s_part / s_dpre_t of the previous step are done with
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = bq + 4 * q;
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
            }
            if (b0 + b < B) {
                float* dp = dpre + (((int64_t)d * B + b0 + b) * T + t) * G;
                dp[u] = pi; dp[H + u] = pf; dp[2 * H + u] = pg; dp[3 * H + u] = po;
            }
            s_dpre_t[u][b] = pi; s_dpre_t[H + u][b] = pf; s_dpre_t[2 * H + u][b] = pg; s_dpre_t[3 * H + u][b] = po;
        }
        __syncthreads();
        // [dx_t | dh_{t-1}][b, k] = sum_j dpre[b, j] * W[j, k]: lane (k, part) sums its share of the rows
        {
            const int k = j % K, part = j / K;
            float acc[LSTM_BT];
#pragma unroll
            for (int b = 0; b < LSTM_BT; ++b) acc[b] = 0.f;
            if (part < NJ) {
                const int j0 = part * (G / NJ), j1 = j0 + G / NJ;
                const float* wp = wcat + (int64_t)d * G * K + k;
#pragma unroll 1
                for (int r0 = j0; r0 < j1; r0 += 16) {      // 16 weight loads in flight, then their products
                    float wv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) wv[i] = wp[(int64_t)(r0 + i) * K];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float4 a = *reinterpret_cast<const float4*>(&s_dpre_t[r0 + i][0]);
                        const float4 c4 = *reinterpret_cast<const float4*>(&s_dpre_t[r0 + i][4]);
                        acc[0] = fmaf(a.x, wv[i], acc[0]); acc[1] = fmaf(a.y, wv[i], acc[1]);
                        acc[2] = fmaf(a.z, wv[i], acc[2]); acc[3] = fmaf(a.w, wv[i], acc[3]);
                        acc[4] = fmaf(c4.x, wv[i], acc[4]); acc[5] = fmaf(c4.y, wv[i], acc[5]);
                        acc[6] = fmaf(c4.z, wv[i], acc[6]); acc[7] = fmaf(c4.w, wv[i], acc[7]);
                    }
                }
            }
            if (NJ > 1) {
                if (part == 1) {
#pragma unroll
                    for (int b = 0; b < LSTM_BT; ++b) s_part[b][k] = acc[b];
                }
                __syncthreads();
                if (part == 0) {
#pragma unroll
                    for (int b = 0; b < LSTM_BT; ++b) acc[b] += s_part[b][k];
                }
            }
            if (part == 0) {
#pragma unroll
                for (int b = 0; b < LSTM_BT; ++b) {
                    if (k < I) {
                        if (b0 + b < B) atomicAdd(&dx[((b0 + b) * T + t) * I + k], acc[b]);   // + the other direction's share
                    } else {
                        s_dh[b][k - I] = acc[b];
                    }
                }
            }
        }
    }
}

// dW[j, :] += sum over the tile's sequences and all steps of dpre[., ., j] * [x_t | h_{t-1}]; db likewise
template <int H, int I>
__global__ __launch_bounds__(4 * H) void lstm_dw_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dpre, int64_t B, int64_t T,
    float* __restrict__ dwcat, float* __restrict__ dbias)
{
    constexpr int G = 4 * H, K = I + H;
    __shared__ __attribute__((aligned(16))) float s_in[LSTM_BT][K];
    const int j = threadIdx.x, d = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * LSTM_BT;
    float dw[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dw[k] = 0.f;
    float db = 0.f;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t tp = d ? t + 1 : t - 1;               // the step before t in this direction's order
        const bool has_prev = d ? (t + 1 < T) : (t > 0);
        __syncthreads();
        for (int idx = j; idx < LSTM_BT * K; idx += G) {
            const int b = idx / K, k = idx % K;
            float v = 0.f;
            if (b0 + b < B) {
                if (k < I) v = x[((b0 + b) * T + t) * I + k];
                else if (has_prev) v = y[((b0 + b) * T + tp) * (2 * H) + d * H + (k - I)];
            }
            s_in[b][k] = v;
        }
        float dp[LSTM_BT];
#pragma unroll
        for (int b = 0; b < LSTM_BT; ++b) {
            dp[b] = (b0 + b < B) ? dpre[(((int64_t)d * B + b0 + b) * T + t) * G + j] : 0.f;
            db += dp[b];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k += 4) {
#pragma unroll
            for (int b = 0; b < LSTM_BT; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(&s_in[b][k]);
                dw[k] = fmaf(dp[b], v.x, dw[k]);
                dw[k + 1] = fmaf(dp[b], v.y, dw[k + 1]);
                dw[k + 2] = fmaf(dp[b], v.z, dw[k + 2]);
                dw[k + 3] = fmaf(dp[b], v.w, dw[k + 3]);
            }
        }
    }
    float* dwp = dwcat + ((int64_t)d * G + j) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) atomicAdd(dwp + k, dw[k]);
    atomicAdd(&dbias[d * G + j], db);
}

template <int H, int I>
static int lstm_launch_fwd(const float* x, const float* wcat, const float* bias, int64_t B, int64_t T, float* y,
                           float* gates, float* cst, hipStream_t st)
{
    hipLaunchKernelGGL((lstm_fwd_kernel<H, I>), dim3((unsigned)((B + LSTM_BT - 1) / LSTM_BT), 2), dim3(4 * H), 0, st, x,
                       wcat, bias, B, T, y, gates, cst);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

template <int H, int I>
static int lstm_launch_bwd(const float* x, const float* wcat, const float* y, const float* gates, const float* cst,
                           const float* dy, int64_t B, int64_t T, float* dx, float* dpre, float* dwcat, float* dbias,
                           hipStream_t st)
{
    const dim3 grid((unsigned)((B + LSTM_BT - 1) / LSTM_BT), 2);
    hipLaunchKernelGGL((lstm_bwd_kernel<H, I>), grid, dim3(4 * H), 0, st, wcat, gates, cst, dy, B, T, dx, dpre);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL((lstm_dw_kernel<H, I>), grid, dim3(4 * H), 0, st, x, y, dpre, B, T, dwcat, dbias);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_lstm_supported(int64_t input_size, int64_t hidden_size)
{
    return ((hidden_size == 64 || hidden_size == 32) && (input_size == hidden_size || input_size == 2 * hidden_size)) ? 1 : 0;
}

extern "C" int sgnn_lstm_fwd(const float* x, const float* wcat, const float* bias, int64_t B, int64_t T,
                             int64_t input_size, int64_t hidden_size, float* y, float* gates, float* cell, void* stream)
{
    if (!x || !wcat || !bias || !y || !gates || !cell || B < 0 || T < 0) return SGNN_ERR_BAD_ARG;
    if (!sgnn_lstm_supported(input_size, hidden_size)) return SGNN_ERR_UNSUPPORTED_D;
    if (B == 0 || T == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hidden_size == 64 && input_size == 64) return lstm_launch_fwd<64, 64>(x, wcat, bias, B, T, y, gates, cell, st);
    if (hidden_size == 64) return lstm_launch_fwd<64, 128>(x, wcat, bias, B, T, y, gates, cell, st);
    if (input_size == 32) return lstm_launch_fwd<32, 32>(x, wcat, bias, B, T, y, gates, cell, st);
    return lstm_launch_fwd<32, 64>(x, wcat, bias, B, T, y, gates, cell, st);
}

extern "C" int sgnn_lstm_bwd(const float* x, const float* wcat, const float* y, const float* gates, const float* cell,
                             const float* dy, int64_t B, int64_t T, int64_t input_size, int64_t hidden_size, float* dx,
                             float* dgates, float* dwcat, float* dbias, void* stream)
{
    if (!x || !wcat || !y || !gates || !cell || !dy || !dx || !dgates || !dwcat || !dbias || B < 0 || T < 0)
        return SGNN_ERR_BAD_ARG;
    if (!sgnn_lstm_supported(input_size, hidden_size)) return SGNN_ERR_UNSUPPORTED_D;
    if (B == 0 || T == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hidden_size == 64 && input_size == 64)
        return lstm_launch_bwd<64, 64>(x, wcat, y, gates, cell, dy, B, T, dx, dgates, dwcat, dbias, st);
    if (hidden_size == 64)
        return lstm_launch_bwd<64, 128>(x, wcat, y, gates, cell, dy, B, T, dx, dgates, dwcat, dbias, st);
    if (input_size == 32) return lstm_launch_bwd<32, 32>(x, wcat, y, gates, cell, dy, B, T, dx, dgates, dwcat, dbias, st);
    return lstm_launch_bwd<32, 64>(x, wcat, y, gates, cell, dy, B, T, dx, dgates, dwcat, dbias, st);
}
