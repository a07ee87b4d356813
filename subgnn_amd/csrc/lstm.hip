// The recurrence of one bidirectional LSTM layer over short sequences, forward and backward, one
// launch each.  Replaces the nn.LSTM(bidirectional=True) of the structure channel's walk aggregator
// (reference SubGNN/SubGNN.py:60-88, called from SubGNN/anchor_patch_samplers.py:413-433 on
// (patches x walks, walk_len, D) inputs: a few hundred sequences of 10-20 steps, hidden size D).
// The vendor library runs such a layer as one small GEMM plus one point-wise launch per time step
// and direction -- ~80 launches forward and ~80 backward per call, each a few microseconds of launch
// latency around nanoseconds of arithmetic; here the recurrence stays inside one kernel.
//
// Split of the work.  Everything that is NOT recurrent is a plain GEMM over all (sequence, step)
// rows at once and stays with the library (the caller): the input projection x W_ih^T + b of both
// directions before the forward kernel, and in the backward pass dx = dgates W_ih, dW_ih = dgates^T x,
// dW_hh = dgates^T h_prev, db = column sums of dgates.  The kernels hold only W_hh, which is what
// makes hidden size 128 fit: a direction's W_hh is 4H x H = 256 KB at H = 128, i.e. H registers in
// each of 4H lanes.
//
// Forward.  A workgroup owns a tile of BT sequences of one direction and has 4H lanes: lane j holds
// row j of W_hh (gate order i, f, g, o as in torch) in registers for the whole sequence.  Per step:
// the lane's pre-activation starts from the projected input (loaded one step ahead), adds
// W_hh[j, :] . h_{t-1} for the BT sequences (h in LDS, broadcast reads), the pre-activations cross
// to the lanes that own (sequence, hidden unit) pairs through LDS, and those apply the
// non-linearities and keep c in registers.  Gates, cell states and h_{t-1} are kept for backward.
// Backward walks the steps in reverse: gate gradients in the owner lanes (written out for the
// caller's GEMMs), then dh_{t-1} = dgates . W_hh as a (BT x 4H) x (4H x H) product whose weight
// columns sit in registers (lane (k, part) holds a quarter of column k).
//
// The barriers inside the step loops order LDS traffic only (fence on the "local" address space):
// a __syncthreads() would also drain the global loads issued one step ahead and the stores of the
// kept activations, which is most of a step's latency.  Nothing a lane writes to global memory is
// read by another lane of the same launch.
// The tile height BT (2, 4 or 8 sequences) is chosen by the host so that a small batch still spreads
// over the chip: the recurrence is a latency chain, a workgroup is one or two wavefronts per SIMD.
#include "common.h"

__device__ __forceinline__ float lstm_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void lstm_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// pre_x: (2, B, T, 4H) projected inputs (direction-major: each direction's rows are one GEMM's contiguous output) incl. b_ih;
// whh_f / whh_r: (4H, H) each; bhh_f / bhh_r: (4H) each, nullable -- added here, so that the caller needs neither a
// concatenation of the two directions' weights nor a sum of the two bias vectors (five small launches per layer before)
// y: (B, T, 2H); gates: (2, B, T, 4H); cst, hprev: (2, B, T, H)
template <int H, int BT>
__global__ __launch_bounds__(4 * H) void lstm_fwd_kernel(
    const float* __restrict__ pre_x, const float* __restrict__ whh_f, const float* __restrict__ whh_r,
    const float* __restrict__ bhh_f, const float* __restrict__ bhh_r, int64_t B, int64_t T,
    float* __restrict__ y, float* __restrict__ gates, float* __restrict__ cst, float* __restrict__ hprev)
{
    constexpr int G = 4 * H;
    constexpr int OWN = (BT * H + G - 1) / G;              // (sequence, unit) pairs per owner lane
    __shared__ __attribute__((aligned(16))) float s_h[BT][H];
    __shared__ float s_pre[BT][G];
    const int j = threadIdx.x, d = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * BT;
    float w[H];
#pragma unroll
    for (int k = 0; k < H; ++k) w[k] = (d ? whh_r : whh_f)[(int64_t)j * H + k];
    const float* __restrict__ bhh = d ? bhh_r : bhh_f;
    const float bj = bhh ? bhh[j] : 0.f;
    float c_own[OWN];
#pragma unroll
    for (int q = 0; q < OWN; ++q) c_own[q] = 0.f;
    for (int idx = j; idx < BT * H; idx += G) s_h[idx / H][idx % H] = 0.f;
    float cur[BT], nxt[BT];
    {
        const int64_t t0 = d ? T - 1 : 0;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            cur[b] = (b0 + b < B) ? pre_x[(((int64_t)d * B + b0 + b) * T + t0) * G + j] + bj : 0.f;
            nxt[b] = 0.f;
        }
    }
    lstm_lds_barrier();
    for (int64_t step = 0; step < T; ++step) {
        const int64_t t = d ? T - 1 - step : step;
        if (step + 1 < T) {                                 // next step's projected input, in flight during this one
            const int64_t tn = d ? t - 1 : t + 1;
#pragma unroll
            for (int b = 0; b < BT; ++b) nxt[b] = (b0 + b < B) ? pre_x[(((int64_t)d * B + b0 + b) * T + tn) * G + j] + bj : 0.f;
        }
        for (int idx = j; idx < BT * H; idx += G) {         // h_{t-1}, kept for the caller's dW_hh
            const int b = idx / H, k = idx % H;
            if (b0 + b < B) hprev[(((int64_t)d * B + b0 + b) * T + t) * H + k] = s_h[b][k];
        }
        float acc[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b] = cur[b];
#pragma unroll
        for (int k = 0; k < H; k += 4) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(&s_h[b][k]);
                acc[b] = fmaf(w[k], v.x, acc[b]);
                acc[b] = fmaf(w[k + 1], v.y, acc[b]);
                acc[b] = fmaf(w[k + 2], v.z, acc[b]);
                acc[b] = fmaf(w[k + 3], v.w, acc[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) s_pre[b][j] = acc[b];
        lstm_lds_barrier();                                 // all dot products done: s_h may be rewritten
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
                s_h[b][u] = h;
                if (b0 + b < B) {
                    float* gp = gates + (((int64_t)d * B + b0 + b) * T + t) * G;
                    gp[u] = gi; gp[H + u] = gf; gp[2 * H + u] = gg; gp[3 * H + u] = go;
                    cst[(((int64_t)d * B + b0 + b) * T + t) * H + u] = c;
                    y[((b0 + b) * T + t) * (2 * H) + d * H + u] = h;
                }
            }
        }
        lstm_lds_barrier();                                 // new h visible; s_pre may be rewritten
#pragma unroll
        for (int b = 0; b < BT; ++b) cur[b] = nxt[b];
    }
}

// dgates: (2, B, T, 4H) written (direction-major like gates: a direction's rows are contiguous for the caller's contractions)
template <int H, int BT>
__global__ __launch_bounds__(4 * H) void lstm_bwd_kernel(
    const float* __restrict__ whh_f, const float* __restrict__ whh_r, const float* __restrict__ gates, const float* __restrict__ cst,
    const float* __restrict__ dy, int64_t B, int64_t T, float* __restrict__ dgates)
{
    constexpr int G = 4 * H;
    constexpr int OWN = (BT * H + G - 1) / G;
    constexpr int NJ = 4;                                   // lanes per output column of the dh product: 4H lanes, H columns
    constexpr int ROWS = G / NJ;                            // = H gate rows per lane
    __shared__ __attribute__((aligned(16))) float s_dpre_t[G][BT];        // gate gradients [row][sequence]: broadcast reads over rows
    __shared__ float s_dh[BT][H];                                          // recurrent dh from the step after
    __shared__ float s_part[NJ - 1][BT][H];
    const int j = threadIdx.x, d = blockIdx.y;
    const int64_t b0 = (int64_t)blockIdx.x * BT;
    const int kcol = j % H, part = j / H;
    float wreg[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wreg[r] = (d ? whh_r : whh_f)[(int64_t)(part * ROWS + r) * H + kcol];
    float dc_own[OWN];
#pragma unroll
    for (int q = 0; q < OWN; ++q) dc_own[q] = 0.f;
    for (int idx = j; idx < BT * H; idx += G) s_dh[idx / H][idx % H] = 0.f;
    lstm_lds_barrier();
    for (int64_t step = 0; step < T; ++step) {
        const int64_t sf = T - 1 - step;                    // forward step being undone
        const int64_t t = d ? T - 1 - sf : sf;
        const int64_t tp = d ? t + 1 : t - 1;               // time index of the forward step before it
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
            const int p = j + q * G;
            if (p < BT * H) {
                const int b = p / H, u = p % H;
                float pi = 0.f, pf = 0.f, pg = 0.f, po = 0.f;
                if (b0 + b < B) {
                    const int64_t rowg = ((int64_t)d * B + b0 + b) * T + t;
                    const float* gp = gates + rowg * G;
                    const float gi = gp[u], gf = gp[H + u], gg = gp[2 * H + u], go = gp[3 * H + u];
                    const float c = cst[(((int64_t)d * B + b0 + b) * T + t) * H + u];
                    const float cprev = sf > 0 ? cst[(((int64_t)d * B + b0 + b) * T + tp) * H + u] : 0.f;
                    const float dh = dy[((b0 + b) * T + t) * (2 * H) + d * H + u] + s_dh[b][u];
                    const float tc = tanhf(c);
                    const float dc = dh * go * (1.f - tc * tc) + dc_own[q];
                    dc_own[q] = dc * gf;
                    pi = dc * gg * gi * (1.f - gi);
                    pf = dc * cprev * gf * (1.f - gf);
                    pg = dc * gi * (1.f - gg * gg);
                    po = dh * tc * go * (1.f - go);
                    float* dp = dgates + rowg * G;
                    dp[u] = pi; dp[H + u] = pf; dp[2 * H + u] = pg; dp[3 * H + u] = po;
                }
                s_dpre_t[u][b] = pi; s_dpre_t[H + u][b] = pf; s_dpre_t[2 * H + u][b] = pg; s_dpre_t[3 * H + u][b] = po;
            }
        }
        lstm_lds_barrier();                                 // gate gradients visible; s_dh fully read
        // dh_{t-1}[b, k] = sum_j dgates[b, j] * W_hh[j, k]: lane (k, part) sums its quarter of the rows
        float acc[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b] = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
#pragma unroll
            for (int b = 0; b < BT; ++b) acc[b] = fmaf(s_dpre_t[part * ROWS + r][b], wreg[r], acc[b]);
        }
        if (part > 0) {
#pragma unroll
            for (int b = 0; b < BT; ++b) s_part[part - 1][b][kcol] = acc[b];
        }
        lstm_lds_barrier();
        if (part == 0) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                float v = acc[b];
#pragma unroll
                for (int pp = 0; pp < NJ - 1; ++pp) v += s_part[pp][b][kcol];
                s_dh[b][kcol] = v;
            }
        }
        lstm_lds_barrier();                                 // s_dh ready; s_dpre_t / s_part may be rewritten
    }
}

template <int H, int BT>
static int lstm_launch_fwd(const float* pre_x, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                           int64_t B, int64_t T, float* y, float* gates, float* cst, float* hprev, hipStream_t st)
{
    hipLaunchKernelGGL((lstm_fwd_kernel<H, BT>), dim3((unsigned)((B + BT - 1) / BT), 2), dim3(4 * H), 0, st, pre_x, whh_f, whh_r,
                       bhh_f, bhh_r, B, T, y, gates, cst, hprev);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

template <int H, int BT>
static int lstm_launch_bwd(const float* whh_f, const float* whh_r, const float* gates, const float* cst, const float* dy,
                           int64_t B, int64_t T, float* dgates, hipStream_t st)
{
    hipLaunchKernelGGL((lstm_bwd_kernel<H, BT>), dim3((unsigned)((B + BT - 1) / BT), 2), dim3(4 * H), 0, st, whh_f, whh_r, gates,
                       cst, dy, B, T, dgates);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// tile height: 8 sequences per workgroup once that fills the chip twice over, fewer for small batches
static int lstm_tile(int64_t B) { return B >= 2048 ? 8 : (B >= 768 ? 4 : 2); }

#define LSTM_DISPATCH(CALL)                                                                  \
    do {                                                                                     \
        const int bt = lstm_tile(B);                                                         \
        if (hidden_size == 128) {                                                            \
            if (bt == 8) return CALL(128, 8); if (bt == 4) return CALL(128, 4); return CALL(128, 2);   \
        } else if (hidden_size == 64) {                                                      \
            if (bt == 8) return CALL(64, 8); if (bt == 4) return CALL(64, 4); return CALL(64, 2);      \
        } else {                                                                             \
            if (bt == 8) return CALL(32, 8); if (bt == 4) return CALL(32, 4); return CALL(32, 2);      \
        }                                                                                    \
    } while (0)

extern "C" int sgnn_lstm_supported(int64_t hidden_size)
{
    return (hidden_size == 128 || hidden_size == 64 || hidden_size == 32) ? 1 : 0;
}

extern "C" int sgnn_lstm_fwd(const float* pre_x, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                             int64_t B, int64_t T, int64_t hidden_size, float* y, float* gates, float* cell, float* hprev,
                             void* stream)
{
    if (!pre_x || !whh_f || !whh_r || !y || !gates || !cell || !hprev || B < 0 || T < 0) return SGNN_ERR_BAD_ARG;
    if (!sgnn_lstm_supported(hidden_size)) return SGNN_ERR_UNSUPPORTED_D;
    if (B == 0 || T == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
#define LSTM_FWD(HH, BB) lstm_launch_fwd<HH, BB>(pre_x, whh_f, whh_r, bhh_f, bhh_r, B, T, y, gates, cell, hprev, st)
    LSTM_DISPATCH(LSTM_FWD);
#undef LSTM_FWD
}

extern "C" int sgnn_lstm_bwd(const float* whh_f, const float* whh_r, const float* gates, const float* cell, const float* dy,
                             int64_t B, int64_t T, int64_t hidden_size, float* dgates, void* stream)
{
    if (!whh_f || !whh_r || !gates || !cell || !dy || !dgates || B < 0 || T < 0) return SGNN_ERR_BAD_ARG;
    if (!sgnn_lstm_supported(hidden_size)) return SGNN_ERR_UNSUPPORTED_D;
    if (B == 0 || T == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
#define LSTM_BWD(HH, BB) lstm_launch_bwd<HH, BB>(whh_f, whh_r, gates, cell, dy, B, T, dgates, st)
    LSTM_DISPATCH(LSTM_BWD);
#undef LSTM_BWD
}

SGNN_DEFINE_WARM(lstm)
