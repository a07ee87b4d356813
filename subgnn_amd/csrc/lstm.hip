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

// ---- the products around the recurrence, in this library's own launches ---------------------------------------------------------
// The walk aggregator (anchor_patch_samplers.py:413-433: embedding lookup of the walks -> LSTM -> last step -> Linear -> sum over a
// patch's walks) is a few hundred sequences of 10-20 steps: its dense products -- input projection, dx, the tail's Linear -- are
// GEMMs of a few thousand rows by 64-256 columns.  As library calls they were 3 + 2 launches forward and ~20 backward per LSTM
// layer (gather, two addmm; in the backward two GEMMs + copies for dx, four weight contractions with their block sums, four
// column sums, the zero-filled gradient of the last-step slice, the Linear's three products), each a 5-40 us launch around
// microseconds of arithmetic.  Here:
//   rows_gemm_kernel     out[z][r][n] = sum_k X[row(r)][k] W_z[n][k] + b_z[n] for z = 0, 1 (the two directions): fp32 MFMA, a lane
//                        reads half of its (gathered) input row straight into registers -- the embedding lookup of the walks IS
//                        the operand load; the gathered rows are written out once for the weight gradients
//   rows_gemm_nt_kernel  dx[r][n] = sum_z sum_k A_z[r][k] W_z[k][n] (both directions into one accumulator)
//   lstm_tail_*          X[p] = (sum over the patch's walks of the last step) W_lin^T + W b_lin, and its backward incl. the dense,
//                        mostly zero d y the recurrence's backward reads
// The weight gradients dW_ih, dW_hh, db of both directions are six jobs of sgnn_contract_rows_partial (head.hip).
typedef float lstm_f32x16 __attribute__((ext_vector_type(16)));
#define LSTM_KC 16

__device__ __forceinline__ int lstm_acc_row(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

// one wavefront per (32 rows, 32 output columns, direction z); K even, K / 2 a multiple of 4
__global__ __launch_bounds__(64) void rows_gemm_kernel(const float* __restrict__ X, const int64_t* __restrict__ ids, int64_t ldx,
                                                       int64_t R, int K, const float* __restrict__ W0, const float* __restrict__ W1,
                                                       const float* __restrict__ b0, const float* __restrict__ b1, int N,
                                                       float* __restrict__ out, float* __restrict__ xcopy)
{
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5, z = blockIdx.z, nt = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int64_t r = row0 + i < R ? row0 + i : R - 1;
    const int64_t src_row = ids ? ids[r] : r;
    const int Kh = K / 2;
    const float* __restrict__ src = X + src_row * ldx + h * Kh;
    const int n = nt * 32 + i;
    const float* __restrict__ W = z ? W1 : W0;
    const float* __restrict__ wrow = W + (int64_t)(n < N ? n : N - 1) * K + h * Kh;
    lstm_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool copy = xcopy && nt == 0 && z == 0 && row0 + i < R;
    for (int kc = 0; kc < Kh; kc += LSTM_KC) {
        float a[LSTM_KC], w[LSTM_KC];
#pragma unroll
        for (int c = 0; c < LSTM_KC / 4; ++c) {
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f), wv = av;
            if (kc + 4 * c < Kh) {
                av = *reinterpret_cast<const float4*>(src + kc + 4 * c);
                wv = *reinterpret_cast<const float4*>(wrow + kc + 4 * c);
                if (copy) *reinterpret_cast<float4*>(xcopy + (row0 + i) * K + h * Kh + kc + 4 * c) = av;
            }
            a[4 * c] = av.x; a[4 * c + 1] = av.y; a[4 * c + 2] = av.z; a[4 * c + 3] = av.w;
            w[4 * c] = wv.x; w[4 * c + 1] = wv.y; w[4 * c + 2] = wv.z; w[4 * c + 3] = wv.w;
        }
#pragma unroll
        for (int s = 0; s < LSTM_KC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w[s], acc, 0, 0, 0);
    }
    if (n >= N) return;
    const float* __restrict__ b = z ? b1 : b0;
    const float bias = b ? b[n] : 0.f;
    float* __restrict__ dst = out + (int64_t)z * R * N;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int64_t rr = row0 + lstm_acc_row(v, h);
        if (rr < R) dst[rr * N + n] = acc[v] + bias;
    }
}

// The same product with a tile's contraction split over the four wavefronts of a workgroup (K % 32 == 0; used from K = 128 on: a
// single wavefront walking K / 2 = 64-128 dependent MFMA + load steps ran the second LSTM layer's projection -- K = 2 H = 256 --
// at a fifth of the matrix cores' rate); partial tiles added in wavefront order through LDS.
__global__ __launch_bounds__(256) void rows_gemm_ksplit_kernel(const float* __restrict__ X, const int64_t* __restrict__ ids, int64_t ldx,
                                                               int64_t R, int K, const float* __restrict__ W0, const float* __restrict__ W1,
                                                               const float* __restrict__ b0, const float* __restrict__ b1, int N,
                                                               float* __restrict__ out, float* __restrict__ xcopy)
{
    __shared__ float s_part[3 * 16 * 64];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, z = blockIdx.z, nt = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int64_t r = row0 + i < R ? row0 + i : R - 1;
    const int64_t src_row = ids ? ids[r] : r;
    const int Kq = K / 8, k0 = wave * (K / 4) + h * Kq;              // positions per (wavefront, k-half); K / 8 is a multiple of 4
    const float* __restrict__ src = X + src_row * ldx + k0;
    const int n = nt * 32 + i;
    const float* __restrict__ wrow = (z ? W1 : W0) + (int64_t)(n < N ? n : N - 1) * K + k0;
    lstm_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool copy = xcopy && nt == 0 && z == 0 && row0 + i < R;
    for (int kc = 0; kc < Kq; kc += LSTM_KC) {
        float a[LSTM_KC], w[LSTM_KC];
#pragma unroll
        for (int c = 0; c < LSTM_KC / 4; ++c) {
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f), wv = av;
            if (kc + 4 * c < Kq) {
                av = *reinterpret_cast<const float4*>(src + kc + 4 * c);
                wv = *reinterpret_cast<const float4*>(wrow + kc + 4 * c);
                if (copy) *reinterpret_cast<float4*>(xcopy + (row0 + i) * K + k0 + kc + 4 * c) = av;
            }
            a[4 * c] = av.x; a[4 * c + 1] = av.y; a[4 * c + 2] = av.z; a[4 * c + 3] = av.w;
            w[4 * c] = wv.x; w[4 * c + 1] = wv.y; w[4 * c + 2] = wv.z; w[4 * c + 3] = wv.w;
        }
#pragma unroll
        for (int s = 0; s < LSTM_KC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w[s], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) s_part[((wave - 1) * 16 + v) * 64 + lane] = acc[v];
    }
    __syncthreads();
    if (wave != 0 || n >= N) return;
    const float* __restrict__ b = z ? b1 : b0;
    const float bias = b ? b[n] : 0.f;
    float* __restrict__ dst = out + (int64_t)z * R * N;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const float sum = ((acc[v] + s_part[v * 64 + lane]) + s_part[(16 + v) * 64 + lane]) + s_part[(32 + v) * 64 + lane];
        const int64_t rr = row0 + lstm_acc_row(v, h);
        if (rr < R) dst[rr * N + n] = sum + bias;
    }
}

// dx[r][n] = sum over z of sum_k A_z[r][k] W_z[k][n]; A_z = A + z * R * K (direction-major gate gradients), W_z (K, N) row-major.
// A workgroup of four wavefronts per (32 rows, 32 columns): wavefront w contracts direction w / 2, half w % 2 of its K positions
// (a tile's 2 K = 512-1024 positions walked by ONE wavefront were a chain of that many dependent MFMA + load steps on a few
// hundred wavefronts: 35 us for 3 700 rows); the four partial tiles are added in wavefront order through LDS (a fixed order).
__global__ __launch_bounds__(256) void rows_gemm_nt_kernel(const float* __restrict__ A, int64_t R, int K, const float* __restrict__ W0,
                                                           const float* __restrict__ W1, int N, float* __restrict__ out)
{
    __shared__ float s_part[3 * 16 * 64];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, nt = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int64_t r = row0 + i < R ? row0 + i : R - 1;
    const int Kh = K / 4;                                   // positions per (wavefront, k-half)
    const int n = nt * 32 + i, nn = n < N ? n : N - 1;
    lstm_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
        const int z = wave >> 1, k0 = (wave & 1) * (K / 2) + h * Kh;
        const float* __restrict__ arow = A + ((int64_t)z * R + r) * K + k0;
        const float* __restrict__ wcol = (z ? W1 : W0) + (int64_t)k0 * N + nn;
        for (int kc = 0; kc < Kh; kc += LSTM_KC) {
            float a[LSTM_KC], w[LSTM_KC];
#pragma unroll
            for (int c = 0; c < LSTM_KC / 4; ++c) {
                float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kc + 4 * c < Kh) av = *reinterpret_cast<const float4*>(arow + kc + 4 * c);
                a[4 * c] = av.x; a[4 * c + 1] = av.y; a[4 * c + 2] = av.z; a[4 * c + 3] = av.w;
            }
#pragma unroll
            for (int s = 0; s < LSTM_KC; ++s) w[s] = kc + s < Kh ? wcol[(int64_t)(kc + s) * N] : 0.f;
#pragma unroll
            for (int s = 0; s < LSTM_KC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w[s], acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) s_part[((wave - 1) * 16 + v) * 64 + lane] = acc[v];
    }
    __syncthreads();
    if (wave != 0 || n >= N) return;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const float sum = ((acc[v] + s_part[v * 64 + lane]) + s_part[(16 + v) * 64 + lane]) + s_part[(32 + v) * 64 + lane];
        const int64_t rr = row0 + lstm_acc_row(v, h);
        if (rr < R) out[rr * N + n] = sum;
    }
}

// X[p][d] = s[p] . W_lin[d] + n_walks b[d], s[p] = sum over the patch's walks of y[walk][T - 1][:] (last_only) or of every step;
// one workgroup per patch; s (n_patches, 2H) is kept for the backward
__global__ __launch_bounds__(256) void lstm_tail_fwd_kernel(const float* __restrict__ y, int64_t T, int H2, int n_walks, int last_only,
                                                            const float* __restrict__ Wl, const float* __restrict__ bl, int D,
                                                            float* __restrict__ s_out, float* __restrict__ X)
{
    extern __shared__ float s_sum[];                                      // [H2]
    const int64_t p = blockIdx.x;
    for (int c = threadIdx.x; c < H2; c += 256) {
        float v = 0.f;
        for (int w = 0; w < n_walks; ++w) {
            const float* __restrict__ yw = y + ((p * n_walks + w) * T) * H2;
            if (last_only) v += yw[(T - 1) * H2 + c];
            else for (int64_t t = 0; t < T; ++t) v += yw[t * H2 + c];
        }
        s_sum[c] = v;
        s_out[p * H2 + c] = v;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        const float* __restrict__ wr = Wl + (int64_t)d * H2;
        float v = 0.f;
        for (int c = 0; c < H2; ++c) v = fmaf(s_sum[c], wr[c], v);
        X[p * D + d] = v + (bl ? (float)n_walks * bl[d] : 0.f);
    }
}

// workgroups [0, B): d y of one sequence -- zeros except the last step (last_only) / the same row at every step: g[b] = dX[b / n_walks] W_lin;
// workgroups [B, B + ceil(D 2H / 256)): dW_lin[d][c] = sum_p dX[p][d] s[p][c] (patches in order); the workgroup after them: d b_lin
__global__ __launch_bounds__(256) void lstm_tail_bwd_kernel(const float* __restrict__ dX, const float* __restrict__ s, int64_t n_patches,
                                                            int64_t B, int64_t T, int H2, int n_walks, int last_only,
                                                            const float* __restrict__ Wl, int D, float* __restrict__ dy,
                                                            float* __restrict__ dWl, float* __restrict__ dbl)
{
    extern __shared__ float s_dx[];                                       // [D]
    const int64_t blk = blockIdx.x;
    if (blk < B) {
        const int64_t p = blk / n_walks;
        for (int d = threadIdx.x; d < D; d += 256) s_dx[d] = dX[p * D + d];
        __syncthreads();
        float* __restrict__ row = dy + blk * T * H2;
        for (int c = threadIdx.x; c < H2; c += 256) {
            float v = 0.f;
            for (int d = 0; d < D; ++d) v = fmaf(s_dx[d], Wl[(int64_t)d * H2 + c], v);
            for (int64_t t = 0; t < T; ++t) row[t * H2 + c] = (last_only && t != T - 1) ? 0.f : v;
        }
        return;
    }
    const int64_t e = (blk - B) * 256 + threadIdx.x, n_w = (int64_t)D * H2;
    const int64_t w_blocks = (n_w + 255) / 256;
    if (blk - B < w_blocks) {
        if (e < n_w && dWl) {
            const int d = (int)(e / H2), c = (int)(e - (int64_t)d * H2);
            float v = 0.f;
            for (int64_t p = 0; p < n_patches; ++p) v = fmaf(dX[p * D + d], s[p * H2 + c], v);
            dWl[e] = v;
        }
        return;
    }
    if (dbl)
        for (int d = threadIdx.x; d < D; d += 256) {
            float v = 0.f;
            for (int64_t p = 0; p < n_patches; ++p) v += dX[p * D + d];
            dbl[d] = (float)n_walks * v;
        }
}

extern "C" int sgnn_rows_gemm(const float* X, const int64_t* ids, int64_t ldx, int64_t R, int64_t K, const float* W0, const float* W1,
                              const float* b0, const float* b1, int64_t N, float* out, float* x_copy, void* stream)
{
    if (!X || !W0 || !out || R < 0 || K < 8 || K % 8 != 0 || N < 1 || ldx < K || ldx % 4 != 0) return SGNN_ERR_BAD_ARG;
    if (R == 0) return SGNN_OK;
    if ((R + 31) / 32 > 0x7fffffff || (N + 31) / 32 > 65535) return SGNN_ERR_BAD_ARG;
    const dim3 grid((unsigned)((R + 31) / 32), (unsigned)((N + 31) / 32), W1 ? 2u : 1u);
    if (K >= 128 && K % 32 == 0)
        hipLaunchKernelGGL(rows_gemm_ksplit_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, ids, ldx, R, (int)K, W0, W1, b0, b1, (int)N,
                           out, x_copy);
    else
        hipLaunchKernelGGL(rows_gemm_kernel, grid, dim3(64), 0, (hipStream_t)stream, X, ids, ldx, R, (int)K, W0, W1, b0, b1, (int)N, out,
                           x_copy);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_rows_gemm_nt(const float* A, int64_t R, int64_t K, const float* W0, const float* W1, int64_t N, float* out,
                                 void* stream)
{
    if (!A || !W0 || !W1 || !out || R < 0 || K < 16 || K % 16 != 0 || N < 1) return SGNN_ERR_BAD_ARG;
    if (R == 0) return SGNN_OK;
    if ((R + 31) / 32 > 0x7fffffff || (N + 31) / 32 > 65535) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(rows_gemm_nt_kernel, dim3((unsigned)((R + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       A, R, (int)K, W0, W1, (int)N, out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_lstm_tail_fwd(const float* y, int64_t n_patches, int64_t n_walks, int64_t T, int64_t H2, int last_only,
                                  const float* W_lin, const float* b_lin, int64_t D, float* s_out, float* X, void* stream)
{
    if (!y || !W_lin || !s_out || !X || n_patches < 0 || n_walks < 1 || T < 1 || H2 < 1 || H2 > 8192 || D < 1) return SGNN_ERR_BAD_ARG;
    if (n_patches == 0) return SGNN_OK;
    if (n_patches > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(lstm_tail_fwd_kernel, dim3((unsigned)n_patches), dim3(256), (size_t)H2 * 4, (hipStream_t)stream, y, T, (int)H2,
                       (int)n_walks, last_only, W_lin, b_lin, (int)D, s_out, X);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_lstm_tail_bwd(const float* dX, const float* s, int64_t n_patches, int64_t n_walks, int64_t T, int64_t H2,
                                  int last_only, const float* W_lin, int64_t D, float* dy, float* dW_lin, float* db_lin, void* stream)
{
    if (!dX || !s || !W_lin || !dy || n_patches < 0 || n_walks < 1 || T < 1 || H2 < 1 || D < 1 || D > 8192) return SGNN_ERR_BAD_ARG;
    if (n_patches == 0) return SGNN_OK;
    const int64_t B = n_patches * n_walks, blocks = B + (D * H2 + 255) / 256 + 1;
    if (blocks > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(lstm_tail_bwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)D * 4, (hipStream_t)stream, dX, s, n_patches, B, T,
                       (int)H2, (int)n_walks, last_only, W_lin, (int)D, dy, dW_lin, db_lin);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(lstm)
