// a12 update():  out = relu([x | aggr] W^T + b)   (reference SubGNN/subgraph_mpn.py:233-241, nn.Linear(2D, D) of
// subgraph_mpn.py:33) and its backward, for one row per component: x = the component embeddings (R, D), aggr = the
// aggregated messages (R, D), W (D, 2D) row-major, b (D).
//
// As torch ops the layer was cat (R x 2D written and read back) + a library GEMM whose 64-column output runs far from
// the chip's width + relu, and five more launches per direction in the backward: 47 us forward, 215 us forward +
// backward per layer at R = 50k, D = 64, six layers per pass (fused: 20 us and 96 us).  Here:
//   forward   one wavefront per (32 rows, 32 output features).  v_mfma_f32_32x32x2_f32 takes A as (row = lane % 32, k = lane / 32): the lane
//             with k-half 0 holds its row of x, the lane with k-half 1 the same row of aggr -- the contraction index
//             is simply walked as (half, position), the same permutation on the W side -- so every lane reads ONE
//             contiguous row of D floats straight into registers, no concatenation, no LDS staging.  Bias and relu are
//             applied to the accumulator fragment.
//   backward  dpre = grad_out * (out > 0) is formed on the fly in both kernels:
//             dx kernel   [grad_x | grad_aggr] = dpre W, one wavefront per (32 rows, 32 columns), contraction over the D outputs;
//             dw kernel   grad_W = dpre^T [x | aggr] and grad_b = column sums of dpre, contracted over the rows: one
//                         workgroup of four wavefronts per (block of UPD_ROWS rows, 32 output features), partial sums per block written
//                         to the workspace and added up in block order by a small second kernel -- a fixed order, so
//                         the gradients are bit-reproducible.
// HBM / cache traffic per layer: forward reads 2 R D and writes R D floats; backward reads 4 R D (+ the row blocks'
// second read in the dw kernel) and writes 2 R D.  All MFMA operands are fp32, accumulation fp32.
#include "common.h"

typedef float upd_f32x16 __attribute__((ext_vector_type(16)));

#define UPD_WAVE_ROWS 64        // rows one wavefront contracts in the weight-gradient kernel (128 / 64 / 32: 54 / 37 / 45 us at R = 50k, D = 64)
#define UPD_ROWS (4 * UPD_WAVE_ROWS)   // rows per partial sum (a workgroup of four wavefronts)
#define UPD_STEPS 8             // contraction steps of the weight gradient whose operands are in flight together
#define UPD_KC 64               // contraction positions held in registers at a time (per k-half)
#define UPD_SPLIT_BELOW 16384    // rows below which the output tiles of a row block go to separate wavefronts
#define UPD_KSPLIT_BELOW 4096    // rows below which a tile's contraction is split over the four wavefronts of a workgroup
#define UPD_WAVE_ROWS_SMALL 16   // rows per wavefront of the weight-gradient kernel for calls below UPD_KSPLIT_BELOW rows

// Several update layers of the same shape in ONE launch each way (the bodies of one message-passing layer: up to six channel
// sides of a batch-sized step): blockIdx.z = body, its operands from this table (n = 0: the launch's own pointer arguments).
#define UPD_MAX_BODIES 8
struct UpdMany {
    const float* x[UPD_MAX_BODIES]; const float* aggr[UPD_MAX_BODIES]; const float* W[UPD_MAX_BODIES]; const float* b[UPD_MAX_BODIES];
    float* out[UPD_MAX_BODIES]; float* aggr_sum[UPD_MAX_BODIES];
    const float* g[UPD_MAX_BODIES]; float* gx[UPD_MAX_BODIES]; float* gaggr[UPD_MAX_BODIES];
    float* gW[UPD_MAX_BODIES]; float* gb[UPD_MAX_BODIES]; float* pW[UPD_MAX_BODIES]; float* pb[UPD_MAX_BODIES];
    int n_chunks[UPD_MAX_BODIES];
    int n;
};

// accumulator element v of lane l: row 8 * (v / 4) + 4 * (l / 32) + v % 4, column l % 32
__device__ __forceinline__ int upd_acc_row(int v, int h) { return 8 * (v >> 2) + 4 * h + (v & 3); }

// SPLIT: one wavefront per (32 rows, 32 output features), blockIdx.y = the feature tile -- for calls with few rows (a
// batch: 448 rows of D = 128 were 14 wavefronts of 512 dependent MFMA + load steps each, 37 us; side by side 9 us);
// else one wavefront per 32 rows walks all tiles with its row fragment kept in registers (50k rows: 20 us, split 26).
template <int D, bool SPLIT>
__global__ __launch_bounds__(64) void update_fwd_kernel(const float* __restrict__ x, const float* __restrict__ aggr,
                                                        const float* __restrict__ W, const float* __restrict__ b,
                                                        int64_t R, float* __restrict__ out)
{
    constexpr int KC = D < UPD_KC ? D : UPD_KC;
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int64_t row = row0 + i < R ? row0 + i : R - 1;                 // (rows past the end: loaded, never stored)
    const float* __restrict__ src = (h ? aggr : x) + row * D;
    float a[KC];
    if (D <= UPD_KC) {
#pragma unroll
        for (int c = 0; c < KC / 4; ++c) {
            const float4 v = reinterpret_cast<const float4*>(src)[c];
            a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
        }
    }
    const int nt0 = SPLIT ? (int)blockIdx.y : 0, nt1 = SPLIT ? nt0 + 1 : D / 32;
    for (int nt = nt0; nt < nt1; ++nt) {
        upd_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ wrow = W + (int64_t)(nt * 32 + i) * (2 * D) + h * D;
#pragma unroll 1
        for (int kc = 0; kc < D; kc += KC) {
            float w[KC];
#pragma unroll
            for (int c = 0; c < KC / 4; ++c) {
                const float4 v = reinterpret_cast<const float4*>(wrow + kc)[c];
                w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
            }
            if (D > UPD_KC) {
#pragma unroll
                for (int c = 0; c < KC / 4; ++c) {
                    const float4 v = reinterpret_cast<const float4*>(src + kc)[c];
                    a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
                }
            }
#pragma unroll
            for (int s = 0; s < KC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w[s], acc, 0, 0, 0);
        }
        const int col = nt * 32 + i;
        const float bias = b ? b[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t r = row0 + upd_acc_row(v, h);
            if (r < R) out[r * D + col] = fmaxf(acc[v] + bias, 0.f);
        }
    }
}

// KSPLIT: batch-sized calls (a few hundred rows: R / 32 x D / 32 tiles do not fill the chip, and one wavefront per tile walks
// 2 D / 2 dependent MFMA + load steps: 12.5 us per call at R = 128, D = 128 -- 20 calls per step of the 4-layer HPO-METAB
// stand-in).  A workgroup of four wavefronts per tile, each contracting a quarter of the positions of both halves; the partial
// accumulators are added in wavefront order through LDS (a fixed order: bit-reproducible).
template <int D>
__global__ __launch_bounds__(256) void update_fwd_ksplit_kernel(const float* __restrict__ x_, const float* __restrict__ aggr_,
                                                                const float* __restrict__ W_, const float* __restrict__ b_,
                                                                int64_t R, float* __restrict__ out_, int n_chunks_,
                                                                float* __restrict__ aggr_sum_, const UpdMany M)
{
    const int z = blockIdx.z;
    const float* __restrict__ x = M.n ? M.x[z] : x_;
    const float* __restrict__ aggr = M.n ? M.aggr[z] : aggr_;
    const float* __restrict__ W = M.n ? M.W[z] : W_;
    const float* __restrict__ b = M.n ? M.b[z] : b_;
    float* __restrict__ out = M.n ? M.out[z] : out_;
    float* __restrict__ aggr_sum = M.n ? M.aggr_sum[z] : aggr_sum_;
    const int n_chunks = M.n ? M.n_chunks[z] : n_chunks_;
    constexpr int Q = D / 4;                                             // positions per wavefront (8 / 16 / 32)
    __shared__ float s_part[3 * 16 * 64];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int64_t row = row0 + i < R ? row0 + i : R - 1;
    const int nt = blockIdx.y;
    const float* __restrict__ src = (h ? aggr : x) + row * D + wave * Q;
    const float* __restrict__ wrow = W + (int64_t)(nt * 32 + i) * (2 * D) + h * D + wave * Q;
    float a[Q], w[Q];
#pragma unroll
    for (int c = 0; c < Q / 4; ++c) {
        const float4 v = reinterpret_cast<const float4*>(src)[c];
        const float4 u = reinterpret_cast<const float4*>(wrow)[c];
        a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
        w[4 * c] = u.x; w[4 * c + 1] = u.y; w[4 * c + 2] = u.z; w[4 * c + 3] = u.w;
    }
    if (n_chunks > 1) {
        // aggr arrives as the anchor-chunk partials of sgnn_mpn_fwd, (n_chunks, R, D): added here in chunk order (the sum the
        // caller used to make with a reduction launch per layer), and written out once for the backward pass.  A chunk's
        // Q / 4 loads are issued together (one memory latency per chunk, not per 16 bytes).
        if (h) {
            for (int k = 1; k < n_chunks; ++k) {
                float4 p[Q / 4];
#pragma unroll
                for (int c = 0; c < Q / 4; ++c) p[c] = reinterpret_cast<const float4*>(src + (int64_t)k * R * D)[c];
#pragma unroll
                for (int c = 0; c < Q / 4; ++c) { a[4 * c] += p[c].x; a[4 * c + 1] += p[c].y; a[4 * c + 2] += p[c].z; a[4 * c + 3] += p[c].w; }
            }
            if (aggr_sum && nt == 0 && row0 + i < R) {
#pragma unroll
                for (int c = 0; c < Q / 4; ++c)
                    reinterpret_cast<float4*>(aggr_sum + row * D + wave * Q)[c] = make_float4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
            }
        }
    }
    upd_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < Q; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w[s], acc, 0, 0, 0);
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) s_part[((wave - 1) * 16 + v) * 64 + lane] = acc[v];
    }
    __syncthreads();
    if (wave != 0) return;
    const int col = nt * 32 + i;
    const float bias = b ? b[col] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const float sum = ((acc[v] + s_part[v * 64 + lane]) + s_part[(16 + v) * 64 + lane]) + s_part[(32 + v) * 64 + lane];
        const int64_t r = row0 + upd_acc_row(v, h);
        if (r < R) out[r * D + col] = fmaxf(sum + bias, 0.f);
    }
}

// [grad_x | grad_aggr](r, c) = sum_n dpre(r, n) W(n, c): contraction index n walked as (half, position)
template <int D, bool SPLIT>
__global__ __launch_bounds__(64) void update_bwd_dx_kernel(const float* __restrict__ g_, const float* __restrict__ out_,
                                                           const float* __restrict__ W_, int64_t R,
                                                           float* __restrict__ gx_, float* __restrict__ gaggr_, const UpdMany M)
{
    const int z = blockIdx.z;
    const float* __restrict__ g = M.n ? M.g[z] : g_;
    if (!g) return;                                                      // (a body whose output received no gradient)
    const float* __restrict__ out = M.n ? M.out[z] : out_;
    const float* __restrict__ W = M.n ? M.W[z] : W_;
    float* __restrict__ gx = M.n ? M.gx[z] : gx_;
    float* __restrict__ gaggr = M.n ? M.gaggr[z] : gaggr_;
    constexpr int H = D / 2;                                             // contraction positions per k-half
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int64_t row = row0 + i < R ? row0 + i : R - 1;
    float a[H];
#pragma unroll
    for (int c = 0; c < H / 4; ++c) {
        const float4 gv = reinterpret_cast<const float4*>(g + row * D + h * H)[c];
        const float4 ov = reinterpret_cast<const float4*>(out + row * D + h * H)[c];
        a[4 * c] = ov.x > 0.f ? gv.x : 0.f; a[4 * c + 1] = ov.y > 0.f ? gv.y : 0.f;
        a[4 * c + 2] = ov.z > 0.f ? gv.z : 0.f; a[4 * c + 3] = ov.w > 0.f ? gv.w : 0.f;
    }
    const int ct0 = SPLIT ? (int)blockIdx.y : 0, ct1 = SPLIT ? ct0 + 1 : 2 * D / 32;     // (SPLIT: see update_fwd_kernel)
    for (int ct = ct0; ct < ct1; ++ct) {
        float* __restrict__ dst = ct * 32 < D ? gx : gaggr;
        if (!dst) continue;
        upd_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ wcol = W + (int64_t)(h * H) * (2 * D) + ct * 32 + i;     // W(h * H + s, ct * 32 + i)
#pragma unroll
        for (int s = 0; s < H; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wcol[(int64_t)s * (2 * D)], acc, 0, 0, 0);
        const int col = (ct * 32) % D + i;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t r = row0 + upd_acc_row(v, h);
            if (r < R) dst[r * D + col] = acc[v];
        }
    }
}

// partial grad_W(n, c) = sum over the block's rows of dpre(r, n) [x | aggr](r, c); partial grad_b(n) = sum dpre(r, n).
// blockIdx.x = row block, blockIdx.y = tile of 32 output features n.  The rows are the contraction: step s covers
// rows r0 + 2 s + (lane / 32).
template <int D, int WR = UPD_WAVE_ROWS>
__global__ __launch_bounds__(256) void update_bwd_dw_kernel(const float* __restrict__ g_, const float* __restrict__ out_,
                                                           const float* __restrict__ x_, const float* __restrict__ aggr_,
                                                           int64_t R, float* __restrict__ pW_, float* __restrict__ pb_, const UpdMany M)
{
    const int z = blockIdx.z;
    const float* __restrict__ g = M.n ? M.g[z] : g_;
    if (!g) return;
    const float* __restrict__ out = M.n ? M.out[z] : out_;
    const float* __restrict__ x = M.n ? M.x[z] : x_;
    const float* __restrict__ aggr = M.n ? M.aggr[z] : aggr_;
    float* __restrict__ pW = M.n ? M.pW[z] : pW_;
    float* __restrict__ pb = M.n ? M.pb[z] : pb_;
    constexpr int CT = 2 * D / 32;                                       // column tiles of [x | aggr]
    __shared__ float s_acc[(CT * 16 + 1) * 64];
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r0 = (int64_t)blockIdx.x * (4 * WR) + (int64_t)wave * WR;
    const int nt = blockIdx.y;
    upd_f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    float bsum = 0.f;
    const int64_t r_end = r0 + WR < R ? r0 + WR : R;
    // UPD_STEPS contraction steps (2 rows each) at a time: all their operands are requested together, then the MFMAs
    // run back to back -- one step at a time the wavefront waited a memory round trip per step (154 us at R = 50k)
    for (int64_t rb = r0; rb < r_end; rb += 2 * UPD_STEPS) {
        float av[UPD_STEPS], bv[UPD_STEPS][CT];
#pragma unroll
        for (int u = 0; u < UPD_STEPS; ++u) {
            const int64_t r = rb + 2 * u + h;
            const bool live = r < r_end;
            const int64_t rr = live ? r : (r_end > 0 ? r_end - 1 : 0);
            const float ov = out[rr * D + nt * 32 + i];
            const float gv = g[rr * D + nt * 32 + i];
            av[u] = (live && ov > 0.f) ? gv : 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float* __restrict__ src = c * 32 < D ? x : aggr;
                bv[u][c] = src[rr * D + (c * 32) % D + i];
            }
        }
#pragma unroll
        for (int u = 0; u < UPD_STEPS; ++u) {
            bsum += av[u];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][c], acc[c], 0, 0, 0);
        }
    }
    bsum += __shfl_xor(bsum, 32, 64);                                    // the two row halves
    // the four wavefronts' sums are added in wavefront order through LDS; the last one holds the workgroup's partial
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int idx = (c * 16 + v) * 64 + lane;
                    if (w > 0) acc[c][v] += s_acc[idx];
                    if (w < 3) s_acc[idx] = acc[c][v];
                }
            if (w > 0) bsum += s_acc[CT * 16 * 64 + lane];
            if (w < 3) s_acc[CT * 16 * 64 + lane] = bsum;
        }
        __syncthreads();
    }
    if (wave != 3) return;
    float* __restrict__ dst = pW + (int64_t)blockIdx.x * D * (2 * D);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v)
            dst[(int64_t)(nt * 32 + upd_acc_row(v, h)) * (2 * D) + c * 32 + i] = acc[c][v];
    if (h == 0) pb[(int64_t)blockIdx.x * D + nt * 32 + i] = bsum;
}

// out[j] = sum over blocks of part[block * n + j], in a fixed order: a workgroup owns 64 outputs; its four wavefronts
// each add up a contiguous quarter of the blocks, and the quarters are added in order through LDS
__global__ __launch_bounds__(256) void update_reduce_kernel(const float* __restrict__ part, int64_t n_blocks, int64_t n,
                                                            float* __restrict__ out, unsigned first_groups,
                                                            const float* __restrict__ part2, int64_t n2, float* __restrict__ out2,
                                                            const UpdMany M)
{
    if (M.n) {
        const int z = blockIdx.z;
        if (!M.g[z]) return;
        part = M.pW[z]; out = M.gW[z]; part2 = M.pb[z]; out2 = M.gb[z];
    }
    __shared__ float s_q[4 * 64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    unsigned group = blockIdx.x;
    if (group >= first_groups) { group -= first_groups; part = part2; n = n2; out = out2; }       // the second array's workgroups
    const int64_t j = (int64_t)group * 64 + o;
    const int64_t per = (n_blocks + 3) / 4;
    const int64_t b0 = q * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    float s = 0.f;
    if (j < n) {
#pragma unroll 8
        for (int64_t b = b0; b < b1; ++b) s += part[b * n + j];
    }
    s_q[q * 64 + o] = s;
    __syncthreads();
    if (q == 0 && j < n) out[j] = ((s_q[o] + s_q[64 + o]) + s_q[128 + o]) + s_q[192 + o];
}

static const UpdMany UPD_NONE = {};

static int update_fwd_run(const float* x, const float* aggr, int n_chunks, const float* W, const float* b, int64_t R, int64_t D,
                          float* out, float* aggr_sum, void* stream);

extern "C" int sgnn_update_fwd(const float* x, const float* aggr, const float* W, const float* b, int64_t R, int64_t D,
                               float* out, void* stream)
{
    return update_fwd_run(x, aggr, 1, W, b, R, D, out, nullptr, stream);
}

extern "C" int64_t sgnn_update_fwd_chunks_max_rows(void) { return UPD_KSPLIT_BELOW - 1; }

extern "C" int sgnn_update_fwd_chunks(const float* x, const float* aggr_chunks, int64_t n_chunks, const float* W, const float* b,
                                      int64_t R, int64_t D, float* out, float* aggr_sum, void* stream)
{
    if (n_chunks < 1 || n_chunks > 4096 || (n_chunks > 1 && R >= UPD_KSPLIT_BELOW)) return SGNN_ERR_BAD_ARG;
    return update_fwd_run(x, aggr_chunks, (int)n_chunks, W, b, R, D, out, aggr_sum, stream);
}

static int update_fwd_run(const float* x, const float* aggr, int n_chunks, const float* W, const float* b, int64_t R, int64_t D,
                          float* out, float* aggr_sum, void* stream)
{
    if (!x || !aggr || !W || !out || R < 0) return SGNN_ERR_BAD_ARG;
    if (D != 32 && D != 64 && D != 128) return SGNN_ERR_UNSUPPORTED_D;
    if (R == 0) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((R + 31) / 32);
    // few rows: the feature tiles side by side (more wavefronts than CUs only from ~8k rows on)
#define UPD_LAUNCH_FWD(DD) do { if (R < UPD_KSPLIT_BELOW) hipLaunchKernelGGL((update_fwd_ksplit_kernel<DD>), dim3(grid, DD / 32), dim3(256), 0, st, x, aggr, W, b, R, out, n_chunks, aggr_sum, UPD_NONE); \
                                else if (split) hipLaunchKernelGGL((update_fwd_kernel<DD, true>), dim3(grid, DD / 32), dim3(64), 0, st, x, aggr, W, b, R, out); \
                                else hipLaunchKernelGGL((update_fwd_kernel<DD, false>), dim3(grid), dim3(64), 0, st, x, aggr, W, b, R, out); } while (0)
    const bool split = R < UPD_SPLIT_BELOW;
    if (D == 32) UPD_LAUNCH_FWD(32); else if (D == 64) UPD_LAUNCH_FWD(64); else UPD_LAUNCH_FWD(128);
#undef UPD_LAUNCH_FWD
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// rows per partial sum: batch-sized calls (a few hundred rows) take 16 rows per wavefront -- with 64 a call of 128 rows was one
// row block: two busy wavefronts per feature tile walking 32 dependent steps (28 us; 8 such calls per HPO-METAB stand-in step)
static inline int64_t upd_block_rows(int64_t R) { return 4 * (R < UPD_KSPLIT_BELOW ? UPD_WAVE_ROWS_SMALL : UPD_WAVE_ROWS); }

extern "C" int64_t sgnn_update_bwd_workspace_bytes(int64_t R, int64_t D)
{
    const int64_t nb = (R + upd_block_rows(R) - 1) / upd_block_rows(R);
    return nb * (D * 2 * D + D) * 4 + 64;
}

extern "C" int sgnn_update_bwd(const float* grad_out, const float* out, const float* x, const float* aggr, const float* W,
                               int64_t R, int64_t D, float* grad_x, float* grad_aggr, float* grad_W, float* grad_b,
                               void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!grad_out || !out || !W || R < 0) return SGNN_ERR_BAD_ARG;
    if ((grad_W || grad_b) && (!x || !aggr)) return SGNN_ERR_BAD_ARG;
    if (D != 32 && D != 64 && D != 128) return SGNN_ERR_UNSUPPORTED_D;
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) {
        hipError_t e = hipSuccess;
        if (grad_W) e = hipMemsetAsync(grad_W, 0, (size_t)(D * 2 * D * 4), st);
        if (e == hipSuccess && grad_b) e = hipMemsetAsync(grad_b, 0, (size_t)(D * 4), st);
        if (e != hipSuccess) { sgnn_set_last_error(e); return SGNN_ERR_LAUNCH; }
        return SGNN_OK;
    }
    if (grad_x || grad_aggr) {
        const unsigned grid = (unsigned)((R + 31) / 32);
#define UPD_LAUNCH_DX(DD) do { if (split) hipLaunchKernelGGL((update_bwd_dx_kernel<DD, true>), dim3(grid, 2 * DD / 32), dim3(64), 0, st, grad_out, out, W, R, grad_x, grad_aggr, UPD_NONE); \
                               else hipLaunchKernelGGL((update_bwd_dx_kernel<DD, false>), dim3(grid), dim3(64), 0, st, grad_out, out, W, R, grad_x, grad_aggr, UPD_NONE); } while (0)
        const bool split = R < UPD_SPLIT_BELOW;
        if (D == 32) UPD_LAUNCH_DX(32); else if (D == 64) UPD_LAUNCH_DX(64); else UPD_LAUNCH_DX(128);
#undef UPD_LAUNCH_DX
        SGNN_CHECK_LAUNCH();
    }
    if (grad_W || grad_b) {
        if (!workspace || workspace_bytes < sgnn_update_bwd_workspace_bytes(R, D)) return SGNN_ERR_BAD_ARG;
        const int64_t nb = (R + upd_block_rows(R) - 1) / upd_block_rows(R);
        float* pW = (float*)workspace;
        float* pb = pW + nb * D * 2 * D;
        const dim3 grid((unsigned)nb, (unsigned)(D / 32));
        if (R < UPD_KSPLIT_BELOW) {
            if (D == 32) hipLaunchKernelGGL((update_bwd_dw_kernel<32, UPD_WAVE_ROWS_SMALL>), grid, dim3(256), 0, st, grad_out, out, x, aggr, R, pW, pb, UPD_NONE);
            else if (D == 64) hipLaunchKernelGGL((update_bwd_dw_kernel<64, UPD_WAVE_ROWS_SMALL>), grid, dim3(256), 0, st, grad_out, out, x, aggr, R, pW, pb, UPD_NONE);
            else hipLaunchKernelGGL((update_bwd_dw_kernel<128, UPD_WAVE_ROWS_SMALL>), grid, dim3(256), 0, st, grad_out, out, x, aggr, R, pW, pb, UPD_NONE);
        }
        else if (D == 32) hipLaunchKernelGGL(update_bwd_dw_kernel<32>, grid, dim3(256), 0, st, grad_out, out, x, aggr, R, pW, pb, UPD_NONE);
        else if (D == 64) hipLaunchKernelGGL(update_bwd_dw_kernel<64>, grid, dim3(256), 0, st, grad_out, out, x, aggr, R, pW, pb, UPD_NONE);
        else hipLaunchKernelGGL(update_bwd_dw_kernel<128>, grid, dim3(256), 0, st, grad_out, out, x, aggr, R, pW, pb, UPD_NONE);
        SGNN_CHECK_LAUNCH();
        // one launch for both: workgroups [0, nW) own 64 elements of grad_W each, the rest 64 of grad_b
        const unsigned nW = grad_W ? (unsigned)((D * 2 * D + 63) / 64) : 0u, nB = grad_b ? (unsigned)((D + 63) / 64) : 0u;
        hipLaunchKernelGGL(update_reduce_kernel, dim3(nW + nB), dim3(256), 0, st, pW, nb, D * 2 * D, grad_W, nW, pb, D, grad_b, UPD_NONE);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

// ---- the update layers of one message-passing layer's bodies in one launch each way ------------------------------------------
// n bodies of the same (R, D), R below UPD_KSPLIT_BELOW (the batch-sized shape): forward 1 launch, backward 3 (dx, dW partials,
// reduce) instead of n and 3 n.  Pointer tables are HOST arrays of DEVICE pointers.
extern "C" int64_t sgnn_update_many_max_bodies(void) { return UPD_MAX_BODIES; }

extern "C" int sgnn_update_fwd_many(int64_t n, const float* const* x, const float* const* aggr_chunks, const int64_t* n_chunks,
                                    const float* const* W, const float* const* b, int64_t R, int64_t D, float* const* out,
                                    float* const* aggr_sum, void* stream)
{
    if (n < 1 || n > UPD_MAX_BODIES || !x || !aggr_chunks || !n_chunks || !W || !b || !out || !aggr_sum || R < 0) return SGNN_ERR_BAD_ARG;
    if (D != 32 && D != 64 && D != 128) return SGNN_ERR_UNSUPPORTED_D;
    if (R >= UPD_KSPLIT_BELOW) return SGNN_ERR_BAD_ARG;
    if (R == 0) return SGNN_OK;
    UpdMany M = {};
    M.n = (int)n;
    for (int k = 0; k < n; ++k) {
        if (!x[k] || !aggr_chunks[k] || !W[k] || !out[k] || n_chunks[k] < 1 || n_chunks[k] > 4096 || (n_chunks[k] > 1 && !aggr_sum[k]))
            return SGNN_ERR_BAD_ARG;
        M.x[k] = x[k]; M.aggr[k] = aggr_chunks[k]; M.W[k] = W[k]; M.b[k] = b[k]; M.out[k] = out[k]; M.aggr_sum[k] = aggr_sum[k];
        M.n_chunks[k] = (int)n_chunks[k];
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((R + 31) / 32), (unsigned)(D / 32), (unsigned)n);
    if (D == 32) hipLaunchKernelGGL((update_fwd_ksplit_kernel<32>), grid, dim3(256), 0, st, nullptr, nullptr, nullptr, nullptr, R, nullptr, 1, nullptr, M);
    else if (D == 64) hipLaunchKernelGGL((update_fwd_ksplit_kernel<64>), grid, dim3(256), 0, st, nullptr, nullptr, nullptr, nullptr, R, nullptr, 1, nullptr, M);
    else hipLaunchKernelGGL((update_fwd_ksplit_kernel<128>), grid, dim3(256), 0, st, nullptr, nullptr, nullptr, nullptr, R, nullptr, 1, nullptr, M);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

/* grad_out[k] NULL: body k received no gradient and is skipped (its outputs are not written).  For the others grad_x[k] /
 * grad_aggr[k] may be NULL; grad_W[k] and grad_b[k] are both written.  workspace: n * sgnn_update_bwd_workspace_bytes(R, D). */
extern "C" int sgnn_update_bwd_many(int64_t n, const float* const* grad_out, const float* const* out, const float* const* x,
                                    const float* const* aggr, const float* const* W, int64_t R, int64_t D, float* const* grad_x,
                                    float* const* grad_aggr, float* const* grad_W, float* const* grad_b, void* workspace,
                                    int64_t workspace_bytes, void* stream)
{
    if (n < 1 || n > UPD_MAX_BODIES || !grad_out || !out || !x || !aggr || !W || !grad_x || !grad_aggr || !grad_W || !grad_b || R < 1)
        return SGNN_ERR_BAD_ARG;
    if (D != 32 && D != 64 && D != 128) return SGNN_ERR_UNSUPPORTED_D;
    if (R >= UPD_KSPLIT_BELOW) return SGNN_ERR_BAD_ARG;
    const int64_t per = sgnn_update_bwd_workspace_bytes(R, D);
    if (!workspace || workspace_bytes < n * per) return SGNN_ERR_BAD_ARG;
    const int64_t nb = (R + upd_block_rows(R) - 1) / upd_block_rows(R);
    UpdMany M = {};
    M.n = (int)n;
    bool any = false;
    for (int k = 0; k < n; ++k) {
        M.g[k] = grad_out[k];
        if (!grad_out[k]) continue;
        if (!out[k] || !x[k] || !aggr[k] || !W[k] || !grad_W[k] || !grad_b[k]) return SGNN_ERR_BAD_ARG;
        any = true;
        M.out[k] = const_cast<float*>(out[k]); M.x[k] = x[k]; M.aggr[k] = aggr[k]; M.W[k] = W[k];
        M.gx[k] = grad_x[k]; M.gaggr[k] = grad_aggr[k]; M.gW[k] = grad_W[k]; M.gb[k] = grad_b[k];
        M.pW[k] = (float*)((char*)workspace + k * per);
        M.pb[k] = M.pW[k] + nb * D * 2 * D;
    }
    if (!any) return SGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        const dim3 grid((unsigned)((R + 31) / 32), (unsigned)(2 * D / 32), (unsigned)n);
        if (D == 32) hipLaunchKernelGGL((update_bwd_dx_kernel<32, true>), grid, dim3(64), 0, st, nullptr, nullptr, nullptr, R, nullptr, nullptr, M);
        else if (D == 64) hipLaunchKernelGGL((update_bwd_dx_kernel<64, true>), grid, dim3(64), 0, st, nullptr, nullptr, nullptr, R, nullptr, nullptr, M);
        else hipLaunchKernelGGL((update_bwd_dx_kernel<128, true>), grid, dim3(64), 0, st, nullptr, nullptr, nullptr, R, nullptr, nullptr, M);
        SGNN_CHECK_LAUNCH();
    }
    {
        const dim3 grid((unsigned)nb, (unsigned)(D / 32), (unsigned)n);
        if (D == 32) hipLaunchKernelGGL((update_bwd_dw_kernel<32, UPD_WAVE_ROWS_SMALL>), grid, dim3(256), 0, st, nullptr, nullptr, nullptr, nullptr, R, nullptr, nullptr, M);
        else if (D == 64) hipLaunchKernelGGL((update_bwd_dw_kernel<64, UPD_WAVE_ROWS_SMALL>), grid, dim3(256), 0, st, nullptr, nullptr, nullptr, nullptr, R, nullptr, nullptr, M);
        else hipLaunchKernelGGL((update_bwd_dw_kernel<128, UPD_WAVE_ROWS_SMALL>), grid, dim3(256), 0, st, nullptr, nullptr, nullptr, nullptr, R, nullptr, nullptr, M);
        SGNN_CHECK_LAUNCH();
    }
    const unsigned nW = (unsigned)((D * 2 * D + 63) / 64), nB = (unsigned)((D + 63) / 64);
    hipLaunchKernelGGL(update_reduce_kernel, dim3(nW + nB, 1, (unsigned)n), dim3(256), 0, st, nullptr, nb, D * 2 * D, nullptr, nW, nullptr, D, nullptr, M);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(update)
