// Adam on one large parameter (the (N+1, D) embedding table: 256 MB at N = 1M, D = 64) in one pass over memory:
// reference SubGNN/SubGNN.py:1156-1161 (torch.optim.Adam on all parameters) + the gradient clipping of the caller
// (train_config.py: Trainer(gradient_clip_val)).  As library calls the table costs a multiply by the clip coefficient
// (read + write of the gradient), the fused multi-tensor Adam in four chunked launches and, next pass, a zero fill of
// the gradient buffer: 7 + 2 + 1 streams of 256 MB.  Here: p, g, m, v read once, the clip coefficient (a device scalar:
// no host round trip) applied on the fly, p, m, v written, and optionally the gradient zeroed in the same pass -- 8
// streams, one launch.  Update rule = torch.optim.Adam (no weight decay, no amsgrad):
//   m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
#include "common.h"

__global__ __launch_bounds__(256) void adam_step_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                        float4* __restrict__ v, int64_t n4, float b1, float b2, float eps,
                                                        float step_size, float rsqrt_bc2, const float* __restrict__ grad_scale,
                                                        int zero_grad, const int64_t* __restrict__ step_counter, float lr)
{
    const float gs = grad_scale ? grad_scale[0] : 1.f;
    if (step_counter) {                       // a recorded step: the count lives on the device, the corrections follow it
        const double t = (double)step_counter[0];
        step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
        rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = G[k] * gs;
            M[k] = M[k] + (1.f - b1) * (gk - M[k]);
            V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
            P[k] -= step_size * M[k] / (sqrtf(V[k]) * rsqrt_bc2 + eps);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (zero_grad) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__global__ void adam_step_tail_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                      int64_t from, int64_t n, float b1, float b2, float eps, float step_size, float rsqrt_bc2,
                                      const float* __restrict__ grad_scale, int zero_grad, const int64_t* __restrict__ step_counter,
                                      float lr)
{
    const int64_t i = from + threadIdx.x;
    if (i >= n) return;
    if (step_counter) {
        const double t = (double)step_counter[0];
        step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
        rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    const float gk = g[i] * (grad_scale ? grad_scale[0] : 1.f);
    m[i] = m[i] + (1.f - b1) * (gk - m[i]);
    v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
    p[i] -= step_size * m[i] / (sqrtf(v[i]) * rsqrt_bc2 + eps);
    if (zero_grad) g[i] = 0.f;
}

__global__ void adam_count_kernel(int64_t* counter) { counter[0] += 1; }

static int adam_run(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                    float eps, int64_t step, int64_t* step_counter, const float* grad_scale, int zero_grad, void* stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || (!step_counter && step < 1)) return SGNN_ERR_BAD_ARG;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return SGNN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (step_counter) {
        hipLaunchKernelGGL(adam_count_kernel, dim3(1), dim3(1), 0, st, step_counter);
        SGNN_CHECK_LAUNCH();
        step = 1;
    }
    if (n == 0) return SGNN_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const int64_t n4 = n / 4;
    if (n4 > 0) {
        hipLaunchKernelGGL(adam_step_kernel, dim3(sgnn_grid_for(n4, 256)), dim3(256), 0, st, (float4*)param, (float4*)grad,
                           (float4*)exp_avg, (float4*)exp_avg_sq, n4, beta1, beta2, eps, step_size, rsqrt_bc2, grad_scale, zero_grad,
                           (const int64_t*)step_counter, lr);
        SGNN_CHECK_LAUNCH();
    }
    if (n4 * 4 < n) {
        hipLaunchKernelGGL(adam_step_tail_kernel, dim3(1), dim3(64), 0, st, param, grad, exp_avg, exp_avg_sq, n4 * 4, n, beta1,
                           beta2, eps, step_size, rsqrt_bc2, grad_scale, zero_grad, (const int64_t*)step_counter, lr);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

extern "C" int sgnn_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, int64_t step, const float* grad_scale, int zero_grad, void* stream)
{
    return adam_run(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, nullptr, grad_scale, zero_grad, stream);
}

extern "C" int sgnn_adam_step_counted(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                      float beta1, float beta2, float eps, int64_t* step_counter, const float* grad_scale,
                                      int zero_grad, void* stream)
{
    if (!step_counter) return SGNN_ERR_BAD_ARG;
    return adam_run(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, 0, step_counter, grad_scale, zero_grad, stream);
}

// ---- clip_grad_norm_'s coefficient without a pass of library launches over the large gradient ----------------------------
// torch.nn.utils.clip_grad_norm_ (the caller's gradient clipping, train_config.py: Trainer(gradient_clip_val)): total = 2-norm
// of all gradients' 2-norms, coefficient = min(1, max_norm / (total + 1e-6)).  The table's gradient (256 MB) took four
// chunked multi-tensor launches + a clean-up, and the coefficient six scalar launches.  Here: per-workgroup sums of squares
// of the large gradient in one launch (fixed order inside a workgroup), then one wavefront adds them and the squared norms
// of the other gradients in a fixed order and writes the coefficient -- a device scalar for sgnn_adam_step.
#define SUMSQ_BLOCKS 2048

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float4* __restrict__ x, int64_t n4, const float* __restrict__ tail,
                                                            int64_t n_tail, float* __restrict__ partial)
{
    __shared__ float sh[256];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = x[i];
        a0 = fmaf(v.x, v.x, a0); a1 = fmaf(v.y, v.y, a1); a2 = fmaf(v.z, v.z, a2); a3 = fmaf(v.w, v.w, a3);
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < n_tail) a0 = fmaf(tail[threadIdx.x], tail[threadIdx.x], a0);
    sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) partial[blockIdx.x] = v;
    }
}

__global__ __launch_bounds__(64) void clip_coefficient_kernel(const float* __restrict__ partial, int64_t n_partial,
                                                              const float* __restrict__ other_norms, int64_t n_other, float max_norm,
                                                              float* __restrict__ coef, float* __restrict__ total_norm)
{
    double acc = 0.0;                                          // 2048 partials of ~1e0..1e3: double keeps the order harmless
    for (int64_t k = threadIdx.x; k < n_partial; k += 64) acc += (double)partial[k];
    for (int64_t k = threadIdx.x; k < n_other; k += 64) acc += (double)other_norms[k] * (double)other_norms[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(acc);
        if (total_norm) total_norm[0] = total;
        coef[0] = fminf(max_norm / (total + 1e-6f), 1.f);
    }
}

extern "C" int64_t sgnn_grad_sumsq_partials(void) { return SUMSQ_BLOCKS; }

extern "C" int sgnn_grad_sumsq(const float* grad, int64_t n, float* partial, void* stream)
{
    if (!grad || !partial || n < 0 || (((uintptr_t)grad) & 15) != 0) return SGNN_ERR_BAD_ARG;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const float4*)grad, n4,
                       grad + n4 * 4, n - n4 * 4, partial);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_clip_coefficient(const float* partial, int64_t n_partial, const float* other_norms, int64_t n_other,
                                     float max_norm, float* coef, float* total_norm, void* stream)
{
    if (!coef || n_partial < 0 || n_other < 0 || (n_partial && !partial) || (n_other && !other_norms)) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(clip_coefficient_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, n_partial, other_norms, n_other,
                       max_norm, coef, total_norm);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}


// ---- the whole optimizer tail in two launches -----------------------------------------------------------------------------
// clip_grad_norm_ + Adam over EVERY parameter (train_config.py: Trainer(gradient_clip_val); SubGNN/SubGNN.py:1156-1161).  As
// library calls a batch-sized step paid ~12 launches and ~155 us here (a multi-tensor norm + clean-up + stack, the coefficient,
// a multi-tensor multiply, torch's fused Adam in two launches of 40-50 us each -- it evaluates pow() in double per thread --
// a count kernel and the table's pass) of a 1.0-2.9 ms step.  Here: launch 1 = per-workgroup sums of squares of every gradient
// (+ the device step counts advance), launch 2 = every workgroup adds the partials in one fixed order (the same value in all
// of them), forms the coefficient and updates its chunk.  Tensors travel as kernel arguments (pointers of gradients change from
// step to step in eager mode and are frozen into a recorded step), OPT_MAXT per launch.
#define OPT_MAXT 72
#define OPT_CHUNK 4096                    // floats per workgroup iteration: 256 lanes x 4 float4
#define OPT_MAX_BLOCKS 2048               // per tensor; larger tensors stride

struct OptTensors {
    float* p[OPT_MAXT]; float* g[OPT_MAXT]; float* m[OPT_MAXT]; float* v[OPT_MAXT];
    long long n[OPT_MAXT];
    int blk[OPT_MAXT + 1];                // first workgroup of tensor t inside this launch
    int zero[OPT_MAXT];
    int slot[OPT_MAXT];                   // which device step count belongs to tensor t
    int count;
    // one tensor of the launch may be a table of rows with a per-row "ever had a gradient" byte: a row whose gradient is all
    // zero and that never had one has m = v = 0, so Adam leaves it exactly as it is (update = step_size * 0 / (0 + eps)) --
    // it costs the read of its gradient instead of four reads and three or four writes.  A shard's subgraphs, border sets and
    // anchor patches reach ~40 % of a million-node table's rows (7 % for one of 8 strong-scaling shards), the same ones every pass.
    int rows_t;                           // index of that tensor in this launch, -1 = none
    int row_lanes;                        // float4 lanes per row: a power of two <= 64
    unsigned char* seen;
};

__device__ __forceinline__ int opt_find(const OptTensors& T, int b)
{
    int lo = 0, hi = T.count - 1;         // largest t with blk[t] <= b
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (T.blk[mid] <= b) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ __launch_bounds__(256) void optim_sumsq_kernel(const OptTensors T, float* __restrict__ partial, int64_t* __restrict__ counters)
{
    __shared__ float sh[256];
    const int b = blockIdx.x, t = opt_find(T, b), nb = T.blk[t + 1] - T.blk[t];
    const float* __restrict__ g = T.g[t];
    const int64_t n = T.n[t];
    if (counters && b == 0 && (int)threadIdx.x < T.count) counters[T.slot[threadIdx.x]] += 1;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const float4* __restrict__ g4 = (const float4*)g;
        for (int64_t i = (int64_t)(b - T.blk[t]) * 256 + threadIdx.x; i < n4; i += (int64_t)nb * 256) {
            const float4 x = g4[i];
            a0 = fmaf(x.x, x.x, a0); a1 = fmaf(x.y, x.y, a1); a2 = fmaf(x.z, x.z, a2); a3 = fmaf(x.w, x.w, a3);
        }
        if (b == T.blk[t] && (int64_t)threadIdx.x < n - n4 * 4) { const float x = g[n4 * 4 + threadIdx.x]; a0 = fmaf(x, x, a0); }
    } else {
        for (int64_t i = (int64_t)(b - T.blk[t]) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) { const float x = g[i]; a0 = fmaf(x, x, a0); }
    }
    sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) partial[b] = v;
    }
}

#ifndef OPT_ROW_SLICES
#define OPT_ROW_SLICES 2
#endif
__global__ __launch_bounds__(256) void optim_adam_kernel(const OptTensors T, const float* __restrict__ partial, int n_partial, float max_norm,
                                                         float* __restrict__ coef_out, float lr, float b1, float b2, float eps,
                                                         long long host_step, const int64_t* __restrict__ counters)
{
    __shared__ double shd[256];
    const int b = blockIdx.x, t = opt_find(T, b), nb = T.blk[t + 1] - T.blk[t];
    float gs = 1.f;
    if (max_norm > 0.f) {                 // the same additions in the same order in every workgroup: one coefficient for all
        double acc = 0.0;
        for (int k = threadIdx.x; k < n_partial; k += 256) acc += (double)partial[k];
        shd[threadIdx.x] = acc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) shd[threadIdx.x] += shd[threadIdx.x + w];
            __syncthreads();
        }
        const float total = (float)sqrt(shd[0]);
        gs = fminf(max_norm / (total + 1e-6f), 1.f);
        if (coef_out && b == 0 && threadIdx.x == 0) { coef_out[0] = gs; coef_out[1] = total; }
    }
    const double st = counters ? (double)counters[T.slot[t]] : (double)host_step;
    const float step_size = (float)((double)lr / (1.0 - pow((double)b1, st)));
    const float rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, st)));
    float* __restrict__ p = T.p[t]; float* __restrict__ g = T.g[t]; float* __restrict__ m = T.m[t]; float* __restrict__ v = T.v[t];
    const int64_t n = T.n[t];
    const int zero = T.zero[t];
    if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        float4* p4 = (float4*)p; float4* g4 = (float4*)g; float4* m4 = (float4*)m; float4* v4 = (float4*)v;
        if (t == T.rows_t) {
            // (n = rows * 4 L and a wavefront's 64 consecutive float4 start at a multiple of 64: a row's lanes share a wavefront)
            const int L = T.row_lanes, lane = threadIdx.x & 63, sh = 31 - __builtin_clz(L);
            const unsigned long long rowmask = (L == 64 ? ~0ull : ((1ull << L) - 1ull)) << (lane & ~(L - 1));
            unsigned char* __restrict__ seen = T.seen;
            // OPT_ROW_SLICES row slices per thread and iteration (round 6), all loads of an iteration issued before the first is used: the
            // gradient slices first, then -- WITHOUT a branch: a lane whose row is skipped reads slot 0 of the tensor, a cached
            // line -- parameter and both moments of both slices.  One slice per iteration with its p / m / v loads behind a
            // branch on the gradient was two dependent memory round trips per 16 bytes of every stream: 2.8 TB/s.
            const int64_t stride = (int64_t)nb * 256;
            constexpr int U = OPT_ROW_SLICES;
            for (int64_t i0 = (int64_t)(b - T.blk[t]) * 256 + threadIdx.x; i0 < n4; i0 += U * stride) {
                float4 gg[U];
                int64_t idx[U];
                bool in[U], any[U], was[U], act[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    idx[u] = i0 + u * stride;
                    in[u] = idx[u] < n4;                         // (wave-uniform: n4 is a multiple of 64 here)
                    gg[u] = in[u] ? g4[idx[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool nz = gg[u].x != 0.f || gg[u].y != 0.f || gg[u].z != 0.f || gg[u].w != 0.f;
                    any[u] = (__ballot(nz) & rowmask) != 0ull;
                    was[u] = in[u] ? seen[idx[u] >> sh] != 0 : false;
                    act[u] = in[u] && (any[u] || was[u]);
                }
                float4 pp[U], mm[U], vv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t j = act[u] ? idx[u] : 0;
                    pp[u] = p4[j]; mm[u] = m4[j]; vv[u] = v4[j];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (act[u]) {
                        float* P = &pp[u].x; float* G = &gg[u].x; float* M = &mm[u].x; float* V = &vv[u].x;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float gk = G[k] * gs;
                            M[k] = M[k] + (1.f - b1) * (gk - M[k]);
                            V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
                            P[k] -= step_size * M[k] / (sqrtf(V[k]) * rsqrt_bc2 + eps);
                        }
                        p4[idx[u]] = pp[u]; m4[idx[u]] = mm[u]; v4[idx[u]] = vv[u];
                        if (zero && any[u]) g4[idx[u]] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (!was[u] && (lane & (L - 1)) == 0) seen[idx[u] >> sh] = 1;
                    }
                }
            }
            return;
        }
        for (int64_t i = (int64_t)(b - T.blk[t]) * 256 + threadIdx.x; i < n4; i += (int64_t)nb * 256) {
            float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
            float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = G[k] * gs;
                M[k] = M[k] + (1.f - b1) * (gk - M[k]);
                V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
                P[k] -= step_size * M[k] / (sqrtf(V[k]) * rsqrt_bc2 + eps);
            }
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
            if (zero) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (b == T.blk[t] && (int64_t)threadIdx.x < n - n4 * 4) {
            const int64_t i = n4 * 4 + threadIdx.x;
            const float gk = g[i] * gs;
            const float mk = m[i] + (1.f - b1) * (gk - m[i]);
            const float vk = b2 * v[i] + (1.f - b2) * gk * gk;
            m[i] = mk; v[i] = vk;
            p[i] -= step_size * mk / (sqrtf(vk) * rsqrt_bc2 + eps);
            if (zero) g[i] = 0.f;
        }
    } else {
        for (int64_t i = (int64_t)(b - T.blk[t]) * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
            const float gk = g[i] * gs;
            const float mk = m[i] + (1.f - b1) * (gk - m[i]);
            const float vk = b2 * v[i] + (1.f - b2) * gk * gk;
            m[i] = mk; v[i] = vk;
            p[i] -= step_size * mk / (sqrtf(vk) * rsqrt_bc2 + eps);
            if (zero) g[i] = 0.f;
        }
    }
}

struct OptSlots { int slot[OPT_MAXT]; int count; };
__global__ void optim_count_kernel(int64_t* counters, const OptSlots S) { if ((int)threadIdx.x < S.count) counters[S.slot[threadIdx.x]] += 1; }

static inline int opt_blocks(int64_t n)
{
    int64_t b = (n + OPT_CHUNK - 1) / OPT_CHUNK;
    return (int)(b < 1 ? 1 : (b > OPT_MAX_BLOCKS ? OPT_MAX_BLOCKS : b));
}

extern "C" int64_t sgnn_optim_partials(const int64_t* numels, int64_t n_tensors)
{
    if (n_tensors < 0 || (n_tensors && !numels)) return -1;
    int64_t total = 0;
    for (int64_t i = 0; i < n_tensors; ++i) { if (numels[i] < 0) return -1; total += opt_blocks(numels[i]); }
    return total;
}

// one launch group = tensors [from, to): fills T, returns its workgroup count
static int opt_fill(OptTensors& T, float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                    const int64_t* numels, const int32_t* zero_grad, const int64_t* slots, int64_t from, int64_t to)
{
    int blocks = 0;
    T.count = (int)(to - from);
    for (int64_t i = from; i < to; ++i) {
        const int k = (int)(i - from);
        T.p[k] = params ? params[i] : nullptr; T.g[k] = grads[i];
        T.m[k] = exp_avg ? exp_avg[i] : nullptr; T.v[k] = exp_avg_sq ? exp_avg_sq[i] : nullptr;
        T.n[k] = numels[i]; T.blk[k] = blocks; T.zero[k] = zero_grad ? zero_grad[i] : 0; T.slot[k] = (int)(slots ? slots[i] : i);
        blocks += opt_blocks(numels[i]);
    }
    T.blk[T.count] = blocks;
    T.rows_t = -1; T.row_lanes = 1; T.seen = nullptr;
    return blocks;
}

extern "C" int sgnn_optim_sumsq(const float* const* grads, const int64_t* numels, int64_t n_tensors, float* partial,
                                int64_t* step_counters, const int64_t* counter_slots, void* stream)
{
    if (n_tensors < 0 || (n_tensors && (!grads || !numels || !partial))) return SGNN_ERR_BAD_ARG;
    for (int64_t i = 0; i < n_tensors; ++i) if (!grads[i] || numels[i] < 0 || (((uintptr_t)grads[i]) & 3)) return SGNN_ERR_BAD_ARG;
    int64_t base = 0;
    for (int64_t from = 0; from < n_tensors; from += OPT_MAXT) {
        const int64_t to = from + OPT_MAXT < n_tensors ? from + OPT_MAXT : n_tensors;
        OptTensors T;
        const int blocks = opt_fill(T, nullptr, (float* const*)grads, nullptr, nullptr, numels, nullptr, counter_slots, from, to);
        hipLaunchKernelGGL(optim_sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T, partial + base, step_counters);
        SGNN_CHECK_LAUNCH();
        base += blocks;
    }
    return SGNN_OK;
}

extern "C" int sgnn_optim_count(int64_t* step_counters, const int64_t* counter_slots, int64_t n_tensors, void* stream)
{
    if (n_tensors < 0 || (n_tensors && !step_counters)) return SGNN_ERR_BAD_ARG;
    for (int64_t from = 0; from < n_tensors; from += OPT_MAXT) {
        OptSlots S;
        S.count = (int)(n_tensors - from < OPT_MAXT ? n_tensors - from : OPT_MAXT);
        for (int k = 0; k < S.count; ++k) S.slot[k] = (int)(counter_slots ? counter_slots[from + k] : from + k);
        hipLaunchKernelGGL(optim_count_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, step_counters, S);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}

extern "C" int sgnn_optim_adam(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                               const int64_t* numels, const int32_t* zero_grad, int64_t n_tensors, float lr, float beta1, float beta2,
                               float eps, const int64_t* steps, const int64_t* step_counters, const int64_t* counter_slots,
                               const int64_t* row_lens, unsigned char* const* row_seen,
                               const float* partial, int64_t n_partial, float max_norm, float* coef_out, void* stream)
{
    if (n_tensors < 0 || (n_tensors && (!params || !grads || !exp_avg || !exp_avg_sq || !numels))) return SGNN_ERR_BAD_ARG;
    if ((steps == nullptr) == (step_counters == nullptr) && n_tensors) return SGNN_ERR_BAD_ARG;      // exactly one of the two
    if (max_norm > 0.f && (!partial || n_partial < 0 || n_partial > 0x7fffffffll)) return SGNN_ERR_BAD_ARG;
    for (int64_t i = 0; i < n_tensors; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numels[i] < 0 || (steps && steps[i] < 1)) return SGNN_ERR_BAD_ARG;
        if ((((uintptr_t)params[i]) | ((uintptr_t)grads[i]) | ((uintptr_t)exp_avg[i]) | ((uintptr_t)exp_avg_sq[i])) & 3) return SGNN_ERR_BAD_ARG;
    }
    int64_t from = 0;
    bool first = true;
    while (from < n_tensors) {
        int64_t to = from + 1;                     // a launch = up to OPT_MAXT consecutive tensors with the same host step count
        while (to < n_tensors && to - from < OPT_MAXT && (!steps || steps[to] == steps[from])) ++to;
        OptTensors T;
        const int blocks = opt_fill(T, params, grads, exp_avg, exp_avg_sq, numels, zero_grad, counter_slots, from, to);
        for (int64_t i = from; i < to && row_lens && row_seen && T.rows_t < 0; ++i) {
            const int64_t D = row_lens[i];
            if (!row_seen[i] || D < 4 || D > 256 || (D & (D - 1)) != 0 || numels[i] % D != 0) continue;      // 1..64 float4 lanes per row
            if ((((uintptr_t)params[i]) | ((uintptr_t)grads[i]) | ((uintptr_t)exp_avg[i]) | ((uintptr_t)exp_avg_sq[i])) & 15) continue;
            T.rows_t = (int)(i - from); T.row_lanes = (int)(D / 4); T.seen = row_seen[i];
        }
        hipLaunchKernelGGL(optim_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T, partial, (int)n_partial, max_norm,
                           first ? coef_out : nullptr, lr, beta1, beta2, eps, (long long)(steps ? steps[from] : 0), step_counters);
        SGNN_CHECK_LAUNCH();
        first = false;
        from = to;
    }
    return SGNN_OK;
}

SGNN_DEFINE_WARM(optim)
