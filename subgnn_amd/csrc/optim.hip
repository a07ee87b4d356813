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

SGNN_DEFINE_WARM(optim)
