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
                                                        int zero_grad)
{
    const float gs = grad_scale ? grad_scale[0] : 1.f;
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
                                      const float* __restrict__ grad_scale, int zero_grad)
{
    const int64_t i = from + threadIdx.x;
    if (i >= n) return;
    const float gk = g[i] * (grad_scale ? grad_scale[0] : 1.f);
    m[i] = m[i] + (1.f - b1) * (gk - m[i]);
    v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
    p[i] -= step_size * m[i] / (sqrtf(v[i]) * rsqrt_bc2 + eps);
    if (zero_grad) g[i] = 0.f;
}

extern "C" int sgnn_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, int64_t step, const float* grad_scale, int zero_grad, void* stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return SGNN_ERR_BAD_ARG;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return SGNN_ERR_BAD_ARG;
    if (n == 0) return SGNN_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipStream_t st = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    if (n4 > 0) {
        hipLaunchKernelGGL(adam_step_kernel, dim3(sgnn_grid_for(n4, 256)), dim3(256), 0, st, (float4*)param, (float4*)grad,
                           (float4*)exp_avg, (float4*)exp_avg_sq, n4, beta1, beta2, eps, step_size, rsqrt_bc2, grad_scale, zero_grad);
        SGNN_CHECK_LAUNCH();
    }
    if (n4 * 4 < n) {
        hipLaunchKernelGGL(adam_step_tail_kernel, dim3(1), dim3(64), 0, st, param, grad, exp_avg, exp_avg_sq, n4 * 4, n, beta1,
                           beta2, eps, step_size, rsqrt_bc2, grad_scale, zero_grad);
        SGNN_CHECK_LAUNCH();
    }
    return SGNN_OK;
}
