// Loss and accuracy of a training / validation step in one pass over the logits: nn.CrossEntropyLoss() (mean over the
// batch, reference SubGNN/SubGNN.py:133, applied at SubGNN.py:1116-1124) and the exact-match accuracy the step logs
// (subgraph_utils.calc_accuracy, SubGNN/subgraph_utils.py:108-124).  The library form is log_softmax, an nll reduction
// that runs on ONE workgroup (48 us for 50k rows), argmax, compare, cast, mean, and in the backward two fills, the nll
// backward and the softmax backward: 12 launches.  Here: one kernel per direction + a one-wavefront finish.
#include "common.h"

#define SGNN_CE_IGNORE_INDEX (-100ll)

// thread per row; per-workgroup partial sums added in a fixed order (bit-reproducible)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int64_t B,
                                                     int32_t K, float* __restrict__ lse, float* __restrict__ partial_loss,
                                                     float* __restrict__ partial_hits, float* __restrict__ partial_rows)
{
    __shared__ float sh_l[256], sh_h[256], sh_n[256];
    const int64_t r = blockIdx.x * 256ll + threadIdx.x;
    float loss = 0.f, hit = 0.f, cnt = 0.f;
    if (r < B) {
        const float* x = logits + r * K;
        float m = x[0];
        int32_t am = 0;
        for (int32_t k = 1; k < K; ++k) { const float v = x[k]; if (v > m) { m = v; am = k; } }      // first maximum, as argmax
        float sum = 0.f;
        for (int32_t k = 0; k < K; ++k) sum += expf(x[k] - m);
        const float l = m + logf(sum);
        lse[r] = l;
        const int64_t y = labels[r];
        if (y >= 0 && y < K) { loss = l - x[y]; hit = (am == (int32_t)y) ? 1.f : 0.f; cnt = 1.f; }
        else if (y != SGNN_CE_IGNORE_INDEX) { loss = __builtin_nanf(""); cnt = 1.f; }
    }
    sh_l[threadIdx.x] = loss;
    sh_h[threadIdx.x] = hit;
    sh_n[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = (sh_l[threadIdx.x] + sh_l[threadIdx.x + 64]) + (sh_l[threadIdx.x + 128] + sh_l[threadIdx.x + 192]);
        float h = (sh_h[threadIdx.x] + sh_h[threadIdx.x + 64]) + (sh_h[threadIdx.x + 128] + sh_h[threadIdx.x + 192]);
        float n = (sh_n[threadIdx.x] + sh_n[threadIdx.x + 64]) + (sh_n[threadIdx.x + 128] + sh_n[threadIdx.x + 192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); h += __shfl_xor(h, o, 64); n += __shfl_xor(n, o, 64); }
        if (threadIdx.x == 0) { partial_loss[blockIdx.x] = a; partial_hits[blockIdx.x] = h; partial_rows[blockIdx.x] = n; }
    }
}

__global__ __launch_bounds__(64) void ce_finish_kernel(const float* __restrict__ partial_loss, const float* __restrict__ partial_hits,
                                                       const float* __restrict__ partial_rows, int64_t nblk, int64_t B,
                                                       float* __restrict__ loss, float* __restrict__ accuracy, float* __restrict__ rows)
{
    float a = 0.f, h = 0.f, n = 0.f;
    for (int64_t k = threadIdx.x; k < nblk; k += 64) { a += partial_loss[k]; h += partial_hits[k]; n += partial_rows[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); h += __shfl_xor(h, o, 64); n += __shfl_xor(n, o, 64); }
    if (threadIdx.x == 0) {
        loss[0] = a / n;                          // every row ignored: 0 / 0 = NaN, as the library gives
        rows[0] = n;                              // lse[B]: the divisor the backward uses (counts are exact in float up to 2^24 per partial sum tree)
        if (accuracy) accuracy[0] = h / (float)B;
    }
}

// d loss / d logits[r, k] = (softmax(r)[k] - [k == label r]) * grad_loss / (rows not ignored)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ grad_loss, int64_t B,
                                                     int32_t K, float* __restrict__ grad_logits)
{
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= B * K) return;
    const int64_t r = t / K;
    const int32_t k = (int32_t)(t - r * K);
    const int64_t y = labels[r];
    const float scale = grad_loss[0] / lse[B];
    float v = 0.f;
    if (y >= 0 && y < K) v = (expf(logits[t] - lse[r]) - (k == (int32_t)y ? 1.f : 0.f)) * scale;
    grad_logits[t] = v;
}

static inline int64_t ce_blocks(int64_t B) { return (B + 255) / 256; }

extern "C" int64_t sgnn_cross_entropy_workspace_bytes(int64_t B)
{
    if (B < 0) return -1;
    return 3 * ce_blocks(B) * (int64_t)sizeof(float) + 16;
}

extern "C" int sgnn_cross_entropy_fwd(const float* logits, const int64_t* labels, int64_t B, int64_t K, float* lse, float* loss,
                                      float* accuracy, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (!logits || !labels || !lse || !loss || B < 1 || K < 1 || K > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < sgnn_cross_entropy_workspace_bytes(B)) return SGNN_ERR_BAD_ARG;
    const int64_t nblk = ce_blocks(B);
    if (nblk > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    float* pl = (float*)workspace;
    float* ph = pl + nblk;
    float* pn = ph + nblk;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)nblk), dim3(256), 0, st, logits, labels, B, (int32_t)K, lse, pl, ph, pn);
    SGNN_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, st, pl, ph, pn, nblk, B, loss, accuracy, lse + B);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* lse, const float* grad_loss, int64_t B,
                                      int64_t K, float* grad_logits, void* stream)
{
    if (!logits || !labels || !lse || !grad_loss || !grad_logits || B < 1 || K < 1 || K > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    if ((B * K + 255) / 256 > 0x7fffffff) return SGNN_ERR_BAD_ARG;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)((B * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, lse,
                       grad_loss, B, (int32_t)K, grad_logits);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(loss)
