// CC embedding initialisation (a12) and the masked read-out over components (a16).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// a12  out[r,:] = sum | max over members of E[id,:]   (reference SubGNN/SubGNN.py:609-622)
// HBM-bound random-row gather: 4*D bytes per member + 8 per member id + 4*D out per component.
// Thread = (component row, 16-byte column slice): the D/4 lanes of a row read one embedding row
// as consecutive float4 (256 B for D=64), different rows of a wavefront proceed independently,
// member loop unrolled for memory-level parallelism.
// ---------------------------------------------------------------------------------------------
template <int AGG, typename T>
__global__ __launch_bounds__(256) void cc_embed_fwd_kernel(
    const T* __restrict__ E, int64_t D4,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    int64_t padded_len, float* __restrict__ out, int32_t* __restrict__ out_arg)
{
    const int64_t total = n_sets * D4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / D4, dv = t % D4;
        const int64_t beg = set_ptr[r];
        const int n = (int)(set_ptr[r + 1] - beg);
        float4 acc;
        int4 arg = make_int4(0, 0, 0, 0);
        if (AGG == 0) {
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int i = 0; i < n; ++i) {
                const float4 x = sgnn_load4<T>(E, (int64_t)set_nodes[beg + i], D4, dv);
                acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
            }
        } else {
            // PAD rows (zeros) take part in the max whenever the padded row is longer than the
            // component (SubGNN.py:622); an empty row is all PAD.
            const bool with_pad = (n < padded_len) || (n == 0);
            const float init = with_pad ? 0.f : -INFINITY;
            acc = make_float4(init, init, init, init);
#pragma unroll 4
            for (int i = 0; i < n; ++i) {
                const int32_t id = set_nodes[beg + i];
                const float4 x = sgnn_load4<T>(E, (int64_t)id, D4, dv);
                if (x.x > acc.x) { acc.x = x.x; arg.x = id; }
                if (x.y > acc.y) { acc.y = x.y; arg.y = id; }
                if (x.z > acc.z) { acc.z = x.z; arg.z = id; }
                if (x.w > acc.w) { acc.w = x.w; arg.w = id; }
            }
            if (out_arg) reinterpret_cast<int4*>(out_arg)[t] = arg;
        }
        reinterpret_cast<float4*>(out)[t] = acc;
    }
}

template <int AGG>
__global__ __launch_bounds__(256) void cc_embed_bwd_kernel(
    const float* __restrict__ grad_out, int64_t D4,
    const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes, int64_t n_sets,
    const int32_t* __restrict__ arg, float* __restrict__ grad_E)
{
    const int64_t total = n_sets * D4;
    const int64_t D = D4 * 4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / D4, dv = t % D4;
        const float4 g = reinterpret_cast<const float4*>(grad_out)[t];
        if (AGG == 0) {
            const int64_t beg = set_ptr[r];
            const int n = (int)(set_ptr[r + 1] - beg);
            for (int i = 0; i < n; ++i) {
                const int32_t id = set_nodes[beg + i];
                if (id == 0) continue;                      // padding_idx row keeps a zero grad
                float* dst = grad_E + (int64_t)id * D + dv * 4;
                atomicAdd(dst + 0, g.x); atomicAdd(dst + 1, g.y); atomicAdd(dst + 2, g.z); atomicAdd(dst + 3, g.w);
            }
        } else {
            const int4 a = reinterpret_cast<const int4*>(arg)[t];
            if (a.x) atomicAdd(grad_E + (int64_t)a.x * D + dv * 4 + 0, g.x);
            if (a.y) atomicAdd(grad_E + (int64_t)a.y * D + dv * 4 + 1, g.y);
            if (a.z) atomicAdd(grad_E + (int64_t)a.z * D + dv * 4 + 2, g.z);
            if (a.w) atomicAdd(grad_E + (int64_t)a.w * D + dv * 4 + 3, g.w);
        }
    }
}

template <typename T>
static int cc_embed_fwd_launch(const T* E, int64_t D, const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                               int aggregator, int64_t padded_len, float* out, int32_t* out_arg, void* stream)
{
    if (!E || !set_ptr || !set_nodes || !out || n_sets < 0 || D <= 0 || aggregator < 0 || aggregator > 1)
        return SGNN_ERR_BAD_ARG;
    if (D % 4 != 0) return SGNN_ERR_UNSUPPORTED_D;
    if (n_sets == 0) return SGNN_OK;
    const int64_t D4 = D / 4;
    const int grid = sgnn_grid_for(n_sets * D4, 256);
    if (aggregator == 0)
        hipLaunchKernelGGL((cc_embed_fwd_kernel<0, T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, E, D4, set_ptr,
                           set_nodes, n_sets, padded_len, out, out_arg);
    else
        hipLaunchKernelGGL((cc_embed_fwd_kernel<1, T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, E, D4, set_ptr,
                           set_nodes, n_sets, padded_len, out, out_arg);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_cc_embed_fwd(const float* E, int64_t D,
                                 const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                 int aggregator, int64_t padded_len, float* out, int32_t* out_arg, void* stream)
{
    return cc_embed_fwd_launch<float>(E, D, set_ptr, set_nodes, n_sets, aggregator, padded_len, out, out_arg, stream);
}

extern "C" int sgnn_cc_embed_fwd_f16(const uint16_t* E_half, int64_t D,
                                     const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                     int aggregator, int64_t padded_len, float* out, int32_t* out_arg, void* stream)
{
    return cc_embed_fwd_launch<__half>(reinterpret_cast<const __half*>(E_half), D, set_ptr, set_nodes, n_sets, aggregator,
                                       padded_len, out, out_arg, stream);
}

extern "C" int sgnn_cc_embed_bwd(const float* grad_out, int64_t D,
                                 const int64_t* set_ptr, const int32_t* set_nodes, int64_t n_sets,
                                 int aggregator, const int32_t* arg, float* grad_E, void* stream)
{
    if (!grad_out || !set_ptr || !set_nodes || !grad_E || n_sets < 0 || D <= 0 || aggregator < 0 || aggregator > 1)
        return SGNN_ERR_BAD_ARG;
    if (aggregator == 1 && !arg) return SGNN_ERR_BAD_ARG;
    if (D % 4 != 0) return SGNN_ERR_UNSUPPORTED_D;
    if (n_sets == 0) return SGNN_OK;
    const int64_t D4 = D / 4;
    const int grid = sgnn_grid_for(n_sets * D4, 256);
    if (aggregator == 0)
        hipLaunchKernelGGL(cc_embed_bwd_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out, D4,
                           set_ptr, set_nodes, n_sets, arg, grad_E);
    else
        hipLaunchKernelGGL(cc_embed_bwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out, D4,
                           set_ptr, set_nodes, n_sets, arg, grad_E);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------
// a16  masked sum over the components of a subgraph (reference SubGNN/subgraph_utils.py:213-237)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_sum_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                             int64_t B, int64_t C, int64_t H, float* __restrict__ out)
{
    const int64_t total = B * H;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / H, h = t % H;
        float acc = 0.f;
        for (int64_t c = 0; c < C; ++c)
            if (mask[b * C + c]) acc += x[(b * C + c) * H + h];
        out[t] = acc;
    }
}

__global__ __launch_bounds__(256) void masked_sum_bwd_kernel(const float* __restrict__ g, const uint8_t* __restrict__ mask,
                                                             int64_t B, int64_t C, int64_t H, float* __restrict__ gx)
{
    const int64_t total = B * C * H;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bc = t / H, h = t % H;
        gx[t] = mask[bc] ? g[(bc / C) * H + h] : 0.f;
    }
}

// the same with four consecutive columns per thread (16-byte loads and stores, one index division per four elements):
// H a multiple of 4 and 16-byte aligned pointers -- the read-out of the benchmark (50k x 448) went from 2.5 to ~5 TB/s
__global__ __launch_bounds__(256) void masked_sum_fwd4_kernel(const float4* __restrict__ x, const uint8_t* __restrict__ mask,
                                                              int64_t B, int64_t C, int64_t H4, float4* __restrict__ out)
{
    const int64_t total = B * H4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / H4, h = t % H4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t c = 0; c < C; ++c)
            if (mask[b * C + c]) {
                const float4 v = x[(b * C + c) * H4 + h];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        out[t] = acc;
    }
}

__global__ __launch_bounds__(256) void masked_sum_bwd4_kernel(const float4* __restrict__ g, const uint8_t* __restrict__ mask,
                                                              int64_t B, int64_t C, int64_t H4, float4* __restrict__ gx)
{
    const int64_t total = B * C * H4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bc = t / H4, h = t % H4;
        gx[t] = mask[bc] ? g[(bc / C) * H4 + h] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

static inline bool ms_vec4_ok(const void* a, const void* b, int64_t H) {
    return H % 4 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

extern "C" int sgnn_masked_sum_fwd(const float* x, const uint8_t* mask, int64_t B, int64_t C, int64_t H,
                                   float* out, void* stream)
{
    if (!x || !mask || !out || B < 0 || C < 0 || H < 0) return SGNN_ERR_BAD_ARG;
    if (B * H == 0) return SGNN_OK;
    if (ms_vec4_ok(x, out, H)) {
        hipLaunchKernelGGL(masked_sum_fwd4_kernel, dim3(sgnn_grid_for(B * (H / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)x, mask, B, C, H / 4, (float4*)out);
        SGNN_CHECK_LAUNCH();
        return SGNN_OK;
    }
    hipLaunchKernelGGL(masked_sum_fwd_kernel, dim3(sgnn_grid_for(B * H, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       mask, B, C, H, out);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

extern "C" int sgnn_masked_sum_bwd(const float* grad_out, const uint8_t* mask, int64_t B, int64_t C, int64_t H,
                                   float* grad_x, void* stream)
{
    if (!grad_out || !mask || !grad_x || B < 0 || C < 0 || H < 0) return SGNN_ERR_BAD_ARG;
    if (B * C * H == 0) return SGNN_OK;
    if (ms_vec4_ok(grad_out, grad_x, H)) {
        hipLaunchKernelGGL(masked_sum_bwd4_kernel, dim3(sgnn_grid_for(B * C * (H / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)grad_out, mask, B, C, H / 4, (float4*)grad_x);
        SGNN_CHECK_LAUNCH();
        return SGNN_OK;
    }
    hipLaunchKernelGGL(masked_sum_bwd_kernel, dim3(sgnn_grid_for(B * C * H, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, mask, B, C, H, grad_x);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

// ---- a batch's rows of several per-split tensors in one launch ------------------------------------------------------------------
// _pad_collate (SubGNN/SubGNN.py:1068-1114) assembles a batch from the per-subgraph tensors of a split: component ids, border
// ids, the similarity rows of three channels, labels -- 7-12 row gathers of a few KB to 30 MB each, one library launch apiece.
// Here: dst[t][i, :] = src[t][idx[i], :] for every listed tensor (rows as raw bytes).  Tensors travel as kernel arguments;
// blockIdx.x = (tensor, batch row) -- the dimension that may be large: a whole split x 24 tensors --, blockIdx.y strides over
// the row.  An index outside its source (index_select raises for it) yields a ZERO row and sets *out_of_range, which the
// caller polls (ops.poll_index_errors): the failure stays observable although nothing here can raise.
#define GM_MAX 24
struct GatherMany {
    const unsigned char* src[GM_MAX];
    unsigned char* dst[GM_MAX];
    long long row_bytes[GM_MAX];
    long long src_rows[GM_MAX];
    int count;
};

__global__ __launch_bounds__(256) void gather_rows_many_kernel(const GatherMany G, const int64_t* __restrict__ idx, int64_t B,
                                                               int32_t* __restrict__ out_of_range)
{
    const int64_t y = blockIdx.x;
    const int t = (int)(y / B);
    const int64_t i = y - (int64_t)t * B;
    const int64_t r = idx[i];
    const int64_t nb = G.row_bytes[t];
    unsigned char* __restrict__ d = G.dst[t] + i * nb;
    const int64_t tid = (int64_t)blockIdx.y * 256 + threadIdx.x, stride = (int64_t)gridDim.y * 256;
    if (r < 0 || r >= G.src_rows[t]) {
        if (out_of_range && tid == 0) *out_of_range = 1;
        for (int64_t k = tid; k < nb; k += stride) d[k] = 0;
        return;
    }
    const unsigned char* __restrict__ s = G.src[t] + r * nb;
    if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)nb) & 15) == 0) {
        for (int64_t k = tid; k < nb / 16; k += stride) reinterpret_cast<uint4*>(d)[k] = reinterpret_cast<const uint4*>(s)[k];
    } else if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)nb) & 3) == 0) {
        for (int64_t k = tid; k < nb / 4; k += stride) reinterpret_cast<uint32_t*>(d)[k] = reinterpret_cast<const uint32_t*>(s)[k];
    } else {
        for (int64_t k = tid; k < nb; k += stride) d[k] = s[k];
    }
}

extern "C" int64_t sgnn_gather_rows_many_max(void) { return GM_MAX; }

extern "C" int sgnn_gather_rows_many(int64_t n, const void* const* src, void* const* dst, const int64_t* row_bytes,
                                     const int64_t* src_rows, const int64_t* idx, int64_t B, int32_t* out_of_range, void* stream)
{
    if (n < 0 || n > GM_MAX || B < 0 || (n && (!src || !dst || !row_bytes || !src_rows)) || (n && B && !idx)) return SGNN_ERR_BAD_ARG;
    if (n == 0 || B == 0) return SGNN_OK;
    GatherMany G;
    G.count = (int)n;
    int64_t longest = 0;
    for (int t = 0; t < n; ++t) {
        if (!src[t] || !dst[t] || row_bytes[t] < 0 || src_rows[t] < 1) return SGNN_ERR_BAD_ARG;
        G.src[t] = (const unsigned char*)src[t]; G.dst[t] = (unsigned char*)dst[t];
        G.row_bytes[t] = row_bytes[t]; G.src_rows[t] = src_rows[t];
        if (row_bytes[t] > longest) longest = row_bytes[t];
    }
    if (n * B > 0x7fffffffll / 2) return SGNN_ERR_BAD_ARG;
    int gx = (int)((longest + 16 * 256 * 4 - 1) / (16 * 256 * 4));          // ~4 vectors per lane for the longest row
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gather_rows_many_kernel, dim3((unsigned)(n * B), (unsigned)gx), dim3(256), 0, (hipStream_t)stream, G, idx, B, out_of_range);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(embed)
