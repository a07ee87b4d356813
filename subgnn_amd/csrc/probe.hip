// Measurement aid: a plain streaming copy of a known byte count, 4 or 16 bytes per lane -- what the
// memory-side counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) are calibrated on before they are read
// for the CSR gather (gfx950 tallies 128-byte read requests as 64 bytes for wide loads; other widths are
// uncalibrated: MI355X_MICROARCH.md, "HBM").  Not on the hot path.
#include "common.h"

template <typename V>
__global__ __launch_bounds__(256) void probe_copy_kernel(const V* __restrict__ src, V* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

extern "C" int sgnn_probe_stream_copy(const void* src, void* dst, int64_t n_bytes, int bytes_per_lane, void* stream) {
    if (!src || !dst || n_bytes < 0 || (bytes_per_lane != 4 && bytes_per_lane != 16) || n_bytes % bytes_per_lane) return SGNN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = n_bytes / bytes_per_lane;
    if (n == 0) return SGNN_OK;
    const int grid = sgnn_grid_for(n, 256, 256 * 32);
    if (bytes_per_lane == 4) probe_copy_kernel<uint32_t><<<grid, 256, 0, st>>>((const uint32_t*)src, (uint32_t*)dst, n);
    else probe_copy_kernel<uint4><<<grid, 256, 0, st>>>((const uint4*)src, (uint4*)dst, n);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
}

SGNN_DEFINE_WARM(probe)
