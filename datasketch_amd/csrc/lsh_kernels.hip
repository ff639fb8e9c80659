// lsh_kernels.hip -- device-side LSH bucketing.
//
// Reference: MinHashLSH keeps one dictionary per band, keyed by the band's key bytes
// (datasketch/lsh.py:326-347 insert, :370-400 query); two signatures are candidates iff they share a
// key in at least one band.  Here the grouping is a sort: per band, the 64-bit digests of the band keys
// (pack_kernels.hip: FNV-1a-64 of exactly the reference's key bytes) are sorted together with the row
// numbers, so every bucket becomes a run of equal digests.  The sort is rocPRIM's device radix sort
// (a library primitive: 8 passes of 8 bits over n 64-bit keys); the kernels around it are ours.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "mhx_internal.h"

namespace mhx {
namespace {

// digests[n, bands] (row-major) -> keys[bands][n], rows[bands][n] = 0..n-1
__global__ __launch_bounds__(256) void band_major_kernel(const uint64_t *__restrict__ digests, int64_t n, int32_t bands,
                                                         uint64_t *__restrict__ keys, uint32_t *__restrict__ rows) {
    const int64_t total = n * (int64_t)bands;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / bands;  // coalesced read, scattered (stride n) write
        const int band = (int)(idx - row * bands);
        keys[(int64_t)band * n + row] = digests[idx];
        rows[(int64_t)band * n + row] = (uint32_t)row;
    }
}

}  // namespace

int launch_lsh_sort_bands(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                          uint64_t *d_sorted_digests, uint32_t *d_sorted_rows) {
    if (n >= ((int64_t)1 << 32)) return fail(MHX_ERR_UNSUPPORTED, "more than 2^32-1 signatures per call");
    // scratch[3]: digests[n, bands] | keys[bands][n] | rows[bands][n] | rocPRIM temporary storage
    const size_t dig_bytes = sizeof(uint64_t) * (size_t)n * bands;
    const size_t rows_bytes = ((sizeof(uint32_t) * (size_t)n * bands) + 255) & ~(size_t)255;
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0, 64, ctx->stream);
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed: %s", hipGetErrorString(e));
    if (int rc = ctx->ensure_scratch(3, 2 * dig_bytes + rows_bytes + tmp_bytes + 512)) return rc;
    uint64_t *d_dig = (uint64_t *)ctx->scratch[3];
    uint64_t *d_keys = (uint64_t *)((char *)ctx->scratch[3] + dig_bytes);
    uint32_t *d_rows = (uint32_t *)((char *)ctx->scratch[3] + 2 * dig_bytes);
    void *d_tmp = (char *)ctx->scratch[3] + 2 * dig_bytes + rows_bytes;
    if (int rc = launch_band_digests(ctx, d_sig, n, k, bands, r, d_dig)) return rc;
    const int64_t want = (n * bands + 255) / 256;
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 16)));
    hipLaunchKernelGGL(band_major_kernel, grid, dim3(256), 0, ctx->stream, d_dig, n, bands, d_keys, d_rows);
    MHX_HIP_CHECK(hipGetLastError());
    for (int32_t j = 0; j < bands; ++j) {
        e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys + (int64_t)j * n, d_sorted_digests + (int64_t)j * n,
                                      d_rows + (int64_t)j * n, d_sorted_rows + (int64_t)j * n, (size_t)n, 0, 64,
                                      ctx->stream);
        if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::radix_sort_pairs failed: %s", hipGetErrorString(e));
    }
    return MHX_OK;
}

}  // namespace mhx
