// weighted_kernels.hip -- WeightedMinHashGenerator.minhash_many on gfx950.
//
// Reference: datasketch/weighted_minhash.py:161-247 (Ioffe's improved consistent weighted
// sampling).  For every row d, sample i and stored column j with log value L = ln(x[d,j]):
//     t    = floor(L / r[i,j] + beta[i,j])                    :216
//     ln_y = (t - beta[i,j] + 1) * r[i,j]                     :217
//     ln_a = ln_c[i,j] - ln_y                                 :218
//     j*   = first argmin_j ln_a                              :229 (np.argmin)
//     out[d,i] = (j*, t at j*)   as int64                     :233-239
// Everything is float32 with one rounding per operation, exactly like numpy: this file is
// compiled with -ffp-contract=off (no FMA fusion), division and floor are IEEE-exact.
//
// Layout: samples on lanes.  The generator tables are transposed once at creation to
// params[dim][3][S_pad] so that, for one column, the 64 samples of a wave read three
// contiguous 256-byte runs (r, ln_c, beta).  Column indices and data values are wave-uniform
// and come through the scalar path.
#include "mhx_internal.h"

#pragma clang fp contract(off)

namespace mhx {
namespace {

constexpr int kWave = 64;
#define MHX_CONST_AS __attribute__((address_space(4)))

// [S, dim] x3  ->  [dim][3][S_pad]
__global__ void wgen_transpose_kernel(const float *__restrict__ rs, const float *__restrict__ ln_cs,
                                      const float *__restrict__ betas, int32_t s, int32_t dim,
                                      int32_t s_pad, float *__restrict__ params) {
    const int64_t total = (int64_t)dim * s_pad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx / s_pad);
        const int i = (int)(idx - (int64_t)j * s_pad);
        float r = 1.0f, c = 0.0f, be = 0.0f;
        if (i < s) {
            r = rs[(int64_t)i * dim + j];
            c = ln_cs[(int64_t)i * dim + j];
            be = betas[(int64_t)i * dim + j];
        }
        float *p = params + (int64_t)j * 3 * s_pad;
        p[i] = r;
        p[s_pad + i] = c;
        p[2 * s_pad + i] = be;
    }
}

struct Best {
    float ln_a;
    float t;
    int32_t k;
};

__device__ __forceinline__ void consider(Best &best, float logx, float r, float ln_c, float beta, int32_t col) {
    const float q = logx / r;             // IEEE-correct division (hipcc default for fp32 '/')
    const float t = floorf(q + beta);     // :216
    const float u = t - beta;             // :217  (t - beta + 1) evaluated left to right
    const float v = u + 1.0f;
    const float ln_y = v * r;
    const float ln_a = ln_c - ln_y;       // :218
    // np.argmin: the first minimum wins; a NaN beats any number and the first NaN is kept.
    const bool take = best.k < 0 || ln_a < best.ln_a || (ln_a != ln_a && best.ln_a == best.ln_a);
    if (take) {
        best.ln_a = ln_a;
        best.t = t;
        best.k = col;
    }
}

// one wave per (row, 64-sample chunk); grid.y = sample chunk
template <bool LOGS>
__global__ __launch_bounds__(256) void weighted_kernel(const int64_t *__restrict__ indptr_,
                                                       const int32_t *__restrict__ indices_,
                                                       const float *__restrict__ values_,
                                                       int64_t n_rows, const float *__restrict__ params,
                                                       int32_t sample_size, int32_t s_pad,
                                                       int64_t *__restrict__ out,
                                                       uint8_t *__restrict__ nonempty) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const int i = blockIdx.y * kWave + lane;  // sample handled by this lane
    const int64_t MHX_CONST_AS *indptr = (const int64_t MHX_CONST_AS *)indptr_;
    const int32_t MHX_CONST_AS *indices = (const int32_t MHX_CONST_AS *)indices_;
    const float MHX_CONST_AS *values = (const float MHX_CONST_AS *)values_;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n_rows;
         row += (int64_t)gridDim.x * waves_per_block) {
        const int64_t beg = indptr[row], end = indptr[row + 1];
        Best best;
        best.ln_a = 0.0f;
        best.t = 0.0f;
        best.k = -1;
        for (int64_t j = beg; j < end; ++j) {
            const int32_t col = indices[j];
            float lx = values[j];
            if (!LOGS) lx = logf(lx);
            const float *p = params + (int64_t)col * 3 * s_pad + i;
            consider(best, lx, p[0], p[s_pad], p[2 * s_pad], col);
        }
        if (i < sample_size) {
            int64_t *o = out + (row * sample_size + i) * 2;
            if (end > beg) {
                o[0] = best.k;
                o[1] = (int64_t)best.t;
            } else {
                o[0] = 0;
                o[1] = 0;
            }
        }
        if (blockIdx.y == 0 && lane == 0) nonempty[row] = end > beg ? 1 : 0;
    }
}

}  // namespace

int launch_wgen_transpose(mhx_wgen *gen, const float *d_rs, const float *d_lncs, const float *d_betas) {
    mhx_ctx *ctx = gen->ctx;
    const int64_t total = (int64_t)gen->dim * gen->s_pad;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)ctx->num_cus * 8));
    hipLaunchKernelGGL(wgen_transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_rs, d_lncs,
                       d_betas, gen->sample_size, gen->dim, gen->s_pad, gen->d_params);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_weighted(mhx_wgen *gen, const int64_t *d_indptr, const int32_t *d_indices, const float *d_values,
                    int values_are_logs, int64_t n_rows, int64_t nnz, int64_t *d_out, uint8_t *d_nonempty) {
    (void)nnz;
    mhx_ctx *ctx = gen->ctx;
    const int64_t want = (n_rows + 3) / 4;
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 8)),
              (unsigned)(gen->s_pad / kWave));
    if (values_are_logs)
        hipLaunchKernelGGL(weighted_kernel<true>, grid, dim3(256), 0, ctx->stream, d_indptr, d_indices, d_values,
                           n_rows, gen->d_params, gen->sample_size, gen->s_pad, d_out, d_nonempty);
    else
        hipLaunchKernelGGL(weighted_kernel<false>, grid, dim3(256), 0, ctx->stream, d_indptr, d_indices, d_values,
                           n_rows, gen->d_params, gen->sample_size, gen->s_pad, d_out, d_nonempty);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

}  // namespace mhx
