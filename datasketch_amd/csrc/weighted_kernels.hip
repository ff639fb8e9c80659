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
// params[dim][5][S_pad] 32-bit words -- per column: S_pad doubles 1/r (see below), then r, ln_c,
// beta -- so that the 64 samples of a wave read contiguous runs.  Column indices and data values
// are wave-uniform and come through the scalar path.
//
// The float32 quotient without a division.  q = RN32(L / r) is what numpy computes.  With
// y = RN64(1/r) (one correctly rounded double per table entry, computed once at creation) the
// kernel evaluates q' = RN32(RN64(L * y)): a double multiply and a conversion instead of the
// ~10-instruction IEEE division sequence.  q' == q for every finite L and r with
// 2^-40 <= |L|, |r| <= 2^40 (and for L = 0, +-inf):
//   * the exact quotient Q = L/r of two 24-bit significands is never closer than 2^-49 (relative)
//     to a midpoint of the float32 grid: a midpoint has an odd 25-bit significand M, so M*r has at
//     least 25 significant bits and cannot equal the 24-bit L; |L - M*r| is then at least one unit
//     of the 49-bit product, i.e. >= 2^-49 relative;
//   * L*y differs from Q by at most 2^-53 (y) + 2^-53 (product rounding) < 2^-51 relative;
//   * so RN64(L*y) lies on the same side of every float32 midpoint as Q, and rounding it to
//     float32 gives RN32(Q); the ranges keep Q within [2^-80, 2^80], far from under/overflow.
// The table is checked for the r range at creation and every row's values are checked by a
// pre-pass (NaN, out-of-range): anything outside goes through the true IEEE division instead, so
// the (k, t) pairs are bit-identical to numpy's in all cases.
//
// (Handing the blocked kernel the logs already converted to double by the pre-pass was measured:
// 8 % slower -- the extra 8 B per element through the scalar cache cost more than the conversion.)
//
// Row blocking.  The table is 5 words per (column, sample): at one row per wave the kernel is
// bound by L2 bandwidth, not arithmetic.  Blocks of 8 consecutive rows that share one column
// list (every block of a dense matrix) are hashed together: table entries are loaded once per
// column and used for 8 rows from registers.
#include <rocprim/device/device_scan.hpp>

#include "mhx_internal.h"

#pragma clang fp contract(off)

namespace mhx {
namespace {

constexpr int kWave = 64;
constexpr int kRowBlock = 8;   // rows hashed together when they share their column list
constexpr int kColChunk = 2;   // columns per software-pipeline stage (2: 72 VGPRs, 7 waves/SIMD; 4 is 10 % slower, 1 and 3 in between)
constexpr int kWords = 5;      // table words per (column, sample)
#define MHX_CONST_AS __attribute__((address_space(4)))

enum : uint8_t { kFlagSamePattern = 1, kFlagSane = 2 };

// [S, dim] x3  ->  [dim][5][S_pad]
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
        float *p = params + (int64_t)j * kWords * s_pad;
        reinterpret_cast<double *>(p)[i] = 1.0 / (double)r;  // correctly rounded (IEEE double division)
        p[2 * s_pad + i] = r;
        p[3 * s_pad + i] = c;
        p[4 * s_pad + i] = be;
    }
}

// ---- pre-pass: one wave per row -------------------------------------------------------------
// logs[j] = ln(x) (device-log mode only), flags[row] = kFlagSamePattern (same column list as the
// first row of its block of 8) | kFlagSane (every log value is 0, +-inf or 2^-40 <= |L| <= 2^40).
__device__ __forceinline__ bool sane_log(float l) {
    const float m = fabsf(l);
    return l == 0.0f || (m >= 0x1p-40f && m <= 0x1p40f) || m == __builtin_inff();
}

template <bool LOGS>
__global__ __launch_bounds__(256) void weighted_prepare_kernel(const int64_t *__restrict__ indptr,
                                                               const int32_t *__restrict__ indices,
                                                               const float *__restrict__ values, int64_t n_rows,
                                                               float *__restrict__ logs,
                                                               uint8_t *__restrict__ flags) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t leader = row / kRowBlock * kRowBlock;
    const int64_t beg = indptr[row], end = indptr[row + 1];
    const int64_t lbeg = indptr[leader], lend = indptr[leader + 1];
    bool same = (end - beg) == (lend - lbeg), sane = true;
    for (int64_t j = beg + lane; j < end; j += kWave) {
        float l = values[j];
        if (!LOGS) {
            l = logf(l);
            logs[j] = l;
        }
        sane &= sane_log(l);
        if (same && row != leader) same = indices[j] == indices[lbeg + (j - beg)];
    }
    same = __all(same);
    sane = __all(sane);
    if (lane == 0) flags[row] = (same ? kFlagSamePattern : 0) | (sane ? kFlagSane : 0);
}

// ---- per-element arithmetic -------------------------------------------------------------------
struct Entry {  // table entry of one (column, sample)
    double rcp;
    float r, ln_c, beta;
};

// col is wave-uniform: the address is a scalar base plus a 32-bit lane offset, which selects the
// "saddr + voffset" form of global_load (no 64-bit VALU address arithmetic per load).
__device__ __forceinline__ Entry load_entry(const float *__restrict__ params, int32_t col, int32_t s_pad, int i) {
    const char *base = reinterpret_cast<const char *>(params) + (int64_t)col * (kWords * 4) * s_pad;
    const uint32_t li = (uint32_t)i, sp = (uint32_t)s_pad;
    Entry e;
    e.rcp = *reinterpret_cast<const double *>(base + (size_t)(li * 8u));
    e.r = *reinterpret_cast<const float *>(base + (size_t)(sp * 8u + li * 4u));
    e.ln_c = *reinterpret_cast<const float *>(base + (size_t)(sp * 12u + li * 4u));
    e.beta = *reinterpret_cast<const float *>(base + (size_t)(sp * 16u + li * 4u));
    return e;
}

template <bool FAST>
__device__ __forceinline__ float quotient(float logx, const Entry &e) {
    if (FAST) return (float)((double)logx * e.rcp);  // == logx / e.r, see the header
    return logx / e.r;                               // IEEE-correct division
}

template <bool FAST>
__device__ __forceinline__ void evaluate(float logx, const Entry &e, float &t, float &ln_a) {
    const float q = quotient<FAST>(logx, e);
    t = floorf(q + e.beta);        // :216
    const float u = t - e.beta;    // :217  (t - beta + 1) evaluated left to right
    const float v = u + 1.0f;
    const float ln_y = v * e.r;
    ln_a = e.ln_c - ln_y;          // :218
}

// ---- exact general path: one row, any values (NaN, out-of-range) -------------------------------
struct Best {
    float ln_a;
    float t;
    int32_t k;
};

__device__ __forceinline__ void consider(Best &best, float logx, const Entry &e, int32_t col) {
    float t, ln_a;
    evaluate<false>(logx, e, t, ln_a);
    // np.argmin: the first minimum wins; a NaN beats any number and the first NaN is kept.
    const bool take = best.k < 0 || ln_a < best.ln_a || (ln_a != ln_a && best.ln_a == best.ln_a);
    if (take) {
        best.ln_a = ln_a;
        best.t = t;
        best.k = col;
    }
}

__device__ __forceinline__ void row_exact(const int32_t MHX_CONST_AS *indices, const float MHX_CONST_AS *logs,
                                          int64_t beg, int64_t end, const float *__restrict__ params, int32_t s_pad,
                                          int i, int64_t &k_out, int64_t &t_out) {
    Best best;
    best.ln_a = 0.0f;
    best.t = 0.0f;
    best.k = -1;
    for (int64_t j = beg; j < end; ++j) {
        const int32_t col = indices[j];
        consider(best, logs[j], load_entry(params, col, s_pad, i), col);
    }
    k_out = best.k;
    t_out = (int64_t)best.t;
}

// ---- fast path: R rows sharing one column list ---------------------------------------------------
// State per row: the smallest ln_a so far and the POSITION of its column in the list; t is
// recomputed for the winner at the end (same arithmetic, same bits).  Values are sane: no NaN can
// arise, so "first minimum wins" is a strict less-than.
template <int R>
__device__ __forceinline__ void rows_fast(const int32_t MHX_CONST_AS *indices, const float MHX_CONST_AS *logs,
                                          const int32_t *__restrict__ indices_vec, const float *__restrict__ logs_vec,
                                          const int64_t (&beg)[R], int32_t nnz, const float *__restrict__ params,
                                          int32_t s_pad, int i, int64_t (&k_out)[R], int64_t (&t_out)[R]) {
    float best[R];
    int32_t pos[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        best[r] = __builtin_inff();
        pos[r] = 0;
    }
    const int32_t MHX_CONST_AS *cols = indices + beg[0];
    const int32_t nfull = nnz / kColChunk * kColChunk;
    // two register buffers of kColChunk table entries, ping-pong: the loads of the next chunk are in
    // flight while the current one is evaluated for all R rows
    Entry bufa[kColChunk], bufb[kColChunk];
    const auto load_chunk = [&](Entry (&buf)[kColChunk], int32_t j) {
        const int32_t jc = j < nfull ? j : nfull - kColChunk;  // clamped prefetch
#pragma unroll
        for (int c = 0; c < kColChunk; ++c) buf[c] = load_entry(params, cols[jc + c], s_pad, i);
    };
    const auto eval_chunk = [&](const Entry (&buf)[kColChunk], int32_t j) {
#pragma unroll
        for (int c = 0; c < kColChunk; ++c) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float t, ln_a;
                evaluate<true>(logs[beg[r] + j + c], buf[c], t, ln_a);
                const bool take = ln_a < best[r];  // first minimum wins (no NaN on this path)
                pos[r] = take ? j + c : pos[r];
                best[r] = take ? ln_a : best[r];
            }
        }
    };
    if (nfull > 0) load_chunk(bufa, 0);
    int32_t j = 0;
    for (; j + 2 * kColChunk <= nfull; j += 2 * kColChunk) {
        load_chunk(bufb, j + kColChunk);
        eval_chunk(bufa, j);
        load_chunk(bufa, j + 2 * kColChunk);
        eval_chunk(bufb, j + kColChunk);
    }
    if (j < nfull) eval_chunk(bufa, j);  // odd chunk left in bufa
    for (int32_t j = nfull; j < nnz; ++j) {
        const Entry e = load_entry(params, cols[j], s_pad, i);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t, ln_a;
            evaluate<true>(logs[beg[r] + j], e, t, ln_a);
            const bool take = ln_a < best[r];
            pos[r] = take ? j : pos[r];
            best[r] = take ? ln_a : best[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {  // the winner's (k, t): per-lane gathers, once per row
        const int32_t col = indices_vec[beg[0] + pos[r]];
        float t, ln_a;
        evaluate<true>(logs_vec[beg[r] + pos[r]], load_entry(params, col, s_pad, i), t, ln_a);
        k_out[r] = col;
        t_out[r] = (int64_t)t;
    }
}

// A block of 8 consecutive rows takes the blocked path iff it is complete, every row shares the
// first row's column list, every value is in the proven range, and the list is not empty.
__device__ __forceinline__ bool block_is_shared(const uint8_t MHX_CONST_AS *flags, const int64_t MHX_CONST_AS *indptr,
                                                int64_t row0, int64_t n_rows, int table_fast) {
    if (!table_fast || row0 + kRowBlock > n_rows) return false;
    uint32_t all_flags = kFlagSamePattern | kFlagSane;
#pragma unroll
    for (int r = 0; r < kRowBlock; ++r) all_flags &= flags[row0 + r];
    const int64_t nnz0 = indptr[row0 + 1] - indptr[row0];
    return all_flags == (kFlagSamePattern | kFlagSane) && nnz0 > 0 && nnz0 < (1ll << 31);
}

// kernel A: one wave per (shared block of 8 rows, 64-sample chunk); grid.y = sample chunk
__global__ __launch_bounds__(256) void weighted_blocks_kernel(const int64_t *__restrict__ indptr_,
                                                              const int32_t *__restrict__ indices_,
                                                              const float *__restrict__ logs_,
                                                              const uint8_t *__restrict__ flags_, int64_t n_rows,
                                                              const float *__restrict__ params, int32_t sample_size,
                                                              int32_t s_pad, int32_t table_fast,
                                                              int64_t *__restrict__ out, uint8_t *__restrict__ nonempty) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const int i = blockIdx.y * kWave + lane;  // sample handled by this lane
    const int64_t MHX_CONST_AS *indptr = (const int64_t MHX_CONST_AS *)indptr_;
    const int32_t MHX_CONST_AS *indices = (const int32_t MHX_CONST_AS *)indices_;
    const float MHX_CONST_AS *logs = (const float MHX_CONST_AS *)logs_;
    const uint8_t MHX_CONST_AS *flags = (const uint8_t MHX_CONST_AS *)flags_;
    const int64_t n_blocks = n_rows / kRowBlock;
    for (int64_t blk = (int64_t)blockIdx.x * waves_per_block + wave; blk < n_blocks;
         blk += (int64_t)gridDim.x * waves_per_block) {
        const int64_t row0 = blk * kRowBlock;
        if (!block_is_shared(flags, indptr, row0, n_rows, table_fast)) continue;  // kernel B's rows
        int64_t beg[kRowBlock];
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) beg[r] = indptr[row0 + r];
        const int32_t nnz = (int32_t)(indptr[row0 + 1] - beg[0]);
        int64_t k[kRowBlock], t[kRowBlock];
        rows_fast<kRowBlock>(indices, logs, indices_, logs_, beg, nnz, params, s_pad, i, k, t);
#pragma unroll
        for (int r = 0; r < kRowBlock; ++r) {
            if (i < sample_size) {
                int64_t *o = out + ((row0 + r) * sample_size + i) * 2;
                o[0] = k[r];
                o[1] = t[r];
            }
            if (blockIdx.y == 0 && lane == 0) nonempty[row0 + r] = 1;
        }
    }
}

// kernel B: one wave per (row, 64-sample chunk) for every row outside the shared blocks (any values)
__global__ __launch_bounds__(256) void weighted_rows_kernel(const int64_t *__restrict__ indptr_,
                                                            const int32_t *__restrict__ indices_,
                                                            const float *__restrict__ logs_,
                                                            const uint8_t *__restrict__ flags_, int64_t n_rows,
                                                            const float *__restrict__ params, int32_t sample_size,
                                                            int32_t s_pad, int32_t table_fast,
                                                            int64_t *__restrict__ out, uint8_t *__restrict__ nonempty) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const int i = blockIdx.y * kWave + lane;
    const int64_t MHX_CONST_AS *indptr = (const int64_t MHX_CONST_AS *)indptr_;
    const int32_t MHX_CONST_AS *indices = (const int32_t MHX_CONST_AS *)indices_;
    const float MHX_CONST_AS *logs = (const float MHX_CONST_AS *)logs_;
    const uint8_t MHX_CONST_AS *flags = (const uint8_t MHX_CONST_AS *)flags_;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n_rows;
         row += (int64_t)gridDim.x * waves_per_block) {
        if (block_is_shared(flags, indptr, row / kRowBlock * kRowBlock, n_rows, table_fast)) continue;
        const int64_t beg = indptr[row], end = indptr[row + 1];
        const int64_t nnz = end - beg;
        int64_t k = 0, t = 0;
        // one row per wave is bound by L2 traffic for the table, not by arithmetic: the IEEE division
        // reads 3 words per element where the reciprocal path would read 5
        if (nnz > 0) row_exact(indices, logs, beg, end, params, s_pad, i, k, t);
        if (i < sample_size) {
            int64_t *o = out + (row * sample_size + i) * 2;
            o[0] = k;
            o[1] = t;
        }
        if (blockIdx.y == 0 && lane == 0) nonempty[row] = nnz > 0 ? 1 : 0;
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

// ---- dense rows: CSR built on the device ------------------------------------------------------------
// The reference turns a dense [N, dim] input into CSR with scipy on one host core (seconds for 10^5 x 4096);
// here the dense matrix is uploaded as it is and compacted by two kernels.  An entry is stored iff its value
// is not zero (what scipy's nonzero() keeps: NaN stays); when the host passes logs, ln(0) = -inf marks the
// absent entries (no stored value has that log).
__device__ __forceinline__ bool dense_present(float v, int values_are_logs) {
    return values_are_logs ? !(v == -INFINITY) : (v != 0.0f);
}

// counts[row] = stored entries of the row; one wave per row
__global__ __launch_bounds__(256) void dense_count_kernel(const float *__restrict__ x, int64_t n_rows, int32_t dim,
                                                          int values_are_logs, int64_t *__restrict__ counts) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t row = wave; row < n_rows; row += n_waves) {
        const float *src = x + row * dim;
        int count = 0;
        for (int c0 = 0; c0 < dim; c0 += kWave) {
            const int c = c0 + lane;
            const bool keep = c < dim && dense_present(src[c], values_are_logs);
            count += __popcll(__ballot(keep));
        }
        if (lane == 0) counts[row] = count;
    }
}

// indices / values of the stored entries, in column order, at indptr[row]
__global__ __launch_bounds__(256) void dense_compact_kernel(const float *__restrict__ x, int64_t n_rows, int32_t dim,
                                                            int values_are_logs, const int64_t *__restrict__ indptr,
                                                            int32_t *__restrict__ indices, float *__restrict__ values) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t row = wave; row < n_rows; row += n_waves) {
        const float *src = x + row * dim;
        int64_t at = indptr[row];
        for (int c0 = 0; c0 < dim; c0 += kWave) {
            const int c = c0 + lane;
            const float v = c < dim ? src[c] : 0.0f;
            const bool keep = c < dim && dense_present(v, values_are_logs);
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                indices[at + below] = c;
                values[at + below] = v;
            }
            at += __popcll(mask);
        }
    }
}

int launch_weighted_dense(mhx_wgen *gen, const float *d_x, int values_are_logs, int64_t n_rows, int64_t *d_out,
                          uint8_t *d_nonempty) {
    mhx_ctx *ctx = gen->ctx;
    const int32_t dim = gen->dim;
    // scratch[4]: counts i64[n+1] | indptr i64[n+1] | scan temporary | indices i32[n*dim] | values f32[n*dim]
    const size_t ptr_bytes = ((sizeof(int64_t) * (size_t)(n_rows + 1)) + 255) & ~(size_t)255;
    const size_t cell_bytes = ((sizeof(float) * (size_t)n_rows * (size_t)dim) + 255) & ~(size_t)255;
    size_t scan_tmp = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_tmp, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0,
                                           (size_t)(n_rows + 1), rocprim::plus<int64_t>(), ctx->stream);
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::exclusive_scan (size query) failed: %s", hipGetErrorString(e));
    scan_tmp = (scan_tmp + 255) & ~(size_t)255;
    if (int rc = ctx->ensure_scratch(4, 2 * ptr_bytes + scan_tmp + 2 * cell_bytes)) return rc;
    char *base = (char *)ctx->scratch[4];
    int64_t *d_counts = (int64_t *)base;
    int64_t *d_indptr = (int64_t *)(base + ptr_bytes);
    void *d_tmp = base + 2 * ptr_bytes;
    int32_t *d_indices = (int32_t *)(base + 2 * ptr_bytes + scan_tmp);
    float *d_values = (float *)(base + 2 * ptr_bytes + scan_tmp + cell_bytes);
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_rows + 3) / 4, (int64_t)ctx->num_cus * 32));
    MHX_HIP_CHECK(hipMemsetAsync(d_counts + n_rows, 0, sizeof(int64_t), ctx->stream));  // the scan's last input
    hipLaunchKernelGGL(dense_count_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_x, n_rows, dim, values_are_logs, d_counts);
    MHX_HIP_CHECK(hipGetLastError());
    e = rocprim::exclusive_scan(d_tmp, scan_tmp, (const int64_t *)d_counts, d_indptr, (int64_t)0, (size_t)(n_rows + 1),
                                rocprim::plus<int64_t>(), ctx->stream);
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::exclusive_scan failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(dense_compact_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_x, n_rows, dim, values_are_logs, d_indptr,
                       d_indices, d_values);
    MHX_HIP_CHECK(hipGetLastError());
    // nnz only sizes the device-log buffer of launch_weighted: n_rows * dim bounds it without a read-back
    return launch_weighted(gen, d_indptr, d_indices, d_values, values_are_logs, n_rows, n_rows * (int64_t)dim, d_out, d_nonempty);
}

// the log of the device-log mode on its own (tests and the bench's tolerance gate look at it)
__global__ __launch_bounds__(256) void weighted_log_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = logf(x[i]);
}

int launch_weighted_log(mhx_ctx *ctx, const float *d_x, int64_t n, float *d_out) {
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(weighted_log_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 16))),
                       dim3(256), 0, ctx->stream, d_x, n, d_out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_weighted(mhx_wgen *gen, const int64_t *d_indptr, const int32_t *d_indices, const float *d_values,
                    int values_are_logs, int64_t n_rows, int64_t nnz, int64_t *d_out, uint8_t *d_nonempty) {
    mhx_ctx *ctx = gen->ctx;
    // scratch slot 3: row flags, then (device-log mode) the logs
    const size_t flag_bytes = ((size_t)n_rows + 255) & ~(size_t)255;
    const size_t log_bytes = values_are_logs ? 0 : sizeof(float) * (size_t)nnz;
    if (int rc = ctx->ensure_scratch(3, flag_bytes + log_bytes + 256)) return rc;
    uint8_t *d_flags = (uint8_t *)ctx->scratch[3];
    float *d_logs = values_are_logs ? const_cast<float *>(d_values) : (float *)((char *)ctx->scratch[3] + flag_bytes);
    const unsigned prep_blocks = (unsigned)((n_rows + 3) / 4);
    if (values_are_logs)
        hipLaunchKernelGGL(weighted_prepare_kernel<true>, dim3(prep_blocks), dim3(256), 0, ctx->stream, d_indptr,
                           d_indices, d_values, n_rows, d_logs, d_flags);
    else
        hipLaunchKernelGGL(weighted_prepare_kernel<false>, dim3(prep_blocks), dim3(256), 0, ctx->stream, d_indptr,
                           d_indices, d_values, n_rows, d_logs, d_flags);
    MHX_HIP_CHECK(hipGetLastError());
    const int table_fast = gen->table_fast && ctx->opt_weighted_path != 1;
    const int64_t max_blocks = (int64_t)ctx->num_cus * 8;
    const unsigned chunks = (unsigned)(gen->s_pad / kWave);
    if (table_fast && n_rows >= kRowBlock) {
        const int64_t want = (n_rows / kRowBlock + 3) / 4;
        hipLaunchKernelGGL(weighted_blocks_kernel, dim3((unsigned)std::max<int64_t>(1, std::min(want, max_blocks)), chunks),
                           dim3(256), 0, ctx->stream, d_indptr, d_indices, d_logs, d_flags, n_rows, gen->d_params,
                           gen->sample_size, gen->s_pad, table_fast, d_out, d_nonempty);
        MHX_HIP_CHECK(hipGetLastError());
    }
    const int64_t want = (n_rows + 3) / 4;
    hipLaunchKernelGGL(weighted_rows_kernel, dim3((unsigned)std::max<int64_t>(1, std::min(want, max_blocks)), chunks),
                       dim3(256), 0, ctx->stream, d_indptr, d_indices, d_logs, d_flags, n_rows, gen->d_params,
                       gen->sample_size, gen->s_pad, table_fast, d_out, d_nonempty);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

}  // namespace mhx
