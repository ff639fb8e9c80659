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

#include <type_traits>

#include "mhx_internal.h"

#pragma clang fp contract(off)

namespace mhx {
namespace {

constexpr int kWave = 64;
constexpr int kRowBlock = 8;   // rows hashed together when they share their column list
constexpr int kColChunk = 2;   // columns per software-pipeline stage (2: 72 VGPRs, 7 waves/SIMD; 4 is 10 % slower, 1 and 3 in between)
constexpr int kWords = 5;      // table words per (column, sample)
constexpr float kKappa = 0x1p-19f;  // relative slack of the candidate filter's lower bound (32 float32 roundoff units)
#define MHX_CONST_AS __attribute__((address_space(4)))

enum : uint8_t { kFlagSamePattern = 1, kFlagSane = 2 };

// [S, dim] x3  ->  [dim][5][S_pad], and the two tables of the dense filter kernel (below):
//   wtab[ceil(dim/4)][S_pad][4]   lower-bound words w' (see "dense rows with a candidate filter"), +inf for padding
//   aos[dim][S_pad] = {r, ln_c, beta, 0}   one 16-byte gather per candidate
__global__ void wgen_transpose_kernel(const float *__restrict__ rs, const float *__restrict__ ln_cs,
                                      const float *__restrict__ betas, int32_t s, int32_t dim,
                                      int32_t s_pad, float *__restrict__ params, float *__restrict__ wtab,
                                      float4 *__restrict__ aos) {
    const int32_t dim_pad = (dim + 3) & ~3;
    const int64_t total = (int64_t)dim_pad * s_pad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx / s_pad);
        const int i = (int)(idx - (int64_t)j * s_pad);
        float r = 1.0f, c = 0.0f, be = 0.0f;
        const bool real = i < s && j < dim;
        if (real) {
            r = rs[(int64_t)i * dim + j];
            c = ln_cs[(int64_t)i * dim + j];
            be = betas[(int64_t)i * dim + j];
        }
        // w' <= ln_c - r - kappa (|ln_c| + r + 1), rounded DOWN to float32 (double arithmetic, then v_cvt toward -inf)
        const double w = (double)c - (double)r - (double)kKappa * (fabs((double)c) + (double)r + 1.0);
        wtab[((int64_t)(j >> 2) * s_pad + i) * 4 + (j & 3)] = real ? __double2float_rd(w) : __builtin_inff();
        if (j >= dim) continue;
        float *p = params + (int64_t)j * kWords * s_pad;
        reinterpret_cast<double *>(p)[i] = 1.0 / (double)r;  // correctly rounded (IEEE double division)
        p[2 * s_pad + i] = r;
        p[3 * s_pad + i] = c;
        p[4 * s_pad + i] = be;
        aos[(int64_t)j * s_pad + i] = make_float4(r, c, be, 0.0f);
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


// ==== dense rows with a candidate filter ============================================================
// weighted_minhash.py:216-218 evaluates every (sample, column).  Almost none of them can be the argmin, and one
// subtraction proves it.  In real arithmetic t = floor(L/r + beta) <= L/r + beta, so ln_y = (t - beta + 1) r lies in
// (L, L + r] and
//         ln_a = ln_c - ln_y  >=  (ln_c - r) - L.
// With float32 roundings (one per operation, relative error <= u = 2^-24 each; no overflow or underflow for the
// ranges checked below) the same chain gives, with A = |L/r|:
//     t - beta      <= L/r + u (2.01 A + 1)                      (quotient, sum; floor only lowers)
//     u1 = t - beta <= L/r + u (3.02 A + 3),    v = u1 + 1 <= L/r + 1 + u (4.03 A + 6.01)
//     ln_y = v r    <= L + r + u (5.04 |L| + 9.02 r)
//     ln_a          >= (ln_c - r - L) - u (6.05 |L| + 12.04 r + |ln_c|).
// The table word  w' = RD(ln_c - r - kappa (|ln_c| + r + 1)),  kappa = 32 u, is computed once in double and rounded
// down; a wave tests  RN(w' - L) < thr'  with  thr' >= thr + kappa max|L|  (max over the finite logs of its rows,
// from the pre-pass; thr = the smallest ln_a evaluated so far for that (row, sample), or +inf).  RN(w' - L) is at
// most u (|w'| + |L|) above w' - L, so an element that fails the test has
//     ln_a >= thr + (32 - 1 - 12.04) u (|L| + |ln_c| + r + 1) > thr >= the row's final minimum:
// it is not the argmin and cannot tie with it.  Elements that pass ("candidates", 0.9 % of config 4) are evaluated
// exactly as weighted_minhash.py does, so (k, t) is bit-identical whatever the filter lets through.
// Ranges (checked: table at creation, logs by the pre-pass; anything else takes the exact kernel below):
// 2^-40 <= r <= 2^40, |ln_c| <= 2^40, 0 <= beta <= 1; L = 0, +-inf or 2^-60 <= |L| <= 2^60, no NaN.
// An absent entry of a dense row is L = -inf: w' - L = +inf never passes, no compaction to CSR is needed.
//
// SIMD shape.  Samples on lanes; a wave owns R rows x 64 samples.  Per column group it loads the rows' logs
// through the scalar path (wave-uniform) and w' for its lanes, and does one v_sub + one v_cmp per (row, column).
// Candidates are rare per lane (1 %) but not per wave (28 % of the tests have one in SOME lane), so they are not
// evaluated in place: the passing lanes append (lane, row, column) to a wave-private LDS queue (ballot + mbcnt
// compaction), and when the queue fills it is drained 64 entries at a time, every lane evaluating whichever entry it
// is handed (table entry and log by one gather each, IEEE division).  Results meet in LDS by a 64-bit atomic min of
// (ordered ln_a << 32 | column): smallest ln_a, then smallest column -- np.argmin's first minimum, in any order.

typedef float vec2f __attribute__((ext_vector_type(2)));
typedef float vec4f __attribute__((ext_vector_type(4)));
typedef float vec16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bool filter_sane_log(float l) {
    const float m = fabsf(l);
    return l == 0.0f || (m >= 0x1p-60f && m <= 0x1p60f) || m == __builtin_inff();  // false for NaN
}

// Where the pre-pass puts the log of (row r, column c) of a block of R rows, in floats from the block's start:
// column pairs, then rows, then the two columns of the pair -- (L[r][c], L[r][c+1]) are one aligned 8-byte pair (the
// scalar operand of a packed subtraction) and the R rows of a column pair one run of 2R floats (one scalar load).
template <int R> __device__ __forceinline__ int64_t log_slot(int64_t c, int r) { return ((c >> 1) * R + r) * 2 + (c & 1); }

// Pre-pass over a dense matrix, one wave per block of R rows: the logs (taken here in device-log mode) in the layout
// above.  (Row-major logs fetched one row at a time did not fit the scalar cache: R rows x the waves of two CUs.)
// A row missing from the last block and the odd column behind the last one are -inf (nothing stored).  Per block:
// "bad" (a value the filter's proof does not cover) and the largest finite |log|; per row: whether it stores anything.
template <bool LOGS, int R>
__global__ __launch_bounds__(256) void weighted_dense_prepare_kernel(const float *__restrict__ x, int64_t n_rows, int32_t dim,
                                                                     float *__restrict__ lt, uint8_t *__restrict__ blockbad,
                                                                     float *__restrict__ blockmax, uint8_t *__restrict__ nonempty) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t n_blocks = (n_rows + R - 1) / R;
    const int32_t dim2 = (dim + 1) & ~1;
    for (int64_t blk = wave; blk < n_blocks; blk += n_waves) {
        const int64_t row0 = blk * R;
        bool bad = false;
        uint32_t present = 0;  // bit r: row r stores something in this lane's columns
        float maxabs = 0.0f;
        for (int c0 = 0; c0 < dim2; c0 += kWave) {
            const int c = c0 + lane;
            float l[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool in = c < dim && row0 + r < n_rows;
                const float v = in ? x[(row0 + r) * dim + c] : (LOGS ? -__builtin_inff() : 0.0f);
                l[r] = LOGS ? v : logf(v);
                if (LOGS ? !(l[r] == -__builtin_inff()) : (v != 0.0f)) present |= 1u << r;
                bad |= !filter_sane_log(l[r]);
                const float m = fabsf(l[r]);
                if (m < __builtin_inff()) maxabs = fmaxf(maxabs, m);
            }
            // the even lane of a column pair stores the pair's rows 0 .. R/2-1, the odd lane rows R/2 .. R-1: R floats each
            float mine[R / 2], theirs[R / 2];
#pragma unroll
            for (int h = 0; h < R / 2; ++h) {
                const float keep = (lane & 1) ? l[R / 2 + h] : l[h];
                const float give = (lane & 1) ? l[h] : l[R / 2 + h];
                mine[h] = keep;
                theirs[h] = __shfl_xor(give, 1);
            }
            if (c < dim2) {
                float *dst = lt + (blk * dim2 + (c & ~1)) * R + (lane & 1) * R;
#pragma unroll
                for (int h = 0; h < R / 2; h += 2) {
                    const vec4f q = (lane & 1) ? vec4f{theirs[h], mine[h], theirs[h + 1], mine[h + 1]}
                                               : vec4f{mine[h], theirs[h], mine[h + 1], theirs[h + 1]};
                    *reinterpret_cast<vec4f *>(dst + 2 * h) = q;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            maxabs = fmaxf(maxabs, __shfl_xor(maxabs, o));
            present |= (uint32_t)__shfl_xor((int)present, o);
        }
        bad = __any(bad);
        if (lane == 0) {
            blockbad[blk] = bad ? 1 : 0;
            blockmax[blk] = maxabs;
        }
        if (lane < R && row0 + lane < n_rows) nonempty[row0 + lane] = (present >> lane) & 1u;
    }
}

__device__ __forceinline__ uint32_t ordered_bits(float f) {  // unsigned order == float order (no NaN here)
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// d = (w - l) - (t, t): two tests in two packed instructions; w per lane, l a scalar pair, t one float of a per-lane
// pair (HI selects which).  One rounding per subtraction, as the proof in the header assumes.
template <int HI> __device__ __forceinline__ vec2f two_tests(vec2f w, vec2f l, vec2f t) {
    vec2f d;
    if constexpr (HI == 0)
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %3 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
            : "=&v"(d) : "v"(w), "s"(l), "v"(t));
    else
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %3 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]"
            : "=&v"(d) : "v"(w), "s"(l), "v"(t));
    return d;
}

// The filter kernel: one wave per (block of R rows, 64 samples).  A batch is 4 columns x R rows = 32 tests (R = 8); the
// signs of the 32 differences are shifted into one mask per lane (v_alignbit), so the hot loop is branch-free
// packed arithmetic: a branch per test costs a wave ~50 cycles between the compare and the jump
// (tools/ubench_filter.hip).  After the batch the set bits are appended to the wave's queue: one round per bit of the
// fullest lane (ballot + mbcnt compaction).
template <int R>
__global__ __launch_bounds__(64) void weighted_dense_filter_kernel(const float *__restrict__ lt_, int64_t n_rows, int32_t dim,
                                                                   const uint8_t *__restrict__ blockbad_,
                                                                   const float *__restrict__ blockmax_,
                                                                   const float *__restrict__ wtab,
                                                                   const float4 *__restrict__ aos, int32_t sample_size,
                                                                   int32_t s_pad, int64_t *__restrict__ out, int32_t debug) {
    static_assert(R == 8, "a batch of 4 columns x R rows fills one 32-bit mask");
    constexpr unsigned long long kEmpty = ~0ull;
    constexpr int kCap = 512;  // queue entries; drained before an append could overflow it
    __shared__ unsigned long long state[R * kWave];
    __shared__ uint32_t queue[kCap];
    const int lane = threadIdx.x;
    const uint8_t MHX_CONST_AS *blockbad = (const uint8_t MHX_CONST_AS *)blockbad_;
    const float MHX_CONST_AS *blockmax = (const float MHX_CONST_AS *)blockmax_;
    const int32_t chunks = s_pad / kWave;
    const int32_t dim2 = (dim + 1) & ~1;
    const int32_t dim4 = (dim + 3) & ~3;
    const int64_t n_blocks = (n_rows + R - 1) / R;
    // a wave keeps its 64 samples and walks over blocks: the columns that won in its previous block are evaluated first in
    // the next one (below); gridDim.x is a multiple of chunks
    const int32_t ch = (int32_t)(blockIdx.x % (uint32_t)chunks);
    uint32_t prevc[R];
    bool have_prev = false;
    for (int64_t blk = blockIdx.x / (uint32_t)chunks; blk < n_blocks; blk += gridDim.x / (uint32_t)chunks) {
        if (blockbad[blk]) continue;  // weighted_dense_exact_kernel's rows
        const int64_t row0 = blk * R;
        const float slack = kKappa * blockmax[blk];
        const int32_t my = ch * kWave + lane;  // sample of this lane
        const float MHX_CONST_AS *ls = (const float MHX_CONST_AS *)(lt_ + blk * dim2 * R);
        const float *lv_base = lt_ + blk * dim2 * R;
        vec2f thr[R / 2];  // thr[r / 2][r % 2]
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // "no bound yet" is the largest finite float, not +inf: an absent entry's +inf minus +inf would be a NaN, whose sign
            // bit (set, on this hardware) would count as a pass
            thr[r / 2][r % 2] = debug == 1 ? -__builtin_inff() : __FLT_MAX__;  // debug 1 (profiling only): nothing ever passes
            state[r * kWave + lane] = kEmpty;
        }
        uint32_t count = 0;  // queue entries; wave-uniform (kept in an SGPR by the readfirstlane at every update)
        __builtin_amdgcn_wave_barrier();

        // the thresholds of this lane's sample from the state: the smallest ln_a evaluated so far + the filter's slack
        const auto refresh = [&]() {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t hi = (uint32_t)(state[r * kWave + lane] >> 32);
                const float best = from_ordered_bits(hi);
                float up = best + slack;
                up = up + fabsf(up) * 0x1p-22f;  // at least one float above thr + slack
                thr[r / 2][r % 2] = hi == 0xFFFFFFFFu ? __FLT_MAX__ : (fabsf(best) == __builtin_inff() ? best : up);
                if (debug == 1) thr[r / 2][r % 2] = -__builtin_inff();
            }
        };
        // Four queue entries per lane and round: their gathers are in flight together (a round is a chain of two
        // dependent memory accesses otherwise).  A lane without an entry works on a copy of entry 0 and drops the result.
        // An entry is (test number << 6 | lane), test number = batch * 32 + column pair * 16 + row * 2 + column in the pair.
        const auto drain = [&]() {
            __builtin_amdgcn_wave_barrier();
            constexpr int kU = 4;
            for (uint32_t b = 0; b < count; b += kU * kWave) {
                uint32_t e[kU];
                float4 ent[kU];
                float lv[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const uint32_t idx = b + u * kWave + lane;
                    e[u] = queue[idx < count ? idx : 0];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const uint32_t il = e[u] & 63u, t = e[u] >> 6;
                    const uint32_t c = (t >> 5) * 4 + ((t >> 3) & 2) + (t & 1), r = (t >> 1) & 7;
                    ent[u] = aos[(int64_t)(c < (uint32_t)dim ? c : 0u) * s_pad + ch * kWave + il];
                    lv[u] = lv_base[log_slot<R>(c, (int)r)];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const uint32_t idx = b + u * kWave + lane;
                    const uint32_t il = e[u] & 63u, t = e[u] >> 6;
                    const uint32_t c = (t >> 5) * 4 + ((t >> 3) & 2) + (t & 1), r = (t >> 1) & 7;
                    Entry en;
                    en.rcp = 0.0;
                    en.r = ent[u].x;
                    en.ln_c = ent[u].y;
                    en.beta = ent[u].z;
                    float tt, ln_a;
                    evaluate<false>(lv[u], en, tt, ln_a);
                    const unsigned long long key = ((unsigned long long)ordered_bits(ln_a + 0.0f) << 32) | c;
                    // c >= dim: a padding column let through by a NaN (inf - inf against whatever lies behind the block's logs)
                    const bool real = idx < count && c < (uint32_t)dim;
                    atomicMin(&state[real ? r * kWave + il : (uint32_t)lane], real ? key : kEmpty);
                }
            }
            count = 0;
            __builtin_amdgcn_wave_barrier();
            refresh();
        };

        // Warm-up: the columns that won the R rows of this wave's previous block, evaluated here for every row before the
        // scan.  Which columns win is mostly a property of the table (a small ln_c - r (1 - beta)), so the thresholds start
        // within rounding of their final values and the scan queues 10 candidates per (row, sample) instead of 36
        // (config 4; 8.1 if the minimum were known in advance).  Only efficiency depends on it.
        if (have_prev) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const uint32_t c = prevc[k];
                const float4 ent = aos[(int64_t)c * s_pad + my];
                Entry en;
                en.rcp = 0.0;
                en.r = ent.x;
                en.ln_c = ent.y;
                en.beta = ent.z;
                const vec4f *src = reinterpret_cast<const vec4f *>(lv_base + (int64_t)(c >> 1) * (2 * R));  // the pair's 2R logs
#pragma unroll
                for (int q = 0; q < R / 2; ++q) {
                    const vec4f v = src[q];  // rows 2q, 2q + 1: (c & ~1, c | 1) each
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = 2 * q + h;
                        const float lv = (c & 1) ? v[2 * h + 1] : v[2 * h];
                        float tt, ln_a;
                        evaluate<false>(lv, en, tt, ln_a);
                        const unsigned long long key = ((unsigned long long)ordered_bits(ln_a + 0.0f) << 32) | c;
                        unsigned long long &st = state[r * kWave + lane];
                        if (!(lv == -__builtin_inff()) && key < st) st = key;  // an absent entry is never a candidate
                    }
                }
            }
            refresh();
        }

        // 16 tests: R rows x one column pair; bit 15 - t of the result is the sign of test t = row * 2 + column
        const auto half_batch = [&](const vec16f &l, vec2f w) {
            uint32_t ma = 0, mb = 0;  // two chains of dependent alignbits
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const vec2f lr = {l[2 * r], l[2 * r + 1]};
                const vec2f d = (r & 1) ? two_tests<1>(w, lr, thr[r / 2]) : two_tests<0>(w, lr, thr[r / 2]);
                uint32_t &m = r < R / 2 ? ma : mb;
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(d.x), 31);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(d.y), 31);
            }
            return (ma << 8) | mb;
        };
        // the set bits of the lanes' masks -> queue entries, one round per bit of the fullest lane
        const auto append = [&](uint32_t mask, uint32_t batch_no) {
            for (;;) {
                const unsigned long long m = __ballot(mask != 0);
                if (!m) break;
                const uint32_t pc = (uint32_t)__popcll(m);
                if (count + pc > (uint32_t)kCap) drain();
                if (mask != 0) {
                    const uint32_t t = (uint32_t)__builtin_clz(mask);
                    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    queue[count + slot] = ((batch_no * 32u + t) << 6) | (uint32_t)lane;
                    mask &= ~(0x80000000u >> t);
                }
                count = __builtin_amdgcn_readfirstlane(count + pc);
            }
        };
        const auto load_w = [&](int32_t c) {  // the 4 table words of this lane, columns c .. c + 3 (c a multiple of 4)
            return *reinterpret_cast<const vec4f *>(wtab + ((int64_t)(c >> 2) * s_pad + my) * 4);
        };
        const auto load_l = [&](int32_t c) {  // 2 columns x R rows (c even): one scalar load
            return *reinterpret_cast<const vec16f MHX_CONST_AS *>(ls + (int64_t)c * R);
        };

        // dim4 columns in whole batches (the table is padded with +inf words, the log tile with -inf: they never pass;
        // behind the block's last column pair the tile reads into the next block or the buffer's padding)
        // The two column pairs' tiles are reloaded as soon as their 16 tests are done (two tiles of SGPRs, not four); the
        // scheduling barriers keep the loads where they are written: a whole half batch ahead of their use.
        vec16f l01 = load_l(0), l23 = load_l(2);
        vec4f w = load_w(0), w1 = load_w(4 < dim4 ? 4 : 0);  // the table words run two batches ahead
        for (int32_t c = 0; c < dim4; c += 4) {
            const int32_t cn = c + 4 < dim4 ? c + 4 : c;  // clamped prefetches
            const vec4f w2 = load_w(c + 8 < dim4 ? c + 8 : c);
            const uint32_t m01 = half_batch(l01, vec2f{w.x, w.y});
            __builtin_amdgcn_sched_barrier(0);
            l01 = load_l(cn);
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t m23 = half_batch(l23, vec2f{w.z, w.w});
            __builtin_amdgcn_sched_barrier(0);
            l23 = load_l(cn + 2);
            __builtin_amdgcn_sched_barrier(0);
            append((m01 << 16) | m23, (uint32_t)c >> 2);
            w = w1, w1 = w2;
        }
        drain();

        // the winners: (k, t) of every (row, sample); t is recomputed from the winning column (same arithmetic)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (row0 + r < n_rows) {
                const unsigned long long key = state[r * kWave + lane];
                int64_t k = 0, tt = 0;
                if (key != kEmpty) {
                    const uint32_t c = (uint32_t)key;
                    const float4 ent = aos[(int64_t)c * s_pad + my];
                    const float lv = lv_base[log_slot<R>(c, r)];
                    Entry en;
                    en.rcp = 0.0;
                    en.r = ent.x;
                    en.ln_c = ent.y;
                    en.beta = ent.z;
                    float t, ln_a;
                    evaluate<false>(lv, en, t, ln_a);
                    k = c;
                    tt = (int64_t)t;
                }
                if (my < sample_size) {
                    int64_t *o = out + ((row0 + r) * sample_size + my) * 2;
                    o[0] = k;
                    o[1] = tt;
                }
            }
            const unsigned long long key = state[r * kWave + lane];
            prevc[r] = key != kEmpty ? (uint32_t)key : (have_prev ? prevc[r] : 0u);
        }
        have_prev = true;
        __builtin_amdgcn_wave_barrier();
    }
}

// rows of blocks the filter kernel leaves alone (a NaN or a log outside the proven range somewhere in the block):
// every stored entry evaluated with the IEEE division, numpy's NaN rule; one wave per (row, 64 samples)
template <int R>
__global__ __launch_bounds__(256) void weighted_dense_exact_kernel(const float *__restrict__ lt_, int64_t n_rows, int32_t dim,
                                                                   const uint8_t *__restrict__ blockbad_,
                                                                   const float *__restrict__ params, int32_t sample_size,
                                                                   int32_t s_pad, int64_t *__restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const int i = blockIdx.y * kWave + lane;
    const uint8_t MHX_CONST_AS *blockbad = (const uint8_t MHX_CONST_AS *)blockbad_;
    const float MHX_CONST_AS *lt = (const float MHX_CONST_AS *)lt_;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n_rows; row += (int64_t)gridDim.x * waves_per_block) {
        const int64_t blk = row / R;
        if (!blockbad[blk]) continue;  // done by the filter kernel
        const int32_t dim2 = (dim + 1) & ~1;
        const float MHX_CONST_AS *l = lt + blk * dim2 * R;
        const int rr = (int)(row - blk * R);
        Best best;
        best.ln_a = 0.0f;
        best.t = 0.0f;
        best.k = -1;
        for (int32_t c = 0; c < dim; ++c) {
            const float lv = l[log_slot<R>(c, rr)];
            if (lv == -__builtin_inff()) continue;  // not stored
            consider(best, lv, load_entry(params, c, s_pad, i), c);
        }
        if (i < sample_size) {
            int64_t *o = out + (row * sample_size + i) * 2;
            o[0] = best.k < 0 ? 0 : best.k;
            o[1] = best.k < 0 ? 0 : (int64_t)best.t;
        }
    }
}

}  // namespace

int launch_wgen_transpose(mhx_wgen *gen, const float *d_rs, const float *d_lncs, const float *d_betas) {
    mhx_ctx *ctx = gen->ctx;
    const int64_t total = (int64_t)((gen->dim + 3) & ~3) * gen->s_pad;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)ctx->num_cus * 8));
    hipLaunchKernelGGL(wgen_transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_rs, d_lncs,
                       d_betas, gen->sample_size, gen->dim, gen->s_pad, gen->d_params, gen->d_wtab,
                       reinterpret_cast<float4 *>(gen->d_aos));
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

// dense rows through the candidate filter: no CSR is built (an absent entry is a log of -inf)
static int launch_weighted_dense_filtered(mhx_wgen *gen, const float *d_x, int values_are_logs, int64_t n_rows, int64_t *d_out,
                                          uint8_t *d_nonempty) {
    constexpr int R = 8;
    mhx_ctx *ctx = gen->ctx;
    const int32_t dim = gen->dim;
    const int64_t dim2 = (dim + 1) & ~1;
    const int64_t n_blocks = (n_rows + R - 1) / R;
    // scratch slot 3: blockbad u8[n_blocks] | blockmax f32[n_blocks] | the logs in the pre-pass's layout + two column pairs of padding
    const size_t bad_bytes = ((size_t)n_blocks + 255) & ~(size_t)255;
    const size_t max_bytes = (sizeof(float) * (size_t)n_blocks + 255) & ~(size_t)255;
    const size_t lt_bytes = sizeof(float) * ((size_t)n_blocks * (size_t)dim2 * R + 4 * R);
    if (int rc = ctx->ensure_scratch(3, bad_bytes + max_bytes + lt_bytes + 256)) return rc;
    uint8_t *d_bad = (uint8_t *)ctx->scratch[3];
    float *d_max = (float *)((char *)ctx->scratch[3] + bad_bytes);
    float *d_lt = (float *)((char *)ctx->scratch[3] + bad_bytes + max_bytes);
    const unsigned prep_blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_blocks + 3) / 4, (int64_t)ctx->num_cus * 32));
    if (values_are_logs)
        hipLaunchKernelGGL((weighted_dense_prepare_kernel<true, R>), dim3(prep_blocks), dim3(256), 0, ctx->stream, d_x, n_rows, dim, d_lt,
                           d_bad, d_max, d_nonempty);
    else
        hipLaunchKernelGGL((weighted_dense_prepare_kernel<false, R>), dim3(prep_blocks), dim3(256), 0, ctx->stream, d_x, n_rows, dim, d_lt,
                           d_bad, d_max, d_nonempty);
    MHX_HIP_CHECK(hipGetLastError());
    const int64_t items = n_blocks * (gen->s_pad / kWave);
    const int64_t per_cu = ctx->opt_blocks_per_cu > 0 ? ctx->opt_blocks_per_cu : 24;
    const int64_t chunks_ = gen->s_pad / kWave;
    const unsigned blocks = (unsigned)std::max<int64_t>(chunks_, std::min<int64_t>(items, per_cu * ctx->num_cus / chunks_ * chunks_));
    hipLaunchKernelGGL((weighted_dense_filter_kernel<R>), dim3(blocks), dim3(64), 0, ctx->stream, d_lt, n_rows, dim, d_bad,
                       d_max, gen->d_wtab, reinterpret_cast<const float4 *>(gen->d_aos), gen->sample_size, gen->s_pad, d_out,
                       (int32_t)ctx->opt_weighted_debug);
    MHX_HIP_CHECK(hipGetLastError());
    const unsigned chunks = (unsigned)(gen->s_pad / kWave);
    const int64_t want = (n_rows + 3) / 4;
    hipLaunchKernelGGL((weighted_dense_exact_kernel<R>), dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 8)), chunks),
                       dim3(256), 0, ctx->stream, d_lt, n_rows, dim, d_bad, gen->d_params, gen->sample_size, gen->s_pad, d_out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_weighted_dense(mhx_wgen *gen, const float *d_x, int values_are_logs, int64_t n_rows, int64_t *d_out,
                          uint8_t *d_nonempty) {
    mhx_ctx *ctx = gen->ctx;
    const int32_t dim = gen->dim;
    if (gen->table_filter && ctx->opt_weighted_path == 0 && dim <= (1 << 22))
        return launch_weighted_dense_filtered(gen, d_x, values_are_logs, n_rows, d_out, d_nonempty);
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
    const int table_fast = gen->table_fast && ctx->opt_weighted_path != 1;  // path 2 (no filter) keeps the fast quotient
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
