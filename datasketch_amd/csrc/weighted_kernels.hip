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
#define MHX_CONST_AS __attribute__((address_space(4)))

enum : uint8_t { kFlagSamePattern = 1, kFlagSane = 2 };

// [S, dim] x3  ->  params[dim][5][S_pad] (the row-block kernels) and aos[dim][S_pad] = {r, ln_c, beta, 1/r}: one
// 16-byte load per lane for a wave-uniform column (the walk kernel's direct evaluations and its table builder)
__global__ void wgen_transpose_kernel(const float *__restrict__ rs, const float *__restrict__ ln_cs,
                                      const float *__restrict__ betas, int32_t s, int32_t dim,
                                      int32_t s_pad, float *__restrict__ params, float4 *__restrict__ aos) {
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
        aos[(int64_t)j * s_pad + i] = make_float4(r, c, be, 1.0f / r);
    }
}

// ---- numpy's float32 logarithm, bit for bit ---------------------------------------------------------------
// WeightedMinHashGenerator.minhash_many takes np.log of the float32 data (ref: datasketch/weighted_minhash.py:212) and
// everything after it is a deterministic function of that log: to reproduce the reference WITHOUT a host pass over the
// matrix the device has to reproduce numpy's log, which is neither correctly rounded (3.83 ulp) nor libm's.  On x86-64
// hosts with AVX2+FMA3 or AVX512F np.log on float32 runs one SIMD loop (numpy/_core/src/umath/
// loops_exponent_log.dispatch.c.src, simd_log_FLOAT; unchanged since numpy 1.17), restated here operation by operation:
//   x = m * 2^e, 0.5 <= m < 1;  m <= 1/sqrt(2): m += m, e -= 1;  m -= 1;  P(m) / Q(m) (degree 5, Horner, FMA; IEEE
//   division);  fma(e, ln 2, P/Q);  x < 0 -> -NaN, +-0 -> -inf, +inf -> +inf, NaN -> the quiet NaN.
// Equal to np.log for ALL 2^32 float32 bit patterns (oracle/np_logf.c, the CPU model of the same algorithm, against the
// installed numpy: tests/test_np_logf_model.py, oracle/check_np_logf.py; this function against numpy on the GPU box's host:
// test_device_log_equals_numpy_log_for_every_float32).  Whether THIS host's numpy runs that loop is checked once per process
// on sentinel values before parity mode relies on it (weighted_minhash.py: device_log_matches_numpy).
__device__ __forceinline__ float np_logf(float x) {
    const uint32_t b = __float_as_uint(x);
    const bool denormal = (b >> 23) == 0;                       // (b == 0 as well: overridden below)
    const int s = denormal ? (int)__clz((int)b) - 8 : 0;        // normalise the mantissa: what getexp / getmant amount to
    const uint32_t mb = b << (s & 31);
    const int e = denormal ? -125 - s : (int)(b >> 23) - 126;
    float m = __uint_as_float((mb & 0x007fffffu) | 0x3f000000u);
    float ef = (float)e;
    const bool low = m <= 0.707106781186547524400844362104849039f;
    m = low ? m + m : m;
    ef = low ? ef - 1.0f : ef;
    m = m - 1.0f;
    float num = __builtin_fmaf(2.589979117907922693523e-02f, m, 3.808837741388407920751e-01f);
    num = __builtin_fmaf(num, m, 1.480000633576506585156e+00f);
    num = __builtin_fmaf(num, m, 2.112677543073053063722e+00f);
    num = __builtin_fmaf(num, m, 9.999999999999998702752e-01f);
    num = __builtin_fmaf(num, m, 0.0f);
    float den = __builtin_fmaf(5.875095403124574342950e-03f, m, 1.546476374983906719538e-01f);
    den = __builtin_fmaf(den, m, 9.864942958519418960339e-01f);
    den = __builtin_fmaf(den, m, 2.453006071784736363091e+00f);
    den = __builtin_fmaf(den, m, 2.612677543073109236779e+00f);
    den = __builtin_fmaf(den, m, 1.0f);
    float r = __builtin_fmaf(ef, 0.693147180559945309417232121458176568f, num / den);
    r = (b >> 31) ? __uint_as_float(0xffc00000u) : r;           // x < 0, -inf
    r = (b << 1) == 0u ? -__builtin_inff() : r;                 // +-0
    r = b == 0x7f800000u ? __builtin_inff() : r;
    r = x != x ? __uint_as_float(0x7fc00000u) : r;
    return r;
}

// The same function for positive normal x only (0x00800000 <= bits < 0x7f800000: no denormal normalisation, no special
// values), arranged for the VALU: the two Horner chains run as ONE packed fused multiply-add per step (v_pk_fma_f32), the
// split at 1/sqrt(2) is a multiplier (m * 2 - 1 and m * 1 - 1 are exact, like m + m - 1 and m - 1), and the quotient is
// the reciprocal refined once and the product corrected twice by its exact remainder -- the hardware's own division
// sequence without the operand scaling it needs for extreme exponents (here 0.44 < Q < 3.4 and |P| < 0.61).  That this
// sequence rounds P/Q like the IEEE division for EVERY (P, Q) that can occur is not argued but checked: the 2^32-pattern
// test covers every mantissa there is.
typedef float vec2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float np_logf_normal(float x) {
    const uint32_t b = __float_as_uint(x);
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
    const bool low = m <= 0.707106781186547524400844362104849039f;
    m = __builtin_fmaf(m, low ? 2.0f : 1.0f, -1.0f);
    const float ef = (float)(b >> 23) - (low ? 127.0f : 126.0f);
    const vec2f_t mm = {m, m};
    vec2f_t pq = __builtin_elementwise_fma(vec2f_t{2.589979117907922693523e-02f, 5.875095403124574342950e-03f}, mm,
                                           vec2f_t{3.808837741388407920751e-01f, 1.546476374983906719538e-01f});
    pq = __builtin_elementwise_fma(pq, mm, vec2f_t{1.480000633576506585156e+00f, 9.864942958519418960339e-01f});
    pq = __builtin_elementwise_fma(pq, mm, vec2f_t{2.112677543073053063722e+00f, 2.453006071784736363091e+00f});
    pq = __builtin_elementwise_fma(pq, mm, vec2f_t{9.999999999999998702752e-01f, 2.612677543073109236779e+00f});
    pq = __builtin_elementwise_fma(pq, mm, vec2f_t{0.0f, 1.0f});
    const float num = pq.x, den = pq.y;
    float r = __builtin_amdgcn_rcpf(den);
    r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
    float q = num * r;
    q = __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
    q = __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
    return __builtin_fmaf(ef, 0.693147180559945309417232121458176568f, q);
}
__device__ __forceinline__ bool np_logf_is_normal(float x) { return __float_as_uint(x) - 0x00800000u < 0x7f000000u; }
// an entry of a row staged in LDS: its log (VALS = false: the stripe holds logs, -inf where nothing is stored) or the log of the
// value held there (VALS = true, round 4: the one-wave-per-row kernel stages VALUES when the device takes the log, and takes it
// of the entries a walk actually meets -- a dozen per sample -- instead of all 4096 of a row; log(+-0) = -inf marks the absent
// ones as before).  Called in wave-uniform control flow (the vote decides between the two forms of the same function).
template <bool VALS>
__device__ __forceinline__ float row_log(const float *row, uint32_t c, bool staged = false) {  // staged (wave-uniform): this row's logs have been taken in place after all
    const float v = row[c];
    if (!VALS || staged) return v;
    const bool zero = v == 0.0f;  // an absent entry: log = -inf (a walk meets many of them on a row that stores a third of its columns)
    if (__builtin_expect(__all(zero || np_logf_is_normal(v)), 1)) return zero ? -__builtin_inff() : np_logf_normal(v);
    return np_logf(v);
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
            l = np_logf(l);
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

// The same t without the division, for the loops that evaluate entry after entry (sparse rows): the IEEE division is
// 11 of their ~27 VALU instructions.  e = {r, ln_c, beta, y} with y = RN(1/r) from the table.  q' = RN(L*y) differs from
// numpy's q = RN(L/r) by less than 2^-22 relative (three roundings of 2^-24) plus, should q' be subnormal, 2^-149
// absolute; so q lies strictly between b1 = RN(q'*(1 - 2^-21) - 2^-100) and b2 = RN(q'*(1 + 2^-20) + 2^-100), whichever
// way round they are.  x -> floor(RN(x + beta)) is monotone: if it gives the same t at b1 and b2 it gives that t at q,
// and ln_y, ln_a follow from t exactly as in evaluate().  Otherwise (q + beta within ~2^-20 relative of an integer: a few
// in 10^6; NaN) the result is "open" and the caller repeats the element with the true division.
// (With the hardware's reciprocal for y -- 1 ulp instead of half an ulp -- the distance grows to 2^-22: still inside.)
// Needs: |L| <= 2^80 or infinite (the table has 2^-40 <= r <= 2^40, so q' overflows only where q does); with `any` set
// the test is written so that infinite b1, b2 (inf - inf: NaN) count as open and L needs no such bound.
typedef float vec2f __attribute__((ext_vector_type(2)));

template <bool ANY>
__device__ __forceinline__ bool evaluate_guarded(float logx, const float4 e, float &t, float &ln_a) {
    const float q = logx * e.w;
    const vec2f b = __builtin_elementwise_fma(vec2f{q, q}, vec2f{0x1.ffffep-1f, 0x1.00001p+0f}, vec2f{-0x1p-100f, 0x1p-100f});
    const vec2f s = b + vec2f{e.z, e.z};
    const float t1 = floorf(s.x), t2 = floorf(s.y);
    t = t1;
    const float u = t1 - e.z;
    const float v = u + 1.0f;
    const float ln_y = v * e.x;
    ln_a = e.y - ln_y;
    return ANY ? !(t2 - t1 == 0.0f) : t1 != t2;
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


// ==== dense rows: a walk over the columns in the order of a lower bound ===========================
// weighted_minhash.py:216-229 evaluates every (sample, column) of a row and takes the argmin.  Here a row's argmin
// costs a handful of evaluations per sample, and the (k, t) pairs are the reference's bit for bit.
//
// The bound.  For one table entry (r > 0, ln_c, beta) the computed ln_a is a non-increasing function of the log L:
// every step of :216-218 -- L / r, + beta, floor, - beta, + 1, * r, ln_c - . -- is a correctly rounded (hence
// monotone) float32 operation or an exact monotone one, and r > 0.  So for every stored entry with L <= Lcut
//         ln_a(L)  >=  LB = ln_a(Lcut)          (the same float32 arithmetic, evaluated once per table entry)
// -- an exact inequality between float32 numbers, no slack.  Per sample the columns are sorted by LB (walk_build_kernel,
// once per Lcut, kept on the generator).  A lane walks its sample's list from the smallest bound, evaluates the row's
// entry at each column it meets (exactly: IEEE division, one rounding per operation) and keeps the smallest ln_a, ties
// to the smaller column (np.argmin's first minimum).  It stops at the first column whose LB is larger than what it
// holds: every column from there on has ln_a >= LB > best and can neither win nor tie.  Which columns can win is mostly a
// property of the table (a small ln_c - r (1 - beta) ...), so the walk is short: 1.9 columns per (row, sample) on
// config 4 (lock step of 64 lanes: 7 rounds) against 4096 evaluations.
// Entries above Lcut ("outliers"; Lcut is a high quantile of a sample of the call's logs, walk_plan_kernel) are
// evaluated first, directly.  A row with few stored entries (or too many outliers, or a NaN) is evaluated entry by
// entry -- for a NaN row with numpy's rule (the first NaN wins).  Only speed depends on Lcut and on the data.
//
// SIMD shape.  One workgroup per row, one wave per 64 samples.  The row's logs are staged in LDS (taken here in
// device-log mode; -inf = not stored: no CSR is built) while they are scanned for NaNs, outliers and the number of
// stored entries: the matrix is read from HBM once, which is what bounds the kernel.  The walk tables are
// [chunk][position][lane]: lanes are at the same position, so a round's table load is one coalesced 1-KB read; the
// row's entries are per-lane LDS reads.
struct WalkPlan {  // device-resident, owned by the generator
    float lcut;       // the tables in d_walk_a / d_walk_c are sorted for this cut (NaN: never built)
    int32_t rebuild;  // walk_plan_kernel's verdict for walk_build_kernel
    float sample_max, sample_quantile;  // what the last plan saw (diagnostics)
};

constexpr int kHistBits = 14;  // sign, exponent and 5 mantissa bits of the order-preserving integer image of a float
constexpr int kWalkCached = 8;  // list positions per 64-sample chunk a workgroup keeps in LDS (1.25 KB each)
constexpr int kCuts = 5;
constexpr float kCutTail[kCuts] = {0.005f, 0.01f, 0.02f, 0.04f, 0.08f};  // Lcut = a (1 - tail) quantile of the sampled logs ...
constexpr float kCutSlack = 0.7f;    // ... or their maximum when that is less than this above the first quantile
constexpr float kCutKeep = 0.35f;    // tables built for a cut in [wanted, wanted + kCutKeep] are kept
// Which tail: the one with the smallest estimated cost per row and wave.  An entry above the cut costs one evaluation
// from the list (kCostListed cycles); the walk visits positions until one of them holds a log close under the cut, about
// kWalkPerNear / (entries of a row within kNearCut under the cut) of them (measured on uniform, lognormal and sorted
// inputs: 900 .. 1300), the first kWalkCached from LDS for next to nothing, the others at kCostPosition cycles.  Both
// costs are latencies of this kernel at its occupancy, fitted to config 4's shape with lognormal (sigma = 2) weights and
// the tail forced (option weighted.tail; 20k rows, ms): 0.5 % 0.78, 1 % 0.65, 2 % 0.63, 4 % 0.81, 8 % 1.30.
// Uniform weights have a quarter of their logs that close under any high quantile: the walk is short whatever the cut
// and the smallest tail wins; the lognormal weights have 10 entries of 4096 within 0.3 under the 99.5 % quantile (the
// slowest lane of a wave walks ~90 positions) and 35 under the 98 % one (~30 positions, 80 listed entries).
constexpr float kNearCut = 0.3f, kWalkPerNear = 1000.0f, kCostListed = 345.0f, kCostPosition = 435.0f;

__device__ __forceinline__ uint32_t ordered_bits(float f) {  // unsigned order == float order (no NaN here)
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__device__ __forceinline__ Entry entry_of(const float4 e) {
    Entry en;
    en.rcp = 0.0;
    en.r = e.x;
    en.ln_c = e.y;
    en.beta = e.z;
    return en;
}

// One workgroup: histogram of n_seg runs of seg_len values spread over v[0 .. total), then the cut.
template <bool LOGS>
__global__ __launch_bounds__(1024) void walk_plan_kernel(const float *__restrict__ v, int64_t total, int32_t seg_len, int32_t n_seg,
                                                         float per_row, int32_t forced, WalkPlan *__restrict__ plan) {
    __shared__ uint32_t hist[1 << kHistBits];
    __shared__ uint32_t part[1024];
    __shared__ uint32_t s_top, s_cut[kCuts], s_above[kCuts];
    const int tid = threadIdx.x;
    for (int b = tid; b < (1 << kHistBits); b += 1024) hist[b] = 0;
    if (tid == 0) s_top = 0;
    if (tid < kCuts) s_cut[tid] = 0, s_above[tid] = 0;
    __syncthreads();
    const int64_t span = total - seg_len;
    for (int sgm = 0; sgm < n_seg; ++sgm) {
        const int64_t start = n_seg > 1 ? span * sgm / (n_seg - 1) : 0;
        for (int j = tid; j < seg_len; j += 1024) {
            float l = v[start + j];
            if (!LOGS) l = np_logf(l);
            if (fabsf(l) < __builtin_inff()) atomicAdd(&hist[ordered_bits(l) >> (32 - kHistBits)], 1u);  // finite, not NaN
        }
    }
    __syncthreads();
    // thread t owns bins [16 t, 16 t + 16); suffix sums over the threads, then inside the thread's bins from the top
    constexpr int kPer = (1 << kHistBits) / 1024;
    uint32_t mine = 0;
    for (int b = 0; b < kPer; ++b) mine += hist[tid * kPer + b];
    part[tid] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive suffix scan (Hillis-Steele)
        const uint32_t add = tid + o < 1024 ? part[tid + o] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    const uint32_t all = part[0];
    uint32_t above = tid + 1 < 1024 ? part[tid + 1] : 0;  // values in bins above this thread's
    for (int b = kPer - 1; b >= 0; --b) {
        const uint32_t h = hist[tid * kPer + b];
        if (h) atomicMax(&s_top, (uint32_t)(tid * kPer + b));
#pragma unroll
        for (int i = 0; i < kCuts; ++i) {
            const uint32_t tail = (uint32_t)((float)all * kCutTail[i]);
            if (above <= tail && above + h > tail) s_cut[i] = (uint32_t)(tid * kPer + b), s_above[i] = above;  // the bin holding the quantile: one thread
        }
        above += h;
    }
    __syncthreads();
    if (tid == 0) {
        // the largest float of a bin: every sampled value of the bin is <= it
        const auto upper = [](uint32_t bin) { return from_ordered_bits(((bin + 1u) << (32 - kHistBits)) - 1u); };
        const auto above_bin = [&](uint32_t bin) {  // sampled values in bins above `bin`
            const uint32_t t = bin / kPer;
            uint32_t n = t + 1 < 1024 ? part[t + 1] : 0;
            for (uint32_t b = bin + 1; b < (t + 1) * kPer; ++b) n += hist[b];
            return n;
        };
        float top = all ? upper(s_top) : 0.0f;
        if (!(top < __builtin_inff())) top = __FLT_MAX__;
        float want = top, q = top, best_cost = __builtin_inff();
        for (int i = 0; i < kCuts && all; ++i) {
            float cut = upper(s_cut[i]);
            if (!(cut < __builtin_inff())) cut = __FLT_MAX__;
            uint32_t listed = s_above[i];
            if (i == 0) {
                q = cut;
                if (top - cut < kCutSlack) cut = top, listed = 0;
            }
            const float near_lo = cut - kNearCut;
            const uint32_t lo_bin = near_lo > -__FLT_MAX__ ? ordered_bits(near_lo) >> (32 - kHistBits) : 0u;
            const uint32_t near = above_bin(lo_bin) - listed;  // sampled logs in (cut - kNearCut, cut], by whole bins
            const float scale = per_row / (float)all;
            const float positions = kWalkPerNear / fmaxf((float)near * scale, 1e-3f);
            const float cost = (float)listed * scale * kCostListed + fmaxf(positions - (float)kWalkCached, 0.0f) * kCostPosition;
            if (forced > 0 ? i == forced - 1 : cost < best_cost) best_cost = cost, want = cut;  // forced: option weighted.tail (profiling)
        }
        const float have = plan->lcut;
        const bool keep = have >= want && have - want <= kCutKeep;  // false for the initial NaN
        plan->rebuild = keep ? 0 : 1;
        if (!keep) plan->lcut = want;
        plan->sample_max = top;
        plan->sample_quantile = q;
    }
}

// One workgroup per sample: LB of every column at the plan's cut, a bitonic sort of (LB, column) in LDS, the walk
// tables.  P = dim rounded up to a power of two (8 P bytes of LDS).
__global__ __launch_bounds__(256) void walk_build_kernel(const WalkPlan *__restrict__ plan, const float4 *__restrict__ aos, int32_t dim,
                                                         int32_t p2, int32_t s_pad, float4 *__restrict__ walk_a,
                                                         uint32_t *__restrict__ walk_c) {
    extern __shared__ unsigned long long keys[];
    if (!plan->rebuild) return;
    const float lcut = plan->lcut;
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < p2; c += 256) {
        unsigned long long key = ~0ull;
        if (c < dim) {
            float t, ln_a;
            evaluate<false>(lcut, entry_of(aos[(int64_t)c * s_pad + i]), t, ln_a);
            key = ((unsigned long long)ordered_bits(ln_a + 0.0f) << 32) | (uint32_t)c;
        }
        keys[c] = key;
    }
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < p2 / 2; t += 256) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const unsigned long long a = keys[lo], b = keys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    const int64_t base = (int64_t)(i / kWave) * dim * kWave + (i % kWave);
    for (int k = tid; k < dim; k += 256) {
        const uint32_t c = (uint32_t)keys[k];
        const float4 e = aos[(int64_t)c * s_pad + i];
        walk_a[base + (int64_t)k * kWave] = make_float4(from_ordered_bits((uint32_t)(keys[k] >> 32)), e.x, e.y, e.z);
        walk_c[base + (int64_t)k * kWave] = c;
    }
}

// Plan and tables in ONE launch (round 4; the two kernels above remain for reference runs, option weighted.plan = 1).
// One workgroup per sample.  Every workgroup computes the plan itself -- the same ~16k sampled logs, the same integer
// histogram, hence the same cut, no communication -- and then, when the tables standing on the generator do not fit that
// cut, sorts its own sample's list.  The plan is double-buffered (`cur` read, `next` written by workgroup 0): a workgroup
// that is late must not read the verdict of one that has already finished.  Against the two launches: the sampled logs are
// all requested before the first one is counted (the old kernel paid a memory round trip per segment: 18 us), the suffix
// sums go through wave shuffles (2 barriers instead of 20), and a call whose tables stand costs one short launch, not two.
template <bool LOGS>
__global__ __launch_bounds__(1024) void walk_plan_build_kernel(const float *__restrict__ v, int64_t total, int32_t seg_len, int32_t n_seg, float per_row,
                                                               int32_t forced, const WalkPlan *__restrict__ cur, WalkPlan *__restrict__ next,
                                                               const float4 *__restrict__ aos, int32_t dim, int32_t p2, int32_t s_pad,
                                                               float4 *__restrict__ walk_a, uint32_t *__restrict__ walk_c, const unsigned int *__restrict__ gate) {
    extern __shared__ unsigned long long keys[];  // p2 sort keys (the build), behind them the histogram
    if (gate && *gate == 0) {  // (uniform) a CSR call none of whose rows is walked: the standing plan and tables stay as they are
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const WalkPlan was = *cur;
            *next = was;
            next->rebuild = 0;
        }
        return;
    }
    uint32_t *hist = reinterpret_cast<uint32_t *>(keys + p2);
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t part_above[1024];  // per thread: the sampled values in bins above the thread's own
    __shared__ uint32_t s_top, s_cut[kCuts], s_above[kCuts];
    __shared__ float s_lcut;
    __shared__ int s_rebuild;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    for (int b = tid; b < (1 << kHistBits); b += 1024) hist[b] = 0;
    if (tid == 0) s_top = 0;
    if (tid < kCuts) s_cut[tid] = 0, s_above[tid] = 0;
    __syncthreads();
    {   // the sample: n_seg runs of seg_len values spread over v[0 .. total); 16 values per thread in flight at a time
        const int64_t span = total - seg_len;
        const int64_t n_sampled = (int64_t)seg_len * n_seg;
        for (int64_t base = 0; base < n_sampled; base += 16 * 1024) {
            float val[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int64_t q = base + u * 1024 + tid;
                const int64_t sgm = q / seg_len, j = q - sgm * seg_len;
                const int64_t start = n_seg > 1 ? span * sgm / (n_seg - 1) : 0;
                val[u] = q < n_sampled ? v[start + j] : __builtin_nanf("");
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float l = val[u];
                if (!LOGS) l = np_logf(l);
                if (fabsf(l) < __builtin_inff()) atomicAdd(&hist[ordered_bits(l) >> (32 - kHistBits)], 1u);  // finite, not NaN
            }
        }
    }
    __syncthreads();
    // thread t owns bins [16 t, 16 t + 16); suffix sums over the threads by wave shuffles, then inside the thread's bins from the top
    constexpr int kPer = (1 << kHistBits) / 1024;
    uint32_t mine = 0;
    for (int b = 0; b < kPer; ++b) mine += hist[tid * kPer + b];
    uint32_t suf = mine;  // inclusive suffix sum inside the wave
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_down((int)suf, o);
        if (lane + o < kWave) suf += up;
    }
    if (lane == 0) wave_sum[wave] = suf;
    __syncthreads();
    uint32_t later = 0, all = 0;
    for (int w = 0; w < 16; ++w) {
        all += wave_sum[w];
        if (w > wave) later += wave_sum[w];
    }
    uint32_t above = later + suf - mine;  // values in bins above this thread's
    part_above[tid] = above;
    for (int b = kPer - 1; b >= 0; --b) {
        const uint32_t h = hist[tid * kPer + b];
        if (h) atomicMax(&s_top, (uint32_t)(tid * kPer + b));
#pragma unroll
        for (int i = 0; i < kCuts; ++i) {
            const uint32_t tail = (uint32_t)((float)all * kCutTail[i]);
            if (above <= tail && above + h > tail) s_cut[i] = (uint32_t)(tid * kPer + b), s_above[i] = above;  // the bin holding the quantile: one thread
        }
        above += h;
    }
    __syncthreads();
    if (wave == 0) {  // the five candidate cuts, one lane each: what the walk would cost with it (see kCutTail above)
        const auto upper = [](uint32_t bin) { return from_ordered_bits(((bin + 1u) << (32 - kHistBits)) - 1u); };
        float top = all ? upper(s_top) : 0.0f;
        if (!(top < __builtin_inff())) top = __FLT_MAX__;
        float cut = top, cost = __builtin_inff(), q = top;
        if (lane < kCuts && all) {
            cut = upper(s_cut[lane]);
            if (!(cut < __builtin_inff())) cut = __FLT_MAX__;
            uint32_t listed = s_above[lane];
            if (lane == 0) {
                q = cut;
                if (top - cut < kCutSlack) cut = top, listed = 0;
            }
            const float near_lo = cut - kNearCut;
            const uint32_t lo_bin = near_lo > -__FLT_MAX__ ? ordered_bits(near_lo) >> (32 - kHistBits) : 0u;
            uint32_t near = part_above[lo_bin / kPer];  // sampled values in bins above lo_bin ...
            for (uint32_t b = lo_bin + 1; b < (lo_bin / kPer + 1) * kPer; ++b) near += hist[b];
            near -= listed;                             // ... and not above the cut: the logs in (cut - kNearCut, cut], by whole bins
            const float scale = per_row / (float)all;
            const float positions = kWalkPerNear / fmaxf((float)near * scale, 1e-3f);
            cost = (float)listed * scale * kCostListed + fmaxf(positions - (float)kWalkCached, 0.0f) * kCostPosition;
            if (forced > 0) cost = lane == forced - 1 ? 0.0f : __builtin_inff();  // option weighted.tail (profiling)
        }
        // the cheapest (the first among equals), across the lanes
        float best_cost = cost, best_cut = cut;
        int best_lane = lane;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float oc = __shfl_xor(best_cost, o), ocut = __shfl_xor(best_cut, o);
            const int ol = __shfl_xor(best_lane, o);
            if (oc < best_cost || (oc == best_cost && ol < best_lane)) best_cost = oc, best_cut = ocut, best_lane = ol;
        }
        if (lane == 0) {
            const float want = all ? best_cut : top;
            const float have = cur->lcut;
            const bool keep = have >= want && have - want <= kCutKeep;  // false for the initial NaN
            s_lcut = keep ? have : want;
            s_rebuild = keep ? 0 : 1;
            if (blockIdx.x == 0) {
                next->lcut = keep ? have : want;
                next->rebuild = keep ? 0 : 1;
                next->sample_max = top;
                next->sample_quantile = __shfl(q, 0);
            }
        }
    }
    __syncthreads();
    if (!s_rebuild) return;
    // this workgroup's sample: LB of every column at the cut, a bitonic sort of (LB, column) in LDS, the walk tables
    const float lcut = s_lcut;
    const int i = blockIdx.x;
    for (int c = tid; c < p2; c += 1024) {
        unsigned long long key = ~0ull;
        if (c < dim) {
            float t, ln_a;
            evaluate<false>(lcut, entry_of(aos[(int64_t)c * s_pad + i]), t, ln_a);
            key = ((unsigned long long)ordered_bits(ln_a + 0.0f) << 32) | (uint32_t)c;
        }
        keys[c] = key;
    }
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < p2 / 2; t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const unsigned long long a = keys[lo], b = keys[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    const int64_t base = (int64_t)(i / kWave) * dim * kWave + (i % kWave);
    for (int k = tid; k < dim; k += 1024) {
        const uint32_t c = (uint32_t)keys[k];
        const float4 e = aos[(int64_t)c * s_pad + i];
        walk_a[base + (int64_t)k * kWave] = make_float4(from_ordered_bits((uint32_t)(keys[k] >> 32)), e.x, e.y, e.z);
        walk_c[base + (int64_t)k * kWave] = c;
    }
}

// plan (and tables, where the standing ones do not fit) for a call's logs; returns the plan the row kernels read
// (gate: device word, may be NULL -- a CSR call whose entry-by-entry launch kept every row leaves it 0 and the launch does nothing)
static int launch_walk_plan(mhx_wgen *gen, const float *d_v, bool logs, int64_t total, int32_t seg_len, int32_t n_seg, float per_row, WalkPlan **plan_out,
                            const unsigned int *gate = nullptr) {
    mhx_ctx *ctx = gen->ctx;
    const int32_t dim = gen->dim;
    WalkPlan *plans = reinterpret_cast<WalkPlan *>(gen->d_walk_plan);
    float4 *walk_a = reinterpret_cast<float4 *>(gen->d_walk_a);
    int32_t p2 = 1;
    while (p2 < dim) p2 <<= 1;
    const size_t lds = sizeof(unsigned long long) * (size_t)p2 + sizeof(uint32_t) * ((size_t)1 << kHistBits);
    if (ctx->opt_weighted_plan == 1 || lds + 8192 > (size_t)ctx->lds_per_block) {  // round 3's two launches (reference runs; dim > 8192: keys and histogram do not fit one workgroup's LDS together)
        WalkPlan *plan = plans + gen->plan_index;
        if (logs)
            hipLaunchKernelGGL(walk_plan_kernel<true>, dim3(1), dim3(1024), 0, ctx->stream, d_v, total, seg_len, n_seg, per_row, (int32_t)ctx->opt_weighted_tail, plan);
        else
            hipLaunchKernelGGL(walk_plan_kernel<false>, dim3(1), dim3(1024), 0, ctx->stream, d_v, total, seg_len, n_seg, per_row, (int32_t)ctx->opt_weighted_tail, plan);
        MHX_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(walk_build_kernel, dim3((unsigned)gen->sample_size), dim3(256), sizeof(unsigned long long) * (size_t)p2, ctx->stream, plan,
                           reinterpret_cast<const float4 *>(gen->d_aos), dim, p2, gen->s_pad, walk_a, gen->d_walk_c);
        MHX_HIP_CHECK(hipGetLastError());
        *plan_out = plan;
        return MHX_OK;
    }
    const WalkPlan *cur = plans + gen->plan_index;
    WalkPlan *next = plans + (gen->plan_index ^ 1);
    if (logs)
        hipLaunchKernelGGL(walk_plan_build_kernel<true>, dim3((unsigned)gen->sample_size), dim3(1024), lds, ctx->stream, d_v, total, seg_len, n_seg, per_row,
                           (int32_t)ctx->opt_weighted_tail, cur, next, reinterpret_cast<const float4 *>(gen->d_aos), dim, p2, gen->s_pad, walk_a, gen->d_walk_c, gate);
    else
        hipLaunchKernelGGL(walk_plan_build_kernel<false>, dim3((unsigned)gen->sample_size), dim3(1024), lds, ctx->stream, d_v, total, seg_len, n_seg, per_row,
                           (int32_t)ctx->opt_weighted_tail, cur, next, reinterpret_cast<const float4 *>(gen->d_aos), dim, p2, gen->s_pad, walk_a, gen->d_walk_c, gate);
    MHX_HIP_CHECK(hipGetLastError());
    gen->plan_index ^= 1;
    *plan_out = next;
    return MHX_OK;
}

// A walk that went beyond the cached positions leaves the entries it fetched ahead for a round that never came on their way.
// The compiler cannot know that most rows never get there: it puts its s_waitcnt vmcnt where those registers are next written
// -- inside the first cached round of the NEXT row, where the wait also covers this row's result stores (vmcnt counts stores on
// gfx9).  Waiting explicitly between the walk and the stores costs nothing on the common path (no load is out) and leaves
// nothing but stores pending at the loop's back edge.
__device__ __forceinline__ void drain_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0), the other counters untouched

// A walked result's t as the int64 the reference stores (ref: weighted_minhash.py:244, astype(int)).  float -> int64 has no instruction
// (fifteen VALU ones); when every lane's t fits an int32 -- always, for weights a float32 log can come from -- one v_cvt_i32_f32 and a
// sign extension do, and give the same integer.  A NaN or a huge t sends the wave down the general conversion.
__device__ __forceinline__ int64_t t_as_int64(float t) {
    if (__builtin_expect(__all(__builtin_fabsf(t) < 2147483520.0f), 1)) return (int64_t)(int32_t)t;
    return (int64_t)t;
}

// The row scan's three running values over four more entries: six instructions (v_max3 / v_min3 / v_pk_add on the register pairs the
// load delivered).  The C++ form -- fmaxf(fmaxf(mx, fmaxf(x, y)), fmaxf(z, w)) and so on -- came out as seventeen: maxnum must not
// return a quieted signalling NaN, so the compiler canonicalises every loaded value first (v_max_f32 v, v, v), and the sum's
// (x + y) + (z + w) shuffles registers into pairs.  Here a signalling NaN may poison mx or mn of its lane instead of being skipped;
// the sum is NaN whenever any entry is, and a row with a NaN never looks at mx or mn again (scan: has_nan -> nan_row).
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void scan4(const float4 v, float &mx, float &mn, float2v &sum) {
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v.x), "v"(v.y));
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v.z), "v"(v.w));
    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(v.x), "v"(v.y));
    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(v.z), "v"(v.w));
    sum += float2v{v.x, v.y};
    sum += float2v{v.z, v.w};
}

// what a lane holds for its sample: the smallest ln_a so far, its column (ties: the smaller one) and its t
struct Held {
    float ln_a = __builtin_inff();
    float t = 0.0f;
    uint32_t c = 0xFFFFFFFFu;
    __device__ __forceinline__ void take(float a, float tt, uint32_t col) {
        if (a < ln_a || (a == ln_a && col < c)) ln_a = a, t = tt, c = col;
    }
    // the same as a select (no branch): inside the walk's rounds, where a branch per position costs more than the position
    // (& and |, not && and ||: the short-circuit form came out of the compiler as three nested exec-mask regions per position --
    // s_and_saveexec + s_cbranch_execz, ~20 instructions -- where three compares and three scalar mask operations do)
    __device__ __forceinline__ void take_if(bool ok, float a, float tt, uint32_t col) {
        const bool better = ok & ((a < ln_a) | ((a == ln_a) & (col < c)));
        ln_a = better ? a : ln_a, t = better ? tt : t, c = better ? col : c;
    }
    __device__ __forceinline__ void offer(float l, const float4 e, uint32_t col) {  // e = {r, ln_c, beta, .}
        float tt, a;
        evaluate<false>(l, entry_of(e), tt, a);
        take(a, tt, col);
    }
};

constexpr int kCachedChunks = 4;

// The second half of a row, one wave per 64 samples: the row's logs are in LDS (row[c]; -inf: not stored), `list` holds
// n_list columns -- every stored one (all_listed: the row is evaluated entry by entry) or the ones above the cut, which
// are evaluated before the walk.
// `part` of `parts` (all_listed only): the waves of a workgroup that share a chunk of samples take a run of the list each;
// the smallest of their results, taken by Held's own rule, is the row's.  (Sharing out the list of a WALKED row with
// many entries above the cut the same way, the walk starting from what the waves found, was measured: lognormal weights
// 0.675 -> 0.652 ms per 20k rows, config 4 0.50 -> 0.535 ms -- the extra code costs the common case registers.  Not kept.)
template <bool VALS = false>
__device__ __forceinline__ Held walk_row(const float *row, const uint16_t *list, int n_list, bool all_listed, int32_t dim, int32_t ch,
                                         int32_t my, int32_t sample_size, const float4 *__restrict__ walk_a,
                                         const uint32_t *__restrict__ walk_c, const float4 *__restrict__ aos, int32_t s_pad,
                                         const float4 *cache_a, const uint32_t *cache_c, int32_t part, int32_t parts, bool staged = false, int32_t cached = kWalkCached) {
    Held held;
    int j = (int)((int64_t)n_list * part / parts);
    n_list = (int)((int64_t)n_list * (part + 1) / parts);
    for (; j + 4 <= n_list; j += 4) {  // four table entries in flight; t without the division (evaluate_guarded)
        uint32_t c[4];
        float4 e[4];
        float l[4], t[4], a[4];
        bool open = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = list[j + u], e[u] = aos[(int64_t)c[u] * s_pad + my];
#pragma unroll
        for (int u = 0; u < 4; ++u) l[u] = row_log<VALS>(row, c[u], staged), open |= evaluate_guarded<true>(l[u], e[u], t[u], a[u]);
        if (__builtin_expect(__any(open), 0)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) evaluate<false>(l[u], entry_of(e[u]), t[u], a[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) held.take(a[u], t[u], c[u]);
    }
    for (; j < n_list; ++j) {
        const uint32_t c = list[j];
        held.offer(row_log<VALS>(row, c, staged), aos[(int64_t)c * s_pad + my], c);
    }
    if (!all_listed) {
        const int lane = my & (kWave - 1);
        const float4 *wa = walk_a + (int64_t)ch * dim * kWave + lane;
        const uint32_t *wc = walk_c + (int64_t)ch * dim * kWave + lane;
        bool done = my >= sample_size;
        int32_t k = 0;
        // the first positions of the lists are the same for every row: this workgroup keeps them in LDS (a load from the
        // L2 behind the rows' stream takes microseconds; most walks end inside the cached positions)
        constexpr int kU = 4;  // positions per round: evaluated side by side (independent division chains), then taken in order
        const int32_t n_cached = cache_a ? (dim < cached ? dim : cached) / kU * kU : 0;
        for (; k < n_cached; k += kU) {
            float4 e[kU];
            uint32_t c[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) e[u] = cache_a[(k + u) * kWave + lane], c[u] = cache_c[(k + u) * kWave + lane];
            done = done || e[0].x > held.ln_a;  // an equal bound may still hide a tie at a smaller column
            if (!__any(!done)) break;
            {
                // e = {LB, r, ln_c, beta}; t without the division (evaluate_guarded with the hardware's reciprocal, 1 ulp:
                // still inside the proof's margin), which is the longest dependent chain of a round; a column the row does
                // not store (-inf: dropped below) is evaluated at 0 so that it does not count as open.  No branch inside a
                // round: every lane evaluates, a lane that has finished drops what it computed (round 3 had `if (!done)`
                // around the round and around every take: ten exec-mask branches per round, 1 400 cycles for ~150
                // instructions of arithmetic)
                float l[kU], t[kU], a[kU];
                bool open = false;
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    l[u] = row_log<VALS>(row, c[u], staged);
                    open |= evaluate_guarded<true>(l[u] == -__builtin_inff() ? 0.0f : l[u],
                                                   make_float4(e[u].y, e[u].z, e[u].w, __builtin_amdgcn_rcpf(e[u].y)), t[u], a[u]);
                }
                if (__builtin_expect(__any(open && !done), 0)) {
#pragma unroll
                    for (int u = 0; u < kU; ++u) evaluate<false>(l[u], entry_of(make_float4(e[u].y, e[u].z, e[u].w, 0.0f)), t[u], a[u]);
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    done = done || e[u].x > held.ln_a;
                    held.take_if(!done && !(l[u] == -__builtin_inff()), a[u], t[u], c[u]);
                }
            }
        }
        if (k == n_cached && k < dim && __any(!done)) {
            // beyond the cached positions (heavy-tailed rows: dozens of them for the slowest lane of a wave): rounds of kG
            // positions (two: the registers that hold the rows fetched ahead leave no room for four) from registers, the next round's entries on their way from the L2 meanwhile (one position per
            // round with one in flight cost an L2 round trip per position)
            constexpr int kG = 2;
            float4 e[kG];
            uint32_t c[kG];
#pragma unroll
            for (int u = 0; u < kG; ++u) {
                const int32_t at = k + u < dim ? k + u : dim - 1;
                e[u] = wa[(int64_t)at * kWave], c[u] = wc[(int64_t)at * kWave];
            }
            for (; k < dim; k += kG) {
                float4 e_next[kG];
                uint32_t c_next[kG];
#pragma unroll
                for (int u = 0; u < kG; ++u) {
                    const int32_t at = k + kG + u < dim ? k + kG + u : dim - 1;
                    e_next[u] = wa[(int64_t)at * kWave], c_next[u] = wc[(int64_t)at * kWave];
                }
                done = done || e[0].x > held.ln_a;
                if (!__any(!done)) break;
                {
                    float l[kG], t[kG], a[kG];
                    bool open = false;
#pragma unroll
                    for (int u = 0; u < kG; ++u) {
                        l[u] = row_log<VALS>(row, c[u], staged);
                        open |= evaluate_guarded<true>(l[u] == -__builtin_inff() ? 0.0f : l[u],
                                                       make_float4(e[u].y, e[u].z, e[u].w, __builtin_amdgcn_rcpf(e[u].y)), t[u], a[u]);
                    }
                    if (__builtin_expect(__any(open && !done), 0)) {
#pragma unroll
                        for (int u = 0; u < kG; ++u) evaluate<false>(l[u], entry_of(make_float4(e[u].y, e[u].z, e[u].w, 0.0f)), t[u], a[u]);
                    }
#pragma unroll
                    for (int u = 0; u < kG; ++u) {
                        done = done || e[u].x > held.ln_a;  // (a position past the end repeats the last one: taken twice, the same)
                        held.take_if(!done && !(l[u] == -__builtin_inff()), a[u], t[u], c[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < kG; ++u) e[u] = e_next[u], c[u] = c_next[u];
            }
        }
    }
    return held;
}

// The tail of a walk, by the whole wave.  The lanes of a wave walk in lock step, so a round costs the same whether one
// lane still walks or all of them, and heavy-tailed rows leave a handful of lanes walking dozens of positions after the
// others have stopped (lognormal weights, sigma 2: the slowest of 128 lanes goes ~90 positions, the mean ~10).  When few
// lanes are left, each of them in turn gets all 64 lanes: lane j evaluates position k + j of THAT sample's list (one trip to
// the L2 for 64 positions instead of one per two), an inclusive prefix minimum gives every position the value the
// sequential walk would hold on reaching it, the first position whose bound exceeds that is where the walk stops, and the
// smallest (ln_a, column) in front of it -- np.argmin's choice, ties to the smaller column -- is its result.  Same
// evaluations (IEEE division here), same stop rule, same answer as position by position.
template <bool VALS = false>
__device__ __forceinline__ void walk_rescue(const float *row, int32_t dim, int32_t ch, int ls, int32_t k, const float4 *__restrict__ walk_a,
                                            const uint32_t *__restrict__ walk_c, int lane, Held &held, bool staged = false) {
    const auto lane_value = [](float v, int from) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), from)); };  // (the builtin is int -> int)
    float best_a = lane_value(held.ln_a, ls), best_t = lane_value(held.t, ls);
    uint32_t best_c = (uint32_t)__builtin_amdgcn_readlane((int)held.c, ls);
    const float4 *wa = walk_a + (int64_t)ch * dim * kWave + ls;
    const uint32_t *wc = walk_c + (int64_t)ch * dim * kWave + ls;
    for (; k < dim; k += kWave) {
        const int32_t p = k + lane;
        const bool in = p < dim;
        const int32_t pc = in ? p : dim - 1;
        const float4 e = wa[(int64_t)pc * kWave];  // {LB, r, ln_c, beta}
        const uint32_t col = wc[(int64_t)pc * kWave];
        const float l = row_log<VALS>(row, col, staged);
        float t, a;
        evaluate<false>(l == -__builtin_inff() ? 0.0f : l, entry_of(make_float4(e.y, e.z, e.w, 0.0f)), t, a);
        const bool valid = in && !(l == -__builtin_inff()) && a == a;
        const float am = valid ? a + 0.0f : __builtin_inff();
        float pm = am;  // inclusive prefix minimum over the lanes
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const float up = __shfl_up(pm, o);
            if (lane >= o) pm = fminf(pm, up);
        }
        float before = __shfl_up(pm, 1);  // what the sequential walk holds when it reaches this position
        before = lane == 0 ? best_a : fminf(before, best_a);
        const unsigned long long stops = __ballot(!in || e.x > before);
        const int first = stops ? __builtin_ctzll(stops) : kWave;
        const bool cand = valid && lane < first;
        unsigned long long key = cand ? (((unsigned long long)ordered_bits(am) << 32) | col) : ~0ull;
        unsigned long long best_key = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), o) << 32) |
                                             (uint32_t)__shfl_xor((int)(uint32_t)best_key, o);
            best_key = other < best_key ? other : best_key;
        }
        if (best_key != ~0ull) {  // (wave-uniform)
            const float wa_ = from_ordered_bits((uint32_t)(best_key >> 32));
            const uint32_t wc_ = (uint32_t)best_key;
            if (wa_ < best_a || (wa_ == best_a && wc_ < best_c)) {
                const int owner = __builtin_ctzll(__ballot(key == best_key));
                best_a = wa_, best_c = wc_, best_t = lane_value(t, owner);
            }
        }
        if (first < kWave) break;
    }
    if (lane == ls) held.ln_a = best_a, held.t = best_t, held.c = best_c;
}

// walk_row for NC chunks of samples of ONE row at a time, as one instruction stream: the walk is a chain of dependent
// LDS round trips and divisions (4 700 cycles per chunk for ~400 instructions), and a wave that owns a row has the row's
// other chunk to fill the gaps with.  Same rules, same results: every lane's evaluations are computed whether the lane
// still walks or not (a finished lane's are dropped), so a finished lane can at most cause the exact re-evaluation of a
// round, never a different value.  The chunks ch0 .. ch0 + NC - 1 all have their first positions cached (cache_a /
// cache_c: chunk ch0's, the others' behind it).
#ifndef MHX_WALK_KU1
#define MHX_WALK_KU1 4  // positions per round when one chunk is walked alone (A/B builds)
#endif
template <int NC, bool VALS = false>
__device__ __forceinline__ void walk_chunks(const float *row, const uint16_t *list, int n_list, bool all_listed, int32_t dim, int32_t ch0,
                                            int lane, int32_t sample_size, const float4 *__restrict__ walk_a,
                                            const uint32_t *__restrict__ walk_c, const float4 *__restrict__ aos, int32_t s_pad,
                                            const float4 *cache_a, const uint32_t *cache_c, int32_t rescue_lanes, Held (&held)[NC], bool staged = false,
                                            int32_t cached = kWalkCached) {  // (cached: list positions per chunk in cache_a / cache_c; NC = 1 only may differ from kWalkCached)
    int32_t my[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) my[i] = (ch0 + i) * kWave + lane;
    int j = 0;
    constexpr int kL = 4;  // listed columns per turn: their table entries are one trip to the L2 together (8: no faster, 40 more VGPRs)
    for (; j + kL <= n_list; j += kL) {  // the listed columns: the row's entry is the same for every chunk
        uint32_t c[kL];
        float l[kL];
        float4 e[NC][kL];
        float t[NC][kL], a[NC][kL];
        bool open = false;
#pragma unroll
        for (int u = 0; u < kL; ++u) {
            c[u] = list[j + u];
#pragma unroll
            for (int i = 0; i < NC; ++i) e[i][u] = aos[(int64_t)c[u] * s_pad + my[i]];
        }
#pragma unroll
        for (int u = 0; u < kL; ++u) {
            l[u] = row_log<VALS>(row, c[u], staged);
#pragma unroll
            for (int i = 0; i < NC; ++i) open |= evaluate_guarded<true>(l[u], e[i][u], t[i][u], a[i][u]);
        }
        if (__builtin_expect(__any(open), 0)) {
#pragma unroll
            for (int u = 0; u < kL; ++u)
#pragma unroll
                for (int i = 0; i < NC; ++i) evaluate<false>(l[u], entry_of(e[i][u]), t[i][u], a[i][u]);
        }
#pragma unroll
        for (int u = 0; u < kL; ++u)
#pragma unroll
            for (int i = 0; i < NC; ++i) held[i].take(a[i][u], t[i][u], c[u]);
    }
    for (; j < n_list; ++j) {
        const uint32_t c = list[j];
        const float l = row_log<VALS>(row, c, staged);
#pragma unroll
        for (int i = 0; i < NC; ++i) held[i].offer(l, aos[(int64_t)c * s_pad + my[i]], c);
    }
    if (all_listed) return;
    bool done[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) done[i] = my[i] >= sample_size;
    const auto walking = [&]() {
        bool w = false;
#pragma unroll
        for (int i = 0; i < NC; ++i) w |= !done[i];
        return __any(w);
    };
    constexpr int kU = NC == 1 ? MHX_WALK_KU1 : 4;
    const int32_t n_cached = (dim < cached ? dim : cached) / kU * kU;
    int32_t k = 0;
    // One chunk walked alone (NC = 1: the walkers of the fetcher / walker kernel) reads a round's entries one round AHEAD: the stop test at
    // the top of a round then waits for nothing, and a round is one LDS round trip (the row's entries) instead of three -- the compiler had
    // sunk the round's other reads below the test, which needs only the first bound.  (Two chunks as one stream have no registers for it.)
    constexpr bool kAheadCached = NC == 1;
    float4 e_ahead[NC][kU];
    uint32_t c_ahead[NC][kU];
    const auto read_round = [&](int32_t at, float4 (&e)[NC][kU], uint32_t (&c)[NC][kU]) {
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                e[i][u] = cache_a[(i * kWalkCached + at + u) * kWave + lane];
                c[i][u] = cache_c[(i * kWalkCached + at + u) * kWave + lane];
            }
    };
    if (kAheadCached && n_cached > 0) read_round(0, e_ahead, c_ahead);
    for (; k < n_cached; k += kU) {
        float4 e[NC][kU];
        uint32_t c[NC][kU];
        if constexpr (kAheadCached) {
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int u = 0; u < kU; ++u) e[i][u] = e_ahead[i][u], c[i][u] = c_ahead[i][u];
        } else {
            read_round(k, e, c);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) done[i] = done[i] || e[i][0].x > held[i].ln_a;  // an equal bound may still hide a tie at a smaller column
        if (!walking()) break;
        if constexpr (kAheadCached) read_round(k + kU < n_cached ? k + kU : k, e_ahead, c_ahead);  // (the last round reads itself again: nobody looks)
        float l[NC][kU], t[NC][kU], a[NC][kU];
        bool open = false;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                l[i][u] = row_log<VALS>(row, c[i][u], staged);
                open |= evaluate_guarded<true>(l[i][u] == -__builtin_inff() ? 0.0f : l[i][u],
                                               make_float4(e[i][u].y, e[i][u].z, e[i][u].w, __builtin_amdgcn_rcpf(e[i][u].y)), t[i][u], a[i][u]);
            }
        if (__builtin_expect(__any(open), 0)) {
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int u = 0; u < kU; ++u) evaluate<false>(l[i][u], entry_of(make_float4(e[i][u].y, e[i][u].z, e[i][u].w, 0.0f)), t[i][u], a[i][u]);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                done[i] = done[i] || e[i][u].x > held[i].ln_a;
                held[i].take_if(!done[i] && !(l[i][u] == -__builtin_inff()), a[i][u], t[i][u], c[i][u]);
            }
    }
    if (k == n_cached && k < dim && walking()) {  // beyond the cached positions: rounds of two from registers (four: no faster on heavy-tailed rows, whose walks wait for the slowest of 128 lanes), the next round's entries on their way
        constexpr int kG = 2;
        const float4 *wa[NC];
        const uint32_t *wc[NC];
        float4 e[NC][kG];
        uint32_t c[NC][kG];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            wa[i] = walk_a + (int64_t)(ch0 + i) * dim * kWave + lane;
            wc[i] = walk_c + (int64_t)(ch0 + i) * dim * kWave + lane;
#pragma unroll
            for (int u = 0; u < kG; ++u) {
                const int32_t at = k + u < dim ? k + u : dim - 1;
                e[i][u] = wa[i][(int64_t)at * kWave], c[i][u] = wc[i][(int64_t)at * kWave];
            }
        }
        for (; k < dim; k += kG) {
            float4 e_next[NC][kG];
            uint32_t c_next[NC][kG];
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int u = 0; u < kG; ++u) {
                    const int32_t at = k + kG + u < dim ? k + kG + u : dim - 1;
                    e_next[i][u] = wa[i][(int64_t)at * kWave], c_next[i][u] = wc[i][(int64_t)at * kWave];
                }
#pragma unroll
            for (int i = 0; i < NC; ++i) done[i] = done[i] || e[i][0].x > held[i].ln_a;
            if (!walking()) break;
            if (rescue_lanes > 0 && k >= n_cached + 2 * kG) {  // few lanes left after two rounds out here: each of them in turn gets the whole wave (walk_rescue; a tail of a round or two is cheaper in lock step: uniform weights)
                unsigned long long act[NC];
                int n_act = 0;
#pragma unroll
                for (int i = 0; i < NC; ++i) act[i] = __ballot(!done[i]), n_act += __popcll(act[i]);
                if (n_act <= rescue_lanes) {
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        while (act[i]) {
                            const int ls = __builtin_ctzll(act[i]);
                            act[i] &= act[i] - 1;
                            walk_rescue<VALS>(row, dim, ch0 + i, ls, k, walk_a, walk_c, lane, held[i], staged);
                        }
                        done[i] = true;
                    }
                    break;
                }
            }
            float l[NC][kG], t[NC][kG], a[NC][kG];
            bool open = false;
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int u = 0; u < kG; ++u) {
                    l[i][u] = row_log<VALS>(row, c[i][u], staged);
                    open |= evaluate_guarded<true>(l[i][u] == -__builtin_inff() ? 0.0f : l[i][u],
                                                   make_float4(e[i][u].y, e[i][u].z, e[i][u].w, __builtin_amdgcn_rcpf(e[i][u].y)), t[i][u], a[i][u]);
                }
            if (__builtin_expect(__any(open), 0)) {
#pragma unroll
                for (int i = 0; i < NC; ++i)
#pragma unroll
                    for (int u = 0; u < kG; ++u) evaluate<false>(l[i][u], entry_of(make_float4(e[i][u].y, e[i][u].z, e[i][u].w, 0.0f)), t[i][u], a[i][u]);
            }
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int u = 0; u < kG; ++u) {
                    done[i] = done[i] || e[i][u].x > held[i].ln_a;  // (a position past the end repeats the last one: taken twice, the same)
                    held[i].take_if(!done[i] && !(l[i][u] == -__builtin_inff()), a[i][u], t[i][u], c[i][u]);
                    e[i][u] = e_next[i][u], c[i][u] = c_next[i][u];
                }
        }
    }
}

// the results of the waves that shared a chunk, through LDS: part p >= 1 of chunk ch writes slot (p - 1) * chunks + ch
// (192 words each), part 0 takes them
__device__ __forceinline__ void put(float *shared, int slot, int lane, const Held &h) {
    float *at = shared + slot * (3 * kWave) + lane;
    at[0] = h.ln_a, at[kWave] = h.t, at[2 * kWave] = __uint_as_float(h.c);
}
__device__ __forceinline__ void take_parts(const float *shared, int32_t ch, int32_t chunks, int32_t parts, int lane, Held &h) {
    for (int32_t p = 1; p < parts; ++p) {
        const float *at = shared + ((p - 1) * chunks + ch) * (3 * kWave) + lane;
        h.take(at[0], at[kWave], __float_as_uint(at[2 * kWave]));
    }
}

// a row with a NaN among its logs: numpy's argmin, the first NaN wins (every stored entry, in column order)
template <bool VALS = false>
__device__ __forceinline__ void nan_row(const float *row, int32_t dim, int32_t my, const float4 *__restrict__ aos, int32_t s_pad,
                                        int64_t &k_out, int64_t &t_out, bool staged = false) {
    Best best;
    best.ln_a = 0.0f, best.t = 0.0f, best.k = -1;
    for (int32_t c = 0; c < dim; ++c) {
        const float l = row_log<VALS>(row, (uint32_t)c, staged);
        if (l == -__builtin_inff()) continue;
        consider(best, l, entry_of(aos[(int64_t)c * s_pad + my]), c);
    }
    k_out = best.k, t_out = (int64_t)best.t;
}

constexpr int kPre = 4;    // 16-byte loads per thread that hold a row (256 threads: dim <= 4096)
constexpr int kAhead = 3;  // rows fetched ahead of the one being walked: the bytes in flight that HBM wants (48 KB per workgroup)

template <bool LOGS, bool AHEAD>
__global__ __launch_bounds__(256, 4) void weighted_walk_dense_kernel(const float *__restrict__ x, int64_t n_rows, int32_t dim,
                                                                  const WalkPlan *__restrict__ plan,
                                                                  const float4 *__restrict__ walk_a,
                                                                  const uint32_t *__restrict__ walk_c,
                                                                  const float4 *__restrict__ aos, int32_t sample_size,
                                                                  int32_t s_pad, int32_t list_cap, int32_t direct_permille,
                                                                  int64_t *__restrict__ out, uint8_t *__restrict__ nonempty, int32_t debug,
                                                                  int32_t split) {
    extern __shared__ float row[];                                             // dim logs of the row (-inf: not stored)
    uint16_t *list = reinterpret_cast<uint16_t *>(row + ((dim + 3) & ~3));    // columns (list_cap of them)
    __shared__ int s_nnz, s_nout, s_nan;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
    const int32_t chunks = s_pad / kWave;
    const int32_t parts = split && chunks * 2 <= n_waves && n_waves <= 4 ? n_waves / chunks : 1;  // waves per chunk of samples
    const float lcut = plan->lcut;
    // behind the row and the list (16-byte aligned): the cached list positions of the first n_cc chunks
    const int32_t n_cc = chunks < kCachedChunks ? chunks : kCachedChunks;
    float4 *s_cache_a = reinterpret_cast<float4 *>(row + ((dim + 3) & ~3) + ((list_cap + 7) & ~7) / 2);
    uint32_t *s_cache_c = reinterpret_cast<uint32_t *>(s_cache_a + n_cc * kWalkCached * kWave);
    for (int j = tid; j < n_cc * kWalkCached * kWave; j += blockDim.x) {
        const int ch = j / (kWalkCached * kWave), k = j / kWave % kWalkCached;
        if (ch < chunks && k < dim) {
            s_cache_a[j] = walk_a[((int64_t)ch * dim + k) * kWave + (j & (kWave - 1))];
            s_cache_c[j] = walk_c[((int64_t)ch * dim + k) * kWave + (j & (kWave - 1))];
        }
    }
    // AHEAD: 16-byte loads and a whole row in kPre of them per thread (the launcher checked): rows are fetched kAhead rows
    // ahead, into registers.  Every thread always issues exactly kPre loads per row (clamped to the matrix), so that the
    // number of loads in flight behind a row's is known at compile time: otherwise the wait for a row's registers is a
    // wait for every load issued since, the next rows' included, and nothing is ahead of anything.
    const bool wide = AHEAD || ((dim & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int64_t stride = gridDim.x;
    const auto fetch = [&](float4 (&pre)[kPre], int64_t d) {
        const float *src = x + (d < n_rows ? d : n_rows - 1) * dim;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int c = (u * (int)blockDim.x + tid) * 4;
            pre[u] = *reinterpret_cast<const float4 *>(src + (c < dim ? c : dim - 4));
        }
    };
    // one row: its logs from `pre` (or memory) into LDS, then `pre` refilled with the row kAhead strides on, then the walk
    const auto one_row = [&](float4 (&pre)[kPre], int64_t d) {
        if (tid == 0) s_nnz = 0, s_nout = 0, s_nan = 0;
        __syncthreads();
        int nnz = 0;
        bool nan = false;
        const auto take = [&](int c, float v) -> float {
            const float l = LOGS ? v : np_logf(v);
            const bool stored = LOGS ? !(l == -__builtin_inff()) : (v != 0.0f);  // scipy's nonzero(): NaN stays
            nnz += stored;
            nan |= l != l;
            if (l > lcut) {  // +inf too
                const int at = atomicAdd(&s_nout, 1);
                if (at < list_cap) list[at] = (uint16_t)c;
            }
            return stored ? l : -__builtin_inff();
        };
        const auto take4 = [&](int c, const float4 v) {
            if (debug == 2) {  // profiling only: the copy without the scan
                *reinterpret_cast<float4 *>(row + c) = v;
                nnz += 4;
                return;
            }
            float4 l;
            l.x = take(c, v.x), l.y = take(c + 1, v.y), l.z = take(c + 2, v.z), l.w = take(c + 3, v.w);
            *reinterpret_cast<float4 *>(row + c) = l;
        };
        if (AHEAD) {
#pragma unroll
            for (int u = 0; u < kPre; ++u) {
                const int c = (u * (int)blockDim.x + tid) * 4;
                if (c < dim) take4(c, pre[u]);
            }
        } else if (wide) {
            for (int c = tid * 4; c < dim; c += blockDim.x * 4) take4(c, *reinterpret_cast<const float4 *>(x + d * dim + c));
        } else {
            for (int c = tid; c < dim; c += blockDim.x) row[c] = take(c, x[d * dim + c]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o);
        if (lane == 0 && nnz) atomicAdd(&s_nnz, nnz);
        if (__any(nan) && lane == 0) s_nan = 1;
        __syncthreads();
        const int n_stored = s_nnz, n_out = s_nout;
        int n_list = n_out;
        const bool has_nan = s_nan != 0;
        // A row with few stored entries is cheaper entry by entry than through the walk, which meets its absent columns
        // too; a row with more entries above the cut than the list holds has no valid stop rule: entry by entry as well.
        bool by_entry = n_out > list_cap || (int64_t)n_stored * 1000 <= (int64_t)direct_permille * dim;
        const bool listable = by_entry && !has_nan && n_stored > 0 && n_stored <= list_cap;
        if (listable) {  // list the stored columns (in any order)
            __syncthreads();
            if (tid == 0) s_nout = 0;
            __syncthreads();
            for (int c = tid; c < dim; c += blockDim.x)
                if (!(row[c] == -__builtin_inff())) list[atomicAdd(&s_nout, 1)] = (uint16_t)c;
            __syncthreads();
            n_list = n_stored;
        }
        if (parts > 1 && listable && debug != 1 && dim >= (parts - 1) * chunks * 3 * kWave) {  // (the row's LDS must hold the waves' results)
            // (workgroup-uniform) a row evaluated entry by entry from its list: the waves that share a chunk of samples
            // take a run of the list each (with one wave per chunk half the workgroup would idle at 128 samples); their
            // results meet in the row's own LDS once every wave has finished reading it.  (The walk is not shared out this
            // way: measured, twice the speculative evaluations and half the cached positions per wave cost 0.35 ms on
            // config 4.)
            const int32_t ch = wave % chunks, part = wave / chunks, my = ch * kWave + lane;
            Held held;
            if (part < parts)
                held = walk_row(row, list, n_list, true, dim, ch, my, sample_size, walk_a, walk_c, aos, s_pad, nullptr, nullptr, part, parts);
            __syncthreads();
            float *s_parts = row;
            if (part > 0 && part < parts) put(s_parts, (part - 1) * chunks + ch, lane, held);
            __syncthreads();
            if (part == 0) {
                take_parts(s_parts, ch, chunks, parts, lane, held);
                if (my < sample_size) {
                    int64_t *o = out + (d * sample_size + my) * 2;
                    o[0] = held.c;
                    o[1] = (int64_t)held.t;
                }
            }
        } else
        for (int32_t ch = wave; ch < chunks; ch += n_waves) {
            const int32_t my = ch * kWave + lane;
            int64_t k_out = 0, t_out = 0;
            if (n_stored == 0) {
                // nothing stored: (0, 0), and the row is reported empty
            } else if (has_nan) {
                nan_row(row, dim, my, aos, s_pad, k_out, t_out);
            } else if (by_entry && !listable) {  // every stored entry, in column order
                Held held;
                for (int32_t c = 0; c < dim; ++c) {
                    const float l = row[c];
                    if (!(l == -__builtin_inff())) held.offer(l, aos[(int64_t)c * s_pad + my], (uint32_t)c);
                }
                k_out = held.c, t_out = (int64_t)held.t;
            } else if (debug != 1) {
                const Held held = walk_row(row, list, n_list, by_entry, dim, ch, my, sample_size, walk_a, walk_c, aos, s_pad,
                                           ch < n_cc ? s_cache_a + ch * kWalkCached * kWave : nullptr,
                                           ch < n_cc ? s_cache_c + ch * kWalkCached * kWave : nullptr, 0, 1);
                k_out = held.c, t_out = (int64_t)held.t;
            }
            if (my < sample_size) {
                int64_t *o = out + (d * sample_size + my) * 2;
                o[0] = k_out;
                o[1] = t_out;
            }
        }
        if (tid == 0) nonempty[d] = n_stored > 0 ? 1 : 0;
        // The refill goes out behind the walk, not before it: vector loads complete in order, so a wait for one of the
        // walk's own loads would otherwise sit out the whole HBM latency of a row that is not needed for three rows.
        if (AHEAD) fetch(pre, d + kAhead * stride);
        __syncthreads();
    };
    float4 pre0[kPre], pre1[kPre], pre2[kPre];
    static_assert(kAhead == 3, "three register buffers below");
    const int64_t d0 = blockIdx.x;
    if (AHEAD) {
        fetch(pre0, d0);
        fetch(pre1, d0 + stride);
        fetch(pre2, d0 + 2 * stride);
    }
    for (int64_t d = d0; d < n_rows; d += kAhead * stride) {
        one_row(pre0, d);
        if (d + stride < n_rows) one_row(pre1, d + stride);
        if (d + 2 * stride < n_rows) one_row(pre2, d + 2 * stride);
    }
}


// ---- dense rows, one WAVE per row (round 4) -----------------------------------------------------------------------
// The workgroup-per-row kernel above spends a row's time waiting: four waves stage, a barrier, two of them walk (at 128
// samples) while the other two idle, a barrier -- 8 700 cycles per row with one workgroup per CU (stage 2 200, reduce +
// barrier 800, walk 4 700, the rest barriers: tools/experiments/r04_walk_phase_cycles.patch), of which the VALU issues a
// sixth, and LDS (a row, a list and 20 KB of cached walk tables per workgroup) caps a CU at four rows in flight.
// Here a wave owns a row: it fetches it (two rows ahead, into registers), stages it in its OWN 16-KB stripe of LDS,
// scans it with ballots (no atomics, no barrier: LDS operations of one wave complete in order) and walks every chunk of
// samples itself.  The waves of a workgroup share nothing but the cached walk tables (read-only after the start), so
// one workgroup of eight waves per CU keeps EIGHT rows in flight in the same 160 KB, no wave ever idles at a barrier,
// and the row's stream (16 x 1-KB loads per row and wave) overlaps with the walks of the seven other waves.
// Same arithmetic, same rules, same helper (walk_row) as the kernel above; rows it does not take (dim > 4096 or not a
// multiple of 4, fewer than two stripes fitting the LDS) stay with that kernel.
// FETCH (A/B, option weighted.refill): bit 0 = the next row's loads go out right after staging instead of behind the walk, bit 1 = non-temporal loads
// SPLIT (round 5): FETCHER waves stream rows from HBM through their registers into the stripes, WALKER waves scan and walk them.
// Vector loads of one wave complete in order, so a wave that does both either has no row in flight while it walks or has its walk's
// table loads (L2 hits) queue behind a row's HBM latency; and the walk itself is a chain of dependent LDS round trips that leaves the
// SIMD idle (one-wave-per-row kernel: walkers alone take 0.35 ms for config 4 at two waves per SIMD, the rows' stream 0.26).  Split,
// the walkers need no registers for rows in flight: sixteen waves of <= 128 VGPRs fit a CU instead of eight of 250 -- twelve walkers
// (a row's two chunks of 64 samples go to two waves) and four fetchers.  See the SPLIT == 2 block below for the hand-over.
constexpr int kHandWords = 12;             // SPLIT 2: per stripe {ready, scanned, what the fetcher found, n_stored, n_list, flags, done[0 .. 5] (one per chunk of samples)}
constexpr int kHandDone = 6;
constexpr int kSplitHandWords2 = 128;      // ... eight stripes at most
constexpr uint32_t kSpinLimit = 1u << 24;  // polls (each ~100 cycles) after which a wait traps -- with option weighted.debug bit 3 only
template <bool LOGS, int NV, bool PAIRS, int FETCH = 0, int SPLIT = 0>  // NV: 16-byte loads per lane that hold a row (dim <= 256 NV); PAIRS: two chunks of samples walked as one instruction stream
__global__ __launch_bounds__(SPLIT != 0 ? 1024 : 512) void weighted_walk_wave_kernel(const float *__restrict__ x, int64_t n_rows, int32_t dim,
                                                                 const WalkPlan *__restrict__ plan, const float4 *__restrict__ walk_a,
                                                                 const uint32_t *__restrict__ walk_c, const float4 *__restrict__ aos,
                                                                 int32_t sample_size, int32_t s_pad, int32_t list_cap, int32_t direct_permille,
                                                                 int32_t stripe_words, int32_t rescue_lanes, int64_t *__restrict__ out,
                                                                 uint8_t *__restrict__ nonempty, int32_t debug, int32_t split_stripes) {
    extern __shared__ float lds[];  // cached list positions of the first n_cc chunks | per wave: row[dim] | list[list_cap] u16
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
    const int32_t chunks = s_pad / kWave;
    const int32_t n_cc = chunks < kCachedChunks ? chunks : kCachedChunks;
    const int32_t wcached = SPLIT == 2 ? split_stripes >> 8 & 255 : kWalkCached;  // list positions per chunk kept in LDS (SPLIT 2: the launcher's choice)
    float4 *s_cache_a = reinterpret_cast<float4 *>(lds);
    uint32_t *s_cache_c = reinterpret_cast<uint32_t *>(s_cache_a + n_cc * wcached * kWave);
    for (int j = tid; j < n_cc * wcached * kWave; j += blockDim.x) {
        const int ch = j / (wcached * kWave), k = j / kWave % wcached;
        if (ch < chunks && k < dim) {
            s_cache_a[j] = walk_a[((int64_t)ch * dim + k) * kWave + (j & (kWave - 1))];
            s_cache_c[j] = walk_c[((int64_t)ch * dim + k) * kWave + (j & (kWave - 1))];
        }
    }
    if (SPLIT == 2 && tid < kSplitHandWords2) reinterpret_cast<uint32_t *>(lds + 5 * n_cc * wcached * kWave)[tid] = 0;
    __syncthreads();  // the only barrier of the kernel
    float *row = lds + 5 * n_cc * wcached * kWave + (int64_t)wave * stripe_words;  // (SPLIT 2: set below)
    uint16_t *list = reinterpret_cast<uint16_t *>(row + ((dim + 3) & ~3));
    const float lcut = plan->lcut;
    // values in (LOGS = false): the stripe holds the VALUES and the log is taken of the entries a walk meets (row_log<true>), not
    // of all of them -- so "above the cut" is first asked of the value: v <= vcut guarantees np_logf(v) <= lcut (vcut = exp of the
    // cut less 10^-5 relative: numpy's log is within 4 ulp of the true one, expf within 2), and only values above vcut have their
    // log taken and compared
    // (below exp(-87) the values are denormal and a rounded exp is no bound: there every positive value has its log taken)
    const float vcut = LOGS ? 0.0f : (lcut >= 88.0f ? __FLT_MAX__ : lcut <= -87.0f ? 0.0f : expf(lcut - 1e-5f * fmaxf(1.0f, fabsf(lcut))));
    const int64_t stride = (int64_t)gridDim.x * n_waves;
    // every lane always issues exactly NV loads per row (clamped to the matrix and to the row), so that the number of loads
    // in flight behind a row's is known at compile time and the wait for a row is not a wait for the one behind it
    const auto fetch = [&](float4 (&pre)[NV], int64_t d) {
        const float *src = x + (d < n_rows ? d : n_rows - 1) * dim;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = (u * kWave + lane) * 4;
            if constexpr ((FETCH & 2) != 0) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(src + (c < dim ? c : dim - 4)));
                pre[u] = make_float4(v.x, v.y, v.z, v.w);
            } else {
                pre[u] = *reinterpret_cast<const float4 *>(src + (c < dim ? c : dim - 4));
            }
        }
    };
    const auto stage = [&](float4 (&pre)[NV], bool &lane_above, bool &lane_odd) {
        // stage + scan: the row's logs into the stripe (-inf: not stored).  What the scan looks for is rare, so it is kept
        // on the scalar unit: a lane's "seen one" flags are lane masks in scalar registers, OR-ed per element (s_or_b64; 128
        // explicit ballots instead made the compiler keep every mask alive: 440 spilled SGPRs), and only a row that has an
        // entry above the cut, a NaN or an absent entry is looked at again, from LDS.
        // A lane behind the row's end holds a clamped copy of the row's last four entries (fetch): what it finds is what the lane
        // that really owns them finds, and what it stores goes where that lane stores the same bytes -- so neither the scan nor the
        // store is predicated.  (Round 4 wrote `if (in)` around the store: sixteen exec-mask regions per row, each with its own
        // conservative s_waitcnt vmcnt(0) -- a wait for the rows fetched ahead as well -- and the masks spilled to VGPR lanes.)
        // The three questions the scan asks -- an entry above the cut? a NaN? an entry that is not stored (LOGS: -inf; values: <=
        // 0)? -- are asked of a maximum, a sum and a minimum over the lane's 64 entries (v_max3 / v_min3 / v_add: ~100 VALU
        // instructions, no scalar ones) instead of two compares and two scalar ORs per entry: max and min skip NaNs, the sum
        // carries them (and is NaN without one only for +inf and -inf together, a row the minimum flags anyway).
        float mx = -__builtin_inff(), mn = __builtin_inff();
        float2v sum2 = {0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = (u * kWave + lane) * 4;
            const float4 v = pre[u];
            scan4(v, mx, mn, sum2);
            *reinterpret_cast<float4 *>(row + (c < dim ? c : dim - 4)) = v;
        }
        const float sum = sum2.x + sum2.y;
        lane_above = LOGS ? mx > lcut : mx > vcut;                               // (+inf too; values: candidates, their logs are taken below)
        lane_odd = sum != sum || (LOGS ? mn == -__builtin_inff()                 // a NaN, or an entry that is not stored: log -inf,
                                       : mn <= 0.0f);                            // a value that is zero (+-0) or negative (its log: NaN)
    };
    // everything behind the staging: what the scan found is looked at again from LDS, the lists are built, the chunks of samples walked
    struct Scanned {
        int n_stored, n_list;
        bool has_nan, by_entry, listable, logs_staged;
    };
    const auto scan = [&](bool any_above, bool any_odd) -> Scanned {
        int n_stored = dim, n_out = 0;
        bool has_nan = false;
        if (any_odd) {  // (wave-uniform) count what is stored, look for NaNs
            n_stored = 0;
            unsigned long long nan_lanes = 0;
            for (int c0 = 0; c0 < dim; c0 += kWave) {
                const int c = c0 + lane;
                const float l = c < dim ? row[c] : (LOGS ? -__builtin_inff() : 0.0f);
                n_stored += __popcll(__ballot(LOGS ? !(l == -__builtin_inff()) : l != 0.0f));
                nan_lanes |= __ballot(LOGS ? l != l : (l != l || l < 0.0f));  // (the log of a negative value is a NaN)
            }
            has_nan = nan_lanes != 0;
        }
        // values in: a row with entries above the cut -- a heavy tail: its walks are long (dozens of positions per sample) -- and
        // a row gone through entry by entry (every sample meets every stored entry) have all their logs taken now, in place; any
        // other row keeps its values and has the log taken of what its walks meet
        const bool few_stored = (int64_t)n_stored * 1000 <= (int64_t)direct_permille * dim;
        bool logs_staged = LOGS;
        if (!LOGS && (any_above || few_stored)) {
#pragma unroll 1
            for (int u = 0; u < NV; ++u) {  // (not unrolled: it runs beside two rows held in registers)
                const int c = (u * kWave + lane) * 4;
                const bool in = c < dim;
                float4 v = in ? *reinterpret_cast<const float4 *>(row + c) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                const bool plain = np_logf_is_normal(v.x) && np_logf_is_normal(v.y) && np_logf_is_normal(v.z) && np_logf_is_normal(v.w);
                if (__builtin_expect(__all(plain), 1)) v = make_float4(np_logf_normal(v.x), np_logf_normal(v.y), np_logf_normal(v.z), np_logf_normal(v.w));
                else v = make_float4(np_logf(v.x), np_logf(v.y), np_logf(v.z), np_logf(v.w));  // zeros (-> -inf: absent), denormals, negatives, NaNs
                if (in) *reinterpret_cast<float4 *>(row + c) = v;
            }
            logs_staged = true;
        }
        if (any_above) {  // (wave-uniform; the stripe holds logs by now) list the columns above the cut
            for (int c0 = 0; c0 < dim; c0 += kWave) {
                const int c = c0 + lane;
                const bool is_above = c < dim && row[c] > lcut;
                const unsigned long long mask = __ballot(is_above);
                if (is_above) {
                    const int at = n_out + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (at < list_cap) list[at] = (uint16_t)c;
                }
                n_out += __popcll(mask);
            }
        }
        int n_list = n_out;
        // as in the workgroup-per-row kernel: few stored entries, or more above the cut than the list holds -> entry by entry
        const bool by_entry = n_out > list_cap || few_stored;
        const bool listable = by_entry && !has_nan && n_stored > 0 && n_stored <= list_cap;
        if (listable) {  // list the stored columns (ascending)
            int at = 0;
            for (int c0 = 0; c0 < dim; c0 += kWave) {
                const int c = c0 + lane;
                const bool keep = c < dim && (logs_staged ? !(row[c] == -__builtin_inff()) : row[c] != 0.0f);
                const unsigned long long mask = __ballot(keep);
                if (keep) list[at + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint16_t)c;
                at += __popcll(mask);
            }
            n_list = n_stored;
        }
        return Scanned{n_stored, n_list, has_nan, by_entry, listable, logs_staged};
    };
    // the chunks [ch_begin, ch_end) of samples of a scanned row
    const auto walk = [&](int64_t d, const Scanned &sc, int32_t ch_begin, int32_t ch_end) {
        const int n_stored = sc.n_stored, n_list = sc.n_list;
        const bool has_nan = sc.has_nan, by_entry = sc.by_entry, listable = sc.listable, logs_staged = sc.logs_staged;
        const bool walked = n_stored > 0 && !has_nan && !(by_entry && !listable);
        {
            constexpr bool VALS = !LOGS;  // (values in: the stripe holds values unless this row's logs were taken in place: logs_staged)
            for (int32_t ch = ch_begin; ch < ch_end; ++ch) {
                if (PAIRS && walked && ch + 1 < n_cc && ch + 1 < ch_end) {  // (wave-uniform) two chunks of samples as one instruction stream
                    Held held[2];
                    walk_chunks<2, VALS>(row, list, n_list, by_entry, dim, ch, lane, sample_size, walk_a, walk_c, aos, s_pad,
                                         s_cache_a + ch * wcached * kWave, s_cache_c + ch * wcached * kWave, rescue_lanes, held, logs_staged);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int32_t my = (ch + i) * kWave + lane;
                        if (my < sample_size) {
                            int64_t *o = out + (d * sample_size + my) * 2;
                            o[0] = held[i].c;
                            o[1] = t_as_int64(held[i].t);
                        }
                    }
                    ++ch;
                    continue;
                }
                if (SPLIT == 2 && walked && ch < n_cc) {  // one chunk, through the same stream as the pairs: the tail of a heavy-tailed row by the whole wave (walk_rescue)
                    Held held[1];
                    walk_chunks<1, VALS>(row, list, n_list, by_entry, dim, ch, lane, sample_size, walk_a, walk_c, aos, s_pad,
                                         s_cache_a + ch * wcached * kWave, s_cache_c + ch * wcached * kWave, rescue_lanes, held, logs_staged, wcached);
                    drain_loads();
                    const int32_t my1 = ch * kWave + lane;
                    if (my1 < sample_size) {
                        int64_t *o = out + (d * sample_size + my1) * 2;
                        o[0] = held[0].c;
                        o[1] = t_as_int64(held[0].t);
                    }
                    continue;
                }
                const int32_t my = ch * kWave + lane;
                int64_t k_out = 0, t_out = 0;
                if (n_stored == 0) {
                    // nothing stored: (0, 0), and the row is reported empty
                } else if (has_nan) {
                    nan_row<VALS>(row, dim, my, aos, s_pad, k_out, t_out, logs_staged);
                } else if (by_entry && !listable) {  // every stored entry, in column order
                    Held held;
                    for (int32_t c = 0; c < dim; ++c) {
                        const float l = row_log<VALS>(row, (uint32_t)c, logs_staged);
                        if (!(l == -__builtin_inff())) held.offer(l, aos[(int64_t)c * s_pad + my], (uint32_t)c);
                    }
                    k_out = held.c, t_out = (int64_t)held.t;
                } else {
                    const Held held = walk_row<VALS>(row, list, n_list, by_entry, dim, ch, my, sample_size, walk_a, walk_c, aos, s_pad,
                                                     ch < n_cc ? s_cache_a + ch * wcached * kWave : nullptr,
                                                     ch < n_cc ? s_cache_c + ch * wcached * kWave : nullptr, 0, 1, logs_staged, wcached);
                    k_out = held.c, t_out = (int64_t)held.t;
                }
                if (my < sample_size) {
                    int64_t *o = out + (d * sample_size + my) * 2;
                    o[0] = k_out;
                    o[1] = t_out;
                }
            }
        }
    };
    // (scan and walk are called where they are needed, not through a third lambda: a closure that captures closures has its captures
    // materialised in scratch memory -- 504 bytes of it and a kernel 3.5 x slower, measured)
#define MHX_SCAN_AND_WALK(d_, above_, odd_)                                                                   \
    do {                                                                                                      \
        const Scanned sc_ = scan(above_, odd_);                                                               \
        if (debug != 1) walk(d_, sc_, 0, chunks); /* (debug 1, profiling only: staged and scanned, not walked) */ \
        if (lane == 0) nonempty[d_] = sc_.n_stored > 0 ? 1 : 0;                                               \
    } while (0)
    if constexpr (SPLIT == 2) {
        // n_walk walkers in GROUPS of `chunks` (one walker per chunk of 64 samples; a group walks one row at a time), n_fetch fetchers, n_stripes
        // stripes.  The workgroup's i-th row (blockIdx.x + i * gridDim.x) goes to stripe i % n_stripes as that stripe's (i / n_stripes)-th row, is
        // fetched by fetcher i % n_fetch and walked by group i % n_groups.  Every hand-over is a counter in LDS with one writer at a time:
        // ready (the fetcher that deposited the stripe's k-th row stores k + 1), done[c] (the walker of chunk c stores k + 1 when it has finished
        // it), scanned (the walker of chunk 0, only for rows the scan has to look at again).  A deposit waits for every chunk's walker to be done
        // with the stripe's row before; rows only ever wait for rows with a smaller i, so nobody waits in a circle.  One stripe more than
        // groups: a row that waits in LDS (the fetchers' and the walkers' rates are close; 3 % on values in and on heavy-tailed rows).
        const int n_stripes = split_stripes & 255, n_fetch = split_stripes >> 16 & 15, n_walk = n_waves - n_fetch;
        const int n_group = chunks, n_pairs = n_walk / n_group;  // (n_pairs: the number of groups -- pairs at 128 samples; chunks divides n_walk: the launcher's rule)
        uint32_t *hands = reinterpret_cast<uint32_t *>(lds + 5 * n_cc * wcached * kWave);
        const auto wait_at_least = [&](uint32_t *word, uint32_t want) {
            for (uint32_t polls = 0; __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want; ++polls) {
                // (the hand-over cannot deadlock -- every counter has one writer and row i waits only for rows before it -- so a long wait is a
                // stall, not a bug: a debugger, a serialising profiler, a preempted queue.  Production waits it out; only option weighted.debug
                // bit 3 (8) turns the wait's limit into a trap, for work on the protocol itself.  ADVICE r5)
                if (polls > kSpinLimit && (debug & 8) != 0) __builtin_trap();
                __builtin_amdgcn_s_sleep(2);
            }
        };
        const auto post = [&](uint32_t *word, uint32_t value) { __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };
        const int64_t gstride = gridDim.x;
        const int64_t my_rows = n_rows > blockIdx.x ? (n_rows - blockIdx.x + gstride - 1) / gstride : 0;  // rows blockIdx.x + i * gridDim.x
        float *stripes = lds + 5 * n_cc * wcached * kWave + kSplitHandWords2;
        if (wave >= n_walk) {
            const int f = wave - n_walk;
            float4 pre[NV];
            if (f < my_rows) fetch(pre, blockIdx.x + (int64_t)f * gstride);
            int st = f % n_stripes;
            uint32_t k = (uint32_t)(f / n_stripes);  // row i is stripe i % n_stripes' (i / n_stripes)-th: kept by counting
            for (int64_t i = f; i < my_rows; i += n_fetch) {
                uint32_t *hand = hands + st * kHandWords;
                float *dst = stripes + (int64_t)st * stripe_words;
                float mx = -__builtin_inff(), mn = __builtin_inff();
                float2v sum2 = {0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < NV; ++u) scan4(pre[u], mx, mn, sum2);
                const float sum = sum2.x + sum2.y;
                const bool any_above = __any(LOGS ? mx > lcut : mx > vcut);
                const bool any_odd = __any(sum != sum || (LOGS ? mn == -__builtin_inff() : mn <= 0.0f));
                for (int w = 0; w < n_group; ++w) wait_at_least(hand + kHandDone + w, k);
                if (debug != 4 || k == 0) {  // (debug 4, profiling only, results wrong: every stripe keeps its first row -- the walkers' side alone)
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const int c = (u * kWave + lane) * 4;
                        *reinterpret_cast<float4 *>(dst + (c < dim ? c : dim - 4)) = pre[u];
                    }
                    if (lane == 0) hand[2] = (any_above ? 1u : 0u) | (any_odd ? 2u : 0u);
                }
                post(hand, k + 1);
                if (i + n_fetch < my_rows && (debug != 4 || i + n_fetch < n_stripes)) fetch(pre, blockIdx.x + (i + n_fetch) * gstride);
                st += n_fetch;
                while (st >= n_stripes) st -= n_stripes, ++k;
            }
        } else {
            drain_loads();  // (a walker has issued no load: this only tells the compiler so -- the structurised control flow runs from the fetchers' loop
                            // into this branch, and with it the compiler's idea that sixteen loads may be out: its s_waitcnt vmcnt(2 / 1 / 0) in the
                            // walkers' cached rounds waited for the walkers' own result stores)
            const int pair = wave % n_pairs, c = wave / n_pairs;  // group `pair`, chunk c; the group's rows: pair, pair + n_pairs, ... -- row i in stripe i % n_stripes
            int st = pair % n_stripes;
            uint32_t k = (uint32_t)(pair / n_stripes);  // (counted, not i / n_stripes: a 64-bit division by a run-time value is ~60 scalar instructions per row)
            for (int64_t i = pair; i < my_rows; i += n_pairs) {
                uint32_t *hand = hands + st * kHandWords;
                row = stripes + (int64_t)st * stripe_words;
                list = reinterpret_cast<uint16_t *>(row + ((dim + 3) & ~3));
                const int64_t d = blockIdx.x + i * gstride;
                wait_at_least(hand, k + 1);
                const uint32_t found = hand[2];
                Scanned sc{dim, 0, false, false, false, LOGS};
                if (found != 0 || (int64_t)dim * 1000 <= (int64_t)direct_permille * dim) {  // (wave-uniform) the scan has to look again: walker 0 does, and says what it saw
                    if (c == 0) {
                        sc = scan((found & 1) != 0, (found & 2) != 0);
                        if (lane == 0) {
                            hand[3] = (uint32_t)sc.n_stored, hand[4] = (uint32_t)sc.n_list;
                            hand[5] = (sc.has_nan ? 1u : 0u) | (sc.by_entry ? 2u : 0u) | (sc.listable ? 4u : 0u) | (sc.logs_staged ? 8u : 0u);
                        }
                        post(hand + 1, k + 1);
                    } else {
                        wait_at_least(hand + 1, k + 1);
                        const uint32_t bits = hand[5];
                        sc = Scanned{(int)hand[3], (int)hand[4], (bits & 1) != 0, (bits & 2) != 0, (bits & 4) != 0, (bits & 8) != 0};
                    }
                }
                if (debug != 1 && debug != 2) walk(d, sc, c, c + 1);
                if (c == 0 && lane == 0) nonempty[d] = sc.n_stored > 0 ? 1 : 0;
                post(hand + kHandDone + c, k + 1);
                st += n_pairs;
                while (st >= n_stripes) st -= n_stripes, ++k;
            }
        }
        return;
    }
    const auto one_row = [&](float4 (&pre)[NV], int64_t d) {
        bool lane_above, lane_odd;
        stage(pre, lane_above, lane_odd);
        // (FETCH bit 0) the refill goes out as soon as the registers are free: two rows per wave are in flight for the whole
        // of the scan and the walk, at the price of the walk's first table load waiting behind it
        if constexpr ((FETCH & 1) != 0) fetch(pre, d + 2 * stride);
        if (debug == 2) {  // profiling only (results are wrong): rows fetched and staged, nothing else
            if (lane == 0) nonempty[d] = lane_above || lane_odd ? 1 : 0;
        } else {
            MHX_SCAN_AND_WALK(d, __any(lane_above), __any(lane_odd));
        }
        // the refill goes out behind the walk (vector loads complete in order: a walk's own table load must not sit out
        // the HBM latency of a row that is not needed for two rows)
        if constexpr ((FETCH & 1) == 0) fetch(pre, d + 2 * stride);
    };
    float4 pre0[NV], pre1[NV];
    const int64_t d0 = (int64_t)blockIdx.x * n_waves + wave;
    fetch(pre0, d0);
    fetch(pre1, d0 + stride);
    for (int64_t d = d0; d < n_rows; d += 2 * stride) {
        one_row(pre0, d);
        if (d + stride < n_rows) one_row(pre1, d + stride);
    }
}

#undef MHX_SCAN_AND_WALK

// ---- CSR rows ---------------------------------------------------------------------------------------
// A row that stores few of the columns is evaluated entry by entry (weighted_csr_direct_kernel: one wave per row and
// 64 samples, four table entries in flight); a row that stores many is spread out in LDS (-inf where nothing is stored)
// and walked like a dense row (weighted_walk_csr_kernel: one workgroup per row).  The two kernels share the rows out by
// the same test on the row's length.
// `mode` > 0: option weighted.direct, a fixed share of the columns in per mille (A/B runs); 0: by cost.  Measured on an MI355X with this
// round's kernels over densities 0.02 .. 0.5 of (1024 columns, 64 samples), (4096, 128), (1024, 256) -- profiles/r06_sweep_weighted_csr.txt --,
// in microseconds per 20 000 rows: entry by entry 0.70 x stored entries x chunks of 64 samples; walked (39 + 0.011 dim + 20 chunks) for
// clearing and spreading the row + (3 + 13.7 chunks + 0.003 dim) x dim / stored for the positions a walk passes before it meets stored
// columns + 0.017 chunks x stored.  The fixed 10 % of rounds 3-5 sat on the crossover of (4096, 128) only: 1024 columns x 64 samples at
// 10 % ran 0.30 ms where entry by entry now takes 0.08.  Only speed depends on the choice: both kernels produce the reference's (k, t).
__host__ __device__ __forceinline__ bool csr_row_is_walked(int64_t nnz, int32_t dim, int32_t chunks, int32_t mode) {
    if (mode > 0) return nnz * 1000 > (int64_t)mode * dim;
    return 683 * (int64_t)chunks * nnz * nnz > (39000 + 11 * (int64_t)dim + 20000 * (int64_t)chunks) * nnz + (3000 + 13700 * (int64_t)chunks + 3 * (int64_t)dim) * dim;
}

// every stored entry of a CSR row with numpy's argmin (first minimum in storage order; the first NaN wins).  A row whose logs are all
// sane -- no NaN, no finite value beyond 2^80 -- takes t without the division (evaluate_guarded: 16 VALU instructions per element where
// the division costs 27), with the strict "smaller" of a NaN-free row; any other row the general rule (csr_row_general).
//
// Round 6: the loop was a chain of dependent memory round trips -- per four elements one scalar load of (column, log), then the four
// table loads that need its answer, then the wait for those: ~25 latencies in a row for a row of 41 entries, which at eight waves per
// SIMD is what the kernel's 0.30 ms per 80 000 rows was made of (VALU busy 0.66 but 0.61 of the wave cycles waiting).  Now a wave
// takes 64 entries of the row with ONE coalesced vector load (lane u holds entry u; the same load serves the sanity test), hands
// them to the whole wave by v_readlane, and keeps eight table entries in flight.  Groups are filled up with the block's last entry
// (evaluated again: the strict "<" ignores it).
__device__ __forceinline__ void csr_row_general(const int32_t MHX_CONST_AS *indices, const float MHX_CONST_AS *logs, int64_t beg, int64_t end,
                                                const float4 *__restrict__ aos, int32_t s_pad, int32_t my, int64_t &k_out, int64_t &t_out) {
    Best best;
    best.ln_a = 0.0f, best.t = 0.0f, best.k = -1;
    for (int64_t j = beg; j < end; ++j) {
        const int32_t c = indices[j];
        consider(best, logs[j], entry_of(aos[(int64_t)c * s_pad + my]), c);
    }
    k_out = best.k, t_out = (int64_t)best.t;
}

template <int G>
__device__ __forceinline__ void csr_group(int32_t c_v, float l_v, int u0, int last, const float4 *__restrict__ aos_my, int32_t s_pad,
                                          float &best_a, float &best_t, int32_t &best_c) {
    int32_t c[G];
    float4 e[G];
    float l[G], t[G], a[G];
    bool open = false;
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const int at = u0 + u < last ? u0 + u : last;  // (uniform)
        c[u] = __builtin_amdgcn_readlane(c_v, at);
        l[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l_v), at));
        e[u] = aos_my[(int64_t)c[u] * s_pad];
    }
#pragma unroll
    for (int u = 0; u < G; ++u) open |= evaluate_guarded<false>(l[u], e[u], t[u], a[u]);
    if (__builtin_expect(__any(open), 0)) {
#pragma unroll
        for (int u = 0; u < G; ++u) evaluate<false>(l[u], entry_of(e[u]), t[u], a[u]);
    }
#pragma unroll
    for (int u = 0; u < G; ++u)
        if (a[u] < best_a) best_a = a[u], best_t = t[u], best_c = c[u];
}

// WIDE: eight table entries in flight (the entry-by-entry kernel: 70 VGPRs); the walk kernel, where this is the rare way out of a row,
// keeps four (eight there cost it 87 VGPRs and three of its eight workgroups per CU: walked rows 0.14 -> 0.33 ms per 20 000)
template <bool WIDE>
__device__ __forceinline__ void csr_row_by_entry(const int32_t MHX_CONST_AS *indices, const float MHX_CONST_AS *logs,
                                                 const int32_t *__restrict__ indices_vec, const float *__restrict__ logs_vec, int64_t beg, int64_t end,
                                                 const float4 *__restrict__ aos, int32_t s_pad, int32_t my, int64_t &k_out, int64_t &t_out) {
    const int lane = my & (kWave - 1);
    const float4 *aos_my = aos + my;
    float best_a = __builtin_inff(), best_t = 0.0f;
    int32_t best_c = 0;
    for (int64_t base = beg; base < end; base += kWave) {
        const int cnt = (int)(end - base < kWave ? end - base : kWave);  // (uniform)
        const int mine = lane < cnt ? lane : cnt - 1;
        const int32_t c_v = indices_vec[base + mine];
        const float l_v = logs_vec[base + mine];
        const float m = fabsf(l_v);
        if (!__all(m <= 0x1p80f || m == __builtin_inff())) {  // a NaN or a huge log somewhere in the row: all of it by the general rule
            csr_row_general(indices, logs, beg, end, aos, s_pad, my, k_out, t_out);
            return;
        }
        int u0 = 0;
        if constexpr (WIDE)
            for (; cnt - u0 > 4; u0 += 8) csr_group<8>(c_v, l_v, u0, cnt - 1, aos_my, s_pad, best_a, best_t, best_c);
        for (; u0 < cnt; u0 += 4) csr_group<4>(c_v, l_v, u0, cnt - 1, aos_my, s_pad, best_a, best_t, best_c);
    }
    if (__builtin_expect(__any(!(best_a < __builtin_inff())), 0)) {  // nothing but +inf (stored zeros): numpy's argmin is the first entry, with its own t
        csr_row_general(indices, logs, beg, end, aos, s_pad, my, k_out, t_out);
        return;
    }
    k_out = best_c, t_out = (int64_t)best_t;
}

// One wave per (row, 64 samples).  The chunk of 64 samples is blockIdx.x % chunks: workgroups go round the 8 XCDs in
// turn, so when chunks divides 8 an XCD only ever reads its chunks' part of the table -- at 128 samples x 4096 columns
// 4.2 MB of {r, ln_c, beta, 1/r}, of which its 4 MB L2 holds nearly all; the whole table (8.4 MB) would not fit.
__global__ __launch_bounds__(256) void weighted_csr_direct_kernel(const int64_t *__restrict__ indptr_, const int32_t *__restrict__ indices_,
                                                                  const float *__restrict__ logs_, int64_t n_rows, int32_t dim,
                                                                  int32_t direct_permille, const float4 *__restrict__ aos,
                                                                  int32_t sample_size, int32_t s_pad, int64_t *__restrict__ out,
                                                                  uint8_t *__restrict__ nonempty) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const uint32_t chunks = (uint32_t)s_pad / kWave;
    const uint32_t ch = blockIdx.x % chunks;
    const int32_t my = (int32_t)ch * kWave + lane;
    const int64_t MHX_CONST_AS *indptr = (const int64_t MHX_CONST_AS *)indptr_;
    const int32_t MHX_CONST_AS *indices = (const int32_t MHX_CONST_AS *)indices_;
    const float MHX_CONST_AS *logs = (const float MHX_CONST_AS *)logs_;
    for (int64_t row = (int64_t)(blockIdx.x / chunks) * waves_per_block + wave; row < n_rows;
         row += (int64_t)(gridDim.x / chunks) * waves_per_block) {
        const int64_t beg = indptr[row], end = indptr[row + 1];
        if (ch == 0 && lane == 0) nonempty[row] = end > beg ? 1 : 0;
        if (direct_permille >= 0 && csr_row_is_walked(end - beg, dim, (int32_t)chunks, direct_permille)) continue;  // the walk kernel's row
        int64_t k = 0, t = 0;
        if (end > beg) csr_row_by_entry<true>(indices, logs, indices_, logs_, beg, end, aos, s_pad, my, k, t);
        if (my < sample_size) {
            int64_t *o = out + (row * sample_size + my) * 2;
            o[0] = k;
            o[1] = t;
        }
    }
}

// Is any row of the call long enough to be walked?  *gate = 1 if so (the caller zeroes it).  The plan and walk launches read the word
// and return at once when it is 0: on a corpus of short rows they were 15 % of the call.  A pass of its own over the row pointers
// (8 bytes per row, a few microseconds) with one store per workgroup that found one: letting the entry-by-entry kernel's 16 384 waves
// report what they skipped -- an atomic each, or even a plain store each, on one word -- serialised in the memory system for 0.19 ms
// and more (profiles/r06_ab_weighted_csr.txt).
__global__ __launch_bounds__(256) void csr_any_walked_kernel(const int64_t *__restrict__ indptr, int64_t n_rows, int32_t dim, int32_t chunks,
                                                             int32_t mode, unsigned int *__restrict__ gate) {
    bool any = false;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += (int64_t)gridDim.x * blockDim.x)
        any |= csr_row_is_walked(indptr[row + 1] - indptr[row], dim, chunks, mode);
    if (__syncthreads_or(any) && threadIdx.x == 0) *gate = 1u;
}

__global__ __launch_bounds__(256, 8) void weighted_walk_csr_kernel(const int64_t *__restrict__ indptr_, const int32_t *__restrict__ indices_,
                                                                const float *__restrict__ logs_, int64_t n_rows, int32_t dim,
                                                                int32_t direct_permille, const WalkPlan *__restrict__ plan,
                                                                const float4 *__restrict__ walk_a, const uint32_t *__restrict__ walk_c,
                                                                const float4 *__restrict__ aos, int32_t sample_size, int32_t s_pad,
                                                                int32_t list_cap, int64_t *__restrict__ out, const unsigned int *__restrict__ n_walked) {
    extern __shared__ float row[];                                             // dim logs of the row (-inf: not stored)
    if (n_walked && *n_walked == 0) return;  // (uniform) the entry-by-entry launch kept every row
    uint16_t *list = reinterpret_cast<uint16_t *>(row + ((dim + 3) & ~3));    // columns above the cut (list_cap of them)
    __shared__ int s_nout, s_odd;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
    const int32_t chunks = s_pad / kWave;
    const float lcut = plan->lcut;
    const int32_t n_cc = chunks < kCachedChunks ? chunks : kCachedChunks;
    float4 *s_cache_a = reinterpret_cast<float4 *>(row + ((dim + 3) & ~3) + ((list_cap + 7) & ~7) / 2);
    uint32_t *s_cache_c = reinterpret_cast<uint32_t *>(s_cache_a + n_cc * kWalkCached * kWave);
    for (int j = tid; j < n_cc * kWalkCached * kWave; j += blockDim.x) {
        const int ch = j / (kWalkCached * kWave), k = j / kWave % kWalkCached;
        if (ch < chunks && k < dim) {
            s_cache_a[j] = walk_a[((int64_t)ch * dim + k) * kWave + (j & (kWave - 1))];
            s_cache_c[j] = walk_c[((int64_t)ch * dim + k) * kWave + (j & (kWave - 1))];
        }
    }
    const int64_t MHX_CONST_AS *indptr = (const int64_t MHX_CONST_AS *)indptr_;
    for (int64_t d = blockIdx.x; d < n_rows; d += gridDim.x) {
        const int64_t beg = indptr[d], end = indptr[d + 1];
        if (!csr_row_is_walked(end - beg, dim, chunks, direct_permille)) continue;  // the direct kernel's row
        if (tid == 0) s_nout = 0, s_odd = 0;
        for (int c = tid; c < dim; c += blockDim.x) row[c] = -__builtin_inff();
        __syncthreads();
        // scatter; "odd": a NaN, a column stored twice, a column outside the matrix -- such a row is evaluated entry by entry
        bool odd = false;
        for (int64_t j = beg + tid; j < end; j += blockDim.x) {
            const int32_t c = indices_[j];
            const float l = logs_[j];
            if (l != l || c < 0 || c >= dim) {
                odd = true;
                continue;
            }
            if (l == -__builtin_inff()) continue;  // a stored zero: +inf for ln_a, never the argmin of a row with anything else
            const float was = __uint_as_float(atomicExch(reinterpret_cast<unsigned int *>(&row[c]), __float_as_uint(l)));
            odd |= !(was == -__builtin_inff());
            if (l > lcut) {
                const int at = atomicAdd(&s_nout, 1);
                if (at < list_cap) list[at] = (uint16_t)c;
            }
        }
        if (__any(odd) && lane == 0) s_odd = 1;
        __syncthreads();
        const int n_out = s_nout;
        // nothing but stored zeros would leave the walk without an answer: entry by entry as well (k = the first column)
        const bool by_entry = s_odd != 0 || n_out > list_cap;
        for (int32_t ch = wave; ch < chunks; ch += n_waves) {
            const int32_t my = ch * kWave + lane;
            int64_t k_out = 0, t_out = 0;
            if (by_entry) {
                csr_row_by_entry<false>((const int32_t MHX_CONST_AS *)indices_, (const float MHX_CONST_AS *)logs_, indices_, logs_, beg, end, aos, s_pad, my, k_out, t_out);
            } else {
                const Held held = walk_row(row, list, n_out, false, dim, ch, my, sample_size, walk_a, walk_c, aos, s_pad,
                                           ch < n_cc ? s_cache_a + ch * kWalkCached * kWave : nullptr,
                                           ch < n_cc ? s_cache_c + ch * kWalkCached * kWave : nullptr, 0, 1);
                k_out = held.c, t_out = (int64_t)held.t;
                const bool nothing = my < sample_size && k_out == 0xFFFFFFFFll;  // the walk met nothing (stored zeros only)
                if (__any(nothing)) {  // (the whole wave goes: csr_row_by_entry hands entries around with v_readlane)
                    int64_t k2 = 0, t2 = 0;
                    csr_row_by_entry<false>((const int32_t MHX_CONST_AS *)indices_, (const float MHX_CONST_AS *)logs_, indices_, logs_, beg, end, aos, s_pad, my, k2, t2);
                    if (nothing) k_out = k2, t_out = t2;
                }
            }
            if (my < sample_size) {
                int64_t *o = out + (d * sample_size + my) * 2;
                o[0] = k_out;
                o[1] = t_out;
            }
        }
        __syncthreads();
    }
}

}  // namespace

int launch_wgen_transpose(mhx_wgen *gen, const float *d_rs, const float *d_lncs, const float *d_betas) {
    mhx_ctx *ctx = gen->ctx;
    const int64_t total = (int64_t)gen->dim * gen->s_pad;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)ctx->num_cus * 8));
    hipLaunchKernelGGL(wgen_transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_rs, d_lncs,
                       d_betas, gen->sample_size, gen->dim, gen->s_pad, gen->d_params, reinterpret_cast<float4 *>(gen->d_aos));
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

// dense rows through the bound-ordered walk: plan (the cut for this call's data), tables (only when the cut moved), rows
static int launch_weighted_dense_walk(mhx_wgen *gen, const float *d_x, int values_are_logs, int64_t n_rows, int64_t *d_out,
                                      uint8_t *d_nonempty) {
    mhx_ctx *ctx = gen->ctx;
    const int32_t dim = gen->dim;
    const int64_t total = n_rows * (int64_t)dim;
    const int32_t n_seg = (int32_t)std::min<int64_t>(n_rows, std::max<int64_t>(1, (16 << 10) / dim));  // about 16k sampled logs
    WalkPlan *plan = nullptr;
    if (int rc = launch_walk_plan(gen, d_x, values_are_logs != 0, total, dim, n_seg, (float)dim, &plan)) return rc;
    float4 *walk_a = reinterpret_cast<float4 *>(gen->d_walk_a);
    const int32_t list_cap = std::max(64, dim / 4);
    const int32_t direct_permille_w = ctx->opt_weighted_direct > 0 ? (int32_t)ctx->opt_weighted_direct : 100;
    // one wave per row (weighted_walk_wave_kernel) when a row is 4 .. 16 sixteen-byte loads per lane and at least four
    // stripes fit the LDS beside the cached tables; option weighted.kernel: 1 = the workgroup-per-row kernel always
    {
        const int32_t n_cc_w = std::min<int32_t>(gen->s_pad / kWave, kCachedChunks);
        const size_t cache_bytes = 20 * (size_t)n_cc_w * kWalkCached * kWave;
        const int32_t list_cap_w = std::max(64, dim / 8);  // (half the other kernel's: eight stripes of a 4096-column row then fit beside the tables)
        const size_t stripe_bytes = (sizeof(float) * (size_t)((dim + 3) & ~3) + sizeof(uint16_t) * (size_t)((list_cap_w + 7) & ~7) + 15) & ~(size_t)15;
        const int64_t fit = ((int64_t)ctx->lds_per_block - (int64_t)cache_bytes - 64) / (int64_t)stripe_bytes;
        const int waves = (int)std::min<int64_t>(8, fit);
        const int32_t min_dim_w = ctx->opt_weighted_min_dim > 0 ? (int32_t)ctx->opt_weighted_min_dim : 4;  // (until round 6: 1024 -- rows of 64 .. 1000 columns x 64 .. 256 samples run 1.5 - 1.75 x faster here than one workgroup per row, profiles/r06_ab_weighted_small_dims.txt)
        const bool shape_ok = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 && dim >= min_dim_w && dim <= 4096;
        if (shape_ok && waves >= 4 && ctx->opt_weighted_kernel != 1) {  // (weighted.debug 1 / 2: this kernel's phases alone, profiling)
            const size_t lds = cache_bytes + stripe_bytes * (size_t)waves;
            const int64_t groups = (n_rows + waves - 1) / waves;
            const int64_t per_cu = ctx->opt_blocks_per_cu > 0 ? ctx->opt_blocks_per_cu : std::max<int64_t>(1, (int64_t)ctx->lds_per_block / (int64_t)(lds + 64));
            const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(groups, per_cu * ctx->num_cus));
            const int nv = dim <= 1024 ? 4 : dim <= 2048 ? 8 : 16;
#define MHX_WALK_WAVE(LOGS, NV_, ...)                                                                                                  \
    hipLaunchKernelGGL((weighted_walk_wave_kernel<LOGS, NV_, __VA_ARGS__>), dim3(split == 2 ? blocks2 : blocks), dim3(split == 2 ? 1024 : 64 * waves), split == 2 ? lds2 : lds, ctx->stream, d_x, n_rows, dim, plan, walk_a, \
                       gen->d_walk_c, reinterpret_cast<const float4 *>(gen->d_aos), gen->sample_size, gen->s_pad, list_cap_w, direct_permille_w,  \
                       (int32_t)(stripe_bytes / 4), rescue_lanes, d_out, d_nonempty, (int32_t)ctx->opt_weighted_debug, split_stripes)
#define MHX_WALK_WAVE_NV(LOGS, PAIRS_)            \
    do {                                          \
        if (split == 2 && nv == 4) MHX_WALK_WAVE(LOGS, 4, false, 2, 2);       \
        else if (split == 2 && nv == 8) MHX_WALK_WAVE(LOGS, 8, false, 2, 2);  \
        else if (split == 2) MHX_WALK_WAVE(LOGS, 16, false, 2, 2);            \
        else if (nv == 4) MHX_WALK_WAVE(LOGS, 4, PAIRS_);      \
        else if (nv == 8) MHX_WALK_WAVE(LOGS, 8, PAIRS_); \
        else if (fetch_mode == 2) MHX_WALK_WAVE(LOGS, 16, PAIRS_, 2); \
        else if (fetch_mode == 3) MHX_WALK_WAVE(LOGS, 16, PAIRS_, 3); \
        else MHX_WALK_WAVE(LOGS, 16, PAIRS_);             \
    } while (0)
            // 4096-column rows (NV = 16), measured on config 4 (profiles/r05_ab_weighted_refill.txt): non-temporal row loads 0.426 -> 0.408 ms with
            // logs in; with values in, chunk after chunk + the refill right after staging + non-temporal 0.526 -> 0.499 ms.  The early refill
            // alone gains nothing (0.427) and costs the chunk-pair walk 17-32 spilled VGPRs (0.489).  Option weighted.refill: 0 auto,
            // 1 = plain loads behind the walk (round 4), 2 / 3 = force that mode.
            // weighted.refill 0 (auto): 4096-column rows and 128 samples go to the fetcher / walker split (SPLIT 2) -- four fetchers, six pairs of
            // walkers (one walker per chunk of samples), SEVEN stripes (one holds a row that waits) and 12 cached list positions per chunk.
            // A/B: 9 = six stripes + 16 positions, 5 = six + 8, 8 = eight + 8, 6 = five stripes, six fetchers, 16 positions; 13 = auto without
            // the split (the one-wave-per-row kernel of this round's first half).  profiles/r05_ab_weighted_split.txt
            const int64_t rf = ctx->opt_weighted_refill;
            // sample counts: a row's chunks of 64 samples go to as many walkers -- 2, 3, 4 or 6 chunks (65 .. 256 and 321 .. 384 samples) share out twelve
            // walkers as 6, 4, 3 or 2 rows at a time, with one stripe more than rows being walked
            const int32_t chunks_w = gen->s_pad / kWave;
            const bool chunks_ok = chunks_w == 2 || chunks_w == 3 || chunks_w == 4 || chunks_w == 6;
            const int32_t groups2 = chunks_ok ? 12 / chunks_w : 1;
            const int32_t n_stripes2 = chunks_w != 2 ? groups2 + 1 : rf == 6 ? 5 : rf == 8 ? 8 : rf == 5 || rf == 9 ? 6 : 7;
            const int32_t cached2 = rf == 5 || rf == 8 ? kWalkCached : rf == 6 || rf == 9 ? 16 : 12;
            const int32_t n_fetch2 = rf == 6 && chunks_w == 2 ? 6 : 4;
            const int32_t split_stripes = n_stripes2 | cached2 << 8 | n_fetch2 << 16;
            const size_t lds2 = 20 * (size_t)n_cc_w * cached2 * kWave + 4 * kSplitHandWords2 + stripe_bytes * (size_t)n_stripes2;
            const int split = (rf == 0 || (rf >= 5 && rf <= 9)) && ctx->opt_weighted_kernel == 0 && chunks_ok && lds2 <= (size_t)ctx->lds_per_block ? 2 : 0;
            const unsigned blocks2 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_rows, ctx->num_cus));  // (SPLIT 2: one workgroup of sixteen waves per CU, rows blockIdx.x + i * gridDim.x)
            const bool auto_fetch = rf == 0 || rf == 13;
            const int fetch_mode = auto_fetch ? (values_are_logs ? 2 : 3) : ctx->opt_weighted_refill == 1 ? 0 : (int)(ctx->opt_weighted_refill & 3);
            const int32_t rescue_lanes = ctx->opt_weighted_rescue < 0 ? 0 : ctx->opt_weighted_rescue > 0 ? (int32_t)ctx->opt_weighted_rescue : 8;  // (lognormal rows at steady clocks: 2: 0.557, 4: 0.535, 8: 0.529, 16: 0.563, 32: 0.68 ms per 20k; config 4 the same for all)
            // two chunks of samples as one stream (0.424 -> 0.405 ms on config 4 with logs in); 2 = chunk after chunk, which values in take at NV = 16
            const bool pairs = ctx->opt_weighted_kernel == 0 ? !(nv == 16 && !values_are_logs && auto_fetch) : ctx->opt_weighted_kernel != 2;
            if (values_are_logs) {
                if (pairs) MHX_WALK_WAVE_NV(true, true);
                else MHX_WALK_WAVE_NV(true, false);
            } else {
                if (pairs) MHX_WALK_WAVE_NV(false, true);
                else MHX_WALK_WAVE_NV(false, false);
            }
#undef MHX_WALK_WAVE_NV
#undef MHX_WALK_WAVE
            MHX_HIP_CHECK(hipGetLastError());
            return MHX_OK;
        }
    }
    const unsigned threads = 256;  // four waves stage a row; its chunks of 64 samples are then shared out among them
    const int32_t direct_permille = ctx->opt_weighted_direct > 0 ? (int32_t)ctx->opt_weighted_direct : 100;  // rows storing less than 10 % of the columns: entry by entry (measured crossover, dense and CSR alike)
    const int32_t n_cc = std::min<int32_t>(gen->s_pad / kWave, kCachedChunks);
    const size_t lds = sizeof(float) * (size_t)((dim + 3) & ~3) + sizeof(uint16_t) * (size_t)((list_cap + 7) & ~7) + 20 * (size_t)n_cc * kWalkCached * kWave;
    const int64_t per_cu = ctx->opt_blocks_per_cu > 0 ? ctx->opt_blocks_per_cu
                                                      : std::max<int64_t>(1, std::min<int64_t>(4, (int64_t)((160 << 10) / (lds + 64))));
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_rows, per_cu * ctx->num_cus));
    const bool ahead = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 && dim >= 4 && dim <= kPre * 4 * (int)threads;
#define MHX_WALK_DENSE(LOGS, AHEAD)                                                                                                    \
    hipLaunchKernelGGL((weighted_walk_dense_kernel<LOGS, AHEAD>), dim3(blocks), dim3(threads), lds, ctx->stream, d_x, n_rows, dim, plan, walk_a, \
                       gen->d_walk_c, reinterpret_cast<const float4 *>(gen->d_aos), gen->sample_size, gen->s_pad, list_cap, direct_permille, \
                       d_out, d_nonempty, (int32_t)ctx->opt_weighted_debug, (int32_t)(ctx->opt_weighted_split != 1))
    if (values_are_logs) {
        if (ahead) MHX_WALK_DENSE(true, true);
        else MHX_WALK_DENSE(true, false);
    } else {
        if (ahead) MHX_WALK_DENSE(false, true);
        else MHX_WALK_DENSE(false, false);
    }
#undef MHX_WALK_DENSE
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_weighted_dense(mhx_wgen *gen, const float *d_x, int values_are_logs, int64_t n_rows, int64_t *d_out,
                          uint8_t *d_nonempty) {
    mhx_ctx *ctx = gen->ctx;
    const int32_t dim = gen->dim;
    if (gen->walk_ok && ctx->opt_weighted_path == 0) return launch_weighted_dense_walk(gen, d_x, values_are_logs, n_rows, d_out, d_nonempty);
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
    {
        const float v = x[i];
        out[i] = np_logf_is_normal(v) ? np_logf_normal(v) : np_logf(v);  // (both paths are what the row kernels take: the 2^32-pattern test runs through here)
    }
}

int launch_weighted_log(mhx_ctx *ctx, const float *d_x, int64_t n, float *d_out) {
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(weighted_log_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 16))),
                       dim3(256), 0, ctx->stream, d_x, n, d_out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

// CSR rows when the generator has walk tables: sparse rows entry by entry, the others through the walk
static int launch_weighted_csr_walk(mhx_wgen *gen, const int64_t *d_indptr, const int32_t *d_indices, const float *d_values,
                                    int values_are_logs, int64_t n_rows, int64_t nnz, int64_t *d_out, uint8_t *d_nonempty) {
    mhx_ctx *ctx = gen->ctx;
    const int32_t dim = gen->dim;
    const float *d_logs = d_values;
    if (!values_are_logs) {  // device-log mode: the logs once, into scratch slot 3
        if (int rc = ctx->ensure_scratch(3, sizeof(float) * (size_t)std::max<int64_t>(nnz, 1) + 256)) return rc;
        if (nnz > 0)
            if (int rc = launch_weighted_log(ctx, d_values, nnz, (float *)ctx->scratch[3])) return rc;
        d_logs = (const float *)ctx->scratch[3];
    }
    const int32_t direct_mode = ctx->opt_weighted_direct > 0 ? (int32_t)ctx->opt_weighted_direct : 0;  // 0: by cost (csr_row_is_walked)
    WalkPlan *plan = reinterpret_cast<WalkPlan *>(gen->d_walk_plan) + gen->plan_index;
    float4 *walk_a = reinterpret_cast<float4 *>(gen->d_walk_a);
    const int64_t chunks = gen->s_pad / kWave;
    const bool any_walk = csr_row_is_walked(std::min<int64_t>(nnz, dim), dim, (int32_t)chunks, direct_mode);  // the longest row there can be
    // A short pass over the row pointers says whether any row is walked at all (d_work word 12); the plan and walk launches read the
    // word and return at once when it is 0.  The two-launch plan (dim > 8192) has no gate: it always runs.
    unsigned int *d_gate = nullptr;
    if (any_walk) {
        if (int rc = ctx->ensure_work()) return rc;
        d_gate = ctx->d_work + 12;
        MHX_HIP_CHECK(hipMemsetAsync(d_gate, 0, sizeof(unsigned int), ctx->stream));
        hipLaunchKernelGGL(csr_any_walked_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_rows + 1023) / 1024, 256))), dim3(256), 0, ctx->stream,
                           d_indptr, n_rows, dim, (int32_t)chunks, direct_mode, d_gate);
        MHX_HIP_CHECK(hipGetLastError());
    }
    const int64_t want = (n_rows + 3) / 4;
    const int64_t groups = std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * (ctx->opt_blocks_per_cu > 0 ? ctx->opt_blocks_per_cu : 16) / chunks));
    hipLaunchKernelGGL(weighted_csr_direct_kernel, dim3((unsigned)(groups * chunks)), dim3(256), 0,
                       ctx->stream, d_indptr, d_indices, d_logs, n_rows, dim, any_walk ? direct_mode : -1,
                       reinterpret_cast<const float4 *>(gen->d_aos), gen->sample_size, gen->s_pad, d_out, d_nonempty);
    MHX_HIP_CHECK(hipGetLastError());
    if (any_walk) {
        const int32_t seg = (int32_t)std::min<int64_t>(nnz, 1024);
        if (int rc = launch_walk_plan(gen, d_logs, true, nnz, seg, (int32_t)std::min<int64_t>(16, nnz / seg),
                                      (float)std::min<double>((double)dim, (double)nnz / (double)std::max<int64_t>(n_rows, 1)), &plan, d_gate))
            return rc;
        const int32_t list_cap = std::max(64, dim / 4);
        const int32_t n_cc = std::min<int32_t>(gen->s_pad / kWave, kCachedChunks);
        const size_t lds = sizeof(float) * (size_t)((dim + 3) & ~3) + sizeof(uint16_t) * (size_t)((list_cap + 7) & ~7) + 20 * (size_t)n_cc * kWalkCached * kWave;
        const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)((160 << 10) / (lds + 64))));
        const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_rows, per_cu * ctx->num_cus));
        hipLaunchKernelGGL(weighted_walk_csr_kernel, dim3(blocks), dim3(256), lds, ctx->stream, d_indptr, d_indices, d_logs, n_rows, dim, direct_mode,
                           plan, walk_a, gen->d_walk_c, reinterpret_cast<const float4 *>(gen->d_aos), gen->sample_size, gen->s_pad, list_cap, d_out, d_gate);
        MHX_HIP_CHECK(hipGetLastError());
    }
    return MHX_OK;
}

int launch_weighted(mhx_wgen *gen, const int64_t *d_indptr, const int32_t *d_indices, const float *d_values,
                    int values_are_logs, int64_t n_rows, int64_t nnz, int64_t *d_out, uint8_t *d_nonempty) {
    mhx_ctx *ctx = gen->ctx;
    if (gen->walk_ok && ctx->opt_weighted_path == 0)
        return launch_weighted_csr_walk(gen, d_indptr, d_indices, d_values, values_are_logs, n_rows, nnz, d_out, d_nonempty);
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
