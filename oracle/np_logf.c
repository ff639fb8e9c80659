/*
 * oracle/np_logf.c -- CPU restatement of numpy's float32 natural logarithm (the SIMD loop `np.log` runs on float32
 * arrays on x86-64 hosts with AVX2+FMA3 or AVX512F).
 *
 * TEST INFRASTRUCTURE ONLY (see the header of mh_oracle.c): the product's copy of this algorithm is the device
 * function np_logf() in datasketch_amd/csrc/weighted_kernels.hip; this file is what tests compare it -- and numpy --
 * with on the CPU.
 *
 * Why it exists.  WeightedMinHashGenerator.minhash_many takes np.log of the float32 data
 * (ref: datasketch/weighted_minhash.py:212) and everything after it is a deterministic function of that log, so a
 * device path that is to reproduce the reference bit for bit WITHOUT a host pass over the matrix has to reproduce
 * numpy's log.  That log lives in a third-party dependency that is not in /root/reference: numpy (the reference pins
 * 2.0.2 / 2.2.6 / 2.3.4 per Python version in uv.lock; 2.2.6 is what this container has), file
 * numpy/_core/src/umath/loops_exponent_log.dispatch.c.src, function simd_log_FLOAT (AVX2 and AVX512F instantiations of
 * one template, unchanged since numpy 1.17).  Its published algorithm, restated:
 *   1. x = m * 2^e with 0.5 <= m < 1 (AVX512F: getexp + 1 / getmant; AVX2: denormals pre-scaled by 2^100 -- both exact);
 *   2. if m <= 1/sqrt(2):  m = m + m, e = e - 1;      m = m - 1                (now -0.2929 <= m < 0.4143)
 *   3. log(1 + m) ~ P(m) / Q(m), two degree-5 polynomials evaluated by Horner's rule with FMA, one IEEE division;
 *   4. result = fma(e, ln 2, P/Q);
 *   5. lanes with x < 0 (and -inf) give -NaN (0xffc00000), +-0 gives -inf, +inf gives +inf, a NaN gives the quiet NaN
 *      0x7fc00000.
 * It is not correctly rounded (numpy states 3.83 ulp), which is exactly why libm's or the GPU's own logf cannot stand
 * in for it.
 *
 * Parity status: PINNED against the dependency itself -- tests/test_np_logf_model.py compares this file with np.log of
 * the installed numpy on a stride through all 2^31 non-negative float32 bit patterns plus every pattern around the
 * algorithm's boundaries, and oracle/check_np_logf.py does all 2^32 patterns (0 mismatches on this container's numpy
 * 2.2.6, AVX512F dispatch; recorded in profiles/r04_np_logf_exhaustive.txt).  On a host whose numpy dispatches to another
 * implementation (no AVX2: libm's logf) the test skips and the product's start-up self-check keeps the log on the host.
 *
 * Build: part of oracle/libmh_oracle.so (oracle/Makefile).  -ffp-contract=off; fmaf() must be a correctly rounded fused
 * multiply-add (glibc's is, in hardware with -mfma or in software without).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

static inline float bits_to_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t float_to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* the constants of numpy's npy_simd_data.h / loops_exponent_log.dispatch.c.src (NPY_COEFF_{P,Q}n_LOGf, NPY_LOGE2f, NPY_SQRT1_2f) */
static const float LOG_P[6] = {0.000000000000000000000e+00f, 9.999999999999998702752e-01f, 2.112677543073053063722e+00f,
                               1.480000633576506585156e+00f, 3.808837741388407920751e-01f, 2.589979117907922693523e-02f};
static const float LOG_Q[6] = {1.000000000000000000000e+00f, 2.612677543073109236779e+00f, 2.453006071784736363091e+00f,
                               9.864942958519418960339e-01f, 1.546476374983906719538e-01f, 5.875095403124574342950e-03f};
static const float LOG_E2 = 0.693147180559945309417232121458176568f;
static const float SQRT1_2 = 0.707106781186547524400844362104849039f;

static float np_logf_one(float x) {
    const uint32_t b = float_to_bits(x);
    if (x != x) return bits_to_float(0x7fc00000u);          /* step 5 */
    if (b == 0u || b == 0x80000000u) return -INFINITY;
    if (b >> 31) return bits_to_float(0xffc00000u);
    if (b == 0x7f800000u) return INFINITY;
    uint32_t mb = b;
    int e;
    if ((b >> 23) == 0) { /* denormal: normalise the mantissa (what getexp / getmant, or the 2^100 pre-scaling, amount to) */
        int s = 0;
        while (!((mb << s) & 0x00800000u)) ++s;
        mb <<= s;
        e = -125 - s;
    } else {
        e = (int)(b >> 23) - 126;
    }
    float m = bits_to_float((mb & 0x007fffffu) | 0x3f000000u); /* step 1 */
    float ef = (float)e;
    if (m <= SQRT1_2) {                                          /* step 2 */
        m = m + m;
        ef = ef - 1.0f;
    }
    m = m - 1.0f;
    float num = fmaf(LOG_P[5], m, LOG_P[4]);                     /* step 3 */
    num = fmaf(num, m, LOG_P[3]);
    num = fmaf(num, m, LOG_P[2]);
    num = fmaf(num, m, LOG_P[1]);
    num = fmaf(num, m, LOG_P[0]);
    float den = fmaf(LOG_Q[5], m, LOG_Q[4]);
    den = fmaf(den, m, LOG_Q[3]);
    den = fmaf(den, m, LOG_Q[2]);
    den = fmaf(den, m, LOG_Q[1]);
    den = fmaf(den, m, LOG_Q[0]);
    const float poly = num / den;
    return fmaf(ef, LOG_E2, poly);                               /* step 4 */
}

ORACLE_API void oracle_np_logf(const float *x, int64_t n, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = np_logf_one(x[i]);
}
