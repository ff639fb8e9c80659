// tools/ubench_mfma.hip -- could the f64 matrix pipe take part of the sieve's multiplies?
//
// The sieve key of a (token, permutation) pair is M = lo32(h*a_lo + b8).  With h = h0 + 2^16*h1 (16-bit halves):
//     M = lo32(h0*a_lo + h1*a' + b8),  a' = (a_lo mod 2^16) * 2^16,
// every product < 2^48 and the sum < 2^50: exact in a double.  So D = A x B + C with rows of A = (h0, h1, 0, 0),
// columns of B = (a_lo, a', 0, 0), C = b8 + 2^52 is ONE v_mfma_f64_16x16x4_f64 for 16 tokens x 16 permutations, and the
// low dword of every D element IS the key (the 2^52 bias puts the integer into the mantissa's low bits).
// This probe measures what the matrix pipe would deliver beside the VALU stream on gfx950:
//   valu   the sieve's own mix: 8 v_mad_u64_u32 + 4 v_min3_u32 per step (16 keys per lane)
//   mfma   v_mfma_f64_16x16x4_f64 back to back, 4 independent accumulators + 2 v_min3 per MFMA on the low dwords
//          of its results (what the sieve would do with the keys)
//   both   one wave issues both streams interleaved (R MFMAs per VALU step), keys/s of the sum
// and checks the identity on random inputs.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef unsigned u32;

__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { return min(min(a, b), c); }
__device__ __forceinline__ u32 lo32(double d) { return (u32)__double_as_longlong(d); }

// ---- identity check: 16 tokens x 16 permutations per wave ------------------------------------------------
__global__ void check_kernel(const u32 *h, const u32 *a_lo, const u32 *b8, u32 *keys) {
    const int lane = threadIdx.x;
    const int idx = lane & 15, kk = lane >> 4;  // A: row idx, k = kk; B: column idx, k = kk
    const u32 hv = h[idx], av = a_lo[idx];
    const double A = kk == 0 ? (double)(hv & 0xFFFFu) : kk == 1 ? (double)(hv >> 16) : 0.0;
    const double B = kk == 0 ? (double)av : kk == 1 ? (double)(av & 0xFFFFu) * 65536.0 : 0.0;
    const double bias = 4503599627370496.0 + (double)b8[idx];  // 2^52 + b8 of the lane's column
    d4 c = {bias, bias, bias, bias};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, c, 0, 0, 0);
    // D layout of the f64 form: register r of lane l holds row (l >> 4) + 4*r, column l & 15
#pragma unroll
    for (int r = 0; r < 4; ++r) keys[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = lo32(c[r]);
}

// ---- rates -----------------------------------------------------------------------------------------------
template <int MODE, int R>  // MODE 0 valu, 1 mfma, 2 both (R MFMAs per VALU step)
__global__ __launch_bounds__(256) void rate_kernel(const u32 *tok, int steps, u32 *sink, u32 a_seed) {
    const int lane = threadIdx.x & 63;
    u32 a0 = a_seed * (2 * lane + 1), a1 = a0 * 2654435761u + 12345u;
    u64 b0 = ((u64)a1 << 32) | a0, b1 = b0 * 3 + 7;
    u32 acc0 = 0xFFFFFFFFu, acc1 = 0xFFFFFFFFu, accm = 0xFFFFFFFFu;
    const double A0 = (double)(a0 & 0xFFFF), B0 = (double)(a1 >> 8);
    double Av = A0, Bv = B0;
    const double bias = 4503599627370496.0 + (double)(a0 >> 3);
    d4 c0 = {bias, bias, bias, bias}, c1 = c0, c2 = c0, c3 = c0;
    const u32 __attribute__((address_space(4))) *t = (const u32 __attribute__((address_space(4))) *)tok;
    for (int s = 0; s < steps; ++s) {
        if constexpr (MODE == 0 || MODE == 2) {
            u32 h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = t[(s * 8 + j) & 1023];
            u32 m[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u64 r0 = (u64)h[2 * j] * a0 + b0, r1 = (u64)h[2 * j + 1] * a0 + b0;
                asm volatile("" : "+v"(r0), "+v"(r1));
                m[2 * j] = (u32)r0;
                m[2 * j + 1] = (u32)r1;
            }
            acc0 = umin3(acc0, m[0], m[1]);
            acc0 = umin3(acc0, m[2], m[3]);
            acc0 = umin3(acc0, m[4], m[5]);
            acc0 = umin3(acc0, m[6], m[7]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u64 r0 = (u64)h[2 * j] * a1 + b1, r1 = (u64)h[2 * j + 1] * a1 + b1;
                asm volatile("" : "+v"(r0), "+v"(r1));
                m[2 * j] = (u32)r0;
                m[2 * j + 1] = (u32)r1;
            }
            acc1 = umin3(acc1, m[0], m[1]);
            acc1 = umin3(acc1, m[2], m[3]);
            acc1 = umin3(acc1, m[4], m[5]);
            acc1 = umin3(acc1, m[6], m[7]);
        }
        if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // fresh accumulators every time (C = bias): the previous result is consumed by the two min3 below
                d4 &c = (r & 3) == 0 ? c0 : (r & 3) == 1 ? c1 : (r & 3) == 2 ? c2 : c3;
                accm = umin3(accm, lo32(c[0]), lo32(c[1]));
                accm = umin3(accm, lo32(c[2]), lo32(c[3]));
                d4 z = {bias, bias, bias, bias};
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(Av, Bv, z, 0, 0, 0);
                asm volatile("" : "+v"(Av), "+v"(Bv));  // keep the optimiser from merging / hoisting equal MFMAs
            }
        }
    }
    accm = umin3(accm, lo32(c0[0]), lo32(c1[1]));
    accm = umin3(accm, lo32(c2[2]), lo32(c3[3]));
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc0 ^ acc1 ^ accm;
}

template <int MODE, int R>
static void run(const char *name, const u32 *d_tok, u32 *d_sink, int cus, int blocks_per_cu) {
    const int steps = 4096, blocks = cus * blocks_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<MODE, R>), dim3(blocks), dim3(256), 0, 0, d_tok, steps, d_sink, 77u + rep);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double waves = (double)blocks * 4;
    const double valu_keys = (MODE == 0 || MODE == 2) ? waves * steps * 16.0 * 64 : 0;  // 16 keys per lane per step
    const double mfma_keys = (MODE == 1 || MODE == 2) ? waves * steps * (double)R * 256 : 0;
    const double simd_ns = best * 1e6 / (waves / (cus * 4.0)) / steps;  // ns per step per SIMD (one wave's step, waves of a SIMD in sequence)
    printf("{\"probe\": \"%s\", \"blocks_per_cu\": %d, \"mfma_per_step\": %d, \"ms\": %.4f, \"ns_per_step_per_simd\": %.3f, \"valu_keys_per_s\": %.4g, "
           "\"mfma_keys_per_s\": %.4g, \"keys_per_s\": %.4g}\n",
           name, blocks_per_cu, (MODE == 0 ? 0 : R), best, simd_ns, valu_keys / best * 1e3, mfma_keys / best * 1e3, (valu_keys + mfma_keys) / best * 1e3);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // identity
    std::vector<u32> h(16), a(16), b(16), keys(256);
    srand(7);
    auto r32 = []() { return ((u32)rand() << 16) ^ (u32)rand() ^ ((u32)rand() << 31); };
    int bad = 0, total = 0;
    u32 *d_h, *d_a, *d_b, *d_k;
    CHK(hipMalloc(&d_h, 64)); CHK(hipMalloc(&d_a, 64)); CHK(hipMalloc(&d_b, 64)); CHK(hipMalloc(&d_k, 1024));
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 16; ++i) { h[i] = r32(); a[i] = r32(); b[i] = r32(); }
        if (trial == 0) { h[0] = 0xFFFFFFFFu; a[0] = 0xFFFFFFFFu; b[0] = 0xFFFFFFFFu; h[1] = 0; a[1] = 0; b[1] = 0; }
        CHK(hipMemcpy(d_h, h.data(), 64, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d_a, a.data(), 64, hipMemcpyHostToDevice));
        CHK(hipMemcpy(d_b, b.data(), 64, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, d_h, d_a, d_b, d_k);
        CHK(hipMemcpy(keys.data(), d_k, 1024, hipMemcpyDeviceToHost));
        for (int t = 0; t < 16; ++t)
            for (int p = 0; p < 16; ++p) {
                const u32 want = (u32)((u64)h[t] * a[p] + b[p]);
                bad += keys[t * 16 + p] != want;
                ++total;
            }
    }
    printf("{\"probe\": \"identity lo32(h*a_lo + b8) == low dword of v_mfma_f64_16x16x4_f64\", \"pairs\": %d, \"mismatches\": %d}\n", total, bad);
    u32 *d_tok, *d_sink;
    CHK(hipMalloc(&d_tok, 4096));
    std::vector<u32> tok(1024);
    for (auto &x : tok) x = r32();
    CHK(hipMemcpy(d_tok, tok.data(), 4096, hipMemcpyHostToDevice));
    CHK(hipMalloc(&d_sink, (size_t)cus * 8 * 256 * 4));
    for (int bpc : {1, 2, 8}) {  // 1, 2, 8 waves per SIMD
        run<0, 0>("valu", d_tok, d_sink, cus, bpc);
        run<1, 1>("mfma", d_tok, d_sink, cus, bpc);
        run<1, 4>("mfma", d_tok, d_sink, cus, bpc);
        run<2, 1>("both", d_tok, d_sink, cus, bpc);
        run<2, 2>("both", d_tok, d_sink, cus, bpc);
        run<2, 4>("both", d_tok, d_sink, cus, bpc);
    }
    return 0;
}
