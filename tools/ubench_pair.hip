// tools/ubench_pair.hip -- throughput of candidate instruction sequences for ONE (token, permutation)
// pair evaluation of the MinHash fast path, in the real mix (slow VOP3 + fast VOP2 ops, token in SGPR).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_pair.hip -o tools/ubench_pair
// Prints ns per 64-lane pair-group per SIMD; the production kernel is compared against these.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned u32;

// All variants: inputs token h (SGPR), a_lo, a_hi (VGPR), b (VGPR pair), produce u (fast fold), min-accumulate.
// V0: what hipcc emits today: mad, mov, mad, lshr, add (+ min3 per two pairs)
// V1: mul_lo first, add into the odd half of the addend pair, one mad, lshr, add
// V2: mad, mul_lo, add, lshr, add
// V3: like V0 but the 4 pairs of a token-quad are software-interleaved by op type (slow ops back to back)
// V4: V1 interleaved by op type
// V5: exact fold (reference cost)
// V6: only the two multiplies + min (lower bound probe)
// V7: V0 without the min (probe)
template <int V>
__device__ __forceinline__ void quad(u32 (&acc)[2], const u32 (&h)[4], const u32 (&alo)[2], const u32 (&ahi)[2], const u64 (&b)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        u32 u[4];
        if constexpr (V == 0 || V == 7) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u64 s0, x;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(s0) : "s"(h[j]), "v"(alo[p]), "v"(b[p]) : "vcc");
                u64 t = (u32)(s0 >> 32);  // v_mov into an even pair
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(x) : "s"(h[j]), "v"(ahi[p]), "v"(t) : "vcc");
                u[j] = (u32)s0 + ((u32)x >> 29);
            }
        } else if constexpr (V == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32 q;
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(q) : "s"(h[j]), "v"(ahi[p]));
                const u64 add = (b[p] & 0xFFFFFFFFull) | ((u64)((u32)(b[p] >> 32) + q) << 32);
                u64 s;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(s) : "s"(h[j]), "v"(alo[p]), "v"(add) : "vcc");
                u[j] = (u32)s + ((u32)(s >> 32) >> 29);
            }
        } else if constexpr (V == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u64 s0;
                u32 q;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(s0) : "s"(h[j]), "v"(alo[p]), "v"(b[p]) : "vcc");
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(q) : "s"(h[j]), "v"(ahi[p]));
                u[j] = (u32)s0 + (((u32)(s0 >> 32) + q) >> 29);
            }
        } else if constexpr (V == 3) {
            u64 s0[4], x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(s0[j]) : "s"(h[j]), "v"(alo[p]), "v"(b[p]) : "vcc");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u64 t = (u32)(s0[j] >> 32);
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(x[j]) : "s"(h[j]), "v"(ahi[p]), "v"(t) : "vcc");
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = (u32)s0[j] + ((u32)x[j] >> 29);
        } else if constexpr (V == 4) {
            u32 q[4];
            u64 s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(q[j]) : "s"(h[j]), "v"(ahi[p]));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u64 add = (b[p] & 0xFFFFFFFFull) | ((u64)((u32)(b[p] >> 32) + q[j]) << 32);
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(s[j]) : "s"(h[j]), "v"(alo[p]), "v"(add) : "vcc");
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = (u32)s[j] + ((u32)(s[j] >> 32) >> 29);
        } else if constexpr (V == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u64 s0 = (u64)h[j] * alo[p] + b[p];
                const u32 s_lo = (u32)s0, s_hi = (u32)(s0 >> 32) + h[j] * ahi[p];
                const u32 top = s_hi >> 29;
                const u64 y = ((((u64)(s_hi & 0x1FFFFFFFu)) << 32) | s_lo) + top;
                u[j] = (u32)y + (y >= ((1ull << 61) - 1) ? 1u : 0u);
            }
        } else if constexpr (V == 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u64 s0;
                u32 q;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(s0) : "s"(h[j]), "v"(alo[p]), "v"(b[p]) : "vcc");
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(q) : "s"(h[j]), "v"(ahi[p]));
                u[j] = (u32)s0 ^ q;
            }
        }
        if constexpr (V == 7) {
            acc[p] ^= u[0] ^ u[1] ^ u[2] ^ u[3];
        } else {
            acc[p] = min(min(acc[p], u[0]), u[1]);
            acc[p] = min(min(acc[p], u[2]), u[3]);
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void pair_kernel(u32 *out, const u32 *tok, int iters) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 alo[2] = {tid * 2654435761u + 1u, tid * 40503u + 7u}, ahi[2] = {(tid * 7919u) & 0x1FFFFFFFu, (tid * 104729u) & 0x1FFFFFFFu};
    u64 b[2] = {((u64)tid << 29) ^ 0x123456789ull, ((u64)tid << 27) ^ 0xABCDEF123ull};
    u32 acc[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    u32 h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = __builtin_amdgcn_readfirstlane(tok[i]);
    for (int it = 0; it < iters; ++it) {
        u32 h0[4] = {h[0], h[1], h[2], h[3]}, h1[4] = {h[4], h[5], h[6], h[7]};
        quad<V>(acc, h0, alo, ahi, b);
        quad<V>(acc, h1, alo, ahi, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] += 0x9E3779B9u;  // SALU: new tokens every iteration
    }
    if ((acc[0] ^ acc[1]) == 0x12345678u) out[tid] = acc[0];
}

template <int V>
void run(const char *name, int cus, int wps, u32 *d_out, u32 *d_tok) {
    const int iters = 4000;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(pair_kernel<V>, dim3(cus * wps), dim3(256), 0, 0, d_out, d_tok, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(pair_kernel<V>, dim3(cus * wps), dim3(256), 0, 0, d_out, d_tok, iters);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    const double groups_per_simd = (double)iters * 16 * wps;  // 8 tokens x 2 perms per iteration per wave
    printf("%-44s w/simd=%d  %7.3f ns per pair-group per SIMD  -> %6.2f ms for 1M x 256 x 128\n", name, wps, ms * 1e6 / groups_per_simd,
           ms * 1e6 / groups_per_simd * 5.24288e8 / 1024 * 1e-6);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u32 *d_out, *d_tok;
    CHK(hipMalloc(&d_out, sizeof(u32) * 256 * cus * 8));
    CHK(hipMalloc(&d_tok, 64));
    u32 h_tok[8] = {0x12345678u, 0x9ABCDEF0u, 0x0F1E2D3Cu, 0x55AA55AAu, 0xDEADBEEFu, 0x01020304u, 0xCAFEBABEu, 0x7F7F7F7Fu};
    CHK(hipMemcpy(d_tok, h_tok, 32, hipMemcpyHostToDevice));
    for (int wps : {4, 8}) {
        run<0>("V0 mad,mov,mad,lshr,add,min3/2 (hipcc today)", cus, wps, d_out, d_tok);
        run<1>("V1 mul_lo,add,mad,lshr,add,min3/2", cus, wps, d_out, d_tok);
        run<2>("V2 mad,mul_lo,add,lshr,add,min3/2", cus, wps, d_out, d_tok);
        run<3>("V3 = V0 grouped by op type over 4 tokens", cus, wps, d_out, d_tok);
        run<4>("V4 = V1 grouped by op type over 4 tokens", cus, wps, d_out, d_tok);
        run<5>("V5 exact fold (compiler)", cus, wps, d_out, d_tok);
        run<6>("V6 two multiplies + xor + min3/2 (probe)", cus, wps, d_out, d_tok);
        run<7>("V7 = V0 with xor instead of min3 (probe)", cus, wps, d_out, d_tok);
    }
    return 0;
}
