// tools/ubench_ldsmin.hip -- can the LDS pipe take a share of the sieve's row minima off the VALU?  (round 6)
// The sieve's hot loop is one v_mad_u64_u32 per (token, permutation) pair and one v_min3_u32 per two pairs: 1.5 VALU instructions per
// pair, and the kernel is VALU-issue bound.  An LDS atomic (ds_min_u32 without return) issues on the LDS port, not the VALU's: if
// LDSK of a row's 16 keys per permutation go to a wave-private LDS cell (first a ds_write_b32 -- the reset --, then ds_min_u32, one
// ds_read_b32 at the end of the row) the VALU is left with the multiplies and (16 - LDSK) / 2 min3.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsmin.hip -o build/ubench_ldsmin
// Prints ns per row (16 tokens x 2 permutations x 64 lanes) per SIMD and the time the headline's 1M x 256 x 128 would take at that rate.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned u32;
typedef __attribute__((address_space(3))) u32 lds_u32;

__device__ __forceinline__ u32 key(u32 h, u32 a, u64 b8) {
    u64 r;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "s"(h), "v"(a), "v"(b8) : "vcc");
    return (u32)r;
}
__device__ __forceinline__ u32 umin3(u32 x, u32 y, u32 z) { return min(min(x, y), z); }

// LDSK: keys of a row (per permutation) whose minimum is taken by the LDS; DEFER: the cell is read one row later (two cells per permutation)
template <int LDSK, bool DEFER>
__global__ __launch_bounds__(256) void row_kernel(u32 *out, const u32 *tok, int iters) {
    __shared__ u32 cells[4][2][2][64];  // [wave][parity][perm][lane]
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 alo[2] = {tid * 2654435761u + 1u, tid * 40503u + 7u};
    u64 b8[2] = {((u64)tid << 29) ^ 0x123456789ull, ((u64)tid << 27) ^ 0xABCDEF123ull};
    u32 k1[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, k2[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    u32 h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = __builtin_amdgcn_readfirstlane(tok[i]);
    if (LDSK > 0) {
        cells[wave][0][0][lane] = cells[wave][0][1][lane] = cells[wave][1][0][lane] = cells[wave][1][1][lane] = 0xFFFFFFFFu;
    }
    for (int it = 0; it < iters; ++it) {
        const int par = DEFER ? (it & 1) : 0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            u32 *cell = &cells[wave][par][p][lane];
            u32 row = 0xFFFFFFFFu;
            u32 prev = 0xFFFFFFFFu;
            if (LDSK > 0 && DEFER) prev = cells[wave][par ^ 1][p][lane];  // last row's LDS share (issued first: it is back when the row ends)
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const u32 m0 = key(h[j], alo[p], b8[p]), m1 = key(h[j + 1], alo[p], b8[p]);
                if (j < LDSK) {
                    if (j == 0) {
                        asm volatile("ds_write_b32 %0, %1" ::"v"((u32)(uintptr_t)(lds_u32 *)cell), "v"(m0) : "memory");
                    } else {
                        asm volatile("ds_min_u32 %0, %1" ::"v"((u32)(uintptr_t)(lds_u32 *)cell), "v"(m0) : "memory");
                    }
                    asm volatile("ds_min_u32 %0, %1" ::"v"((u32)(uintptr_t)(lds_u32 *)cell), "v"(m1) : "memory");
                } else {
                    row = umin3(row, m0, m1);
                }
            }
            if (LDSK > 0 && !DEFER) prev = *(volatile u32 *)cell;
            if (LDSK > 0) row = min(row, prev);
            // the row's tag and the (smallest, second) record, as in the kernel
            const u32 tagged = (row & ~15u) | (u32)(it & 15);
            u32 med;
            asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(k1[p]), "v"(k2[p]), "v"(tagged));
            k2[p] = med;
            k1[p] = min(k1[p], tagged);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] += 0x9E3779B9u;  // SALU: new tokens every iteration
    }
    if ((k1[0] ^ k1[1] ^ k2[0] ^ k2[1]) == 0x12345678u) out[tid] = k1[0];
}

template <int LDSK, bool DEFER>
void run(const char *name, int cus, int wps, u32 *d_out, u32 *d_tok) {
    const int iters = 4000;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((row_kernel<LDSK, DEFER>), dim3(cus * wps), dim3(256), 0, 0, d_out, d_tok, 10);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((row_kernel<LDSK, DEFER>), dim3(cus * wps), dim3(256), 0, 0, d_out, d_tok, iters);
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double rows_per_simd = (double)iters * wps;  // one row (16 tokens x 2 perms) per iteration per wave
    const double ns_row = best * 1e6 / rows_per_simd;
    // headline: 1M sets x 16 rows over 1024 SIMDs
    printf("%-64s w/simd=%d  %8.3f ns per row per SIMD -> %6.3f ms for 1M x 256 x 128 (hot loop + row tags only)\n", name, wps, ns_row,
           ns_row * 16.0e6 / 1024 * 1e-6);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u32 *d_out, *d_tok;
    CHK(hipMalloc(&d_out, sizeof(u32) * 256 * cus * 8));
    CHK(hipMalloc(&d_tok, 64));
    u32 h_tok[16] = {0x12345678u, 0x9ABCDEF0u, 0x0F1E2D3Cu, 0x55AA55AAu, 0xDEADBEEFu, 0x01020304u, 0xCAFEBABEu, 0x7F7F7F7Fu,
                     0x31415926u, 0x27182818u, 0x16180339u, 0x14142135u, 0x17320508u, 0x22360679u, 0x24494897u, 0x26457513u};
    CHK(hipMemcpy(d_tok, h_tok, 64, hipMemcpyHostToDevice));
    for (int wps : {2, 4, 8}) {
        run<0, false>("VALU only: 32 mad + 16 min3 per row", cus, wps, d_out, d_tok);
        run<2, false>("2 of 16 keys per permutation through ds_min_u32", cus, wps, d_out, d_tok);
        run<4, false>("4 of 16", cus, wps, d_out, d_tok);
        run<4, true>("4 of 16, cell read a row later", cus, wps, d_out, d_tok);
        run<6, true>("6 of 16, cell read a row later", cus, wps, d_out, d_tok);
        run<8, true>("8 of 16, cell read a row later", cus, wps, d_out, d_tok);
        run<12, true>("12 of 16, cell read a row later", cus, wps, d_out, d_tok);
        run<16, true>("16 of 16, cell read a row later", cus, wps, d_out, d_tok);
    }
    return 0;
}
