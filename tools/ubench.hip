// tools/ubench.hip -- issue-rate microbenchmark of the integer VALU instructions the MinHash
// kernel is built from, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
// Prints wave-instructions per ns per SIMD and the cost relative to v_add_u32.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int CHAINS = 8;
constexpr int UNROLL = 16;

template <int OP>
__device__ __forceinline__ void op(unsigned &x0, unsigned &x1, unsigned y, unsigned z, unsigned sc) {
    // x0,x1: chain state (x1 used by 64-bit ops); y,z: loop-invariant vgprs; sc: sgpr
    if constexpr (OP == 0) { unsigned long long d, c = ((unsigned long long)x1 << 32) | x0; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(y), "v"(z), "v"(c) : "vcc"); x0 = (unsigned)d; x1 = (unsigned)(d >> 32); }
    if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 5) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 6) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x0));
    if constexpr (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 9) { unsigned long long d, c = ((unsigned long long)x1 << 32) | x0; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "s"(sc), "v"(z), "v"(c) : "vcc"); x0 = (unsigned)d; x1 = (unsigned)(d >> 32); }
    if constexpr (OP == 10) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(x0), "+v"(x1) : "v"(y), "v"(z) : "vcc");
    if constexpr (OP == 11) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x0) : "v"(c), "v"(d) : "vcc"); }
    if constexpr (OP == 12) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 13) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 14) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 15) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 16) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x0) : "s"(sc));
    if constexpr (OP == 17) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 19) asm volatile("v_mov_b32 %0, %1" : "=v"(x0) : "v"(x1));
    if constexpr (OP == 20) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x0) : "v"(y));
    if constexpr (OP == 21) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(x0));
    if constexpr (OP == 22) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 30) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 31) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 33) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x0));
    if constexpr (OP == 34) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 35) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 36) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 37) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 38) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 39) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(y) : );
    if constexpr (OP == 40) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x0) : "v"(y) : "vcc");
    if constexpr (OP == 41) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(y) : "vcc");
    if constexpr (OP == 42) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 43) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 44) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 45) { unsigned long long c = ((unsigned long long)x1 << 32) | x0; asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(c)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 46) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 47) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 48) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 49) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 50) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "s"(sc));
    if constexpr (OP == 51) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 52) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x0));
    if constexpr (OP == 53) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 54) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 55) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 56) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 57) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 58) asm volatile("v_min_f16 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 59) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 60) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 61) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x0) : "v"(y));
    if constexpr (OP == 62) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 63) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_min_f64 %0, %0, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 70) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_mul_f64 %0, %0, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 71) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 72) { unsigned long long c; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c) : "v"(x0)); x0 ^= (unsigned)c; }
    if constexpr (OP == 73) { unsigned long long c = ((unsigned long long)x1 << 32) | x0; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x0) : "v"(c)); }
    if constexpr (OP == 74) asm volatile("v_floor_f32 %0, %0" : "+v"(x0));
    if constexpr (OP == 75) asm volatile("v_rcp_f32 %0, %0" : "+v"(x0));
    if constexpr (OP == 76) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(y) : "vcc");
    if constexpr (OP == 77) { unsigned long long c; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c) : "s"(sc)); x0 ^= (unsigned)c; }
    if constexpr (OP == 78) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x0) : "v"(y) : "vcc");
    if constexpr (OP == 79) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(z));
    if constexpr (OP == 80) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
    if constexpr (OP == 81) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x0));
    if constexpr (OP == 82) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(y));
    if constexpr (OP == 64) { unsigned long long c = ((unsigned long long)x1 << 32) | x0, d = ((unsigned long long)z << 32) | y; asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(d)); x0 = (unsigned)c; x1 = (unsigned)(c >> 32); }
}

template <int OP>
__global__ __launch_bounds__(256) void bench_kernel(unsigned *out, int iters, unsigned seed) {
    unsigned x0[CHAINS], x1[CHAINS];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int c = 0; c < CHAINS; ++c) { x0[c] = tid * 2654435761u + c * 40503u + seed; x1[c] = tid ^ (c * 7919u); }
    unsigned y = tid * 3u + 1u, z = tid * 5u + 7u;
    const unsigned sc = __builtin_amdgcn_readfirstlane(seed * 77u + 13u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) op<OP>(x0[c], x1[c], y, z, sc);
    }
    unsigned acc = 0;
    for (int c = 0; c < CHAINS; ++c) acc ^= x0[c] ^ x1[c];
    if (acc == 0x12345678u) out[tid] = acc;  // keep results live
}

__global__ void clock_kernel(unsigned long long *out, int iters) {
    unsigned x = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime: shader clock
    const unsigned long long w0 = wall_clock64();                // constant 100 MHz
    for (int i = 0; i < iters; ++i) asm volatile("v_add_u32 %0, %0, %0\n\tv_mul_lo_u32 %0, %0, %0" : "+v"(x));
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (x == 0x12345) out[2] = x;
}

template <int OP>
double run(const char *name, int instr_per_op, int blocks, int waves_per_simd, unsigned *d_out, double base) {
    const int iters = 2000;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(bench_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(bench_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 2u);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr_per_simd = (double)iters * UNROLL * CHAINS * instr_per_op * waves_per_simd;
    const double ns_per = ms * 1e6 / wave_instr_per_simd;
    printf("%-34s w/simd=%d  %8.3f ns/wave-instr/SIMD  (%5.2f cyc @2.4GHz)  x%.2f vs v_add_u32\n", name, waves_per_simd, ns_per, ns_per * 2.4, base > 0 ? ns_per / base : 1.0);
    fflush(stdout);
    return ns_per;
}

int main(int argc, char **argv) {
    const bool fp_only = argc > 1 && argv[1][0] == 'f';
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    unsigned *d_out; CHK(hipMalloc(&d_out, sizeof(unsigned) * 256 * cus * 8));
    {
        unsigned long long *d_clk, h_clk[2];
        CHK(hipMalloc(&d_clk, 64));
        hipLaunchKernelGGL(clock_kernel, dim3(cus * 8), dim3(256), 0, 0, d_clk, 2000000);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
        printf("shader clock under full VALU load: %.0f MHz (s_memtime ticks %llu over %llu wall ticks @100MHz)\n",
               (double)h_clk[0] / ((double)h_clk[1] / 100.0), h_clk[0], h_clk[1]);
    }
    for (int wps : {2, 8}) {
        const int blocks = cus * wps;  // 256-thread block = 1 wave per SIMD
        double base = run<3>("v_add_u32", 1, blocks, wps, d_out, 0);
        run<70>("v_mul_f64", 1, blocks, wps, d_out, base);
        run<71>("v_fma_f64", 1, blocks, wps, d_out, base);
        run<72>("v_cvt_f64_f32 (+xor)", 2, blocks, wps, d_out, base);
        run<77>("v_cvt_f64_f32 sgpr src (+xor)", 2, blocks, wps, d_out, base);
        run<73>("v_cvt_f32_f64", 1, blocks, wps, d_out, base);
        run<74>("v_floor_f32", 1, blocks, wps, d_out, base);
        run<75>("v_rcp_f32", 1, blocks, wps, d_out, base);
        run<76>("v_cmp_lt_f32+v_cndmask (pair)", 2, blocks, wps, d_out, base);
        run<78>("v_div_scale_f32", 1, blocks, wps, d_out, base);
        run<79>("v_div_fixup_f32", 1, blocks, wps, d_out, base);
        run<80>("v_pk_mul_f32", 1, blocks, wps, d_out, base);
        run<81>("v_cvt_f32_u32", 1, blocks, wps, d_out, base);
        run<36>("v_min_f32", 1, blocks, wps, d_out, base);
        run<37>("v_add_f32", 1, blocks, wps, d_out, base);
        run<48>("v_mul_f32", 1, blocks, wps, d_out, base);
        run<49>("v_pk_add_f32", 1, blocks, wps, d_out, base);
        if (fp_only) continue;
        run<4>("v_add3_u32", 1, blocks, wps, d_out, base);
        run<5>("v_min3_u32", 1, blocks, wps, d_out, base);
        run<17>("v_min_u32", 1, blocks, wps, d_out, base);
        run<6>("v_lshrrev_b32", 1, blocks, wps, d_out, base);
        run<18>("v_and_b32", 1, blocks, wps, d_out, base);
        run<19>("v_mov_b32", 1, blocks, wps, d_out, base);
        run<20>("v_alignbit_b32", 1, blocks, wps, d_out, base);
        run<21>("v_bfe_u32", 1, blocks, wps, d_out, base);
        run<22>("v_lshl_add_u32", 1, blocks, wps, d_out, base);
        run<1>("v_mul_lo_u32", 1, blocks, wps, d_out, base);
        run<16>("v_mul_lo_u32 (sgpr src)", 1, blocks, wps, d_out, base);
        run<2>("v_mul_hi_u32", 1, blocks, wps, d_out, base);
        run<0>("v_mad_u64_u32", 1, blocks, wps, d_out, base);
        run<9>("v_mad_u64_u32 (sgpr src)", 1, blocks, wps, d_out, base);
        run<7>("v_mul_u32_u24", 1, blocks, wps, d_out, base);
        run<8>("v_mad_u32_u24", 1, blocks, wps, d_out, base);
        run<14>("v_mul_hi_u32_u24", 1, blocks, wps, d_out, base);
        run<15>("v_pk_mul_lo_u16", 1, blocks, wps, d_out, base);
        run<13>("v_dot4_u32_u8", 1, blocks, wps, d_out, base);
        run<10>("v_add_co+v_addc_co (pair)", 2, blocks, wps, d_out, base);
        run<11>("v_cmp_lt_u64+v_addc_co (pair)", 2, blocks, wps, d_out, base);
        run<12>("v_lshl_add_u64", 1, blocks, wps, d_out, base);
        run<30>("v_sub_u32", 1, blocks, wps, d_out, base);
        run<51>("v_subrev_u32", 1, blocks, wps, d_out, base);
        run<50>("v_add_u32 (sgpr src)", 1, blocks, wps, d_out, base);
        run<31>("v_or_b32", 1, blocks, wps, d_out, base);
        run<32>("v_xor_b32", 1, blocks, wps, d_out, base);
        run<33>("v_lshlrev_b32", 1, blocks, wps, d_out, base);
        run<52>("v_ashrrev_i32", 1, blocks, wps, d_out, base);
        run<34>("v_max_u32", 1, blocks, wps, d_out, base);
        run<35>("v_min_i32", 1, blocks, wps, d_out, base);
        run<36>("v_min_f32", 1, blocks, wps, d_out, base);
        run<57>("v_max_f32", 1, blocks, wps, d_out, base);
        run<37>("v_add_f32", 1, blocks, wps, d_out, base);
        run<48>("v_mul_f32", 1, blocks, wps, d_out, base);
        run<38>("v_fma_f32", 1, blocks, wps, d_out, base);
        run<49>("v_pk_add_f32", 1, blocks, wps, d_out, base);
        run<62>("v_pk_fma_f32", 1, blocks, wps, d_out, base);
        run<63>("v_min_f64", 1, blocks, wps, d_out, base);
        run<64>("v_add_f64", 1, blocks, wps, d_out, base);
        run<39>("v_cndmask_b32", 1, blocks, wps, d_out, base);
        run<40>("v_add_co_u32", 1, blocks, wps, d_out, base);
        run<41>("v_cmp_lt_u32+v_cndmask (pair)", 2, blocks, wps, d_out, base);
        run<42>("v_bfi_b32", 1, blocks, wps, d_out, base);
        run<43>("v_and_or_b32", 1, blocks, wps, d_out, base);
        run<44>("v_xad_u32", 1, blocks, wps, d_out, base);
        run<45>("v_lshrrev_b64", 1, blocks, wps, d_out, base);
        run<46>("v_pk_add_u16", 1, blocks, wps, d_out, base);
        run<47>("v_pk_min_u16", 1, blocks, wps, d_out, base);
        run<61>("v_pk_min_i16", 1, blocks, wps, d_out, base);
        run<60>("v_pk_mad_u16", 1, blocks, wps, d_out, base);
        run<56>("v_min_u16", 1, blocks, wps, d_out, base);
        run<58>("v_min_f16", 1, blocks, wps, d_out, base);
        run<59>("v_add_u16", 1, blocks, wps, d_out, base);
        run<53>("v_sad_u32", 1, blocks, wps, d_out, base);
        run<54>("v_med3_u32", 1, blocks, wps, d_out, base);
        run<55>("v_max3_u32", 1, blocks, wps, d_out, base);
    }
    return 0;
}
