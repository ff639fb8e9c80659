// tools/ubench_stream.hip -- which streaming recipe gets the most out of HBM3E for the re-packing kernels
// (one 16-byte load per lane, a byte swap, one 16-byte store: band_keys / merge / lean_serialize are this shape)?
// Variants: loads in flight per lane (1, 2, 4), non-temporal loads / stores, blocks per CU (8, 16, 32).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/ubench_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;

template <int U, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void swap_kernel(const ulonglong2 *__restrict__ in, long n2, ulonglong2 *__restrict__ out) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += U * stride) {
        ulonglong2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (j < n2) {
                if (NT_LOAD) {
                    v[u].x = __builtin_nontemporal_load(&in[j].x);
                    v[u].y = __builtin_nontemporal_load(&in[j].y);
                } else {
                    v[u] = in[j];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (j < n2) {
                ulonglong2 r;
                r.x = __builtin_bswap64(v[u].x);
                r.y = __builtin_bswap64(v[u].y);
                if (NT_STORE) {
                    __builtin_nontemporal_store(r.x, &out[j].x);
                    __builtin_nontemporal_store(r.y, &out[j].y);
                } else {
                    out[j] = r;
                }
            }
        }
    }
}

template <int U, bool NTL, bool NTS>
static void run(const ulonglong2 *in, ulonglong2 *out, long n2, int cus, int bpc) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((swap_kernel<U, NTL, NTS>), dim3(cus * bpc), dim3(256), 0, 0, in, n2, out);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("{\"in_flight\": %d, \"nt_load\": %d, \"nt_store\": %d, \"blocks_per_cu\": %d, \"ms\": %.4f, \"TBps_read_plus_write\": %.3f}\n", U, (int)NTL, (int)NTS,
           bpc, best, 2.0 * n2 * 16 / (best * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const long n2 = 128l << 20;  // 2 GiB in, 2 GiB out (the 1M x 256 uint64 matrix)
    ulonglong2 *in, *out;
    CHK(hipMalloc(&in, n2 * 16));
    CHK(hipMalloc(&out, n2 * 16));
    CHK(hipMemset(in, 1, n2 * 16));
    CHK(hipMemset(out, 0, n2 * 16));
    for (int bpc : {8, 16, 32}) {
        run<1, false, false>(in, out, n2, cus, bpc);
        run<2, false, false>(in, out, n2, cus, bpc);
        run<4, false, false>(in, out, n2, cus, bpc);
        run<4, true, false>(in, out, n2, cus, bpc);
        run<4, false, true>(in, out, n2, cus, bpc);
        run<4, true, true>(in, out, n2, cus, bpc);
        run<1, true, true>(in, out, n2, cus, bpc);
    }
    return 0;
}
