// tools/ubench_rows.hip -- how fast can a READ-ONLY pass over a [100 000 x 4096] float32 matrix go, by who reads what?
// Round 4 ended with config 4's row kernel bound by its stream: fetch + stage + scan alone 0.36-0.37 ms of a 0.39 ms call,
// 4.4-4.5 TB/s, where the read-mostly re-pack kernels reach 5.1-5.8 (DESIGN.md section 7 (b')).  Each variant below reads
// the matrix once, keeps a trivial per-row result (the row's maximum: one v_max per loaded value, so the loads cannot be
// dropped) and writes 4 bytes per row; what differs is the mapping of rows to waves and the number of bytes in flight:
//   own<NV, AHEAD, NT>    one wave owns a row (NV 16-byte loads per lane = 16 KB at NV = 16), AHEAD rows requested ahead of the
//                         one being reduced -- the row kernel's pattern (AHEAD = 2, eight waves per CU, one workgroup per CU)
//   own<...> with more workgroups per CU (the row kernel cannot: its LDS stripes; here: what the pattern could give)
//   shared<NT>            the eight waves of a workgroup read ONE row together (two 1-KB loads each), row after row: a CU streams
//                         one sequential run of rows instead of eight -- what a cooperative fetch into the stripes would look like
//   flat<U, NT>           grid-stride over the whole matrix, U loads in flight per lane (the copy kernels' pattern; row results
//                         are not formed: a grand maximum per thread): the box's read-only line for comparison
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_rows.hip -o tools/ubench_rows      Run: tools/ubench_rows  (one JSON line per variant)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kWave = 64;

template <bool NT>
__device__ __forceinline__ f4 load16(const float *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));  // (the 16-byte form keeps its `nt`; two 8-byte halves lose it when merged)
    return *reinterpret_cast<const f4 *>(p);
}
__device__ __forceinline__ float max4(float m, f4 v) { return fmaxf(fmaxf(fmaxf(m, v.x), fmaxf(v.y, v.z)), v.w); }
__device__ __forceinline__ float wave_max(float m) {
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, kWave));
    return m;
}

template <int NV, int AHEAD, bool NT>
__global__ __launch_bounds__(512) void own_kernel(const float *__restrict__ x, long n_rows, int dim, float *__restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const long stride = (long)gridDim.x * n_waves;
    f4 pre[AHEAD][NV];
    const auto fetch = [&](f4 (&p)[NV], long d) {
        const float *src = x + (d < n_rows ? d : n_rows - 1) * dim;
#pragma unroll
        for (int u = 0; u < NV; ++u) p[u] = load16<NT>(src + (u * kWave + lane) * 4);
    };
    const long d0 = (long)blockIdx.x * n_waves + wave;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) fetch(pre[a], d0 + a * stride);
    for (long d = d0; d < n_rows; d += AHEAD * stride) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const long r = d + a * stride;
            if (r >= n_rows) break;
            float m = -1e30f;
#pragma unroll
            for (int u = 0; u < NV; ++u) m = max4(m, pre[a][u]);
            m = wave_max(m);
            if (lane == 0) out[r] = m;
            fetch(pre[a], r + AHEAD * stride);
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(512) void shared_kernel(const float *__restrict__ x, long n_rows, int dim, float *__restrict__ out) {
    // 512 threads x 16 bytes = 8 KB per load instruction of the workgroup; a 16-KB row is two of them; four rows in flight
    __shared__ float part[4][8];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    constexpr int R = 4;
    f4 pre[R][2];
    const auto fetch = [&](f4 (&p)[2], long d) {
        const float *src = x + (d < n_rows ? d : n_rows - 1) * dim;
        p[0] = load16<NT>(src + threadIdx.x * 4);
        p[1] = load16<NT>(src + 2048 + threadIdx.x * 4);
    };
    const long d0 = (long)blockIdx.x * R, stride = (long)gridDim.x * R;
    for (int a = 0; a < R; ++a) fetch(pre[a], d0 + a);
    for (long d = d0; d < n_rows; d += stride) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
            float m = wave_max(max4(max4(-1e30f, pre[a][0]), pre[a][1]));
            if (lane == 0) part[a][wave] = m;
            fetch(pre[a], d + stride + a);
        }
        __syncthreads();
        if (threadIdx.x < R && d + threadIdx.x < n_rows) {
            float m = part[threadIdx.x][0];
            for (int w = 1; w < 8; ++w) m = fmaxf(m, part[threadIdx.x][w]);
            out[d + threadIdx.x] = m;
        }
        __syncthreads();
    }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void flat_kernel(const float *__restrict__ x, long n4, float *__restrict__ out) {
    const long stride = (long)gridDim.x * blockDim.x;
    float m = -1e30f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load16<NT>(x + 4 * ((i + u * stride) < n4 ? (i + u * stride) : i));
#pragma unroll
        for (int u = 0; u < U; ++u) m = max4(m, v[u]);
    }
    if (m > 1e29f) out[0] = m;  // (never: keeps the loads alive)
}

static hipEvent_t e0, e1;
template <typename F>
static void timed(const char *name, int bpc, double bytes, F launch) {
    // steady clocks first (profiles/r04_clock_ramp.txt): ~0.3 s of the same launch untimed
    for (int i = 0; i < 600; ++i) launch();
    CHK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("{\"variant\": \"%s\", \"workgroups_per_cu\": %d, \"ms_best\": %.4f, \"ms_mean\": %.4f, \"TBps_read\": %.3f}\n", name, bpc, best, sum / reps,
           bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const long n_rows = 100000;
    const int dim = 4096;
    float *x, *out;
    CHK(hipMalloc(&x, n_rows * dim * sizeof(float)));
    CHK(hipMalloc(&out, n_rows * sizeof(float)));
    CHK(hipMemset(x, 0x3f, n_rows * dim * sizeof(float)));
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const double bytes = (double)n_rows * dim * 4;
#define OWN(NV, AHEAD, NT, BPC, WAVES) \
    timed("own<" #NV "," #AHEAD "," #NT "> x " #WAVES " waves", BPC, bytes, [&] { hipLaunchKernelGGL((own_kernel<NV, AHEAD, NT>), dim3(cus * BPC), dim3(64 * WAVES), 0, 0, x, n_rows, dim, out); })
    OWN(16, 2, false, 1, 8);  // the row kernel's pattern
    OWN(16, 2, true, 1, 8);
    OWN(16, 1, false, 1, 8);
    OWN(16, 3, false, 1, 8);
    OWN(16, 2, false, 2, 8);  // what the pattern gives with more waves per CU than the stripes allow
    OWN(16, 2, false, 4, 4);
    OWN(16, 1, false, 4, 8);
    OWN(16, 2, true, 2, 8);
#undef OWN
    for (int bpc : {1, 2, 4}) {
        timed("shared<plain>", bpc, bytes, [&] { hipLaunchKernelGGL((shared_kernel<false>), dim3(cus * bpc), dim3(512), 0, 0, x, n_rows, dim, out); });
        timed("shared<nt>", bpc, bytes, [&] { hipLaunchKernelGGL((shared_kernel<true>), dim3(cus * bpc), dim3(512), 0, 0, x, n_rows, dim, out); });
    }
    const long n4 = n_rows * dim / 4;
    for (int bpc : {8, 32}) {
        timed("flat<4,plain>", bpc, bytes, [&] { hipLaunchKernelGGL((flat_kernel<4, false>), dim3(cus * bpc), dim3(256), 0, 0, x, n4, out); });
        timed("flat<4,nt>", bpc, bytes, [&] { hipLaunchKernelGGL((flat_kernel<4, true>), dim3(cus * bpc), dim3(256), 0, 0, x, n4, out); });
        timed("flat<8,plain>", bpc, bytes, [&] { hipLaunchKernelGGL((flat_kernel<8, false>), dim3(cus * bpc), dim3(256), 0, 0, x, n4, out); });
    }
    return 0;
}
