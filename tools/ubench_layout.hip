// tools/ubench_layout.hip -- the layout BASELINE.json's north_star sketches ("LDS-staged permutation coefficients and
// wavefront __shfl/min reductions"), built and timed against the production mapping, as the kept artefact for DESIGN.md's
// claim that it is the slower design.
//
//   layout B (this file): TOKENS on lanes.  A wave takes one set of 256 tokens, lane l holding tokens l, l+64, l+128,
//   l+192 in registers; the K coefficient pairs (a, b) sit in LDS (staged once per workgroup); for every permutation k
//   the lane hashes its four tokens -- with the SAME arithmetic the production sieve uses for a key, one v_mad_u64_u32
//   per pair, low word only, so that this is a comparison of layouts, not of hash formulations -- keeps the smallest,
//   and the wave reduces across lanes with a 6-step DPP / shuffle min; lane 0 of the reduction stores out[k].
//   (This computes min of the low-word keys, which is what the production sieve's hot loop computes too; the exact
//   candidate step that follows there is the same in both layouts and is left out here: it only adds to layout B.)
//
//   production (permutations on lanes, tokens on the scalar path): per (token, perm) pair 1 mad + 1/2 min3, NO cross-lane
//   step; layout B: per pair 1 mad + 3/4 min, plus per (set, perm) an LDS broadcast read of (a, b) and 6 cross-lane
//   min steps that 64 lanes execute to produce ONE value.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_layout.hip -o tools/ubench_layout ; run: tools/ubench_layout
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned u32;

__device__ __forceinline__ u32 key(u32 h, u32 a_lo, u64 b8) {
    u64 r = (u64)h * a_lo + b8;
    asm("" : "+v"(r));
    return (u32)r;
}

__device__ __forceinline__ u32 wave_min(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (u32)__shfl_xor((int)v, d));
    return v;
}

template <int K>
__global__ __launch_bounds__(256) void layout_b(const u64 *__restrict__ tokens, const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                long n_sets, u32 *__restrict__ out) {
    __shared__ u32 s_alo[K];
    __shared__ u64 s_b8[K];
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        s_alo[k] = (u32)a[k];
        s_b8[k] = b[k] + 8;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long set = (long)blockIdx.x * 4 + wave; set < n_sets; set += (long)gridDim.x * 4) {
        const u64 *t = tokens + set * 256;
        const u32 h0 = (u32)t[lane], h1 = (u32)t[lane + 64], h2 = (u32)t[lane + 128], h3 = (u32)t[lane + 192];
        u32 mine = 0;  // lane k % 64 keeps result k of the current group of 64 permutations
        for (int k = 0; k < K; ++k) {
            const u32 alo = s_alo[k];  // LDS broadcast reads
            const u64 b8 = s_b8[k];
            u32 m = min(min(key(h0, alo, b8), key(h1, alo, b8)), min(key(h2, alo, b8), key(h3, alo, b8)));
            m = wave_min(m);
            if ((k & 63) == lane) mine = m;
            if ((k & 63) == 63) out[set * K + (k & ~63) + lane] = mine;  // coalesced row store, 64 results at a time
        }
    }
}

// production mapping reduced to the same work (keys only, no candidate step): permutations on lanes, tokens scalar
template <int K>
__global__ __launch_bounds__(256) void layout_a(const u64 *__restrict__ tokens, const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                long n_sets, u32 *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int P = K / 64;
    u32 alo[P];
    u64 b8[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        alo[p] = (u32)a[p * 64 + lane];
        b8[p] = b[p * 64 + lane] + 8;
    }
    typedef const u32 __attribute__((address_space(4))) *cptr;
    for (long set = (long)blockIdx.x * 4 + wave; set < n_sets; set += (long)gridDim.x * 4) {
        cptr t = (cptr)(tokens + set * 256);
        u32 m[P];
#pragma unroll
        for (int p = 0; p < P; ++p) m[p] = 0xFFFFFFFFu;
        for (int c = 0; c < 256; c += 8) {
            u32 h[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = t[2 * (c + i)];  // low words; scalar loads
#pragma unroll
            for (int i = 0; i < 8; i += 2)
#pragma unroll
                for (int p = 0; p < P; ++p) m[p] = min(min(m[p], key(h[i], alo[p], b8[p])), key(h[i + 1], alo[p], b8[p]));
        }
#pragma unroll
        for (int p = 0; p < P; ++p) out[set * K + p * 64 + lane] = m[p];
    }
}

int main() {
    constexpr int K = 128;
    const long n = 1000000;
    std::vector<u64> tok((size_t)n * 256), a(K), b(K);
    u64 s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &v : tok) v = rnd() & 0xFFFFFFFFull;
    for (int k = 0; k < K; ++k) { a[k] = (rnd() % ((1ull << 61) - 2)) + 1; b[k] = rnd() % ((1ull << 61) - 1); }
    u64 *d_tok, *d_a, *d_b;
    u32 *d_o1, *d_o2;
    CHK(hipMalloc(&d_tok, tok.size() * 8)); CHK(hipMalloc(&d_a, K * 8)); CHK(hipMalloc(&d_b, K * 8));
    CHK(hipMalloc(&d_o1, (size_t)n * K * 4)); CHK(hipMalloc(&d_o2, (size_t)n * K * 4));
    CHK(hipMemcpy(d_tok, tok.data(), tok.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_a, a.data(), K * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_b, b.data(), K * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        for (int blocks_per_cu : {8, 16, 64}) {
            const int grid = 256 * blocks_per_cu;
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(layout_a<K>, dim3(grid), dim3(256), 0, 0, d_tok, d_a, d_b, n, d_o1);
                else hipLaunchKernelGGL(layout_b<K>, dim3(grid), dim3(256), 0, 0, d_tok, d_a, d_b, n, d_o2);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("%s  grid %5d x 256: %.3f ms per 1M sets x 256 tokens x %d permutations (keys only)\n",
                   which == 0 ? "layout A (perms on lanes, scalar tokens)   " : "layout B (tokens on lanes, LDS coeffs, shfl)", grid, best, K);
        }
    }
    std::vector<u32> o1((size_t)4096 * K), o2((size_t)4096 * K);
    CHK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < o1.size(); ++i) bad += o1[i] != o2[i];
    // host check of a few values
    for (int set = 0; set < 3; ++set)
        for (int k : {0, 77, 127}) {
            u32 m = 0xFFFFFFFFu;
            for (int t = 0; t < 256; ++t) { u32 v = (u32)((u64)(u32)tok[(size_t)set * 256 + t] * (u32)a[k] + b[k] + 8); if (v < m) m = v; }
            bad += m != o1[(size_t)set * K + k];
        }
    printf("both layouts agree with each other and with the host on the sampled rows: %s\n", bad ? "NO" : "yes");
    return bad != 0;
}
