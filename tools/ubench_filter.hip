// tools/ubench_filter.hip -- what one candidate-filter test ("w - L < thr", w and thr per lane, L wave-uniform)
// costs on gfx950 in the instruction sequences the weighted dense kernel could be built from.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_filter.hip -o tools/ubench_filter
// Prints cycles per test per SIMD at 1, 2, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kTestsPerIter = 32;

// The tests never pass (thr = -inf) unless TAKEN says so: every TAKEN-th test uses an L that passes in all lanes.
template <int SEQ>
__global__ __launch_bounds__(64) void k(float *out, const float *in, int iters, float lpass) {
    float w0 = in[threadIdx.x], w1 = in[64 + threadIdx.x];
    float thr = -__builtin_inff();
    float l0 = in[128], l1 = in[129];  // wave-uniform -> SGPRs
    l0 = __builtin_amdgcn_readfirstlane(l0);
    l1 = __builtin_amdgcn_readfirstlane(l1);
    float lp = __builtin_amdgcn_readfirstlane(lpass);
    unsigned mask = 0, cnt = 0;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (SEQ == 0) {  // v_sub + v_cmp, no branch
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t)
                asm volatile("v_subrev_f32 %0, %2, %1\n\tv_cmp_lt_f32 vcc, %0, %3" : "=&v"(acc) : "v"(w0), "s"(l0), "v"(thr) : "vcc");
        }
        if constexpr (SEQ == 1) {  // + a branch per test, never taken
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t)
                asm volatile("v_subrev_f32 %0, %2, %1\n\tv_cmp_lt_f32 vcc, %0, %3\n\ts_cbranch_vccnz 1f\n\ts_branch 2f\n1:\n\tv_add_u32 %4, %4, 1\n2:"
                             : "=&v"(acc), "+v"(cnt) : "v"(w0), "s"(l0), "v"(thr) : "vcc");
        }
        if constexpr (SEQ == 2) {  // the same with the fall-through on the not-taken side (one branch instruction per test)
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t)
                asm volatile("v_subrev_f32 %0, %2, %1\n\tv_cmp_lt_f32 vcc, %0, %3\n\ts_cbranch_vccz 1f\n\tv_add_u32 %4, %4, 1\n1:"
                             : "=&v"(acc), "+v"(cnt) : "v"(w0), "s"(l0), "v"(thr) : "vcc");
        }
        if constexpr (SEQ == 3) {  // masks OR-ed on the scalar unit, one branch per 8 tests
#pragma unroll
            for (int t = 0; t < kTestsPerIter; t += 8) {
                unsigned long long m;
                asm volatile("v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_mov_b64 %1, vcc\n\t"
                             "v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_or_b64 %1, %1, vcc\n\t"
                             "v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_or_b64 %1, %1, vcc\n\t"
                             "v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_or_b64 %1, %1, vcc\n\t"
                             "v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_or_b64 %1, %1, vcc\n\t"
                             "v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_or_b64 %1, %1, vcc\n\t"
                             "v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_or_b64 %1, %1, vcc\n\t"
                             "v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\ts_or_b64 %1, %1, vcc\n\t"
                             "s_cbranch_scc0 1f\n\tv_add_u32 %5, %5, 1\n1:"
                             : "=&v"(acc), "=&s"(m), "+v"(cnt) : "s"(l0), "v"(thr), "v"(w0) : "vcc", "scc");
            }
        }
        if constexpr (SEQ == 4) {  // sign of (w - L) - thr shifted into a per-lane bit mask (no vcc), branch per 32
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t)
                asm volatile("v_subrev_f32 %0, %3, %2\n\tv_sub_f32 %0, %0, %4\n\tv_alignbit_b32 %1, %1, %0, 31"
                             : "=&v"(acc), "+v"(mask) : "v"(w0), "s"(l0), "v"(thr));
            if (__builtin_expect(__any(mask != 0), 0)) cnt += mask;
        }
        if constexpr (SEQ == 5) {  // the same with packed subtractions: two tests per v_pk_add_f32
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f w2 = {w0, w1}, t2 = {thr, thr}, d;
            const unsigned long long lpair = ((unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(l1)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(l0));
#pragma unroll
            for (int t = 0; t < kTestsPerIter; t += 2) {
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]"
                             : "=&v"(d) : "v"(w2), "s"(lpair), "v"(t2));
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(d.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(d.y), 31);
            }
            if (__builtin_expect(__any(mask != 0), 0)) cnt += mask;
        }
        if constexpr (SEQ == 12 || SEQ == 13) {  // the kernel's form: two mask chains; 13: the logs in a VGPR pair
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f w2 = {w0, w1}, t2 = {thr, thr}, d, e, lv = {l0, l1};
            asm volatile("" : "+v"(lv));
            const unsigned long long lpair = ((unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(l1)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(l0));
            unsigned mb = cnt;
#pragma unroll
            for (int t = 0; t < kTestsPerIter; t += 4) {
                if constexpr (SEQ == 12) {
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %3 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                                 : "=&v"(d) : "v"(w2), "s"(lpair), "v"(t2));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %3 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]"
                                 : "=&v"(e) : "v"(w2), "s"(lpair), "v"(t2));
                } else {
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %3 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                                 : "=&v"(d) : "v"(w2), "v"(lv), "v"(t2));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %3 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]"
                                 : "=&v"(e) : "v"(w2), "v"(lv), "v"(t2));
                }
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(d.x), 31);
                mb = __builtin_amdgcn_alignbit(mb, __float_as_uint(e.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(d.y), 31);
                mb = __builtin_amdgcn_alignbit(mb, __float_as_uint(e.y), 31);
            }
            cnt = mb;
            if (__builtin_expect(__any((mask | mb) != 0), 0)) cnt += mask;
        }
        if constexpr (SEQ == 6) {  // v_sub + v_cmp + v_addc (mask = 2 mask + pass)
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t)
                asm volatile("v_subrev_f32 %0, %3, %2\n\tv_cmp_lt_f32 vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                             : "=&v"(acc), "+v"(mask) : "v"(w0), "s"(l0), "v"(thr) : "vcc");
            if (__builtin_expect(__any(mask != 0), 0)) cnt += mask;
        }
        if constexpr (SEQ == 7) {  // branch per test, every 4th test taken to an out-of-line push-like stub (5 VALU + LDS write)
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t) {
                if (t % 4 == 3)
                    asm volatile("v_subrev_f32 %0, %2, %1\n\tv_cmp_gt_f32 vcc, %0, %3\n\ts_cbranch_vccz 1f\n\t"
                                 "s_and_saveexec_b64 s[20:21], vcc\n\tv_mbcnt_lo_u32_b32 %0, vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %0, vcc_hi, %0\n\t"
                                 "v_lshlrev_b32 %0, 2, %0\n\tv_add_u32 %4, %4, 1\n\tds_write_b32 %0, %4\n\ts_or_b64 exec, exec, s[20:21]\n1:"
                                 : "=&v"(acc), "+v"(cnt) : "v"(w0), "s"(lp), "v"(thr) : "vcc", "s20", "s21", "memory");
                else
                    asm volatile("v_subrev_f32 %0, %2, %1\n\tv_cmp_lt_f32 vcc, %0, %3\n\ts_cbranch_vccz 1f\n\tv_add_u32 %4, %4, 1\n1:"
                                 : "=&v"(acc), "+v"(cnt) : "v"(w0), "s"(l0), "v"(thr) : "vcc");
            }
        }
        if constexpr (SEQ == 8) {  // v_sub with the log in a VGPR (2.5-cycle form) + v_cmp
            float lv = l0;
            asm volatile("" : "+v"(lv));
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t)
                asm volatile("v_sub_f32 %0, %1, %2\n\tv_cmp_lt_f32 vcc, %0, %3" : "=&v"(acc) : "v"(w0), "v"(lv), "v"(thr) : "vcc");
        }
        if constexpr (SEQ == 9) {  // v_cmp only (w < thr + L folded elsewhere): the floor of a compare per test
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(w0), "v"(thr) : "vcc");
        }
        if constexpr (SEQ == 10) {  // v_cmp with the scalar as src0 (w - thr precomputed per row: compare against L directly)
#pragma unroll
            for (int t = 0; t < kTestsPerIter; ++t) asm volatile("v_cmp_gt_f32 vcc, %1, %0" : : "v"(w0), "s"(l0) : "vcc");
        }
        if constexpr (SEQ == 11) {  // v_cmp_e64 into SGPR pairs + s_or, one branch per 8 (no v_sub)
#pragma unroll
            for (int t = 0; t < kTestsPerIter; t += 8) {
                unsigned long long m;
                asm volatile("v_cmp_gt_f32 vcc, %3, %1\n\ts_mov_b64 %0, vcc\n\t"
                             "v_cmp_gt_f32 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc\n\t"
                             "v_cmp_gt_f32 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc\n\t"
                             "v_cmp_gt_f32 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc\n\t"
                             "v_cmp_gt_f32 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc\n\t"
                             "v_cmp_gt_f32 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc\n\t"
                             "v_cmp_gt_f32 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc\n\t"
                             "v_cmp_gt_f32 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc\n\t"
                             "s_cbranch_scc0 1f\n\tv_add_u32 %2, %2, 1\n1:"
                             : "=&s"(m), "+v"(w0), "+v"(cnt) : "s"(l0) : "vcc", "scc");
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + (float)cnt + (float)mask;
}

template <int SEQ>
void run(const char *name, float *d_out, const float *d_in, int cus, float lpass) {
    const int iters = 1500;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * 4 * wps;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<SEQ>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, 100, lpass);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<SEQ>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters, lpass);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double tests_per_simd = (double)iters * kTestsPerIter * wps;
        const double ns = ms * 1e6 / tests_per_simd;
        printf("%-64s waves/SIMD=%d  %7.3f ns/test/SIMD  (%6.2f cyc @2.4GHz)\n", name, wps, ns, ns * 2.4);
    }
}

int main(int argc, char **argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device: %s, %d CUs\n", p.name, cus);
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + i * 0.001f;
    float *d_in, *d_out;
    CHK(hipMalloc(&d_in, sizeof(h)));
    CHK(hipMalloc(&d_out, sizeof(float) * 64 * cus * 32));
    CHK(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));
    if (only < 0 || only == 0) run<0>("v_sub(sgpr) + v_cmp", d_out, d_in, cus, 0);
    if (only < 0 || only == 8) run<8>("v_sub(vgpr) + v_cmp", d_out, d_in, cus, 0);
    if (only < 0 || only == 9) run<9>("v_cmp alone (vgpr, vgpr)", d_out, d_in, cus, 0);
    if (only < 0 || only == 10) run<10>("v_cmp alone (sgpr, vgpr)", d_out, d_in, cus, 0);
    if (only < 0 || only == 1) run<1>("v_sub + v_cmp + branch (pass side falls through; never passes)", d_out, d_in, cus, 0);
    if (only < 0 || only == 2) run<2>("v_sub + v_cmp + branch (skip side is the taken branch)", d_out, d_in, cus, 0);
    if (only < 0 || only == 3) run<3>("v_sub + v_cmp + s_or, branch per 8", d_out, d_in, cus, 0);
    if (only < 0 || only == 11) run<11>("v_cmp + s_or, branch per 8", d_out, d_in, cus, 0);
    if (only < 0 || only == 4) run<4>("v_sub + v_sub + v_alignbit (bit mask), branch per 32", d_out, d_in, cus, 0);
    if (only < 0 || only == 5) run<5>("2 x v_pk_add per 2 tests + v_alignbit, branch per 32", d_out, d_in, cus, 0);
    if (only < 0 || only == 12) run<12>("2 x v_pk_add (sgpr logs) per 2 tests + v_alignbit, two chains", d_out, d_in, cus, 0);
    if (only < 0 || only == 13) run<13>("2 x v_pk_add (vgpr logs) per 2 tests + v_alignbit, two chains", d_out, d_in, cus, 0);
    if (only < 0 || only == 6) run<6>("v_sub + v_cmp + v_addc (bit mask), branch per 32", d_out, d_in, cus, 0);
    if (only < 0 || only == 7) run<7>("branch per test, every 4th taken into a push stub", d_out, d_in, cus, -1e30f);
    return 0;
}
