#!/usr/bin/env python3
"""tools/experiments/r04_minhash_phase_cycles.py -- instrumented copy of csrc/minhash_kernels.hip (_variant_tmp.hip): every
wave of the sieve launch adds up the shader-clock cycles per set spent (a) until its offsets and first tokens are there,
(b) in the rows + rescan of the sieve, (c) in the < 16-token tail, (d) storing; lane 0 adds them to a device array at the end
and launch_typed prints the per-set averages.  Build: tools/build_variant.sh datasketch_amd/csrc/_variant_tmp.hip prof minhash_kernels"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "datasketch_amd", "csrc", "minhash_kernels.hip")).read()


def sub(old, new, count=1):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, count)


sub("constexpr int kRowTokens = 16;", "__device__ unsigned long long mhx_dbg[16];\n__device__ unsigned long long mhx_t[4];\nconstexpr int kRowTokens = 16;")
sub("      while (todo) {\n        const int bit = __builtin_ctzll(todo);", "      while (todo) {\n        const unsigned long long p0 = (unsigned long long)clock64();\n        const int bit = __builtin_ctzll(todo);")
sub("        bool defer = false;  // MODE_SIEVE: leave this set to the MODE_FULL launch",
    "        asm volatile(\"\" ::\"s\"(beg), \"s\"(end));\n        const unsigned long long p1 = (unsigned long long)clock64();\n        bool defer = false;  // MODE_SIEVE: leave this set to the MODE_FULL launch")
sub("        asm volatile(\"\" ::\"v\"(warm));  // the warm-up load retires here, a whole set later",
    "        {\n            const unsigned long long p2 = (unsigned long long)clock64();\n            acc0 += p1 - p0, acc1 += p2 - p1, acc2 += 1;\n        }\n        asm volatile(\"\" ::\"v\"(warm));  // the warm-up load retires here, a whole set later")
sub("    SieveBackoff backoff;\n    TiesBackoff ties;\n    int tried = 0", "    unsigned long long acc0 = 0, acc1 = 0, acc2 = 0;\n    SieveBackoff backoff;\n    TiesBackoff ties;\n    int tried = 0")
sub("    // (a sample of the waves publishes: atomics of all 65 536 waves on one word would serialise for over a millisecond)",
    "    if (kSieve && lane == 0 && acc2) atomicAdd(&mhx_dbg[0], acc0), atomicAdd(&mhx_dbg[3], acc1), atomicAdd(&mhx_dbg[4], acc2);\n    // (a sample of the waves publishes: atomics of all 65 536 waves on one word would serialise for over a millisecond)")
sub("    MHX_HIP_CHECK(hipGetLastError());\n    return MHX_OK;\n}\n\ntemplate <typename TokT, typename OutT>\nint launch_p(",
    "    MHX_HIP_CHECK(hipGetLastError());\n    {\n        unsigned long long h[16];\n        MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));\n        MHX_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(mhx_dbg), sizeof(h)));\n"
    "        const double ns = (double)(h[4] ? h[4] : 1);\n        fprintf(stderr, \"per set: offsets+warm=%.0f whole_set_after_offsets=%.0f cycles (sets=%.0f)\\n\", h[0] / ns, h[3] / ns, ns);\n"
    "        unsigned long long z[16] = {0};\n        MHX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mhx_dbg), z, sizeof(z)));\n    }\n    return MHX_OK;\n}\n\ntemplate <typename TokT, typename OutT>\nint launch_p(")
out = os.path.join(ROOT, "datasketch_amd", "csrc", "_variant_tmp.hip")
open(out, "w").write(s)
print(out)
