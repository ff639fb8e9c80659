#!/usr/bin/env python3
"""tools/experiments/r04_walk_phase_cycles.py -- writes an instrumented copy of csrc/weighted_kernels.hip
(datasketch_amd/csrc/_variant_tmp.hip) in which every wave of weighted_walk_wave_kernel adds up the shader-clock cycles
(clock64) it spends waiting for its row, staging + scanning it, listing, walking and refilling, and the rounds its walks
take; the launcher prints the averages per row to stderr after every call.  Build and run:

    python tools/experiments/r04_walk_phase_cycles.py && bash tools/build_variant.sh datasketch_amd/csrc/_variant_tmp.hip prof weighted_kernels
    MHX_LIBRARY=build/variants/libmhx_prof.so python tools/bench_weighted.py --check 0 --reps 1 --variants "kernel=0"

(The instrumented kernel is ~1.3x slower than the product's: read the proportions, not the sum.)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "datasketch_amd", "csrc", "weighted_kernels.hip")).read()


def sub(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)


sub("// walk_row for NC chunks of samples of ONE row at a time", "__device__ unsigned long long mhx_dbg[32];\n// walk_row for NC chunks of samples of ONE row at a time")
# rounds inside walk_chunks
sub("        if (!walking()) break;\n        float l[NC][kU], t[NC][kU], a[NC][kU];", "        if (!walking()) break;\n        if (lane == 0) atomicAdd(&mhx_dbg[8], 1ull);\n        float l[NC][kU], t[NC][kU], a[NC][kU];")
sub("            if (!walking()) break;\n            float l[NC][kG], t[NC][kG], a[NC][kG];", "            if (!walking()) break;\n            if (lane == 0) atomicAdd(&mhx_dbg[9], 1ull);\n            float l[NC][kG], t[NC][kG], a[NC][kG];")
sub("    if (all_listed) return;\n    bool done[NC];", "    if (lane == 0) atomicAdd(&mhx_dbg[10], (unsigned long long)n_list);\n    if (all_listed) return;\n    bool done[NC];")
# phases in the wave kernel
sub("    const auto one_row = [&](float4 (&pre)[NV], int64_t d) {\n        // stage + scan",
    "    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n    const auto one_row = [&](float4 (&pre)[NV], int64_t d) {\n        const unsigned long long t0 = (unsigned long long)clock64();\n"
    "        asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(NV));\n        const unsigned long long t1 = (unsigned long long)clock64();\n        // stage + scan")
sub("        int n_stored = dim, n_out = 0;\n        bool has_nan = false;", "        const unsigned long long t2 = (unsigned long long)clock64();\n        int n_stored = dim, n_out = 0;\n        bool has_nan = false;")
sub("        const bool walked = n_stored > 0 && !has_nan && !(by_entry && !listable);", "        const unsigned long long t3 = (unsigned long long)clock64();\n        const bool walked = n_stored > 0 && !has_nan && !(by_entry && !listable);")
sub("        if (lane == 0) nonempty[d] = n_stored > 0 ? 1 : 0;\n        // the refill goes out behind the walk", "        const unsigned long long t4 = (unsigned long long)clock64();\n        if (lane == 0) nonempty[d] = n_stored > 0 ? 1 : 0;\n        // the refill goes out behind the walk")
sub("        fetch(pre, d + 2 * stride);\n    };\n    float4 pre0[NV], pre1[NV];",
    "        fetch(pre, d + 2 * stride);\n        const unsigned long long t5 = (unsigned long long)clock64();\n"
    "        acc[0] += t1 - t0, acc[1] += t2 - t1, acc[2] += t3 - t2, acc[3] += t4 - t3, acc[4] += t5 - t4, acc[6] += 1;\n    };\n    float4 pre0[NV], pre1[NV];")
sub("        if (d + stride < n_rows) one_row(pre1, d + stride);\n    }\n}\n",
    "        if (d + stride < n_rows) one_row(pre1, d + stride);\n    }\n    if (lane == 0)\n        for (int i = 0; i < 7; ++i) atomicAdd(&mhx_dbg[i], acc[i]);\n}\n")
sub("#undef MHX_WALK_WAVE\n            MHX_HIP_CHECK(hipGetLastError());\n            return MHX_OK;",
    "#undef MHX_WALK_WAVE\n            MHX_HIP_CHECK(hipGetLastError());\n            {\n                unsigned long long h[32];\n"
    "                MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));\n                MHX_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(mhx_dbg), sizeof(h)));\n"
    "                const double nr = (double)(h[6] ? h[6] : 1);\n"
    "                fprintf(stderr, \"per row: wait_row=%.0f stage+scan=%.0f recount+lists=%.0f walk+store=%.0f refill=%.0f cycles; per row: cached rounds=%.2f global rounds=%.2f listed=%.1f (rows=%.0f)\\n\",\n"
    "                        h[0] / nr, h[1] / nr, h[2] / nr, h[3] / nr, h[4] / nr, h[8] / nr, h[9] / nr, h[10] / nr, nr);\n"
    "                unsigned long long z[32] = {0};\n                MHX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mhx_dbg), z, sizeof(z)));\n            }\n            return MHX_OK;")
out = os.path.join(ROOT, "datasketch_amd", "csrc", "_variant_tmp.hip")
open(out, "w").write(s)
print(out)
