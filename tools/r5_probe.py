#!/usr/bin/env python3
"""tools/r5_probe.py -- the launches round 5's counter and trace passes look at (tools/r5_passes.sh wraps it in rocprofv3):
config 3 / 5's per-GPU shapes (1.25M x 256 uint32 signatures, 32 bands x 8), each launch a few times, nothing checked here
(tests/test_gpu_round5.py does that):

  calibration   minhash_merge_kernel over two 1.024 GB matrices (reads 2.048 GB, writes 1.024 GB)
  sort          mhx_lsh_sort_digests_dev: lsh_bin_scatter_kernel<Digest64 / Digest64BM> + lsh_bin_sort_kernel (row-major and band-major digest input)
  c5            bbit_digest_fused_kernel against bbit1_wide_kernel + band_digest_kernel

With no profiler around it prints HIP-event times of the same launches (interleaved A/B, after a clock warm-up)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._warm import warm  # noqa: E402

from datasketch_amd import _native  # noqa: E402

ctx = _native.Context(0)
lib = ctx.lib
n, k, bands, r = int(os.environ.get("N", 1_250_000)), 256, 32, 8
reps = int(os.environ.get("REPS", 3))
sig = np.random.RandomState(3).randint(0, 2**32, (n, k), dtype=np.uint64).astype(np.uint32)
d_sig = ctx.to_device(sig)
d_dig, d_blk = ctx.alloc(n * bands * 8), ctx.alloc(n * (k // 64) * 8)
d_sd, d_sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
d_x, d_y, d_z = ctx.alloc(128_000_000 * 8), ctx.alloc(128_000_000 * 8), ctx.alloc(128_000_000 * 8)


def timed(fn, reps=5):
    evs = [ctx.event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    ctx.synchronize()
    return [round(evs[i].elapsed_ms(evs[i + 1]), 4) for i in range(reps)]


digests = lambda: _native.check(lib.mhx_band_digests_dev_typed(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, bands, r, d_dig.ptr))
pack = lambda: _native.check(lib.mhx_bbit_pack_dev_typed(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, 1, d_blk.ptr))
fused = lambda: ctx.bbit_pack_band_digests_dev(d_sig.ptr, _native.MHX_U32, n, k, 1, bands, r, d_blk.ptr, d_dig.ptr)
sort = lambda: _native.check(lib.mhx_lsh_sort_digests_dev(ctx.handle, d_dig.ptr, n, bands, d_sd.ptr, d_sr.ptr))
d_dig_bm = ctx.alloc(n * bands * 8)
digests_bm = lambda: _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, bands, r, _native.BAND_MAJOR, d_dig_bm.ptr))
fused_bm = lambda: ctx.bbit_pack_band_digests_dev(d_sig.ptr, _native.MHX_U32, n, k, 1, bands, r, d_blk.ptr, d_dig_bm.ptr, _native.BAND_MAJOR)
sort_bm = lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig_bm.ptr, n, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr))
merge = lambda: _native.check(lib.mhx_minhash_merge_dev(ctx.handle, d_x.ptr, d_y.ptr, 128_000_000, d_z.ptr))

digests()
digests_bm()
ctx.synchronize()
profiled = bool(os.environ.get("ROCPROFILER_REGISTER_ROOT") or os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("R5_PROFILED"))
if not profiled:
    warm(fused, ctx.synchronize, 0.4)
for _ in range(reps):
    merge()
for _ in range(reps):
    sort()
for _ in range(reps):
    sort_bm()
for _ in range(reps):
    pack()
    digests()
    fused()
    digests_bm()
    fused_bm()
ctx.synchronize()
if not profiled:
    out = {"n": n}
    for rnd in range(2):
        out.setdefault("sort_digests_row_major_ms", []).extend(timed(sort, 3))
        out.setdefault("sort_digests_band_major_ms", []).extend(timed(sort_bm, 3))
        ctx.set_option("lsh.chunk", 8)
        out.setdefault("sort_digests_band_major_chunk8_ms", []).extend(timed(sort_bm, 3))
        ctx.set_option("lsh.chunk", 0)
        out.setdefault("digests_band_major_ms", []).extend(timed(digests_bm, 3))
        out.setdefault("fused_band_major_ms", []).extend(timed(fused_bm, 3))
        out.setdefault("pack_ms", []).extend(timed(pack, 3))
        out.setdefault("digests_ms", []).extend(timed(digests, 3))
        out.setdefault("fused_ms", []).extend(timed(fused, 3))
    for key in list(out):
        if key.endswith("_ms"):
            out[key[:-3] + "_min_ms"] = min(out[key])
    print(json.dumps(out), flush=True)
print("done")
