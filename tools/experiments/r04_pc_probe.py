#!/usr/bin/env python3
"""Launches for a PC-sampling run (tools/experiments/r04_pc_sampling.sh): CASE=ragged100 (10^6 sets of 1..100 tokens, K=128, CSR)
or CASE=c4 (config 4's dense weighted rows), the same launch REPS times at steady clocks."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datasketch_amd import MinHash, WeightedMinHashGenerator, _native  # noqa: E402

case, reps = os.environ.get("CASE", "ragged100"), int(os.environ.get("REPS", "200"))
rng = np.random.RandomState(7)
if case == "c4":
    n, dim, s = 100_000, 4096, 128
    x = rng.uniform(0, 100, (n, dim)).astype(np.float32)
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    ctx, handle = g._device_handle()
    d_x, d_o, d_ne = ctx.to_device(np.log(x)), ctx.alloc(n * s * 16), ctx.alloc(n)
    run = lambda: _native.check(ctx.lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 1, n, d_o.ptr, d_ne.ptr))
else:
    ctx = _native.context()
    n, lo, hi, k = (1_000_000, 1, 100, 128) if case == "ragged100" else (1_000_000, 256, 256, 128)
    lens = rng.randint(lo, hi + 1, size=n).astype(np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    hv = rng.randint(0, 2**32, size=int(off[-1]), dtype=np.uint64)
    a, b = MinHash(num_perm=k, seed=1).permutations
    d_hv, d_off, d_out = ctx.to_device(hv), ctx.to_device(off), ctx.alloc(n * k * 8)
    run = lambda: ctx.minhash_bulk_dev((a, b), d_hv.ptr, _native.MHX_U64, d_off.ptr, 0, n, hv.size, None, 0, d_out.ptr, _native.MHX_U64)
for _ in range(reps):
    run()
ctx.synchronize()
print("done", case, reps)
