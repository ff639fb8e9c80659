#!/usr/bin/env python3
"""tools/traffic_probe.py -- the launches whose HBM traffic rocprofv3's TCC counters are read for
(tools/traffic.sh wraps it in the --pmc passes; tools/traffic_summary.py turns the databases into
profiles/r02_traffic_minhash_bulk.json).

  1. the headline launch (1M sets x 256 tokens, K=128, uint64 in / out), 3 times;
  2. calibration with a known byte count: minhash_merge_kernel over two 1.024 GB matrices (reads 2.048 GB with
     16 B per lane, writes 1.024 GB), 3 times;
  3. the headline launch with option minhash.alias = 4095: every set reads the tokens of set (i & 4095), an 8 MB
     working set that stays in L2 / Infinity Cache -- the same requests, but none of the token reads reaches HBM.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datasketch_amd import MinHash, _native  # noqa: E402

ctx = _native.Context(0)
n, t, k = 1_000_000, 256, 128
perms = MinHash(num_perm=k, seed=1).permutations
tok = np.random.RandomState(42).randint(0, 2**32, (n, t), dtype=np.uint64)
d_tok, d_out = ctx.to_device(tok), ctx.alloc(n * k * 8)
run = lambda: ctx.minhash_bulk_dev(perms, d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_out.ptr, _native.MHX_U64)
for _ in range(4):
    run()
ctx.synchronize()
d_x, d_y, d_z = ctx.alloc(n * k * 8), ctx.alloc(n * k * 8), ctx.alloc(n * k * 8)
ctx.copy_dev(d_x.ptr, d_out.ptr, n * k * 8)
ctx.copy_dev(d_y.ptr, d_out.ptr, n * k * 8)
for _ in range(3):
    _native.check(ctx.lib.mhx_minhash_merge_dev(ctx.handle, d_x.ptr, d_y.ptr, n * k, d_z.ptr))
ctx.synchronize()
ctx.set_option("minhash.alias", 4095)
for _ in range(3):
    run()
ctx.synchronize()
print("done")
