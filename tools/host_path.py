#!/usr/bin/env python3
"""tools/host_path.py -- PCIe-inclusive rate of the host entry point (numpy in -> numpy out) at the
headline configuration, whole-call vs pipelined in pieces (option host.chunk_bytes)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from datasketch_amd import _native  # noqa: E402
from datasketch_amd import MinHash  # noqa: E402

n, t, k = 1_000_000, 256, 128
tok = np.random.RandomState(42).randint(0, 2**32, (n, t), dtype=np.uint64).reshape(-1)
perms = MinHash(num_perm=k, seed=1).permutations
ctx = _native.context()
for label, opt in (("whole call", -1), ("pieces 96 MiB", 0), ("pieces 32 MiB", 32 << 20), ("pieces 256 MiB", 256 << 20)):
    ctx.set_option("host.chunk_bytes", opt)
    best = None
    out = None
    for rep in range(5):
        # rep 0 fills a fresh array (page faults included), later reps reuse it
        t0 = time.perf_counter()
        out = ctx.minhash_bulk(perms, tok, None, t, n, out=out)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        print(f"{label:16s} rep {rep}: {dt*1e3:7.1f} ms  {n/dt/1e6:6.2f} M signatures/s", flush=True)
    print(f"{label:16s} best : {best*1e3:7.1f} ms  {n/best/1e6:6.2f} M signatures/s  checksum {int(out.sum()) & 0xFFFFFFFF:#x}", flush=True)
