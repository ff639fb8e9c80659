#!/usr/bin/env python3
"""tools/cpu_reference_vs_port.py -- the REAL reference (`/root/reference`, build container only) timed beside the
numpy restatement that bench.py's `cpu_baseline` times (oracle/oracle.py:np_minhash_bulk), same process, same sample,
same core: how much of the reference's CPU time is the Python object churn of `MinHash.bulk` (one `copy()` per set)
that the restatement does not have.  Rows are compared bit for bit."""
import sys
import time

import numpy as np

sys.path.insert(0, "/root/reference")
sys.path.insert(0, ".")
import datasketch as ref  # noqa: E402
from oracle import oracle as O  # noqa: E402


def identity(x):
    return x


n, t, k = 20_000, 256, 128
tok = np.random.RandomState(42).randint(0, 2**32, (n, t), dtype=np.uint64)
a, b = O.np_init_permutations(k, 1)
best_ref = best_port = None
for rep in range(3):
    t0 = time.perf_counter()
    objs = ref.MinHash.bulk(tok, num_perm=k, seed=1, hashfunc=identity)
    dt = time.perf_counter() - t0
    best_ref = dt if best_ref is None else min(best_ref, dt)
    t0 = time.perf_counter()
    sig = O.np_minhash_bulk(list(tok), a, b)
    dt = time.perf_counter() - t0
    best_port = dt if best_port is None else min(best_port, dt)
got = np.stack([m.hashvalues for m in objs])
assert np.array_equal(got, sig)
print(f"sample: {n} sets x {t} tokens, num_perm={k}, identity hashfunc, one core, best of 3")
print(f"reference MinHash.bulk        : {best_ref:.2f} s  {n / best_ref:9.0f} signatures/s")
print(f"restatement np_minhash_bulk   : {best_port:.2f} s  {n / best_port:9.0f} signatures/s")
print(f"reference / restatement time  : {best_ref / best_port:.3f}  (rows equal bit for bit)")
