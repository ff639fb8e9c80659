#!/usr/bin/env python3
"""tools/bench_sort.py -- mhx_lsh_sort_bands on config 3's per-GPU shard (1.25M x 256 uint32 signatures, 32 bands x 8),
the two-pass bucketing (lsh.sort = 0) against the radix sort (lsh.sort = 1), same box, same matrix; results compared."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._warm import warm  # noqa: E402


def main():
    from datasketch_amd import _native

    ctx = _native.context()
    n, k, b, r = int(os.environ.get("N", 1_250_000)), 256, 32, 8
    rng = np.random.RandomState(3)
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64).astype(np.uint32)
    d_sig = ctx.to_device(sig)
    d_dig, d_rows = ctx.alloc(n * b * 8), ctx.alloc(n * b * 4)
    first = None
    for mode, prehash in ((0, 0), (0, 1), (1, 0), (0, 0), (0, 1)):
        ctx.set_option("lsh.sort", mode)
        ctx.set_option("lsh.prehash", prehash)
        run = lambda: _native.check(ctx.lib.mhx_lsh_sort_bands_dev_typed(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, b, r, d_dig.ptr, d_rows.ptr))
        run()
        ctx.synchronize()
        warm(run, ctx.synchronize, 0.3)  # GPU clocks (tools/_warm.py)
        ms = []
        for _ in range(5):
            e0, e1 = ctx.event(), ctx.event()
            e0.record()
            run()
            e1.record()
            ctx.synchronize()
            ms.append(e0.elapsed_ms(e1))
        dig, rows = d_dig.download((b, n), np.uint64), d_rows.download((b, n), np.uint32)
        rec = {"lsh.sort": mode, "lsh.prehash": prehash, "ms_min": round(min(ms), 4), "ms": [round(x, 4) for x in ms], "keys_per_s": n * b / (min(ms) * 1e-3)}
        if first is None:
            first = (dig, rows)
            rec["sorted"] = bool((dig[:, 1:] >= dig[:, :-1]).all())  # uint64 compared as uint64 (an int64 view turns digests >= 2^63 negative)
        else:
            rec["equal_to_first"] = bool(np.array_equal(dig, first[0]) and np.array_equal(rows, first[1]))
        print(json.dumps(rec), flush=True)
    ctx.set_option("lsh.sort", 0)
    ctx.set_option("lsh.prehash", 0)


if __name__ == "__main__":
    main()
