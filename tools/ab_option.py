#!/usr/bin/env python3
"""tools/ab_option.py <option> <v0,v1,..> -- the band-major bucketing (mhx_lsh_sort_digests_layout_dev) of N x 32 digests timed under the values of
one context option, interleaved on one box, results compared (round 6: lsh.ahead).  N from the environment (default 1.25M)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._warm import warm  # noqa: E402

from datasketch_amd import _native  # noqa: E402

opt, values = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
ctx = _native.Context(0)
lib = ctx.lib
n, bands = int(os.environ.get("N", 1_250_000)), 32
dig = np.random.RandomState(5).randint(0, 2**63, (bands, n), dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
d_dig = ctx.to_device(dig)
d_sd, d_sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
sort_bm = lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig.ptr, n, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr))


def timed(fn, reps=6):
    evs = [ctx.event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    ctx.synchronize()
    return [round(evs[i].elapsed_ms(evs[i + 1]), 4) for i in range(reps)]


warm(sort_bm, ctx.synchronize, 0.4)
ref = None
out = {"n": n, "option": opt}
for rnd in range(3):
    for v in values:
        ctx.set_option(opt, v)
        sort_bm()
        ctx.synchronize()
        out.setdefault(f"{opt}={v}", []).append(min(timed(sort_bm)))
        if rnd == 0:
            got = (d_sd.download((bands, n), np.uint64), d_sr.download((bands, n), np.uint32))
            if ref is None:
                ref = got
                srt = np.sort(dig, axis=1)
                out["sorted_equal_numpy"] = bool(np.array_equal(got[0], srt))
            else:
                out[f"{opt}={v}_equal_first"] = bool(np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]))
print(json.dumps(out))
