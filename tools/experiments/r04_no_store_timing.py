#!/usr/bin/env python3
"""Round 4 experiment (profiles/r04_ab_short_ragged_sets.txt (d), profiles/r04_clock_ramp.txt): how much of the first launch is its stores?

NOT runnable against the committed library: setting 3 of option minhash.prefetch was a two-line hack in minhash_bulk_kernel that existed
only for this measurement --
    first run:  `if (kSieve && args.prefetch == 3) continue;` in front of `out[...] = v` (and no flag byte): no stores at all;
    second run: `__builtin_nontemporal_store(v, &out[...])` under the same condition: streaming stores.
With the committed library both settings run the same kernel (3 counts as "on").  Kept because its FIRST form (no clock warm-up,
the settings alternating 1, 3, 1, 3) is what exposed the GPU's clock ramp: every later group was faster than the one before it,
whatever its setting.  The form below (tools/_warm.py in front of every group) is the one that measured "no difference".
"""
import sys, json, numpy as np
sys.path.insert(0, '.')
from datasketch_amd import MinHash, _native
from tools._warm import warm
ctx = _native.context()
rng = np.random.RandomState(7)
for case, (n, lo, hi) in {"ragged100": (1_000_000, 1, 100), "ragged480": (500_000, 32, 480), "k128": (1_000_000, 256, 256)}.items():
    lens = rng.randint(lo, hi + 1, size=n).astype(np.int64)
    off = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
    hv = rng.randint(0, 2**32, size=int(off[-1]), dtype=np.uint64)
    a, b = MinHash(num_perm=128, seed=1).permutations
    d_hv, d_off, d_out = ctx.to_device(hv), ctx.to_device(off), ctx.alloc(n * 128 * 8)
    for pf in (1, 3, 1, 3):
        ctx.set_option("minhash.prefetch", pf)
        def run():
            ctx.minhash_bulk_dev((a, b), d_hv.ptr, _native.MHX_U64, d_off.ptr, 0, n, hv.size, None, 0, d_out.ptr, _native.MHX_U64)
        run(); ctx.synchronize()
        warm(run, ctx.synchronize, 0.3)
        evs = [ctx.event() for _ in range(21)]
        evs[0].record()
        for i in range(20):
            run(); evs[i + 1].record()
        ctx.synchronize()
        print(case, "prefetch", pf, round(min(evs[i].elapsed_ms(evs[i + 1]) for i in range(20)), 4), round(sum(evs[i].elapsed_ms(evs[i + 1]) for i in range(20)) / 20, 4), flush=True)
    ctx.set_option("minhash.prefetch", 1)
