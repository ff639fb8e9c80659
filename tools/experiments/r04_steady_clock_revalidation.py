#!/usr/bin/env python3
"""Round 4: the tuning choices of rounds 1-3 were made on launches timed inside the GPU's clock ramp (profiles/r04_clock_ramp.txt),
with a bias towards whichever setting ran second.  This re-measures the ones that decide a default, at steady clocks
(tools/_warm.py), every setting twice and interleaved.  Output: one line per (case, setting)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datasketch_amd import MinHash, _native  # noqa: E402
from tools._warm import warm  # noqa: E402


def timed(ctx, run, reps=20):
    run()
    ctx.synchronize()
    warm(run, ctx.synchronize, 0.3)
    evs = [ctx.event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        run()
        evs[i + 1].record()
    ctx.synchronize()
    ms = [evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]
    return min(ms), float(np.mean(ms))


def main():
    ctx = _native.context()
    rng = np.random.RandomState(7)
    cases = {"k128_dense": (1_000_000, 256, 256, 128), "ragged100": (1_000_000, 1, 100, 128), "k256_dense": (1_000_000, 256, 256, 256),
             "k192_dense": (1_000_000, 256, 256, 192), "t64_dense": (2_000_000, 64, 64, 128), "t128_dense": (1_000_000, 128, 128, 128),
             "t512_dense": (500_000, 512, 512, 128), "t100_dense": (1_000_000, 100, 100, 128), "k64_dense": (1_000_000, 256, 256, 64), "ragged480": (500_000, 32, 480, 128)}
    if len(sys.argv) > 1:
        cases = {c: cases[c] for c in sys.argv[1].split(",")}
    sweeps = (("blocks_per_cu", (64, 32, 128, 16, 64, 32, 128, 16)), ("minhash.prefetch", (1, 0, 1, 0)), ("minhash.adapt", (0, 1, 0, 1)))
    if len(sys.argv) > 2:
        key, _, vals = sys.argv[2].partition("=")
        sweeps = ((key if vals else "blocks_per_cu", tuple(int(x) for x in (vals or key).split(","))),)
    for case, (n, lo, hi, k) in cases.items():
        lens = rng.randint(lo, hi + 1, size=n).astype(np.int64)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        hv = rng.randint(0, 2**32, size=int(off[-1]), dtype=np.uint64)
        a, b = MinHash(num_perm=k, seed=1).permutations
        d_hv, d_off, d_out = ctx.to_device(hv), ctx.to_device(off), ctx.alloc(n * k * 8)
        dense = lo == hi

        def run():
            ctx.minhash_bulk_dev((a, b), d_hv.ptr, _native.MHX_U64, None if dense else d_off.ptr, lo if dense else 0, n, hv.size, None, 0,
                                 d_out.ptr, _native.MHX_U64)

        for key, values in sweeps:
            for v in values:
                ctx.set_option(key, v)
                best, mean = timed(ctx, run)
                print(f"{case:12s} {key}={v:<4d} best {best:.4f} mean {mean:.4f} ms", flush=True)
            ctx.set_option(key, 0 if key != "minhash.prefetch" else 1)
        for d in (d_hv, d_off, d_out):
            d.free()


if __name__ == "__main__":
    main()
