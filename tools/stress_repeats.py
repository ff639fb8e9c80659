#!/usr/bin/env python3
"""tools/stress_repeats.py -- GPU signatures against the C oracle on corpora with repeated tokens at several rates, ragged
lengths, several num_perm, narrow and wide tokens, with and without the tie-tolerant proof.  Prints one line per case;
exit status 1 on the first mismatch.  (A one-off confidence run for the second launch's proofs; the same generator as the
repeats corpora of tools/bench_extra.py, plus clustered repeats and tokens repeated many times.)"""
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datasketch_amd import _native  # noqa: E402
from oracle import oracle as O  # noqa: E402


def corpus(rng, n, lo, hi, rate, wide, cluster):
    lens = rng.randint(lo, hi + 1, size=n).astype(np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    hv = rng.randint(0, 2**32, size=total, dtype=np.uint64)
    if wide:
        w = rng.rand(total) < 0.02
        hv[w] |= rng.randint(1, 2**32, size=int(w.sum()), dtype=np.uint64) << np.uint64(32)
    m = int(rate * total)
    if m:
        dst = rng.randint(0, total, size=m)
        owner = np.searchsorted(off, dst, side="right") - 1
        span = lens[owner]
        if cluster:  # copies of a token next to it (same row mostly)
            src = np.minimum(off[owner] + span - 1, np.maximum(off[owner], dst - rng.randint(1, 4, size=m)))
        else:
            src = off[owner] + (rng.rand(m) * span).astype(np.int64)
        hv[dst] = hv[src]
    return hv, off


def main():
    ctx = _native.context()
    rng = np.random.RandomState(int(os.environ.get("STRESS_SEED", "1")))
    bad = 0
    t0 = time.time()
    for k in (64, 128, 200, 256):
        a, b = O.np_init_permutations(k, 3)
        for rate in (0.002, 0.01, 0.05, 0.1, 0.3, 0.7):
            for (lo, hi) in ((1, 100), (16, 300), (256, 256), (250, 700)):
                for wide in (False, True):
                    cluster = bool(rng.randint(0, 2))
                    n = 6000
                    hv, off = corpus(rng, n, lo, hi, rate, wide, cluster)
                    want = O.c_minhash_bulk(hv, off, a, b)
                    # every setting twice: the second call's first launch is the one the first call's counters chose (round 4:
                    # the tie-tolerant kernel when a quarter of the sets defeated the one-candidate proof), and adapt = 1 pins the old one
                    for ties, adapt in ((0, 0), (0, 0), (1, 0), (1, 0), (0, 1)):
                        ctx.set_option("minhash.ties", ties)
                        ctx.set_option("minhash.adapt", adapt)
                        got = ctx.minhash_bulk((a, b), hv, off, 0, n)
                        rows = np.flatnonzero((got != want).any(axis=1))
                        if rows.size:
                            bad += 1
                            print(f"MISMATCH k={k} rate={rate} len={lo}..{hi} wide={wide} cluster={cluster} ties_off={ties} adapt_off={adapt}: rows {rows[:8]}", flush=True)
                    ctx.set_option("minhash.ties", 0)
                    ctx.set_option("minhash.adapt", 0)
        print(f"k={k} done, {time.time() - t0:.0f} s, mismatching cases so far: {bad}", flush=True)
    print("cases with mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
