#!/usr/bin/env python3
"""tools/bench_shapes.py -- MinHash kernel time for signature lengths and set shapes off the headline, under the
settings of minhash.packed (0 auto, 1 one set per wave, 2 kernel C wherever it can run), same box, same corpus.

    python tools/bench_shapes.py [--cases k48,k64,k96,k128,k200,k256,ragged100,ragged480] [--packed 0,1,2]

One JSON line per (case, setting): ms (HIP events), pairs/s; every result is compared with the first setting's.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="k48,k64,k96,k128,k200,k256,ragged100,ragged480")
    ap.add_argument("--packed", default="0,1,2")
    ap.add_argument("--p3", default="0", help="settings of minhash.p3 to run (0 auto: three permutations per lane for 129 .. 192; 1: four)")
    ap.add_argument("--share", default="0", help="settings of minhash.share to run (0 auto: lane groups share a last slot of <= 32 permutations; 1: off)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ramp", type=float, default=0.3, help="seconds of untimed launches in front of the timed ones (GPU clocks)")
    args = ap.parse_args()
    from datasketch_amd import MinHash, _native
    from datasketch_amd.hashfunc import prehashed
    from tools._warm import warm

    ctx = _native.context()
    rng = np.random.RandomState(7)
    for case in args.cases.split(","):
        if case.startswith("k"):
            k, n, lo, hi = int(case[1:]), 1_000_000, 256, 256
        else:
            k, n, lo, hi = 128, (1_000_000 if case == "ragged100" else 500_000), (1 if case == "ragged100" else 32), int(case[6:])
        lens = rng.randint(lo, hi + 1, size=n).astype(np.int64)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        hv = rng.randint(0, 2**32, size=int(off[-1]), dtype=np.uint64)
        a, b = MinHash(num_perm=k, seed=1).permutations
        d_hv, d_off, d_out = ctx.to_device(hv), ctx.to_device(off), ctx.alloc(n * k * 8)
        dense = lo == hi
        first = None
        for packed, p3, share in [(int(x), int(y), int(z)) for x in args.packed.split(",") for y in args.p3.split(",") for z in args.share.split(",")]:
            ctx.set_option("minhash.packed", packed)
            ctx.set_option("minhash.p3", p3)
            ctx.set_option("minhash.share", share)

            def run():
                ctx.minhash_bulk_dev((a, b), d_hv.ptr, _native.MHX_U64, None if dense else d_off.ptr, lo if dense else 0, n, hv.size, None, 0,
                                     d_out.ptr, _native.MHX_U64)

            run()
            ctx.synchronize()
            warm(run, ctx.synchronize, args.ramp)  # GPU clocks (tools/_warm.py)
            evs = [ctx.event() for _ in range(args.reps + 1)]
            evs[0].record()
            for i in range(args.reps):
                run()
                evs[i + 1].record()
            ctx.synchronize()
            ms = min(evs[i].elapsed_ms(evs[i + 1]) for i in range(args.reps))
            got = d_out.download((n, k), np.uint64)
            rec = {"case": case, "num_perm": k, "sets": n, "tokens": int(hv.size), "minhash.packed": packed, "minhash.p3": p3, "minhash.share": share, "ms": round(ms, 4),
                   "pairs_per_s": hv.size * k / (ms * 1e-3)}
            if first is None:
                first = got
                want = MinHash.bulk_signatures((hv[: off[512]], off[:513]), num_perm=k, seed=1, hashfunc=prehashed, gpu_mode="disable")
                rec["first_512_rows_equal_numpy"] = bool(np.array_equal(got[:512], want))
            else:
                rec["equal_to_first"] = bool(np.array_equal(got, first))
            print(json.dumps(rec), flush=True)
        ctx.set_option("minhash.packed", 0)
        ctx.set_option("minhash.p3", 0)
        ctx.set_option("minhash.share", 0)
        for d in (d_hv, d_off, d_out):
            d.free()


if __name__ == "__main__":
    main()
