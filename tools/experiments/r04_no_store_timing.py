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
