#!/usr/bin/env python3
"""tools/bench_weighted.py -- config 4 (dense weighted rows) kernel timings under option sets, on one box.

    python tools/bench_weighted.py [--rows 100000] [--dim 4096] [--samples 128] [--check 2048]
                                   [--variants "path=0;path=2"] [--density 1.0]

Every variant runs mhx_weighted_minhash_many_dense_dev on the same resident logs (HIP events on the context's
stream) and is compared with the first variant's result; `--check` rows are compared with the C oracle (tests'
checker; never timed).  One JSON line per variant.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools._warm import warm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--check", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ramp", type=float, default=0.3, help="seconds of untimed calls in front of the timed ones (GPU clocks)")
    ap.add_argument("--density", type=float, default=1.0, help="fraction of stored entries (the rest are zeros)")
    ap.add_argument("--dist", default="uniform", help="uniform | lognormal | sorted (columns by increasing weight)")
    ap.add_argument("--variants", default="path=0;path=2")
    ap.add_argument("--values", action="store_true", help="hand over the values, not their logs: the device takes the log (np_logf)")
    ap.add_argument("--csr", action="store_true", help="hand the rows over as CSR (mhx_weighted_minhash_many_dev) instead of dense")
    args = ap.parse_args()

    from datasketch_amd import WeightedMinHashGenerator, _native

    n, dim, s = args.rows, args.dim, args.samples
    rs = np.random.RandomState(42)
    x = np.empty((n, dim), dtype=np.float32)
    for i in range(0, n, 10_000):
        m = min(10_000, n - i)
        if args.dist == "lognormal":
            x[i:i + m] = rs.lognormal(0.0, 2.0, (m, dim))
        else:
            x[i:i + m] = rs.uniform(0, 100, (m, dim))
    if args.dist == "sorted":
        x.sort(axis=1)
    if args.density < 1.0:
        for i in range(0, n, 10_000):
            m = min(10_000, n - i)
            x[i:i + m][rs.random_sample((m, dim)) >= args.density] = 0
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    ctx, handle = g._device_handle()
    lib = ctx.lib
    with np.errstate(invalid="ignore", divide="ignore"):
        logs = np.log(x)
    d_o = ctx.alloc(n * s * 16)
    d_ne = ctx.alloc(n)
    if args.csr:
        import scipy.sparse as sp

        csr = sp.csr_matrix(x)
        csr.sort_indices()
        with np.errstate(invalid="ignore", divide="ignore"):
            d_ptr, d_idx = ctx.to_device(csr.indptr.astype(np.int64)), ctx.to_device(csr.indices.astype(np.int32))
            d_val = ctx.to_device(np.log(csr.data).astype(np.float32))
        nnz = int(csr.nnz)
    else:
        d_x = ctx.to_device(x if args.values else logs)
    first = None
    for variant in args.variants.split(";"):
        opts = dict(kv.split("=") for kv in variant.split(",") if kv)
        ctx.set_option("weighted.path", int(opts.get("path", 0)))
        ctx.set_option("blocks_per_cu", int(opts.get("bpc", 0)))
        ctx.set_option("weighted.direct", int(opts.get("direct", 0)))
        ctx.set_option("weighted.debug", int(opts.get("debug", 0)))
        ctx.set_option("weighted.split", int(opts.get("split", 0)))
        ctx.set_option("weighted.tail", int(opts.get("tail", 0)))
        ctx.set_option("weighted.kernel", int(opts.get("kernel", 0)))
        ctx.set_option("weighted.plan", int(opts.get("plan", 0)))
        ctx.set_option("weighted.rescue", int(opts.get("rescue", 0)))
        ctx.set_option("weighted.refill", int(opts.get("refill", 0)))
        ctx.set_option("weighted.min_dim", int(opts.get("min_dim", 0)))

        def call():
            if args.csr:
                _native.check(lib.mhx_weighted_minhash_many_dev(handle, d_ptr.ptr, d_idx.ptr, d_val.ptr, 1, n, nnz, d_o.ptr, d_ne.ptr))
            else:
                _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 0 if args.values else 1, n, d_o.ptr, d_ne.ptr))

        call()
        ctx.synchronize()
        warm(call, ctx.synchronize, args.ramp)  # GPU clocks (tools/_warm.py)
        times = []
        for _ in range(args.reps):
            e0, e1 = ctx.event(), ctx.event()
            e0.record()
            call()
            e1.record()
            ctx.synchronize()
            times.append(e0.elapsed_ms(e1))
        hv = d_o.download((n, s, 2), np.int64)
        ne = d_ne.download((n,), np.uint8)
        rec = {"variant": variant, "ms": [round(t, 3) for t in times], "ms_min": round(min(times), 3),
               "element_evaluations_per_s": n * dim * s * args.density / (min(times) * 1e-3)}
        if first is None:
            first = (hv, ne)
            if args.check:
                import scipy.sparse as sp

                from oracle import oracle as O

                rows = np.unique(np.linspace(0, n - 1, args.check).astype(np.int64))
                t0 = time.perf_counter()
                csr = sp.csr_matrix(x[rows])
                csr.sort_indices()
                wo, wn = O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)
                rec["oracle_rows"] = int(len(rows))
                rec["oracle_seconds"] = round(time.perf_counter() - t0, 2)
                rec["oracle_equal"] = bool(np.array_equal(hv[rows], wo) and np.array_equal(ne[rows], wn))
        else:
            rec["equal_to_first"] = bool(np.array_equal(hv, first[0]) and np.array_equal(ne, first[1]))
        print(json.dumps(rec), flush=True)
    ctx.set_option("weighted.path", 0)
    ctx.set_option("blocks_per_cu", 0)
    ctx.set_option("weighted.direct", 0)
    ctx.set_option("weighted.debug", 0)
    ctx.set_option("weighted.kernel", 0)
    ctx.set_option("weighted.refill", 0)
    ctx.set_option("weighted.plan", 0)
    ctx.set_option("weighted.rescue", 0)


if __name__ == "__main__":
    main()
