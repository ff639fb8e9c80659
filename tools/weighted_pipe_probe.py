#!/usr/bin/env python3
"""tools/weighted_pipe_probe.py -- config 4 from Python in parity mode (np.log on the host, pipelined with the
device): wall time against the piece size and the number of log threads.  One JSON line per setting.

    python tools/weighted_pipe_probe.py [--rows 100000] [--dim 4096] [--samples 128]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datasketch_amd import WeightedMinHashGenerator  # noqa: E402


def parts(g, x):
    """Where the time of the pipelined call goes: the logs alone (thread pool, ring of buffers), the device calls alone
    (pieces of ready logs), one whole-matrix log + one device call."""
    from concurrent.futures import ThreadPoolExecutor

    n, dim = x.shape
    ctx, handle = g._device_handle()
    s = g.sample_size
    for threads, mib in ((8, 32), (8, 64), (16, 64)):
        rows = (mib << 20) // (4 * dim)
        starts = list(range(0, n, rows))
        ring = [np.empty((rows, dim), dtype=np.float32) for _ in range(threads + 2)]

        def take_log(i, lo):
            hi = min(n, lo + rows)
            np.log(x[lo:hi], out=ring[i % len(ring)][: hi - lo])

        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda a: take_log(*a), enumerate(starts)))
        t_log = time.perf_counter() - t0
        out = np.zeros((n, s, 2), dtype=np.int64)
        ne = np.zeros(n, dtype=np.uint8)
        t0 = time.perf_counter()
        for lo in starts:
            hi = min(n, lo + rows)
            ctx.weighted_minhash_many_dense(handle, s, ring[0][: hi - lo], True, out=out[lo:hi], nonempty=ne[lo:hi])
        t_dev_fresh = time.perf_counter() - t0
        t0 = time.perf_counter()
        for lo in starts:
            hi = min(n, lo + rows)
            ctx.weighted_minhash_many_dense(handle, s, ring[0][: hi - lo], True, out=out[lo:hi], nonempty=ne[lo:hi])
        t_dev = time.perf_counter() - t0
        print(json.dumps({"part": "pieces", "log_threads": threads, "piece_MiB": mib, "logs_only_s": round(t_log, 4),
                          "device_calls_only_fresh_out_s": round(t_dev_fresh, 4), "device_calls_only_s": round(t_dev, 4)}), flush=True)
    t0 = time.perf_counter()
    logs = np.log(x)
    t_log1 = time.perf_counter() - t0
    res = {"part": "whole", "np_log_one_core_s": round(t_log1, 4)}
    out = np.zeros((n, s, 2), dtype=np.int64)
    ne = np.zeros(n, dtype=np.uint8)
    for name, data, is_logs in (("logs", logs, True), ("values_device_log", x, False)):
        for label, opt in (("pieces", 0), ("one_shot", -1)):
            ctx.set_option("host.chunk_bytes", opt)
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                ctx.weighted_minhash_many_dense(handle, s, data, is_logs, out=out, nonempty=ne)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res[f"one_call_{name}_{label}_s"] = round(best, 4)
    ctx.set_option("host.chunk_bytes", 0)
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--settings", default="8:32,12:32,16:32,8:64,12:64,16:16,12:16,24:32")
    a = ap.parse_args()
    x = np.random.RandomState(42).uniform(0, 100, (a.rows, a.dim)).astype(np.float32)
    g = WeightedMinHashGenerator(a.dim, a.samples, seed=1, gpu_mode="always")
    first = g.minhash_many_arrays(x[:4096])  # warm-up: tables on the device, scratch sized
    parts(g, x)
    ref = None
    for setting in a.settings.split(","):
        threads, mib = map(int, setting.split(":"))
        type(g)._PIPE_LOG_THREADS, type(g)._PIPE_PIECE_BYTES = threads, mib << 20
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            out, ok = g.minhash_many_arrays(x)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if ref is None:
            ref = out
            assert np.array_equal(out[:4096], first[0])
        else:
            assert np.array_equal(out, ref)
        print(json.dumps({"log_threads": threads, "piece_MiB": mib, "seconds_best_of_3": round(best, 4),
                          "vectors_per_s": a.rows / best, "usable_cpus": len(os.sched_getaffinity(0))}), flush=True)


if __name__ == "__main__":
    main()
