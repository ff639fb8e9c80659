#!/usr/bin/env python3
"""tools/bench_extra.py -- device-resident timings of the non-headline kernels (DESIGN.md numbers).

    python tools/bench_extra.py [--weighted-rows 20000] [--sigs 1000000]

Prints one JSON object per measurement: kernel-only time from HIP events on the context's stream,
algorithmic bytes and the implied GB/s.  Every result is first checked
against the package's numpy path (gpu_mode="disable") on a sample.  Not the driver's bench (that is bench.py).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from datasketch_amd import MinHash, WeightedMinHashGenerator, _native, prehashed  # noqa: E402
from datasketch_amd import lsh_bulk as LB  # noqa: E402
from tools._warm import warm  # noqa: E402
from datasketch_amd.b_bit_minhash import pack_matrix  # noqa: E402


def timed(ctx, fn, reps=5, warmup=1):
    for _ in range(warmup):
        fn()
    ctx.synchronize()
    warm(fn, ctx.synchronize, 0.25)  # GPU clocks (tools/_warm.py)
    evs = [ctx.event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    ctx.synchronize()
    return float(np.median([evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]))


def report(name, ms, units, unit_name, alg_bytes, **extra):
    out = {"name": name, "ms": round(ms, 4), f"{unit_name}_per_s": units / (ms * 1e-3), "algorithmic_GBps": alg_bytes / (ms * 1e-3) / 1e9,
           "hbm_frac_of_8TBps": alg_bytes / (ms * 1e-3) / 8e12}
    out.update(extra)
    print(json.dumps(out), flush=True)


def weighted(ctx, n_rows, dim, s, density):
    rng = np.random.RandomState(42)
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    x = rng.uniform(0, 100, (n_rows, dim)).astype(np.float32)
    if density < 1.0:
        x[rng.random_sample(x.shape) >= density] = 0
    import scipy.sparse as sp

    csr = sp.csr_matrix(x)
    csr.sort_indices()
    indptr, indices = csr.indptr.astype(np.int64), csr.indices.astype(np.int32)
    logs = np.log(csr.data)
    _, handle = g._device_handle()
    d_ptr, d_idx, d_val = ctx.to_device(indptr), ctx.to_device(indices), ctx.to_device(logs)
    d_out, d_ne = ctx.alloc(n_rows * s * 16), ctx.alloc(n_rows)
    lib = ctx.lib

    def run():
        _native.check(lib.mhx_weighted_minhash_many_dev(handle, d_ptr.ptr, d_idx.ptr, d_val.ptr, 1, n_rows, indices.size, d_out.ptr, d_ne.ptr))

    ms = timed(ctx, run, reps=3)
    got = d_out.download((n_rows, s, 2), np.int64)
    chk = min(n_rows, 64)
    host = WeightedMinHashGenerator(dim, s, seed=g.seed, gpu_mode="disable")  # the package's numpy path = the reference's arithmetic
    want, _ = host._minhash_many_host(indptr[: chk + 1], indices[: indptr[chk]], csr.data[: indptr[chk]])
    assert np.array_equal(got[:chk], want), "weighted parity failure"
    nnz = int(indices.size)
    report(f"weighted_minhash_many dim={dim} S={s} density={density}", ms, n_rows, "vectors", 4 * nnz + 16 * s * n_rows,
           rows=n_rows, nnz=nnz, evals_per_s=nnz * s / (ms * 1e-3))


def packing(ctx, n, k):
    rng = np.random.RandomState(1)
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    d_sig = ctx.to_device(sig)
    lib = ctx.lib
    for b in (1, 4, 16):
        nb = ctypes.c_int32(0)
        _native.check(lib.mhx_bbit_num_blocks(k, b, ctypes.byref(nb)))
        d_out = ctx.alloc(n * nb.value * 8)
        ms = timed(ctx, lambda: _native.check(lib.mhx_bbit_pack_dev(ctx.handle, d_sig.ptr, n, k, b, d_out.ptr)))
        got = d_out.download((n, nb.value), np.uint64)
        assert np.array_equal(got[:512], pack_matrix(sig[:512], b, gpu_mode="disable"))
        report(f"bbit_pack b={b} K={k}", ms, n, "signatures", n * (8 * k + 8 * nb.value))
    for bands, r in ((32, 8), (k // 4, 4)):
        d_out = ctx.alloc(n * bands * r * 8)
        ms = timed(ctx, lambda: _native.check(lib.mhx_band_keys_dev(ctx.handle, d_sig.ptr, n, k, bands, r, d_out.ptr)))
        got = d_out.download((n, bands * r), np.uint64)
        assert got[:512].tobytes() == LB.band_keys(sig[:512], bands, r, gpu_mode="disable").tobytes()
        report(f"band_keys bands={bands} r={r} K={k}", ms, n, "signatures", n * 16 * bands * r)
    d_out = ctx.alloc(n * (12 + 4 * k))
    ms = timed(ctx, lambda: _native.check(lib.mhx_lean_serialize_dev(ctx.handle, d_sig.ptr, n, k, 1, d_out.ptr)))
    report(f"lean_serialize K={k}", ms, n, "signatures", n * (8 * k + 12 + 4 * k))
    d_dig = ctx.alloc(n * 32 * 8)
    ms = timed(ctx, lambda: _native.check(lib.mhx_band_digests_dev(ctx.handle, d_sig.ptr, n, k, 32, 8, d_dig.ptr)))
    assert np.array_equal(d_dig.download((n, 32), np.uint64)[:256], LB.band_digests(sig[:256], 32, 8, gpu_mode="disable"))
    report(f"band_digests bands=32 r=8 K={k}", ms, n, "signatures", n * (8 * k + 8 * 32))
    d_sd, d_sr = ctx.alloc(n * 32 * 8), ctx.alloc(n * 32 * 4)
    ms = timed(ctx, lambda: _native.check(lib.mhx_lsh_sort_bands_dev(ctx.handle, d_sig.ptr, n, k, 32, 8, d_sd.ptr, d_sr.ptr)), reps=3)
    sd = d_sd.download((32, n), np.uint64)
    assert np.all(sd[:, 1:] >= sd[:, :-1])
    report(f"lsh_sort_bands bands=32 r=8 K={k} (digests + one radix sort of 32 x {n} keys by (band, digest))", ms, n, "signatures", n * (8 * k + 12 * 32))
    m = 4_000_000
    pairs = rng.randint(0, n, (m, 2)).astype(np.int64)
    d_pairs, d_cnt = ctx.to_device(pairs), ctx.alloc(m * 4)
    ms = timed(ctx, lambda: _native.check(lib.mhx_jaccard_pairs_dev(ctx.handle, d_sig.ptr, d_sig.ptr, k, d_pairs.ptr, m, d_cnt.ptr)))
    got = d_cnt.download((m,), np.int32)
    assert np.array_equal(got[:512], np.count_nonzero(sig[pairs[:512, 0]] == sig[pairs[:512, 1]], axis=1))
    report(f"jaccard_pairs {m} random pairs K={k}", ms, m, "pairs", m * (16 * k + 16 + 4))
    d_y = ctx.to_device(rng.randint(0, 2**32, (n, k), dtype=np.uint64))
    d_o = ctx.alloc(n * k * 8)
    ms = timed(ctx, lambda: _native.check(lib.mhx_minhash_merge_dev(ctx.handle, d_sig.ptr, d_y.ptr, n * k, d_o.ptr)))
    report(f"minhash_merge K={k}", ms, n, "signatures", n * k * 24)


def minhash_shapes(ctx):
    rng = np.random.RandomState(3)
    for n, t, k in ((1000, 64, 16), (1_000_000, 256, 256), (200_000, 256, 512), (1, 50_000, 128), (1, 50_000, 256), (1, 50_000, 512), (64, 100_000, 128)):
        tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
        a, b = MinHash(num_perm=k, seed=1).permutations
        d_tok, d_out = ctx.to_device(tok), ctx.alloc(n * k * 8)
        ms = timed(ctx, lambda: ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_out.ptr, _native.MHX_U64))
        got = d_out.download((n, k), np.uint64)
        chk = min(n, 256)
        assert np.array_equal(got[:chk], MinHash.bulk_signatures(tok[:chk], num_perm=k, seed=1, hashfunc=prehashed, gpu_mode="disable"))
        report(f"minhash_bulk N={n} T={t} K={k}", ms, n, "signatures", n * (8 * t + 8 * k), pairs_per_s=n * t * k / (ms * 1e-3))
        if n == 1:
            t0 = time.perf_counter()
            ctx.minhash_update_batch((a, b), tok.reshape(-1), np.full(k, 2**32 - 1, dtype=np.uint64))
            print(json.dumps({"name": f"update_batch host->host n={t} K={k}", "ms": round(1e3 * (time.perf_counter() - t0), 4)}), flush=True)


def minhash_ragged(ctx, only_repeats=False):
    """Realistic corpora: ragged sets (CSR), and sets with repeated tokens (failed sieve proofs)."""
    rng = np.random.RandomState(7)
    k = 128
    a, b = MinHash(num_perm=k, seed=1).permutations
    cases = (("ragged 32..480", 500_000, 32, 480, 0.0), ("ragged 1..100", 1_000_000, 1, 100, 0.0),
             ("dense 256 with 10% repeated tokens", 500_000, 256, 256, 0.1), ("dense 256 with 1% repeated tokens", 500_000, 256, 256, 0.01))
    if only_repeats and os.environ.get("MHX_REPEATS_RATE"):  # one corpus only (kernel traces)
        cases = tuple(c for c in cases if c[4] == float(os.environ["MHX_REPEATS_RATE"]))
        only_repeats = False
    for name, n, lo, hi, dup in (cases[2:] if only_repeats else cases):
        lens = rng.randint(lo, hi + 1, size=n).astype(np.int64)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        hv = rng.randint(0, 2**32, size=int(off[-1]), dtype=np.uint64)
        if dup:  # repeated tokens at random places of the same set (dense corpus: set i = tokens [256 i, 256 i + 256))
            m = int(dup * hv.size)
            dst = rng.randint(0, hv.size, size=m)
            src = (dst // 256) * 256 + rng.randint(0, 256, size=m)
            hv[dst] = hv[src]
        d_hv, d_off, d_out = ctx.to_device(hv), ctx.to_device(off), ctx.alloc(n * k * 8)
        run = lambda: ctx.minhash_bulk_dev((a, b), d_hv.ptr, _native.MHX_U64, d_off.ptr, 0, n, hv.size, None, 0, d_out.ptr, _native.MHX_U64)
        ms = timed(ctx, run)
        ctx.counters(True)
        run()
        c = ctx.counters(False)
        got = d_out.download((n, k), np.uint64)
        assert np.array_equal(got[:256], MinHash.bulk_signatures((hv[: off[256]], off[:257]), num_perm=k, seed=1, hashfunc=prehashed, gpu_mode="disable"))
        report(f"minhash_bulk {name} N={n} K={k}", ms, n, "signatures", 8 * hv.size + 8 * k * n, pairs_per_s=hv.size * k / (ms * 1e-3),
               tokens=int(hv.size), **c)


def sha1(ctx):
    """Device SHA-1 of byte tokens (row f2) and what it does to bulk() on raw byte tokens end to end."""
    from datasketch_amd import MinHash

    rng = np.random.RandomState(11)
    n = 32_000_000
    lens = rng.randint(3, 13, size=n).astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    buf = rng.randint(0, 256, int(offs[-1]), dtype=np.uint8)
    d_buf, d_off, d_out = ctx.to_device(buf), ctx.to_device(offs), ctx.alloc(n * 4)
    run = lambda: _native.check(ctx.lib.mhx_sha1_tokens_dev(ctx.handle, d_buf.ptr, d_off.ptr, n, _native.MHX_U32, d_out.ptr))
    ms = timed(ctx, run)
    import hashlib, struct
    got = d_out.download((n,), np.uint32)
    for i in (0, 1, 12345, n - 1):
        assert got[i] == struct.unpack("<I", hashlib.sha1(buf[offs[i]:offs[i + 1]].tobytes()).digest()[:4])[0]
    report(f"sha1_hash32 of {n} byte tokens (3..12 bytes)", ms, n, "tokens", buf.size + 8 * n + 4 * n)
    # end to end from Python objects: 20k sets x 100 byte tokens
    sets = [[b"w%d" % v for v in rng.randint(0, 1 << 20, 100)] for _ in range(20_000)]
    for mode in ("always", "disable"):
        t0 = time.perf_counter()
        sig = MinHash.bulk_signatures(sets, num_perm=128, seed=1, gpu_mode=mode)
        dt = time.perf_counter() - t0
        print(json.dumps({"name": f"bulk_signatures on byte tokens, default hashfunc, gpu_mode={mode}", "sets": len(sets),
                          "tokens_per_set": 100, "seconds": round(dt, 3), "sets_per_s": len(sets) / dt,
                          "checksum": int(sig.sum() % (1 << 61))}), flush=True)
    # the same corpus already packed (what a tokenizer writing into one buffer hands over): MinHash.bulk_signatures(packed=...), round 6
    flat = [t for s_ in sets for t in s_]
    pbuf = np.frombuffer(b"".join(flat), dtype=np.uint8)
    pboff = np.concatenate([[0], np.cumsum([len(t) for t in flat])]).astype(np.int64)
    psoff = np.arange(len(sets) + 1, dtype=np.int64) * 100
    want = sig
    for rep in range(2):
        t0 = time.perf_counter()
        sig = MinHash.bulk_signatures(packed=(pbuf, pboff, psoff), num_perm=128, seed=1, gpu_mode="always")
        dt = time.perf_counter() - t0
    assert np.array_equal(sig, want)
    print(json.dumps({"name": "bulk_signatures(packed=...) on the same byte tokens, gpu_mode=always (second call)", "sets": len(sets), "tokens_per_set": 100,
                      "seconds": round(dt, 4), "sets_per_s": len(sets) / dt, "tokens_per_s": len(flat) / dt}), flush=True)
    # a corpus large enough for the link to matter: 1M sets x 100 tokens of 3..12 bytes, packed
    n_sets, per = 1_000_000, 100
    lens = rng.randint(3, 13, size=n_sets * per).astype(np.int64)
    boff = np.zeros(n_sets * per + 1, dtype=np.int64)
    np.cumsum(lens, out=boff[1:])
    big = rng.randint(0, 256, int(boff[-1]), dtype=np.uint8)
    soff = np.arange(n_sets + 1, dtype=np.int64) * per
    for rep in range(2):
        t0 = time.perf_counter()
        sig = MinHash.bulk_signatures(packed=(big, boff, soff), num_perm=128, seed=1, gpu_mode="always", out_dtype=np.uint32)
        dt = time.perf_counter() - t0
    print(json.dumps({"name": "bulk_signatures(packed=...) 1M sets x 100 byte tokens (3..12 bytes), uint32 out, host numpy in -> host numpy out (second call)",
                      "sets": n_sets, "seconds": round(dt, 3), "sets_per_s": n_sets / dt, "tokens_per_s": n_sets * per / dt,
                      "bytes_over_pcie": int(big.size + boff.nbytes + soff.nbytes + sig.nbytes)}), flush=True)


def reference_gpu_benchmark(ctx):
    """The reference's own GPU benchmark (benchmark/sketches/minhash_gpu_benchmark.py: one update_batch of
    n byte tokens "token-i", seed 7, SHA-1 included, mean of 5 after a warm-up) -- the only numbers the
    reference publishes for this path (CPU 163/258/485 ms, CuPy GPU 60/63/72 ms at
    n = 50 000 and num_perm = 128/256/512, hardware unstated)."""
    from datasketch_amd import MinHash

    for n in (1000, 10_000, 50_000):
        data = [f"token-{i}".encode("utf-8") for i in range(n)]
        for k in (128, 256, 512):
            row = {"name": f"update_batch of {n} byte tokens, num_perm={k} (reference GPU benchmark shape)"}
            digests = {}
            for mode in ("always", "disable"):
                times = []
                for rep in range(6):
                    m = MinHash(num_perm=k, seed=7, gpu_mode=mode)
                    t0 = time.perf_counter()
                    m.update_batch(data)
                    times.append((time.perf_counter() - t0) * 1e3)
                row[f"ms_gpu_mode_{mode}"] = round(float(np.mean(times[1:])), 3)
                digests[mode] = m.hashvalues
            assert np.array_equal(digests["always"], digests["disable"])
            print(json.dumps(row), flush=True)


def lsh(ctx, n):
    """Candidate pairs by sort (rows f1/f4): a corpus with near-duplicate rows, K=128, (b, r) = (32, 4)."""

    rng = np.random.RandomState(21)
    k, b, r = 128, 32, 4
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    dup = rng.randint(0, n, n // 10)           # 10% of the rows share about half of their positions with another row
    src = rng.randint(0, n, n // 10)
    keep = rng.random_sample((n // 10, k)) < 0.5
    sig[dup] = np.where(keep, sig[src], sig[dup])
    d_sig = ctx.to_device(sig)
    d_dig, d_rows = ctx.alloc(8 * n * b), ctx.alloc(4 * n * b)
    cap = 4 * n
    d_pairs = ctx.alloc(16 * cap)
    found, raw = ctypes.c_int64(0), ctypes.c_int64(0)

    def chain():
        _native.check(ctx.lib.mhx_lsh_sort_bands_dev(ctx.handle, d_sig.ptr, n, k, b, r, d_dig.ptr, d_rows.ptr))
        _native.check(ctx.lib.mhx_lsh_candidate_pairs_dev(ctx.handle, d_dig.ptr, d_rows.ptr, n, b, d_pairs.ptr, cap,
                                                          ctypes.byref(found), ctypes.byref(raw)))

    ms = timed(ctx, chain)
    ms_sort = timed(ctx, lambda: _native.check(ctx.lib.mhx_lsh_sort_bands_dev(ctx.handle, d_sig.ptr, n, k, b, r, d_dig.ptr, d_rows.ptr)))
    pairs = d_pairs.download((cap, 2), np.int64)[: found.value]
    d_counts = ctx.alloc(4 * max(1, found.value))
    ms_j = timed(ctx, lambda: _native.check(ctx.lib.mhx_jaccard_pairs_dev(ctx.handle, d_sig.ptr, d_sig.ptr, k, d_pairs.ptr, found.value, d_counts.ptr)))
    report(f"lsh candidate pairs (sort {b} bands + runs + emit + sort/unique) N={n}", ms, n, "signatures", n * k * 8,
           sort_bands_ms=round(ms_sort, 4), unique_pairs=int(found.value), raw_pairs=int(raw.value), jaccard_pairs_ms=round(ms_j, 4))
    m = min(n, 100_000)  # the numpy bucketing on a sample, and equality of the two on it
    t0 = time.perf_counter()
    want = LB.candidate_pairs(sig[:m], b, r, gpu_mode="disable")
    dt = time.perf_counter() - t0
    got = LB.candidate_pairs(sig[:m], b, r, gpu_mode="always")
    assert np.array_equal(got, want)
    print(json.dumps({"name": f"candidate_pairs numpy path N={m}", "seconds": round(dt, 3), "pairs": int(len(want))}), flush=True)
    t0 = time.perf_counter()
    LB.candidate_pairs(sig[:m], b, r, gpu_mode="always")
    print(json.dumps({"name": f"candidate_pairs device path host->host N={m}", "seconds": round(time.perf_counter() - t0, 4)}), flush=True)


class _DictLSH:
    """The state of the reference's MinHashLSH with its in-memory storage (ref: datasketch/lsh.py:178-199,
    storage.py:210-259) -- enough of it for insert_bulk / query_bulk and for the per-key loop below, so that the
    timing does not need the reference repository on the GPU box (tests/test_lsh_bulk.py checks both against the
    real class where it is mounted)."""

    def __init__(self, h, b, r):
        import collections

        class Store:
            def __init__(self, factory):
                self._dict = collections.defaultdict(factory)

        self.h, self.b, self.r, self.prepickle, self.hashfunc = h, b, r, False, None
        self.keys = Store(list)
        self.hashtables = [Store(set) for _ in range(b)]
        self.hashranges = [(i * r, (i + 1) * r) for i in range(b)]

    def insert(self, key, hashvalues):  # lsh.py:326-347 on dict storage
        hs = [bytes(hashvalues[s:e].byteswap().data) for s, e in self.hashranges]
        self.keys._dict[key].extend(hs)
        for h, table in zip(hs, self.hashtables):
            table._dict[h].add(key)

    def query(self, hashvalues):  # lsh.py:423-431
        cand = set()
        for (s, e), table in zip(self.hashranges, self.hashtables):
            cand.update(table._dict.get(bytes(hashvalues[s:e].byteswap().data), ()))
        return list(cand)


def lsh_index(ctx, n):
    """MinHashLSH insertion and query in bulk (row f1): K=256, (b, r) = (32, 8) as config 3/5."""
    rng = np.random.RandomState(33)
    k, b, r = 256, 32, 8
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    dup = rng.randint(0, n, n // 20)
    sig[dup, : k // 2] = sig[rng.randint(0, n, n // 20), : k // 2]   # 5 % of the rows share half their bands with another row
    keys = [b"%d" % i for i in range(n)]
    one, bulk = _DictLSH(k, b, r), _DictLSH(k, b, r)
    m = min(n, 20_000)
    t0 = time.perf_counter()
    for i in range(m):
        one.insert(keys[i], sig[i])
    loop_s = time.perf_counter() - t0
    frozen = _DictLSH(k, b, r)
    t0 = time.perf_counter()
    LB.insert_bulk(frozen, keys, sig, gpu_mode="always", settle="freeze")
    freeze_s = time.perf_counter() - t0
    del frozen
    import gc

    gc.unfreeze()
    gc.collect()
    t0 = time.perf_counter()
    LB.insert_bulk(bulk, keys, sig, gpu_mode="always")  # settle="collect": one full collection at the end, in the time
    bulk_s = time.perf_counter() - t0
    for j in (0, b - 1):
        sample = list(one.hashtables[j]._dict.items())[:2000]
        assert all(bulk.hashtables[j]._dict[h] >= v for h, v in sample)
    probes = sig[rng.randint(0, n, 20_000)]
    t0 = time.perf_counter()
    got = LB.query_bulk(bulk, probes, gpu_mode="always")
    qbulk_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = [bulk.query(row) for row in probes[:2000]]
    qloop_s = time.perf_counter() - t0
    assert all(set(a) == set(c) for a, c in zip(got[:2000], want))
    print(json.dumps({"name": f"MinHashLSH dict index, K={k} ({b} x {r}): insert_bulk of {n} keys", "seconds": round(bulk_s, 3),
                      "keys_per_s": n / bulk_s, "seconds_settle_freeze": round(freeze_s, 3), "keys_per_s_settle_freeze": n / freeze_s,
                      "per_key_loop_keys_per_s": m / loop_s, "per_key_loop_sample": m,
                      "query_bulk_probes_per_s": len(probes) / qbulk_s, "per_probe_loop_probes_per_s": 2000 / qloop_s}), flush=True)
    del one, bulk, got, want
    # the same index resident on the GPU as sorted bands: build + query 1M probes
    t0 = time.perf_counter()
    idx = LB.SortedBandsIndex(sig, b, r)
    ctx.synchronize()
    build_s = time.perf_counter() - t0
    probes = sig[rng.randint(0, n, min(n, 1_000_000))].copy()
    probes[::2, rng.randint(0, k, 40)] = 3
    idx.query(probes[:1000])
    t0 = time.perf_counter()
    offsets, rows = idx.query(probes)
    q_s = time.perf_counter() - t0
    print(json.dumps({"name": f"SortedBandsIndex (device), K={k} ({b} x {r}), {n} rows: build from host matrix", "seconds": round(build_s, 4),
                      "query_probes": int(len(probes)), "query_seconds_host_to_host": round(q_s, 4), "probes_per_s": len(probes) / q_s,
                      "candidates": int(rows.size)}), flush=True)
    # b-bit Jaccard of candidate pairs on packed rows
    blocks = ctx.bbit_pack(sig, 1)
    pairs = np.stack([np.repeat(np.arange(len(offsets) - 1), np.diff(offsets)), rows], axis=1)[:2_000_000]
    d_blocks, d_pairs, d_cnt = ctx.to_device(blocks), ctx.to_device(pairs), ctx.alloc(4 * len(pairs))
    ms = timed(ctx, lambda: _native.check(ctx.lib.mhx_bbit_jaccard_pairs_dev(ctx.handle, d_blocks.ptr, d_blocks.ptr, k, 1, d_pairs.ptr, len(pairs), d_cnt.ptr)))
    report(f"bbit_jaccard_pairs b=1 K={k}, {len(pairs)} pairs", ms, len(pairs), "pairs", len(pairs) * (2 * 32 + 16 + 4))


def weighted_python_level(ctx, n, dim, s):
    """WeightedMinHashGenerator.minhash_many_arrays from Python on a dense matrix: the device-built CSR (dense
    entry point) against the scipy CSR the reference route needs first."""
    import scipy.sparse as sp

    rng = np.random.RandomState(42)
    x = rng.uniform(0, 100, (n, dim)).astype(np.float32)
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    g.minhash_many_arrays(x[:64])
    t0 = time.perf_counter()
    dense = g.minhash_many_arrays(x)
    t1 = time.perf_counter()
    csr = g.minhash_many_arrays(sp.csr_matrix(x))
    t2 = time.perf_counter()
    assert np.array_equal(dense[0], csr[0])
    print(json.dumps({"name": f"minhash_many_arrays from Python, dense {n} x {dim}, S={s}", "dense_entry_seconds": round(t1 - t0, 3),
                      "scipy_csr_route_seconds": round(t2 - t1, 3), "vectors_per_s_dense_entry": n / (t1 - t0)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weighted-rows", type=int, default=20000)
    ap.add_argument("--sigs", type=int, default=1_000_000)
    ap.add_argument("--only", default="")
    ap.add_argument("--opt", action="append", default=[], help="context option key=value (mhx_ctx_set_option), repeatable")
    args = ap.parse_args()
    ctx = _native.context()
    for kv in args.opt:
        key, value = kv.split("=")
        ctx.set_option(key, int(value))
    print(json.dumps(ctx.info()), flush=True)
    if args.only in ("", "minhash"):
        minhash_shapes(ctx)
    if args.only == "repeats":
        minhash_ragged(ctx, only_repeats=True)
    if args.only in ("", "minhash", "ragged"):
        minhash_ragged(ctx)
    if args.only in ("", "sha1"):
        sha1(ctx)
    if args.only in ("", "packing"):
        packing(ctx, args.sigs, 256)
    if args.only in ("", "refbench"):
        reference_gpu_benchmark(ctx)
    if args.only in ("", "lsh", "lsh_index"):
        lsh_index(ctx, args.sigs)
    if args.only in ("", "lsh"):
        lsh(ctx, args.sigs)
    if args.only in ("", "weighted", "weighted_py"):
        weighted_python_level(ctx, args.weighted_rows, 4096, 128)
    if args.only in ("", "weighted"):
        weighted(ctx, args.weighted_rows, 4096, 128, 1.0)
        weighted(ctx, args.weighted_rows * 4, 4096, 128, 0.01)
        weighted(ctx, args.weighted_rows, 1024, 64, 0.1)


if __name__ == "__main__":
    main()
