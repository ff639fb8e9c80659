"""bench.py's `extra` entries on one GPU: BASELINE.json configs 3, 4, 5 at their per-GPU shapes and at their stated size."""
from __future__ import annotations

import json
import os
import time

import numpy as np

from benchmarks.common import HBM_PEAK_GBS, ROOT, _download_rows, _fnv1a64, _roof, _timed
from benchmarks.cpu import _usable_cores, cpu_model

def extra_configs(ctx, tokens, d_tok, seed, only):
    """BASELINE.json configs 3, 4, 5 at their per-GPU shapes (the 8-GPU configs divide by 8), each timed with HIP
    events and parity-gated on a sample: a wrong result aborts the bench."""
    import ctypes

    from datasketch_amd import _native
    from datasketch_amd.minhash import MinHash
    from oracle import oracle as O

    lib = ctx.lib
    res = {}
    n3, t, k3, bands, r = 1_250_000, tokens.shape[1], 256, 32, 8
    state = {}

    def c3_corpus():
        # the config-2 corpus (already resident) + 250k more rows = one rank's 1.25M-set shard of config 3
        if "d_tok3" not in state:
            more = np.random.RandomState(4242).randint(0, 2**32, size=(n3 - tokens.shape[0], t), dtype=np.uint64)
            d = ctx.alloc(n3 * t * 8)
            ctx.copy_dev(d.ptr, d_tok.ptr, tokens.size * 8)
            d.upload(more, offset=tokens.size * 8)
            state["d_tok3"], state["more"] = d, more
            p3 = MinHash(num_perm=k3, seed=seed, hashfunc=lambda x: x).permutations
            state["perms3"] = p3
            state["d_sig3"] = ctx.alloc(n3 * k3 * 4)
            ctx.minhash_bulk_dev(p3, d.ptr, _native.MHX_U64, None, t, n3, n3 * t, None, 0, state["d_sig3"].ptr, _native.MHX_U32)
        return state

    def sample_rows():
        rows = np.unique(np.concatenate([np.linspace(0, tokens.shape[0] - 1, 384).astype(np.int64),
                                         np.arange(tokens.shape[0], tokens.shape[0] + 128)]))
        tok = np.concatenate([tokens[rows[rows < tokens.shape[0]]], state["more"][: 128]])
        return rows, tok

    if "c3" in only:
        st = c3_corpus()
        p3, d3, dsig = st["perms3"], st["d_tok3"], st["d_sig3"]
        ms_sig = _timed(ctx, lambda: ctx.minhash_bulk_dev(p3, d3.ptr, _native.MHX_U64, None, t, n3, n3 * t, None, 0, dsig.ptr, _native.MHX_U32))
        d_dig = ctx.alloc(n3 * bands * 8)
        d_sd = ctx.alloc(n3 * bands * 8)
        d_sr = ctx.alloc(n3 * bands * 4)
        ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, d_dig.ptr)))
        sort = lambda: _native.check(lib.mhx_lsh_sort_bands_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, d_sd.ptr, d_sr.ptr))
        ctx.set_option("lsh.sort", 1)  # A/B: the library radix sort (round 2's path, now the fallback), same call
        try:
            ms_sort_radix = _timed(ctx, sort, reps=3)
            sd_radix, sr_radix = d_sd.download((bands, n3), np.uint64), d_sr.download((bands, n3), np.uint32)
        finally:
            ctx.set_option("lsh.sort", 0)
        ms_sort = _timed(ctx, sort)
        # config 3 as a chain computes the digests once: the bucketing takes the [n, bands] digest matrix that was just written
        sort_dig = lambda: _native.check(lib.mhx_lsh_sort_digests_dev(ctx.handle, d_dig.ptr, n3, bands, d_sd.ptr, d_sr.ptr))
        ms_sort_dig_rm = _timed(ctx, sort_dig)
        sd_dig, sr_dig = d_sd.download((bands, n3), np.uint64), d_sr.download((bands, n3), np.uint32)
        # ... and the layout the chain runs on: the digests band-major ([bands, n]), read by the bucketing with unit stride
        d_dig_bm = ctx.alloc(n3 * bands * 8)
        ms_dig_bm = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, _native.BAND_MAJOR, d_dig_bm.ptr)))
        ms_sort_dig = _timed(ctx, lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig_bm.ptr, n3, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr)))
        if not (np.array_equal(d_sd.download((bands, n3), np.uint64), sd_dig) and np.array_equal(d_sr.download((bands, n3), np.uint32), sr_dig)):
            raise SystemExit("PARITY FAILURE (extra.c3): bucketing from the band-major digests differs from bucketing from the row-major ones")
        dig_bm = d_dig_bm.download((bands, n3), np.uint64)
        d_dig_bm.free()
        ms_sort = _timed(ctx, sort)  # (d_sd / d_sr hold the sort-from-signatures result again for the checks below)
        # parity: signature rows against the C oracle, digests against FNV-1a of the reference's key bytes, order of the sort
        rows, tok = sample_rows()
        a3, b3 = p3
        want = O.c_minhash_bulk_dense(tok, a3, b3)
        sig = dsig.download((n3, k3), np.uint32)
        if not np.array_equal(sig[rows].astype(np.uint64), want):
            raise SystemExit("PARITY FAILURE (extra.c3): K=256 signatures differ from the oracle")
        keys = O.c_band_keys(want[:64], bands, r)
        dig = d_dig.download((n3, bands), np.uint64)
        if not np.array_equal(dig_bm, dig.T):
            raise SystemExit("PARITY FAILURE (extra.c3): band-major digests differ from the row-major ones")
        del dig_bm
        for i in range(64):
            for j in range(bands):
                if int(dig[rows[i], j]) != _fnv1a64(keys[i, j * r:(j + 1) * r].tobytes()):
                    raise SystemExit("PARITY FAILURE (extra.c3): band digest differs from FNV-1a-64 of the reference's key bytes")
        sd = d_sd.download((bands, n3), np.uint64)
        sr = d_sr.download((bands, n3), np.uint32)
        for j in (0, bands - 1):
            if np.any(sd[j, 1:] < sd[j, :-1]) or not np.array_equal(dig[sr[j].astype(np.int64), j], sd[j]):
                raise SystemExit("PARITY FAILURE (extra.c3): sorted bands are not the digests in ascending order")
        if not (np.array_equal(sd, sd_radix) and np.array_equal(sr, sr_radix)):
            raise SystemExit("PARITY FAILURE (extra.c3): the bucketing passes and the stable radix sort disagree")
        if not (np.array_equal(sd, sd_dig) and np.array_equal(sr, sr_dig)):
            raise SystemExit("PARITY FAILURE (extra.c3): bucketing from the digest matrix differs from bucketing from the signatures")
        del sig, dig, sd, sr, sd_radix, sr_radix, sd_dig, sr_dig
        res["c3"] = {
            "workload": f"config 3 per-GPU shard: {n3} sets x {t} tokens, num_perm={k3} (uint64 tokens in, uint32 signatures out = the all-gather's wire format), then LSH band digests ({bands} bands x {r}) and the bucketing sort",
            "signatures": dict(_roof(n3 * (8 * t + 4 * k3), ms_sig), signatures_per_s=n3 / (ms_sig * 1e-3),
                               note="algorithmic bytes 8*T + 4*K per signature (uint32 out); SURVEY 8d's 4096 B/sig assumes uint64 out"),
            "band_digests": dict(_roof(n3 * (4 * k3 + 8 * bands), ms_dig_bm), layout="band-major [bands, n] (MHX_BAND_MAJOR), written through an LDS tile"),
            "band_digests_row_major": _roof(n3 * (4 * k3 + 8 * bands), ms_dig),
            "lsh_sort_bands": dict(_roof(n3 * (4 * k3 + 12 * bands), ms_sort), keys_per_s=n3 * bands / (ms_sort * 1e-3),
                                   kernels="band_digest_bm_kernel (band-major digests into scratch) + lsh_bin_scatter_kernel + lsh_bin_sort_kernel",
                                   note="digests computed (their own pass since round 5: 0.86 -> 0.77 ms), scattered to bins by their top bits, every bin ordered in LDS: exact (band, digest, row) "
                                        "order; bytes = signatures in, (digest, row) out"),
            "lsh_sort_bands_radix": dict(_roof(n3 * (4 * k3 + 12 * bands), ms_sort_radix), keys_per_s=n3 * bands / (ms_sort_radix * 1e-3),
                                         note="lsh.sort=1: digests + the library radix sort of (band, digest prefix, row) + exact clean-up (round 2's path, "
                                              "now the fallback), same call, same box"),
            "lsh_sort_digests": dict(_roof(n3 * (8 * bands + 12 * bands), ms_sort_dig), keys_per_s=n3 * bands / (ms_sort_dig * 1e-3),
                                     note="mhx_lsh_sort_digests_layout_dev on the band-major digest matrix band_digests has just written (8 B read per key with "
                                          "unit stride, no hashing); bytes = digests in, (digest, row) out"),
            "lsh_sort_digests_row_major": dict(_roof(n3 * (8 * bands + 12 * bands), ms_sort_dig_rm), keys_per_s=n3 * bands / (ms_sort_dig_rm * 1e-3),
                                               note="the same from an [n, bands] matrix: every 128-byte input line is fetched by the four XCDs whose bands share it "
                                                    "(profiles/r05_pmc_scatter_work_orders.txt)"),
            "pipeline_ms": ms_sig + ms_dig_bm + ms_sort_dig,
            "pipeline": "signatures -> band_digests (band-major, kept: one key array per hashtable) -> lsh_sort_digests; digests computed once",
            "pipeline_ms_digests_twice": ms_sig + ms_dig + ms_sort,
            "parity": f"{len(rows)} signature rows vs the C oracle, 64 x {bands} digests vs FNV-1a-64 of the reference's key bytes, 2 bands' order, all {bands} sorted bands equal to the stable radix sort's and to the sort from the digest matrix",
        }
        for d in (d_dig, d_sd, d_sr):
            d.free()

    if "c5" in only:
        st = c3_corpus()
        dsig = st["d_sig3"]
        nb = k3 // 64
        d_pack = ctx.alloc(n3 * nb * 8)
        d_dig = ctx.alloc(n3 * bands * 8)
        ms_pack = _timed(ctx, lambda: _native.check(lib.mhx_bbit_pack_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, 1, d_pack.ptr)))
        ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, d_dig.ptr)))
        rows, tok = sample_rows()
        a3, b3 = st["perms3"]
        want = O.c_minhash_bulk_dense(tok, a3, b3)
        pack = d_pack.download((n3, nb), np.uint64)
        dig = d_dig.download((n3, bands), np.uint64)
        if not np.array_equal(pack[rows], O.c_bbit_pack(want, 1)):
            raise SystemExit("PARITY FAILURE (extra.c5): b=1 blocks differ from the oracle's bBitMinHash packing")
        # the same two outputs from ONE read of the matrix (bbit_digest_fused_kernel), over buffers cleared in between
        import ctypes as _ct

        for d in (d_pack, d_dig):
            _native.check(lib.mhx_memset_dev(ctx.handle, _ct.c_void_p(d.ptr), 0, d.nbytes))
        one_read = []
        ms_fused = _timed(ctx, lambda: one_read.append(ctx.bbit_pack_band_digests_dev(dsig.ptr, _native.MHX_U32, n3, k3, 1, bands, r, d_pack.ptr, d_dig.ptr)))
        if not (np.array_equal(d_pack.download((n3, nb), np.uint64), pack) and np.array_equal(d_dig.download((n3, bands), np.uint64), dig)):
            raise SystemExit("PARITY FAILURE (extra.c5): the fused kernel's blocks / digests differ from the two kernels'")
        keys = O.c_band_keys(want[:64], bands, r)
        for i in range(64):
            for j in range(bands):
                if int(dig[rows[i], j]) != _fnv1a64(keys[i, j * r:(j + 1) * r].tobytes()):
                    raise SystemExit("PARITY FAILURE (extra.c5): band digest differs from FNV-1a-64 of the reference's key bytes")
        for d in (d_pack, d_dig):
            _native.check(lib.mhx_memset_dev(ctx.handle, _ct.c_void_p(d.ptr), 0, d.nbytes))
        ms_fused_bm = _timed(ctx, lambda: one_read.append(ctx.bbit_pack_band_digests_dev(dsig.ptr, _native.MHX_U32, n3, k3, 1, bands, r, d_pack.ptr, d_dig.ptr, _native.BAND_MAJOR)))
        if not (np.array_equal(d_pack.download((n3, nb), np.uint64), pack) and np.array_equal(d_dig.download((bands, n3), np.uint64), dig.T)):
            raise SystemExit("PARITY FAILURE (extra.c5): the fused kernel's band-major digests / blocks differ from the two kernels'")
        # the same with the reference's own signature width: uint64 hashvalues in (8K + K/8 + 8*bands = 2336 B per signature)
        sig64 = st["d_sig3"].download((n3, k3), np.uint32).astype(np.uint64)
        d_sig64 = ctx.to_device(sig64)
        del sig64
        for d in (d_pack, d_dig):
            _native.check(lib.mhx_memset_dev(ctx.handle, _ct.c_void_p(d.ptr), 0, d.nbytes))
        one_read64 = []
        ms_fused64 = _timed(ctx, lambda: one_read64.append(ctx.bbit_pack_band_digests_dev(d_sig64.ptr, _native.MHX_U64, n3, k3, 1, bands, r, d_pack.ptr, d_dig.ptr, _native.BAND_MAJOR)))
        if not (np.array_equal(d_pack.download((n3, nb), np.uint64), pack) and np.array_equal(d_dig.download((bands, n3), np.uint64), dig.T)):
            raise SystemExit("PARITY FAILURE (extra.c5): the fused kernel on uint64 signatures differs from the uint32 run")
        d_sig64.free()
        del pack, dig
        res["c5"] = {
            "workload": f"config 5 per-GPU shard: b=1 packing of {n3} x {k3} signatures (uint32, as all-gathered) + LSH band hashing ({bands} x {r})",
            "fused": dict(_roof(n3 * (4 * k3 + k3 // 8 + 8 * bands), ms_fused), one_read=bool(one_read and all(one_read)),
                          kernel="bbit_digest_fused_kernel: blocks and digests from one read of the matrix (algorithmic bytes: 4K in, K/8 + 8*bands out)"),
            "bbit_pack_b1": _roof(n3 * (4 * k3 + k3 // 8), ms_pack),
            "band_digests": _roof(n3 * (4 * k3 + 8 * bands), ms_dig),
            "fused_band_major": dict(_roof(n3 * (4 * k3 + k3 // 8 + 8 * bands), ms_fused_bm),
                                     note="the same kernel writing the digests [bands, n] through an LDS tile: the layout the bucketing reads with unit stride"),
            "fused_band_major_uint64_in": dict(_roof(n3 * (8 * k3 + k3 // 8 + 8 * bands), ms_fused64), one_read=bool(one_read64 and all(one_read64)),
                                               note="the reference's own width: uint64 hashvalues in (SURVEY.md 8d: 8K + K/8 + 8*bands = 2336 B per signature at K = 256, "
                                                    "32 bands); blocks and band-major digests equal to the uint32 run's on all rows"),
            "pipeline_ms": ms_fused_bm,
            "pipeline_ms_two_kernels": ms_pack + ms_dig,
            "parity": f"{len(rows)} packed rows vs the C oracle (b_bit_minhash.py:82-101 bit order), 64 x {bands} digests vs FNV-1a-64 of the reference's key "
                      f"bytes, and the fused kernel's outputs equal to the two kernels' on all {n3} rows",
        }
        d_pack.free()
        d_dig.free()

    for key in ("d_tok3", "d_sig3"):
        if key in state:
            state[key].free()
    state.clear()

    if "c4" in only:
        res["c4"] = extra_c4(ctx)
        res["c4_sparse"] = extra_c4_sparse(ctx)
    return res


# ------------------------------------------------------------------------------------------------


def full_corpus(ctx, n, t, shards=8, piece=50_000, sample=4096, blocks=(0, 0)):
    """Config 3's corpus, resident in HBM: shard q = RandomState(42 + q).randint(0, 2**32, (rows_q, t), uint64) (SURVEY.md
    section 8d), drawn by one host thread per shard (numpy releases the GIL inside the draw) in pieces of 50k sets that
    go up as they are made -- 20.5 GB on the device, 100 MB per thread on the host.  Returns the device buffer, `sample`
    row numbers spread over the whole corpus and their tokens (what the oracle will be given) -- and, with blocks = (count,
    length), the first rows and the tokens of `count` runs of `length` consecutive sets spread over the corpus (a
    million rows of config 3 against the oracle, not four thousand; runs, so that their signatures come back in `count` copies)."""
    from concurrent.futures import ThreadPoolExecutor

    from datasketch_amd.dist import shard_rows

    d_tok = ctx.alloc(n * t * 8)
    rows = np.unique(np.concatenate([np.linspace(0, n - 1, sample).astype(np.int64), [0, n - 1]]))
    count, length = blocks
    length = min(length, n)
    starts = np.unique(np.linspace(0, n - length, count).astype(np.int64)) if count and length else np.empty(0, dtype=np.int64)
    starts = starts[np.concatenate([[True], np.diff(starts) >= length])] if starts.size else starts  # (no overlapping runs on a small corpus)
    run_rows = (starts[:, None] + np.arange(length, dtype=np.int64)[None, :]).reshape(-1)

    def make(q):
        b, e = shard_rows(n, shards, q)
        rng = np.random.RandomState(42 + q)
        kept, kept_runs = [], []
        for lo in range(b, e, piece):
            m = min(piece, e - lo)
            part = rng.randint(0, 2**32, size=(m, t), dtype=np.uint64)
            sel = rows[(rows >= lo) & (rows < lo + m)]
            kept.append(part[sel - lo].copy())
            sel = run_rows[(run_rows >= lo) & (run_rows < lo + m)]
            kept_runs.append(part[sel - lo].copy())
            d_tok.upload(part, offset=lo * t * 8)
        none = np.empty((0, t), dtype=np.uint64)
        return (np.concatenate(kept) if kept else none), (np.concatenate(kept_runs) if kept_runs else none)

    with ThreadPoolExecutor(max(1, min(shards, _usable_cores()))) as pool:
        made = list(pool.map(make, range(shards)))
    sample_tokens = np.concatenate([m[0] for m in made])
    if count and length:
        return d_tok, rows, sample_tokens, (starts, length, np.concatenate([m[1] for m in made]))
    return d_tok, rows, sample_tokens


def extra_full(ctx, n, seed, checks="sample", t=256, k=256, bands=32, r=8):
    """BASELINE.json configs[2] and [4] at their STATED size on one GPU (an 8-GPU job holds exactly this on every GPU after
    the all-gather): n = 10M sets x 256 tokens (2.56e9 tokens: past 2^31 elements in every kernel), num_perm = 256 ->
    uint32 signatures (10.2 GB) -> band digests (32 x 8) -> bucketing of 320M (band, digest) keys; b = 1 blocks + band
    digests of the same matrix from one read.  checks = "sample": the spread sample rows against the C oracle at every
    stage, four bands of the sorted output in full; "all": every band, the fused outputs against the two kernels'
    everywhere, the bucketing against the stable radix sort everywhere."""
    from datasketch_amd import _native, lsh_bulk
    from datasketch_amd.minhash import MinHash
    from oracle import oracle as O

    lib = ctx.lib
    t0 = time.perf_counter()
    n_runs = 250 if checks == "all" else 25  # x 4 000 consecutive sets: 1 000 000 rows ("all": the GPU test) or 100 000 (every bench run) against the oracle
    d_tok, rows, tok, (run_starts, run_len, run_tok) = full_corpus(ctx, n, t, blocks=(n_runs, 4000))
    gen_s = time.perf_counter() - t0
    perms = MinHash(num_perm=k, seed=seed, hashfunc=lambda x: x).permutations
    nb = k // 64
    d_sig = ctx.alloc(n * k * 4)
    ms_sig = _timed(ctx, lambda: ctx.minhash_bulk_dev(perms, d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_sig.ptr, _native.MHX_U32), reps=2, ramp=0.1)
    d_tok.free()
    d_dig, d_sd, d_sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
    BM = _native.BAND_MAJOR  # the digests [bands, n]: one array per hashtable, read by the bucketing with unit stride
    ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, bands, r, BM, d_dig.ptr)), reps=3, ramp=0.1)
    sort_dig = lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig.ptr, n, bands, BM, d_sd.ptr, d_sr.ptr))
    ms_sort = _timed(ctx, sort_dig, reps=3, ramp=0.1)
    # ---- parity, config 3
    a, b = perms
    want = O.c_minhash_bulk_dense(tok, a, b)
    if not np.array_equal(_download_rows(d_sig, rows, k, np.uint32).astype(np.uint64), want):
        raise SystemExit("PARITY FAILURE (extra.c3_full): signatures differ from the oracle")
    # ... and a million rows of them (runs of consecutive sets spread over the corpus): the C oracle on every host thread
    w0 = time.perf_counter()
    want_runs = O.c_minhash_bulk_dense_parallel(run_tok, a, b)
    for i, st in enumerate(run_starts):
        got = d_sig.download((run_len, k), np.uint32, offset=int(st) * k * 4)
        if not np.array_equal(got.astype(np.uint64), want_runs[i * run_len: (i + 1) * run_len]):
            raise SystemExit(f"PARITY FAILURE (extra.c3_full): signatures of rows {int(st)} .. {int(st) + run_len} differ from the oracle")
    runs_checked, runs_s = int(len(run_starts) * run_len), time.perf_counter() - w0
    del want_runs, run_tok
    want_dig = lsh_bulk.band_digests(want, bands, r, gpu_mode="disable")  # FNV-1a-64 of the reference's key bytes (lsh.py:537-538), numpy
    for i in range(8):
        keys = O.c_band_keys(want[i: i + 1], bands, r)
        if int(want_dig[i, bands - 1]) != _fnv1a64(keys[0, (bands - 1) * r:].tobytes()):
            raise SystemExit("PARITY FAILURE (extra.c3_full): the numpy digests differ from FNV-1a-64 of the key bytes")
    dig = d_dig.download((bands, n), np.uint64)
    if not np.array_equal(dig[:, rows].T, want_dig):
        raise SystemExit("PARITY FAILURE (extra.c3_full): band digests differ from FNV-1a-64 of the reference's key bytes")
    check_bands = list(range(bands)) if checks == "all" else sorted({0, bands // 3, 2 * bands // 3, bands - 1})
    for j in check_bands:
        sd = d_sd.download((n,), np.uint64, offset=j * n * 8)
        sr = d_sr.download((n,), np.uint32, offset=j * n * 4)
        col = dig[j]
        if np.any(sd[1:] < sd[:-1]) or not np.array_equal(col[sr.astype(np.int64)], sd):
            raise SystemExit(f"PARITY FAILURE (extra.c3_full): band {j} is not the band's digests in ascending order")
        tie = sd[1:] == sd[:-1]
        if np.any(sr[1:][tie] <= sr[:-1][tie]) or np.unique(sr).size != n:
            raise SystemExit(f"PARITY FAILURE (extra.c3_full): band {j}: rows not ascending inside a bucket, or not a permutation")
    radix = None
    if checks == "all":  # the whole output against the stable radix sort (the fallback path), every band
        sd_all, sr_all = d_sd.download((bands, n), np.uint64), d_sr.download((bands, n), np.uint32)
        ctx.set_option("lsh.sort", 1)
        try:
            radix = _timed(ctx, sort_dig, reps=1, ramp=0.0)
            same = np.array_equal(d_sd.download((bands, n), np.uint64), sd_all) and np.array_equal(d_sr.download((bands, n), np.uint32), sr_all)
        finally:
            ctx.set_option("lsh.sort", 0)
        del sd_all, sr_all
        if not same:
            raise SystemExit("PARITY FAILURE (extra.c3_full): the bucketing passes and the stable radix sort disagree")
    for d in (d_sd, d_sr):
        d.free()
    ctx.release_scratch()
    c3 = {
        "workload": f"config 3 at its stated size on one GPU: {n} sets x {t} tokens ({n * t:.3e} tokens), num_perm={k} (uint64 tokens in, uint32 signatures out), "
                    f"band digests ({bands} x {r}, band-major), bucketing of {n * bands} (band, digest) keys",
        "signatures": dict(_roof(n * (8 * t + 4 * k), ms_sig), signatures_per_s=n / (ms_sig * 1e-3)),
        "band_digests": _roof(n * (4 * k + 8 * bands), ms_dig),
        "lsh_sort_digests": dict(_roof(n * (8 * bands + 12 * bands), ms_sort), keys_per_s=n * bands / (ms_sort * 1e-3)),
        "pipeline_ms": ms_sig + ms_dig + ms_sort,
        "corpus_seconds_on_host": gen_s,
        "signature_rows_vs_oracle": runs_checked + len(rows),
        "parity": f"{runs_checked} rows in {len(run_starts)} runs of {run_len} consecutive sets spread over the corpus: signatures vs the C oracle ({runs_s:.1f} s on the host's threads); "
                  f"{len(rows)} rows spread over the corpus: signatures vs the C oracle, {bands} digests each vs FNV-1a-64 of the reference's key bytes; "
                  f"sorted bands {check_bands if checks != 'all' else 'all'}: ascending, equal to the digest column gathered by the sorted rows, rows ascending "
                  f"inside every bucket, a permutation" + ("; all bands equal to the stable radix sort's" if radix is not None else ""),
    }
    if radix is not None:
        c3["lsh_sort_digests_radix_ms"] = radix
    # ---- config 5: b = 1 blocks + band digests of the same 10M x 256 matrix
    d_blk, d_dig2 = ctx.alloc(n * nb * 8), ctx.alloc(n * bands * 8)
    fused_flag = []
    fused = lambda: fused_flag.append(ctx.bbit_pack_band_digests_dev(d_sig.ptr, _native.MHX_U32, n, k, 1, bands, r, d_blk.ptr, d_dig2.ptr, BM))
    ms_fused = _timed(ctx, fused, reps=3, ramp=0.1)
    blk_rows = _download_rows(d_blk, rows, nb, np.uint64)
    if not np.array_equal(blk_rows, O.c_bbit_pack(want, 1)):
        raise SystemExit("PARITY FAILURE (extra.c5_full): fused b=1 blocks differ from the oracle")
    if not np.array_equal(d_dig2.download((bands, n), np.uint64), dig):  # (dig: checked against the oracle on the sample rows above)
        raise SystemExit("PARITY FAILURE (extra.c5_full): the fused kernel's digests differ from band_digest_kernel's")
    del dig
    ms_pack = _timed(ctx, lambda: _native.check(lib.mhx_bbit_pack_dev_typed(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, 1, d_dig.ptr)), reps=3, ramp=0.1)  # (into d_dig: free by now)
    if checks == "all" and not np.array_equal(d_dig.download((n, nb), np.uint64), d_blk.download((n, nb), np.uint64)):
        raise SystemExit("PARITY FAILURE (extra.c5_full): the fused kernel's blocks differ from bbit1_wide_kernel's")
    c5 = {
        "workload": f"config 5 at its stated size on one GPU: b=1 packing of {n} x {k} signatures (uint32) + LSH band hashing ({bands} x {r})",
        "fused": dict(_roof(n * (4 * k + k // 8 + 8 * bands), ms_fused), one_read=bool(fused_flag and all(fused_flag)),
                      kernel="bbit_digest_fused_kernel: blocks and band-major digests from one read of the matrix"),
        "two_kernels_ms": ms_pack + ms_dig,
        "bbit_pack_b1_ms": ms_pack,
        "band_digests_ms": ms_dig,
        "pipeline_ms": ms_fused,
        "parity": f"{len(rows)} spread rows: blocks vs the C oracle (b_bit_minhash.py:82-101), digests vs FNV-1a-64 of the key bytes; all {n} x {bands} "
                  f"digests equal to band_digest_kernel's" + (f"; all {n} x {nb} blocks equal to bbit1_wide_kernel's" if checks == "all" else ""),
    }
    for d in (d_sig, d_dig, d_blk, d_dig2):
        d.free()
    return {"c3_full": c3, "c5_full": c5}


def extra_c4(ctx, n=100_000, dim=4096, s=128):
    """Config 4: WeightedMinHashGenerator(4096, 128, seed=1).minhash_many on X = RandomState(42).uniform(0, 100,
    (100k, 4096)) float32 -- kernel-only on resident input, and from Python (numpy in, numpy out) in parity mode
    (np.log on the host) and device-log mode; the device-log mode's (k, t) mismatches against parity mode are
    counted and gated as BASELINE.md section 3 prescribes."""
    import scipy.sparse as sp

    from datasketch_amd import WeightedMinHashGenerator, _native
    from oracle import oracle as O

    rs = np.random.RandomState(42)
    x = np.empty((n, dim), dtype=np.float32)
    for i in range(0, n, 10_000):  # the same stream as one call; bounds the float64 temporary
        x[i:i + 10_000] = rs.uniform(0, 100, (min(10_000, n - i), dim))
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always", device_log=False)  # np.log on the host, whatever the device could do
    gl = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always", device_log=True)
    ga = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")  # the default: the log on the device where it is numpy's bit for bit
    log_matches = bool(ctx.device_log_matches_numpy()) if hasattr(ctx, "device_log_matches_numpy") else False
    out = {"workload": f"config 4: {n} dense vectors x dim {dim}, sample_size {s}, float32 (the reference's arithmetic type)"}
    # from Python, parity mode and device-log mode
    g.minhash_many_arrays(x[:2048])
    t0 = time.perf_counter()
    hv, ne = g.minhash_many_arrays(x)
    dt_par_first = time.perf_counter() - t0  # takes the page-locked log buffers (kept on the generator) on top
    t0 = time.perf_counter()
    hv, ne = g.minhash_many_arrays(x)
    dt_par = time.perf_counter() - t0
    gl.minhash_many_arrays(x[:2048])
    t0 = time.perf_counter()
    hv_l, ne_l = gl.minhash_many_arrays(x)
    dt_log = time.perf_counter() - t0
    ga.minhash_many_arrays(x[:2048])
    t0 = time.perf_counter()
    hv_a, ne_a = ga.minhash_many_arrays(x)
    dt_auto = time.perf_counter() - t0
    if not (np.array_equal(hv_a, hv) and np.array_equal(ne_a, ne)):
        raise SystemExit("PARITY FAILURE (extra.c4): the default mode (log on the device after the start-up check) differs from the host-log results")
    # kernel only: logs resident on the device (the generator lives on the process-wide context: its stream is the
    # one the events must be recorded on)
    wctx, handle = g._device_handle()
    lib = wctx.lib
    with np.errstate(invalid="ignore", divide="ignore"):
        logs = np.log(x)
    d_x = wctx.to_device(logs)
    d_o = wctx.alloc(n * s * 16)
    d_ne = wctx.alloc(n)
    ms = _timed(wctx, lambda: _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 1, n, d_o.ptr, d_ne.ptr)), reps=5)
    hv_dev = d_o.download((n, s, 2), np.int64)  # the timed entry point's own result
    wctx.set_option("weighted.path", 2)  # A/B: every element evaluated (round 2's kernels), same call
    try:
        ms_every = _timed(wctx, lambda: _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 1, n, d_o.ptr, d_ne.ptr)), reps=1)
        hv_every = d_o.download((n, s, 2), np.int64)
    finally:
        wctx.set_option("weighted.path", 0)
    d_x.upload(x)
    ms_log = _timed(wctx, lambda: _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 0, n, d_o.ptr, d_ne.ptr)), reps=5)
    for d in (d_x, d_o, d_ne):
        d.free()
    # parity gates: 2 048 rows spread over the matrix against the C oracle (bit-exact (k, t) in parity mode), and EVERY
    # row of the walk against the kernels that evaluate every element
    rows = np.unique(np.linspace(0, n - 1, 2048).astype(np.int64))
    csr = sp.csr_matrix(x[rows])
    csr.sort_indices()
    t0 = time.perf_counter()
    wo, wn = O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)
    oracle_s = time.perf_counter() - t0
    if not (np.array_equal(hv[rows], wo) and np.array_equal(ne[rows], wn) and np.array_equal(hv_dev[rows], wo)):
        raise SystemExit("PARITY FAILURE (extra.c4): weighted (k, t) differ from the oracle in parity mode")
    if not np.array_equal(hv_dev, hv_every):
        raise SystemExit("PARITY FAILURE (extra.c4): the walk and the evaluate-every-element kernels disagree")
    cpu = weighted_cpu_baseline(x, g, wo[:64] if np.array_equal(rows[:64], np.arange(64)) else None, oracle_rows=len(rows), oracle_s=oracle_s)
    # fast-mode acceptance gate: every (k, t) mismatch must come from two smallest ln_a
    # within 1e-6 relative of each other
    mism = np.argwhere(np.any(hv != hv_l, axis=2))
    gate = weighted_gap_gate(x, g, hv, hv_l, mism[:5000])
    if gate["unexplained"]:
        raise SystemExit(f"PARITY FAILURE (extra.c4): {gate['unexplained']} device-log mismatches outside the 1e-6 ln_a tolerance")
    alg = n * (4 * dim + 16 * s)
    out.update({
        # the reference's function takes VALUES (weighted_minhash.py:212 takes np.log itself): that is config 4's primary number
        "kernel": dict(_roof(alg, ms_log), vectors_per_s=n / (ms_log * 1e-3), element_evaluations_per_s=n * dim * s / (ms_log * 1e-3),
                       kernels="walk_plan_kernel + walk_build_kernel (no-op once the tables stand) + weighted_walk_wave_kernel<values in>",
                       note="VALUES in, as the reference's minhash_many takes them: the device takes numpy's float32 log (np_logf) of the entries a walk "
                            "meets; bound-ordered walk: ~2 exact evaluations per (row, sample) instead of 4096 (the element rate counts the "
                            "evaluations the reference makes); the matrix is read once; algorithmic bytes = 4*dim + 16*S per vector"),
        "kernel_logs_in": dict(_roof(alg, ms), vectors_per_s=n / (ms * 1e-3),
                               note="the same with np.log of the matrix precomputed and resident (what the host-log parity mode hands over)"),
        "kernel_every_element": dict(_roof(alg, ms_every), vectors_per_s=n / (ms_every * 1e-3),
                                     note="weighted.path=2: round 2's kernels (every element evaluated), same call, same box"),
        "from_python_parity_mode": {"seconds": dt_auto, "vectors_per_s": n / dt_auto, "log_taken_on": "device" if log_matches else "host",
                                    "note": "numpy in -> numpy out, the default mode: (k, t) bit-identical to the reference; the log is taken on the device when "
                                            "its float32 log reproduces this host's np.log on the start-up sentinels (device_log_matches_numpy), else on the host; "
                                            "equal to the host-log results on all rows (checked above)"},
        "from_python_host_log": {"seconds": dt_par, "vectors_per_s": n / dt_par, "first_call_seconds": dt_par_first,
                                 "note": "device_log=False: np.log on the host; first call = with the one-time allocation of the page-locked log buffers"},
        "device_log_matches_numpy": log_matches,
        "from_python_device_log": {"seconds": dt_log, "vectors_per_s": n / dt_log},
        "device_log_mismatch_rate": float(len(mism)) / (n * s),
        "device_log_mismatches": int(len(mism)),
        "device_log_gate": gate,
        "cpu_baseline": cpu,
        "parity": f"{len(rows)} rows bit-exact (k, t) vs the C oracle in parity mode; all {n} rows equal to the evaluate-every-element "
                  f"kernels; all-rows nonempty = {bool(ne.all())}",
    })
    return out


def weighted_cpu_baseline(x, g, want64, oracle_rows, oracle_s):
    """The reference's CPU paths for config 4 on a bounded sample of its input, one core (numpy's float32 ufuncs are
    single-threaded): `minhash_many` as the reference evaluates it (weighted_minhash.py:205-239: per row the (S, nnz)
    arrays) through this package's gpu_mode='disable' path (the same numpy statements), the per-vector `minhash` loop
    (weighted_minhash.py:123-159; SURVEY.md section 8d: the reference's faster CPU alternative), and the scalar C
    oracle.  kind 'port': the reference itself is not on the GPU box."""
    from datasketch_amd import WeightedMinHashGenerator
    from oracle import oracle as O
    import scipy.sparse as sp

    gd = WeightedMinHashGenerator(g.dim, g.sample_size, seed=g.seed, gpu_mode="disable")
    m = 64
    t0 = time.perf_counter()
    res = gd.minhash_many(x[:m])
    dt_many = time.perf_counter() - t0
    if want64 is None:
        c = sp.csr_matrix(x[:m])
        c.sort_indices()
        want64 = O.c_weighted_minhash_many(c.indptr, c.indices, c.data, g.rs, g.ln_cs, g.betas)[0]
    if not np.array_equal(np.stack([r.hashvalues for r in res]), want64):
        raise SystemExit("PARITY FAILURE (extra.c4): the numpy path differs from the oracle on the cpu_baseline sample")
    t0 = time.perf_counter()
    for v in x[:m]:
        gd.minhash(v)
    dt_each = time.perf_counter() - t0
    return {"value": m / dt_many, "unit": "vectors/s", "cores": 1, "kind": "port",
            "sample": f"first {m} rows of config 4's input (dim {g.dim}, sample_size {g.sample_size}): minhash_many, numpy, {dt_many:.1f} s",
            "per_vector_minhash_loop_value": m / dt_each,
            "per_vector_minhash_loop_sample": f"the same {m} rows through minhash() one at a time, {dt_each:.1f} s",
            "c_oracle_value": oracle_rows / oracle_s, "c_oracle_sample": f"{oracle_rows} rows, scalar C, {oracle_s:.1f} s",
            "cpu_model": cpu_model()}


def weighted_gap_gate(x, g, hv_par, hv_log, mism, tol=1e-6):
    """BASELINE.md section 3's acceptance rule for the device-log mode: a (k, t) pair may differ from parity mode
    only where the choice was within rounding -- the two competing columns' ln_a (float32, the reference's formula,
    weighted_minhash.py:212-218) within `tol` relative of each other, or (same effect one step earlier) a column's
    ln(x)/r + beta within `tol` relative of an integer, where one ulp of the log moves the floor.  Returns the
    largest relative gap seen among the accepted mismatches and the number of mismatches neither rule explains."""
    worst_gap, worst_edge, unexplained, edge_cases = None, None, 0, 0
    one = np.float32(1)
    for row, smp in mism:
        k0, k1 = int(hv_par[row, smp, 0]), int(hv_log[row, smp, 0])
        ln_a, edge = [], []
        for kk in (k0, k1):
            lg = np.log(np.float32(x[row, kk]))
            r, be, lc = g.rs[smp, kk], g.betas[smp, kk], g.ln_cs[smp, kk]
            y = np.float32(np.float32(lg / r) + be)
            tt = np.floor(y)
            ln_a.append(float(np.float32(lc - np.float32(np.float32(np.float32(tt - be) + one) * r))))
            edge.append(float(min(y - tt, tt + one - y)) / max(abs(float(y)), 1.0))
        if min(edge) <= tol:  # a floor boundary: t (and with it ln_a) flips with the last bit of the log
            edge_cases += 1
            worst_edge = min(edge) if worst_edge is None else max(worst_edge, min(edge))
            continue
        gap = abs(ln_a[0] - ln_a[1]) / max(abs(ln_a[0]), abs(ln_a[1]), 1e-30)
        if k0 != k1 and gap <= tol:
            worst_gap = gap if worst_gap is None else max(worst_gap, gap)
        else:
            unexplained += 1
    return {"mismatches_examined": int(len(mism)), "worst_relative_ln_a_gap": worst_gap, "floor_boundary_cases": edge_cases,
            "worst_floor_boundary_distance": worst_edge, "unexplained": unexplained, "tolerance": tol,
            "rule": "BASELINE.md section 3: (k,t) may differ from parity mode only where the two smallest ln_a "
                    "(or ln(x)/r+beta and an integer) are within 1e-6 relative"}



def extra_c4_sparse(ctx, n=100_000, dim=4096, s=128, density=0.01):
    """SURVEY.md 8d's "1 %-dense CSR variant" of config 4: the form the reference natively takes (weighted_minhash.py:192-203 converts
    whatever it is given to CSR).  100k rows x ~41 stored entries of 4096 columns, sample_size 128, float32; logs resident (parity
    mode's hand-over).  Rows this short are evaluated entry by entry (weighted_csr_direct_kernel): nnz x S exact evaluations, each
    one a 16-byte table entry from the L2 -- the bound is VALU issue and L2 -> CU bandwidth, not HBM, and the roofline record says so."""
    import scipy.sparse as sp

    from datasketch_amd import WeightedMinHashGenerator, _native
    from oracle import oracle as O

    rng = np.random.RandomState(42)
    nnz_row = max(1, int(round(dim * density)))
    # every row stores nnz_row distinct, ascending columns: one drawn uniformly from each of nnz_row equal strata of the columns
    indptr = np.arange(n + 1, dtype=np.int64) * nnz_row
    width = dim // nnz_row
    indices = (np.arange(nnz_row, dtype=np.int32) * width + rng.randint(0, width, (n, nnz_row)).astype(np.int32)).reshape(-1)
    data = rng.uniform(0, 100, n * nnz_row).astype(np.float32)
    data[data == 0] = 1.0
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    # the generator lives on the process-wide context: its stream is the one the events must be recorded on (as in extra_c4)
    ctx, handle = g._device_handle()
    lib = ctx.lib
    logs = np.log(data)
    d_ptr, d_idx, d_log, d_val = ctx.to_device(indptr), ctx.to_device(indices), ctx.to_device(logs), ctx.to_device(data)
    d_out, d_ne = ctx.alloc(n * s * 16), ctx.alloc(n)
    nnz = int(indices.size)
    run_logs = lambda: _native.check(lib.mhx_weighted_minhash_many_dev(handle, d_ptr.ptr, d_idx.ptr, d_log.ptr, 1, n, nnz, d_out.ptr, d_ne.ptr))
    run_vals = lambda: _native.check(lib.mhx_weighted_minhash_many_dev(handle, d_ptr.ptr, d_idx.ptr, d_val.ptr, 0, n, nnz, d_out.ptr, d_ne.ptr))
    ms_logs = _timed(ctx, run_logs, reps=5)
    hv = d_out.download((n, s, 2), np.int64)
    rows = np.unique(np.linspace(0, n - 1, 2048).astype(np.int64))
    sub = sp.csr_matrix((data.reshape(n, nnz_row)[rows].reshape(-1), indices.reshape(n, nnz_row)[rows].reshape(-1), np.arange(len(rows) + 1) * nnz_row), shape=(len(rows), dim))
    wo, wn = O.c_weighted_minhash_many(sub.indptr, sub.indices, sub.data, g.rs, g.ln_cs, g.betas)
    if not np.array_equal(hv[rows], wo) or not np.all(d_ne.download((n,), np.uint8) == 1):
        raise SystemExit("PARITY FAILURE (extra.c4_sparse): (k, t) differ from the C oracle")
    ms_vals = _timed(ctx, run_vals, reps=5)
    log_on_device = bool(g._log_on_device(ctx))
    if log_on_device and not np.array_equal(d_out.download((n, s, 2), np.int64), hv):
        raise SystemExit("PARITY FAILURE (extra.c4_sparse): values in (device log) differs from logs in on a host whose np.log the device log reproduces")
    alg = 4 * nnz + 16 * s * n  # 4 * nnz_row + 16 * S bytes per vector (float32 values in; the column indices are 4 more bytes per entry)
    evals = nnz * s
    table_bytes = evals * 16
    return {
        "workload": f"config 4, the 1 %-dense CSR variant (SURVEY.md 8d): {n} rows x {nnz_row} stored entries of {dim} columns, sample_size {s}, float32",
        "kernel": dict(_roof(alg, ms_logs), vectors_per_s=n / (ms_logs * 1e-3), exact_evaluations_per_s=evals / (ms_logs * 1e-3),
                       table_GBps_from_l2=table_bytes / (ms_logs * 1e-3) / 1e9,
                       kernels="weighted_csr_direct_kernel (+ the plan / walk launches, which return at once: no row of this call is long enough to walk)",
                       bound="not HBM: every (entry, sample) is one exact evaluation -- 17 VALU instructions and a 16-byte table entry {r, ln_c, beta, 1/r} gathered "
                             "from the L2 (the 8.4 MB table does not fit a 4 MB L2 slice whole: 14 % of the requests miss); `frac` prices the algorithmic bytes "
                             "(4 nnz + 16 S per vector) against 8 TB/s as the contract asks and is not what binds this kernel -- exact_evaluations_per_s against "
                             "the chip's VALU issue (1024 SIMDs x 2.4 GHz / 4 cycles / 17 instructions x 64 lanes = 2.3e12 /s) and table_GBps_from_l2 against "
                             "64 B/clk/CU (39 TB/s) are"),
        "kernel_values_in": dict(_roof(alg, ms_vals), note="values in: the device takes numpy's float32 log of the nnz entries first (one more launch over 4 * nnz bytes)"),
        "parity": f"{len(rows)} rows bit-exact (k, t) vs the C oracle; values-in equal to logs-in on all {n} rows: {log_on_device}",
    }
