"""GPU parity tests added in round 2 (run on an MI355X: python -m pytest tests -m gpu -x -q).

uint32 host entry, sha1_hash64 bulk, uint32 signature matrices through the pack / digest / sort kernels, bulk LSH
query on the device, weighted fast mode against the stated tolerance, full-size K=256 and config-4 input,
context sharing between threads, and the N>1 launch path of bench.py.  Everything goes through the C ABI; the
oracle / hashlib / brute force are the checkers.
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import scipy.sparse as sp

from datasketch_amd import MinHash, WeightedMinHashGenerator, _native, prehashed
from datasketch_amd import lsh_bulk as LB
from datasketch_amd.hashfunc import sha1_hash32, sha1_hash64
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    assert _native.gpu_available(), "these tests need an MI355X"
    return _native.context()


# ------------------------------------------------------------------ uint32 host entry, out_dtype
@pytest.mark.parametrize("chunk_bytes", [-1, 1 << 20])
def test_host_entry_with_uint32_tokens_and_signatures(ctx, chunk_bytes):
    rng = np.random.RandomState(11)
    n, t, k = 5000, 96, 128
    tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
    a, b = O.np_init_permutations(k, 1)
    want = O.c_minhash_bulk_dense(tok, a, b)
    ctx.set_option("host.chunk_bytes", chunk_bytes)
    try:
        tok32 = tok.astype(np.uint32)
        for out_dtype in (np.uint64, np.uint32):
            got = ctx.minhash_bulk((a, b), tok32.reshape(-1), None, t, n, None, out_dtype=out_dtype)
            assert got.dtype == out_dtype and np.array_equal(got.astype(np.uint64), want)
        got = ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n, None, out_dtype=np.uint32)
        assert got.dtype == np.uint32 and np.array_equal(got.astype(np.uint64), want)
        # ragged CSR with uint32 tokens and an initial state
        lens = rng.randint(0, 70, size=n)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        hv = rng.randint(0, 2**32, int(offsets[-1]), dtype=np.uint64)
        init = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
        want_r = O.c_minhash_bulk(hv, offsets, a, b, init)
        got_r = ctx.minhash_bulk((a, b), hv.astype(np.uint32), offsets, 0, n, init, out_dtype=np.uint32)
        assert np.array_equal(got_r.astype(np.uint64), want_r)
    finally:
        ctx.set_option("host.chunk_bytes", 0)
    sig32 = MinHash.bulk_signatures(tok32, num_perm=k, seed=1, hashfunc=prehashed, gpu_mode="always", out_dtype=np.uint32)
    assert sig32.dtype == np.uint32 and np.array_equal(sig32.astype(np.uint64), want)
    with pytest.raises(ValueError):
        MinHash.bulk_signatures(tok32, num_perm=k, seed=1, hashfunc=prehashed, gpu_mode="always", out_dtype=np.int32)


def test_bulk_with_sha1_hash64_runs_on_the_device(ctx):
    """MinHash.bulk with hashfunc=sha1_hash64 (ref: hashfunc.py:17-28): bytes -> SHA-1 (8 bytes) -> signatures on
    the device, equal to hashing every token with hashlib on the host."""
    rng = np.random.RandomState(5)
    sets = [[bytes(rng.randint(0, 256, rng.randint(0, 40), dtype=np.uint8)) for _ in range(rng.randint(0, 60))] for _ in range(400)]
    sets[7] = []
    dev = MinHash.bulk_signatures(sets, num_perm=96, seed=4, hashfunc=sha1_hash64, gpu_mode="always")
    host = MinHash.bulk_signatures(sets, num_perm=96, seed=4, hashfunc=sha1_hash64, gpu_mode="disable")
    assert np.array_equal(dev, host)
    one = [int.from_bytes(hashlib.sha1(tk).digest()[:8], "little") for tk in sets[3]]
    a, b = O.np_init_permutations(96, 4)
    assert np.array_equal(dev[3], O.np_minhash_bulk([np.array(one, dtype=np.uint64)], a, b)[0])
    dev32 = MinHash.bulk_signatures(sets, num_perm=96, seed=4, hashfunc=sha1_hash32, gpu_mode="always")
    assert np.array_equal(dev32, MinHash.bulk_signatures(sets, num_perm=96, seed=4, hashfunc=sha1_hash32, gpu_mode="disable"))
    assert not np.array_equal(dev32, dev)


# ------------------------------------------------------------------ uint32 signature matrices downstream
def test_uint32_signatures_through_pack_digest_sort(ctx):
    rng = np.random.RandomState(2)
    n, k, bands, r = 20_000, 256, 32, 8
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    sig[100:120] = sig[0:20]
    d64, d32 = ctx.to_device(sig), ctx.to_device(sig.astype(np.uint32))
    lib = ctx.lib
    for b in (1, 2, 3, 8, 13, 32):
        nb = ctypes.c_int32(0)
        _native.check(lib.mhx_bbit_num_blocks(k, b, ctypes.byref(nb)))
        out = ctx.alloc(n * nb.value * 8)
        _native.check(lib.mhx_bbit_pack_dev_typed(ctx.handle, d32.ptr, _native.MHX_U32, n, k, b, out.ptr))
        ctx.synchronize()
        assert np.array_equal(out.download((n, nb.value), np.uint64), O.c_bbit_pack(sig, b)), b
    for (bb, rr) in ((bands, r), (9, 7), (64, 4), (3, 2)):
        dg32, dg64 = ctx.alloc(n * bb * 8), ctx.alloc(n * bb * 8)
        _native.check(lib.mhx_band_digests_dev_typed(ctx.handle, d32.ptr, _native.MHX_U32, n, k, bb, rr, dg32.ptr))
        _native.check(lib.mhx_band_digests_dev_typed(ctx.handle, d64.ptr, _native.MHX_U64, n, k, bb, rr, dg64.ptr))
        ctx.synchronize()
        got = dg32.download((n, bb), np.uint64)
        assert np.array_equal(got, dg64.download((n, bb), np.uint64))
        assert np.array_equal(got[:50], LB.band_digests(sig[:50], bb, rr, gpu_mode="disable"))
    sd, sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
    _native.check(lib.mhx_lsh_sort_bands_dev_typed(ctx.handle, d32.ptr, _native.MHX_U32, n, k, bands, r, sd.ptr, sr.ptr))
    ctx.synchronize()
    want_d, want_r = ctx.lsh_sort_bands(sig, bands, r)
    assert np.array_equal(sd.download((bands, n), np.uint64), want_d) and np.array_equal(sr.download((bands, n), np.uint32), want_r)
    pairs = np.stack([rng.randint(0, n, 500), rng.randint(0, n, 500)], axis=1).astype(np.int64)
    pairs[:20, 0], pairs[:20, 1] = np.arange(20), np.arange(100, 120)
    d_p, d_c = ctx.to_device(pairs), ctx.alloc(500 * 4)
    _native.check(lib.mhx_jaccard_pairs_dev_typed(ctx.handle, d32.ptr, d32.ptr, _native.MHX_U32, k, d_p.ptr, 500, d_c.ptr))
    ctx.synchronize()
    cnt = d_c.download((500,), np.int32)
    assert np.array_equal(cnt, np.count_nonzero(sig[pairs[:, 0]] == sig[pairs[:, 1]], axis=1)) and cnt[:20].tolist() == [k] * 20
    with pytest.raises(ValueError):
        _native.check(lib.mhx_bbit_pack_dev_typed(ctx.handle, d32.ptr, 7, n, k, 1, sd.ptr))


@pytest.mark.parametrize("k", [128, 256, 384, 512, 768, 1024, 200])
def test_bbit_packing_with_wide_loads(ctx, k):
    """Every slot width takes 16-byte loads when a row is whole groups of 64 * V values (V = 4 uint32 / 2 uint64 per lane): every
    such K, the K that fall back, odd row counts, against the oracle's bit order (b_bit_minhash.py:82-97)."""
    rng = np.random.RandomState(k)
    n = 3001
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    sig[5] = 0
    sig[6] = 2**32 - 1
    sig[7, ::2] |= 1
    sig[7, 1::2] &= ~np.uint64(1)
    for b in (1, 2, 3, 4, 7, 8, 11, 16, 17, 32):  # slots of 1, 2, 4, 8, 16 and 32 bits
        want = O.c_bbit_pack(sig, b)
        nb = want.shape[1]
        for dtype, code in ((np.uint64, _native.MHX_U64), (np.uint32, _native.MHX_U32)):
            d_sig, out = ctx.to_device(sig.astype(dtype)), ctx.alloc(n * nb * 8)
            _native.check(ctx.lib.mhx_bbit_pack_dev_typed(ctx.handle, d_sig.ptr, code, n, k, b, out.ptr))
            ctx.synchronize()
            assert np.array_equal(out.download((n, nb), np.uint64), want), (b, dtype)


# ------------------------------------------------------------------ bulk query on the device
def _brute_query(index, probes, bands, r):
    ib = index[:, : bands * r].reshape(index.shape[0], bands, r)
    res = []
    for q in probes:
        qb = q[: bands * r].reshape(bands, r)
        res.append(np.flatnonzero(np.any(np.all(ib == qb[None], axis=2), axis=1)))
    return res


@pytest.mark.parametrize("dtype", [np.uint64, np.uint32])
@pytest.mark.parametrize("n,bands,r", [(4000, 16, 4), (50_000, 32, 8), (300, 5, 3)])
def test_sorted_bands_index_query_is_the_reference_query(ctx, dtype, n, bands, r):
    """What MinHashLSH.query returns (ref: lsh.py:423-431: union over bands of the bucket of the probe's band key),
    for a whole matrix of probes: checked against brute force over the band values themselves."""
    rng = np.random.RandomState(n + bands)
    k = 256 if bands * r > 64 else 64
    index = rng.randint(0, 2**32, (n, k)).astype(np.uint64)
    index[n // 2 : n // 2 + 40] = index[:40]                     # identical rows
    index[n // 3 : n // 3 + 64, : 2 * r] = index[7, : 2 * r]     # a 65-row bucket in bands 0 and 1
    m = 600
    probes = index[rng.randint(0, n, m)].copy()
    probes[::2, rng.randint(0, k, 12)] = 5                       # damage some bands
    probes[5::40] = rng.randint(0, 2**32, (len(probes[5::40]), k))
    idx = LB.SortedBandsIndex(index.astype(dtype), bands, r)
    offsets, rows = idx.query(probes.astype(dtype), capacity=16)  # tiny capacity: the retry path
    want = _brute_query(index, probes, bands, r)
    assert offsets[0] == 0 and offsets[-1] == rows.size
    for i in range(m):
        assert np.array_equal(rows[offsets[i] : offsets[i + 1]], want[i]), i
    assert max(len(w) for w in want) >= 65 and any(len(w) == 0 for w in want)
    off2, rows2 = idx.query(probes[:0].astype(dtype))
    assert off2.tolist() == [0] and rows2.size == 0
    with pytest.raises(ValueError):
        idx.query(probes[:, :8].astype(dtype))


@pytest.mark.parametrize("dtype", [np.uint64, np.uint32])
def test_sorted_bands_index_grown_in_batches(ctx, dtype):
    """SortedBandsIndex.extend: an index grown in three batches (the first one empty) answers like the index built
    from the whole matrix and like the brute-force band comparison; row numbers continue across batches."""
    rng = np.random.RandomState(17)
    n, k, bands, r = 9000, 64, 16, 4
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64).astype(dtype)
    sig[5000:5200, :32] = sig[100:300, :32]      # rows of the second batch share half their bands with rows of the first
    sig[8000:8100] = sig[6000:6100]              # duplicates inside the last batch's span
    probes = np.concatenate([sig[rng.randint(0, n, 300)], rng.randint(0, 2**32, (50, k), dtype=np.uint64).astype(dtype)])
    whole = LB.SortedBandsIndex(sig, bands, r)
    grown = LB.SortedBandsIndex(sig[:0], bands, r)
    assert list(grown.extend(sig[:4000])) == list(range(0, 4000))
    assert grown.extend(sig[4000:4000]) == range(4000, 4000)
    assert grown.extend(sig[4000:7500]) == range(4000, 7500)
    assert grown.extend(sig[7500:]) == range(7500, n) and grown.n == n
    off_w, rows_w = whole.query(probes)
    off_g, rows_g = grown.query(probes)
    assert np.array_equal(off_w, off_g) and np.array_equal(rows_w, rows_g)
    brute = _brute_query(sig.astype(np.uint64), probes.astype(np.uint64), bands, r)
    for i in (0, 7, 150, 299, 320):
        assert np.array_equal(rows_g[off_g[i] : off_g[i + 1]], brute[i])
    with pytest.raises(ValueError):
        grown.extend(sig[:10, :32])


def test_query_verification_rejects_digest_only_matches(ctx):
    """A probe is located by its 64-bit band digest; with the index matrix given, a candidate counts only if the band's
    values are equal.  Forced here by handing the kernel sorted bands of a DIFFERENT matrix than the one it verifies
    against: every digest match is then a 'collision' and must be dropped."""
    rng = np.random.RandomState(0)
    n, k, bands, r = 2000, 64, 8, 8
    a = rng.randint(0, 2**32, (n, k)).astype(np.uint64)
    other = rng.randint(0, 2**32, (n, k)).astype(np.uint64)
    d_a, d_other = ctx.to_device(a), ctx.to_device(other)
    sd, sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
    lib = ctx.lib
    _native.check(lib.mhx_lsh_sort_bands_dev_typed(ctx.handle, d_a.ptr, _native.MHX_U64, n, k, bands, r, sd.ptr, sr.ptr))
    d_pairs = ctx.alloc(n * bands * 16)
    found = ctypes.c_int64(-1)
    _native.check(lib.mhx_lsh_query_dev(ctx.handle, sd.ptr, sr.ptr, n, bands, r, d_a.ptr, None, _native.MHX_U64, k, n, d_pairs.ptr, n * bands, ctypes.byref(found)))
    assert found.value == n  # unverified: every row finds itself
    pairs = d_pairs.download((n, 2), np.int64)
    assert np.array_equal(pairs[:, 0], np.arange(n)) and np.array_equal(pairs[:, 1], np.arange(n))
    _native.check(lib.mhx_lsh_query_dev(ctx.handle, sd.ptr, sr.ptr, n, bands, r, d_a.ptr, d_other.ptr, _native.MHX_U64, k, n, d_pairs.ptr, n * bands, ctypes.byref(found)))
    assert found.value == 0  # verified against a matrix whose bands differ: nothing survives


# ------------------------------------------------------------------ weighted fast mode: the stated tolerance
def test_device_log_is_within_1e6_relative_of_numpy(ctx):
    """north_star: "within 1e-6 for WeightedMinHash's gamma/log draws".  The device-log mode's only deviation from the
    reference is its float32 log (ref: weighted_minhash.py:212 takes np.log): check it over config 4's input range
    (uniform(0, 100)), around 1 where the log crosses zero, and over the float32 range."""
    rng = np.random.RandomState(42)
    xs = [rng.uniform(0, 100, 2_000_000).astype(np.float32),
          (1 + rng.uniform(-1e-3, 1e-3, 200_000)).astype(np.float32),
          np.exp(rng.uniform(-80, 80, 200_000)).astype(np.float32),
          np.array([1.0, 2.0, 0.5, 100.0, np.float32(1e-38), np.float32(3e38)], dtype=np.float32)]
    for x in xs:
        x = x[x > 0]
        dev = ctx.weighted_logf(x)
        ref = np.log(x)
        exact = ref == 0
        assert np.array_equal(dev[exact], ref[exact])
        rel = np.abs(dev[~exact].astype(np.float64) - ref[~exact]) / np.abs(ref[~exact].astype(np.float64))
        assert rel.max() <= 1e-6, rel.max()
    sp_in = np.array([0.0, np.inf, -1.0, np.nan], dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        want = np.log(sp_in)
    got = ctx.weighted_logf(sp_in)
    assert got[0] == want[0] and got[1] == want[1] and np.isnan(got[2]) and np.isnan(got[3])


def test_device_log_mode_mismatches_pass_the_acceptance_gate(ctx):
    """in fast mode a (k, t) pair may differ from parity mode only where the two smallest ln_a
    are within 1e-6 relative (or, one step earlier, ln(x)/r + beta within 1e-6 of an integer).  4000 x 4096 vectors
    of config 4's distribution; the mismatch rate is reported by bench.py's extra.c4 at full size."""
    sys.path.insert(0, ROOT)
    import bench

    rng = np.random.RandomState(42)
    n, dim, s = 4000, 4096, 128
    x = rng.uniform(0, 100, (n, dim)).astype(np.float32)
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    gl = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always", device_log=True)
    hv, ne = g.minhash_many_arrays(x)
    hv_l, ne_l = gl.minhash_many_arrays(x)
    assert ne.all() and ne_l.all()
    mism = np.argwhere(np.any(hv != hv_l, axis=2))
    rate = len(mism) / (n * s)
    assert rate < 1e-4, rate
    gate = bench.weighted_gap_gate(x, g, hv, hv_l, mism)
    assert gate["unexplained"] == 0, gate
    # the gate itself must bite: a fabricated mismatch between two columns that are far apart is reported
    fake = hv.copy()
    fake[0, 0, 0] = (hv[0, 0, 0] + 1) % dim
    bad = bench.weighted_gap_gate(x, g, hv, fake, np.array([[0, 0]]))
    assert bad["unexplained"] == 1


# ------------------------------------------------------------------ full size: K = 256 and config 4's own input
def test_full_size_k256_1m_sets(ctx):
    """Config 3's kernel shape at full per-call size (1M x 256, K=256: the four-permutations-per-lane sieve variant):
    oracle rows over the whole matrix, idempotence, the union identity, uint32 output equal to uint64."""
    n, t, k = 1_000_000, 256, 256
    tok = np.random.RandomState(43).randint(0, 2**32, (n, t), dtype=np.uint64)
    a, b = O.np_init_permutations(k, 1)
    d_tok = ctx.to_device(tok)
    d_sig = ctx.alloc(n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_sig.ptr, _native.MHX_U64)
    ctx.synchronize()
    sig = d_sig.download((n, k), np.uint64)
    rows = np.unique(np.concatenate([np.arange(0, 2048), np.linspace(0, n - 1, 4096).astype(np.int64), np.arange(n - 2048, n)]))
    assert len(rows) >= 8000
    assert np.array_equal(sig[rows], O.c_minhash_bulk_dense(tok[rows], a, b))
    assert int(sig.max()) < 2**32
    d_32 = ctx.alloc(n * k * 4)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_32.ptr, _native.MHX_U32)
    ctx.synchronize()
    assert np.array_equal(d_32.download((n, k), np.uint32), sig.astype(np.uint32))
    d_32.free()
    d_again = ctx.alloc(n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, d_sig.ptr, k, d_again.ptr, _native.MHX_U64)
    ctx.synchronize()
    assert np.array_equal(d_again.download((n, k), np.uint64), sig)
    d_again.free()
    half = t // 2
    d_halves = ctx.alloc(2 * n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, half, 2 * n, n * t, None, 0, d_halves.ptr, _native.MHX_U64)
    ctx.synchronize()
    halves = d_halves.download((n, 2, k), np.uint64)
    assert np.array_equal(np.minimum(halves[:, 0], halves[:, 1]), sig)


def test_full_size_weighted_config4_input(ctx):
    """BASELINE.json configs[3] on ITS input: RandomState(42).uniform(0, 100, (100k, 4096)) float32, generator
    (4096, 128, seed=1): oracle rows over the matrix (bit-exact (k, t) in parity mode), row-order equivariance."""
    n, dim, s = 100_000, 4096, 128
    rs_ = np.random.RandomState(42)
    x = np.empty((n, dim), dtype=np.float32)
    for i in range(0, n, 10_000):
        x[i : i + 10_000] = rs_.uniform(0, 100, (10_000, dim))
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    out, ne = g.minhash_many_arrays(x)
    assert ne.all()
    rows = np.unique(np.concatenate([np.arange(0, 16), np.linspace(0, n - 1, 40).astype(np.int64), np.arange(n - 16, n)]))
    csr = sp.csr_matrix(x[rows])
    csr.sort_indices()
    wo, wn = O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)
    assert wn.all() and np.array_equal(out[rows], wo)
    rev, _ = g.minhash_many_arrays(x[::-1])
    assert np.array_equal(rev[::-1], out)


def _weighted_oracle_rows(g, x, rows):
    csr = sp.csr_matrix(x[rows])
    csr.sort_indices()
    return O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)


def test_page_locked_host_arrays(ctx):
    """mhx_host_alloc through Context.pinned_empty: an ordinary numpy array as far as numpy is concerned, accepted by
    the host entry points, released with its last view."""
    import gc

    x = ctx.pinned_empty((1000, 96), np.float32)
    assert x.shape == (1000, 96) and x.dtype == np.float32 and x.flags.c_contiguous and x.flags.writeable
    x[:] = np.random.RandomState(3).uniform(1, 9, x.shape)
    d = ctx.to_device(x)
    assert np.array_equal(d.download(x.shape, np.float32), x)
    view = x[10:20]
    del x
    gc.collect()
    assert float(view.sum()) > 0  # the buffer lives as long as a view does
    del view
    gc.collect()
    assert ctx.pinned_empty((0,), np.uint8).size == 0


@pytest.mark.parametrize("values_are_logs", [True, False])
def test_weighted_dense_feed_in_ragged_pieces(ctx, values_are_logs):
    """mhx_weighted_dense_begin / feed / end: pieces of different sizes (one of a single row, one full, a partial last
    one), rows without entries, buffers overwritten right after feed returns -- the same (k, t) as the one-call entry
    and, in parity mode, as the oracle."""
    n, dim, s, piece = 3517, 96, 40, 1000
    rs_ = np.random.RandomState(5)
    x = rs_.uniform(0, 50, (n, dim)).astype(np.float32)
    x[rs_.rand(n, dim) < 0.6] = 0
    x[[0, 17, 1000, n - 1]] = 0  # rows without entries, also first / last of a piece
    g = WeightedMinHashGenerator(dim, s, seed=3, gpu_mode="always")
    gctx, handle = g._device_handle()
    with np.errstate(divide="ignore"):
        values = np.log(x) if values_are_logs else x
    want, want_ne = gctx.weighted_minhash_many_dense(handle, s, values, values_are_logs)
    out = np.full((n, s, 2), -7, dtype=np.int64)
    ne = np.full(n, 9, dtype=np.uint8)
    cuts = [0, 1, 1001, 1500, 2500, 3400, n]
    buf = np.empty((piece, dim), dtype=np.float32)
    with gctx.weighted_dense_feed(handle, s, dim, values_are_logs, piece) as feed:
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            buf[: hi - lo] = values[lo:hi]
            feed.feed(buf[: hi - lo], out[lo:hi], ne[lo:hi])
            buf[:] = np.nan  # the piece is on the device: the caller's buffer is free
        with pytest.raises(ValueError, match="at most 1000 rows"):
            feed.feed(np.zeros((piece + 1, dim), dtype=np.float32), np.zeros((piece + 1, s, 2), dtype=np.int64), np.zeros(piece + 1, dtype=np.uint8))
    assert np.array_equal(ne.view(bool), want_ne) and np.array_equal(out, want)
    assert not ne[[0, 17, 1000, n - 1]].any() and ne.sum() == n - 4
    if values_are_logs:
        rows = np.arange(0, n, 7)
        wo, wn = _weighted_oracle_rows(g, x, rows)
        assert np.array_equal(wn, want_ne[rows]) and np.array_equal(out[rows][wn], wo[wn])


def test_weighted_one_call_and_python_paths_in_pieces(ctx):
    """A matrix of more than two pieces (64 MiB each): the one-call host entry pipelines internally, the Python
    parity path takes np.log of piece i+1 in threads while piece i is fed -- both equal the unpipelined call
    (option host.chunk_bytes = -1) and the oracle on a sample of rows."""
    n, dim, s = 300_000, 64, 16
    rs_ = np.random.RandomState(11)
    x = rs_.uniform(0, 100, (n, dim)).astype(np.float32)
    x[rs_.rand(n, dim) < 0.3] = 0
    x[[5, 131072, n - 1]] = 0
    g = WeightedMinHashGenerator(dim, s, seed=2, gpu_mode="always")
    assert n >= 2 * max(4096, ((64 << 20) // (4 * dim + 16 * s)) & ~7)
    got, got_ne = g.minhash_many_arrays(x)  # Python: threads + feed
    again, _ = g.minhash_many_arrays(x)     # the kept log buffers are reused
    gctx, handle = g._device_handle()
    with np.errstate(divide="ignore"):
        logs = np.log(x)
    piped, piped_ne = gctx.weighted_minhash_many_dense(handle, s, logs, True)
    gctx.set_option("host.chunk_bytes", -1)
    try:
        plain, plain_ne = gctx.weighted_minhash_many_dense(handle, s, logs, True)
    finally:
        gctx.set_option("host.chunk_bytes", 0)
    assert np.array_equal(plain_ne, piped_ne) and np.array_equal(plain_ne, got_ne)
    assert np.array_equal(plain, piped) and np.array_equal(plain, got) and np.array_equal(plain, again)
    assert not got_ne[[5, 131072, n - 1]].any()
    rows = np.unique(np.concatenate([np.arange(0, 64), rs_.randint(0, n, 400), np.arange(131072 - 8, 131072 + 8), np.arange(n - 64, n)]))
    wo, wn = _weighted_oracle_rows(g, x, rows)
    assert np.array_equal(wn, got_ne[rows]) and np.array_equal(got[rows][wn], wo[wn])
    g.release_buffers()


# ------------------------------------------------------------------ the second launch's tie-tolerant proof
@pytest.mark.parametrize("k,tok_dtype", [(128, np.uint64), (64, np.uint64), (256, np.uint64), (128, np.uint32), (200, np.uint32)])
def test_repeated_tokens_in_every_position_pattern(ctx, k, tok_dtype):
    """Sets with one repeated token (same row, another row, the ragged tail, another 256-token block), two
    different repeated tokens, a token occurring three and four times, near-tied keys -- what the first launch's
    uniqueness proof refuses.  The second launch settles the pairs with its tie-tolerant proof (two candidates
    hashed exactly) and everything else with the dedup pass; all of it must equal the oracle bit for bit."""
    rng = np.random.RandomState(21)
    hi = 2**32
    sets = []

    def fresh(t):
        return rng.randint(0, hi, t, dtype=np.uint64)

    for t in (16, 32, 47, 256, 300, 600):
        for rep in range(12):
            base = fresh(t)
            i, j = rng.choice(t, 2, replace=False)
            v = base.copy(); v[j] = v[i]; sets.append(v)                        # a pair anywhere
            v = base.copy(); r = (i // 16) * 16; v[r + (i + 1) % min(16, t - r) if t - r >= 16 else j] = v[i]; sets.append(v)  # same row where there is one
            if t >= 48:
                v = base.copy(); v[(i + 16) % t] = v[i]; sets.append(v)         # next row, same column
                i2, j2 = rng.choice(t, 2, replace=False)
                v = base.copy(); v[j] = v[i]; v[j2] = v[i2]; sets.append(v)     # two repeated tokens
                v = base.copy(); v[j] = v[i]; v[(j + 7) % t] = v[i]; sets.append(v)   # three times
                v = base.copy(); v[rng.choice(t, 4, replace=False)] = v[i]; sets.append(v)  # four times
            if t > 256:
                v = base.copy(); v[256 + (i % (t - 256))] = v[i % 256]; sets.append(v)  # across blocks
            if t % 16:
                v = base.copy(); v[t - 1] = v[i]; sets.append(v)                # repeat in the ragged tail
    sets += [np.repeat(fresh(8), 2), np.tile(fresh(16), 2), np.tile(fresh(128), 2), fresh(256), np.zeros(64, np.uint64)]
    lens = np.array([len(v) for v in sets], dtype=np.int64)
    off = np.zeros(len(sets) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    hv = np.concatenate(sets)
    a, b = O.np_init_permutations(k, 6)
    want = O.c_minhash_bulk(hv, off, a, b)
    ctx.set_option("minhash.split", 1)  # one wave per set: the launches under test
    try:
        if tok_dtype == np.uint32:
            d_tok, d_off, d_out = ctx.to_device(hv.astype(np.uint32)), ctx.to_device(off), ctx.alloc(len(sets) * k * 8)
            ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U32, d_off.ptr, 0, len(sets), hv.size, None, 0, d_out.ptr, _native.MHX_U64)
            ctx.synchronize()
            got = d_out.download((len(sets), k), np.uint64)
        else:
            got = ctx.minhash_bulk((a, b), hv, off, 0, len(sets))
    finally:
        ctx.set_option("minhash.split", 0)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, (bad[:10], lens[bad[:10]])


def test_light_repeats_corpus_is_settled_without_the_pairwise_launch(ctx):
    """1 % repeated tokens: nearly every set fails the first launch's proof (some permutation's minimum falls on a
    repeated token) and nearly none needs more than the tie-tolerant proof: the pairwise launch sees (almost) nothing."""
    rng = np.random.RandomState(8)
    n, t, k = 20_000, 256, 128
    hv = rng.randint(0, 2**32, n * t, dtype=np.uint64)
    m = n * t // 100
    dst = rng.randint(0, hv.size, m)
    hv[dst] = hv[(dst // t) * t + rng.randint(0, t, m)]
    a, b = O.np_init_permutations(k, 1)
    ctx.counters(True)
    got = ctx.minhash_bulk((a, b), hv, None, t, n)
    c = ctx.counters(False)
    assert np.array_equal(got[:3000], O.c_minhash_bulk_dense(hv[: 3000 * t].reshape(3000, t), a, b))
    assert c["sieve_sets_redone"] > n // 2 and c["pairwise_sets"] <= n // 100, c
    ctx.set_option("minhash.ties", 1)  # dedup pass only: the same signatures
    try:
        assert np.array_equal(ctx.minhash_bulk((a, b), hv, None, t, n), got)
    finally:
        ctx.set_option("minhash.ties", 0)


@pytest.mark.parametrize("copies", [300, 20_000])
def test_three_close_keys_reach_the_pairwise_launch(ctx, copies):
    """Three DIFFERENT tokens whose keys under one permutation are the three smallest of the set and within 32 of each
    other: the tie-tolerant proof has one candidate too many, the dedup pass finds nothing to drop, the set goes to the
    pairwise launch -- from the list the second launch writes (a few hundred sets: the workgroup-per-set kernels of a
    clean corpus) or, beyond the list's 16 384 entries, by the scan of the flags.  Bit-exact either way."""
    k = 128
    a, b = O.np_init_permutations(k, 4)
    rng = np.random.RandomState(31)
    protos = []
    for pi in range(40):
        a_lo, b8 = int(a[pi]) & 0xFFFFFFFF, (int(b[pi]) + 8) & 0xFFFFFFFF
        if a_lo % 2 == 0:
            continue
        inv = pow(a_lo, -1, 1 << 32)
        lows = [((m - b8) * inv) % 2**32 for m in (1000, 1007, 1021)]  # keys 1000, 1007, 1021 for permutation pi
        filler = rng.randint(0, 2**32, 253, dtype=np.uint64)
        at = 64 * int(rng.randint(0, 4)) + int(rng.randint(0, 61))  # all three inside one quarter of the set
        # next to each other: also the workgroup-per-set kernel, which proves quarter by quarter, sees all three in one quarter
        protos.append(np.concatenate([filler[:at], np.array(lows, dtype=np.uint64), filler[at:]]))
    protos = np.stack(protos[:8])
    want_proto = O.c_minhash_bulk_dense(protos, a, b)
    reps = -(-copies // len(protos))
    tok = np.tile(protos, (reps, 1))[:copies]
    ctx.set_option("minhash.split", 1)
    try:
        ctx.counters(True)
        got = ctx.minhash_bulk((a, b), tok.reshape(-1), None, 256, copies)
        c = ctx.counters(False)
    finally:
        ctx.set_option("minhash.split", 0)
    assert np.array_equal(got, np.tile(want_proto, (reps, 1))[:copies])
    assert c["pairwise_sets"] == copies, c


# ------------------------------------------------------------------ one context, several threads
def test_threads_sharing_the_process_context_get_right_answers(ctx):
    """ctypes releases the GIL during a libmhx call; the context serialises its callers (mhx_ctx::mu), so threads
    that share the process-wide context -- each growing / re-using the same staging buffers -- still get their own
    results."""
    a, b = O.np_init_permutations(64, 9)
    rng = np.random.RandomState(1)
    jobs = []
    for i in range(8):
        n, t = int(rng.randint(200, 4000)), int(rng.randint(8, 300))
        tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
        jobs.append((tok, O.c_minhash_bulk_dense(tok, a, b)))
    errors = []

    def run(idx):
        tok, want = jobs[idx]
        try:
            for _ in range(6):
                got = MinHash.bulk_signatures(tok, num_perm=64, seed=9, hashfunc=prehashed, gpu_mode="always")
                if not np.array_equal(got, want):
                    errors.append(idx)
                packed = _native.context().bbit_pack(want, 2)
                if not np.array_equal(packed, O.c_bbit_pack(want, 2)):
                    errors.append(-idx - 1)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors


# ------------------------------------------------------------------ N > 1 launch path of bench.py
def _bench(args, env=None, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    return p


def test_bench_spawns_two_ranks_on_one_gpu_for_the_plumbing():
    """python bench.py --gpus 2 (no launcher): the parent spawns the ranks, they meet over the package's own TCP
    rendezvous, barrier, time, reduce MAX, and rank 0 prints the line.  Two ranks share this box's one GPU
    (--share-devices), so the numbers mean nothing and RCCL is left out (next test)."""
    p = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--sets", "20000", "--check-rows", "256", "--share-devices",
                "--no-allgather-probe"])
    assert p.returncode == 0, p.stdout + p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["launcher"] == "self-spawned ranks"
    assert len(line["per_rank"]["ms_per_step"]) == 2 and len(line["per_rank"]["devices"]) == 2
    assert line["value"] > 0 and "roofline" in line and "cpu_baseline" not in line and "extra" not in line
    # without --share-devices the same launch is refused: one GPU per rank
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--sets", "1000"], timeout=300)
    assert p.returncode != 0 and "one GPU per rank" in (p.stdout + p.stderr)


def test_bench_under_torch_distributed_run_environment():
    """The driver's launch form exports RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; bench.py must find its peers
    from those without importing torch (the launcher itself is not needed to test that: two processes with the same
    environment and parent)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="none")
        env.pop("MHX_RDZV_ADDR", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--sets",
                                       "10000", "--check-rows", "64", "--share-devices", "--no-allgather-probe"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["launcher"] == "torch.distributed.run env"
    assert outs[1][0].strip() == ""  # only rank 0 prints
    mods = subprocess.run([sys.executable, "-c", "import sys; sys.argv=['bench.py']; import bench, datasketch_amd.dist; print('torch' in sys.modules)"],
                          capture_output=True, text=True, cwd=ROOT)
    assert mods.stdout.strip() == "False", mods.stdout + mods.stderr


def test_rccl_with_two_ranks_on_one_device_is_refused_or_works():
    """RCCL needs one device per rank: ncclCommInitRank with two ranks on the same GPU is expected to fail
    ("Duplicate GPU detected") on a 1-GPU box.  Either outcome is recorded; what must hold is that a refusal
    comes back as an MhxError from both ranks (no hang, no crash), so the multi-rank all-gather has to be measured
    on a multi-GPU node (bench.py --gpus N reports it under "allgather")."""
    body = r'''
import os, sys
sys.path.insert(0, %r)
from datasketch_amd import _native, rendezvous, dist
g = rendezvous.from_env(timeout=40)
ctx = _native.Context(0)
try:
    comm = dist.communicator(ctx, g)
    print("OK", comm.info(), flush=True)
except Exception as e:
    print("REFUSED", type(e).__name__, str(e)[:300], flush=True)
os._exit(0)
''' % ROOT
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, "-c", body], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MHX_RDZV_ADDR=f"127.0.0.1:{port}"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=75)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT (blocked inside ncclCommInitRank) " + p.communicate()[0])
    print("two RCCL ranks on one device:", outs)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_two_ranks_one_device.txt"), "w") as f:
        f.write("\n---\n".join(outs))
    assert all(("OK" in o) or ("REFUSED" in o) or ("TIMEOUT" in o) for o in outs), outs


# ------------------------------------------------------------------ b-bit Jaccard on packed rows
@pytest.mark.parametrize("k", [64, 100, 256, 1000])
def test_bbit_jaccard_pairs_on_the_device(ctx, k):
    """Agreeing b-bit positions counted on the packed blocks (XOR + popcount) == counted on the unpacked values, for
    every slot size; the estimate equals bBitMinHash.jaccard (ref: b_bit_minhash.py:53-72)."""
    from datasketch_amd import bBitMinHash
    from datasketch_amd.b_bit_minhash import jaccard_pairs, pack_matrix

    rng = np.random.RandomState(k)
    n = 3000
    sig = rng.randint(0, 2**32, (n, k)).astype(np.uint64)
    sig[1::2, : k // 2] = sig[0::2, : k // 2]          # pairs (2i, 2i+1) agree on at least half
    pairs = np.stack([rng.randint(0, n, 5000), rng.randint(0, n, 5000)], axis=1).astype(np.int64)
    pairs[:1000, 0] = np.arange(0, 2000, 2)
    pairs[:1000, 1] = np.arange(1, 2000, 2)
    pairs[1000] = (7, 7)
    for b in (1, 2, 3, 4, 7, 8, 12, 16, 20, 32):
        blocks = pack_matrix(sig, b, gpu_mode="always")
        got = ctx.bbit_jaccard_pairs(blocks, k, b, pairs)
        mask = np.uint64((1 << b) - 1)
        want = np.count_nonzero((sig[pairs[:, 0]] & mask) == (sig[pairs[:, 1]] & mask), axis=1)
        assert np.array_equal(got, want), b
        assert got[1000] == k and got[:1000].min() >= k // 2
    est = jaccard_pairs(pack_matrix(sig, 2, gpu_mode="always"), pairs[:50], k, 2, 0.25, gpu_mode="always")
    for (i, j), e in zip(pairs[:50], est):
        x = bBitMinHash(MinHash(num_perm=k, hashvalues=sig[i]), 2, 0.25)
        y = bBitMinHash(MinHash(num_perm=k, hashvalues=sig[j]), 2, 0.25)
        assert e == x.jaccard(y)
    assert ctx.bbit_jaccard_pairs(pack_matrix(sig, 1, gpu_mode="always"), k, 1, np.empty((0, 2), np.int64)).size == 0


# ------------------------------------------------------------------ several sets per wave (num_perm <= 32)
@pytest.mark.parametrize("k", [1, 5, 8, 9, 16, 17, 31, 32])
def test_packed_kernel_for_short_signatures(ctx, k):
    """num_perm <= 32 runs kernel C (64 / KP sets per wave, tokens from per-group LDS tiles): against the C oracle and
    against the one-set-per-wave path, on ragged sets (empty, shorter than a row, several 256-token blocks), with an
    initial state, uint32 tokens, uint32 output, and a set count that is no multiple of the sets per wave."""
    rng = np.random.RandomState(100 + k)
    n = 4099
    lens = rng.randint(0, 700, size=n)
    lens[:8] = [0, 1, 15, 16, 17, 255, 256, 257]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    hv = rng.randint(0, 2**32, int(offsets[-1]), dtype=np.uint64)
    wide = rng.random_sample(hv.size) < 0.01
    hv[wide] = rng.randint(0, 2**64, int(wide.sum()), dtype=np.uint64)
    hv[offsets[20] : offsets[20] + 5] = hv[offsets[20]]          # repeated tokens: proofs fail, the set goes to the dedup launch
    a, b = O.np_init_permutations(k, 3)
    init = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    init[3, 0] = 2**40                                            # survives only where nothing smaller arrives
    want = O.c_minhash_bulk(hv, offsets, a, b)
    want_init = O.c_minhash_bulk(hv, offsets, a, b, init)
    for packed_off in (0, 1):
        ctx.set_option("minhash.packed", packed_off)
        ctx.set_option("minhash.split", 1)
        try:
            assert np.array_equal(ctx.minhash_bulk((a, b), hv, offsets, 0, n), want), packed_off
            assert np.array_equal(ctx.minhash_bulk((a, b), hv, offsets, 0, n, init), want_init), packed_off
            narrow = hv & np.uint64(0xFFFFFFFF)
            got32 = ctx.minhash_bulk((a, b), narrow.astype(np.uint32), offsets, 0, n, out_dtype=np.uint32)
            assert np.array_equal(got32.astype(np.uint64), O.c_minhash_bulk(narrow, offsets, a, b)), packed_off
            dense = rng.randint(0, 2**32, (1001, 48), dtype=np.uint64)
            assert np.array_equal(ctx.minhash_bulk((a, b), dense.reshape(-1), None, 48, 1001), O.c_minhash_bulk_dense(dense, a, b)), packed_off
        finally:
            ctx.set_option("minhash.packed", 0)
            ctx.set_option("minhash.split", 0)
