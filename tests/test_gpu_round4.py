"""GPU parity tests added in round 4 (run on an MI355X: python -m pytest tests -m gpu -x -q).

The weighted kernels with MORE ROWS THAN WORKGROUPS (and than waves, for the one-wave-per-row kernel of round 4) for every
instantiation: a workgroup of the dense walk kernel keeps
its cached walk tables, its prefetched rows and the row's LDS (which doubles as scratch for shared-out lists) across
rows, so a bug in what survives from one row to the next only shows when a workgroup takes several rows
(reference: datasketch/weighted_minhash.py:191-247).  Everything goes through the C ABI; the C
oracle and the evaluate-every-element kernels (weighted.path = 2) are the checkers.
"""
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import scipy.sparse as sp

from datasketch_amd import WeightedMinHashGenerator, _native
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert _native.gpu_available(), "these tests need an MI355X"
    return _native.context()


_THREADS = max(1, min(16, len(os.sched_getaffinity(0))))


def _oracle_rows(g, csr, rows):
    """The C oracle on the given rows, the rows shared out among host threads (ctypes releases the GIL)."""
    sub = csr[rows]
    sub.sort_indices()
    indptr, indices = sub.indptr.astype(np.int64), sub.indices.astype(np.int32)

    def piece(lo, hi):
        with np.errstate(invalid="ignore", divide="ignore"):
            return O.c_weighted_minhash_many(indptr[lo : hi + 1] - indptr[lo], indices[indptr[lo] : indptr[hi]], sub.data[indptr[lo] : indptr[hi]],
                                             g.rs, g.ln_cs, g.betas)

    n = len(rows)
    cuts = np.linspace(0, n, min(_THREADS, n) + 1).astype(np.int64)
    with ThreadPoolExecutor(_THREADS) as pool:
        parts = list(pool.map(lambda i: piece(cuts[i], cuts[i + 1]), range(len(cuts) - 1)))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def _walk_dense_blocks(dim, s, cus=256):
    """The grid launch_weighted_dense_walk chooses (weighted_kernels.hip): workgroups = min(rows, per_cu * CUs)."""
    s_pad = (s + 63) // 64 * 64
    list_cap = max(64, dim // 4)
    n_cc = min(s_pad // 64, 4)
    lds = 4 * ((dim + 3) & ~3) + 2 * ((list_cap + 7) & ~7) + 20 * n_cc * 8 * 64
    return max(1, min(4, (160 << 10) // (lds + 64))) * cus, max(1, min(8, (160 << 10) // (lds + 64))) * cus


def _fuzz_matrix(rng, n, dim, heavy):
    """Rows of mixed density (0.02 .. 1.0 per row: entry-by-entry rows, shared-out lists and walked rows interleaved
    inside every workgroup), uniform or heavy-tailed values, plus a row that stores nothing and rows with NaN / inf."""
    x = (rng.lognormal(0, 2.0, (n, dim)) if heavy else rng.uniform(0, 100, (n, dim))).astype(np.float32)
    dens = rng.choice([0.02, 0.05, 0.09, 0.12, 0.3, 0.6, 1.0], size=(n, 1), p=[0.1, 0.1, 0.1, 0.1, 0.15, 0.15, 0.3])
    if dim >= 16:
        x[rng.random_sample(x.shape) >= dens] = 0
    x[7] = 0
    if dim >= 8:
        x[5, rng.randint(0, dim)] = np.nan
        x[6, rng.randint(0, dim)] = np.inf
        x[n - 3, rng.randint(0, dim)] = np.nan  # ... and in a workgroup's last round
    x[n - 2] = 0
    return x


_DIMS = [63, 301, 513, 4095, 4097, 5000, 10000, 16384]   # dim % 4 != 0 or dim > 4096: weighted_walk_dense_kernel<*, AHEAD = false>
_SAMPLES = [1, 65, 129, 300, 513]


def _case_id(dim, s):
    ahead = dim % 4 == 0 and 4 <= dim <= 4096
    kernel = "walk_wave" if dim % 4 == 0 and 1024 <= dim <= 4096 else "walk_dense"
    return f"{kernel}_AHEAD_{str(ahead).lower()}-dim{dim}-S{s}-chunks{(s + 63) // 64}"


# ... and dims the one-wave-per-row kernel takes (1024 <= dim <= 4096, a multiple of 4): both kernels, every row
_CASES = [(d, s) for d in _DIMS for s in _SAMPLES] + [(1024, 300), (1024, 513), (4096, 300), (4096, 513), (4096, 128), (2048, 128), (1500, 65), (4092, 129),
                                                       (3000, 1),
                                                       # round 5: 1024 <= dim <= 4096 (a multiple of 4) with 65 .. 256 or 321 .. 384 samples is the fetcher / walker kernel's
                                                       (4092, 128), (2052, 100), (3000, 65), (4096, 127), (1024, 128), (1028, 100),
                                                       # ... and with 3, 4 or 6 chunks of 64 samples (three, four, six walkers per row)
                                                       (4096, 192), (4096, 256), (2048, 384), (1024, 200)]


def _wave_kernel_takes(dim):
    return dim % 4 == 0 and 1024 <= dim <= 4096


@pytest.mark.parametrize("dim,s", _CASES, ids=[_case_id(d, s) for d, s in _CASES])
def test_weighted_kernels_with_several_rows_per_workgroup(ctx, dim, s):
    """Every (dim, sample_size) class with n_rows >= 3 x the number of workgroups, dense and CSR, logs (parity mode) and
    values (device log), weighted.split 0 / 1: a sample of rows against the C oracle (all rows of small shapes), EVERY
    row against the evaluate-every-element kernels (weighted.path = 2)."""
    rng = np.random.RandomState(zlib.crc32(f"{dim}/{s}".encode()))
    dense_blocks, csr_blocks = _walk_dense_blocks(dim, s)
    heavy = (dim + s) % 2 == 1
    g = WeightedMinHashGenerator(dim, s, seed=11, gpu_mode="always")
    gv = WeightedMinHashGenerator(dim, s, seed=11, gpu_mode="always", device_log=True)
    wctx, _ = g._device_handle()

    def every_element(gen, x):
        wctx.set_option("weighted.path", 2)
        try:
            return gen.minhash_many_arrays(x)
        finally:
            wctx.set_option("weighted.path", 0)

    def check(x, is_csr):
        n = x.shape[0]
        csr = sp.csr_matrix(x)
        arg = csr if is_csr else x
        out, ne = g.minhash_many_arrays(arg)
        # the oracle on as many rows as ~0.6 s of the host's threads buy (all of them for small shapes), spread over the matrix so
        # that first, middle and last rows of the workgroups' strides are among them
        budget = int(1.5e8 * _THREADS // max(1, dim * s // 2))
        rows = np.arange(n) if budget >= n else np.unique(np.concatenate([np.arange(min(n, 8)), np.linspace(0, n - 1, max(budget, 64)).astype(np.int64), np.arange(n - 8, n)]))
        want, wn = _oracle_rows(g, csr, rows)
        assert np.array_equal(ne[rows].astype(bool), wn)
        got = out[rows]
        odd = np.isin(rows, [5, 6, n - 3]) & (dim >= 8)  # the NaN / inf rows: floor(NaN), floor(inf) cast to int64 platform by platform
        assert np.array_equal(got[wn & ~odd], want[wn & ~odd]) and not got[~wn].any()
        assert np.array_equal(got[wn & odd][:, :, 0], want[wn & odd][:, :, 0])  # the winning columns
        every, ne2 = every_element(g, arg)
        assert np.array_equal(ne2, ne) and np.array_equal(every[ne.astype(bool)], out[ne.astype(bool)])
        # values in, log on the device: the same logf in both implementations, so the sketches must agree exactly
        out_v, ne_v = gv.minhash_many_arrays(arg)
        every_v, ne_v2 = every_element(gv, arg)
        assert np.array_equal(ne_v, ne_v2) and np.array_equal(out_v[ne_v.astype(bool)], every_v[ne_v.astype(bool)])
        return out, ne

    n_dense = 3 * (2048 if _wave_kernel_takes(dim) else dense_blocks) + 37  # (256 workgroups of eight waves: one row per wave and turn)
    x = _fuzz_matrix(rng, n_dense, dim, heavy)
    out0, ne0 = check(x, False)
    if _wave_kernel_takes(dim):  # the workgroup-per-row kernel on the same rows (its AHEAD instantiation)
        wctx.set_option("weighted.kernel", 1)
        try:
            out_w, ne_w = g.minhash_many_arrays(x)
            out_wv, ne_wv = gv.minhash_many_arrays(x)
        finally:
            wctx.set_option("weighted.kernel", 0)
        assert np.array_equal(out_w, out0) and np.array_equal(ne_w, ne0)
        out_v, ne_v = gv.minhash_many_arrays(x)
        assert np.array_equal(out_wv[ne_v.astype(bool)], out_v[ne_v.astype(bool)]) and np.array_equal(ne_wv, ne_v)
    if (s + 63) // 64 * 2 <= 4:  # the waves of a workgroup can share a chunk's list: both settings of weighted.split
        wctx.set_option("weighted.split", 1)
        try:
            out1, ne1 = g.minhash_many_arrays(x)
        finally:
            wctx.set_option("weighted.split", 0)
        assert np.array_equal(out1, out0) and np.array_equal(ne1, ne0)
    # CSR: walked rows (more than 10 % stored) and entry-by-entry rows in one call, three rows per workgroup of the walk kernel
    n_csr = 3 * csr_blocks + 37
    check(x if n_csr == n_dense else _fuzz_matrix(rng, n_csr, dim, heavy), True)


@pytest.mark.parametrize("n", [2_600_000, 5_200_000])
def test_bands_bucketed_beyond_two_and_a_half_million_rows(ctx, n):
    """with 2048 / 4096 bins per band four teams per workgroup need 176 / 272 KB of LDS -- more than a
    workgroup can have -- and the two-pass bucketing used to fail the call instead of sharing fewer bands per workgroup
    (or leaving it to the radix sort).  uint32 signatures, r = 8 (32-byte band pieces: four bands share a 128-byte line),
    against the radix path on the device."""
    rng = np.random.RandomState(n % 1000)
    bands, r = 4, 8
    sig = rng.randint(0, 2**32, size=(n, bands * r), dtype=np.uint64).astype(np.uint32)
    sig[n // 2 : n // 2 + 1000] = sig[:1000]  # some shared buckets
    d_sig = ctx.to_device(sig)
    d_dig, d_rows = ctx.alloc(8 * bands * n), ctx.alloc(4 * bands * n)
    got = {}
    for opt in (0, 1):
        ctx.set_option("lsh.sort", opt)
        try:
            _native.check(ctx.lib.mhx_lsh_sort_bands_dev_typed(ctx.handle, d_sig.ptr, _native.MHX_U32, n, bands * r, bands, r, d_dig.ptr, d_rows.ptr))
            got[opt] = (d_dig.download((bands, n), np.uint64), d_rows.download((bands, n), np.uint32))
        finally:
            ctx.set_option("lsh.sort", 0)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    dig = got[0][0]
    assert (dig[:, 1:] >= dig[:, :-1]).all()
    ctx.release_scratch()


def test_device_log_equals_numpy_log_for_every_float32(ctx):
    """np_logf (weighted_kernels.hip) against np.log of THIS host's numpy for all 2^32 float32 bit patterns, in pieces
    (ref: datasketch/weighted_minhash.py:212 takes np.log of float32 data).  Where the host's numpy does not run the
    AVX2 / AVX512F loop the device function restates (the C model disagrees with np.log too), there is nothing to be equal
    to: the product's start-up check then keeps the log on the host (test_parity_mode_takes_the_log_where_it_is_numpys)."""
    probe = np.random.RandomState(1).randint(0, 0x7F800000, size=1 << 16).astype(np.uint32).view(np.float32)
    if not np.array_equal(O.c_np_logf(probe).view(np.uint32), np.log(probe).view(np.uint32)):
        pytest.skip("this host's numpy does not use the SIMD float32 log the device function restates")
    piece = 1 << 26
    bad = 0
    with np.errstate(all="ignore"):
        for start in range(0, 1 << 32, piece):
            x = np.arange(start, start + piece, dtype=np.uint32).view(np.float32)
            got = ctx.weighted_logf(x).view(np.uint32)
            want = np.log(x).view(np.uint32)
            bad += int(np.count_nonzero(got != want))
    assert bad == 0


@pytest.mark.parametrize("k", list(range(129, 257, 7)) + [192, 193, 255, 256, 257, 300, 320, 384, 385, 448, 512, 513, 576, 600])
def test_num_perm_sweep_129_to_256(ctx, k):
    """Every shape class of 129 <= num_perm <= 256 (ref: datasketch/minhash.py:113-132 allows any num_perm): three
    permutations per lane up to 192 (round 4), four beyond; dense rows of whole 16-token rows, dense rows with a tail, ragged
    sets with empty ones and 64-bit tokens, uint32 tokens, an initial state -- against the C oracle, and the three-per-lane
    launch against the four-per-lane one.  Beyond 256 (appended to the sweep): several passes over the permutations with the number of
    permutations per lane that walks the fewest slots -- 257..384 three per lane twice, 385..512 four twice, 513..576 three three times."""
    from oracle import oracle as O2

    rng = np.random.RandomState(k)
    a, b = O2.np_init_permutations(k, 11)
    for n, t in ((1500, 256), (700, 100)):
        tok = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
        want = O2.c_minhash_bulk_dense(tok, a, b)
        assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n), want)
        assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1).astype(np.uint32), None, t, n, out_dtype=np.uint32), want.astype(np.uint32))
        if k <= 192 or 256 < k <= 384 or 512 < k <= 576:
            ctx.set_option("minhash.p3", 1)
            try:
                assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n), want)
            finally:
                ctx.set_option("minhash.p3", 0)
    lens = rng.randint(0, 90, size=900)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    hv = rng.randint(0, 2**32, size=int(off[-1]), dtype=np.uint64)
    hv[rng.randint(0, hv.size, size=hv.size // 10)] += np.uint64(2**45)
    hv[5:9] = hv[4]  # repeated tokens: the dedup launch
    init = rng.randint(0, 2**32, size=(900, k), dtype=np.uint64)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, 900, init=init), O2.c_minhash_bulk(hv, off, a, b, init=init))


@pytest.mark.parametrize("n,bands", [(1, 1), (5, 3), (3000, 32), (70000, 16), (300000, 4), (20000, 128)])
def test_bands_bucketed_from_the_digest_matrix(ctx, n, bands):
    """mhx_lsh_sort_digests_dev (config 3 computes its band digests once, ref: datasketch/lsh.py:326-347,537-543): the order
    is the sort from the signatures' -- (band, digest, row) -- on both sort paths, with shared buckets."""
    rng = np.random.RandomState(n + bands)
    r = 4
    sig = rng.randint(0, 2**32, size=(n, bands * r), dtype=np.uint64)
    if n > 10:
        sig[n // 2 : n // 2 + n // 10] = sig[: n // 10]
    dig = ctx.band_digests(sig, bands, r)
    want_d, want_r = ctx.lsh_sort_bands(sig, bands, r)
    d_dig, d_sd, d_sr = ctx.to_device(dig), ctx.alloc(8 * bands * n), ctx.alloc(4 * bands * n)
    for opt in (0, 1):
        ctx.set_option("lsh.sort", opt)
        try:
            _native.check(ctx.lib.mhx_lsh_sort_digests_dev(ctx.handle, d_dig.ptr, n, bands, d_sd.ptr, d_sr.ptr))
            got_d, got_r = d_sd.download((bands, n), np.uint64), d_sr.download((bands, n), np.uint32)
        finally:
            ctx.set_option("lsh.sort", 0)
        assert np.array_equal(got_d, want_d) and np.array_equal(got_r, want_r), opt


@pytest.mark.parametrize("kind", ["lognormal2", "pareto", "uniform", "sparse_lognormal"])
def test_a_walks_last_lanes_taken_by_the_whole_wave(ctx, kind):
    """walk_rescue (round 4): when few lanes of a wave still walk, each of them gets all 64 lanes -- 64 list positions per
    turn, a prefix minimum for the stop rule, np.argmin's tie rule for the winner.  Same (k, t) whatever the threshold
    (never, 1 lane, the default, every lane from the third round on), against the C oracle
    (ref: datasketch/weighted_minhash.py:216-229)."""
    rng = np.random.RandomState(zlib.crc32(kind.encode()))
    n, dim, s = 5000, 2048, 128
    if kind == "lognormal2":
        x = rng.lognormal(0, 2.0, (n, dim))
    elif kind == "pareto":
        x = rng.pareto(1.2, (n, dim)) + 1e-3
    elif kind == "uniform":
        x = rng.uniform(0, 100, (n, dim))
    else:
        x = rng.lognormal(0, 2.5, (n, dim))
        x[rng.random_sample(x.shape) < 0.7] = 0
    x = np.ascontiguousarray(x, dtype=np.float32)
    x[:, 7] = x[:, 9]  # equal logs in two columns: ties are decided by the column
    g = WeightedMinHashGenerator(dim, s, seed=5, gpu_mode="always")
    rows = np.arange(0, n, 7)
    want, wn = _oracle_rows(g, sp.csr_matrix(x), rows)
    wctx, _ = g._device_handle()
    outs = []
    for lanes in (-1, 1, 0, 4, 64):
        wctx.set_option("weighted.rescue", lanes)
        try:
            out, ne = g.minhash_many_arrays(x)
        finally:
            wctx.set_option("weighted.rescue", 0)
        assert ne.all() and wn.all() and np.array_equal(out[rows], want), lanes
        outs.append(out)
    assert all(np.array_equal(outs[0], o) for o in outs[1:])


def test_first_launch_follows_the_corpus_and_results_do_not(ctx):
    """Round 4: a context remembers (on the device) whether the last call's sets mostly defeated the one-candidate proof and
    then makes the tie-tolerant proof the first launch -- or not, when that leaves too many sets to the dedup pass.  The
    signatures must not depend on what was learned: clean, lightly and heavily repeating corpora in every order, twice each,
    against the C oracle (ref: datasketch/minhash.py:262-263 accepts any iterable, repeats included)."""
    from oracle import oracle as O2

    rng = np.random.RandomState(99)
    k, n, t = 128, 6000, 256
    a, b = O2.np_init_permutations(k, 4)

    def corpus(rate):
        hv = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
        m = int(rate * hv.size)
        if m:
            rows, dst, src = rng.randint(0, n, m), rng.randint(0, t, m), rng.randint(0, t, m)
            hv[rows, dst] = hv[rows, src]
        return hv

    corpora = {rate: corpus(rate) for rate in (0.0, 0.01, 0.1, 0.5)}
    want = {rate: O2.c_minhash_bulk_dense(hv, a, b) for rate, hv in corpora.items()}
    order = [0.0, 0.01, 0.01, 0.1, 0.1, 0.1, 0.01, 0.01, 0.0, 0.0, 0.5, 0.5, 0.01, 0.0, 0.1, 0.01, 0.01]
    for csr in (False, True):
        for rate in order:
            hv = corpora[rate]
            off = np.arange(0, (n + 1) * t, t, dtype=np.int64) if csr else None
            got = ctx.minhash_bulk((a, b), hv.reshape(-1), off, 0 if csr else t, n)
            assert np.array_equal(got, want[rate]), (csr, rate)


@pytest.mark.parametrize("k", [130, 132, 136, 137, 145, 150, 160, 161, 194, 196, 200, 209, 224, 225])
def test_lane_groups_share_a_partly_filled_last_slot(ctx, k):
    """Round 4: with 3 or 4 permutations per lane and at most 32 of them left for the last slot (K = 129..160, 193..224) lane
    groups hold the same permutations and take every G-th row of a 256-token block each (share_last_slot: spans 4, 8, 16, 32).
    Blocks of 16 rows, partial blocks (1..15 rows), tails, empty sets, both token widths, the tie-tolerant first launch (a
    corpus full of repeats, called until the context has switched), an initial state -- against the C oracle and against the
    same launch with the sharing switched off (ref: datasketch/minhash.py:113-132 allows any num_perm)."""
    from oracle import oracle as O2

    rng = np.random.RandomState(1000 + k)
    a, b = O2.np_init_permutations(k, 5)
    for n, t in ((600, 400), (300, 16), (200, 48), (150, 1024), (400, 250)):  # 25 rows; 1 row; 3 rows; 4 blocks; 15 rows + tail
        tok = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
        want = O2.c_minhash_bulk_dense(tok, a, b)
        assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n), want), (n, t)
        assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1).astype(np.uint32), None, t, n, out_dtype=np.uint32), want.astype(np.uint32)), (n, t)
    lens = rng.randint(0, 700, size=800)
    lens[:40] = np.arange(40)  # every short length
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    hv = rng.randint(0, 2**32, size=int(off[-1]), dtype=np.uint64)
    hv[rng.randint(0, hv.size, size=hv.size // 10)] += np.uint64(2**45)
    init = rng.randint(0, 2**32, size=(800, k), dtype=np.uint64)
    want = O2.c_minhash_bulk(hv, off, a, b, init=init)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, 800, init=init), want)
    want = O2.c_minhash_bulk(hv, off, a, b)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, 800), want)
    ctx.set_option("minhash.share", 1)
    try:
        assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, 800), want)
    finally:
        ctx.set_option("minhash.share", 0)
    # repeats in most sets: after a call or two the first launch is the tie-tolerant one (Three records, merged across groups)
    n, t = 5000, 256
    rep = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
    m = n * t // 50
    rows, dst, src = rng.randint(0, n, m), rng.randint(0, t, m), rng.randint(0, t, m)
    rep[rows, dst] = rep[rows, src]
    want = O2.c_minhash_bulk_dense(rep, a, b)
    for _ in range(4):
        assert np.array_equal(ctx.minhash_bulk((a, b), rep.reshape(-1), None, t, n), want)
    roff = np.arange(0, (n + 1) * t, t, dtype=np.int64)
    for _ in range(2):
        assert np.array_equal(ctx.minhash_bulk((a, b), rep.reshape(-1), roff, 0, n), want)
    clean = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
    want = O2.c_minhash_bulk_dense(clean, a, b)
    for _ in range(2):  # and back
        assert np.array_equal(ctx.minhash_bulk((a, b), clean.reshape(-1), None, t, n), want)
