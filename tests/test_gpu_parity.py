"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden vectors.

Run on an MI355X:  python -m pytest tests -m gpu -x -q
Every test calls libmhx through ctypes (datasketch_amd._native -> include/mhx.h).  The bar is
bit-exact equality for the integer paths and exact (k, t) pairs for the weighted path in parity
mode.  /root/reference is NOT needed: fixtures in tests/golden/ came from it.
"""
import ctypes
import pickle

import numpy as np
import pytest
import scipy.sparse as sp

from datasketch_amd import LeanMinHash, MinHash, WeightedMinHashGenerator, _native, bBitMinHash, prehashed
from datasketch_amd.b_bit_minhash import pack_matrix
from datasketch_amd.lean_minhash import serialize_matrix
from oracle import oracle as O
from tests.conftest import identity as fake_hash_func

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert _native.gpu_available(), "these tests need an MI355X"
    c = _native.context()
    yield c
    for key in ("minhash.path", "minhash.split", "blocks_per_cu"):
        c.set_option(key, 0)


def _ragged(rng, n_sets, max_len, wide_fraction):
    lens = rng.randint(0, max_len + 1, size=n_sets)
    offsets = np.zeros(n_sets + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    hv = rng.randint(0, 2**32, size=int(offsets[-1]), dtype=np.uint64)
    wide = rng.random_sample(hv.size) < wide_fraction
    hv[wide] = rng.randint(0, 2**64, size=int(wide.sum()), dtype=np.uint64)
    return hv, offsets


# ------------------------------------------------------------------ golden vectors of the reference
def test_library_is_the_hip_one(ctx):
    info = ctx.info()
    assert "gfx950" in info["name"], info
    assert info["compute_units"] >= 200


def test_known_answer_vector(ctx):
    """test/test_minhash.py:109-115 through update_batch on the device."""
    m = MinHash(4, 1, gpu_mode="always")
    m.update_batch([b"Hello"])
    assert m.hashvalues.tolist() == [734825475, 960773806, 359816889, 342714745]


def test_small_golden_vectors(ctx, golden):
    _, meta = golden
    m = MinHash(4, 1, hashfunc=fake_hash_func, gpu_mode="always")
    m.update_batch([12, 24])
    assert m.hashvalues.tolist() == meta["identity_12_24_k4_seed1"]
    m = MinHash(4, 7, hashfunc=fake_hash_func, gpu_mode="always")
    m.update_batch([0, 1, 2**32 - 1, 2**61 - 1, 2**64 - 1])
    assert m.hashvalues.tolist() == meta["identity_edge_k4_seed7"]
    m = MinHash(4, 1, gpu_mode="always")
    m.update_batch([f"token-{i}".encode() for i in range(1000)])
    assert m.hashvalues.tolist() == meta["sha1_token1000_k4_seed1"]


def test_config1_matrix(ctx, golden):
    """BASELINE.json configs[0]: 1k x 64, num_perm=16."""
    arrays, _ = golden
    tok = np.random.RandomState(42).randint(0, 2**32, (1000, 64), dtype=np.uint64)
    got = MinHash.bulk_signatures(tok, num_perm=16, seed=1, hashfunc=prehashed, gpu_mode="always")
    assert np.array_equal(got, arrays["c1_matrix"])
    objs = MinHash.bulk(tok, num_perm=16, seed=1, hashfunc=fake_hash_func, gpu_mode="always")
    assert np.array_equal(np.stack([m.hashvalues for m in objs]), arrays["c1_matrix"])


def test_config2_sample(ctx, golden):
    arrays, _ = golden
    tok = np.random.RandomState(42).randint(0, 2**32, (64, 256), dtype=np.uint64)
    got = MinHash.bulk_signatures(tok, num_perm=128, seed=1, hashfunc=prehashed, gpu_mode="always")
    assert np.array_equal(got, arrays["c2_sample_matrix"])


@pytest.mark.parametrize("path", [0, 1, 2])
@pytest.mark.parametrize("split", [0, 1, 2])
def test_ragged_golden(ctx, golden, path, split):
    arrays, meta = golden
    ctx.set_option("minhash.path", path)
    ctx.set_option("minhash.split", split)
    try:
        for idx, cfg in enumerate(meta["ragged"]):
            hv, off, want = (arrays[f"ragged{idx}_{n}"] for n in ("hv", "offsets", "sig"))
            got = MinHash.bulk_signatures((hv, off), num_perm=cfg["k"], seed=cfg["seed"], hashfunc=prehashed, gpu_mode="always")
            assert np.array_equal(got, want), (cfg, path, split)
    finally:
        ctx.set_option("minhash.path", 0)
        ctx.set_option("minhash.split", 0)


def test_two_batches_with_state(ctx, golden):
    """test/test_minhash_gpu.py:26-52: CPU == GPU bit-exact, also on a non-trivial state."""
    arrays, _ = golden
    d1 = [f"token-{i}".encode() for i in range(500)]
    d2 = [f"token-{i}".encode() for i in range(700)]
    m_cpu = MinHash(num_perm=128, seed=7, gpu_mode="disable")
    m_gpu = MinHash(num_perm=128, seed=7, gpu_mode="always")
    m_auto = MinHash(num_perm=128, seed=7, gpu_mode="detect")
    for m in (m_cpu, m_gpu, m_auto):
        m.update_batch(d1)
    assert np.array_equal(m_gpu.hashvalues, arrays["two_batches_after1"])
    for m in (m_cpu, m_gpu, m_auto):
        m.update_batch(d2)
    assert np.array_equal(m_gpu.hashvalues, arrays["two_batches_after2"])
    assert np.array_equal(m_cpu.hashvalues, m_gpu.hashvalues) and np.array_equal(m_cpu.hashvalues, m_auto.hashvalues)
    data = [f"token-{i}".encode() for i in range(1000)]
    m_gpu = MinHash(num_perm=256, seed=7, gpu_mode="always")
    m_gpu.update_batch(data)
    assert np.array_equal(m_gpu.hashvalues, arrays["sha1_token1000_k256_seed7"])


def test_pickle_roundtrip_is_portable(ctx):
    """test/test_minhash_gpu.py:54-71."""
    m = MinHash(num_perm=128, seed=7, gpu_mode="detect")
    m2 = pickle.loads(pickle.dumps(m))
    m2.update_batch([f"token-{i}".encode() for i in range(64)])
    ref = MinHash(num_perm=128, seed=7)
    ref.update_batch([f"token-{i}".encode() for i in range(64)])
    assert m2 == ref


def test_adversarial_fold_boundaries_golden(ctx, golden):
    arrays, _ = golden
    adv = arrays["adv_tokens"]
    off = np.arange(len(adv) + 1, dtype=np.int64)
    for path in (0, 1, 2):
        ctx.set_option("minhash.path", path)
        got = MinHash.bulk_signatures((adv, off), num_perm=8, seed=1, hashfunc=prehashed, gpu_mode="always")
        assert np.array_equal(got, arrays["adv_per_token_sig"]), path
    ctx.set_option("minhash.path", 0)


# ------------------------------------------------------------------ randomized parity vs the oracle
@pytest.mark.parametrize("k", [1, 7, 16, 63, 64, 65, 100, 128, 129, 200, 256, 257, 320, 512, 600, 1024, 1100])
def test_num_perm_sweep(ctx, k):
    rng = np.random.RandomState(k)
    hv, off = _ragged(rng, 300, 90, 0.1)
    a, b = O.np_init_permutations(k, 11)
    want = O.c_minhash_bulk(hv, off, a, b)
    got = ctx.minhash_bulk((a, b), hv, off, 0, 300)
    assert np.array_equal(got, want)


def test_dense_config2_shape_20k(ctx):
    tok = np.random.RandomState(1).randint(0, 2**32, (20000, 256), dtype=np.uint64)
    a, b = O.np_init_permutations(128, 1)
    want = O.c_minhash_bulk_dense(tok, a, b)
    got = ctx.minhash_bulk((a, b), tok.reshape(-1), None, 256, 20000)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("wide", [0.0, 0.02, 1.0])
@pytest.mark.parametrize("k", [128, 256])
def test_wide_tokens(ctx, wide, k):
    rng = np.random.RandomState(int(wide * 100) + k)
    hv, off = _ragged(rng, 2000, 300, wide)
    a, b = O.np_init_permutations(k, 5)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, 2000), O.c_minhash_bulk(hv, off, a, b))


def test_arbitrary_uint64_permutations(ctx):
    """User-supplied permutations are not range-checked by the reference: any uint64 must work."""
    rng = np.random.RandomState(3)
    a = rng.randint(0, 2**64, 128, dtype=np.uint64)
    b = rng.randint(0, 2**64, 128, dtype=np.uint64)
    a[:4] = [0, 1, 2**64 - 1, 2**63]
    b[:4] = [2**64 - 1, 0, 2**64 - 1, 2**61 - 1]
    hv, off = _ragged(rng, 500, 64, 0.3)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, 500), O.c_minhash_bulk(hv, off, a, b))


def _solve_tokens(a, b, targets):
    out = []
    for ai, bi in zip(a.tolist(), b.tolist()):
        if ai % 2 == 0:
            continue
        inv = pow(ai, -1, 1 << 64)
        out.extend(((s - bi) * inv) % (1 << 64) for s in targets)
    return np.array(out, dtype=np.uint64)


def test_fold_boundaries_exhaustive(ctx):
    """Tokens whose hv*a+b (mod 2^64) sits on every kind of fold boundary, hidden inside otherwise
    ordinary sets so that the fast fold must notice and recompute exactly."""
    p = (1 << 61) - 1
    m = (1 << 29) - 1
    targets = []
    for top in range(8):
        base = top << 61
        for low in (0, 1, p - 8, p - 7, p - 2, p - 1, p, (m << 32) | 0xFFFFFFF0, (m << 32) | 0xFFFFFFFF,
                    0xFFFFFFFF, 0xFFFFFFF8, 0x1_0000_0000 - 1 - top, ((m - 1) << 32) | 0xFFFFFFFF):
            targets.append((base + (low & p)) % (1 << 64))
            targets.append((base + ((low - top) & p)) % (1 << 64))
            targets.append((base + ((low - top - 1) & p)) % (1 << 64))
    for k in (8, 128):
        a, b = O.np_init_permutations(k, 1)
        adv = _solve_tokens(a[:8], b[:8], targets)
        rng = np.random.RandomState(0)
        sets, off = [], [0]
        for t in adv:
            filler = rng.randint(0, 2**32, rng.randint(0, 40), dtype=np.uint64)
            s = np.concatenate([filler, [t]])
            rng.shuffle(s)
            sets.append(s)
            off.append(off[-1] + s.size)
        hv, off = np.concatenate(sets), np.array(off, dtype=np.int64)
        want = O.c_minhash_bulk(hv, off, a, b)
        for path in (0, 1, 2):
            ctx.set_option("minhash.path", path)
            assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, len(sets)), want), (k, path)
        ctx.set_option("minhash.path", 0)


def test_fuzz_random_configurations(ctx):
    """150 seeded random configurations: set counts, ragged lengths (empty sets, rows of exactly 16,
    blocks of exactly 256), share of wide tokens, repeated tokens, num_perm, initial state, launch shape."""
    rng = np.random.RandomState(20260922)
    try:
        for case in range(150):
            n = int(rng.choice([1, 2, 5, 33, 200]))
            kind = rng.randint(0, 4)
            if kind == 0:
                lens = rng.randint(0, 40, n)
            elif kind == 1:
                lens = rng.choice([0, 15, 16, 17, 31, 32, 255, 256, 257, 272, 511, 512, 600], n)
            elif kind == 2:
                lens = rng.randint(200, 700, n)
            else:
                lens = np.full(n, int(rng.choice([16, 64, 256, 300])))
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            hv = rng.randint(0, 2**32, int(off[-1]), dtype=np.uint64)
            wide = rng.random_sample(hv.size) < rng.choice([0.0, 0.0, 0.01, 0.5])
            hv[wide] = rng.randint(0, 2**64, int(wide.sum()), dtype=np.uint64)
            if hv.size > 4 and rng.random_sample() < 0.4:  # repeated tokens, anywhere
                m = int(hv.size * rng.choice([0.02, 0.3]))
                hv[rng.randint(0, hv.size, m)] = hv[rng.randint(0, hv.size, m)]
            k = int(rng.choice([1, 16, 64, 65, 128, 130, 256, 300]))
            a, b = O.np_init_permutations(k, int(rng.randint(0, 1000)))
            init = [None, rng.randint(0, 2**32, k, dtype=np.uint64), rng.randint(0, 2**33, (n, k), dtype=np.uint64)][rng.randint(0, 3)]
            ctx.set_option("minhash.split", int(rng.choice([0, 0, 1, 2])))
            ctx.set_option("minhash.path", int(rng.choice([0, 0, 0, 1, 2])))
            got = ctx.minhash_bulk((a, b), hv, off, 0, n, init)
            want = O.c_minhash_bulk(hv, off, a, b, init)
            assert np.array_equal(got, want), (case, n, kind, k)
    finally:
        ctx.set_option("minhash.split", 0)
        ctx.set_option("minhash.path", 0)


@pytest.mark.parametrize("chunk_bytes", [4096, 70_000, 1 << 20])
def test_host_call_pipelined_in_pieces(ctx, chunk_bytes):
    """mhx_minhash_bulk cut into pieces (upload / kernels / download overlapped): ragged CSR with empty
    sets, one set larger than a piece, offsets[0] > 0, every form of initial state; fixed-length too."""
    rng = np.random.RandomState(chunk_bytes)
    n, k = 700, 128
    lens = rng.randint(0, 300, n)
    lens[100] = 40_000  # larger than the smallest piece sizes
    lens[200:230] = 0
    off = (17 + np.concatenate([[0], np.cumsum(lens)])).astype(np.int64)  # tokens before offsets[0] are unused
    hv = rng.randint(0, 2**32, int(off[-1]), dtype=np.uint64)
    hv[::97] = rng.randint(2**32, 2**64, len(hv[::97]), dtype=np.uint64)
    a, b = O.np_init_permutations(k, 5)
    inits = [None, rng.randint(0, 2**32, k, dtype=np.uint64), rng.randint(0, 2**33, (n, k), dtype=np.uint64)]
    try:
        for init in inits:
            want = O.c_minhash_bulk(hv, off, a, b, init)
            ctx.set_option("host.chunk_bytes", -1)
            whole = ctx.minhash_bulk((a, b), hv, off, 0, n, init)
            ctx.set_option("host.chunk_bytes", chunk_bytes)
            pieces = ctx.minhash_bulk((a, b), hv, off, 0, n, init)
            assert np.array_equal(whole, want)
            assert np.array_equal(pieces, want)
        tok = rng.randint(0, 2**32, (3000, 64), dtype=np.uint64)
        want = O.c_minhash_bulk_dense(tok, a, b)
        assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, 64, 3000), want)
        assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, 64, 3000, inits[1]),
                              np.minimum(want, inits[1][None, :]))
    finally:
        ctx.set_option("host.chunk_bytes", 0)


# ------------------------------------------------------------------ sieve path (minhash.path = 0)
@pytest.mark.parametrize("t", [31, 32, 33, 63, 64, 65, 255, 256, 257, 287, 288, 511, 512, 513, 1000, 5000])
@pytest.mark.parametrize("k", [64, 128])
def test_sieve_block_boundaries(ctx, t, k):
    """Set lengths around the 32-token group and 256-token block edges, uint64 and uint32 tokens."""
    rng = np.random.RandomState(t + k)
    n = 300
    tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
    tok[::7, t // 2] = rng.randint(2**32, 2**64, len(tok[::7]), dtype=np.uint64)  # some wide tokens
    a, b = O.np_init_permutations(k, 3)
    want = O.c_minhash_bulk_dense(tok, a, b)
    assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n), want)
    tok32 = (tok & 0xFFFFFFFF).astype(np.uint32)
    want32 = O.c_minhash_bulk_dense(tok32.astype(np.uint64), a, b)
    d_tok, d_out = ctx.to_device(tok32), ctx.alloc(n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U32, None, t, n, n * t, None, 0, d_out.ptr, _native.MHX_U64)
    ctx.synchronize()
    assert np.array_equal(d_out.download((n, k), np.uint64), want32)


def test_sieve_duplicates_and_constant_sets(ctx):
    """Equal tokens defeat the sieve's uniqueness proof: those sets must come out right through
    the fallback, and the counters must show that the fallback ran."""
    rng = np.random.RandomState(12)
    n, t, k = 400, 256, 128
    tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
    tok[:100, 128:] = tok[:100, :128]            # every token twice
    tok[100:150] = tok[100:150, :1]              # one token repeated 256 times
    tok[150:200, ::3] = tok[150:200, 1:2]        # a third of the set is one token
    tok[200:250, 17] = tok[200:250, 200]         # a single duplicated pair
    a, b = O.np_init_permutations(k, 9)
    want = O.c_minhash_bulk_dense(tok, a, b)
    ctx.set_option("minhash.split", 1)  # wave per set, so that the counters count sets
    ctx.counters(True)
    got = ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n)
    c = ctx.counters(False)
    ctx.set_option("minhash.split", 0)
    assert np.array_equal(got, want)
    assert c["sieve_blocks"] == n and 150 <= c["sieve_sets_redone"] <= 260, c
    assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n), want)  # split over waves


def test_sieve_counters_on_random_sets(ctx):
    """Distinct random tokens: the proof almost never fails (that is what makes the path fast)."""
    n, t, k = 20000, 256, 128
    tok = np.random.RandomState(5).randint(0, 2**32, (n, t), dtype=np.uint64)
    a, b = O.np_init_permutations(k, 1)
    ctx.counters(True)
    got = ctx.minhash_bulk((a, b), tok.reshape(-1), None, t, n)
    c = ctx.counters(False)
    assert np.array_equal(got, O.c_minhash_bulk_dense(tok, a, b))
    assert c["sieve_blocks"] == n and c["sieve_sets_redone"] <= n // 200 and c["exact_sets_redone"] <= 5, c


def test_sieve_adversarial_near_minima(ctx):
    """Two tokens whose low hash words differ by d = 1..48 at the very bottom of the range (so they
    are the two smallest keys of the set), including low words in [2^32-8, 2^32) where the exact
    value wraps around; wide variants change the top bits.  Every proof outcome is exercised."""
    k = 128
    a, b = O.np_init_permutations(k, 4)
    rng = np.random.RandomState(2)
    sets = []
    for pi in range(0, 24):
        a_lo, b_lo = int(a[pi]) & 0xFFFFFFFF, int(b[pi]) & 0xFFFFFFFF
        if a_lo % 2 == 0:
            continue
        inv = pow(a_lo, -1, 1 << 32)
        for base in (2**32 - 9, 2**32 - 8, 2**32 - 1, 0, 1, 7, 8, 9, 15, 16, 17, 40, 1000):
            for d in (0, 1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48):
                lows = [(base - b_lo) * inv % 2**32, (base + d - b_lo) * inv % 2**32]  # s_lo = base, base + d
                for wide in (False, True):
                    toks = [lo | ((int(rng.randint(0, 2**32)) << 32) if wide else 0) for lo in lows]
                    filler = rng.randint(0, 2**32, 254, dtype=np.uint64)
                    s = np.concatenate([filler, np.array(toks, dtype=np.uint64)])
                    rng.shuffle(s)
                    sets.append(s)
    tok = np.stack(sets)
    want = O.c_minhash_bulk_dense(tok, a, b)
    for path in (0, 2):
        ctx.set_option("minhash.path", path)
        got = ctx.minhash_bulk((a, b), tok.reshape(-1), None, 256, len(sets))
        ctx.set_option("minhash.path", 0)
        assert np.array_equal(got, want), path


def test_initial_state_variants(ctx):
    rng = np.random.RandomState(9)
    k, n = 128, 400
    a, b = O.np_init_permutations(k, 2)
    hv, off = _ragged(rng, n, 50, 0.05)
    proto = rng.randint(0, 2**32, k, dtype=np.uint64)
    proto[:3] = [0, 2**32 - 1, 2**40]  # a state value >= 2^32 must survive an empty set untouched
    full = rng.randint(0, 2**33, (n, k), dtype=np.uint64)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, n, proto), O.c_minhash_bulk(hv, off, a, b, proto))
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, n, full), O.c_minhash_bulk(hv, off, a, b, full))
    ctx.set_option("minhash.split", 2)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, n, full), O.c_minhash_bulk(hv, off, a, b, full))
    ctx.set_option("minhash.split", 0)


@pytest.mark.parametrize("n_tokens", [1, 7, 8, 9, 63, 1000, 50_000, 1_000_003])
def test_update_batch_one_long_set(ctx, n_tokens):
    """The reference's own GPU use case: one big set per call (split over waves + atomic min)."""
    rng = np.random.RandomState(n_tokens % 1000)
    hv = rng.randint(0, 2**32, n_tokens, dtype=np.uint64)
    a, b = O.np_init_permutations(256, 7)
    state = rng.randint(0, 2**32, 256, dtype=np.uint64)
    want = O.c_minhash_bulk(hv, np.array([0, n_tokens]), a, b, state)[0]
    assert np.array_equal(ctx.minhash_update_batch((a, b), hv, state), want)


def test_few_long_ragged_sets(ctx):
    rng = np.random.RandomState(4)
    lens = np.array([0, 70000, 3, 0, 123457, 1, 40000], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    hv = rng.randint(0, 2**64, int(off[-1]), dtype=np.uint64)
    a, b = O.np_init_permutations(128, 1)
    assert np.array_equal(ctx.minhash_bulk((a, b), hv, off, 0, len(lens)), O.c_minhash_bulk(hv, off, a, b))


def test_device_resident_u32_variants(ctx):
    """Device-pointer entry point with uint32 tokens in / uint32 signatures out."""
    rng = np.random.RandomState(6)
    n, t, k = 5000, 96, 128
    tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
    a, b = O.np_init_permutations(k, 1)
    want = O.c_minhash_bulk_dense(tok, a, b)
    d64 = ctx.to_device(tok)
    d32 = ctx.to_device(tok.astype(np.uint32))
    for tok_buf, tok_dt in ((d64, _native.MHX_U64), (d32, _native.MHX_U32)):
        for out_dt, np_dt in ((_native.MHX_U64, np.uint64), (_native.MHX_U32, np.uint32)):
            d_out = ctx.alloc(n * k * np.dtype(np_dt).itemsize)
            ctx.minhash_bulk_dev((a, b), tok_buf.ptr, tok_dt, None, t, n, n * t, None, 0, d_out.ptr, out_dt)
            ctx.synchronize()
            assert np.array_equal(d_out.download((n, k), np_dt).astype(np.uint64), want)
    # CSR on the device with uint32 tokens and an odd row start (4-byte aligned scalar loads)
    hv, off = _ragged(rng, 700, 40, 0.0)
    d_hv, d_off = ctx.to_device(hv.astype(np.uint32)), ctx.to_device(off)
    d_out = ctx.alloc(700 * k * 8)
    ctx.minhash_bulk_dev((a, b), d_hv.ptr, _native.MHX_U32, d_off.ptr, 0, 700, hv.size, None, 0, d_out.ptr, _native.MHX_U64)
    ctx.synchronize()
    assert np.array_equal(d_out.download((700, k), np.uint64), O.c_minhash_bulk(hv, off, a, b))


def test_merge(ctx):
    rng = np.random.RandomState(8)
    for shape in ((1, 1), (3, 7), (1000, 128), (257, 129)):
        x = rng.randint(0, 2**32, shape, dtype=np.uint64)
        y = rng.randint(0, 2**32, shape, dtype=np.uint64)
        assert np.array_equal(ctx.minhash_merge(x, y), O.c_minhash_merge(x, y))


def test_invalid_arguments_raise_value_error(ctx):
    a, b = O.np_init_permutations(8, 1)
    with pytest.raises(ValueError):
        ctx.minhash_bulk((a, b), np.zeros(4, np.uint64), np.array([0, 3, 2]), 0, 2)
    with pytest.raises(ValueError):
        ctx.minhash_bulk((a, b), np.zeros(4, np.uint64), None, 4, 1, init=np.zeros(3, np.uint64))
    with pytest.raises(ValueError):
        ctx.bbit_pack(np.zeros((2, 8), np.uint64), 40)
    with pytest.raises(ValueError):
        ctx.band_keys(np.zeros((2, 8), np.uint64), 3, 4)
    assert ctx.minhash_bulk((a, b), np.zeros(0, np.uint64), None, 0, 0).shape == (0, 8)


# ------------------------------------------------------------------ full-size properties (config 2)
def test_full_size_properties_1m_sets(ctx):
    """BASELINE.json configs[1] at full size (1M x 256, K=128): size-independent properties plus
    an oracle check on rows spread over the whole matrix."""
    n, t, k = 1_000_000, 256, 128
    tok = np.random.RandomState(42).randint(0, 2**32, (n, t), dtype=np.uint64)
    a, b = O.np_init_permutations(k, 1)
    d_tok = ctx.to_device(tok)
    d_sig = ctx.alloc(n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_sig.ptr, _native.MHX_U64)
    ctx.synchronize()
    sig = d_sig.download((n, k), np.uint64)
    rows = np.unique(np.concatenate([np.arange(0, 2048), np.linspace(0, n - 1, 4096).astype(np.int64), np.arange(n - 2048, n)]))
    assert np.array_equal(sig[rows], O.c_minhash_bulk_dense(tok[rows], a, b))
    assert int(sig.max()) < 2**32
    # idempotence: hashing the same sets again on top of their own signatures changes nothing
    d_again = ctx.alloc(n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, d_sig.ptr, k, d_again.ptr, _native.MHX_U64)
    ctx.synchronize()
    assert np.array_equal(d_again.download((n, k), np.uint64), sig)
    # union identity: signature(set) == min(signature(first half), signature(second half))
    # (the dense [n, t] corpus viewed as [2n, t/2]: set 2i / 2i+1 are the halves of row i)
    half = t // 2
    del d_again
    d_halves = ctx.alloc(2 * n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, half, 2 * n, n * t, None, 0, d_halves.ptr, _native.MHX_U64)
    ctx.synchronize()
    halves = d_halves.download((n, 2, k), np.uint64)
    assert np.array_equal(np.minimum(halves[:, 0], halves[:, 1]), sig)
    # checksum of checksums against the oracle on a 50k-row slice
    sl = slice(300_000, 350_000)
    assert int(sig[sl].sum(dtype=np.uint64)) == int(O.c_minhash_bulk_dense(tok[sl], a, b).sum(dtype=np.uint64))


# ------------------------------------------------------------------ weighted MinHash
def test_weighted_golden(ctx, golden):
    arrays, meta = golden
    g = WeightedMinHashGenerator(8, 4, 1, gpu_mode="always")
    res = g.minhash_many(np.array([[1, 0, 3, 0, 0.5, 2, 0, 7], [0] * 8, [2] * 8], dtype=np.float64))
    assert [None if r is None else r.hashvalues.tolist() for r in res] == meta["weighted_small"]
    g = WeightedMinHashGenerator(64, 32, 5, gpu_mode="always")
    out, nonempty = g.minhash_many_arrays(arrays["w_dense_in"])
    assert np.array_equal(out, arrays["w_dense_out"]) and np.array_equal(nonempty, arrays["w_dense_nonempty"])
    X = sp.csr_matrix((arrays["w_csr_data"], arrays["w_csr_indices"], arrays["w_csr_indptr"]), shape=(30, 64))
    out, nonempty = g.minhash_many_arrays(X)
    assert np.array_equal(out, arrays["w_csr_out"]) and np.array_equal(nonempty, arrays["w_csr_nonempty"])
    g2 = WeightedMinHashGenerator(512, 128, 1, gpu_mode="always")
    out, nonempty = g2.minhash_many_arrays(arrays["w2_in"])
    assert nonempty.all() and np.array_equal(out, arrays["w2_out"])


@pytest.mark.parametrize("dim,s,density", [(300, 1, 0.5), (300, 100, 0.02), (4096, 128, 1.0), (1000, 200, 0.3)])
def test_weighted_random_vs_oracle(ctx, dim, s, density):
    rng = np.random.RandomState(dim + s)
    g = WeightedMinHashGenerator(dim, s, seed=3, gpu_mode="always")
    x = rng.uniform(0, 100, (48, dim)).astype(np.float32)
    if density < 1.0:
        x[rng.random_sample(x.shape) >= density] = 0
    x[7] = 0
    out, nonempty = g.minhash_many_arrays(x)
    csr = sp.csr_matrix(x)
    csr.sort_indices()
    want, wn = O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)
    assert np.array_equal(nonempty, wn) and not nonempty[7]
    assert np.array_equal(out, want)


def _weighted_logs_case(ctx, rs, ln_cs, betas, indptr, indices, logs):
    h = ctx.wgen_create(rs, ln_cs, betas)
    try:
        got, ne = ctx.weighted_minhash_many(h, rs.shape[0], indptr, indices, logs, True)
        ctx.set_option("weighted.path", 1)
        forced, ne2 = ctx.weighted_minhash_many(h, rs.shape[0], indptr, indices, logs, True)
    finally:
        ctx.set_option("weighted.path", 0)
        ctx.wgen_destroy(h)
    want, wn = O.c_weighted_minhash_many(indptr, indices, None, rs, ln_cs, betas, logs=logs)
    assert np.array_equal(ne, wn) and np.array_equal(ne2, wn)
    assert np.array_equal(forced, want)  # IEEE division everywhere
    assert np.array_equal(got, want)     # reciprocal-multiply quotient + row blocks + guarded rows


def test_weighted_row_blocks_and_guards(ctx):
    """Blocks of 8 rows with a shared column list take the blocked path; rows with values outside
    the proven range of the reciprocal-multiply quotient (huge / tiny logs), rows with other column
    lists, empty rows and ragged block ends take the other paths.  All must agree with the oracle."""
    rng = np.random.RandomState(21)
    dim, s = 256, 70
    rs, ln_cs, betas = O.np_weighted_params(dim, s, 9)
    rows_idx, rows_log = [], []
    dense = np.arange(dim, dtype=np.int32)
    sparse = np.sort(rng.choice(dim, 37, replace=False)).astype(np.int32)
    def logs_for(n):
        l = np.log(rng.uniform(1e-3, 100, n).astype(np.float32))
        l[rng.randint(0, n, 3)] = 0.0            # x == 1
        l[rng.randint(0, n)] = -np.inf           # stored zero
        return l.astype(np.float32)
    for _ in range(16):                          # two clean dense blocks
        rows_idx.append(dense); rows_log.append(logs_for(dim))
    for r in range(8):                           # dense block with one out-of-range row
        l = logs_for(dim)
        if r == 3:
            l[10] = np.float32(2.0**50); l[77] = np.float32(2.0**-60); l[200] = np.float32(-2.0**45)
        rows_idx.append(dense); rows_log.append(l)
    for r in range(8):                           # mixed column lists, one empty row
        idx = np.sort(rng.choice(dim, rng.randint(1, 60), replace=False)).astype(np.int32) if r != 5 else np.zeros(0, np.int32)
        rows_idx.append(idx); rows_log.append(logs_for(max(len(idx), 4))[: len(idx)])
    for _ in range(8):                           # sparse block with a shared column list
        rows_idx.append(sparse); rows_log.append(logs_for(len(sparse)))
    for _ in range(5):                           # ragged end: 5 rows, same list
        rows_idx.append(sparse); rows_log.append(logs_for(len(sparse)))
    indptr = np.concatenate([[0], np.cumsum([len(i) for i in rows_idx])]).astype(np.int64)
    indices, logs = np.concatenate(rows_idx), np.concatenate(rows_log).astype(np.float32)
    _weighted_logs_case(ctx, rs, ln_cs, betas, indptr, indices, logs)
    # a table with an r outside [2^-40, 2^40]: the generator itself falls back to IEEE division
    rs2 = rs.copy()
    rs2[3, 17] = np.float32(2.0**-50)
    rs2[60, 200] = np.float32(2.0**45)
    _weighted_logs_case(ctx, rs2, ln_cs, betas, indptr, indices, logs)


def test_weighted_quotient_stress(ctx):
    """The float32 quotient through a double reciprocal: many (log, r) pairs whose exact quotient
    sits as close to a rounding boundary as float32 operands allow (r = odd * 2^e, log = r * m
    rounded, for m near half-integers of the grid)."""
    rng = np.random.RandomState(33)
    dim, s, n = 512, 64, 64
    rs, ln_cs, betas = O.np_weighted_params(dim, s, 2)
    # q = L/r close to k + 0.5 ulp boundaries: L = fl(r * (j + 0.5 + tiny) * 2^-12)
    j = rng.randint(1, 2**23, size=(n, dim)).astype(np.float64)
    pick_r = rs[rng.randint(0, s, size=(n, dim)), np.arange(dim)[None, :]].astype(np.float64)
    logs = (pick_r * (j + 0.5) * 2.0**-12 * (1 + rng.choice([-1, 1], size=(n, dim)) * 2.0**-24)).astype(np.float32)
    logs *= rng.choice([-1.0, 1.0], size=logs.shape).astype(np.float32)
    indptr = (np.arange(n + 1) * dim).astype(np.int64)
    indices = np.tile(np.arange(dim, dtype=np.int32), n)
    _weighted_logs_case(ctx, rs, ln_cs, betas, indptr, indices, logs.reshape(-1))


def test_weighted_device_log_mode_is_close(ctx):
    """Fast mode: logf on the device.  (k, t) may differ from numpy's log only where two ln_a are
    within float32 rounding of each other; the mismatch rate must be tiny."""
    rng = np.random.RandomState(1)
    x = rng.uniform(0, 100, (64, 2048)).astype(np.float32)
    exact = WeightedMinHashGenerator(2048, 128, seed=1, gpu_mode="always").minhash_many_arrays(x)[0]
    fast = WeightedMinHashGenerator(2048, 128, seed=1, gpu_mode="always", device_log=True).minhash_many_arrays(x)[0]
    mismatch = np.mean(np.any(exact != fast, axis=2))
    assert mismatch < 1e-3, mismatch


# ------------------------------------------------------------------ packing
@pytest.mark.parametrize("k", [8, 48, 64, 100, 128, 256, 300])
def test_bbit_pack(ctx, k):
    rng = np.random.RandomState(k)
    sig = rng.randint(0, 2**32, (257, k), dtype=np.uint64)
    for b in (0, 1, 2, 3, 4, 5, 8, 9, 13, 16, 27, 32):
        assert np.array_equal(pack_matrix(sig, b, gpu_mode="always"), O.c_bbit_pack(sig, b)), (k, b)


def test_bbit_states_golden(ctx, golden):
    arrays, meta = golden
    sig = arrays["misc_sig_k48"][None, :]
    for b, want in meta["bbit_states_k48"].items():
        blocks = pack_matrix(sig, int(b), gpu_mode="always")
        assert blocks.astype("<u8").tobytes().hex() == want[42:]
        m = MinHash(seed=3, hashvalues=arrays["misc_sig_k48"])
        assert bytes(bBitMinHash(m, int(b)).__getstate__()).hex() == want


def test_band_keys(ctx, golden):
    arrays, meta = golden
    for name, bands, r in (("k8", 2, 4), ("k48", 6, 8)):
        keys = ctx.band_keys(arrays[f"misc_sig_{name}"][None, :], bands, r)
        got = [keys[0, i * r : (i + 1) * r].tobytes().hex() for i in range(bands)]
        assert got == meta[f"lsh_keys_{name}_b{bands}_r{r}"]
    rng = np.random.RandomState(2)
    sig = rng.randint(0, 2**32, (1001, 256), dtype=np.uint64)
    for bands, r in ((32, 8), (20, 5), (1, 256), (3, 3)):
        assert np.array_equal(ctx.band_keys(sig, bands, r), O.c_band_keys(sig, bands, r))


def test_lean_serialize(ctx, golden):
    arrays, meta = golden
    assert serialize_matrix(arrays["misc_sig_k8"][None, :], 1, gpu_mode="always")[0].tobytes().hex() == meta["lean_serialize_le"]
    rng = np.random.RandomState(3)
    for k in (1, 8, 100, 128):
        sig = rng.randint(0, 2**32, (513, k), dtype=np.uint64)
        rows = serialize_matrix(sig, -5, gpu_mode="always")
        assert np.array_equal(rows, O.c_lean_serialize(sig, -5))
        lm = LeanMinHash.deserialize(rows[17].tobytes(), "<")
        assert lm.seed == -5 and np.array_equal(lm.hashvalues, sig[17])


# ------------------------------------------------------------------ full-size properties (configs 4 and 5)
def test_full_size_weighted_100k_vectors(ctx):
    """BASELINE.json configs[3] at full size (100k dense vectors of dim 4096, 128 samples): oracle on
    rows spread over the matrix, and row-order equivariance -- reversing the rows must reverse the
    output, i.e. nothing leaks between the rows of an 8-row block or between blocks."""
    n, dim, s = 100_000, 4096, 128
    rs, ln_cs, betas = O.np_weighted_params(dim, s, 1)
    rng = np.random.RandomState(42)
    logs = np.log(rng.randint(1, 2**20, (n, dim)).astype(np.float32) * np.float32(1e-4))
    logs[12345, 77] = 0.0
    indptr = np.arange(n + 1, dtype=np.int64) * dim
    indices = np.tile(np.arange(dim, dtype=np.int32), n)
    h = ctx.wgen_create(rs, ln_cs, betas)
    try:
        out, ne = ctx.weighted_minhash_many(h, s, indptr, indices, logs.reshape(-1), True)
        rev, _ = ctx.weighted_minhash_many(h, s, indptr, indices, logs[::-1].reshape(-1), True)
    finally:
        ctx.wgen_destroy(h)
    assert ne.all() and np.array_equal(rev[::-1], out)
    rows = np.unique(np.concatenate([np.arange(0, 24), np.linspace(0, n - 1, 40).astype(np.int64), np.arange(n - 24, n)]))
    want, _ = O.c_weighted_minhash_many(np.arange(len(rows) + 1, dtype=np.int64) * dim, np.tile(np.arange(dim, dtype=np.int32), len(rows)),
                                        None, rs, ln_cs, betas, logs=logs[rows].reshape(-1))
    assert np.array_equal(out[rows], want)
    assert out[..., 0].min() >= 0 and out[..., 0].max() < dim


def test_full_size_packing_1m_signatures_k256(ctx):
    """BASELINE.json configs[4] per-GPU shape (1.25M x 256 rounded to 1M): b-bit blocks, band keys,
    band digests and Lean records of a whole matrix -- checksums that do not need the oracle at full
    size, plus the oracle on a slice."""
    n, k = 1_000_000, 256
    sig = np.random.RandomState(9).randint(0, 2**32, (n, k), dtype=np.uint64)
    sl = slice(499_000, 500_000)
    blocks = ctx.bbit_pack(sig, 1)
    assert blocks.shape == (n, 4)
    ones = np.unpackbits(blocks.view(np.uint8)).sum(dtype=np.int64)
    assert int(ones) == int((sig & 1).sum(dtype=np.int64))          # every low bit arrived, nothing else
    assert np.array_equal(blocks[sl], O.c_bbit_pack(sig[sl], 1))
    keys = ctx.band_keys(sig, 32, 8)
    assert np.array_equal(keys.byteswap(), sig)                      # byte-swapping twice is the identity
    from datasketch_amd import lsh_bulk as LB

    dig = ctx.band_digests(sig, 32, 8)
    assert np.array_equal(dig[sl], LB.band_digests(sig[sl], 32, 8, gpu_mode="disable"))
    assert len(np.unique(dig[:, 0])) == n                            # random signatures: no shared bucket
    rec = serialize_matrix(sig, 7, gpu_mode="always")
    assert rec.shape == (n, 12 + 4 * k)
    body = rec[:, 12:].copy().view("<u4")
    assert np.array_equal(body, sig.astype(np.uint32))
    assert np.array_equal(rec[:, :12], np.tile(np.frombuffer(np.array([7], "<i8").tobytes() + np.array([k], "<i4").tobytes(), np.uint8), (n, 1)))


# ------------------------------------------------------------------ RCCL binding (single rank)
def test_rccl_allgather_single_rank(ctx):
    lib = ctx.lib
    uid = (ctypes.c_uint8 * _native.COMM_ID_BYTES)()
    _native.check(lib.mhx_comm_unique_id(uid))
    comm = ctypes.c_void_p()
    _native.check(lib.mhx_comm_create(ctx.handle, uid, 0, 1, ctypes.byref(comm)))
    x = np.random.RandomState(0).randint(0, 2**32, (1000, 128), dtype=np.uint64)
    d_x, d_y = ctx.to_device(x), ctx.alloc(x.nbytes)
    _native.check(lib.mhx_comm_allgather_dev(comm, d_x.ptr, d_y.ptr, x.nbytes))
    ctx.synchronize()
    assert np.array_equal(d_y.download(x.shape, np.uint64), x)
    _native.check(lib.mhx_comm_destroy(comm))


def test_device_allgather_through_dist_single_rank(ctx):
    """datasketch_amd.dist device path at world size 1: kernel -> uint32 shard on the device ->
    RCCL all-gather (libmhx communicator, id handed out over the package's own rendezvous group) ->
    device-resident matrix -> host."""
    from datasketch_amd import rendezvous
    from datasketch_amd.dist import allgather_signatures_dev

    group = rendezvous.Group(0, 1)
    n, t, k = 3000, 100, 128
    tok = np.random.RandomState(3).randint(0, 2**32, (n, t), dtype=np.uint64)
    a, b = O.np_init_permutations(k, 1)
    d_tok, d_out = ctx.to_device(tok), ctx.alloc(n * k * 4)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_out.ptr, _native.MHX_U32)
    full = allgather_signatures_dev(ctx, d_out, n, k, [n], group)
    assert full.rows == n and full.k == k
    assert np.array_equal(full.to_host(), O.c_minhash_bulk_dense(tok, a, b))
    from datasketch_amd.dist import communicator

    info = communicator(ctx, group).info()
    assert info["ranks_seen"] == 1 and info["rank"] == 0 and info["device"] == ctx.device and info["rccl_version"] > 0


# ------------------------------------------------------------------ device SHA-1 (row f2)
def test_sha1_tokens_against_hashlib(ctx):
    """sha1_hash32 / sha1_hash64 of byte tokens: every length 0..260 (padding boundaries 55/56,
    63/64, 119/120), every start alignment (tokens are packed back to back), random bytes."""
    import hashlib
    import struct

    from datasketch_amd import sha1_hash_many

    rng = np.random.RandomState(0)
    tokens = [bytes(rng.randint(0, 256, n, dtype=np.uint8)) for n in range(0, 261)]
    tokens += [bytes(rng.randint(0, 256, rng.randint(0, 40), dtype=np.uint8)) for _ in range(5000)]
    tokens += [b"Hello", b"", b"a" * 1000, bytearray(b"xyz"), memoryview(b"memory")]
    want32 = np.array([struct.unpack("<I", hashlib.sha1(t).digest()[:4])[0] for t in tokens], dtype=np.uint32)
    want64 = np.array([struct.unpack("<Q", hashlib.sha1(t).digest()[:8])[0] for t in tokens], dtype=np.uint64)
    assert np.array_equal(sha1_hash_many(tokens, 32, gpu_mode="always"), want32)
    assert np.array_equal(sha1_hash_many(tokens, 64, gpu_mode="always"), want64)
    with pytest.raises(TypeError):
        sha1_hash_many(["not bytes"], gpu_mode="always")


def test_bulk_on_byte_tokens_matches_host_hashing(ctx):
    """MinHash.bulk / update_batch with the default hashfunc: device SHA-1 + device MinHash equals
    the reference arithmetic with hashlib on the host (gpu_mode='disable')."""
    rng = np.random.RandomState(2)
    sets = [[f"tok-{rng.randint(0, 5000)}".encode() for _ in range(rng.randint(0, 120))] for _ in range(400)]
    dev = MinHash.bulk_signatures(sets, num_perm=128, seed=3, gpu_mode="always")
    host = MinHash.bulk_signatures(sets, num_perm=128, seed=3, gpu_mode="disable")
    assert np.array_equal(dev, host)
    objs = MinHash.bulk(sets[:50], num_perm=64, seed=3, gpu_mode="always")
    for m, s in zip(objs, sets[:50]):
        ref = MinHash(num_perm=64, seed=3, gpu_mode="disable")
        ref.update_batch(s)
        assert m == ref
    from datasketch_amd import sha1_hash64

    m64 = MinHash(num_perm=32, seed=1, hashfunc=sha1_hash64, gpu_mode="always")
    r64 = MinHash(num_perm=32, seed=1, hashfunc=sha1_hash64, gpu_mode="disable")
    for m in (m64, r64):
        m.update_batch(sets[7] + sets[8])
    assert m64 == r64


# ------------------------------------------------------------------ consumer side in bulk (rows f1, f4)
def test_band_digests_candidates_and_jaccard_on_device(ctx):
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(4)
    tok = rng.randint(0, 2**32, (3000, 64), dtype=np.uint64)
    tok[5::11] = tok[3:4]
    sig = MinHash.bulk_signatures(tok, num_perm=128, seed=1, hashfunc=prehashed, gpu_mode="always")
    for b, r in ((32, 4), (16, 8), (5, 25), (1, 128), (128, 1)):
        assert np.array_equal(LB.band_digests(sig, b, r, gpu_mode="always"), LB.band_digests(sig, b, r, gpu_mode="disable")), (b, r)
        dev, host = LB.band_keys(sig, b, r, gpu_mode="always"), LB.band_keys(sig, b, r, gpu_mode="disable")
        assert dev.tobytes() == host.tobytes()
    dev_dig, dev_rows = LB.sorted_bands(sig, 32, 4, gpu_mode="always")
    host_dig, _ = LB.sorted_bands(sig, 32, 4, gpu_mode="disable")
    assert np.array_equal(dev_dig, host_dig)                       # same multiset per band, sorted
    full = LB.band_digests(sig, 32, 4, gpu_mode="disable")
    for j in (0, 17, 31):                                          # rows are a permutation consistent with the keys
        assert np.array_equal(np.sort(dev_rows[j]), np.arange(sig.shape[0]))
        assert np.array_equal(full[dev_rows[j].astype(np.int64), j], dev_dig[j])
    pairs = LB.candidate_pairs(sig, 32, 4, gpu_mode="always")
    assert np.array_equal(pairs, LB.candidate_pairs(sig, 32, 4, gpu_mode="disable")) and len(pairs) > 1000
    extra = rng.randint(0, 3000, (5000, 2))
    allp = np.concatenate([pairs, extra])
    assert np.array_equal(LB.jaccard_pairs(sig, allp, gpu_mode="always"), LB.jaccard_pairs(sig, allp, gpu_mode="disable"))
    for k in (1, 63, 64, 65, 200):
        s2 = rng.randint(0, 3, (100, k)).astype(np.uint64)
        p2 = rng.randint(0, 100, (400, 2))
        assert np.array_equal(LB.jaccard_pairs(s2, p2, gpu_mode="always"), LB.jaccard_pairs(s2, p2, gpu_mode="disable")), k
    with pytest.raises(ValueError):
        ctx.jaccard_pairs(sig, np.array([[0, 3000]]))


@pytest.mark.parametrize("n,b,r,clusters", [(1, 4, 2, 0), (2, 4, 2, 1), (500, 8, 4, 0), (5000, 32, 4, 40), (20000, 16, 8, 300), (3000, 1, 16, 30)])
def test_candidate_pairs_on_device(ctx, n, b, r, clusters):
    """mhx_lsh_candidate_pairs against the numpy bucketing: near-duplicates sharing a few bands, clusters
    of identical rows (quadratic buckets), no pairs at all, the capacity protocol and the raw count."""
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(n + b)
    k = b * r + 3
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    for c in range(clusters):  # copy whole rows (all bands shared) or single bands onto other rows
        src, size = rng.randint(0, n), rng.randint(2, 9)
        members = rng.randint(0, n, size)
        if c % 2 == 0:
            sig[members] = sig[src]
        else:
            j = rng.randint(0, b)
            sig[members, j * r:(j + 1) * r] = sig[src, j * r:(j + 1) * r]
    if n == 2:
        sig[1] = sig[0]
    want = LB.candidate_pairs(sig, b, r, gpu_mode="disable")
    got, raw = ctx.lsh_candidate_pairs(sig, b, r)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert raw >= len(want)
    if clusters == 0:
        assert len(want) == 0 and raw == 0
    else:
        assert len(want) > 0
        tight, raw2 = ctx.lsh_candidate_pairs(sig, b, r, capacity=1)  # too small: the call is repeated with the answer
        assert np.array_equal(tight, want) and raw2 == raw
    # the raw count is the sum over buckets of L(L-1)/2
    dig = LB.band_digests(sig, b, r, gpu_mode="disable")
    expect_raw = 0
    for j in range(b):
        _, counts = np.unique(dig[:, j], return_counts=True)
        expect_raw += int((counts * (counts - 1) // 2).sum())
    assert raw == expect_raw
    assert np.array_equal(LB.candidate_pairs(sig, b, r, gpu_mode="always"), want)


@pytest.mark.parametrize("n,dim,s", [(1, 1, 1), (37, 100, 20), (300, 257, 64), (70, 4096, 128)])
def test_weighted_dense_rows_are_compacted_on_device(ctx, n, dim, s):
    """A dense ndarray goes up as it is (mhx_weighted_minhash_many_dense builds the CSR form on the device): same
    (k, t) as the scipy CSR route and the oracle -- all-zero rows, -0.0, NaN, inf, negative and denormal values."""
    rng = np.random.RandomState(n + dim)
    x = rng.uniform(0, 50, (n, dim)).astype(np.float32)
    x[rng.random_sample(x.shape) < 0.6] = 0
    if n > 3:
        x[3] = 0
        x[5, : dim // 2] = -0.0
        x[7, rng.randint(0, dim)] = np.nan
        x[8, rng.randint(0, dim)] = np.inf
        x[9, rng.randint(0, dim)] = -3.5
        x[10, rng.randint(0, dim)] = 1e-42  # float32 denormal: stored, finite log
    g = WeightedMinHashGenerator(dim, s, seed=11, gpu_mode="always")
    dense_out, dense_ne = g.minhash_many_arrays(x)
    csr = sp.csr_matrix(x)
    csr_out, csr_ne = g.minhash_many_arrays(csr)
    assert np.array_equal(dense_ne, csr_ne)
    assert np.array_equal(dense_out, csr_out)
    clean = np.nan_to_num(np.abs(x), nan=1.0, posinf=1.0)  # the oracle on well-defined input
    c = sp.csr_matrix(clean)
    c.sort_indices()
    want, want_ne = O.c_weighted_minhash_many(c.indptr, c.indices, c.data, g.rs, g.ln_cs, g.betas)
    got, got_ne = g.minhash_many_arrays(clean)
    assert np.array_equal(got, want) and np.array_equal(got_ne, want_ne)
    objs = g.minhash_many(x)
    assert [o is None for o in objs] == [not v for v in dense_ne]


def test_weighted_signatures_through_the_lsh_helpers_on_device(ctx):
    """[N, S, 2] int64 WeightedMinHash matrices are [N, 2S] words to the band kernels (keys of 2r words)."""
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(8)
    g = WeightedMinHashGenerator(64, 32, seed=5, gpu_mode="always")
    x = rng.uniform(0, 9, (800, 64)).astype(np.float32)
    x[rng.randint(0, 800, 200)] = x[rng.randint(0, 800, 200)]
    sig, nonempty = g.minhash_many_arrays(x)
    assert sig.shape == (800, 32, 2) and nonempty.all()
    for b, r in ((8, 4), (32, 1), (3, 7)):
        assert LB.band_keys(sig, b, r, gpu_mode="always").tobytes() == LB.band_keys(sig, b, r, gpu_mode="disable").tobytes()
        assert np.array_equal(LB.band_digests(sig, b, r, gpu_mode="always"), LB.band_digests(sig, b, r, gpu_mode="disable"))
        dev, host = LB.candidate_pairs(sig, b, r, gpu_mode="always"), LB.candidate_pairs(sig, b, r, gpu_mode="disable")
        assert np.array_equal(dev, host) and len(host) > 50


def test_near_duplicates_example_same_answer_on_device(ctx):
    """examples/near_duplicates.py: byte tokens -> signatures -> candidate pairs -> Jaccard, device vs numpy."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "near_duplicates.py")
    spec = importlib.util.spec_from_file_location("near_duplicates", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dev = mod.main(["--docs", "1500", "--gpu-mode", "always"])
    host = mod.main(["--docs", "1500", "--gpu-mode", "disable"])
    for key in ("signatures", "pairs", "kept"):
        assert np.array_equal(dev[key], host[key]), key
    assert dev["recall"] > 0.9


@pytest.mark.parametrize("sort_bits", [0, 6, 12, 20, 33, 64])
def test_sorted_bands_exact_for_any_radix_prefix(ctx, sort_bits):
    """mhx_lsh_sort_bands sorts a (band, digest-prefix) key and repairs runs of equal prefixes afterwards:
    digests and rows must equal the stable host sort whatever the prefix length (short prefixes make
    nearly every run a mixed one)."""
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(sort_bits)
    n, b, r = 2500, 24, 3
    sig = rng.randint(0, 2**32, (n, b * r + 5), dtype=np.uint64)
    sig[rng.randint(0, n, 500)] = sig[rng.randint(0, n, 500)]          # buckets of equal keys
    sig[rng.randint(0, n, 300), :r] = sig[rng.randint(0, n, 300), :r]  # ... and of single bands
    want_dig, want_rows = LB.sorted_bands(sig, b, r, gpu_mode="disable")
    try:
        ctx.set_option("lsh.sort_bits", sort_bits)
        got_dig, got_rows = LB.sorted_bands(sig, b, r, gpu_mode="always")
        pairs = LB.candidate_pairs(sig, b, r, gpu_mode="always")
    finally:
        ctx.set_option("lsh.sort_bits", 0)
    assert np.array_equal(got_dig, want_dig)
    assert np.array_equal(got_rows, want_rows)
    assert np.array_equal(pairs, LB.candidate_pairs(sig, b, r, gpu_mode="disable"))


def test_sorted_bands_with_huge_buckets(ctx):
    """Tens of thousands of identical rows (one bucket per band) next to ordinary rows, default and short keys."""
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(31)
    n, b, r = 60_000, 8, 4
    sig = rng.randint(0, 2**32, (n, b * r), dtype=np.uint64)
    sig[rng.permutation(n)[:25_000]] = sig[0]
    want_dig, want_rows = LB.sorted_bands(sig, b, r, gpu_mode="disable")
    try:
        for bits in (0, 16):
            ctx.set_option("lsh.sort_bits", bits)
            got_dig, got_rows = LB.sorted_bands(sig, b, r, gpu_mode="always")
            assert np.array_equal(got_dig, want_dig) and np.array_equal(got_rows, want_rows), bits
    finally:
        ctx.set_option("lsh.sort_bits", 0)


def test_release_scratch_and_regrow(ctx):
    """mhx_ctx_release_scratch frees the staging buffers; the next host calls re-create them."""
    rng = np.random.RandomState(2)
    tok = rng.randint(0, 2**32, (500, 40), dtype=np.uint64)
    a, b = O.np_init_permutations(64, 9)
    want = O.c_minhash_bulk_dense(tok, a, b)
    assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, 40, 500), want)
    ctx.release_scratch()
    ctx.release_scratch()  # idempotent
    assert np.array_equal(ctx.minhash_bulk((a, b), tok.reshape(-1), None, 40, 500), want)
    sig = want
    from datasketch_amd import lsh_bulk as LB
    ctx.release_scratch()
    assert np.array_equal(LB.candidate_pairs(sig, 16, 4, gpu_mode="always"), LB.candidate_pairs(sig, 16, 4, gpu_mode="disable"))


def test_candidate_pairs_device_entry(ctx):
    """The _dev entry point on the output of mhx_lsh_sort_bands_dev, buffers owned by the caller."""
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(77)
    n, b, r = 4000, 20, 5
    sig = rng.randint(0, 2**32, (n, b * r), dtype=np.uint64)
    sig[rng.randint(0, n, 600)] = sig[rng.randint(0, n, 600)]
    want = LB.candidate_pairs(sig, b, r, gpu_mode="disable")
    d_sig = ctx.to_device(sig)
    d_dig, d_rows = ctx.alloc(8 * n * b), ctx.alloc(4 * n * b)
    cap = len(want) + 10
    d_pairs = ctx.alloc(16 * cap)
    _native.check(ctx.lib.mhx_lsh_sort_bands_dev(ctx.handle, d_sig.ptr, n, b * r, b, r, d_dig.ptr, d_rows.ptr))
    found, raw = ctypes.c_int64(0), ctypes.c_int64(0)
    _native.check(ctx.lib.mhx_lsh_candidate_pairs_dev(ctx.handle, d_dig.ptr, d_rows.ptr, n, b, d_pairs.ptr, cap,
                                                      ctypes.byref(found), ctypes.byref(raw)))
    ctx.synchronize()
    assert found.value == len(want) and raw.value >= found.value
    assert np.array_equal(d_pairs.download((cap, 2), np.int64)[: found.value], want)
    # capacity 0 with a NULL buffer only counts
    _native.check(ctx.lib.mhx_lsh_candidate_pairs_dev(ctx.handle, d_dig.ptr, d_rows.ptr, n, b, None, 0,
                                                      ctypes.byref(found), None))
    assert found.value == len(want)
