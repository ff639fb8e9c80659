"""Host-side mirror of the reference API (datasketch_amd.*) -- CPU tests.

These read like the reference's own tests (test/test_minhash.py, test_lean_minhash.py,
test_weighted_minhash.py, test_minhash_gpu.py) and additionally pin results to the golden
vectors produced by the real reference.  gpu_mode='disable' everywhere: no device needed.
"""
import pickle
import struct

import numpy as np
import pytest
import scipy.sparse as sp

from datasketch_amd import (
    LeanMinHash,
    MinHash,
    WeightedMinHash,
    WeightedMinHashGenerator,
    _native,
    bBitMinHash,
    prehashed,
    sha1_hash32,
    sha1_hash64,
)
from datasketch_amd.b_bit_minhash import pack_matrix
from datasketch_amd.lean_minhash import serialize_matrix
from tests.conftest import identity as fake_hash_func


# ------------------------------------------------------------------------------- MinHash
class TestMinHash:
    def test_init(self):
        m1 = MinHash(4, 1, hashfunc=fake_hash_func)
        m2 = MinHash(4, 1, hashfunc=fake_hash_func)
        assert np.array_equal(m1.hashvalues, m2.hashvalues)
        assert np.array_equal(m1.permutations, m2.permutations)
        assert m1.hashvalues.dtype == np.uint64 and m1.permutations.shape == (2, 4)

    def test_permutations_match_reference(self, golden):
        arrays, _ = golden
        for k, seed in ((4, 1), (128, 1), (256, 7), (200, 12345)):
            assert np.array_equal(MinHash(k, seed).permutations, arrays[f"perm_k{k}_s{seed}"])

    def test_is_empty(self):
        assert MinHash().is_empty()

    def test_update(self):
        m1 = MinHash(4, 1, hashfunc=fake_hash_func)
        m2 = MinHash(4, 1, hashfunc=fake_hash_func)
        m1.update(12)
        assert all(m1.hashvalues[i] < m2.hashvalues[i] for i in range(4))

    def test_update_batch(self, golden):
        _, meta = golden
        m1 = MinHash(4, 1, hashfunc=fake_hash_func)
        m1.update(12)
        m1.update(24)
        m2 = MinHash(4, 1, hashfunc=fake_hash_func)
        m2.update_batch([12, 24])
        assert all(m1.hashvalues == m2.hashvalues)
        assert m2.hashvalues.tolist() == meta["identity_12_24_k4_seed1"]
        m2.update_batch([])  # no-op
        assert m2.hashvalues.tolist() == meta["identity_12_24_k4_seed1"]

    def test_wide_and_edge_tokens(self, golden):
        _, meta = golden
        m = MinHash(4, 7, hashfunc=fake_hash_func)
        m.update_batch([0, 1, 2**32 - 1, 2**61 - 1, 2**64 - 1])
        assert m.hashvalues.tolist() == meta["identity_edge_k4_seed7"]
        with pytest.raises(OverflowError):
            MinHash(4, 1, hashfunc=fake_hash_func).update_batch([2**64])
        with pytest.raises(OverflowError):
            MinHash(4, 1, hashfunc=fake_hash_func).update_batch([-1])

    def test_jaccard_merge_union(self):
        m1 = MinHash(4, 1, hashfunc=fake_hash_func)
        m2 = MinHash(4, 1, hashfunc=fake_hash_func)
        assert m1.jaccard(m2) == 1.0
        m2.update(12)
        assert m1.jaccard(m2) == 0.0
        m1.update(13)
        assert m1.jaccard(m2) < 1.0
        m3 = MinHash(4, 1, hashfunc=fake_hash_func)
        m3.merge(m2)
        assert m3.jaccard(m2) == 1.0
        u = MinHash.union(MinHash(4, 1, gpu_mode="detect", hashfunc=fake_hash_func), m2)
        assert u.jaccard(m2) == 1.0 and u.hashfunc is fake_hash_func and u._gpu_mode == "detect"
        with pytest.raises(ValueError):
            MinHash.union(m1)
        with pytest.raises(ValueError):
            m1.merge(MinHash(4, 2))
        with pytest.raises(ValueError):
            m1.jaccard(MinHash(8, 1))

    def test_validation(self):
        with pytest.raises(ValueError):
            MinHash(hashfunc=42)
        with pytest.raises(ValueError):
            MinHash(num_perm=4, permutations=MinHash(8, 1).permutations)
        with pytest.warns(DeprecationWarning):
            MinHash(4, hashobj=object())
        assert len(MinHash(hashvalues=[1, 2, 3], num_perm=99)) == 3

    def test_pickle_eq_copy(self):
        m = MinHash(4, 1, hashfunc=fake_hash_func, gpu_mode="detect")
        m.update(123)
        m.update(45)
        p = pickle.loads(pickle.dumps(m))
        assert p.seed == m.seed and p == m and p._gpu_mode == "detect"
        assert np.array_equal(p.permutations, m.permutations)
        c = m.copy()
        assert c == m and c._gpu_mode == "detect" and c.hashvalues is not m.hashvalues
        c.update_batch(list(range(1000, 1040)))
        assert c != m
        assert MinHash(4, 1) != MinHash(4, 2) and MinHash(4, 1) != MinHash(8, 1)

    def test_count(self):
        m = MinHash(hashfunc=fake_hash_func)
        for v in (11, 123, 92, 98, 123218, 32):
            m.update(v)
        assert m.count() >= 0

    def test_byte_tokens_known_answer(self):
        """Reference known-answer vector, test/test_minhash.py:109-115."""
        m = MinHash(4, 1)
        m.update(b"Hello")
        assert m.hashvalues.tolist() == [734825475, 960773806, 359816889, 342714745]
        m2 = MinHash(4, 1)
        m2.update_batch([b"Hello"])
        assert m2 == m

    def test_sha1_tokens(self, golden):
        arrays, meta = golden
        m = MinHash(4, 1)
        m.update_batch([f"token-{i}".encode() for i in range(1000)])
        assert m.hashvalues.tolist() == meta["sha1_token1000_k4_seed1"]
        m = MinHash(num_perm=128, seed=7)
        m.update_batch([f"token-{i}".encode() for i in range(500)])
        assert np.array_equal(m.hashvalues, arrays["two_batches_after1"])
        m.update_batch([f"token-{i}".encode() for i in range(700)])
        assert np.array_equal(m.hashvalues, arrays["two_batches_after2"])

    def test_bulk_and_generator(self):
        kwargs = dict(num_perm=4, seed=1, hashfunc=fake_hash_func)
        b = [[n * 4 for n in range(4)]] * 2
        m1 = MinHash(**kwargs)
        m1.update_batch(b[0])
        m2, m3 = MinHash.bulk(b, **kwargs)
        assert np.array_equal(m1.hashvalues, m2.hashvalues) and np.array_equal(m1.hashvalues, m3.hashvalues)
        assert isinstance(m2, MinHash) and m2.permutations is m3.permutations
        m2.update(99)  # results are independent, usable objects
        assert not np.array_equal(m2.hashvalues, m3.hashvalues)
        gen = MinHash.generator(iter(b), **kwargs)
        assert next(gen) == m1

    def test_bulk_config1_matches_reference(self, golden):
        arrays, _ = golden
        tok = np.random.RandomState(42).randint(0, 2**32, (1000, 64), dtype=np.uint64)
        got = np.stack([m.hashvalues for m in MinHash.bulk(tok, num_perm=16, seed=1, hashfunc=fake_hash_func)])
        assert np.array_equal(got, arrays["c1_matrix"])
        assert np.array_equal(MinHash.bulk_signatures(tok, num_perm=16, seed=1, hashfunc=prehashed), arrays["c1_matrix"])

    def test_bulk_ragged_matches_reference(self, golden):
        arrays, meta = golden
        for idx, cfg in enumerate(meta["ragged"]):
            hv, off, want = (arrays[f"ragged{idx}_{n}"] for n in ("hv", "offsets", "sig"))
            sets = [hv[off[i] : off[i + 1]] for i in range(cfg["n_sets"])]
            kw = dict(num_perm=cfg["k"], seed=cfg["seed"])
            assert np.array_equal(MinHash.bulk_signatures(sets, hashfunc=fake_hash_func, **kw), want)
            assert np.array_equal(MinHash.bulk_signatures((hv, off), hashfunc=prehashed, **kw), want)

    def test_bulk_with_initial_state(self):
        base = MinHash(8, 3, hashfunc=fake_hash_func)
        base.update_batch([5, 6, 7])
        sets = [[1, 2], [], [9]]
        got = MinHash.bulk(sets, hashfunc=fake_hash_func, seed=3, hashvalues=base.hashvalues, permutations=base.permutations)
        for s, g in zip(sets, got):
            want = base.copy()
            want.update_batch(s)
            assert np.array_equal(g.hashvalues, want.hashvalues)

    def test_always_mode_raises_when_no_device(self):
        """Reference: test/test_minhash_gpu.py:73-79."""
        if _native.gpu_node_present() and _native.gpu_available():
            pytest.skip("GPU available; cannot force negative path.")
        m = MinHash(num_perm=64, seed=1, gpu_mode="always")
        with pytest.raises(RuntimeError):
            m.update_batch([f"token-{i}".encode() for i in range(32)])
        with pytest.raises(RuntimeError):
            MinHash.bulk([[b"a"]], num_perm=8, gpu_mode="always")

    def test_detect_mode_on_cpu_host(self):
        if _native.gpu_node_present():
            pytest.skip("host has a GPU")
        m = MinHash(num_perm=16, seed=1, gpu_mode="detect")
        m.update_batch([b"a", b"b"])
        ref = MinHash(num_perm=16, seed=1)
        ref.update_batch([b"a", b"b"])
        assert m == ref

    def test_hashfuncs(self):
        assert sha1_hash32(b"Hello") < 2**32 and sha1_hash64(b"Hello") < 2**64
        assert sha1_hash64(b"Hello") & 0xFFFFFFFF == sha1_hash32(b"Hello")


# ------------------------------------------------------------------------------- bBit
class TestbBitMinHash:
    def setup_method(self):
        self.m = MinHash(hashfunc=fake_hash_func)
        self.m.update(11)
        self.m.update(123)
        self.m.update(92)

    def test_init(self):
        for b in (1, 2, 3, 9, 27, 32):
            bm = bBitMinHash(self.m, b)
            assert bm.hashvalues.dtype == np.uint32 and bm.hashvalues.max() < 2**b
        with pytest.raises(ValueError):
            bBitMinHash(self.m, 33)
        with pytest.raises(ValueError):
            bBitMinHash(self.m, 1, r=1.5)

    def test_jaccard_and_eq(self):
        m2 = self.m.copy()
        m2.update(999)
        b1, b2 = bBitMinHash(self.m, 4), bBitMinHash(m2, 4)
        assert b1.jaccard(b1) == pytest.approx(1.0)
        assert b1.jaccard(b2) <= 1.0
        assert b1 == bBitMinHash(self.m, 4) and b1 != b2
        with pytest.raises(ValueError):
            b1.jaccard(bBitMinHash(self.m, 5))

    def test_pickle_roundtrip(self):
        """Reference: test/test_minhash.py:189-201."""
        for num_perm in range(16, 513, 16):
            m = MinHash(num_perm, hashfunc=fake_hash_func)
            m.update(11)
            m.update(123)
            for b in (1, 2, 3, 9, 27, 32):
                bm = bBitMinHash(m, b)
                assert pickle.loads(pickle.dumps(bm)) == bm
                assert bm.bytesize() == len(bm.__getstate__())

    def test_states_match_reference(self, golden):
        arrays, meta = golden
        for name, seed, k in (("k8", 1, 8), ("k48", 3, 48)):
            m = MinHash(seed=seed, hashvalues=arrays[f"misc_sig_{name}"])
            for b, want in meta[f"bbit_states_{name}"].items():
                bm = bBitMinHash(m, int(b))
                assert bytes(bm.__getstate__()).hex() == want
                back = bBitMinHash.__new__(bBitMinHash)
                back.__setstate__(bytes.fromhex(want))
                assert back == bm

    def test_pack_matrix_host(self, golden):
        arrays, meta = golden
        sig = arrays["misc_sig_k48"][None, :]
        for b, want in meta["bbit_states_k48"].items():
            blocks = pack_matrix(sig, int(b), gpu_mode="disable")
            assert blocks.astype("<u8").tobytes().hex() == want[42:]  # after the 21-byte header


# ------------------------------------------------------------------------------- Lean
class TestLeanMinHash:
    def _mh(self):
        m = MinHash(8, 1, hashfunc=fake_hash_func)
        m.update_batch([11, 12, 13, 99999, 2**40 + 5])
        return m

    def test_init_and_frozen(self):
        m = self._mh()
        lm = LeanMinHash(m)
        assert lm.seed == m.seed and np.array_equal(lm.hashvalues, m.hashvalues)
        assert LeanMinHash(seed=m.seed, hashvalues=m.hashvalues) == lm
        with pytest.raises(ValueError):
            LeanMinHash()
        with pytest.raises(TypeError):
            lm.update(1)
        assert lm.jaccard(m) == 1.0 and len(lm) == 8 and lm.copy() == lm

    def test_bytesize(self):
        lm = LeanMinHash(self._mh())
        assert lm.bytesize() == 8 + 4 + 4 * 8

    def test_serialize_all_byteorders(self, golden):
        _, meta = golden
        lm = LeanMinHash(self._mh())
        for bo in ("@", "=", "<", ">", "!"):
            buf = bytearray(lm.bytesize(bo))
            lm.serialize(buf, bo)
            assert bytes(buf) == struct.pack("%sqi8I" % bo, lm.seed, 8, *[int(v) for v in lm.hashvalues])
            assert LeanMinHash.deserialize(buf, bo) == lm
            assert LeanMinHash.deserialize(bytes(buf), bo) == lm
        buf = bytearray(lm.bytesize("<"))
        lm.serialize(buf, "<")
        assert bytes(buf).hex() == meta["lean_serialize_le"]
        buf = bytearray(lm.bytesize(">"))
        lm.serialize(buf, ">")
        assert bytes(buf).hex() == meta["lean_serialize_be"]
        with pytest.raises(ValueError):
            lm.serialize(bytearray(5))

    def test_pickle_hash_union(self, golden):
        _, meta = golden
        lm = LeanMinHash(self._mh())
        assert bytes(lm.__getstate__()).hex() == meta["lean_pickle_state"]
        assert pickle.loads(pickle.dumps(lm)) == lm
        assert hash(lm) == hash(LeanMinHash(self._mh()))
        other = MinHash(8, 1, hashfunc=fake_hash_func)
        other.update(77)
        u = LeanMinHash.union(lm, LeanMinHash(other))
        assert np.array_equal(u.hashvalues, np.minimum(lm.hashvalues, other.hashvalues))

    def test_serialize_matrix_host(self, golden):
        arrays, meta = golden
        rows = serialize_matrix(arrays["misc_sig_k8"][None, :], 1, gpu_mode="disable")
        assert rows[0].tobytes().hex() == meta["lean_serialize_le"]


# ------------------------------------------------------------------------------- Weighted
class TestWeightedMinHash:
    def test_generator_tables(self, golden):
        arrays, _ = golden
        g = WeightedMinHashGenerator(64, 32, 5)
        assert g.rs.dtype == np.float32
        for name, arr in (("w_rs", g.rs), ("w_ln_cs", g.ln_cs), ("w_betas", g.betas)):
            assert np.array_equal(arr, arrays[name])

    def test_minhash_single(self, golden):
        arrays, _ = golden
        g = WeightedMinHashGenerator(64, 32, 5)
        v = arrays["w_dense_in"][0]
        keep = v.copy()
        m = g.minhash(v)
        assert isinstance(m, WeightedMinHash) and m.hashvalues.shape == (32, 2) and m.hashvalues.dtype == int
        assert np.array_equal(v, keep)  # input not mutated
        got = np.stack([g.minhash(arrays["w_dense_in"][i]).hashvalues for i in (0, 1, 2)])
        assert np.array_equal(got, arrays["w_single_out"])
        with pytest.raises(ValueError):
            g.minhash(np.zeros(64))
        with pytest.raises(ValueError):
            g.minhash(np.ones(3))
        with pytest.raises(TypeError):
            g.minhash(5)

    def test_minhash_many(self, golden):
        arrays, meta = golden
        g = WeightedMinHashGenerator(8, 4, 1)
        res = g.minhash_many(np.array([[1, 0, 3, 0, 0.5, 2, 0, 7], [0] * 8, [2] * 8], dtype=np.float64))
        assert [None if r is None else r.hashvalues.tolist() for r in res] == meta["weighted_small"]
        g = WeightedMinHashGenerator(64, 32, 5)
        res = g.minhash_many(arrays["w_dense_in"])
        assert [r is not None for r in res] == arrays["w_dense_nonempty"].tolist()
        for r, want in zip(res, arrays["w_dense_out"]):
            if r is not None:
                assert np.array_equal(r.hashvalues, want) and r.hashvalues.dtype == np.int64
        X = sp.csr_matrix((arrays["w_csr_data"], arrays["w_csr_indices"], arrays["w_csr_indptr"]), shape=(30, 64))
        out, nonempty = g.minhash_many_arrays(X)
        assert np.array_equal(out, arrays["w_csr_out"]) and np.array_equal(nonempty, arrays["w_csr_nonempty"])
        with pytest.raises(TypeError):
            g.minhash_many([[1.0] * 64])
        with pytest.raises(ValueError):
            g.minhash_many(np.ones((2, 3)))
        with pytest.raises(ValueError):
            g.minhash_many(np.ones(64))

    def test_value_type(self):
        g = WeightedMinHashGenerator(8, 4, 1)
        a = g.minhash([1, 2, 3, 4, 5, 6, 7, 8])
        b = g.minhash([1, 2, 3, 4, 5, 6, 7, 9])
        assert a.jaccard(a) == 1.0 and 0.0 <= a.jaccard(b) <= 1.0 and len(a) == 4
        assert a.copy() == a and pickle.loads(pickle.dumps(a)) == a
        with pytest.raises(ValueError):
            a.jaccard(WeightedMinHashGenerator(8, 4, 2).minhash([1] * 8))


# ------------------------------------------------------------------ helpers added around the path
def test_pack_tokens_and_sha1_hash_many_on_the_host():
    """Packing byte tokens for the device and the host fallback of sha1_hash_many."""
    import hashlib
    import struct

    from datasketch_amd import _native, sha1_hash_many

    tokens = [b"", b"a", bytearray(b"bc"), memoryview(b"def"), b"\x00" * 70]
    buf, offs = _native.Context.pack_tokens(tokens)
    assert buf.dtype == np.uint8 and offs.tolist() == [0, 0, 1, 3, 6, 76]
    assert bytes(buf[1:3]) == b"bc" and bytes(buf[3:6]) == b"def"
    with pytest.raises(TypeError):
        _native.Context.pack_tokens(["text"])
    want32 = [struct.unpack("<I", hashlib.sha1(bytes(t)).digest()[:4])[0] for t in tokens]
    want64 = [struct.unpack("<Q", hashlib.sha1(bytes(t)).digest()[:8])[0] for t in tokens]
    assert sha1_hash_many(tokens, 32, gpu_mode="disable").tolist() == want32
    assert sha1_hash_many(tokens, 64, gpu_mode="disable").tolist() == want64
    with pytest.raises(ValueError):
        sha1_hash_many(tokens, 16, gpu_mode="disable")


@pytest.mark.parametrize("helper", ["c", "python"])
def test_pack_sets_c_helper_and_python_packer_agree(helper, monkeypatch):
    """csrc/pack_module.c and the pure-Python packer: same arrays, same TypeErrors (those hashlib raises)."""
    from datasketch_amd import _native

    if helper == "c" and _native._mhxpack is None:
        pytest.skip("_mhxpack.so not built")
    if helper == "python":
        monkeypatch.setattr(_native, "_mhxpack", None)
    rng = np.random.RandomState(3)
    sets = [[b"w%d" % v for v in rng.randint(0, 1000, rng.randint(0, 9))] for _ in range(200)]
    sets += [[], {b"solo"}, (bytearray(b"bc"), memoryview(b"def"), b""), iter([b"g", b"\x00\xff"])]
    sets = [list(s) if not isinstance(s, (list, tuple, set)) else s for s in sets]
    buf, byte_offs, set_offs = _native.Context.pack_sets(sets)
    flat = [bytes(t) for s in sets for t in s]
    assert buf.dtype == np.uint8 and byte_offs.dtype == np.int64 and set_offs.dtype == np.int64
    assert bytes(buf) == b"".join(flat)
    assert byte_offs.tolist() == [0] + np.cumsum([len(t) for t in flat]).tolist()
    assert set_offs.tolist() == [0] + np.cumsum([len(s) for s in sets]).tolist()
    b2, o2 = _native.Context.pack_tokens(flat)
    assert bytes(b2) == bytes(buf) and o2.tolist() == byte_offs.tolist()
    empty = _native.Context.pack_sets([])
    assert empty[0].size == 0 and empty[1].tolist() == [0] and empty[2].tolist() == [0]
    for bad in ([["text"]], [[b"ok", 7]], [[b"ok"], 5]):
        with pytest.raises(TypeError):
            _native.Context.pack_sets(bad)


def test_bulk_with_repeated_tokens_equals_update_batch():
    """bulk() may drop repeated tokens of a set before hashing them: the signatures must not change."""
    rng = np.random.RandomState(3)
    sets = [[b"w%d" % v for v in rng.randint(0, 30, rng.randint(0, 80))] for _ in range(40)]
    sig = MinHash.bulk_signatures(sets, num_perm=32, seed=2, gpu_mode="disable")
    for row, s in zip(sig, sets):
        m = MinHash(num_perm=32, seed=2, gpu_mode="disable")
        m.update_batch(s)
        assert np.array_equal(row, m.hashvalues)


def test_near_duplicates_example_runs_on_the_numpy_paths():
    """examples/near_duplicates.py end to end with gpu_mode='disable': planted near-duplicates are found."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "near_duplicates.py")
    spec = importlib.util.spec_from_file_location("near_duplicates", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(["--docs", "1200", "--gpu-mode", "disable"])
    assert out["signatures"].shape == (1200, 128) and out["recall"] > 0.9 and len(out["kept"]) >= 100


def test_bbit_jaccard_pairs_matches_the_object_method():
    """b_bit_minhash.jaccard_pairs on packed rows == bBitMinHash.jaccard of the two objects
    (ref: datasketch/b_bit_minhash.py:53-72), for every slot size and for r = 0 and r > 0."""
    from datasketch_amd.b_bit_minhash import jaccard_pairs, pack_matrix

    rng = np.random.RandomState(0)
    sig = rng.randint(0, 2**32, (40, 100)).astype(np.uint64)
    sig[1, :60] = sig[0, :60]
    pairs = np.array([[0, 1], [2, 3], [4, 4], [0, 39]])
    for b in (1, 2, 3, 5, 8, 13, 32):
        for r in (0.0, 0.3):
            blocks = pack_matrix(sig, b, gpu_mode="disable")
            est = jaccard_pairs(blocks, pairs, 100, b, r, gpu_mode="disable")
            for (i, j), e in zip(pairs, est):
                x = bBitMinHash(MinHash(num_perm=100, hashvalues=sig[i]), b, r)
                y = bBitMinHash(MinHash(num_perm=100, hashvalues=sig[j]), b, r)
                assert e == x.jaccard(y)
    with pytest.raises(ValueError):
        jaccard_pairs(pack_matrix(sig, 1, gpu_mode="disable"), pairs, 100, 2, gpu_mode="disable")  # blocks of another b
    with pytest.raises(ValueError):
        jaccard_pairs(pack_matrix(sig, 1, gpu_mode="disable"), [[0, 40]], 100, 1, gpu_mode="disable")


def test_bulk_signatures_out_dtype_and_uint32_tokens_on_the_numpy_path():
    """out_dtype / uint32 token arrays are host-API additions (not in the reference); on gpu_mode='disable' they are
    plain casts of the reference's arithmetic (minhash.py:293-297) and must agree with the uint64 result."""
    rng = np.random.RandomState(4)
    tok = rng.randint(0, 2**32, (50, 20), dtype=np.uint64)
    want = MinHash.bulk_signatures(tok, num_perm=32, seed=2, hashfunc=prehashed, gpu_mode="disable")
    assert want.dtype == np.uint64
    got32 = MinHash.bulk_signatures(tok.astype(np.uint32), num_perm=32, seed=2, hashfunc=prehashed, gpu_mode="disable", out_dtype=np.uint32)
    assert got32.dtype == np.uint32 and np.array_equal(got32.astype(np.uint64), want)
    csr = (tok.reshape(-1), np.arange(0, 1001, 20))
    assert np.array_equal(MinHash.bulk_signatures(csr, num_perm=32, seed=2, hashfunc=prehashed, gpu_mode="disable", out_dtype=np.uint32), got32)
    sets = [[b"a", b"bb"], [], [b"ccc"]]
    s32 = MinHash.bulk_signatures(sets, num_perm=8, seed=1, gpu_mode="disable", out_dtype=np.uint32)
    assert s32.dtype == np.uint32 and np.array_equal(s32.astype(np.uint64), MinHash.bulk_signatures(sets, num_perm=8, seed=1, gpu_mode="disable"))
    assert MinHash.bulk_signatures([], num_perm=8, gpu_mode="disable", out_dtype=np.uint32).shape == (0, 8)
    with pytest.raises(ValueError):
        MinHash.bulk_signatures(tok, num_perm=8, hashfunc=prehashed, gpu_mode="disable", out_dtype=np.float32)
    big = MinHash(num_perm=8, seed=1, hashfunc=prehashed, hashvalues=np.full(8, 2**40, dtype=np.uint64))
    with pytest.raises(ValueError):
        MinHash.bulk_signatures(tok, num_perm=8, seed=1, hashfunc=prehashed, gpu_mode="disable", out_dtype=np.uint32, hashvalues=big.hashvalues)


def test_serialize_matrix_rejects_values_that_do_not_fit_the_format():
    sig = np.array([[1, 2, 2**32]], dtype=np.uint64)
    with pytest.raises(struct.error):
        serialize_matrix(sig, 1, gpu_mode="disable")
    m = MinHash(num_perm=3, seed=1, hashvalues=sig[0])
    with pytest.raises(struct.error):  # what the object method does for the same value (lean_minhash.py:174-175)
        LeanMinHash(m).serialize(bytearray(LeanMinHash(m).bytesize()))


def test_prehashed_lists_of_ints_take_the_c_packer_and_keep_numpy_semantics():
    """hashfunc=prehashed on lists / tuples of Python ints: packed by csrc/pack_module.c (pack_int_sets); anything else
    (numpy arrays, floats, bools, numpy scalars) goes the numpy way with numpy's conversions, values outside uint64 raise
    OverflowError as np.array(..., dtype=uint64) does (ref: datasketch/minhash.py:294)."""
    from datasketch_amd import _native

    m = MinHash(num_perm=16, seed=1, hashfunc=prehashed, gpu_mode="disable")
    rng = np.random.RandomState(4)
    sets = [list(map(int, rng.randint(0, 2**32, n))) for n in (5, 0, 1, 40)] + [(2**64 - 1, 0, 2**63)]
    hv, off = m._hash_sets(sets)
    assert hv.dtype == np.uint64 and off.tolist() == [0, 5, 5, 6, 46, 49]
    assert np.array_equal(hv, np.concatenate([np.array(s, dtype=np.uint64) for s in sets]))
    if _native._mhxpack is not None:
        assert _native.Context.pack_int_sets(sets) is not None
        for other in ([np.array([1, 2])], [[1.0, 2]], [[True, 2]], [[np.uint64(3)]], [iter([1, 2])], [{1, 2}]):
            assert _native.Context.pack_int_sets(other) is None
    assert m._hash_sets([[1, 2.0], np.array([7, 8]), [True]])[0].tolist() == [1, 2, 7, 8, 1]
    for bad in ([[1, -2]], [[2**64]]):
        with pytest.raises(OverflowError):
            m._hash_sets(bad)
    want = MinHash.bulk_signatures([np.array(s, dtype=np.uint64) for s in sets], num_perm=16, seed=1, hashfunc=prehashed, gpu_mode="disable")
    assert np.array_equal(MinHash.bulk_signatures(sets, num_perm=16, seed=1, hashfunc=prehashed, gpu_mode="disable"), want)
