"""Pin the oracle (numpy + C restatements) to the reference's golden vectors.

The fixtures in tests/golden/ were produced by importing the real reference
(oracle/gen_golden.py).  Everything here runs on CPU.
"""
import hashlib

import numpy as np
import pytest

from oracle import oracle as O


def _sha1_32(data: bytes) -> int:
    # datasketch/hashfunc.py:5-15
    return int.from_bytes(hashlib.sha1(data).digest()[:4], "little")


def test_kat_hello(golden):
    """Reference's own known-answer vector, test/test_minhash.py:109-115."""
    _, meta = golden
    a, b = O.np_init_permutations(4, 1)
    hv = np.array([_sha1_32(b"Hello")], dtype=np.uint64)
    want = [734825475, 960773806, 359816889, 342714745]
    assert meta["kat_hello_k4_seed1"] == want
    init = np.ones(4, dtype=np.uint64) * O.MAX_HASH
    assert O.np_update_batch(init, hv, a, b).tolist() == want
    assert O.c_minhash_bulk(hv, np.array([0, 1]), a, b)[0].tolist() == want


@pytest.mark.parametrize("k,seed", [(4, 1), (128, 1), (256, 7), (200, 12345)])
def test_permutations(golden, k, seed):
    arrays, _ = golden
    assert np.array_equal(O.np_init_permutations(k, seed), arrays[f"perm_k{k}_s{seed}"])


def test_small_vectors(golden):
    _, meta = golden
    a, b = O.np_init_permutations(4, 1)
    hv = np.array([12, 24], dtype=np.uint64)
    assert O.c_minhash_bulk(hv, np.array([0, 2]), a, b)[0].tolist() == meta["identity_12_24_k4_seed1"]
    a, b = O.np_init_permutations(4, 7)
    hv = np.array([0, 1, 2**32 - 1, 2**61 - 1, 2**64 - 1], dtype=np.uint64)
    assert O.c_minhash_bulk(hv, np.array([0, 5]), a, b)[0].tolist() == meta["identity_edge_k4_seed7"]
    init = np.ones(4, dtype=np.uint64) * O.MAX_HASH
    assert O.np_update_batch(init, hv, a, b).tolist() == meta["identity_edge_k4_seed7"]
    a, b = O.np_init_permutations(4, 1)
    hv = np.array([_sha1_32(f"token-{i}".encode()) for i in range(1000)], dtype=np.uint64)
    assert O.c_minhash_bulk(hv, np.array([0, 1000]), a, b)[0].tolist() == meta["sha1_token1000_k4_seed1"]


def test_config1_matrix(golden):
    """BASELINE.json configs[0]: 1k sets x 64 tokens, num_perm=16."""
    arrays, meta = golden
    tok = np.random.RandomState(42).randint(0, 2**32, (1000, 64), dtype=np.uint64)
    a, b = O.np_init_permutations(16, 1)
    got_c = O.c_minhash_bulk_dense(tok, a, b)
    got_np = O.np_minhash_bulk(list(tok), a, b)
    assert np.array_equal(got_c, arrays["c1_matrix"])
    assert np.array_equal(got_np, arrays["c1_matrix"])
    assert hashlib.sha256(got_c.tobytes()).hexdigest() == meta["c1_sha256"]


def test_config2_sample(golden):
    arrays, _ = golden
    tok = np.random.RandomState(42).randint(0, 2**32, (64, 256), dtype=np.uint64)
    a, b = O.np_init_permutations(128, 1)
    assert np.array_equal(O.c_minhash_bulk_dense(tok, a, b), arrays["c2_sample_matrix"])


def test_ragged_corpora(golden):
    arrays, meta = golden
    for idx, cfg in enumerate(meta["ragged"]):
        a, b = O.np_init_permutations(cfg["k"], cfg["seed"])
        hv, off, want = (arrays[f"ragged{idx}_{n}"] for n in ("hv", "offsets", "sig"))
        assert np.array_equal(O.c_minhash_bulk(hv, off, a, b), want), cfg
        sets = [hv[off[i] : off[i + 1]] for i in range(cfg["n_sets"])]
        assert np.array_equal(O.np_minhash_bulk(sets, a, b), want), cfg


def test_two_batches_with_state(golden):
    """test/test_minhash_gpu.py:39-52 shape: a second update_batch on a non-trivial state."""
    arrays, _ = golden
    a, b = O.np_init_permutations(128, 7)
    d1 = np.array([_sha1_32(f"token-{i}".encode()) for i in range(500)], dtype=np.uint64)
    d2 = np.array([_sha1_32(f"token-{i}".encode()) for i in range(700)], dtype=np.uint64)
    s1 = O.c_minhash_bulk(d1, np.array([0, 500]), a, b)
    assert np.array_equal(s1[0], arrays["two_batches_after1"])
    s2 = O.c_minhash_bulk(d2, np.array([0, 700]), a, b, init=s1)
    assert np.array_equal(s2[0], arrays["two_batches_after2"])
    s2b = O.c_minhash_bulk(d2, np.array([0, 700]), a, b, init=s1[0])
    assert np.array_equal(s2b[0], arrays["two_batches_after2"])
    a, b = O.np_init_permutations(256, 7)
    d = np.array([_sha1_32(f"token-{i}".encode()) for i in range(1000)], dtype=np.uint64)
    assert np.array_equal(O.c_minhash_bulk(d, np.array([0, 1000]), a, b)[0], arrays["sha1_token1000_k256_seed7"])


def test_adversarial_fold_boundaries(golden):
    """Tokens chosen so hv*a+b (mod 2^64) lands on p, 2p, 8p+7, 2^64-1, ... for some permutation."""
    arrays, _ = golden
    a, b = O.np_init_permutations(8, 1)
    adv = arrays["adv_tokens"]
    off = np.arange(len(adv) + 1, dtype=np.int64)
    assert np.array_equal(O.c_minhash_bulk(adv, off, a, b), arrays["adv_per_token_sig"])


def test_merge():
    rng = np.random.RandomState(0)
    x = rng.randint(0, 2**32, (50, 16), dtype=np.uint64)
    y = rng.randint(0, 2**32, (50, 16), dtype=np.uint64)
    assert np.array_equal(O.c_minhash_merge(x, y), np.minimum(x, y))


def test_weighted_params(golden):
    arrays, _ = golden
    rs, ln_cs, betas = O.np_weighted_params(64, 32, 5)
    assert np.array_equal(rs, arrays["w_rs"])
    assert np.array_equal(ln_cs, arrays["w_ln_cs"])
    assert np.array_equal(betas, arrays["w_betas"])


def _csr_from_dense(x):
    import scipy.sparse as sp

    m = sp.csr_matrix(x, dtype=np.float32)
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data


def test_weighted_small(golden):
    _, meta = golden
    rs, ln_cs, betas = O.np_weighted_params(8, 4, 1)
    x = np.array([[1, 0, 3, 0, 0.5, 2, 0, 7], [0] * 8, [2] * 8], dtype=np.float32)
    for fn in (O.np_weighted_minhash_many, O.c_weighted_minhash_many):
        out, nonempty = fn(*_csr_from_dense(x), rs, ln_cs, betas)
        got = [out[i].tolist() if nonempty[i] else None for i in range(3)]
        assert got == meta["weighted_small"]


def test_weighted_dense_and_csr(golden):
    arrays, _ = golden
    rs, ln_cs, betas = arrays["w_rs"], arrays["w_ln_cs"], arrays["w_betas"]
    for fn in (O.np_weighted_minhash_many, O.c_weighted_minhash_many):
        out, nonempty = fn(*_csr_from_dense(arrays["w_dense_in"]), rs, ln_cs, betas)
        assert np.array_equal(nonempty, arrays["w_dense_nonempty"])
        assert np.array_equal(out, arrays["w_dense_out"])
        out, nonempty = fn(arrays["w_csr_indptr"], arrays["w_csr_indices"], arrays["w_csr_data"], rs, ln_cs, betas)
        assert np.array_equal(nonempty, arrays["w_csr_nonempty"])
        assert np.array_equal(out, arrays["w_csr_out"])


def test_weighted_config4_shaped(golden):
    arrays, _ = golden
    rs, ln_cs, betas = O.np_weighted_params(512, 128, 1)
    out, nonempty = O.c_weighted_minhash_many(*_csr_from_dense(arrays["w2_in"]), rs, ln_cs, betas)
    assert nonempty.all()
    assert np.array_equal(out, arrays["w2_out"])


def test_bbit_states(golden):
    arrays, meta = golden
    for name, seed in (("k8", 1), ("k48", 3)):
        sig = arrays[f"misc_sig_{name}"]
        for b, want in meta[f"bbit_states_{name}"].items():
            b = int(b)
            assert O.np_bbit_state(sig, seed, b).hex() == want, (name, b)
            blocks_c = O.c_bbit_pack(sig[None, :], b)
            assert np.array_equal(blocks_c, O.np_bbit_pack(sig[None, :], b)), (name, b)


def test_band_keys(golden):
    arrays, meta = golden
    for name, bands, r in (("k8", 2, 4), ("k48", 6, 8)):
        sig = arrays[f"misc_sig_{name}"][None, :]
        for fn in (O.np_band_keys, O.c_band_keys):
            keys = np.ascontiguousarray(fn(sig, bands, r))
            got = [keys[0, i * r : (i + 1) * r].tobytes().hex() for i in range(bands)]
            assert got == meta[f"lsh_keys_{name}_b{bands}_r{r}"]


def test_lean_serialize(golden):
    arrays, meta = golden
    sig = arrays["misc_sig_k8"]
    assert O.np_lean_serialize(sig, 1, "<").hex() == meta["lean_serialize_le"]
    assert O.np_lean_serialize(sig, 1, ">").hex() == meta["lean_serialize_be"]
    assert O.c_lean_serialize(sig[None, :], 1)[0].tobytes().hex() == meta["lean_serialize_le"]
