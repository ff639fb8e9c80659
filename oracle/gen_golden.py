#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (ekzhu/datasketch @ /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python oracle/gen_golden.py            # rewrites tests/golden/golden.npz + golden.json

The fixtures pin oracle/ (tests/test_oracle.py) and, through the same files, the HIP path
(tests/test_gpu_parity.py).  Inputs are seeded; outputs are whatever the reference computes.
Nothing here is copied from the reference: it is imported and called.
"""
from __future__ import annotations

import hashlib
import json
import os
import pickle
import sys

import numpy as np

REF = os.environ.get("DATASKETCH_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import scipy.sparse as sp  # noqa: E402
from datasketch import LeanMinHash, MinHash, MinHashLSH, WeightedMinHashGenerator  # noqa: E402
from datasketch.b_bit_minhash import bBitMinHash  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def identity(x):
    return x


def ragged_corpus(rng, n_sets, max_len, wide_fraction):
    """Seeded ragged corpus: lengths 0..max_len, a fraction of tokens >= 2^32 (sha1_hash64 range)."""
    lens = rng.randint(0, max_len + 1, size=n_sets)
    lens[0] = 0  # an empty set first
    lens[-1] = 0  # and last
    offsets = np.zeros(n_sets + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    hv = rng.randint(0, 2**32, size=int(offsets[-1]), dtype=np.uint64)
    wide = rng.random_sample(hv.size) < wide_fraction
    hv[wide] = rng.randint(0, 2**64, size=int(wide.sum()), dtype=np.uint64)
    return hv, offsets


def main():
    arrays = {}
    meta = {"reference": "ekzhu/datasketch v1.10.0 imported from /root/reference", "numpy": np.__version__}

    # ---- 1. reference's own KAT (test/test_minhash.py:109-115) re-derived from the reference
    m = MinHash(4, 1)
    m.update(b"Hello")
    meta["kat_hello_k4_seed1"] = m.hashvalues.tolist()
    assert meta["kat_hello_k4_seed1"] == [734825475, 960773806, 359816889, 342714745]

    # ---- 2. permutations
    for k, seed in ((4, 1), (128, 1), (256, 7), (200, 12345)):
        arrays[f"perm_k{k}_s{seed}"] = MinHash(k, seed, hashfunc=identity).permutations

    # ---- 3. small known-answer vectors
    m = MinHash(4, 1, hashfunc=identity)
    m.update_batch([12, 24])
    meta["identity_12_24_k4_seed1"] = m.hashvalues.tolist()
    m = MinHash(4, 7, hashfunc=identity)
    edge = [0, 1, 2**32 - 1, 2**61 - 1, 2**64 - 1]
    m.update_batch(edge)
    meta["identity_edge_k4_seed7"] = m.hashvalues.tolist()
    m = MinHash(4, 1)
    m.update_batch([f"token-{i}".encode() for i in range(1000)])
    meta["sha1_token1000_k4_seed1"] = m.hashvalues.tolist()

    # ---- 4. config 1 (BASELINE.json configs[0]): 1k x 64, K=16, identity hashfunc
    tok = np.random.RandomState(42).randint(0, 2**32, (1000, 64), dtype=np.uint64)
    mat = np.stack([x.hashvalues for x in MinHash.bulk(tok, num_perm=16, seed=1, hashfunc=identity)])
    arrays["c1_matrix"] = mat
    meta["c1_sha256"] = hashlib.sha256(np.ascontiguousarray(mat).tobytes()).hexdigest()

    # ---- 5. config-2-shaped sample (256 tokens, K=128) - 64 rows are enough to pin the shape
    tok = np.random.RandomState(42).randint(0, 2**32, (64, 256), dtype=np.uint64)
    arrays["c2_sample_matrix"] = np.stack(
        [x.hashvalues for x in MinHash.bulk(tok, num_perm=128, seed=1, hashfunc=identity)]
    )

    # ---- 6. ragged corpora across K, with 64-bit tokens and empty sets
    rng = np.random.RandomState(2024)
    ragged = []
    for idx, (k, seed, n_sets, max_len, wide) in enumerate(
        [
            (1, 3, 40, 20, 0.0),
            (4, 1, 60, 33, 0.3),
            (16, 1, 50, 70, 0.0),
            (64, 9, 40, 130, 0.1),
            (100, 5, 30, 90, 0.5),
            (128, 1, 40, 300, 0.0),
            (200, 12345, 20, 65, 0.2),
            (256, 7, 24, 257, 1.0),
            (320, 11, 10, 40, 0.05),
        ]
    ):
        hv, offsets = ragged_corpus(rng, n_sets, max_len, wide)
        sets = [hv[offsets[i] : offsets[i + 1]] for i in range(n_sets)]
        sig = np.stack([x.hashvalues for x in MinHash.bulk(sets, num_perm=k, seed=seed, hashfunc=identity)])
        arrays[f"ragged{idx}_hv"] = hv
        arrays[f"ragged{idx}_offsets"] = offsets
        arrays[f"ragged{idx}_sig"] = sig
        ragged.append({"k": k, "seed": seed, "n_sets": n_sets})
    meta["ragged"] = ragged

    # ---- 7. update_batch with a non-trivial state: two successive batches (test_minhash_gpu.py:39-52)
    d1 = [f"token-{i}".encode() for i in range(500)]
    d2 = [f"token-{i}".encode() for i in range(700)]
    m = MinHash(num_perm=128, seed=7)
    m.update_batch(d1)
    arrays["two_batches_after1"] = m.hashvalues.copy()
    m.update_batch(d2)
    arrays["two_batches_after2"] = m.hashvalues.copy()
    m = MinHash(num_perm=256, seed=7)
    m.update_batch([f"token-{i}".encode() for i in range(1000)])
    arrays["sha1_token1000_k256_seed7"] = m.hashvalues.copy()

    # ---- 8. adversarial tokens: solutions of hv*a+b == s (mod 2^64) for s near the fold boundaries
    perms = MinHash(8, 1, hashfunc=identity).permutations
    P = (1 << 61) - 1
    adv = []
    for a, b in zip(perms[0].tolist(), perms[1].tolist()):
        if a % 2 == 0:
            continue
        inv = pow(a, -1, 1 << 64)
        for s in (P, P - 1, P + 1, 2 * P, 2 * P + 1, 8 * P + 6, 8 * P + 7, 2**64 - 1, 0, 2**61, 2**61 - 2,
                  7 * P, 7 * P + 7, 3 * (1 << 61) - 1, (1 << 61) | 0xFFFFFFFF, ((1 << 29) - 1) << 32 | 0xFFFFFFF9):
            adv.append(((s - b) * inv) % (1 << 64))
    adv = np.array(adv, dtype=np.uint64)
    m = MinHash(8, 1, hashfunc=identity)
    per_token = np.stack([MinHash(8, 1, hashfunc=identity).hashvalues for _ in adv])
    for i, t in enumerate(adv):
        mm = MinHash(8, 1, hashfunc=identity)
        mm.update_batch([int(t)])
        per_token[i] = mm.hashvalues
    arrays["adv_tokens"] = adv
    arrays["adv_per_token_sig"] = per_token

    # ---- 9. LeanMinHash wire format + bBit states + LSH band keys for one signature
    m = MinHash(8, 1, hashfunc=identity)
    m.update_batch([11, 12, 13, 99999, 2**40 + 5])
    arrays["misc_sig_k8"] = m.hashvalues.copy()
    lm = LeanMinHash(m)
    for bo, name in (("<", "le"), (">", "be")):
        buf = bytearray(lm.bytesize(bo))
        lm.serialize(buf, bo)
        meta[f"lean_serialize_{name}"] = bytes(buf).hex()
    meta["lean_pickle_state"] = bytes(lm.__getstate__()).hex()
    bbit = {}
    for b in (0, 1, 2, 3, 4, 5, 8, 9, 16, 27, 32):
        bbit[str(b)] = bytes(bBitMinHash(m, b).__getstate__()).hex()
    meta["bbit_states_k8"] = bbit
    m48 = MinHash(48, 3, hashfunc=identity)
    m48.update_batch(list(range(100, 400, 7)))
    arrays["misc_sig_k48"] = m48.hashvalues.copy()
    meta["bbit_states_k48"] = {str(b): bytes(bBitMinHash(m48, b).__getstate__()).hex() for b in (1, 2, 3, 7, 13, 32)}
    lsh = MinHashLSH(num_perm=8, params=(2, 4))
    meta["lsh_keys_k8_b2_r4"] = [lsh._H(m.hashvalues[s:e]).hex() for s, e in lsh.hashranges]
    lsh48 = MinHashLSH(num_perm=48, params=(6, 8))
    meta["lsh_keys_k48_b6_r8"] = [lsh48._H(m48.hashvalues[s:e]).hex() for s, e in lsh48.hashranges]

    # ---- 10. weighted MinHash
    g = WeightedMinHashGenerator(8, 4, 1)
    res = g.minhash_many(np.array([[1, 0, 3, 0, 0.5, 2, 0, 7], [0] * 8, [2] * 8], dtype=np.float64))
    meta["weighted_small"] = [None if r is None else r.hashvalues.tolist() for r in res]
    g = WeightedMinHashGenerator(64, 32, 5)
    arrays["w_rs"], arrays["w_ln_cs"], arrays["w_betas"] = g.rs, g.ln_cs, g.betas
    rng = np.random.RandomState(77)
    dense = rng.uniform(0, 100, (40, 64)).astype(np.float32)
    dense[rng.random_sample(dense.shape) < 0.6] = 0
    dense[3] = 0  # all-zero row -> None
    dense[17] = 0
    arrays["w_dense_in"] = dense
    res = g.minhash_many(dense)
    arrays["w_dense_nonempty"] = np.array([r is not None for r in res])
    arrays["w_dense_out"] = np.stack([np.zeros((32, 2), dtype=np.int64) if r is None else r.hashvalues for r in res])
    X = sp.random(30, 64, density=0.15, format="csr", dtype=np.float32, random_state=rng)
    X.data = (X.data * 50 + 0.001).astype(np.float32)
    res = g.minhash_many(X)
    arrays["w_csr_indptr"], arrays["w_csr_indices"], arrays["w_csr_data"] = X.indptr, X.indices, X.data
    arrays["w_csr_nonempty"] = np.array([r is not None for r in res])
    arrays["w_csr_out"] = np.stack([np.zeros((32, 2), dtype=np.int64) if r is None else r.hashvalues for r in res])
    # single-vector minhash() uses a differently rounded formula (weighted_minhash.py:155-156)
    arrays["w_single_out"] = np.stack([g.minhash(dense[i]).hashvalues for i in (0, 1, 2)])
    # a config-4-shaped slice: dense strictly positive, dim 4096 would be big; use dim=512,S=128 here
    g2 = WeightedMinHashGenerator(512, 128, 1)
    x2 = np.random.RandomState(42).uniform(0, 100, (6, 512)).astype(np.float32)
    arrays["w2_in"] = x2
    arrays["w2_out"] = np.stack([r.hashvalues for r in g2.minhash_many(x2)])

    # ---- 10b. the inverse wire formats as the reference itself reads them (round 6: bulk deserialize / unpack)
    for bo, name in (("<", "le"), (">", "be"), ("@", "native"), ("!", "network")):
        buf = bytearray(lm.bytesize(bo))
        lm.serialize(buf, bo)
        back = LeanMinHash.deserialize(buf, bo)
        meta[f"lean_deserialize_{name}"] = {"bytes": bytes(buf).hex(), "seed": int(back.seed), "hashvalues": [int(v) for v in back.hashvalues]}
    restored = {}
    for b in (1, 2, 3, 7, 13, 32):
        back = pickle.loads(pickle.dumps(bBitMinHash(m48, b)))   # bBitMinHash.__setstate__
        restored[str(b)] = [int(v) for v in back.hashvalues]
    meta["bbit_restored_k48"] = restored

    # ---- 11. pickles of the reference classes (cross-load check for our mirrors' state layout)
    meta["minhash_pickle_keys"] = sorted(MinHash(4, 1).__getstate__().keys())

    os.makedirs(OUT_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(OUT_DIR, "golden.npz"), **arrays)
    with open(os.path.join(OUT_DIR, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    size = os.path.getsize(os.path.join(OUT_DIR, "golden.npz"))
    print(f"wrote {len(arrays)} arrays ({size/1024:.0f} KiB) + golden.json to {os.path.normpath(OUT_DIR)}")


if __name__ == "__main__":
    main()
