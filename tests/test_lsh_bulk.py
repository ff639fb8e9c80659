"""Consumer-side bulk helpers (datasketch_amd/lsh_bulk.py): band keys, digests, bulk LSH insert,
candidate pairs, batched Jaccard.  The CPU tests use the numpy fallbacks; where the reference
repository is mounted (/root/reference, build container only) its own MinHashLSH is the oracle."""
import os
import sys

import numpy as np
import pytest

from datasketch_amd import MinHash, prehashed
from datasketch_amd import lsh_bulk as LB

REFERENCE = "/root/reference"


def _signatures(n=300, t=40, k=64, seed=0, dup_every=7):
    rng = np.random.RandomState(seed)
    tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
    tok[dup_every::dup_every] = tok[0:1]  # clusters of identical sets
    near = tok[1].copy()
    near[:4] = rng.randint(0, 2**32, 4, dtype=np.uint64)
    tok[2] = near  # a near duplicate of row 1
    return MinHash.bulk_signatures(tok, num_perm=k, seed=1, hashfunc=prehashed, gpu_mode="disable")


def test_band_keys_are_the_reference_key_bytes():
    sig = _signatures()
    keys = LB.band_keys(sig, 8, 8, gpu_mode="disable")
    assert keys.shape == (sig.shape[0], 8)
    for i in (0, 5, 299):
        for j in (0, 3, 7):
            assert bytes(keys[i, j]) == bytes(sig[i, j * 8 : (j + 1) * 8].byteswap().data)  # lsh.py:537-538
    assert keys.T.tolist()[2][5] == bytes(keys[5, 2])
    with pytest.raises(ValueError):
        LB.band_keys(sig, 9, 8, gpu_mode="disable")


def test_band_digests_are_fnv1a_of_the_key_bytes():
    sig = _signatures(n=50)
    dig = LB.band_digests(sig, 4, 16, gpu_mode="disable")
    keys = LB.band_keys(sig, 4, 16, gpu_mode="disable")
    for i in (0, 7, 49):
        for j in range(4):
            assert int(dig[i, j]) == LB.fnv1a_64(bytes(keys[i, j]))
    assert LB.fnv1a_64(b"") == 0xCBF29CE484222325 and LB.fnv1a_64(b"a") == 0xAF63DC4C8601EC8C  # published test vectors


def test_candidate_pairs_and_jaccard_pairs():
    sig = _signatures()
    pairs = LB.candidate_pairs(sig, 16, 4, gpu_mode="disable")
    # brute force: rows sharing any band
    bands = sig[:, : 16 * 4].reshape(sig.shape[0], 16, 4)
    want = set()
    for i in range(sig.shape[0]):
        same = np.any(np.all(bands[i][None] == bands, axis=2), axis=1)
        want.update((i, int(j)) for j in np.flatnonzero(same) if j > i)
    assert set(map(tuple, pairs.tolist())) == want and len(pairs) == len(want)
    jac = LB.jaccard_pairs(sig, pairs, gpu_mode="disable")
    for (i, j), est in list(zip(pairs.tolist(), jac))[:200]:
        a, b = MinHash(seed=1, hashvalues=sig[i]), MinHash(seed=1, hashvalues=sig[j])
        assert est == a.jaccard(b)
    assert LB.jaccard_pairs(sig, np.empty((0, 2), np.int64), gpu_mode="disable").size == 0


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference repository not mounted")
@pytest.mark.parametrize("prepickle,use_hashfunc", [(False, False), (True, False), (False, True)])
def test_insert_bulk_leaves_the_reference_index_in_the_same_state(prepickle, use_hashfunc):
    sys.path.insert(0, REFERENCE)
    try:
        import datasketch as ref
    finally:
        sys.path.remove(REFERENCE)
    sig = _signatures(n=200, k=64)
    keys = [f"doc-{i}" for i in range(sig.shape[0])]
    kw = dict(threshold=0.5, num_perm=64, prepickle=prepickle)
    if use_hashfunc:
        kw["hashfunc"] = LB.fnv1a_64
    one, bulk = ref.MinHashLSH(**kw), ref.MinHashLSH(**kw)
    for key, row in zip(keys, sig):
        one.insert(key, ref.MinHash(num_perm=64, seed=1, hashvalues=row))
    LB.insert_bulk(bulk, keys, sig, gpu_mode="disable")
    assert one.b == bulk.b and one.r == bulk.r
    for t1, t2 in zip(one.hashtables, bulk.hashtables):
        assert dict(t1._dict) == dict(t2._dict)
    assert dict(one.keys._dict) == dict(bulk.keys._dict)
    probe = ref.MinHash(num_perm=64, seed=1, hashvalues=sig[0])
    assert sorted(one.query(probe)) == sorted(bulk.query(probe)) and len(one.query(probe)) > 1
    with pytest.raises(ValueError):
        LB.insert_bulk(bulk, keys[:1], sig[:1], gpu_mode="disable")  # duplicate key
    with pytest.raises(ValueError):
        LB.insert_bulk(bulk, ["x"], sig[:1, :32], gpu_mode="disable")  # wrong length
    if use_hashfunc:  # the digests are what the index stores
        dig = LB.band_digests(sig, one.b, one.r, gpu_mode="disable")
        assert set(one.hashtables[0]._dict) == set(int(x) for x in dig[:, 0])


def _weighted_signatures(n=60, dim=32, s=16):
    from datasketch_amd import WeightedMinHashGenerator

    g = WeightedMinHashGenerator(dim, s, seed=3, gpu_mode="disable")
    x = np.random.RandomState(0).uniform(0, 5, (n, dim)).astype(np.float32)
    x[7] = x[3]
    x[11, :4] += 1.0
    wm = g.minhash_many(x)
    return wm, np.stack([w.hashvalues for w in wm])


def test_weighted_signatures_go_through_the_same_helpers():
    """An [N, S, 2] int64 WeightedMinHash matrix: a band key is the 16*r bytes the reference builds from
    hashvalues[i*r:(i+1)*r] (lsh.py:537-538 on an [r, 2] int64 slice)."""
    wm, sig = _weighted_signatures()
    assert sig.shape == (60, 16, 2) and sig.dtype == np.int64
    keys = LB.band_keys(sig, 4, 4, gpu_mode="disable")
    for i in (0, 5, 59):
        for j in range(4):
            assert bytes(keys[i, j]) == bytes(sig[i, j * 4 : (j + 1) * 4].byteswap().data)
    dig = LB.band_digests(sig, 4, 4, gpu_mode="disable")
    assert int(dig[5, 2]) == LB.fnv1a_64(bytes(keys[5, 2]))
    pairs = LB.candidate_pairs(sig, 4, 4, gpu_mode="disable")
    assert [3, 7] in pairs.tolist()
    jac = LB.weighted_jaccard_pairs(sig, pairs)
    for (i, j), est in zip(pairs.tolist(), jac):
        assert est == wm[i].jaccard(wm[j])
    with pytest.raises(ValueError):
        LB.jaccard_pairs(sig, pairs, gpu_mode="disable")
    with pytest.raises(ValueError):
        LB.band_keys(sig, 5, 4, gpu_mode="disable")  # 5 * 4 > 16 samples


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference repository not mounted")
def test_insert_bulk_of_weighted_signatures_matches_the_reference_index():
    sys.path.insert(0, REFERENCE)
    try:
        import datasketch as ref
    finally:
        sys.path.remove(REFERENCE)
    wm, sig = _weighted_signatures()
    one, bulk = ref.MinHashLSH(threshold=0.5, num_perm=16), ref.MinHashLSH(threshold=0.5, num_perm=16)
    for i, w in enumerate(wm):
        one.insert(i, ref.WeightedMinHash(3, w.hashvalues))
    LB.insert_bulk(bulk, list(range(len(wm))), sig, gpu_mode="disable")
    for t1, t2 in zip(one.hashtables, bulk.hashtables):
        assert dict(t1._dict) == dict(t2._dict)
    probe = ref.WeightedMinHash(3, wm[3].hashvalues)
    assert sorted(one.query(probe)) == sorted(bulk.query(probe)) and 7 in bulk.query(probe)
    with pytest.raises(ValueError):
        LB.insert_bulk(bulk, ["x"], sig[:1, :8], gpu_mode="disable")  # wrong sample count


def _reference():
    sys.path.insert(0, REFERENCE)
    try:
        import datasketch as ref
    finally:
        sys.path.remove(REFERENCE)
    return ref


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference repository not mounted")
@pytest.mark.parametrize("prepickle", [True, False])
def test_insert_bulk_in_batches_and_query_bulk_equal_the_per_key_calls(prepickle):
    """Two bulk batches (the second onto a non-empty index, with band keys shared inside the batch and with the
    first batch) leave every dictionary as the per-key loop does; query_bulk == [lsh.query(m) for m in probes]
    for 1000 probes (ref: datasketch/lsh.py:326-347, 370-431; storage.py:210-259)."""
    ref = _reference()
    rng = np.random.RandomState(3)
    n, k = 1500, 64
    sig = rng.randint(0, 2**32, (n, k)).astype(np.uint64)
    sig[700:760] = sig[0:60]            # whole rows again, across the batch boundary
    sig[900:940, :32] = sig[5, :32]     # a big bucket in the first bands
    sig[1200, 16:] = sig[1100, 16:]     # a near duplicate inside the second batch
    keys = [("doc", i) for i in range(n)] if prepickle else [b"doc-%d" % i for i in range(n)]
    kw = dict(threshold=0.6, num_perm=k, prepickle=prepickle)
    one, bulk = ref.MinHashLSH(**kw), ref.MinHashLSH(**kw)
    for key, row in zip(keys, sig):
        one.insert(key, ref.MinHash(num_perm=k, seed=1, hashvalues=row))
    LB.insert_bulk(bulk, keys[:800], sig[:800], gpu_mode="disable")
    LB.insert_bulk(bulk, keys[800:], sig[800:], gpu_mode="disable")
    assert dict(one.keys._dict) == dict(bulk.keys._dict)
    for t1, t2 in zip(one.hashtables, bulk.hashtables):
        assert dict(t1._dict) == dict(t2._dict)
    probes = sig[rng.randint(0, n, 1000)].copy()
    probes[::3, rng.randint(0, k, 20)] = 7          # perturbed probes: some bands still match
    probes[1::50] = rng.randint(0, 2**32, (20, k))  # probes that match nothing
    got = LB.query_bulk(bulk, probes, gpu_mode="disable")
    assert len(got) == 1000
    for row, res in zip(probes, got):
        want = one.query(ref.MinHash(num_perm=k, seed=1, hashvalues=row))
        assert sorted(map(repr, res)) == sorted(map(repr, want))
    assert any(len(r) > 30 for r in got) and any(len(r) == 0 for r in got)
    with pytest.raises(ValueError):
        LB.query_bulk(bulk, probes[:, :32], gpu_mode="disable")
    with pytest.raises(ValueError):
        LB.insert_bulk(bulk, keys[10:12], sig[10:12], gpu_mode="disable")   # present already
    with pytest.raises(ValueError):
        LB.insert_bulk(bulk, [keys[0] + (1,) if prepickle else b"x", keys[0] + (1,) if prepickle else b"x"], sig[:2], gpu_mode="disable")  # twice in one batch


class _ListStore:
    """A storage that is NOT the in-memory dict one (no _dict): insert_bulk / query_bulk must go through
    its API key by key (what a Redis / Cassandra back end gets, ref: storage.py)."""

    def __init__(self, factory):
        self.data, self.factory, self.calls = {}, factory, 0

    def insert(self, key, *vals, **kwargs):
        self.calls += 1
        box = self.data.setdefault(key, self.factory())
        (box.extend if isinstance(box, list) else box.update)(vals)

    def get(self, key):
        return self.data.get(key, self.factory())

    def __contains__(self, key):
        return key in self.data


def test_other_storages_take_the_per_key_path():
    sig = _signatures(n=120, k=64)

    class Index:
        h, b, r, prepickle, hashfunc = 64, 8, 8, False, None

        def __init__(self):
            self.keys = _ListStore(list)
            self.hashtables = [_ListStore(set) for _ in range(8)]

    idx = Index()
    keys = [b"k%d" % i for i in range(120)]
    LB.insert_bulk(idx, keys, sig, gpu_mode="disable")
    assert idx.keys.calls == 120 and all(t.calls == 120 for t in idx.hashtables)
    bk = LB.band_keys(sig, 8, 8, gpu_mode="disable")
    assert idx.keys.data[keys[3]] == [bytes(bk[3, j]) for j in range(8)]
    assert keys[7] in idx.hashtables[2].data[bytes(bk[7, 2])]
    got = LB.query_bulk(idx, sig[:10], gpu_mode="disable")
    assert all(keys[i] in got[i] for i in range(10)) and set(got[0]) >= {keys[0], keys[7], keys[14]}  # rows 0, 7, 14 ... are identical sets
    with pytest.raises(ValueError):
        LB.insert_bulk(idx, keys[:1], sig[:1], gpu_mode="disable")


def test_insert_bulk_settle_choices(monkeypatch):
    """The new containers are settled as asked: one full collection, gc.freeze(), or nothing -- and only when the
    collector was on and the batch is big enough to matter."""
    import collections
    import gc

    class Store:
        def __init__(self, factory):
            self._dict = collections.defaultdict(factory)

    class Index:
        h, b, r, prepickle, hashfunc = 64, 8, 8, False, None

        def __init__(self):
            self.keys = Store(list)
            self.hashtables = [Store(set) for _ in range(8)]

    sig = _signatures(n=120, k=64)
    keys = [b"k%d" % i for i in range(120)]
    calls = []
    monkeypatch.setattr(gc, "collect", lambda *a: calls.append("collect") or 0)
    monkeypatch.setattr(gc, "freeze", lambda: calls.append("freeze"))
    monkeypatch.setattr(LB, "_SETTLE_MIN_KEYS", 100)
    states = []
    for settle in ("collect", "freeze", "leave"):
        idx = Index()
        LB.insert_bulk(idx, keys, sig, gpu_mode="disable", settle=settle)
        states.append(dict(idx.hashtables[3]._dict))
    assert calls == ["collect", "freeze"] and states[0] == states[1] == states[2]
    monkeypatch.setattr(LB, "_SETTLE_MIN_KEYS", 1000)
    LB.insert_bulk(Index(), keys, sig, gpu_mode="disable")
    assert calls == ["collect", "freeze"]  # small batch: left alone
    assert gc.isenabled()
    with pytest.raises(ValueError):
        LB.insert_bulk(Index(), keys, sig, gpu_mode="disable", settle="later")
