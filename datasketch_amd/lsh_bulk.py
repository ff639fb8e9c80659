"""Bulk helpers on the consumer side of the signature matrix.

The reference's ``MinHashLSH`` stays what it is -- host-side control plane that duck-types on
``.hashvalues`` -- and keeps working on our objects unchanged.  What this module adds is the
whole-matrix form of the three per-object steps around it, so that 10^6 signatures do not cost 10^6
Python round trips:

* :func:`band_keys` / :func:`band_digests` -- every band key of every signature in one pass
  (ref: datasketch/lsh.py:199,344,537-543), as the exact key bytes or as 64-bit FNV-1a digests of them;
* :func:`insert_bulk` -- ``MinHashLSH.insert`` for a whole matrix (ref: lsh.py:326-347), leaving the index
  in exactly the state the per-key loop would: for the in-memory storage (``storage.py:210-259``) every band's
  ``defaultdict(set)`` is built by ONE ``dict.update`` over C-level iterators instead of N ``insert`` calls;
  other back ends (Redis, Cassandra) go through the index's own storage API key by key;
* :func:`query_bulk` -- ``MinHashLSH.query`` (ref: lsh.py:370-431) for a whole matrix of probes;
  :class:`SortedBandsIndex` -- the same question answered on the device against an index held as sorted
  bands (binary search of M x b probe digests, exact band-key verification, sort + unique);
* :func:`sorted_bands` / :func:`candidate_pairs` -- LSH bucketing by sort: per band the digests in
  ascending order with their rows (device radix sort), and from that the pairs of rows that share at
  least one band (what ``query`` would find), without probing dictionaries -- on the device the
  whole chain (digests, sorts, run detection, pair emission, sort + unique) is one call;
* :func:`jaccard_pairs` -- ``MinHash.jaccard`` for a list of pairs (ref: datasketch/minhash.py:299-324).

Every helper except :func:`jaccard_pairs` also takes a WeightedMinHash matrix ``[N, S, 2]`` int64
(``WeightedMinHashGenerator.minhash_many_arrays``): its band keys are the reference's too
(16*r bytes per band); :func:`weighted_jaccard_pairs` is ``WeightedMinHash.jaccard`` for pairs.
"""
from __future__ import annotations

import pickle
from typing import Hashable, Iterable, List, Optional, Sequence

import numpy as np

from datasketch_amd import _native

_FNV_OFFSET = np.uint64(0xCBF29CE484222325)
_FNV_PRIME = np.uint64(0x100000001B3)


def fnv1a_64(data: bytes) -> int:
    """64-bit FNV-1a of a byte string.  ``MinHashLSH(hashfunc=fnv1a_64)`` stores exactly the values
    :func:`band_digests` computes."""
    h = 0xCBF29CE484222325
    for byte in bytes(data):
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _use_gpu(gpu_mode: str) -> bool:
    if gpu_mode == "always":
        if not _native.gpu_available():
            raise RuntimeError("GPU mode 'always' requested but no MI355X / libmhx.so is available.")
        return True
    return gpu_mode == "detect" and _native.gpu_detected()


def _matrix(signatures) -> np.ndarray:
    """``[N, K]`` uint64 view of a signature matrix.  A WeightedMinHash matrix ``[N, S, 2]`` int64 (rows of
    ``(k, t)`` pairs, ref: weighted_minhash.py:11-30) is viewed as ``[N, 2S]`` words: the reference's band
    key of ``hashvalues[i*r:(i+1)*r]`` is then the key of ``2r`` consecutive words (see :func:`_words`)."""
    sig = np.asarray(signatures)
    if sig.ndim == 3 and sig.shape[2] == 2:
        sig = np.ascontiguousarray(sig, dtype=np.int64).view(np.uint64).reshape(sig.shape[0], 2 * sig.shape[1])
        return sig
    sig = np.ascontiguousarray(sig, dtype=np.uint64)
    if sig.ndim != 2:
        raise ValueError("signatures must be an [N, K] matrix (or [N, S, 2] for WeightedMinHash)")
    return sig


def _words(signatures) -> int:
    """uint64 words per hash value: 2 for a WeightedMinHash matrix ``[N, S, 2]``, else 1."""
    return 2 if np.ndim(signatures) == 3 else 1


def _check_params(k: int, b: int, r: int) -> None:
    if b <= 0 or r <= 0 or b * r > k:
        raise ValueError("b*r must be in (0, num_perm]")


def band_keys(signatures, b: int, r: int, gpu_mode: str = "detect") -> np.ndarray:
    """``[N, b]`` array of ``numpy.void`` items of ``8*r`` bytes: item ``[i, j]`` holds exactly
    ``MinHashLSH._H(hashvalues[j*r:(j+1)*r])`` of row ``i`` (big-endian words, ref: lsh.py:537-538).
    ``.tolist()`` gives nested lists of ``bytes``; ``bytes(out[i, j])`` one key."""
    sig, r = _matrix(signatures), r * _words(signatures)
    n, k = sig.shape
    _check_params(k, b, r)
    if _use_gpu(gpu_mode):
        swapped = _native.context().band_keys(sig, b, r)
    else:
        swapped = sig[:, : b * r].byteswap()
    swapped = np.ascontiguousarray(swapped).reshape(n, b * r)
    return swapped.view(np.dtype((np.void, 8 * r))).reshape(n, b)


def band_digests(signatures, b: int, r: int, gpu_mode: str = "detect") -> np.ndarray:
    """``[N, b]`` uint64: FNV-1a-64 of every band key.  Equal band keys give equal digests; different band keys
    give different digests except for 64-bit hash collisions (one expected among ~6*10^9 keys of one band).
    :func:`sorted_bands` / :func:`candidate_pairs` group by digest, so such a collision would add one candidate
    pair the reference's byte-keyed dictionaries would not report -- harmless where candidates are verified
    with :func:`jaccard_pairs` afterwards; :class:`SortedBandsIndex` compares the band's values themselves."""
    sig, r = _matrix(signatures), r * _words(signatures)
    n, k = sig.shape
    _check_params(k, b, r)
    if _use_gpu(gpu_mode):
        return _native.context().band_digests(sig, b, r)
    key_bytes = np.ascontiguousarray(sig[:, : b * r].byteswap()).view(np.uint8).reshape(n, b, 8 * r)
    h = np.full((n, b), _FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for c in range(8 * r):
            h = (h ^ key_bytes[:, :, c].astype(np.uint64)) * _FNV_PRIME
    return h


def _is_dict_index(lsh) -> bool:
    """The reference's in-memory storage: ``keys`` a DictListStorage, every hashtable a DictSetStorage
    (ref: datasketch/storage.py:210-259), recognised by what they are made of."""
    import collections

    def plain(store, factory):
        d = getattr(store, "_dict", None)
        return isinstance(d, collections.defaultdict) and d.default_factory is factory

    return plain(lsh.keys, list) and all(plain(h, set) for h in lsh.hashtables)


def insert_bulk(lsh, keys: Iterable[Hashable], signatures, check_duplication: bool = True, gpu_mode: str = "detect",
                settle: str = "collect") -> None:
    """``for key, row in zip(keys, signatures): lsh.insert(key, MinHash(hashvalues=row))`` in bulk.

    ``lsh`` is a ``datasketch.MinHashLSH`` (or anything with its attributes ``h, b, r, keys,
    hashtables, prepickle, hashfunc``).  Validation, key pickling and the duplicate check are those of
    ref: datasketch/lsh.py:326-347; the ``b`` band keys per signature come from one pass over the matrix.
    With the in-memory storage the dictionaries are filled wholesale: ``keys`` by one ``dict.update`` of
    ``key -> [H_0 .. H_{b-1}]`` and every band's table by one ``dict.update`` of ``H -> {key}`` for the band
    keys that occur once and are new, plus a loop over the (few) band keys shared by several rows or already
    present -- the resulting state equals the per-key loop's.  Other storages take the per-key calls.

    ``settle`` says what happens to the ``N * (b + 1)`` new containers once the dictionaries are built (the
    cyclic collector is held off while they are): ``"collect"`` runs one full collection, which moves them
    into the oldest generation in a single walk (left young they would be walked three times, by whatever
    code allocates next -- 4.9 s landed on the first ``query_bulk`` after 300 000 keys); ``"freeze"`` calls
    ``gc.freeze()`` instead (no walk now or ever, process-wide effect); ``"leave"`` does neither."""
    if settle not in ("collect", "freeze", "leave"):
        raise ValueError("settle must be 'collect', 'freeze' or 'leave'")
    sig, words = _matrix(signatures), _words(signatures)
    n, k = sig.shape
    if k != lsh.h * words:
        raise ValueError("Expecting minhash with length %d, got %d" % (lsh.h, k // words))
    keys = list(keys)
    if len(keys) != n:
        raise ValueError("keys and signatures must have the same length")
    if getattr(lsh, "_require_bytes_keys", False):
        for key in keys:
            if not isinstance(key, bytes):
                raise TypeError(
                    f"prepickle=False requires bytes keys for non-dict storage, got {type(key).__name__}. "
                    "Either pass bytes keys or use prepickle=True for automatic serialization."
                )
    if lsh.prepickle:
        keys = list(map(pickle.dumps, keys))
    dict_index = _is_dict_index(lsh)
    if check_duplication:
        if dict_index:
            if len(set(keys)) != n or not lsh.keys._dict.keys().isdisjoint(keys):
                raise ValueError("The given key already exists")
        else:
            seen = set()
            for key in keys:
                if key in seen or key in lsh.keys:
                    raise ValueError("The given key already exists")
                seen.add(key)
    columns = band_keys(sig, lsh.b, lsh.r * words, gpu_mode=gpu_mode).T.tolist()  # b lists of N bytes objects
    hashfunc = getattr(lsh, "hashfunc", None)
    if hashfunc is not None:
        columns = [list(map(hashfunc, col)) for col in columns]
    if not dict_index or (not check_duplication and len(set(keys)) != n):
        for i, key in enumerate(keys):
            lsh.keys.insert(key, *[col[i] for col in columns], buffer=False)
        for col, hashtable in zip(columns, lsh.hashtables):
            for h, key in zip(col, keys):
                hashtable.insert(h, key, buffer=False)
        return
    # ---- in-memory storage: whole dictionaries at a time (every iterator below runs in C).  The cyclic garbage
    # collector is held off meanwhile: N*(b+1) new containers, none of them part of a cycle, would otherwise trigger
    # full collections that walk everything built so far (17 s instead of 2.8 s for 200k keys x 25 bands)
    import gc

    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        _fill_dict_index(lsh, keys, columns, n)
    finally:
        if gc_was_on:
            gc.enable()
    if gc_was_on and n >= _SETTLE_MIN_KEYS:
        if settle == "collect":
            gc.collect()
        elif settle == "freeze":
            gc.freeze()


_SETTLE_MIN_KEYS = 20000  # below this the young containers cost a later collection milliseconds


def _fill_dict_index(lsh, keys, columns, n) -> None:
    kd = lsh.keys._dict
    fresh = kd.keys().isdisjoint(keys)
    rows = map(list, zip(*columns))
    if fresh:
        kd.update(zip(keys, rows))  # DictListStorage.insert: _dict[key].extend(Hs) on an absent key
    else:  # re-inserting existing keys without the duplicate check extends their lists, as the reference does
        for key, hs in zip(keys, rows):
            kd[key].extend(hs)
    for col, hashtable in zip(columns, lsh.hashtables):
        d = hashtable._dict
        first = dict(zip(col, keys))  # band key -> LAST row holding it
        if len(first) == n and d.keys().isdisjoint(first):
            d.update(zip(col, map(set, zip(keys))))  # every bucket is new and holds one key
            continue
        # some band keys are shared (near-duplicate rows) or present already: those few go one by one
        counts: dict = {}
        for h in col:
            counts[h] = counts.get(h, 0) + 1
        shared = {h for h, c in counts.items() if c > 1}
        shared.update(d.keys() & counts.keys())
        if shared:
            single_h, single_k = [], []
            for h, key in zip(col, keys):
                if h in shared:
                    d[h].add(key)
                else:
                    single_h.append(h)
                    single_k.append(key)
            d.update(zip(single_h, map(set, zip(single_k))))
        else:
            d.update(zip(col, map(set, zip(keys))))


def query_bulk(lsh, signatures, gpu_mode: str = "detect") -> List[list]:
    """``[lsh.query(MinHash(hashvalues=row)) for row in signatures]`` (ref: datasketch/lsh.py:370-431) with the band
    keys of all probes taken in one pass and, for the in-memory storage, the ``b`` dictionary probes per
    signature done by C-level ``map(dict.get, ...)``.  Returns one list of keys per row (order within a list
    is unspecified, as in the reference)."""
    sig, words = _matrix(signatures), _words(signatures)
    n, k = sig.shape
    if k != lsh.h * words:
        raise ValueError("Expecting minhash with length %d, got %d" % (lsh.h, k // words))
    columns = band_keys(sig, lsh.b, lsh.r * words, gpu_mode=gpu_mode).T.tolist()
    hashfunc = getattr(lsh, "hashfunc", None)
    if hashfunc is not None:
        columns = [list(map(hashfunc, col)) for col in columns]
    if _is_dict_index(lsh):
        # (no cyclic garbage is made here, and a full collection would walk the whole index: 32 million sets at 10^6
        # keys -- with the collector left on, 20 000 probes against such an index took 3.9 s)
        import gc

        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            empty = frozenset()
            found = [list(map(ht._dict.get, col, [empty] * n)) for col, ht in zip(columns, lsh.hashtables)]  # b lists of N buckets
            results = [set().union(*buckets) for buckets in zip(*found)] if n else []
            if lsh.prepickle:
                return [[pickle.loads(key) for key in cand] for cand in results]
            return [list(cand) for cand in results]
        finally:
            if gc_was_on:
                gc.enable()
    else:
        results = []
        for i in range(n):
            cand = set()
            for col, hashtable in zip(columns, lsh.hashtables):
                cand.update(hashtable.get(col[i]))
            results.append(cand)
    if lsh.prepickle:
        return [[pickle.loads(key) for key in cand] for cand in results]
    return [list(cand) for cand in results]


class SortedBandsIndex:
    """An LSH index over a signature matrix (grown in batches with :meth:`extend`), resident on the GPU as sorted bands: per band the FNV-1a-64
    digests of the band keys in ascending order with their rows (``mhx_lsh_sort_bands``) -- every bucket of
    the reference's per-band dictionary (ref: datasketch/lsh.py:326-347) is a run of equal digests.

    :meth:`query` answers what ``MinHashLSH.query`` would for a whole matrix of probes: binary search of the
    ``M x b`` probe digests, candidates confirmed by comparing the band's ``r`` hash values themselves (so the
    answer is the reference's even under a 64-bit digest collision), sort + unique across bands.  Rows are
    numbers ``0 .. N-1`` of the indexed matrix; map them to keys with your own array.  uint32 signature
    matrices (the compact / all-gathered form) are taken as they are."""

    def __init__(self, signatures, b: int, r: int, device: Optional[int] = None):
        sig = np.asarray(signatures)
        if sig.ndim != 2:
            raise ValueError("signatures must be an [N, K] matrix")
        if sig.dtype != np.uint32:
            sig = np.ascontiguousarray(sig, dtype=np.uint64)
        sig = np.ascontiguousarray(sig)
        _check_params(sig.shape[1], b, r)
        if not _native.gpu_available():
            raise RuntimeError("SortedBandsIndex needs an MI355X / libmhx.so")
        self.ctx = _native.context(device)
        self.n, self.k = sig.shape
        self.b, self.r = int(b), int(r)
        self.dtype = sig.dtype
        self._code = _native.MHX_U32 if sig.dtype == np.uint32 else _native.MHX_U64
        self._d_sig = self.ctx.to_device(sig)
        self._sort()

    def _sort(self) -> None:
        self._d_dig = self.ctx.alloc(max(1, self.n * self.b * 8))
        self._d_rows = self.ctx.alloc(max(1, self.n * self.b * 4))
        if self.n:
            _native.check(self.ctx.lib.mhx_lsh_sort_bands_dev_typed(self.ctx.handle, self._d_sig.ptr, self._code, self.n, self.k,
                                                                    self.b, self.r, self._d_dig.ptr, self._d_rows.ptr))

    def _as_index_dtype(self, sig: np.ndarray) -> np.ndarray:
        """``sig`` in the index's signature type; a wider matrix must fit (a wrapped value would match band keys the
        reference's byte-keyed dictionaries keep apart)."""
        if sig.dtype.kind not in "ui":
            raise ValueError("signatures are unsigned integers (hashvalues), not %s" % sig.dtype)
        if sig.size and sig.dtype.kind == "i" and int(sig.min()) < 0:
            raise ValueError("negative signature values")
        if self.dtype == np.uint32 and sig.dtype.itemsize > 4 and sig.size and int(sig.max()) > 0xFFFFFFFF:
            raise ValueError("signature values >= 2**32 do not fit this uint32 index")
        return np.ascontiguousarray(sig, dtype=self.dtype)

    def extend(self, signatures) -> range:
        """Add rows (what ``MinHashLSH.insert`` does key by key, ref: datasketch/lsh.py:326-347) and return their row
        numbers ``range(old N, new N)``.  The matrix grows on the device (the rows already there are not uploaded
        again) and the bands are sorted afresh -- 40 ms per 10^6 rows, so add in batches."""
        more = np.asarray(signatures)
        if more.ndim != 2 or more.shape[1] != self.k:
            raise ValueError("Expecting minhash with length %d, got %d" % (self.k, more.shape[-1]))
        more = self._as_index_dtype(more)
        first, m = self.n, more.shape[0]
        if m == 0:
            return range(first, first)
        if (first + m) >> 32:
            raise ValueError("a SortedBandsIndex holds fewer than 2^32 rows")
        row_bytes = self.k * self.dtype.itemsize
        grown = self.ctx.alloc((first + m) * row_bytes)
        if first:
            self.ctx.copy_dev(grown.ptr, self._d_sig.ptr, first * row_bytes)
        grown.upload(more, offset=first * row_bytes)
        self.ctx.synchronize()
        self._d_sig, self.n = grown, first + m
        self._sort()
        return range(first, first + m)

    def query(self, signatures, capacity: Optional[int] = None):
        """``(offsets int64[M+1], rows int64[...])``: the index rows sharing at least one band key with probe
        ``i`` are ``rows[offsets[i]:offsets[i+1]]``, ascending."""
        import ctypes

        q = np.asarray(signatures)
        if q.ndim != 2 or q.shape[1] != self.k:
            raise ValueError("Expecting minhash with length %d, got %d" % (self.k, q.shape[-1]))
        q = self._as_index_dtype(q)
        m = q.shape[0]
        offsets = np.zeros(m + 1, dtype=np.int64)
        if m == 0 or self.n == 0:
            return offsets, np.empty(0, dtype=np.int64)
        d_q = self.ctx.to_device(q)
        cap = int(capacity) if capacity is not None else max(4 * m, 1 << 16)
        while True:
            d_pairs = self.ctx.alloc(cap * 16)
            found = ctypes.c_int64(0)
            _native.check(self.ctx.lib.mhx_lsh_query_dev(self.ctx.handle, self._d_dig.ptr, self._d_rows.ptr, self.n, self.b, self.r,
                                                         d_q.ptr, self._d_sig.ptr, self._code, self.k, m, d_pairs.ptr, cap,
                                                         ctypes.byref(found)))
            if found.value <= cap:
                break
            cap = int(found.value)
        self.ctx.synchronize()
        pairs = d_pairs.download((found.value, 2), np.int64) if found.value else np.empty((0, 2), dtype=np.int64)
        np.cumsum(np.bincount(pairs[:, 0], minlength=m), out=offsets[1:])
        return offsets, np.ascontiguousarray(pairs[:, 1])


def sorted_bands(signatures, b: int, r: int, gpu_mode: str = "detect"):
    """``(digests [b, N] uint64 ascending per band, rows [b, N] uint32 in the same order)``: every LSH
    bucket of band ``j`` is a run of equal values in ``digests[j]``.  On the device this is one digest
    pass plus ``b`` radix sorts (mhx_lsh_sort_bands); the numpy fallback is ``argsort`` per band."""
    sig, r = _matrix(signatures), r * _words(signatures)
    n, k = sig.shape
    _check_params(k, b, r)
    if _use_gpu(gpu_mode) and n:
        return _native.context().lsh_sort_bands(sig, b, r)
    dig = band_digests(sig, b, r, gpu_mode="disable").T
    order = np.argsort(dig, axis=1, kind="stable")
    return np.take_along_axis(dig, order, axis=1), order.astype(np.uint32)


def candidate_pairs(signatures, b: int, r: int, gpu_mode: str = "detect") -> np.ndarray:
    """``[M, 2]`` int64, sorted, unique pairs ``i < j`` of rows that share the key of at least one band
    -- the pairs ``MinHashLSH(params=(b, r))`` would report for each other.  On the device: digests,
    per-band sort, run detection, pair emission, sort + unique in one call (mhx_lsh_candidate_pairs)."""
    sig, r = _matrix(signatures), r * _words(signatures)
    _check_params(sig.shape[1], b, r)
    if _use_gpu(gpu_mode) and sig.shape[0]:
        return _native.context().lsh_candidate_pairs(sig, b, r)[0]
    dig, rows = sorted_bands(sig, b, r, gpu_mode=gpu_mode)
    n = dig.shape[1]
    found: List[np.ndarray] = []
    for j in range(b):
        s, order = dig[j], rows[j].astype(np.int64)
        starts = np.flatnonzero(np.concatenate([[True], s[1:] != s[:-1]])) if n else np.empty(0, np.int64)
        ends = np.concatenate([starts[1:], [n]]) if n else starts
        size = ends - starts
        two = starts[size == 2]  # the common bucket: exactly two rows
        if two.size:
            pair = np.stack([order[two], order[two + 1]], axis=1)
            found.append(np.sort(pair, axis=1))
        for a, e in zip(starts[size > 2], ends[size > 2]):
            members = np.sort(order[a:e])
            ii, jj = np.triu_indices(members.size, k=1)
            found.append(np.stack([members[ii], members[jj]], axis=1))
    if not found:
        return np.empty((0, 2), dtype=np.int64)
    return np.unique(np.concatenate(found).astype(np.int64), axis=0)


def jaccard_pairs(signatures, pairs, gpu_mode: str = "detect") -> np.ndarray:
    """``MinHash.jaccard`` of rows ``pairs[:, 0]`` and ``pairs[:, 1]``: float64, equal positions / K.
    (For a WeightedMinHash matrix use :func:`weighted_jaccard_pairs`: a position is a ``(k, t)`` pair.)"""
    if np.ndim(signatures) == 3:
        raise ValueError("jaccard_pairs takes an [N, K] matrix; use weighted_jaccard_pairs for [N, S, 2]")
    sig = _matrix(signatures)
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    if pairs.size and (pairs.min() < 0 or pairs.max() >= sig.shape[0]):
        raise ValueError("pair index out of range")
    if _use_gpu(gpu_mode):
        counts = _native.context().jaccard_pairs(sig, pairs)
    else:
        counts = np.count_nonzero(sig[pairs[:, 0]] == sig[pairs[:, 1]], axis=1)
    return counts.astype(np.float64) / float(sig.shape[1])


def weighted_jaccard_pairs(signatures, pairs) -> np.ndarray:
    """``WeightedMinHash.jaccard`` (ref: weighted_minhash.py:41-60) of rows ``pairs[:, 0]`` and ``pairs[:, 1]`` of an
    ``[N, S, 2]`` matrix: the fraction of samples whose ``(k, t)`` pairs are equal."""
    sig = np.asarray(signatures, dtype=np.int64)
    if sig.ndim != 3 or sig.shape[2] != 2:
        raise ValueError("signatures must be [N, S, 2]")
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    if pairs.size and (pairs.min() < 0 or pairs.max() >= sig.shape[0]):
        raise ValueError("pair index out of range")
    same = np.all(sig[pairs[:, 0]] == sig[pairs[:, 1]], axis=2)
    return np.count_nonzero(same, axis=1).astype(np.float64) / float(sig.shape[1])
