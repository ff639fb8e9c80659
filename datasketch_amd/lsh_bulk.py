"""Bulk helpers on the consumer side of the signature matrix.

The reference's ``MinHashLSH`` stays what it is -- host-side control plane that duck-types on
``.hashvalues`` -- and keeps working on our objects unchanged.  What this module adds is the
whole-matrix form of the three per-object steps around it, so that 10^6 signatures do not cost 10^6
Python round trips:

* :func:`band_keys` / :func:`band_digests` -- every band key of every signature in one pass
  (ref: datasketch/lsh.py:199,344,537-543), as the exact key bytes or as 64-bit FNV-1a digests of them;
* :func:`insert_bulk` -- ``MinHashLSH.insert`` for a whole matrix (ref: lsh.py:326-347), through the
  index's own storage API, leaving it in exactly the state the per-key loop would;
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
    return gpu_mode == "detect" and _native.gpu_available()


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
    """``[N, b]`` uint64: FNV-1a-64 of every band key (equal digests <=> same bucket)."""
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


def insert_bulk(lsh, keys: Iterable[Hashable], signatures, check_duplication: bool = True, gpu_mode: str = "detect") -> None:
    """``for key, row in zip(keys, signatures): lsh.insert(key, MinHash(hashvalues=row))`` in bulk.

    ``lsh`` is a ``datasketch.MinHashLSH`` (or anything with its attributes ``h, b, r, keys,
    hashtables, prepickle, hashfunc``).  Validation, key pickling, duplicate check and the storage
    calls are those of ref: datasketch/lsh.py:326-347; only the ``b`` band keys per signature come
    from one pass over the matrix instead of ``b`` numpy slices per object."""
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
        keys = [pickle.dumps(key) for key in keys]
    if check_duplication:
        seen = set()
        for key in keys:
            if key in seen or key in lsh.keys:
                raise ValueError("The given key already exists")
            seen.add(key)
    columns = band_keys(sig, lsh.b, lsh.r * words, gpu_mode=gpu_mode).T.tolist()  # b lists of N bytes objects
    hashfunc = getattr(lsh, "hashfunc", None)
    if hashfunc is not None:
        columns = [[hashfunc(h) for h in col] for col in columns]
    for i, key in enumerate(keys):
        lsh.keys.insert(key, *[col[i] for col in columns], buffer=False)
    for col, hashtable in zip(columns, lsh.hashtables):
        for h, key in zip(col, keys):
            hashtable.insert(h, key, buffer=False)


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
