"""LeanMinHash with the reference's API and wire format (datasketch/lean_minhash.py).

A frozen MinHash holding only ``seed`` and ``hashvalues``.  The binary format is the
reference's: ``<byteorder> q i {K}I`` = seed (int64), K (int32), K hash values (uint32), so
buffers written by either implementation can be read by the other.  Whole signature matrices
can be serialised on the device with :func:`serialize_matrix`.
"""
from __future__ import annotations

import struct
from typing import Iterable, Optional

import numpy as np

from datasketch_amd import _native
from datasketch_amd.minhash import MinHash


def _header(byteorder: str) -> struct.Struct:
    return struct.Struct(byteorder + "qi")


class LeanMinHash(MinHash):
    """Drop-in for ``datasketch.LeanMinHash``: build from a MinHash, or from ``seed`` + ``hashvalues``."""

    __slots__ = ("hashvalues", "seed")

    def _initialize_slots(self, seed, hashvalues) -> None:
        self.seed = seed
        self.hashvalues = self._parse_hashvalues(hashvalues)

    def __init__(self, minhash: Optional[MinHash] = None, seed: Optional[int] = None, hashvalues: Optional[Iterable] = None):
        if minhash is not None:
            self._initialize_slots(minhash.seed, minhash.hashvalues)
        elif hashvalues is not None and seed is not None:
            self._initialize_slots(seed, hashvalues)
        else:
            raise ValueError("Init parameters cannot be None: make sure to set either minhash or both of hash values and seed")

    @classmethod
    def _from_state(cls, seed, hashvalues: np.ndarray) -> "LeanMinHash":
        """Fast constructor (no validation, no copy), the analogue of the reference's
        ``object.__new__`` + ``_initialize_slots`` idiom (lean_minhash.py:212-214)."""
        lmh = object.__new__(cls)
        lmh.seed = seed
        lmh.hashvalues = hashvalues
        return lmh

    def update(self, b) -> None:
        raise TypeError("Cannot update a LeanMinHash")

    def update_batch(self, b) -> None:
        raise TypeError("Cannot update a LeanMinHash")

    def copy(self) -> "LeanMinHash":
        # The reference's copy() (lean_minhash.py:99-102) passes the slot NAMES and raises; a
        # working copy is what its docstring promises.
        return LeanMinHash._from_state(self.seed, self.hashvalues.copy())

    def bytesize(self, byteorder: str = "@") -> int:
        """Serialized size: 8 (seed) + 4 (length) + 4 per hash value (reference: lean_minhash.py:104-124)."""
        return _header(byteorder).size + len(self) * struct.calcsize(byteorder + "I")

    def serialize(self, buf, byteorder: str = "@") -> None:
        """Write ``seed, K, hashvalues`` into ``buf`` (reference: lean_minhash.py:126-175)."""
        if len(buf) < self.bytesize(byteorder):
            raise ValueError("The buffer does not have enough space for holding this MinHash.")
        head = _header(byteorder)
        head.pack_into(buf, 0, self.seed, len(self))
        # hash values are < 2**32 by construction; astype wraps like a C cast would
        order = byteorder if byteorder in "<>" else ("<" if byteorder in "@=" and np.little_endian else ">")
        if byteorder == "!":
            order = ">"
        payload = self.hashvalues.astype(order + "u4").tobytes()
        if np.any(self.hashvalues > np.uint64(0xFFFFFFFF)):
            raise struct.error("argument out of range")  # what struct.pack raises in the reference
        memoryview(buf)[head.size : head.size + len(payload)] = payload

    @classmethod
    def deserialize(cls, buf, byteorder: str = "@") -> "LeanMinHash":
        """Read a LeanMinHash written by :meth:`serialize` (reference: lean_minhash.py:177-214)."""
        head = _header(byteorder)
        view = memoryview(buf)
        seed, num_perm = head.unpack_from(view, 0)
        order = ">" if byteorder in (">", "!") or (byteorder in "@=" and not np.little_endian) else "<"
        hv = np.frombuffer(view, dtype=order + "u4", count=num_perm, offset=head.size)
        return cls._from_state(seed, hv.astype(np.uint64))

    def __getstate__(self):
        buf = bytearray(self.bytesize())
        self.serialize(buf)
        return buf

    def __setstate__(self, buf):
        other = LeanMinHash.deserialize(buf)
        self.seed, self.hashvalues = other.seed, other.hashvalues

    def __hash__(self) -> int:
        return hash((self.seed, tuple(self.hashvalues)))

    @classmethod
    def union(cls, *lmhs: "LeanMinHash") -> "LeanMinHash":
        if len(lmhs) < 2:
            raise ValueError("Cannot union less than 2 MinHash")
        num_perm = len(lmhs[0])
        seed = lmhs[0].seed
        if any((seed != m.seed or num_perm != len(m)) for m in lmhs):
            raise ValueError("The unioning MinHash must have the same seed, number of permutation functions.")
        return cls._from_state(seed, np.minimum.reduce([m.hashvalues for m in lmhs]))

    # ------------------------------------------------------------------ bulk helpers (new)
    @classmethod
    def from_matrix(cls, signatures: np.ndarray, seed: int) -> list:
        """Wrap the rows of an ``[N, K]`` signature matrix as LeanMinHash objects (views, no copy)."""
        signatures = np.asarray(signatures, dtype=np.uint64)
        return [cls._from_state(seed, row) for row in signatures]


def serialize_matrix(signatures: np.ndarray, seed: int, gpu_mode: str = "always") -> np.ndarray:
    """``LeanMinHash.serialize`` (little-endian) of every row of an ``[N, K]`` matrix at once.

    Returns a uint8 array ``[N, 12 + 4*K]``; row ``i`` equals what ``LeanMinHash(seed=seed,
    hashvalues=signatures[i]).serialize(buf, '<')`` writes.  Runs on the device unless
    ``gpu_mode='disable'``.  A hash value above 2**32-1 does not fit the format's ``I`` field:
    ``struct.error``, as ``LeanMinHash.serialize`` raises for it (lean_minhash.py:174-175).
    """
    signatures = np.ascontiguousarray(signatures, dtype=np.uint64)
    if signatures.size and int(signatures.max()) > 0xFFFFFFFF:
        raise struct.error("'I' format requires 0 <= number <= 4294967295")
    if gpu_mode != "disable" and (gpu_mode == "always" or _native.gpu_detected()):
        return _native.context().lean_serialize(signatures, seed)
    n, k = signatures.shape
    out = np.zeros((n, 12 + 4 * k), dtype=np.uint8)
    out[:, :8] = np.frombuffer(struct.pack("<q", seed), dtype=np.uint8)
    out[:, 8:12] = np.frombuffer(struct.pack("<i", k), dtype=np.uint8)
    out[:, 12:] = signatures.astype("<u4").view(np.uint8).reshape(n, 4 * k)
    return out


_BIG = {">": True, "!": True, "<": False, "=": struct.pack("=I", 1) != struct.pack("<I", 1), "@": struct.pack("@I", 1) != struct.pack("<I", 1)}


def deserialize_matrix(buf, num_perm: Optional[int] = None, byteorder: str = "@", gpu_mode: str = "always"):
    """``LeanMinHash.deserialize`` (ref: datasketch/lean_minhash.py:177-214) of a buffer of N records laid back to back --
    what :func:`serialize_matrix` writes, or N ``serialize`` calls as in the reference's docstring example -- without N
    Python objects.  Returns ``(seeds int64[N], hashvalues uint64[N, K])``.  ``byteorder``: any of the reference's
    (``@ = < > !``).  ``num_perm``: taken from the first record when not given; a record with another length field is a
    ``ValueError`` (the reference would read a different number of values there and mis-frame everything behind it).
    Runs on the device unless ``gpu_mode='disable'``."""
    if byteorder not in _BIG:
        raise struct.error("bad char in struct format")
    raw = np.frombuffer(memoryview(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    big = _BIG[byteorder]
    if raw.size == 0:
        return np.empty(0, dtype=np.int64), np.empty((0, int(num_perm or 0)), dtype=np.uint64)
    if raw.size < 12:
        raise struct.error("unpack_from requires a buffer of at least 12 bytes")
    k = int(np.frombuffer(raw[8:12].tobytes(), dtype=">i4" if big else "<i4")[0]) if num_perm is None else int(num_perm)
    rec = 12 + 4 * k
    if k <= 0 or raw.size % rec:
        raise ValueError("the buffer is not a whole number of %d-byte records (num_perm = %d)" % (rec, k))
    if gpu_mode != "disable" and (gpu_mode == "always" or _native.gpu_detected()):
        try:
            return _native.context().lean_deserialize(raw, k, big)
        except ValueError as e:
            raise ValueError(str(e)) from None
    rows = raw.reshape(-1, rec)
    lengths = rows[:, 8:12].copy().view(">i4" if big else "<i4").reshape(-1)
    if np.any(lengths != k):
        raise ValueError("%d of %d records do not hold %d hash values (length field)" % (int(np.count_nonzero(lengths != k)), rows.shape[0], k))
    seeds = rows[:, :8].copy().view(">i8" if big else "<i8").reshape(-1).astype(np.int64)
    return seeds, rows[:, 12:].copy().view(">u4" if big else "<u4").astype(np.uint64)
