"""b-bit MinHash with the reference's API and pickle format (datasketch/b_bit_minhash.py).

Keeps the lowest ``b`` bits of every hash value.  The packed state is the reference's:
header ``<qBdi`` (seed, b, r, num_perm) followed by uint64 blocks holding ``64/slot`` values
each, value ``j`` of a block at bit ``(n-1-j)*slot``.  :func:`pack_matrix` packs a whole
signature matrix on the device in the same bit order.
"""
from __future__ import annotations

import struct

import numpy as np

from datasketch_amd import _native


def _slot_size(b: int) -> int:
    """Storage slot width for a b-bit value (reference: b_bit_minhash.py:147-160)."""
    for width in (1, 2):
        if b == width:
            return width
    for width in (4, 8, 16, 32):
        if b <= width:
            return width
    raise ValueError("Incorrect value of b")


def _pack_rows(values: np.ndarray, slot: int) -> np.ndarray:
    """values [n, K] (already masked, < 2**slot) -> uint64 blocks [n, ceil(K/(64/slot))]."""
    per = 64 // slot
    n, k = values.shape
    nb = -(-k // per)
    padded = np.zeros((n, nb * per), dtype=np.uint64)
    padded[:, :k] = values
    shifts = (np.arange(per - 1, -1, -1, dtype=np.uint64) * np.uint64(slot)).astype(np.uint64)
    return np.bitwise_or.reduce(padded.reshape(n, nb, per) << shifts, axis=2)


def _unpack_rows(blocks: np.ndarray, slot: int, k: int) -> np.ndarray:
    per = 64 // slot
    shifts = (np.arange(per - 1, -1, -1, dtype=np.uint64) * np.uint64(slot)).astype(np.uint64)
    mask = np.uint64((1 << slot) - 1)
    vals = (blocks[:, :, None] >> shifts) & mask
    return vals.reshape(blocks.shape[0], -1)[:, :k]


class bBitMinHash:
    """Drop-in for ``datasketch.bBitMinHash``."""

    __slots__ = ("b", "hashvalues", "r", "seed")

    _serial_fmt_params = "<qBdi"
    _serial_fmt_block = "Q"

    def __init__(self, minhash, b: int = 1, r: float = 0.0):
        b = int(b)
        r = float(r)
        if b > 32 or b < 0:
            raise ValueError("b must be an integer in [0, 32]")
        if r > 1.0:
            raise ValueError("r must be a float in [0.0, 1.0]")
        bmask = (1 << b) - 1
        self.hashvalues = np.bitwise_and(minhash.hashvalues, np.uint64(bmask)).astype(np.uint32)
        self.seed = minhash.seed
        self.b = b
        self.r = r

    def __eq__(self, other):
        return (
            type(self) is type(other)
            and self.seed == other.seed
            and self.b == other.b
            and self.r == other.r
            and np.array_equal(self.hashvalues, other.hashvalues)
        )

    __hash__ = None

    def jaccard(self, other) -> float:
        """Bias-corrected resemblance estimate (reference: b_bit_minhash.py:53-72)."""
        if self.b != other.b:
            raise ValueError("Cannot compare two b-bit MinHashes with different b values")
        if self.seed != other.seed:
            raise ValueError("Cannot compare two b-bit MinHashes with different set of permutations")
        raw_est = float(np.count_nonzero(self.hashvalues == other.hashvalues)) / float(self.hashvalues.size)
        a1 = self._calc_a(self.r, self.b)
        a2 = self._calc_a(other.r, other.b)
        c1, c2 = self._calc_c(a1, a2, self.r, other.r)
        return (raw_est - c1) / (1 - c2)

    def bytesize(self) -> int:
        return self._bytesize()[-1]

    def __getstate__(self):
        slot, _per, nb, total = self._bytesize()
        buf = bytearray(total)
        struct.pack_into(self._serial_fmt_params, buf, 0, self.seed, self.b, self.r, self.hashvalues.size)
        blocks = _pack_rows(self.hashvalues.astype(np.uint64)[None, :], slot)[0]
        off = struct.calcsize(self._serial_fmt_params)
        buf[off : off + 8 * nb] = blocks.astype("<u8").tobytes()
        return buf

    def __setstate__(self, buf):
        view = memoryview(buf)
        self.seed, self.b, self.r, num_perm = struct.unpack_from(self._serial_fmt_params, view, 0)
        off = struct.calcsize(self._serial_fmt_params)
        self.hashvalues = np.zeros((num_perm,), dtype=np.uint32)
        slot, _per, nb, _total = self._bytesize()
        blocks = np.frombuffer(view, dtype="<u8", count=nb, offset=off).astype(np.uint64)
        self.hashvalues = _unpack_rows(blocks[None, :], slot, num_perm)[0].astype(np.uint32)

    def _calc_a(self, r, b):
        if r == 0.0:
            return 1.0 / (1 << b)
        return r * (1 - r) ** (2**b - 1) / (1 - (1 - r) ** (2 * b))

    def _calc_c(self, a1, a2, r1, r2):
        if r1 == 0.0 and r2 == 0.0:
            return a1, a2
        div = 1 / (r1 + r2)
        return (a1 * r2 + a2 * r1) * div, (a1 * r1 + a2 * r2) * div

    def _find_slot_size(self, b):
        return _slot_size(b)

    def _bytesize(self):
        slot = _slot_size(self.b)
        per = 64 // slot
        nb = -(-int(self.hashvalues.size) // per)
        total = struct.calcsize(self._serial_fmt_params) + 8 * nb
        return slot, per, nb, total


def pack_matrix(signatures: np.ndarray, b: int, gpu_mode: str = "always") -> np.ndarray:
    """b-bit pack every row of an ``[N, K]`` signature matrix: returns uint64 ``[N, num_blocks]``
    whose rows are exactly the blocks ``bBitMinHash.__getstate__`` writes after its 21-byte header."""
    b = int(b)
    if b > 32 or b < 0:
        raise ValueError("b must be an integer in [0, 32]")
    signatures = np.ascontiguousarray(signatures, dtype=np.uint64)
    if gpu_mode != "disable" and (gpu_mode == "always" or _native.gpu_detected()):
        return _native.context().bbit_pack(signatures, b)
    masked = np.bitwise_and(signatures, np.uint64((1 << b) - 1))
    return _pack_rows(masked, _slot_size(b))


def unpack_matrix(blocks: np.ndarray, num_perm: int, b: int, gpu_mode: str = "always") -> np.ndarray:
    """The inverse of :func:`pack_matrix`: ``bBitMinHash.__setstate__`` (ref: datasketch/b_bit_minhash.py:103-125) of every
    row of a uint64 ``[N, num_blocks]`` block matrix -- the b-bit values as uint32 ``[N, num_perm]``, what N restored
    ``bBitMinHash`` objects would hold as ``hashvalues``."""
    b = int(b)
    if b > 32 or b < 0:
        raise ValueError("b must be an integer in [0, 32]")
    blocks = np.ascontiguousarray(blocks, dtype=np.uint64)
    slot = _slot_size(b)
    nb = -(-int(num_perm) // (64 // slot))
    if blocks.ndim != 2 or blocks.shape[1] != nb:
        raise ValueError("blocks must be [n, %d] for num_perm=%d, b=%d" % (nb, num_perm, b))
    if gpu_mode != "disable" and (gpu_mode == "always" or _native.gpu_detected()):
        return _native.context().bbit_unpack(blocks, num_perm, b)
    return _unpack_rows(blocks, slot, int(num_perm)).astype(np.uint32)


def jaccard_pairs(blocks: np.ndarray, pairs, num_perm: int, b: int, r: float = 0.0, gpu_mode: str = "always") -> np.ndarray:
    """``bBitMinHash.jaccard`` (ref: datasketch/b_bit_minhash.py:53-72) for rows ``pairs[:, 0]`` and ``pairs[:, 1]`` of a
    packed matrix (:func:`pack_matrix` output): float64 estimates ``(agreeing / num_perm - C1) / (1 - C2)`` with the
    reference's ``A(r, b)``, ``C1``, ``C2`` for a common ``r``.  On the device the agreeing positions are counted on the
    packed blocks (XOR, fold every slot to one bit, popcount); the numpy path unpacks."""
    b = int(b)
    if b > 32 or b < 0:
        raise ValueError("b must be an integer in [0, 32]")
    if r > 1.0:
        raise ValueError("r must be a float in [0.0, 1.0]")
    blocks = np.ascontiguousarray(blocks, dtype=np.uint64)
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    if pairs.size and (pairs.min() < 0 or pairs.max() >= blocks.shape[0]):
        raise ValueError("pair index out of range")
    slot = _slot_size(b)
    nb = -(-int(num_perm) // (64 // slot))
    if blocks.ndim != 2 or blocks.shape[1] != nb:
        raise ValueError("blocks must be [n, %d] for num_perm=%d, b=%d" % (nb, num_perm, b))
    if gpu_mode != "disable" and (gpu_mode == "always" or _native.gpu_detected()):
        same = _native.context().bbit_jaccard_pairs(blocks, num_perm, b, pairs)
    else:
        vals = _unpack_rows(blocks, slot, num_perm)
        same = np.count_nonzero(vals[pairs[:, 0]] == vals[pairs[:, 1]], axis=1)
    proto = object.__new__(bBitMinHash)
    a = proto._calc_a(float(r), b)
    c1, c2 = proto._calc_c(a, a, float(r), float(r))
    return (same.astype(np.float64) / float(num_perm) - c1) / (1 - c2)
