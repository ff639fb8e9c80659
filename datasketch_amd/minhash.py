"""MinHash with the reference's API (ekzhu/datasketch ``datasketch.MinHash``) and a HIP back end.

Mirror of datasketch/minhash.py: same constructor, attributes, methods, validation and error
behaviour, so an object of this class can be handed to MinHashLSH & co. (they read
``.hashvalues``, ``.seed`` and ``len()`` only).  What differs is below the ``gpu_mode`` seam:

* ``gpu_mode='always'`` / ``'detect'`` run the permutation + min step on an MI355X through
  libmhx (``datasketch_amd/csrc``), replacing the reference's CuPy branch
  (datasketch/minhash.py:281-291).  ``'always'`` raises ``RuntimeError`` when no device (or no
  libmhx.so) is usable; ``'detect'`` uses the device when there is one.
* ``bulk`` / ``generator`` hand the WHOLE corpus to one fused kernel instead of one call per
  set, and ``bulk_signatures`` returns the ``[N, K]`` matrix without building N objects.
* ``gpu_mode='disable'`` is the reference's numpy arithmetic (kept because it is part of the
  API contract; it is not a fallback: nothing switches to it implicitly on a GPU host).

Results are bit-identical to the reference in every mode, including numpy's uint64 wrap-around
in ``(a*hv + b) % (2**61-1) & 0xFFFFFFFF``.
"""
from __future__ import annotations

import copy
import warnings
from typing import Callable, Generator, Iterable, List, Optional, Tuple

import numpy as np

from datasketch_amd import _native
from datasketch_amd.hashfunc import prehashed, sha1_hash32, sha1_hash64

# The size of a hash value in number of bytes (reference: datasketch/minhash.py:27)
hashvalue_byte_size = len(bytes(np.int64(42).data))

_mersenne_prime = np.uint64((1 << 61) - 1)  # reference: datasketch/minhash.py:30
_max_hash = np.uint64((1 << 32) - 1)  # :31
_hash_range = 1 << 32  # :32

_GPU_MODES = ("disable", "detect", "always")

# sets per device launch in bulk/generator (bounds host staging memory, keeps generator lazy)
_BULK_CHUNK_SETS = 1 << 16
_BULK_CHUNK_TOKENS = 1 << 25


def _gpu_available() -> bool:
    """Reference: datasketch/minhash.py:38-48, with a HIP device count instead of CuPy's."""
    return _native.gpu_available()


def _no_device_error() -> RuntimeError:
    # message kept compatible with datasketch/minhash.py:274
    return RuntimeError("GPU mode 'always' requested but no HIP device (or libmhx.so) is available.")


def _as_hash_array(values) -> np.ndarray:
    """Same conversion the reference applies (minhash.py:294): raises OverflowError for values
    outside uint64, TypeError/ValueError for non-integers."""
    return np.array(values, dtype=np.uint64)


class MinHash:
    """MinHash sketch for Jaccard similarity; drop-in for ``datasketch.MinHash``.

    Args:
        num_perm: number of permutation functions (ignored when ``hashvalues`` is given).
        seed: seed of the permutation functions.
        gpu_mode: ``'disable'`` | ``'detect'`` | ``'always'`` (see module docstring).
        hashfunc: callable mapping a token to an unsigned integer hash value (< 2**64).
        hashobj: deprecated, as in the reference.
        hashvalues: optional initial state.
        permutations: optional ``(a, b)`` permutation parameters to reuse.
    """

    # The reference keeps its CuPy copies of (a, b) in these two attributes (datasketch/minhash.py:
    # 156-165) and nulls them when pickling; here device state lives in the process-wide libmhx
    # context, never in the object, so they exist for compatibility and stay None.
    _a_gpu = None
    _b_gpu = None

    def __init__(
        self,
        num_perm: int = 128,
        seed: int = 1,
        gpu_mode: str = "disable",
        hashfunc: Callable = sha1_hash32,
        hashobj: Optional[object] = None,
        hashvalues=None,
        permutations=None,
    ) -> None:
        if hashvalues is not None:
            num_perm = len(hashvalues)
        if num_perm > _hash_range:
            # reference: datasketch/minhash.py:125-132
            raise ValueError("Cannot have more than %d number of permutation functions" % _hash_range)
        self.seed = seed
        self.num_perm = num_perm
        if not callable(hashfunc):
            raise ValueError("The hashfunc must be a callable.")
        self.hashfunc = hashfunc
        if hashobj is not None:
            warnings.warn("hashobj is deprecated, use hashfunc instead.", DeprecationWarning, stacklevel=2)
        if hashvalues is not None:
            self.hashvalues = self._parse_hashvalues(hashvalues)
        else:
            self.hashvalues = self._init_hashvalues(num_perm)
        if permutations is not None:
            self.permutations = permutations
        else:
            self.permutations = self._init_permutations(num_perm)
        if len(self) != len(self.permutations[0]):
            raise ValueError("Numbers of hash values and permutations mismatch")
        self._gpu_mode = gpu_mode

    # ------------------------------------------------------------------ state helpers
    def _init_hashvalues(self, num_perm: int) -> np.ndarray:
        return np.full(num_perm, _max_hash, dtype=np.uint64)

    def _init_permutations(self, num_perm: int) -> np.ndarray:
        """(a_i, b_i) drawn alternately from the legacy ``RandomState(seed)`` stream, a in [1, p),
        b in [0, p); same stream as datasketch/minhash.py:170-184 (kept on the host)."""
        gen = np.random.RandomState(self.seed)
        ab = np.empty((2, num_perm), dtype=np.uint64)
        for i in range(num_perm):
            ab[0, i] = gen.randint(1, _mersenne_prime, dtype=np.uint64)
            ab[1, i] = gen.randint(0, _mersenne_prime, dtype=np.uint64)
        return ab

    def _parse_hashvalues(self, hashvalues) -> np.ndarray:
        return np.array(hashvalues, dtype=np.uint64)

    # ------------------------------------------------------------------ back-end selection
    def _use_gpu(self) -> bool:
        """The seam of datasketch/minhash.py:268-279."""
        mode = self._gpu_mode
        if mode == "always":
            try:
                ok = _gpu_available()
            except _native.MhxError as e:
                raise _no_device_error() from e
            if not ok:
                raise _no_device_error()
            return True
        if mode == "detect":  # the reference's fallback (minhash.py:276-277); a GPU host with a broken library is announced once
            return _native.gpu_detected()
        return False

    # ------------------------------------------------------------------ updates
    def update(self, b) -> None:
        """Add one token.  As in the reference (minhash.py:221-224) a single token is folded in
        on the host: one K-vector of numpy arithmetic, no device round trip."""
        hv = self.hashfunc(b)
        a, c = self.permutations
        phv = np.bitwise_and((a * hv + c) % _mersenne_prime, _max_hash)
        self.hashvalues = np.minimum(phv, self.hashvalues)

    def update_batch(self, b: Iterable) -> None:
        """Add many tokens.  Hashing runs on the host (reference contract, minhash.py:262-263);
        permutation + min run on the device for ``gpu_mode`` 'always' / 'detect'."""
        if self.hashfunc is prehashed and isinstance(b, np.ndarray):
            hv_list = b
            if b.size == 0:
                return
        elif self.hashfunc in (sha1_hash32, sha1_hash64) and self._use_gpu():
            # the reference's default hash, for the whole batch at once on the device
            b = b if isinstance(b, (list, tuple)) else list(b)
            if not b:
                return
            buf, offs = _native.Context.pack_tokens(b)
            if self.hashfunc is sha1_hash32:  # one call: bytes up, SHA-1 + permutations + min on the device, K values back
                one_set = np.array([0, len(b)], dtype=np.int64)
                self.hashvalues = _native.context().minhash_bulk_bytes(self.permutations, buf, offs, one_set, self.hashvalues)[0]
                return
            hv_list = _native.context().sha1_tokens(buf, offs, 64)
        else:
            hv_list = [self.hashfunc(_b) for _b in b]
            if not hv_list:  # empty batch is a no-op (minhash.py:265-266)
                return
        if self._use_gpu():
            hv = _as_hash_array(hv_list).reshape(-1)
            self.hashvalues = _native.context().minhash_update_batch(self.permutations, hv, self.hashvalues)
            return
        a, c = self.permutations
        hv = np.array(hv_list, dtype=np.uint64, ndmin=2).T
        phv = np.bitwise_and((hv * a + c) % _mersenne_prime, _max_hash)
        self.hashvalues = np.minimum(self.hashvalues, phv.min(axis=0))

    # ------------------------------------------------------------------ estimators
    def jaccard(self, other: "MinHash") -> float:
        if other.seed != self.seed:
            raise ValueError("Cannot compute Jaccard given MinHash with different seeds")
        if len(self) != len(other):
            raise ValueError("Cannot compute Jaccard given MinHash with different numbers of permutation functions")
        return float(np.count_nonzero(self.hashvalues == other.hashvalues)) / float(len(self))

    def count(self) -> float:
        k = len(self)
        return float(k) / np.sum(self.hashvalues / float(_max_hash)) - 1.0

    def merge(self, other: "MinHash") -> None:
        if other.seed != self.seed:
            raise ValueError("Cannot merge MinHash with different seeds")
        if len(self) != len(other):
            raise ValueError("Cannot merge MinHash with different numbers of permutation functions")
        self.hashvalues = np.minimum(other.hashvalues, self.hashvalues)

    def digest(self) -> np.ndarray:
        return copy.copy(self.hashvalues)

    def is_empty(self) -> bool:
        return not np.any(self.hashvalues != _max_hash)

    def clear(self) -> None:
        self.hashvalues = self._init_hashvalues(len(self))

    def copy(self) -> "MinHash":
        return MinHash(
            seed=self.seed,
            hashfunc=self.hashfunc,
            hashvalues=self.digest(),
            permutations=self.permutations,
            gpu_mode=self._gpu_mode,
        )

    def __len__(self) -> int:
        return len(self.hashvalues)

    def __eq__(self, other) -> bool:
        return type(self) is type(other) and self.seed == other.seed and np.array_equal(self.hashvalues, other.hashvalues)

    __hash__ = None  # mutable, like the reference (defines __eq__ without __hash__)

    @classmethod
    def union(cls, *mhs: "MinHash") -> "MinHash":
        if len(mhs) < 2:
            raise ValueError("Cannot union less than 2 MinHash")
        num_perm = len(mhs[0])
        seed = mhs[0].seed
        if any((seed != m.seed or num_perm != len(m)) for m in mhs):
            raise ValueError("The unioning MinHash must have the same seed and number of permutation functions")
        hashvalues = np.minimum.reduce([m.hashvalues for m in mhs])
        return cls(
            num_perm=num_perm,
            seed=seed,
            hashfunc=mhs[0].hashfunc,
            hashvalues=hashvalues,
            permutations=mhs[0].permutations,
            gpu_mode=mhs[0]._gpu_mode,
        )

    # ------------------------------------------------------------------ bulk
    def _spawn(self, hashvalues: np.ndarray) -> "MinHash":
        """A sibling sharing seed / hashfunc / permutations / gpu_mode, without re-running
        ``__init__`` (what ``copy()`` produces in datasketch/minhash.py:385-393, ~10x cheaper)."""
        m = object.__new__(type(self))
        m.seed = self.seed
        m.num_perm = self.num_perm
        m.hashfunc = self.hashfunc
        m.hashvalues = hashvalues
        m.permutations = self.permutations
        m._gpu_mode = self._gpu_mode
        return m

    def _hash_sets(self, sets: List) -> Tuple[np.ndarray, np.ndarray]:
        """Apply ``hashfunc`` per token on the host and pack the result as CSR (values, offsets)."""
        f = self.hashfunc
        if f is prehashed:
            packed = _native.Context.pack_int_sets(sets)  # lists of Python ints: the C helper, ~10x np.array per set
            if packed is not None:
                return packed
        offsets = np.zeros(len(sets) + 1, dtype=np.int64)
        if f is prehashed:
            parts = [_as_hash_array(s).reshape(-1) if not isinstance(s, np.ndarray) or s.dtype != np.uint64 else s.reshape(-1) for s in sets]
        else:
            parts = [_as_hash_array([f(t) for t in s]).reshape(-1) for s in sets]
        for i, p in enumerate(parts):
            offsets[i + 1] = offsets[i] + p.size
        hv = np.concatenate(parts) if parts else np.empty(0, dtype=np.uint64)
        return hv.astype(np.uint64, copy=False), offsets

    def _bulk_chunks(self, b: Iterable, out_dtype=np.uint64) -> Generator[np.ndarray, None, None]:
        """Yield ``[n_i, K]`` signature blocks for consecutive chunks of the corpus ``b``."""
        init = None if self.is_empty() else self.hashvalues
        if self.hashfunc is prehashed and isinstance(b, np.ndarray) and b.ndim == 2:
            # dense corpus of fixed-length sets: no per-set Python work at all; a uint32 array (the range of
            # sha1_hash32) travels to the device as it is -- half the bytes over PCIe
            if b.dtype == np.uint32 and self._use_gpu():
                tok = np.ascontiguousarray(b)
            else:
                tok = _as_hash_array(b) if b.dtype != np.uint64 else np.ascontiguousarray(b)
            n, t = tok.shape
            step = max(1, min(n, _BULK_CHUNK_TOKENS // max(t, 1)))
            for s in range(0, n, step):
                blk = tok[s : s + step]
                yield self._signatures_csr(blk.reshape(-1), None, t, blk.shape[0], init, out_dtype)
            return
        if self.hashfunc is prehashed and isinstance(b, tuple) and len(b) == 2:
            values, offsets = b
            values = _as_hash_array(values).reshape(-1)
            offsets = np.asarray(offsets, dtype=np.int64)
            n = offsets.size - 1
            s = 0
            while s < n:
                e = min(n, s + _BULK_CHUNK_SETS)
                local = offsets[s : e + 1] - offsets[s]
                yield self._signatures_csr(values[offsets[s] : offsets[e]], local, 0, e - s, init, out_dtype)
                s = e
            return
        # the reference's two default token hashes run on the device (32: hashfunc.py:5-15, 64: :17-28)
        device_sha1 = 32 if self.hashfunc is sha1_hash32 else 64 if self.hashfunc is sha1_hash64 else 0
        if device_sha1 and not self._use_gpu():
            device_sha1 = 0
        chunk: List = []
        tokens = 0
        for s in b:
            s = s if hasattr(s, "__len__") else list(s)
            chunk.append(s)
            tokens += len(s)
            if len(chunk) >= _BULK_CHUNK_SETS or tokens >= _BULK_CHUNK_TOKENS:
                yield self._signatures_of_sets(chunk, init, device_sha1).astype(out_dtype, copy=False)
                chunk, tokens = [], 0
        if chunk:
            yield self._signatures_of_sets(chunk, init, device_sha1).astype(out_dtype, copy=False)

    def _signatures_of_sets(self, sets: List, init, device_sha1: int) -> np.ndarray:
        if not device_sha1:
            hv, offsets = self._hash_sets(sets)
            return self._signatures_csr(hv, offsets, 0, len(sets), init)
        # default hashfunc + device: pack the byte tokens once, SHA-1 and MinHash both on the device
        # (repeated tokens are left in: dropping them here costs more host time per set -- a dict per set --
        # than the kernel's slow path for such sets costs on the device)
        buf, byte_offsets, set_offsets = _native.Context.pack_sets(sets)
        return _native.context().minhash_bulk_bytes(self.permutations, buf, byte_offsets, set_offsets, init, bits=device_sha1)

    def _signatures_csr(self, hv, offsets, fixed_len, n_sets, init, out_dtype=np.uint64) -> np.ndarray:
        if self._use_gpu():
            return _native.context().minhash_bulk(self.permutations, hv, offsets, fixed_len, n_sets, init, out_dtype=out_dtype)
        # gpu_mode='disable': the reference's per-set numpy arithmetic (minhash.py:293-297)
        a, c = self.permutations
        k = len(a)
        out = np.empty((n_sets, k), dtype=np.uint64)
        proto = self._init_hashvalues(k) if init is None else np.asarray(init, dtype=np.uint64)
        for i in range(n_sets):
            beg, end = (offsets[i], offsets[i + 1]) if offsets is not None else (i * fixed_len, (i + 1) * fixed_len)
            if end == beg:
                out[i] = proto
                continue
            col = hv[beg:end].reshape(-1, 1)
            phv = np.bitwise_and((col * a + c) % _mersenne_prime, _max_hash)
            out[i] = np.minimum(proto, phv.min(axis=0))
        return out.astype(out_dtype, copy=False)

    @classmethod
    def bulk(cls, b: Iterable, **minhash_kwargs) -> List["MinHash"]:
        """Compute one MinHash per element of ``b`` (reference: datasketch/minhash.py:464-489)."""
        return list(cls.generator(b, **minhash_kwargs))

    @classmethod
    def generator(cls, b: Iterable, **minhash_kwargs) -> Generator["MinHash", None, None]:
        """Lazily yield one MinHash per element of ``b`` (reference: datasketch/minhash.py:491-522).

        With a device back end the corpus is consumed in chunks of up to 65 536 sets; each chunk
        is one fused kernel launch.
        """
        m = cls(**minhash_kwargs)
        for block in m._bulk_chunks(b):
            for row in block:
                yield m._spawn(row.copy())

    @classmethod
    def bulk_signatures(cls, b=None, out_dtype=np.uint64, packed=None, **minhash_kwargs) -> np.ndarray:
        """Not in the reference: the ``[N, K]`` signature matrix of a corpus, without creating N
        Python objects.  ``b`` is any iterable of token iterables, or -- with ``hashfunc=prehashed``
        -- a 2-D integer array (fixed-length sets; a uint32 array is uploaded as it is) or a
        ``(values, offsets)`` CSR pair of already hashed tokens.  ``out_dtype``: uint64 (the
        reference's ``hashvalues`` type) or uint32 (values are < 2**32: half the bytes back).

        ``packed=(buf, byte_offsets, set_offsets)`` instead of ``b``: byte tokens already packed back to back --
        token ``i`` is ``buf[byte_offsets[i]:byte_offsets[i+1]]``, set ``j`` owns tokens ``set_offsets[j] ..
        set_offsets[j+1]`` (what a tokenizer writing into one buffer produces; the layout of
        ``mhx_minhash_bulk_bytes``).  No per-object packing on the host at all: with the default ``hashfunc``
        (``sha1_hash32``, or ``sha1_hash64``) SHA-1 and MinHash both run on the device; any other ``hashfunc``, or
        ``gpu_mode='disable'``, is applied per token on the host (ref: minhash.py:262-263)."""
        if np.dtype(out_dtype) not in (np.dtype(np.uint64), np.dtype(np.uint32)):
            raise ValueError("out_dtype must be uint64 or uint32")
        if (b is None) == (packed is None):
            raise ValueError("give the corpus either as b or as packed=(buf, byte_offsets, set_offsets)")
        m = cls(**minhash_kwargs)
        if np.dtype(out_dtype) == np.uint32 and not m.is_empty() and int(m.hashvalues.max()) > 0xFFFFFFFF:
            raise ValueError("initial hashvalues >= 2**32 do not fit uint32 signatures")
        blocks = list(m._packed_chunks(packed, np.dtype(out_dtype)) if packed is not None else m._bulk_chunks(b, np.dtype(out_dtype)))
        if not blocks:
            return np.empty((0, len(m)), dtype=out_dtype)
        return blocks[0] if len(blocks) == 1 else np.concatenate(blocks, axis=0)

    def _packed_chunks(self, packed, out_dtype=np.uint64) -> Generator[np.ndarray, None, None]:
        """Signature blocks of a corpus of packed byte tokens, in chunks of whole sets (see :meth:`bulk_signatures`)."""
        if not (isinstance(packed, tuple) and len(packed) == 3):
            raise ValueError("packed is a (buf, byte_offsets, set_offsets) triple")
        buf = np.frombuffer(memoryview(packed[0]), dtype=np.uint8) if not isinstance(packed[0], np.ndarray) else np.ascontiguousarray(packed[0]).view(np.uint8).reshape(-1)
        byte_offsets = np.ascontiguousarray(packed[1], dtype=np.int64).reshape(-1)
        set_offsets = np.ascontiguousarray(packed[2], dtype=np.int64).reshape(-1)
        if byte_offsets.size < 1 or set_offsets.size < 1:
            raise ValueError("byte_offsets has tokens + 1 entries and set_offsets sets + 1")
        n_tokens, n_sets = byte_offsets.size - 1, set_offsets.size - 1
        if byte_offsets[0] != 0 or byte_offsets[-1] > buf.size or np.any(np.diff(byte_offsets) < 0):
            raise ValueError("byte_offsets must start at 0, never decrease and end inside buf")
        if set_offsets[0] != 0 or set_offsets[-1] != n_tokens or np.any(np.diff(set_offsets) < 0):
            raise ValueError("set_offsets must start at 0, never decrease and end at the number of tokens")
        init = None if self.is_empty() else self.hashvalues
        bits = 32 if self.hashfunc is sha1_hash32 else 64 if self.hashfunc is sha1_hash64 else 0
        on_device = bits and self._use_gpu()
        s = 0
        while s < n_sets:
            # whole sets, at most _BULK_CHUNK_SETS of them and about _BULK_CHUNK_TOKENS tokens (one set may exceed that alone)
            e = min(n_sets, s + _BULK_CHUNK_SETS)
            t0 = int(set_offsets[s])
            if int(set_offsets[e]) - t0 > _BULK_CHUNK_TOKENS:
                e = max(s + 1, int(np.searchsorted(set_offsets, t0 + _BULK_CHUNK_TOKENS, side="right")) - 1)
            t1 = int(set_offsets[e])
            b0 = int(byte_offsets[t0])
            local_bytes = byte_offsets[t0: t1 + 1] - b0
            local_sets = set_offsets[s: e + 1] - t0
            piece = buf[b0: int(byte_offsets[t1])]
            if on_device:
                blk = _native.context().minhash_bulk_bytes(self.permutations, piece, local_bytes, local_sets, init, bits=bits)
            else:
                f = self.hashfunc
                raw = piece.tobytes()
                hv = _as_hash_array([f(raw[local_bytes[i]: local_bytes[i + 1]]) for i in range(t1 - t0)]).reshape(-1) if t1 > t0 else np.empty(0, dtype=np.uint64)
                blk = self._signatures_csr(hv, local_sets, 0, e - s, init)
            yield blk.astype(out_dtype, copy=False)
            s = e

    # ------------------------------------------------------------------ pickling
    # State is plain numpy + the gpu_mode string: device handles live in the process-wide
    # context (datasketch_amd._native), never in the object, so pickles are portable
    # (the reference drops its CuPy arrays for the same reason, minhash.py:524-538).
    def __getstate__(self):
        return self.__dict__.copy()

    def __setstate__(self, state):
        state = dict(state)
        state.pop("_a_gpu", None)  # tolerate pickles of the reference class layout
        state.pop("_b_gpu", None)
        self.__dict__.update(state)
