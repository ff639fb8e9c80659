"""Token hash functions.

The reference hashes tokens in Python on the CPU and hands integers to the permutation step
(datasketch/hashfunc.py:5-28, datasketch/minhash.py:262-263); these callables are that boundary,
unchanged.  The bulk paths recognise the function objects ``sha1_hash32`` / ``sha1_hash64`` (the
reference's default) and, with a device back end, hash whole chunks of byte tokens with the SHA-1
kernel of libmhx instead of calling them per token (``sha1_hash_many``); results are identical.
"""
import hashlib
import struct

_U32 = struct.Struct("<I")
_U64 = struct.Struct("<Q")


def sha1_hash32(data):
    """First 4 bytes of SHA1(data) as a little-endian unsigned 32-bit integer
    (same function as the reference's ``datasketch.hashfunc.sha1_hash32``)."""
    return _U32.unpack_from(hashlib.sha1(data).digest())[0]


def sha1_hash64(data):
    """First 8 bytes of SHA1(data) as a little-endian unsigned 64-bit integer
    (same function as the reference's ``datasketch.hashfunc.sha1_hash64``)."""
    return _U64.unpack_from(hashlib.sha1(data).digest())[0]


def prehashed(value):
    """Identity hash function for tokens that already are hash values (non-negative integers < 2**64).

    Semantically the same as the ``fake_hash_func`` idiom of the reference's tests
    (test/utils.py:4-6).  The bulk entry points recognise this exact function object and skip the
    per-token Python call: a numpy integer array (or a ``(values, offsets)`` CSR pair) is then
    handed to the device as is.
    """
    return value


def sha1_hash_many(tokens, bits: int = 32, gpu_mode: str = "detect"):
    """``[sha1_hash32(t) for t in tokens]`` (``bits=64``: ``sha1_hash64``) as a numpy array; on the
    device when one is available (``gpu_mode`` as for MinHash), else with hashlib."""
    import numpy as np

    from datasketch_amd import _native

    if bits not in (32, 64):
        raise ValueError("bits must be 32 or 64")
    tokens = tokens if isinstance(tokens, (list, tuple)) else list(tokens)
    use_gpu = gpu_mode == "always" or (gpu_mode == "detect" and _native.gpu_detected())
    if gpu_mode == "always" and not _native.gpu_available():
        raise RuntimeError("GPU mode 'always' requested but no MI355X / libmhx.so is available.")
    if not use_gpu:
        f = sha1_hash32 if bits == 32 else sha1_hash64
        return np.array([f(t) for t in tokens], dtype=np.uint32 if bits == 32 else np.uint64)
    buf, offsets = _native.Context.pack_tokens(tokens)
    return _native.context().sha1_tokens(buf, offsets, bits)
