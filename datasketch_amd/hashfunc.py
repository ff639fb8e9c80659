"""Token hash functions (host side).

The reference hashes tokens in Python on the CPU and hands integers to the permutation step
(datasketch/hashfunc.py:5-28, datasketch/minhash.py:262-263); this boundary is kept.
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
