"""datasketch_amd -- MI355X-native MinHash signature engine with the datasketch API.

Drop-in for the bulk-hashing hot path of ekzhu/datasketch (``MinHash.update_batch`` / ``bulk`` /
``generator`` and ``WeightedMinHashGenerator.minhash_many``): same class names, arguments and
results, with the permutation + min kernels written in HIP for gfx950 and reached through a
ctypes C ABI (``include/mhx.h``).  No PyTorch / CuPy / Triton on the path.
"""
from datasketch_amd.b_bit_minhash import bBitMinHash
from datasketch_amd import lsh_bulk
from datasketch_amd.hashfunc import prehashed, sha1_hash32, sha1_hash64, sha1_hash_many
from datasketch_amd.lean_minhash import LeanMinHash
from datasketch_amd.minhash import MinHash
from datasketch_amd.weighted_minhash import WeightedMinHash, WeightedMinHashGenerator

__version__ = "0.1.0"

__all__ = [
    "LeanMinHash",
    "MinHash",
    "WeightedMinHash",
    "WeightedMinHashGenerator",
    "bBitMinHash",
    "prehashed",
    "sha1_hash32",
    "sha1_hash64",
    "sha1_hash_many",
    "lsh_bulk",
]
