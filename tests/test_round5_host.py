"""Round 5's host-side decisions, on the CPU.

* `gpu_mode='detect'` keeps the reference's semantics (datasketch/minhash.py:272-279: use the device if there is one,
  else the numpy path, never an error) -- with one RuntimeWarning when the host has an AMD GPU but libmhx does not load;
  `gpu_mode='always'` stays strict (ref :272-275).
"""
import warnings

import numpy as np
import pytest

from datasketch_amd import MinHash, WeightedMinHashGenerator, _native, lsh_bulk
from datasketch_amd.b_bit_minhash import pack_matrix


@pytest.fixture
def broken_library_on_a_gpu_host(monkeypatch):
    """A host with /dev/kfd whose libmhx.so cannot be loaded."""

    def boom():
        raise _native.MhxError("libmhx.so could not be loaded from /nowhere/libmhx.so: cannot open shared object file")

    monkeypatch.setattr(_native, "device_count", boom)
    monkeypatch.setattr(_native, "gpu_node_present", lambda: True)
    monkeypatch.setattr(_native, "_detect_warned", False)


def test_detect_falls_back_to_numpy_with_one_warning(broken_library_on_a_gpu_host):
    tokens = [f"t{i}".encode() for i in range(50)]
    ref = MinHash(num_perm=32, seed=5, gpu_mode="disable")
    ref.update_batch(tokens)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        m = MinHash(num_perm=32, seed=5, gpu_mode="detect")
        m.update_batch(tokens)                                              # ref minhash.py:276-277: no error
        sigs = MinHash.bulk_signatures(np.arange(60, dtype=np.uint64).reshape(6, 10), num_perm=16, seed=2, hashfunc=lambda x: x, gpu_mode="detect")
        blocks = pack_matrix(sigs, 1, gpu_mode="detect")
        dig = lsh_bulk.band_digests(sigs, 4, 4, gpu_mode="detect")
        g = WeightedMinHashGenerator(8, 4, seed=1, gpu_mode="detect")
        out = g.minhash_many([[1, 0, 3, 0, .5, 2, 0, 7], [0] * 8])
    assert np.array_equal(m.hashvalues, ref.hashvalues)
    assert sigs.shape == (6, 16) and blocks.shape == (6, 1) and dig.shape == (6, 4)
    assert out[1] is None and out[0].hashvalues.tolist() == [[0, 0], [7, 3], [5, 0], [2, 1]]   # SURVEY.md 8c golden
    mine = [w for w in seen if issubclass(w.category, RuntimeWarning) and "datasketch_amd" in str(w.message)]
    assert len(mine) == 1 and "cannot open shared object file" in str(mine[0].message)  # announced once, names the load error


def test_always_stays_strict_on_the_same_host(broken_library_on_a_gpu_host):
    m = MinHash(num_perm=8, seed=1, gpu_mode="always")
    with pytest.raises(RuntimeError, match="GPU mode 'always' requested"):     # ref minhash.py:272-275
        m.update_batch([b"a"])
    with pytest.raises((RuntimeError, _native.MhxError)):
        lsh_bulk.band_digests(np.zeros((2, 8), dtype=np.uint64), 2, 4, gpu_mode="always")
    with pytest.raises(_native.MhxError):
        _native.gpu_available()                                                 # the strict probe still raises here


def test_detect_is_silent_on_a_host_without_a_gpu(monkeypatch):
    def boom():
        raise _native.MhxError("no library")

    monkeypatch.setattr(_native, "device_count", boom)
    monkeypatch.setattr(_native, "gpu_node_present", lambda: False)
    monkeypatch.setattr(_native, "_detect_warned", False)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        assert _native.gpu_detected() is False and _native.gpu_available() is False
    assert not seen
