"""Round 4's host-side decisions, on the CPU (no device: the objects that would own one are stubbed).

  * parity mode of the weighted sketch takes the log on the device only where the device's float32 log reproduces THIS
    host's np.log on the start-up sentinels (ref: datasketch/weighted_minhash.py:212 takes np.log on the host);
  * the device index refuses signature matrices it would have to wrap or reinterpret (ref: datasketch/lsh.py:537-538
    keys a band by the bytes of its uint64 hashvalues: two values that differ above bit 31 are different keys).
"""
import types

import numpy as np
import pytest

from datasketch_amd import WeightedMinHashGenerator, _native
from datasketch_amd import lsh_bulk as LB
from oracle import oracle as O


class _FakeContext:
    """What Context.device_log_matches_numpy needs: a weighted_logf; `flip` corrupts one result bit."""

    def __init__(self, flip=False):
        self.flip, self.calls = flip, 0

    def weighted_logf(self, x):
        self.calls += 1
        out = O.c_np_logf(np.ascontiguousarray(x, dtype=np.float32))  # the CPU model of numpy's loop (oracle/np_logf.c)
        if self.flip:
            bits = out.view(np.uint32).copy()
            bits[bits.size // 2] ^= 1
            out = bits.view(np.float32)
        return out

    device_log_matches_numpy = _native.Context.device_log_matches_numpy


def test_start_up_check_accepts_a_log_equal_to_this_hosts_numpy_and_runs_once():
    with np.errstate(all="ignore"):
        probe = np.log(np.float32(0.3)).view(np.uint32) == O.c_np_logf(np.array([0.3], dtype=np.float32)).view(np.uint32)[0]
    ctx = _FakeContext()
    first = ctx.device_log_matches_numpy()
    assert ctx.device_log_matches_numpy() == first and ctx.calls == 1  # checked once per context
    if probe:
        # (the model equals np.log on every pattern on hosts whose numpy dispatches its AVX2 / AVX512F loop:
        # tests/test_np_logf_model.py; elsewhere the check's job is to say no, which the next test covers)
        assert first is True


def test_start_up_check_rejects_a_single_wrong_bit():
    ctx = _FakeContext(flip=True)
    assert ctx.device_log_matches_numpy() is False


@pytest.mark.parametrize("setting, device_says, want", [(None, True, True), (None, False, False), (True, False, True), (False, True, False)])
def test_parity_mode_follows_the_check_unless_told_otherwise(setting, device_says, want):
    g = WeightedMinHashGenerator(8, sample_size=4, seed=1, gpu_mode="disable", device_log=setting)
    ctx = types.SimpleNamespace(device_log_matches_numpy=lambda: device_says)
    assert g._log_on_device(ctx) is want


def _index(dtype):
    return types.SimpleNamespace(dtype=np.dtype(dtype))


def test_device_index_refuses_what_it_would_have_to_wrap():
    as_index = LB.SortedBandsIndex._as_index_dtype
    ok = as_index(_index(np.uint32), np.array([[1, 2, 0xFFFFFFFF]], dtype=np.uint64))
    assert ok.dtype == np.uint32 and ok.flags.c_contiguous and ok.tolist() == [[1, 2, 0xFFFFFFFF]]
    with pytest.raises(ValueError, match="do not fit"):
        as_index(_index(np.uint32), np.array([[1, 1 << 32]], dtype=np.uint64))
    with pytest.raises(ValueError, match="negative"):
        as_index(_index(np.uint64), np.array([[3, -1]], dtype=np.int64))
    with pytest.raises(ValueError, match="unsigned integers"):
        as_index(_index(np.uint64), np.array([[1.0, 2.0]]))
    wide = as_index(_index(np.uint64), np.array([[1, 1 << 40]], dtype=np.uint64))
    assert wide.dtype == np.uint64 and int(wide[0, 1]) == 1 << 40
    empty = as_index(_index(np.uint32), np.empty((0, 4), dtype=np.int64))
    assert empty.shape == (0, 4) and empty.dtype == np.uint32
