"""oracle/np_logf.c -- the CPU restatement of numpy's float32 SIMD logarithm -- against np.log of the installed numpy.

The weighted path's only dependency on a third-party algorithm: ref datasketch/weighted_minhash.py:212 takes np.log of
float32 data, and numpy's float32 log (AVX2 / AVX512F loop) is neither correctly rounded nor libm's.  The device function
np_logf() (datasketch_amd/csrc/weighted_kernels.hip) restates the same operations; this test pins the restatement
itself on the CPU: a stride through all 2^31 non-negative bit patterns, every pattern near the algorithm's boundaries
(mantissa around 1/sqrt(2) in every binade, powers of two, the denormal range, the ends), negatives and NaNs.
oracle/check_np_logf.py runs ALL 2^32 patterns (46 s here; result in profiles/r04_np_logf_exhaustive.txt).
"""
import numpy as np
import pytest

from oracle import oracle as O


def _equal_bits(x):
    with np.errstate(all="ignore"):
        want = np.log(x)
    return np.count_nonzero(O.c_np_logf(x).view(np.uint32) != want.view(np.uint32))


@pytest.fixture(scope="module")
def simd_log():
    probe = np.random.RandomState(1).randint(0, 0x7F800000, size=1 << 16).astype(np.uint32).view(np.float32)
    if _equal_bits(probe):
        pytest.skip("this host's numpy does not run the AVX2 / AVX512F float32 log loop (the model restates that loop)")


def test_stride_through_all_non_negative_patterns(simd_log):
    bad = 0
    for start in range(0, 1 << 31, 1 << 28):
        bad += _equal_bits(np.arange(start, start + (1 << 28), 127, dtype=np.uint32).view(np.float32))
    assert bad == 0


def test_boundaries_specials_and_negatives(simd_log):
    near = np.arange(-96, 97, dtype=np.int64)
    sqrt_half = int(np.float32(0.70710678).view(np.uint32)) & 0x007FFFFF
    pats = [np.arange(0, 1 << 16), np.arange((1 << 23) - 4096, (1 << 23) + 4096), np.arange(0x7F800000 - 4096, 0x7F800000 + 4096),
            np.arange(0, 1 << 23, 5)]
    for e in range(0, 255):
        pats.append((e << 23) + near)                      # around every power of two
        pats.append((e << 23) + sqrt_half + near)          # around m = 1/sqrt(2) in every binade
    bits = np.unique(np.clip(np.concatenate(pats), 0, 0x7FFFFFFF)).astype(np.uint32)
    assert _equal_bits(bits.view(np.float32)) == 0
    for d in range(1, 23):  # denormals whose mantissa, normalised, sits at the 1/sqrt(2) boundary
        b = ((0x00800000 | sqrt_half) >> d) + near
        assert _equal_bits(np.clip(b, 0, None).astype(np.uint32).view(np.float32)) == 0
    rng = np.random.RandomState(2)
    neg = (rng.randint(0, 1 << 31, size=1 << 20).astype(np.uint32) | np.uint32(0x80000000)).view(np.float32)
    assert _equal_bits(neg) == 0
    nans = np.concatenate([np.arange(0x7F800001, 0x7F800001 + 4096), np.arange(0x7FC00000, 0x7FC00000 + 4096),
                           np.arange(0xFF800000, 0xFF800000 + 4096), np.arange(0xFFC00000, 0xFFC00000 + 4096)]).astype(np.uint32)
    assert _equal_bits(nans.view(np.float32)) == 0


def test_value_threshold_below_the_cut_is_conservative(simd_log):
    """The one-wave-per-row kernel, values in: an entry counts as 'above the cut' only after its log has been taken, and the log
    is taken only of values above vcut = exp(lcut - 1e-5 max(1, |lcut|)) (0 for cuts <= -87).  That is safe iff v <= vcut implies np.log(v) <= lcut --
    checked here on the floats around vcut for cuts over the whole range, with vcut moved a few ulps either way (the device's
    expf is not numpy's)."""
    rng = np.random.RandomState(5)
    cuts = np.concatenate([rng.uniform(-100, 88, 4000), rng.uniform(-2, 6, 4000), [0.0, 1e-3, -1e-3, 4.6051702, 87.9]]).astype(np.float32)
    with np.errstate(all="ignore"):
        vcut = np.exp((cuts - np.float32(1e-5) * np.maximum(np.float32(1), np.abs(cuts))).astype(np.float32)).astype(np.float32)
    keep = cuts > -87.0  # (at and below -87 the kernel takes vcut = 0: denormal values, where a rounded exp bounds nothing)
    cuts, vcut = cuts[keep], vcut[keep]
    bits = vcut.view(np.uint32).astype(np.int64)
    for shift in (-4, 0, 4):
        top = np.clip(bits + shift, 1, 0x7F7FFFFF)
        near = (top[:, None] - np.arange(0, 256)[None, :]).clip(1, None).astype(np.uint32).view(np.float32)  # the 256 floats at and below vcut
        with np.errstate(all="ignore"):
            logs = np.log(near)
        assert (logs <= cuts[:, None]).all(), shift
