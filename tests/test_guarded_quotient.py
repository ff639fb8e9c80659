"""CPU model of `evaluate_guarded` (datasketch_amd/csrc/weighted_kernels.hip): t = floor(RN(L / r) + beta) taken without the
division.  The device computes, in float32, q' = RN(L * y) with y ~ 1/r, b1 = RN(q'(1 - 2^-21) - 2^-100),
b2 = RN(q'(1 + 2^-20) + 2^-100), t1 = floor(RN(b1 + beta)), t2 = floor(RN(b2 + beta)) and accepts t1 when t1 == t2.
The claim behind it: the reference's q = RN(L / r) (ref weighted_minhash.py:216) lies between b1 and b2, so an accepted t
is the reference's t.  Checked here with numpy's float32 arithmetic (the fused multiply-add evaluated in float64: the product
of two float32 is exact there, and the 2^-100 term only matters where q' is tiny) on random and constructed inputs, with
the table's correctly rounded reciprocal and with one perturbed by an ulp either way (the hardware reciprocal of the walk's
rounds)."""
import numpy as np

F = np.float32


def _guarded(L, r, beta, y):
    q = (L * y).astype(F)
    b1 = (q.astype(np.float64) * np.float64(F(1 - 2.0**-21)) - 2.0**-100).astype(F)
    b2 = (q.astype(np.float64) * np.float64(F(1 + 2.0**-20)) + 2.0**-100).astype(F)
    t1 = np.floor((b1 + beta).astype(F))
    t2 = np.floor((b2 + beta).astype(F))
    return t1, t1 == t2, b1, b2


def _exact(L, r, beta):
    q = (L / r).astype(F)
    return np.floor((q + beta).astype(F)), q


def _check(L, r, beta):
    L, r, beta = (np.ascontiguousarray(v, dtype=F) for v in (L, r, beta))
    want, q = _exact(L, r, beta)
    y0 = (F(1) / r).astype(F)
    settled_share = []
    for y in (y0, np.nextafter(y0, F(np.inf)), np.nextafter(y0, F(0))):
        t, settled, b1, b2 = _guarded(L, r, beta, y.astype(F))
        lo, hi = np.minimum(b1, b2), np.maximum(b1, b2)
        finite = np.isfinite(q)
        assert np.all((lo <= q)[finite] & (q <= hi)[finite]), "the reference's quotient left the bracket"
        assert np.array_equal(t[settled], want[settled]), "an accepted t differs from the reference's"
        settled_share.append(settled.mean())
    return settled_share


def test_random_logs_and_table_entries():
    rng = np.random.RandomState(0)
    n = 4_000_000
    r = rng.gamma(2.0, 1.0, n).astype(F)
    r = np.maximum(r, F(2.0**-40))
    beta = rng.uniform(0, 1, n).astype(F)
    with np.errstate(divide="ignore"):
        L = np.log(rng.uniform(0, 100, n).astype(F)).astype(F)
    L = np.where(np.isfinite(L), L, F(0)).astype(F)
    share = _check(L, r, beta)
    assert min(share) > 0.9999  # open elements are a few in 10^6 (they go through the true division)


def test_logs_on_floor_boundaries_and_extremes():
    rng = np.random.RandomState(1)
    n = 1_000_000
    r = np.exp(rng.uniform(np.log(2.0**-40), np.log(2.0**40), n)).astype(F)  # the whole range the table check admits
    beta = rng.uniform(0, 1, n).astype(F)
    beta[::7] = 0
    k = rng.randint(-60, 60, n).astype(F)
    L = ((k - beta) * r).astype(F)  # q + beta within an ulp or two of an integer
    L[1::5] = np.nextafter(L[1::5], F(np.inf))
    L[2::5] = np.nextafter(L[2::5], F(-np.inf))
    L[3::11] = 0
    L[4::13] = F(1e-42) * rng.choice([-1, 1], len(L[4::13])).astype(F)  # subnormal logs
    L[5::17] = F(2.0**80) * rng.choice([-1, 1], len(L[5::17])).astype(F)  # the largest finite logs the fast loops admit
    with np.errstate(over="ignore", invalid="ignore"):
        _check(L, r, beta)


def test_infinite_logs_are_accepted_only_with_the_reference_result():
    r = np.array([0.5, 2.0, 1e-3, 7.0], dtype=F)
    beta = np.array([0.25, 0.0, 0.9, 0.5], dtype=F)
    for L in (np.full(4, -np.inf, dtype=F), np.full(4, np.inf, dtype=F)):
        with np.errstate(invalid="ignore", over="ignore"):
            want, _ = _exact(L, r, beta)
            t, settled, _, _ = _guarded(L, r, beta, (F(1) / r).astype(F))
        assert settled.all() and np.array_equal(t, want)
