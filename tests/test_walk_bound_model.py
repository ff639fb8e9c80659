"""CPU model of the weighted path's bound-ordered walk (datasketch_amd/csrc/weighted_kernels.hip).

The kernel's claim: for one table entry (r > 0, finite ln_c and beta) the float32 value the reference computes,
    ln_a(L) = ln_c - (floor(L / r + beta) - beta + 1) * r            (ref weighted_minhash.py:216-218, one rounding per step)
is a non-increasing function of the log L.  Hence LB = ln_a(Lcut) bounds ln_a(L) from below for every L <= Lcut, and a walk
over a sample's columns in the order of LB that stops at the first LB **larger** than the smallest ln_a seen (ties go to
the smaller column) returns numpy's argmin over all stored columns.  Both are checked here in numpy's float32 arithmetic:
the monotonicity on random and adversarial entries, and a line-by-line model of the walk (entries above the cut first,
then the list, strict stop rule) against the brute-force argmin of the reference's formula."""
import numpy as np

F = np.float32


def ln_a(L, r, ln_c, beta):
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        q = (L / r).astype(F)
        t = np.floor((q + beta).astype(F))
        u = (t - beta).astype(F)
        v = (u + F(1)).astype(F)
        return (ln_c - (v * r).astype(F)).astype(F), t


def test_ln_a_is_monotone_in_the_log():
    rng = np.random.RandomState(0)
    m = 20000
    r = np.concatenate([rng.gamma(2.0, 1.0, m // 2), np.exp(rng.uniform(-27, 27, m // 2))]).astype(F)
    r = np.maximum(r, F(2.0**-40))
    ln_c = np.log(rng.gamma(2.0, 1.0, m)).astype(F)
    beta = rng.uniform(0, 1, m).astype(F)
    beta[::9] = 0
    # per entry: 64 logs in increasing order -- random, neighbouring floats, floor boundaries of the entry's own quotient, extremes
    L = np.sort(np.concatenate([
        rng.uniform(-20, 12, (m, 24)),
        ((rng.randint(-50, 50, (m, 16)) - beta[:, None]) * r[:, None]),
        np.exp(rng.uniform(-80, 80, (m, 8))) * rng.choice([-1, 1], (m, 8)),
        np.zeros((m, 4)),
    ], axis=1).astype(F), axis=1)
    L = np.concatenate([L, np.nextafter(L[:, ::4], F(np.inf)), np.nextafter(L[:, ::4], F(-np.inf)),
                        np.full((m, 1), -np.inf, F), np.full((m, 1), np.inf, F)], axis=1)
    L = np.sort(L, axis=1)
    a, _ = ln_a(L, r[:, None], ln_c[:, None], beta[:, None])
    assert not np.isnan(a).any()
    assert np.all(a[:, 1:] <= a[:, :-1]), "a larger log gave a larger ln_a"


def _walk(logs, r, ln_c, beta, lcut):
    """One sample's walk over a row: logs[dim] (-inf = not stored); returns (column, t) or (None, None) for an empty row."""
    dim = len(logs)
    lb, _ = ln_a(np.full(dim, lcut, F), r, ln_c, beta)
    order = np.lexsort((np.arange(dim), lb))  # by bound, then column (the build kernel sorts (ordered LB, column))
    best, best_c, best_t = F(np.inf), None, None

    def offer(c):
        nonlocal best, best_c, best_t
        a, t = ln_a(logs[c:c + 1], r[c:c + 1], ln_c[c:c + 1], beta[c:c + 1])
        a, t = a[0], t[0]
        if a < best or (a == best and (best_c is None or c < best_c)):
            best, best_c, best_t = a, c, t

    for c in np.nonzero(logs > lcut)[0]:  # entries above the cut have no valid bound: evaluated first
        offer(int(c))
    visited = 0
    for c in order:
        if lb[c] > best:  # strictly larger: an equal bound may still hide a tie at a smaller column
            break
        visited += 1
        if logs[c] != -np.inf:
            offer(int(c))
    return best_c, best_t, visited


def test_the_walk_returns_the_argmin_of_every_row():
    rng = np.random.RandomState(1)
    dim, samples = 96, 12
    total_visits = 0
    for trial in range(60):
        r = rng.gamma(2.0, 1.0, (samples, dim)).astype(F)
        ln_c = np.log(rng.gamma(2.0, 1.0, (samples, dim))).astype(F)
        beta = rng.uniform(0, 1, (samples, dim)).astype(F)
        kind = trial % 6
        if kind == 0:
            x = rng.uniform(0, 100, dim)
        elif kind == 1:
            x = rng.lognormal(0, 2, dim)
        elif kind == 2:
            x = np.full(dim, rng.uniform(0.5, 50))  # one value everywhere: bounds tight, ties likely
        elif kind == 3:
            x = rng.poisson(2.0, dim).astype(float)  # zeros (not stored) and small integers: many equal logs
        elif kind == 4:
            x = np.where(rng.random_sample(dim) < 0.9, 0.0, rng.uniform(0, 100, dim))
        else:
            x = np.exp(rng.uniform(-60, 60, dim))
        with np.errstate(divide="ignore"):
            logs = np.log(x.astype(F)).astype(F)
        stored = np.nonzero(logs != -np.inf)[0]
        if len(stored) == 0:
            continue
        for lcut in (np.quantile(logs[stored], 0.9), logs[stored].max(), logs[stored].min(), F(0.0)):
            lcut = F(lcut)
            for i in range(samples):
                a, t = ln_a(logs[stored], r[i, stored], ln_c[i, stored], beta[i, stored])
                j = int(np.argmin(a))  # numpy: the first minimum = the smallest column among ties
                c, tt, visited = _walk(logs, r[i], ln_c[i], beta[i], lcut)
                assert c == stored[j] and tt == t[j], (trial, float(lcut), i)
                total_visits += visited
    assert total_visits > 0
