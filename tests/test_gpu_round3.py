"""GPU parity tests added in round 3 (run on an MI355X: python -m pytest tests -m gpu -x -q).

The weighted sketch's bound-ordered walk (dense rows and CSR rows): bit-identical (k, t) against the C oracle and
against the evaluate-every-element kernels of round 2 on config 4's own input at full size, on heavy-tailed,
sorted, sparse and degenerate inputs, across table rebuilds, and for every (dim, sample_size) shape class of the
kernels.  Everything goes through the C ABI; the oracle is the checker.
"""
import os
import zlib

import numpy as np
import pytest
import scipy.sparse as sp

from datasketch_amd import WeightedMinHashGenerator, _native
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    assert _native.gpu_available(), "these tests need an MI355X"
    return _native.context()


def _oracle(g, x, rows=None):
    xs = x if rows is None else x[rows]
    csr = sp.csr_matrix(xs)
    csr.sort_indices()
    with np.errstate(invalid="ignore", divide="ignore"):
        return O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)


def _same(got, want, wn):
    """(k, t) of the rows that store something; rows that store nothing report empty."""
    out, ne = got
    return np.array_equal(ne.astype(bool), wn) and np.array_equal(out[wn], want[wn]) and not out[~wn].any()


def _config4_input(n, dim=4096):
    rs_ = np.random.RandomState(42)
    x = np.empty((n, dim), dtype=np.float32)
    for i in range(0, n, 10_000):
        x[i : i + 10_000] = rs_.uniform(0, 100, (min(10_000, n - i), dim))
    return x


def test_walk_on_config4_input_at_full_size(ctx):
    """BASELINE.json configs[3] on its own input: 2 500 rows spread over the matrix against the C oracle, and EVERY row
    against the kernels that evaluate every element (weighted.path = 2, round 2's path) -- two implementations that
    share nothing but the arithmetic of one evaluation."""
    n, dim, s = 100_000, 4096, 128
    x = _config4_input(n, dim)
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    out, ne = g.minhash_many_arrays(x)
    assert ne.all()
    rows = np.unique(np.concatenate([np.arange(0, 32), np.linspace(0, n - 1, 2500).astype(np.int64), np.arange(n - 32, n)]))
    want, wn = _oracle(g, x, rows)
    assert wn.all() and np.array_equal(out[rows], want)
    wctx, _ = g._device_handle()
    wctx.set_option("weighted.path", 2)
    try:
        every, ne2 = g.minhash_many_arrays(x)
    finally:
        wctx.set_option("weighted.path", 0)
    assert np.array_equal(every, out) and np.array_equal(ne2, ne)


@pytest.mark.parametrize("kind", ["lognormal2", "sorted_up", "sorted_down", "constant", "spikes", "tiny_and_huge", "counts", "half_zeros", "few_percent", "one_entry"])
def test_walk_on_inputs_that_stress_the_cut_and_the_stop_rule(ctx, kind):
    """Heavy tails (entries above the cut, evaluated before the walk), sorted rows, rows of one value (every bound is
    tight: ties everywhere), rare spikes, logs near the float32 range ends, small integers, and rows that store little."""
    rng = np.random.RandomState(zlib.crc32(kind.encode()))
    n, dim, s = 96, 1024, 128
    if kind == "lognormal2":
        x = rng.lognormal(0, 2.0, (n, dim))
    elif kind in ("sorted_up", "sorted_down"):
        x = np.sort(rng.uniform(0, 100, (n, dim)), axis=1)
        x = x if kind == "sorted_up" else x[:, ::-1]
    elif kind == "constant":
        x = np.repeat(rng.uniform(0.5, 50, (n, 1)), dim, axis=1)
    elif kind == "spikes":
        x = rng.uniform(0, 1, (n, dim))
        x[rng.random_sample(x.shape) < 0.002] = 1e6
    elif kind == "tiny_and_huge":
        x = np.exp(rng.uniform(-80, 80, (n, dim)))
    elif kind == "counts":
        x = rng.poisson(3.0, (n, dim)).astype(np.float64)
    elif kind == "half_zeros":
        x = rng.uniform(0, 100, (n, dim))
        x[rng.random_sample(x.shape) < 0.5] = 0
    elif kind == "few_percent":
        x = rng.uniform(0, 100, (n, dim))
        x[rng.random_sample(x.shape) < rng.uniform(0.85, 0.995, (n, 1))] = 0
    else:
        x = np.zeros((n, dim))
        x[np.arange(n), rng.randint(0, dim, n)] = rng.uniform(0.1, 9, n)
    x = np.ascontiguousarray(x, dtype=np.float32)
    x[5] = 0  # a row that stores nothing
    g = WeightedMinHashGenerator(dim, s, seed=3, gpu_mode="always")
    want, wn = _oracle(g, x)
    assert not wn[5]
    assert _same(g.minhash_many_arrays(x), want, wn)
    # the same rows as CSR (mhx_weighted_minhash_many): short rows entry by entry, long rows walked
    assert _same(g.minhash_many_arrays(sp.csr_matrix(x)), want, wn)


@pytest.mark.parametrize("dim,s", [(1, 1), (2, 3), (5, 64), (7, 70), (63, 128), (300, 300), (513, 128), (4096, 129), (5000, 64), (16384, 65), (16385, 64)])
def test_walk_shape_classes(ctx, dim, s):
    """Dimensions that are not multiples of 4 (no 16-byte loads), above 4096 (rows not held in registers ahead), at and
    above the walk's limit of 16384 (above it the round-2 kernels take over), sample sizes that are not multiples of 64
    and above 256 (more chunks than a workgroup caches), in dense and CSR form, full and 20 % stored."""
    rng = np.random.RandomState(dim * 7 + s)
    n = 37 if dim < 5000 else 11
    g = WeightedMinHashGenerator(dim, s, seed=2, gpu_mode="always")
    for density in (1.0, 0.2):
        x = rng.uniform(0, 30, (n, dim)).astype(np.float32)
        if density < 1.0:
            x[rng.random_sample(x.shape) >= density] = 0
        want, wn = _oracle(g, x)
        assert _same(g.minhash_many_arrays(x), want, wn)
        assert _same(g.minhash_many_arrays(sp.csr_matrix(x)), want, wn)


def test_walk_rows_with_nan_and_infinite_values(ctx):
    """numpy's argmin on a row with a NaN log (a negative value): the first NaN wins; +inf values give ln_a = -inf;
    rows of such values sit between ordinary rows of the same call."""
    rng = np.random.RandomState(8)
    n, dim, s = 64, 512, 100
    x = rng.uniform(0, 10, (n, dim)).astype(np.float32)
    x[3, 100] = -1.0            # log -> NaN
    x[3, 7] = -2.0
    x[9, 5] = np.inf            # log -> +inf: ln_a = -inf
    x[9, 300] = np.inf          # a tie at -inf: the first column wins
    x[12, :] = 0
    x[12, 44] = np.nan
    x[20, rng.random_sample(dim) < 0.97] = 0
    x[20, 17] = -5.0            # a NaN in a short row
    g = WeightedMinHashGenerator(dim, s, seed=4, gpu_mode="always")
    with np.errstate(invalid="ignore", divide="ignore"):
        want, wn = _oracle(g, x)
    dense = g.minhash_many_arrays(x)
    csr = g.minhash_many_arrays(sp.csr_matrix(x))
    odd = np.array([3, 9, 12, 20])  # their t is floor(NaN) or floor(inf) cast to int64: the cast is platform-defined
    rest = np.setdiff1d(np.arange(n), odd)
    for out, ne in (dense, csr):
        assert np.array_equal(ne.astype(bool), wn)
        assert np.array_equal(out[rest], want[rest])
        assert np.array_equal(out[odd][:, :, 0], want[odd][:, :, 0])  # the winning columns
    assert np.array_equal(want[3][:, 0], np.full(s, 7)) and np.array_equal(want[9][:, 0], np.full(s, 5))
    assert np.array_equal(dense[0], csr[0])


def test_walk_tables_follow_the_scale_of_the_data(ctx):
    """The cut is planned per call from a sample of the call's logs and the sorted tables are rebuilt when it moves:
    calls at scales 1, 1e4, 1e-3 and 1 again on one generator, each against the oracle -- results never depend on
    which tables happen to be in place."""
    rng = np.random.RandomState(31)
    n, dim, s = 64, 2048, 128
    g = WeightedMinHashGenerator(dim, s, seed=9, gpu_mode="always")
    base = rng.uniform(0, 1, (n, dim)).astype(np.float32)
    for scale in (1.0, 1e4, 1e-3, 1.0, 3.0):
        x = (base * np.float32(scale)).astype(np.float32)
        want, wn = _oracle(g, x)
        assert _same(g.minhash_many_arrays(x), want, wn)
    # one call whose rows live on very different scales: most rows have all their entries above or far below the cut
    x = (base * np.float32(10.0) ** rng.randint(-6, 7, (n, 1)).astype(np.float32)).astype(np.float32)
    want, wn = _oracle(g, x)
    assert _same(g.minhash_many_arrays(x), want, wn)


def test_sparse_rows_at_scale_against_the_oracle(ctx):
    """1 %-dense CSR rows, 100 000 of them (dim 4096, 128 samples): 2 500 rows against the C oracle, every row against
    the IEEE-division-everywhere kernels (weighted.path = 1)."""
    n, dim, s = 100_000, 4096, 128
    rng = np.random.RandomState(77)
    x = sp.random(n, dim, density=0.01, format="csr", dtype=np.float32, random_state=rng, data_rvs=lambda k: rng.uniform(0.01, 100, k).astype(np.float32))
    x.sort_indices()
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    out, ne = g.minhash_many_arrays(x)
    rows = np.unique(np.linspace(0, n - 1, 2500).astype(np.int64))
    sub = x[rows]
    want, wn = O.c_weighted_minhash_many(sub.indptr, sub.indices, sub.data, g.rs, g.ln_cs, g.betas)
    assert np.array_equal(ne[rows].astype(bool), wn) and np.array_equal(out[rows][wn], want[wn])
    wctx, _ = g._device_handle()
    wctx.set_option("weighted.path", 1)
    try:
        every, ne2 = g.minhash_many_arrays(x)
    finally:
        wctx.set_option("weighted.path", 0)
    assert np.array_equal(every, out) and np.array_equal(ne2, ne)


def test_csr_rows_with_stored_zeros_and_through_the_c_abi_in_log_form(ctx):
    """Logs handed over as they are (values_are_logs): a stored zero is a log of -inf (ln_a = +inf: it wins only a row
    that stores nothing else, and then its first column does); rows long enough to be walked and short ones."""
    rng = np.random.RandomState(5)
    dim, s = 256, 64
    g = WeightedMinHashGenerator(dim, s, seed=6, gpu_mode="always")
    wctx, handle = g._device_handle()
    lens = [0, 1, 3, 40, 200, 256, 2, 100, 256, 30]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(dim, k, replace=False)) for k in lens]).astype(np.int32)
    logs = rng.uniform(-3, 5, int(indptr[-1])).astype(np.float32)
    logs[indptr[3] : indptr[4] : 3] = -np.inf     # stored zeros inside a short row
    logs[indptr[4] : indptr[5] : 2] = -np.inf     # ... inside a walked row
    logs[indptr[6] : indptr[7]] = -np.inf         # a short row of nothing but stored zeros
    logs[indptr[8] : indptr[9]] = -np.inf         # a long row of nothing but stored zeros
    want, wn = O.c_weighted_minhash_many(indptr, indices, None, g.rs, g.ln_cs, g.betas, logs=logs)
    got, ne = wctx.weighted_minhash_many(handle, s, indptr, indices, logs, True)
    zeros_only = np.array([6, 8])  # t = floor(-inf) cast to int64: platform-defined; the column is the row's first
    rest = np.setdiff1d(np.arange(len(lens)), zeros_only)
    assert np.array_equal(ne.astype(bool), wn) and np.array_equal(got[rest], want[rest])
    assert np.array_equal(got[zeros_only][:, :, 0], want[zeros_only][:, :, 0])
    assert (want[6][:, 0] == indices[indptr[6]]).all() and (want[8][:, 0] == indices[indptr[8]]).all()


def _boundary_logs(g, rng, cols, count):
    """Logs L with L / r + beta within a few ulps of an integer for some (sample, column): the quotient the entry-by-entry
    loops take without a division (evaluate_guarded) is "open" there and the true division has to decide."""
    i = rng.randint(0, g.sample_size, count)
    n = rng.randint(-40, 40, count).astype(np.float32)
    r, beta = g.rs[i, cols].astype(np.float32), g.betas[i, cols].astype(np.float32)
    base = ((n - beta) * r).astype(np.float32)
    return np.nextafter(base, np.float32(np.inf) * rng.choice([-1.0, 1.0], count).astype(np.float32)).astype(np.float32) \
        if rng.rand() < 0.5 else base


@pytest.mark.parametrize("dim,s", [(4096, 128), (300, 70), (64, 256)])
def test_entry_by_entry_rows_without_the_division_at_floor_boundaries(ctx, dim, s):
    """Sparse CSR rows (the direct kernel) and sparse dense rows (the walk kernel's entry-by-entry mode), in log form
    through the C ABI: logs on floor boundaries of some sample's quotient, zeros, subnormal and huge logs (beyond 2^80: the
    row leaves the division-free loop), +-inf, NaN rows; (k, t) bit-identical to the oracle wherever the winner's t is
    finite, the column everywhere."""
    rng = np.random.RandomState(dim + s)
    g = WeightedMinHashGenerator(dim, s, seed=9, gpu_mode="always")
    wctx, handle = g._device_handle()
    n_rows = 600
    lens = rng.randint(1, max(2, min(dim, 64)), n_rows)
    lens[:5] = [1, 2, 4, 5, 9]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(dim, k, replace=False)) for k in lens]).astype(np.int32)
    logs = rng.uniform(-6, 5, int(indptr[-1])).astype(np.float32)
    special = rng.rand(len(logs))
    pick = special < 0.5
    logs[pick] = _boundary_logs(g, rng, indices[pick], int(pick.sum()))
    logs[(special >= 0.5) & (special < 0.52)] = 0.0
    logs[(special >= 0.52) & (special < 0.53)] = -np.inf
    logs[(special >= 0.53) & (special < 0.535)] = np.float32(1e-42)    # subnormal
    logs[(special >= 0.535) & (special < 0.54)] = np.float32(-3e-39)
    for row, value in ((50, 1e30), (51, -1e30), (52, np.inf), (53, np.nan), (54, 3e38), (55, -2e25)):
        logs[indptr[row] + (lens[row] // 2)] = np.float32(value)
    want, wn = O.c_weighted_minhash_many(indptr, indices, None, g.rs, g.ln_cs, g.betas, logs=logs)
    got, ne = wctx.weighted_minhash_many(handle, s, indptr, indices, logs, True)
    finite = (want[:, :, 1] > -(2**62)) & (want[:, :, 1] < 2**62)  # (a float t beyond int64 casts platform by platform)
    assert np.array_equal(ne.astype(bool), wn)
    assert np.array_equal(got[:, :, 0], want[:, :, 0])
    assert np.array_equal(got[:, :, 1][finite], want[:, :, 1][finite])
    # the same rows as a dense matrix of logs (absent = -inf): sparse enough for the walk kernel's entry-by-entry mode
    dense = np.full((n_rows, dim), -np.inf, dtype=np.float32)
    for d in range(n_rows):
        dense[d, indices[indptr[d] : indptr[d + 1]]] = logs[indptr[d] : indptr[d + 1]]
    stored = dense != -np.inf
    lib = wctx.lib
    d_x, d_o, d_ne = wctx.to_device(dense), wctx.alloc(n_rows * s * 16), wctx.alloc(n_rows)
    try:
        _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 1, n_rows, d_o.ptr, d_ne.ptr))
        got_d = d_o.download((n_rows, s, 2), np.int64)
        ne_d = d_ne.download((n_rows,), np.uint8)
    finally:
        for d in (d_x, d_o, d_ne):
            d.free()
    # (a stored -inf is "absent" in the dense form: compare the rows that store no -inf)
    rows = np.array([d for d in range(n_rows) if stored[d].sum() == lens[d]])
    assert np.array_equal(ne_d[rows].astype(bool), wn[rows])
    assert np.array_equal(got_d[rows][:, :, 0], want[rows][:, :, 0])
    assert np.array_equal(got_d[rows][:, :, 1][finite[rows]], want[rows][:, :, 1][finite[rows]])


@pytest.mark.parametrize("dist", ["lognormal", "uniform", "pareto"])
def test_every_cut_the_plan_can_choose_gives_the_same_sketch(ctx, dist):
    """The walk's cut is chosen per call from an estimate of the cost (the share of the logs above it: 0.5 .. 8 %);
    heavy-tailed weights move it.  Whatever the cut -- each one forced in turn (option weighted.tail), tables rebuilt --
    the (k, t) pairs are the oracle's; walks that run past the positions cached in LDS (lognormal, pareto) included."""
    n, dim, s = 1500, 4096, 128
    rng = np.random.RandomState(31)
    if dist == "lognormal":
        x = rng.lognormal(0.0, 2.0, (n, dim)).astype(np.float32)
    elif dist == "pareto":
        x = (rng.pareto(1.1, (n, dim)) + 1e-3).astype(np.float32)
    else:
        x = rng.uniform(0, 100, (n, dim)).astype(np.float32)
    g = WeightedMinHashGenerator(dim, s, seed=3, gpu_mode="always")
    wctx, _ = g._device_handle()
    want, wn = _oracle(g, x, np.arange(0, n, 5))
    try:
        for tail in (0, 1, 2, 3, 4, 5, 0):
            wctx.set_option("weighted.tail", tail)
            out, ne = g.minhash_many_arrays(x)
            assert ne.all() and np.array_equal(out[::5], want), (dist, tail)
            if tail == 0:
                first = out
            assert np.array_equal(out, first), (dist, tail)
    finally:
        wctx.set_option("weighted.tail", 0)


def test_device_log_mode_through_the_walk(ctx):
    """device_log=True takes logf on the device inside the walk kernel's staging pass: its (k, t) may differ from parity
    mode only under BASELINE.md section 3's rule (bench.weighted_gap_gate)."""
    import sys

    sys.path.insert(0, ROOT)
    from bench import weighted_gap_gate

    n, dim, s = 4096, 4096, 128
    x = _config4_input(n, dim)
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")
    gl = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always", device_log=True)
    hv, _ = g.minhash_many_arrays(x)
    hv_l, _ = gl.minhash_many_arrays(x)
    mism = np.argwhere(np.any(hv != hv_l, axis=2))
    assert len(mism) < 1e-5 * n * s
    assert weighted_gap_gate(x, g, hv, hv_l, mism)["unexplained"] == 0


# ------------------------------------------------------------------ bench.py at N = 8 (one GPU shared: plumbing only)
def _bench(argv, env=None, timeout=900):
    import subprocess
    import sys

    e = dict(os.environ)
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MHX_RDZV_ADDR", "MHX_RDZV_NONCE", "TORCHELASTIC_RUN_ID"):
        e.pop(var, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _keys(obj, prefix=""):
    if isinstance(obj, dict):
        out = set()
        for k, v in obj.items():
            out |= {prefix + k} | _keys(v, prefix + k + ".")
        return out
    return set()


def test_bench_with_eight_ranks_on_one_gpu():
    """The driver's SCALE run is the first launch with eight ranks anybody makes.  Here eight ranks share this box's one
    GPU (--share-devices: the numbers mean nothing): eight HIP runtimes start at once, the star rendezvous forms at
    world 8, every rank generates its shard in bounded pieces, the MAX over ranks is taken, rank 0 alone prints.  With
    the all-gather probe on, RCCL refuses eight ranks on one device; that must come back as an error in the line, from
    every rank, before the watchdog -- not as a hang."""
    import json

    p = _bench(["--gpus", "8", "--steps", "3", "--warmup", "1", "--sets", "20000", "--check-rows", "256", "--share-devices", "--no-allgather-probe"])
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # only rank 0 prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    assert len(line["per_rank"]["ms_per_step"]) == 8 and len(line["per_rank"]["kernel_ms"]) == 8 and len(line["per_rank"]["devices"]) == 8
    assert "cpu_baseline" not in line and "extra" not in line
    p = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--sets", "10000", "--check-rows", "64", "--share-devices", "--probe-timeout", "120"])
    assert p.returncode == 0, p.stdout + p.stderr
    line = json.loads([ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and "allgather" in line
    ag = line["allgather"]
    assert "error" in ag or ag.get("rccl_ranks_seen") == [8] * 8  # one device: refused by RCCL (or, if it ever works, complete)
    if "error" in ag:
        assert "did not finish" not in str(ag["error"])  # the refusal came back from RCCL, the watchdog never fired


def test_bench_line_is_the_same_with_and_without_a_launcher_environment_at_one_gpu():
    """N = 1 under `python bench.py` and under a launcher that exports RANK=0 / WORLD_SIZE=1: the same keys at every
    level of the line, and the same number within the spread of two short runs."""
    import json

    args = ["--sets", "200000", "--steps", "10", "--warmup", "2", "--cpu-sample", "0", "--no-e2e", "--no-extra", "--check-rows", "512"]
    plain = _bench(args)
    assert plain.returncode == 0, plain.stdout + plain.stderr
    launched = _bench(args, env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "LOCAL_WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511"})
    assert launched.returncode == 0, launched.stdout + launched.stderr
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    b = json.loads(launched.stdout.strip().splitlines()[-1])
    assert _keys(a) == _keys(b)
    assert abs(a["value"] - b["value"]) <= 0.05 * a["value"]
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["metric"] == b["metric"]


# ------------------------------------------------------------------ MinHash: several permutations per lane in kernel C
@pytest.mark.parametrize("k", [33, 40, 48, 49, 64, 65, 96, 97, 128])
def test_packed_kernel_with_several_permutations_per_lane(ctx, k):
    """Kernel C with P permutations per lane (a token read from the LDS tile serves all P): 33 .. 48 permutations run
    there by default (16 lanes x 3, four sets per wave), 49 .. 128 when minhash.packed = 2 asks for it.  Against the C
    oracle on ragged sets (empty, shorter than a row, several 256-token blocks, wide tokens, repeated tokens), with an
    initial state, uint32 tokens and output -- under every setting of the option."""
    rng = np.random.RandomState(300 + k)
    n = 2051
    lens = rng.randint(0, 600, size=n)
    lens[:8] = [0, 1, 15, 16, 17, 255, 256, 257]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    hv = rng.randint(0, 2**32, int(offsets[-1]), dtype=np.uint64)
    wide = rng.random_sample(hv.size) < 0.01
    hv[wide] = rng.randint(0, 2**64, int(wide.sum()), dtype=np.uint64)
    hv[offsets[20] : offsets[20] + 5] = hv[offsets[20]]  # repeated tokens: proofs fail, the set goes to the dedup launch
    a, b = O.np_init_permutations(k, 3)
    init = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    init[3, 0] = 2**40
    want = O.c_minhash_bulk(hv, offsets, a, b)
    want_init = O.c_minhash_bulk(hv, offsets, a, b, init)
    narrow = hv & np.uint64(0xFFFFFFFF)
    want32 = O.c_minhash_bulk(narrow, offsets, a, b)
    dense = rng.randint(0, 2**32, (1001, 48), dtype=np.uint64)
    want_dense = O.c_minhash_bulk_dense(dense, a, b)
    for packed in (0, 1, 2):
        ctx.set_option("minhash.packed", packed)
        ctx.set_option("minhash.split", 1)
        try:
            assert np.array_equal(ctx.minhash_bulk((a, b), hv, offsets, 0, n), want), packed
            assert np.array_equal(ctx.minhash_bulk((a, b), hv, offsets, 0, n, init), want_init), packed
            got32 = ctx.minhash_bulk((a, b), narrow.astype(np.uint32), offsets, 0, n, out_dtype=np.uint32)
            assert np.array_equal(got32.astype(np.uint64), want32), packed
            assert np.array_equal(ctx.minhash_bulk((a, b), dense.reshape(-1), None, 48, 1001), want_dense), packed
        finally:
            ctx.set_option("minhash.packed", 0)
            ctx.set_option("minhash.split", 0)


# ------------------------------------------------------------------ LSH: the bands bucketed in two passes
@pytest.mark.parametrize("n,b,r,dtype", [(1, 4, 2, np.uint64), (100, 32, 8, np.uint32), (2999, 20, 5, np.uint64), (3001, 20, 5, np.uint32),
                                         (50_000, 32, 8, np.uint32), (300_000, 16, 4, np.uint64), (70_000, 128, 2, np.uint32), (40_000, 1, 6, np.uint64)])
def test_bands_bucketed_in_two_passes_match_the_host_sort(ctx, n, b, r, dtype):
    """mhx_lsh_sort_bands' default path (a scatter into bins by the digest's top bits, every bin finished in LDS) against
    the stable host sort of (digest, row) per band and against the radix-sort path (lsh.sort = 1), for sizes on both
    sides of every bin-count step, one band and 128, uint32 and uint64 signatures, with buckets of equal keys."""
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(n % 1000 + b)
    sig = rng.randint(0, 2**32, (n, b * r + 3), dtype=np.uint64).astype(dtype)
    if n > 10:
        sig[rng.randint(0, n, n // 5)] = sig[rng.randint(0, n, n // 5)]          # buckets of equal rows
        sig[rng.randint(0, n, n // 8), :r] = sig[rng.randint(0, n, n // 8), :r]  # ... and of single bands
    want_dig, want_rows = LB.sorted_bands(sig, b, r, gpu_mode="disable")
    got_dig, got_rows = LB.sorted_bands(sig, b, r, gpu_mode="always")
    assert np.array_equal(got_dig, want_dig) and np.array_equal(got_rows, want_rows)
    ctx.set_option("lsh.sort", 1)
    try:
        radix_dig, radix_rows = LB.sorted_bands(sig, b, r, gpu_mode="always")
    finally:
        ctx.set_option("lsh.sort", 0)
    assert np.array_equal(radix_dig, want_dig) and np.array_equal(radix_rows, want_rows)
    assert np.array_equal(LB.candidate_pairs(sig, b, r, gpu_mode="always"), LB.candidate_pairs(sig, b, r, gpu_mode="disable"))


def test_bands_bucketed_with_clustered_and_identical_signatures(ctx):
    """Thousands of copies of a few rows (large buckets inside a bin: the bin is finished by the bitonic sort), then tens
    of thousands of identical rows (a bin overflows: the call falls back to the radix sort) -- same answer as the host."""
    from datasketch_amd import lsh_bulk as LB

    rng = np.random.RandomState(77)
    n, b, r = 60_000, 8, 4
    sig = rng.randint(0, 2**32, (n, b * r), dtype=np.uint64)
    for c in range(6):  # six clusters of 1500 copies: buckets of 1500 equal digests inside bins of ~3000
        sig[rng.permutation(n)[:1500]] = sig[c]
    want_dig, want_rows = LB.sorted_bands(sig, b, r, gpu_mode="disable")
    got_dig, got_rows = LB.sorted_bands(sig, b, r, gpu_mode="always")
    assert np.array_equal(got_dig, want_dig) and np.array_equal(got_rows, want_rows)
    sig[rng.permutation(n)[:25_000]] = sig[0]  # one bucket of 25 000 per band: more than a bin holds
    want_dig, want_rows = LB.sorted_bands(sig, b, r, gpu_mode="disable")
    got_dig, got_rows = LB.sorted_bands(sig, b, r, gpu_mode="always")
    assert np.array_equal(got_dig, want_dig) and np.array_equal(got_rows, want_rows)


# (kept last in the file: added after the round's last GPU call, it first runs on the round-end box)
@pytest.mark.parametrize("dim,s", [(300, 70), (200, 64), (64, 256), (1024, 128)])
def test_sparse_and_dense_rows_interleaved_with_more_rows_than_workgroups(ctx, dim, s):
    """Rows evaluated entry by entry (a few per cent stored; their list shared between the waves of a chunk, results
    meeting in the row's LDS) between rows that are walked from the tables a workgroup caches in LDS, 6 000 rows so that
    every workgroup takes several of both kinds: every row against the oracle.  (Small dim: the row's LDS is smaller than
    what the waves' results take -- such rows must not be shared out.)"""
    rng = np.random.RandomState(dim * 7 + s)
    n = 6000
    x = np.zeros((n, dim), dtype=np.float32)
    density = rng.choice([0.02, 0.05, 0.09, 0.3, 0.6, 1.0], n)
    mask = rng.random_sample((n, dim)) < density[:, None]
    x[mask] = rng.uniform(0.01, 100, int(mask.sum())).astype(np.float32)
    g = WeightedMinHashGenerator(dim, s, seed=11, gpu_mode="always")
    want, wn = _oracle(g, x)
    got = g.minhash_many_arrays(x)
    assert _same(got, want, wn)
    wctx, _ = g._device_handle()
    wctx.set_option("weighted.split", 1)
    try:
        assert _same(g.minhash_many_arrays(x), want, wn)
    finally:
        wctx.set_option("weighted.split", 0)
