"""GPU parity tests added in round 5 (run on an MI355X: python -m pytest tests -m gpu -x -q).

* the N > 1 path END TO END on one GPU: 2 and 8 ranks share the device, the shards travel over the explicit host-staged
  all-gather transport (RCCL refuses two ranks on one device), equal and unequal shards, and the gathered matrix goes
  through pack / digests / bucketing on every rank -- against the single-process results and the oracle
  (reference for the stages: minhash.py:293-297, b_bit_minhash.py:78-101, lsh.py:326-347,537-543);
* configs 3 and 5 at their stated size on one GPU: 10M x 256 tokens, num_perm = 256 (2.56e9 tokens: the first pass above
  2^31 elements through every kernel), band digests, the bucketing of 320M keys, b = 1 packing;
* the fused pack + band-digest kernel against the two oracles, every slot size, uint32 and uint64;
* the rewritten bucketing passes against numpy's stable order, round 3's work order and the radix sort.
Everything goes through the C ABI.
"""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from datasketch_amd import _native, lsh_bulk
from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def ctx():
    assert _native.gpu_available(), "these tests need an MI355X"
    return _native.context()


# ------------------------------------------------------------------ fused b-bit blocks + band digests
_FUSED = [(256, 32, 8), (128, 16, 8), (64, 16, 4), (256, 16, 16), (512, 64, 8), (128, 32, 4), (64, 4, 16)]


@pytest.mark.parametrize("k,bands,r", _FUSED)
@pytest.mark.parametrize("dtype", [np.uint32, np.uint64])
def test_fused_pack_and_digests_against_both_oracles(ctx, k, bands, r, dtype):
    """mhx_bbit_pack_band_digests_dev: blocks == the C oracle's bBitMinHash packing (b_bit_minhash.py:82-101), digests ==
    FNV-1a-64 of the reference's band key bytes (lsh.py:537-538), for every b -- from one read of the matrix where the
    shape allows (reported through *fused), and equal to the two separate kernels (pack.fused = 1) bit for bit."""
    rng = np.random.RandomState(k + bands)
    n = 5003
    sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
    if dtype == np.uint64:
        wide = rng.random_sample(sig.shape) < 0.02  # hashvalues given by hand may exceed 2^32: the key's leading bytes are then not zero
        sig[wide] = rng.randint(0, 2**63, int(wide.sum()), dtype=np.uint64)
    sig = sig.astype(dtype)
    want_dig = lsh_bulk.band_digests(sig.astype(np.uint64), bands, r, gpu_mode="disable")
    for i in range(3):  # the numpy digests themselves against the byte-wise definition
        keys = O.c_band_keys(sig[i: i + 1].astype(np.uint64), bands, r)
        assert int(want_dig[i, 0]) == lsh_bulk.fnv1a_64(keys[0, :r].tobytes())
    for b in (1, 2, 3, 4, 7, 8, 12, 16, 20, 32):
        slot = 1 if b == 1 else 2 if b == 2 else 4 if b <= 4 else 8 if b <= 8 else 16 if b <= 16 else 32
        g = max(1, (64 // slot) // r)
        blocks, dig, fused = ctx.bbit_pack_band_digests(sig, b, bands, r)
        assert fused == (bands % g == 0), (b, fused)
        assert np.array_equal(blocks, O.c_bbit_pack(sig.astype(np.uint64), b)), b
        assert np.array_equal(dig, want_dig), b
        blocks_bm, dig_bm, fused_bm = ctx.bbit_pack_band_digests(sig, b, bands, r, layout=_native.BAND_MAJOR)   # digests [bands, n]
        assert fused_bm == fused and np.array_equal(blocks_bm, blocks) and np.array_equal(dig_bm, want_dig.T), b
        ctx.set_option("pack.fused", 1)
        try:
            blocks2, dig2, fused2 = ctx.bbit_pack_band_digests(sig, b, bands, r)
        finally:
            ctx.set_option("pack.fused", 0)
        assert not fused2 and np.array_equal(blocks2, blocks) and np.array_equal(dig2, dig), b


@pytest.mark.parametrize("k,bands,r", [(100, 10, 10), (256, 32, 4), (96, 12, 8), (256, 25, 10), (24, 3, 8), (256, 128, 2)])
def test_fused_entry_point_on_shapes_the_fused_kernel_does_not_take(ctx, k, bands, r):
    """bands * r < num_perm, bands not a power of two, r not 4 / 8 / 16: the same entry point runs the two kernels."""
    rng = np.random.RandomState(k)
    sig = rng.randint(0, 2**32, (1001, k), dtype=np.uint64)
    for b in (1, 5, 32):
        blocks, dig, fused = ctx.bbit_pack_band_digests(sig, b, bands, r)
        assert not fused
        assert np.array_equal(blocks, O.c_bbit_pack(sig, b))
        assert np.array_equal(dig, lsh_bulk.band_digests(sig, bands, r, gpu_mode="disable"))
        assert np.array_equal(ctx.bbit_pack_band_digests(sig, b, bands, r, layout=_native.BAND_MAJOR)[1], dig.T)  # band_digest_kernel, band-major
    blocks, dig, fused = ctx.bbit_pack_band_digests(sig[:0], 1, bands, r)
    assert blocks.shape[0] == 0 and dig.shape == (0, bands)


# ------------------------------------------------------------------ bucketing: the round-5 passes
@pytest.mark.parametrize("n", [1, 63, 2500, 2501, 10_001, 70_001, 400_000, 3_000_000])
def test_bucketing_passes_in_both_layouts_equal_numpy_and_the_radix_sort(ctx, n):
    """mhx_lsh_sort_digests_dev ([n, bands] input) and mhx_lsh_sort_digests_layout_dev with MHX_BAND_MAJOR ([bands, n]: the layout
    the chain runs on, read with unit stride): the (band, digest, row) order must be numpy's stable order and the stable
    radix sort's (lsh.sort = 1, with and without the digests riding through the sort) -- on uniform digests, on clusters of
    equal digests inside a bin's capacity and beyond it (the fallback), with rows shared by many bands; one scatter level and
    two (lsh.levels = 2 forces what more than 2^10 bins per band -- 2.56M rows -- select by themselves: the 3M-row case)."""
    rng = np.random.RandomState(n)
    bands = 16 if n <= 400_000 else 8
    dig = rng.randint(0, 2**63, (n, bands), dtype=np.uint64) * np.uint64(2) + rng.randint(0, 2, (n, bands)).astype(np.uint64)
    if n > 100:
        dig[rng.randint(0, n, min(n // 7, 600)), 3] = dig[0, 3]  # one big bucket in band 3 (within a bin's capacity: the two passes run)
        dig[rng.randint(0, n, n // 7), 7] = dig[1, 7]            # ... and one in band 7 that overflows its bin for large n: the radix fallback
        dup = rng.randint(0, n, n // 3)
        dig[dup, 5] = dig[(dup * 7) % n, 5]                       # many small ones in band 5
    d_dig, d_dig_bm = ctx.to_device(dig), ctx.to_device(np.ascontiguousarray(dig.T))
    d_sd, d_sr = ctx.alloc(max(1, n * bands * 8)), ctx.alloc(max(1, n * bands * 4))
    res = {}
    for name, opts in (("lds", {}), ("two_levels", {"lsh.levels": 2}), ("radix", {"lsh.sort": 1}), ("band_major_two_levels", {"lsh.levels": 2}),
                       ("band_major", {}), ("band_major_radix", {"lsh.sort": 1}), ("band_major_gather", {"lsh.sort": 1, "lsh.gather": 1}),
                       # round 6: bins of up to 11 264 elements finished by the big form of the bin pass (auto between 2.56M and 10.2M rows: the 3M-row
                       # case takes it by itself, lsh.bigbins = 1 is the three passes there; 2 forces it from 4 bins on), teams of 256 / 1024 threads
                       ("big_bins", {"lsh.bigbins": 2}), ("band_major_big_bins", {"lsh.bigbins": 2}), ("band_major_never_big", {"lsh.bigbins": 1}),
                       ("band_major_teams_256", {"lsh.team": 256}), ("band_major_teams_1024", {"lsh.team": 1024}),
                       ("band_major_teams_1024_two_levels", {"lsh.team": 1024, "lsh.levels": 2})):
        for key, v in opts.items():
            ctx.set_option(key, v)
        try:
            if name.startswith("band_major"):  # the same digests handed over [bands, n]
                _native.check(ctx.lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig_bm.ptr, n, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr))
            else:
                _native.check(ctx.lib.mhx_lsh_sort_digests_dev(ctx.handle, d_dig.ptr, n, bands, d_sd.ptr, d_sr.ptr))
            ctx.synchronize()
            res[name] = (d_sd.download((bands, n), np.uint64), d_sr.download((bands, n), np.uint32))
        finally:
            for key in opts:
                ctx.set_option(key, 0)
    for j in range(bands):
        order = np.lexsort((np.arange(n), dig[:, j]))
        assert np.array_equal(res["lds"][1][j], order.astype(np.uint32)), j
        assert np.array_equal(res["lds"][0][j], dig[order, j]), j
    for name in res:
        assert np.array_equal(res[name][0], res["lds"][0]) and np.array_equal(res[name][1], res["lds"][1]), name


# ------------------------------------------------------------------ RCCL: unequal shards in place (one rank: what a 1-GPU box can run)
def test_allgatherv_with_one_rank_through_rccl(ctx):
    """mhx_comm_allgatherv_dev (grouped ncclBroadcast per root) with a communicator of one rank: the shard lands at its
    offset and nothing else is touched.  N > 1 over RCCL needs one GPU per rank (the driver's 8-GPU run)."""
    from datasketch_amd import dist, rendezvous

    g = rendezvous.Group(0, 1)
    try:
        comm = dist.communicator(ctx, g)
    except _native.MhxError as e:
        pytest.skip(f"no RCCL communicator on this box: {e}")
    shard = np.random.RandomState(1).randint(0, 2**32, (1000, 64), dtype=np.uint64).astype(np.uint32)
    d_local = ctx.to_device(shard)
    d_all = ctx.to_device(np.full(shard.size + 64, 0xABABABAB, dtype=np.uint32))
    comm.allgatherv_dev(d_local.ptr, d_all.ptr + 128, [0], [shard.nbytes])
    ctx.synchronize()
    got = d_all.download((shard.size + 64,), np.uint32)
    assert np.array_equal(got[32: 32 + shard.size], shard.reshape(-1))
    assert np.all(got[:32] == 0xABABABAB) and np.all(got[32 + shard.size:] == 0xABABABAB)
    out = dist.allgather_signatures_dev(ctx, d_local, 1000, 64, [1000], g)
    assert out.transport == "rccl" and np.array_equal(out.to_host(np.uint32), shard)
    g.close()


# ------------------------------------------------------------------ N > 1 on one GPU: the whole config-3 chain per rank
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, n, t, k, bands, r, scheme, tmp_path, transport="host", timeout=600):
    out = str(tmp_path / f"c3_{world}_{scheme}")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = {key: v for key, v in os.environ.items() if key not in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MHX_RDZV_ADDR=f"127.0.0.1:{port}",
                   MHX_RDZV_NONCE="round5", MHX_TEST_TRANSPORT=transport)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_c3.py"), str(n), str(t), str(k), str(bands), str(r), scheme, out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    deadline = time.time() + timeout
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=max(1.0, deadline - time.time()))[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append("TIMEOUT " + p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    recs = [json.load(open(f"{out}.{rank}.json")) for rank in range(world)]
    return recs, np.load(f"{out}.0.npz")


@pytest.mark.parametrize("world,scheme", [(2, "equal"), (2, "unequal"), (8, "equal"), (8, "unequal")])
def test_sharded_chain_with_ranks_sharing_one_gpu(ctx, tmp_path, world, scheme):
    """`world` processes, one GPU: every rank hashes its shard, the uint32 shards are all-gathered over the host-staged
    transport (explicitly chosen; /dev/shm on one node), and EVERY rank runs pack + digests + bucketing on the gathered
    [N, K] matrix.  All ranks must hold the same bytes at every stage, and those must be the single-process results:
    signatures == the C oracle's, blocks == its b=1 packing, digests == FNV-1a of the reference's key bytes, sorted
    bands == the stable (digest, row) order."""
    import rank_c3

    n, t, k, bands, r = 40_000, 96, 128, 16, 8
    recs, arrays = _run_ranks(world, n, t, k, bands, r, scheme, tmp_path)
    counts = rank_c3.split(n, world, scheme)
    assert [rec["counts"] for rec in recs] == [counts] * world
    if scheme == "unequal":
        assert counts[0] == 1 and len(set(counts)) >= min(world, 3)
    assert all(rec["transport"] == "host-shm" and rec["fused"] for rec in recs)
    assert all(rec["sha"] == recs[0]["sha"] for rec in recs)          # every rank: the same bytes at every stage
    tokens = rank_c3.corpus(n, t)
    a, b = O.np_init_permutations(k, 3)
    want = O.c_minhash_bulk_dense(tokens, a, b)
    assert np.array_equal(arrays["sig"].astype(np.uint64), want)
    assert np.array_equal(arrays["blocks"], O.c_bbit_pack(want, 1))
    dig = lsh_bulk.band_digests(want, bands, r, gpu_mode="disable")
    assert np.array_equal(arrays["digests"], dig)
    for j in range(bands):
        order = np.lexsort((np.arange(n), dig[:, j]))
        assert np.array_equal(arrays["sorted_rows"][j], order.astype(np.uint32)) and np.array_equal(arrays["sorted_digests"][j], dig[order, j])
    # ... and the single-process device path gives the same matrix
    from datasketch_amd import MinHash, prehashed

    assert np.array_equal(MinHash.bulk_signatures(tokens, num_perm=k, seed=3, hashfunc=prehashed, gpu_mode="always"), want)


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    """tests/fake_rccl.c built next to the test run: RCCL's entry points for ranks that share one device (see its header)."""
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "c", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        os.path.join(ROOT, "tests", "fake_rccl.c"), "-o", out, "-L/opt/rocm/lib", "-lamdhip64"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return out


@pytest.mark.parametrize("world,scheme", [(2, "equal"), (8, "unequal")])
def test_rccl_binding_with_world_above_one_through_the_stand_in_library(ctx, tmp_path, fake_rccl, monkeypatch, world, scheme):
    """libmhx's RCCL binding (csrc/comm.hip) with world > 1 on one GPU: MHX_RCCL_LIBRARY points the dlopen at the stand-in, so
    ncclCommInitRank / ncclAllGather (equal shards) and the grouped per-root ncclBroadcasts of mhx_comm_allgatherv_dev
    (unequal shards, each written at its final offset) run for real from `world` processes.  What this pins is our marshalling
    (unique-id exchange over the rendezvous, byte counts, offsets, group bracketing); RCCL itself is not in the picture."""
    import rank_c3

    monkeypatch.setenv("MHX_RCCL_LIBRARY", fake_rccl)
    n, t, k, bands, r = 24_000, 64, 128, 16, 8
    recs, arrays = _run_ranks(world, n, t, k, bands, r, scheme, tmp_path, transport="rccl")
    assert all(rec["transport"] == "rccl" for rec in recs) and all(rec["sha"] == recs[0]["sha"] for rec in recs)
    a, b = O.np_init_permutations(k, 3)
    want = O.c_minhash_bulk_dense(rank_c3.corpus(n, t), a, b)
    assert np.array_equal(arrays["sig"].astype(np.uint64), want)
    assert np.array_equal(arrays["blocks"], O.c_bbit_pack(want, 1))


def test_an_override_library_that_does_not_load_is_an_error(ctx):
    """MHX_RCCL_LIBRARY naming a file that is not there does not fall back to the installed librccl.so."""
    body = "import sys; sys.path.insert(0, %r); from datasketch_amd import _native\n" \
           "try:\n    _native.Communicator.unique_id(); print('LOADED')\nexcept Exception as e:\n    print('REFUSED', e)" % ROOT
    p = subprocess.run([sys.executable, "-c", body], env=dict(os.environ, MHX_RCCL_LIBRARY="/nonexistent/librccl.so"), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "REFUSED" in p.stdout and "LOADED" not in p.stdout, p.stdout + p.stderr


def test_host_transport_over_sockets_with_two_ranks(ctx, tmp_path, monkeypatch):
    """The same chain when the ranks do not share /dev/shm files (forced here): the shards travel through the rendezvous
    sockets in bounded pieces."""
    import rank_c3

    env_body = "import sys; sys.path.insert(0, %r); from datasketch_amd import dist; dist._FORCE_TCP = True; dist._HOST_PIECE = 1 << 20; " \
               "sys.argv = sys.argv[1:]; import runpy; runpy.run_path(sys.argv[0], run_name='__main__')" % ROOT
    n, t, k, bands, r = 20_000, 64, 128, 16, 8
    out = str(tmp_path / "tcp")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MHX_RDZV_ADDR=f"127.0.0.1:{port}", MHX_TEST_TRANSPORT="host")
        procs.append(subprocess.Popen([sys.executable, "-c", env_body, os.path.join(ROOT, "tests", "rank_c3.py"), str(n), str(t), str(k), str(bands), str(r),
                                       "unequal", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    recs = [json.load(open(f"{out}.{rank}.json")) for rank in range(2)]
    assert all(rec["transport"] == "host-tcp" for rec in recs) and recs[0]["sha"] == recs[1]["sha"]
    a, b = O.np_init_permutations(k, 3)
    assert np.array_equal(np.load(f"{out}.0.npz")["sig"].astype(np.uint64), O.c_minhash_bulk_dense(rank_c3.corpus(n, t), a, b))


# ------------------------------------------------------------------ bench.py at N > 1 with the host-staged transport
def _bench(argv, env=None, timeout=900):
    e = dict(os.environ)
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MHX_RDZV_ADDR", "MHX_RDZV_NONCE", "TORCHELASTIC_RUN_ID",
                "MHX_ALLGATHER_TRANSPORT"):
        e.pop(var, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _line(p):
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # only rank 0 prints
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 8])
def test_bench_ranks_share_one_gpu_and_run_config_3_end_to_end(world):
    """`bench.py --gpus N --share-devices --allgather-transport host`: the driver's N > 1 command line with the one
    substitution a 1-GPU box needs.  The all-gather probe and extra.c3_sharded (shard -> K=256 signatures -> gather ->
    band-partitioned digests + bucketing, parity-gated on every rank) run to the end; the transport is named in the line."""
    line = _line(_bench(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--sets", "20000", "--c3-rows", "30000", "--check-rows", "256",
                         "--share-devices", "--allgather-transport", "host", "--clock-warmup", "0"]))
    assert line["n_gpus"] == world and line["config"]["allgather_transport"] == "host"
    ag = line["allgather"]
    assert ag["transport"] == "host-shm" and ag["rccl_ranks_seen"] is None and len(ag["ms_per_rank"]) == world
    c3 = line["extra"]["c3_sharded"]
    assert "error" not in c3, c3
    assert c3["rows_total"] == world * 30000 and sum(c3["bands_per_rank"]) == 32
    assert c3["allgather"]["transport"] == "host-shm" and c3["allgather"]["bytes_received_per_gpu"] == (world - 1) * 30000 * 256 * 4
    for stage in ("signatures", "allgather", "band_digests", "bucketing"):
        assert len(c3["per_rank_ms"][stage]) == world and all(v > 0 for v in c3["per_rank_ms"][stage]), stage


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rccl_branch_runs_to_the_end_through_the_stand_in(fake_rccl, world):
    """The command line the driver uses on an 8-GPU node, with the one thing a 1-GPU box cannot give it -- RCCL across devices -- replaced
    by the stand-in library (MHX_RCCL_LIBRARY; named in the line): transport "rccl", the all-gather probe, extra.c3_sharded with its in-place
    gather and parity gates.  What runs is every line of bench.py and libmhx the multi-GPU run will run, except RCCL itself."""
    line = _line(_bench(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--sets", "20000", "--c3-rows", "30000", "--check-rows", "256",
                         "--share-devices", "--clock-warmup", "0"], env={"MHX_RCCL_LIBRARY": fake_rccl}))
    assert line["n_gpus"] == world and line["config"]["allgather_transport"] == "rccl" and line["config"]["rccl_library_override"] == fake_rccl
    ag = line["allgather"]
    assert "error" not in ag and ag["transport"] == "rccl" and ag["rccl_ranks_seen"] == [world] * world
    c3 = line["extra"]["c3_sharded"]
    assert "error" not in c3, c3
    assert c3["rows_total"] == world * 30000 and c3["allgather"]["transport"] == "rccl"
    assert c3["allgather"]["bytes_received_per_gpu"] == (world - 1) * 30000 * 256 * 4


def test_bench_allgather_inside_every_step_over_the_host_transport():
    line = _line(_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--sets", "20000", "--check-rows", "128", "--share-devices", "--allgather",
                         "--allgather-transport", "host", "--no-c3-sharded"]))
    assert line["config"]["parallelism"] == "shard2+allgather" and line["allgather"]["transport"] == "host-shm"
    assert "extra" not in line


def test_bench_without_the_opt_in_still_asks_rccl_and_reports_its_refusal():
    """The host transport is never chosen silently: the same shared-device launch without --allgather-transport host goes to
    RCCL, which refuses two ranks on one device -- reported in the line, no c3_sharded, exit code 0."""
    line = _line(_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--sets", "20000", "--check-rows", "128", "--share-devices", "--probe-timeout", "120"]))
    assert line["config"]["allgather_transport"] == "rccl"
    assert "error" in line["allgather"] or line["allgather"].get("rccl_ranks_seen") == [2, 2]
    if "error" in line["allgather"]:
        assert "extra" not in line


# ------------------------------------------------------------------ configs 3 and 5 at their stated size
def _full(rows, env=None, timeout=1500):
    line = _line(_bench(["--full-only", "--full-rows", str(rows)], env=env, timeout=timeout))
    c3, c5 = line["extra"]["c3_full"], line["extra"]["c5_full"]
    assert c5["fused"]["one_read"] is True
    assert "all bands equal to the stable radix sort" in c3["parity"] and "blocks equal to bbit1_wide_kernel" in c5["parity"]
    return line, c3, c5


def test_configs_3_and_5_at_full_size_on_one_gpu():
    """BASELINE.json configs[2] / [4] as stated: 10M sets x 256 tokens, num_perm = 256 -- 2.56e9 tokens (> 2^31) through
    mhx_minhash_bulk_dev in one call, 2.56e9 signature values through the digest / pack / fused kernels, 320M keys through
    the bucketing (bin_bits = 12) -- with every check bench.py's extra_full knows ("all": 4 098 spread rows against the C
    oracle at every stage, every sorted band verified in full and against the stable radix sort, fused against separate)."""
    line, c3, c5 = _full(10_000_000)
    assert "1e+07" in c3["workload"] or "10000000" in c3["workload"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "full_size_c3_c5.json"), "w") as f:
        json.dump(line, f)


def test_full_size_chain_under_guard_pages():
    """The same sizes with every device allocation of the library abutting an unmapped page (MHX_GUARD_ALLOC): an index that
    wrapped at 2^31 or 2^32 elements would read or write outside its buffer and kill the process."""
    probe = subprocess.run([sys.executable, "-c", "from datasketch_amd import _native as n; n.guard_alloc(16); c = n.context(); c.alloc(100); print('ok')"],
                           capture_output=True, text=True, cwd=ROOT, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT))
    assert "ok" in probe.stdout, "the HIP virtual-memory API is not usable on this box:\n" + probe.stdout[-1000:] + probe.stderr[-1000:]
    line, c3, c5 = _full(10_000_000, env={"MHX_GUARD_ALLOC": "16"})
    assert line["guard_alloc"] == "16"


# ------------------------------------------------------------------ what a context learned about the corpus is visible
def test_minhash_mode_word_is_readable_follows_the_corpus_and_resets(ctx):
    """mhx_ctx_minhash_mode: 0 on a clean corpus, non-zero after calls whose sets mostly defeat the one-candidate proof (every
    set full of repeated tokens), 0 again after a reset -- and the signatures are the oracle's whatever the word says."""
    rng = np.random.RandomState(5)
    k, n, t = 128, 6000, 256
    a, b = O.np_init_permutations(k, 4)
    clean = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
    dirty = clean.copy()
    dirty[:, 1::2] = dirty[:, 0::2]                       # every token twice in every set
    want_clean, want_dirty = O.c_minhash_bulk_dense(clean, a, b), O.c_minhash_bulk_dense(dirty, a, b)
    assert ctx.minhash_mode(reset=True) in (0, 1, 2)
    assert ctx.minhash_mode() == 0
    for _ in range(3):
        assert np.array_equal(ctx.minhash_bulk((a, b), clean.reshape(-1), None, t, n), want_clean)
    assert ctx.minhash_mode() == 0
    for _ in range(3):
        assert np.array_equal(ctx.minhash_bulk((a, b), dirty.reshape(-1), None, t, n), want_dirty)
    learned = ctx.minhash_mode()
    assert learned in (1, 2), learned
    assert np.array_equal(ctx.minhash_bulk((a, b), clean.reshape(-1), None, t, n), want_clean)   # first launch chosen for the dirty corpus: same result
    assert ctx.minhash_mode(reset=True) in (0, 1, 2) and ctx.minhash_mode() == 0
    assert np.array_equal(ctx.minhash_bulk((a, b), dirty.reshape(-1), None, t, n), want_dirty)


# ------------------------------------------------------------------ config 4: fetcher and walker waves
@pytest.mark.parametrize("values", [False, True])
@pytest.mark.parametrize("n,dim", [(1, 4096), (3, 4096), (5, 4096), (7, 4096), (13, 4096), (255, 4096), (257, 4096), (3000, 4096), (6, 1536), (777, 1536), (9, 1024), (1500, 2048)])
def test_weighted_fetcher_walker_kernel_with_few_and_odd_row_counts(ctx, n, dim, values):
    """The kernel that splits a workgroup into four fetcher and twelve walker waves (1024 .. 4096 columns, 128 samples; config 4's shape first) with
    fewer rows than fetchers, than stripes, than workgroups, and a ragged last round; rows of every kind in one call (dense, sparse,
    empty, NaN, inf, heavy-tailed).  Every row against the C oracle, and against the one-wave-per-row kernel it replaced
    (weighted.refill 13) and the other stripe / cache settings (5, 6, 8, 9)."""
    from datasketch_amd import WeightedMinHashGenerator

    s = 128
    rng = np.random.RandomState(1000 + n)
    x = rng.uniform(0, 1, (n, dim)).astype(np.float32)
    kinds = rng.randint(0, 8, n)
    for i, kind in enumerate(kinds):
        if kind == 1:
            x[i, rng.random_sample(dim) >= 0.05] = 0          # few stored: entry by entry
        elif kind == 2:
            x[i] = rng.lognormal(0, 2.0, dim)                  # heavy tail: entries above the cut, long walks
        elif kind == 3 and n > 4:
            x[i] = 0                                           # stores nothing
        elif kind == 4:
            x[i, rng.random_sample(dim) >= 0.5] = 0
    if n >= 7:
        x[5, 17] = np.nan
        x[6, dim - 96] = np.inf
    g = WeightedMinHashGenerator(dim, s, seed=5, gpu_mode="always", device_log=values)
    wctx, _ = g._device_handle()
    out, ne = g.minhash_many_arrays(x)
    ref = WeightedMinHashGenerator(dim, s, seed=5, gpu_mode="disable")
    for i in range(n):
        if not x[i].any():
            assert not ne[i] and not out[i].any()
            continue
        assert ne[i]
        if np.isnan(x[i]).any() or np.isinf(x[i]).any():
            continue  # (floor(NaN / inf) cast to int64 differs platform by platform: compared between the kernels below)
        if not values and i % max(1, n // 40) == 0:  # (values in: the device's log, compared between kernels; logs in: the oracle)
            want = ref.minhash(x[i]).hashvalues
            assert np.array_equal(out[i], want), i
    for code in (13, 5, 6, 8, 9):
        wctx.set_option("weighted.refill", code)
        try:
            o2, n2 = g.minhash_many_arrays(x)
        finally:
            wctx.set_option("weighted.refill", 0)
        assert np.array_equal(n2, ne) and np.array_equal(o2[ne.astype(bool)], out[ne.astype(bool)]), code
