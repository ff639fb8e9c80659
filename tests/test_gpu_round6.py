"""GPU parity tests added in round 6 (run on an MI355X: python -m pytest tests -m gpu -x -q).  Everything goes through the C ABI.

* the headline corpus compared with the C oracle on EVERY row (1M x 256, K = 128), the sets the launches' own hand-over flags name
  (mhx_ctx_minhash_flags: the rare-event paths) checked by id, and the same on corpora full of repeated tokens where nearly all
  sets take the tie-tolerant proof (ref: minhash.py:293-297);
* configs 3 / 5 with the index partitioned by band: worlds of 2 and 8 on one GPU, equal and unequal shards, the by-band exchange of
  band digests (mhx_comm_exchange_dev) over the host-staged transport and over the RCCL binding with the stand-in library -- every
  rank's sorted bands byte-identical to the single-process result and the oracle (ref: lsh.py:199,326-347; b_bit_minhash.py:78-101);
* ragged (heavy-tailed) shards through dist.bulk_signatures_sharded (ref: minhash.py:491-522);
* the inverse wire formats in bulk (ref: lean_minhash.py:177-214, b_bit_minhash.py:103-125) and packed byte-token input.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from datasketch_amd import MinHash, _native, dist, lsh_bulk, prehashed
from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_gpu_round5 import _free_port, fake_rccl  # noqa: E402,F401  (the stand-in RCCL fixture)


@pytest.fixture(scope="module")
def ctx():
    assert _native.gpu_available(), "these tests need an MI355X"
    return _native.context()


# ------------------------------------------------------------------ every row of the headline, and the flagged sets by id
def _bulk_with_flags(ctx, tok, k, seed=1, out_dtype=np.uint64):
    n, t = tok.shape
    a, b = O.np_init_permutations(k, seed)
    d_tok, d_sig = ctx.to_device(tok), ctx.alloc(n * k * np.dtype(out_dtype).itemsize)
    code = _native.MHX_U64 if out_dtype == np.uint64 else _native.MHX_U32
    ctx.minhash_mode(reset=True)  # a fresh context's first launch: the one-candidate proof
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_sig.ptr, code)
    flags = ctx.minhash_flags(n)
    return d_sig.download((n, k), out_dtype), flags, (a, b)


def test_every_row_of_the_headline_corpus_equals_the_oracle_and_the_flagged_sets_do(ctx):
    """BASELINE.json configs[1], all 1 000 000 rows (rounds 1-5 sampled ~1 %).  The flags name the sets the first launch's
    certificate rejected -- a few hundred per million of distinct random tokens -- and every one of them is compared by id, so the
    'unconditionally exact' claim no longer rests on a sample happening to contain them."""
    n, t, k = 1_000_000, 256, 128
    tok = np.random.RandomState(42).randint(0, 2**32, (n, t), dtype=np.uint64)
    sig, flags, (a, b) = _bulk_with_flags(ctx, tok, k)
    t0 = time.time()
    want = O.c_minhash_bulk_dense_parallel(tok, a, b)
    oracle_s = time.time() - t0
    assert np.array_equal(sig, want), f"rows differing: {np.flatnonzero((sig != want).any(axis=1))[:10].tolist()}"
    flagged = np.flatnonzero(flags)
    assert set(np.unique(flags).tolist()) <= {0, 1, 2}
    assert 20 <= flagged.size <= 5000, flagged.size  # (r05: 383 + 60 per million)
    assert np.array_equal(sig[flagged], want[flagged])          # (implied by the full comparison; kept as the by-id statement)
    pairwise = np.flatnonzero(flags == 2)
    assert np.array_equal(sig[pairwise], O.c_minhash_bulk_dense(tok[pairwise], a, b)) if pairwise.size else True
    print(f"[round6] 1M rows vs the C oracle in {oracle_s:.1f} s on {O.usable_threads()} threads; flagged {flagged.size} (pairwise {pairwise.size})")


@pytest.mark.parametrize("repeat", [0.01, 0.10])
def test_repeat_corpora_where_nearly_every_set_leaves_the_first_proof_all_rows(ctx, repeat):
    """A share of every set's tokens are copies of its other tokens (equal tokens at the minimum defeat the one-candidate
    proof): with a fresh context's first launch nearly all sets are flagged and go through the second / third launch.  All rows
    against the oracle, the flagged ones by id; then the same corpus again, now with the tie-tolerant proof FIRST (what the
    context learned), again all rows."""
    n, t, k = 200_000, 256, 128
    rng = np.random.RandomState(7)
    tok = rng.randint(0, 2**32, (n, t), dtype=np.uint64)
    m = max(1, int(t * repeat))
    src = rng.randint(0, t - m, (n, m))
    tok[:, t - m:] = np.take_along_axis(tok, src, axis=1)  # the last m tokens repeat earlier ones of the same set
    sig, flags, (a, b) = _bulk_with_flags(ctx, tok, k)
    want = O.c_minhash_bulk_dense_parallel(tok, a, b)
    assert np.array_equal(sig, want)
    flagged = np.flatnonzero(flags)
    assert flagged.size > 0.2 * n, flagged.size  # the rare-event path is the common one here
    assert np.array_equal(sig[flagged], want[flagged])
    mode = ctx.minhash_mode()
    d_tok, d_sig = ctx.to_device(tok), ctx.alloc(n * k * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_sig.ptr, _native.MHX_U64)  # first launch by `mode`
    flags2 = ctx.minhash_flags(n)
    assert np.array_equal(d_sig.download((n, k), np.uint64), want)
    print(f"[round6] repeat {repeat}: flagged {flagged.size} of {n} (pairwise {int((flags == 2).sum())}); learned mode {mode}, then flagged {int((flags2 != 0).sum())}")
    ctx.minhash_mode(reset=True)


def test_flags_entry_point_refuses_what_it_cannot_answer(ctx):
    a, b = O.np_init_permutations(64, 1)
    tok = np.random.RandomState(3).randint(0, 2**32, (1000, 40), dtype=np.uint64)
    d_tok, d_sig = ctx.to_device(tok), ctx.alloc(1000 * 64 * 8)
    ctx.minhash_bulk_dev((a, b), d_tok.ptr, _native.MHX_U64, None, 40, 1000, tok.size, None, 0, d_sig.ptr, _native.MHX_U64)
    assert ctx.minhash_flags(1000).shape == (1000,)
    with pytest.raises(ValueError):
        ctx.minhash_flags(999)  # not the last call's set count
    one = np.random.RandomState(4).randint(0, 2**32, (1, 60_000), dtype=np.uint64)  # one huge set: split over waves, no flags
    d_one, d_o = ctx.to_device(one), ctx.alloc(64 * 8)
    ctx.minhash_bulk_dev((a, b), d_one.ptr, _native.MHX_U64, None, 60_000, 1, one.size, None, 0, d_o.ptr, _native.MHX_U64)
    with pytest.raises(ValueError):
        ctx.minhash_flags(1)


# ------------------------------------------------------------------ the index partitioned by band, N > 1 on one GPU
def _run_ranks(args, world, out, transport="host", timeout=600, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = {key: v for key, v in os.environ.items() if key not in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MHX_RDZV_ADDR=f"127.0.0.1:{port}",
                   MHX_RDZV_NONCE="round6", MHX_TEST_TRANSPORT=transport)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_c5.py")] + [str(a) for a in args] + [out],
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
    return [json.load(open(f"{out}.{rank}.json")) for rank in range(world)], [np.load(f"{out}.{rank}.npz") for rank in range(world)]


def _check_by_band(recs, arrays, n, t, k, bands, r, world, scheme, transport_name):
    import rank_c3

    counts = rank_c3.split(n, world, scheme)
    tokens = rank_c3.corpus(n, t)
    a, b = O.np_init_permutations(k, 3)
    want = O.c_minhash_bulk_dense(tokens, a, b)
    dig = lsh_bulk.band_digests(want, bands, r, gpu_mode="disable")      # [n, bands]: FNV-1a-64 of the reference's key bytes
    blocks = O.c_bbit_pack(want, 1)
    part = dist.band_partition(bands, world)
    begin = 0
    for q in range(world):
        rec, arr = recs[q], arrays[q]
        lo, hi = part[q]
        assert rec["counts"] == counts and (rec["lo_band"], rec["hi_band"]) == (lo, hi) and rec["rows"] == n
        assert rec["transport"] == transport_name and rec["fused"] == (counts[q] > 0)
        assert rec["bytes_received"] == (hi - lo) * (n - counts[q]) * 8      # 8 bytes per (row, band) it buckets, from the other ranks only
        assert np.array_equal(arr["blocks"], blocks[begin: begin + counts[q]])   # b = 1 blocks of the rank's OWN rows
        assert np.array_equal(arr["digests"], np.ascontiguousarray(dig[:, lo:hi].T))
        for j in range(lo, hi):
            order = np.lexsort((np.arange(n), dig[:, j]))                   # the stable (digest, row) order of the whole column
            assert np.array_equal(arr["sorted_rows"][j - lo], order.astype(np.uint32))
            assert np.array_equal(arr["sorted_digests"][j - lo], dig[order, j])
        begin += counts[q]
    if scheme == "unequal":
        assert counts[0] == 1 and len(set(counts)) >= min(world, 3)
    return want, dig


@pytest.mark.parametrize("world,scheme", [(2, "equal"), (2, "unequal"), (8, "equal"), (8, "unequal")])
def test_band_partitioned_index_with_ranks_sharing_one_gpu(ctx, tmp_path, world, scheme):
    """`world` processes, one GPU, host-staged transport: every rank hashes its shard, packs b = 1 blocks and band-major digests of
    ITS rows in one read, sends every peer that peer's bands and buckets its own bands over all rows.  No signature matrix is ever
    assembled.  Blocks == the oracle's packing of the rank's rows; the exchanged digests == FNV-1a-64 of the reference's key bytes
    for (all rows, its bands); every sorted band == the stable order of the whole column; and the single-process device chain agrees."""
    n, t, k, bands, r = 40_000, 96, 128, 16, 8
    recs, arrays = _run_ranks(["byband", n, t, k, bands, r, scheme], world, str(tmp_path / f"bb_{world}_{scheme}"))
    want, dig = _check_by_band(recs, arrays, n, t, k, bands, r, world, scheme, "host-shm")
    # single process, same kernels: sort of the band-major digests of the whole matrix
    d_sig = ctx.to_device(want.astype(np.uint32))
    d_dig, d_sd, d_sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
    _native.check(ctx.lib.mhx_band_digests_layout_dev(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, bands, r, _native.BAND_MAJOR, d_dig.ptr))
    _native.check(ctx.lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig.ptr, n, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr))
    ctx.synchronize()
    sd, sr = d_sd.download((bands, n), np.uint64), d_sr.download((bands, n), np.uint32)
    part = dist.band_partition(bands, world)
    for q in range(world):
        lo, hi = part[q]
        assert arrays[q]["sorted_digests"].tobytes() == sd[lo:hi].tobytes() and arrays[q]["sorted_rows"].tobytes() == sr[lo:hi].tobytes()


@pytest.mark.parametrize("world,scheme", [(2, "unequal"), (8, "equal"), (8, "unequal")])
def test_by_band_exchange_through_the_rccl_binding_with_the_stand_in_library(ctx, tmp_path, fake_rccl, world, scheme):
    """mhx_comm_exchange_dev (one group of ncclSend / ncclRecv, a run per (peer, band), self-runs as device copies) executed for
    real from `world` processes against tests/fake_rccl.c: pins OUR marshalling -- peers, offsets, sizes, order -- with worlds of 2
    and 8; RCCL and xGMI are not in the picture."""
    n, t, k, bands, r = 24_000, 64, 128, 16, 8
    recs, arrays = _run_ranks(["byband", n, t, k, bands, r, scheme], world, str(tmp_path / f"bbr_{world}_{scheme}"), transport="rccl",
                              extra_env={"MHX_RCCL_LIBRARY": fake_rccl})
    _check_by_band(recs, arrays, n, t, k, bands, r, world, scheme, "rccl")


def test_by_band_exchange_with_more_ranks_than_bands_and_a_non_fused_shape(ctx, tmp_path):
    """bands = 6 over 8 ranks: two ranks own no band (they still send); K = 96, r = 16 is a shape the fused kernel declines."""
    n, t, k, bands, r = 12_000, 48, 96, 6, 16
    world = 8
    recs, arrays = _run_ranks(["byband", n, t, k, bands, r, "unequal"], world, str(tmp_path / "bb_few_bands"))
    import rank_c3

    counts = rank_c3.split(n, world, "unequal")
    a, b = O.np_init_permutations(k, 3)
    want = O.c_minhash_bulk_dense(rank_c3.corpus(n, t), a, b)
    dig = lsh_bulk.band_digests(want, bands, r, gpu_mode="disable")
    part = dist.band_partition(bands, world)
    assert sum(1 for lo, hi in part if hi == lo) == 2
    blocks, begin = O.c_bbit_pack(want, 1), 0
    for q in range(world):
        lo, hi = part[q]
        assert not recs[q]["fused"] and recs[q]["bytes_received"] == (hi - lo) * (n - counts[q]) * 8
        assert np.array_equal(arrays[q]["blocks"], blocks[begin: begin + counts[q]])
        begin += counts[q]
        if hi == lo:
            assert "sorted_rows" not in arrays[q].files
            continue
        for j in range(lo, hi):
            order = np.lexsort((np.arange(n), dig[:, j]))
            assert np.array_equal(arrays[q]["sorted_rows"][j - lo], order.astype(np.uint32)) and np.array_equal(arrays[q]["sorted_digests"][j - lo], dig[order, j])


def test_exchange_with_one_rank_through_rccl(ctx):
    """The real librccl, a communicator of one rank: every run is a run the rank owes itself (device copies, an empty group)."""
    from datasketch_amd import rendezvous

    group = rendezvous.Group(0, 1)
    n, bands = 1000, 8
    dig = np.random.RandomState(5).randint(0, 2**63, (bands, n), dtype=np.uint64)
    comm = dist.communicator(ctx, group)
    d_in, d_out = ctx.to_device(dig), ctx.alloc(dig.nbytes)
    sends, recvs = dist._band_runs([n], bands, 0)
    comm.exchange_dev(d_in.ptr, d_out.ptr, sends, recvs)
    ctx.synchronize()
    assert np.array_equal(d_out.download((bands, n), np.uint64), dig)
    with pytest.raises(ValueError):
        comm.exchange_dev(d_in.ptr, d_out.ptr, [(0, 0, 64)], [(0, 0, 32)])  # a self-run whose two ends disagree
    with pytest.raises(ValueError):
        comm.exchange_dev(d_in.ptr, d_out.ptr, [(3, 0, 64)], [])            # a peer outside the communicator
    group.close()


# ------------------------------------------------------------------ ragged shards
@pytest.mark.parametrize("world,transport", [(8, "host"), (2, "rccl")])
def test_ragged_shards_balanced_by_tokens_equal_the_single_process_csr_call(ctx, tmp_path, fake_rccl, world, transport):
    """A heavy-tailed corpus (Pareto lengths, empty sets, a few 64-bit tokens) cut by token count, every rank hashing its CSR shard
    (ref: minhash.py:491-522 takes arbitrary iterables), all-gathered in unequal shards: equal to the single-process CSR call and to
    the oracle on every row."""
    import rank_c5

    n, k = 30_000, 64
    env = {"MHX_RCCL_LIBRARY": fake_rccl} if transport == "rccl" else None
    recs, arrays = _run_ranks(["ragged", n, k], world, str(tmp_path / f"rag_{world}"), transport=transport, extra_env=env)
    values, offsets = rank_c5.ragged_corpus(n)
    a, b = O.np_init_permutations(k, 3)
    want = O.c_minhash_bulk(values, offsets, a, b)
    assert np.array_equal(MinHash.bulk_signatures((values, offsets), num_perm=k, seed=3, hashfunc=prehashed, gpu_mode="always"), want)
    for q in (0, world - 1):
        assert np.array_equal(arrays[q]["sig"].astype(np.uint64), want)
    rows = [rec["rows"] for rec in recs]
    assert rows[0][0] == 0 and rows[-1][1] == n and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    tokens = [rec["tokens"] for rec in recs]
    lens = np.diff(offsets)
    assert (lens == 0).any() and max(tokens) - min(tokens) <= 2 * int(lens.max())     # balanced by tokens ...
    assert len({e - b for b, e in rows}) > 1                                            # ... not by rows
    assert all(rec["transport"] == ("rccl" if transport == "rccl" else "host-shm") for rec in recs)


# ------------------------------------------------------------------ inverse wire formats, packed byte input
@pytest.mark.parametrize("k", [1, 7, 64, 130, 256])
def test_bbit_unpack_is_the_inverse_of_pack_for_every_b(ctx, k):
    from datasketch_amd.b_bit_minhash import pack_matrix, unpack_matrix

    sig = np.random.RandomState(k).randint(0, 2**32, (513, k), dtype=np.uint64)
    for b in range(0, 33):
        blocks = O.c_bbit_pack(sig, b)
        got = unpack_matrix(blocks, k, b, gpu_mode="always")
        assert got.dtype == np.uint32 and np.array_equal(got, (sig & np.uint64((1 << b) - 1)).astype(np.uint32)), (k, b)
        assert np.array_equal(got, unpack_matrix(blocks, k, b, gpu_mode="disable"))
    assert np.array_equal(pack_matrix(sig, 5, gpu_mode="always"), O.c_bbit_pack(sig, 5))


def test_bbit_unpack_and_lean_deserialize_read_the_reference_s_own_bytes(ctx):
    from datasketch_amd.b_bit_minhash import unpack_matrix
    from datasketch_amd.lean_minhash import deserialize_matrix

    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        meta = json.load(f)
    for b, state in meta["bbit_states_k48"].items():
        blocks = np.frombuffer(bytes.fromhex(state)[21:], dtype="<u8").reshape(1, -1)
        assert unpack_matrix(np.repeat(blocks, 5, axis=0), 48, int(b), gpu_mode="always").tolist() == [meta["bbit_restored_k48"][b]] * 5
    for bo, name in (("<", "le"), (">", "be"), ("@", "native"), ("!", "network")):
        g = meta[f"lean_deserialize_{name}"]
        seeds, sig = deserialize_matrix(bytes.fromhex(g["bytes"]) * 4, byteorder=bo, gpu_mode="always")
        assert seeds.tolist() == [g["seed"]] * 4 and sig.tolist() == [g["hashvalues"]] * 4


@pytest.mark.parametrize("n,k", [(1, 1), (1000, 128), (257, 129), (50_000, 256)])
def test_lean_records_round_trip_in_both_byte_orders(ctx, n, k):
    from datasketch_amd.lean_minhash import deserialize_matrix, serialize_matrix

    sig = np.random.RandomState(n + k).randint(0, 2**32, (n, k), dtype=np.uint64)
    raw = serialize_matrix(sig, -12345, gpu_mode="always")
    assert np.array_equal(raw, O.c_lean_serialize(sig, -12345))
    seeds, back = deserialize_matrix(raw, gpu_mode="always")
    assert np.array_equal(back, sig) and np.all(seeds == -12345)
    big = b"".join(O.np_lean_serialize(row, 77, ">") for row in sig[:64])
    seeds, back = deserialize_matrix(big, num_perm=k, byteorder=">", gpu_mode="always")
    assert np.array_equal(back, sig[:64]) and np.all(seeds == 77)
    bad = raw.copy()
    bad[n // 2, 9] ^= 2
    with pytest.raises(ValueError):
        deserialize_matrix(bad, num_perm=k, gpu_mode="always")
    # device to device, uint32 signatures, big-endian: what the typed entry points add
    d_sig, d_rec, d_back = ctx.to_device(sig.astype(np.uint32)), ctx.alloc(n * (12 + 4 * k)), ctx.alloc(n * k * 4)
    _native.check(ctx.lib.mhx_lean_serialize_dev_typed(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, 5, 1, d_rec.ptr))
    _native.check(ctx.lib.mhx_lean_deserialize_dev(ctx.handle, d_rec.ptr, n, k, 1, _native.MHX_U32, d_back.ptr, None, None))
    ctx.synchronize()
    assert d_rec.download((min(n, 64) * (12 + 4 * k),), np.uint8).tobytes() == b"".join(O.np_lean_serialize(row, 5, ">") for row in sig[:64])
    assert np.array_equal(d_back.download((n, k), np.uint32), sig.astype(np.uint32))


def test_packed_byte_tokens_go_to_the_device_without_per_object_packing(ctx):
    """MinHash.bulk_signatures(packed=(buf, byte_offsets, set_offsets)): SHA-1 + MinHash on the device from one packed buffer;
    equal to the list-of-lists corpus on the device and to the numpy path with hashlib (ref: hashfunc.py:5-28, minhash.py:262-263)."""
    from datasketch_amd import sha1_hash64

    rng = np.random.RandomState(21)
    sets = [[bytes(rng.randint(0, 256, rng.randint(0, 70), dtype=np.uint8)) for _ in range(rng.randint(0, 60))] for _ in range(3000)]
    sets[17] = []
    flat = [tk for s in sets for tk in s]
    buf = np.frombuffer(b"".join(flat), dtype=np.uint8)
    byte_offsets = np.concatenate([[0], np.cumsum([len(tk) for tk in flat])]).astype(np.int64)
    set_offsets = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int64)
    for kw in ({}, {"hashfunc": sha1_hash64}):
        want = MinHash.bulk_signatures(sets, num_perm=128, seed=9, gpu_mode="disable", **kw)
        assert np.array_equal(MinHash.bulk_signatures(packed=(buf, byte_offsets, set_offsets), num_perm=128, seed=9, gpu_mode="always", **kw), want)
        assert np.array_equal(MinHash.bulk_signatures(sets, num_perm=128, seed=9, gpu_mode="always", **kw), want)
    from datasketch_amd import minhash as mh_mod

    old = mh_mod._BULK_CHUNK_SETS, mh_mod._BULK_CHUNK_TOKENS
    try:
        mh_mod._BULK_CHUNK_SETS, mh_mod._BULK_CHUNK_TOKENS = 700, 9000
        got = MinHash.bulk_signatures(packed=(buf, byte_offsets, set_offsets), num_perm=128, seed=9, gpu_mode="always", out_dtype=np.uint32)
    finally:
        mh_mod._BULK_CHUNK_SETS, mh_mod._BULK_CHUNK_TOKENS = old
    assert got.dtype == np.uint32 and np.array_equal(got, MinHash.bulk_signatures(sets, num_perm=128, seed=9, gpu_mode="disable"))


# ------------------------------------------------------------------ weighted CSR: the crossover by cost, the gated launches
@pytest.mark.parametrize("dim,s,density", [(1024, 64, 0.01), (1024, 64, 0.1), (1024, 64, 0.3), (4096, 128, 0.01), (4096, 128, 0.09), (1024, 256, 0.17), (512, 192, 0.5)])
def test_weighted_csr_rows_on_both_sides_of_the_cost_crossover_equal_the_oracle(ctx, dim, s, density):
    """mhx_weighted_minhash_many_dev over CSR rows of mixed lengths around csr_row_is_walked's crossover: some rows entry by entry,
    some walked, in one call -- and a call NONE of whose rows is walked (the gated plan / walk launches do nothing), followed by one
    where all are (the plan must still be built then).  (k, t) bit-exact against the C oracle (ref: weighted_minhash.py:192-247)."""
    import scipy.sparse as sp

    from datasketch_amd import WeightedMinHashGenerator

    rng = np.random.RandomState(int(dim + s + density * 1000))
    g = WeightedMinHashGenerator(dim, s, seed=3, gpu_mode="always")
    n = 600
    x = rng.uniform(0, 100, (n, dim)).astype(np.float32)
    dens = np.clip(density * rng.uniform(0.2, 2.5, n), 0.0, 1.0)  # rows on both sides of the crossover
    x[rng.random_sample(x.shape) >= dens[:, None]] = 0
    x[5] = 0
    for part in (x, x[dens < density * 0.5], x[dens > density * 1.5]):
        if part.shape[0] == 0:
            continue
        csr = sp.csr_matrix(part)
        csr.sort_indices()
        out, ne = g.minhash_many_arrays(csr)
        wo, wn = O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)
        assert np.array_equal(ne, wn) and np.array_equal(out, wo)


# ------------------------------------------------------------------ bucketing: the big bins' range of sizes
@pytest.mark.parametrize("n", [2_560_001, 10_480_000, 10_490_000])
def test_big_bin_bucketing_at_the_ends_of_its_range(ctx, n):
    """Round 6: between 2.56M and 10.2M rows a band is spread over 1024 bins of up to 11 264 elements and finished by the big form of the bin pass
    (launch_lsh_bucket_bands); just above 2^10 x 2500 rows, at the last size that still takes it ((n >> 10) <= 10 240) and just beyond (three passes
    again) the sorted bands must be numpy's stable order, with a cluster of equal digests inside one bin's capacity."""
    bands = 2
    rng = np.random.RandomState(n % 1000)
    dig = rng.randint(0, 2**63, (bands, n), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, (bands, n)).astype(np.uint64)
    dig[1, rng.randint(0, n, 5000)] = dig[1, 7]  # a bucket of ~5 000 equal digests: ranked inside one sub-bucket by the row
    d_dig = ctx.to_device(dig)
    d_sd, d_sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
    _native.check(ctx.lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig.ptr, n, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr))
    ctx.synchronize()
    sd, sr = d_sd.download((bands, n), np.uint64), d_sr.download((bands, n), np.uint32)
    for j in range(bands):
        order = np.argsort(dig[j], kind="stable")
        assert np.array_equal(sr[j], order.astype(np.uint32)), j
        assert np.array_equal(sd[j], dig[j][order]), j
    for d in (d_dig, d_sd, d_sr):
        d.free()
    ctx.release_scratch()
