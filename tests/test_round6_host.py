"""Round 6, host side (no GPU): the by-band exchange of band digests, ragged shards through the sharded path, the host-staged
transport's error agreement.  World-2 runs use a torch.distributed gloo group wrapped as the three-member protocol dist.py needs."""
import glob
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from datasketch_amd import dist, rendezvous

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_dist_cpu import _FakeBuffer, _FakeContext, _free_port  # noqa: E402  (stand-in device buffers over a bytearray)


def test_band_partition_covers_every_band_once():
    for bands in (1, 4, 5, 32, 33):
        for world in (1, 2, 3, 8, 40):
            part = dist.band_partition(bands, world)
            assert len(part) == world and part[0][0] == 0 and part[-1][1] == bands
            assert all(part[q][1] == part[q + 1][0] for q in range(world - 1))
            sizes = [hi - lo for lo, hi in part]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("counts,bands", [([3, 5], 4), ([4, 4, 4], 5), ([1, 0, 7, 2], 9), ([2] * 8, 32), ([6, 1], 1)])
def test_the_runs_of_the_by_band_exchange_match_pairwise_and_tile_every_destination(counts, bands):
    """What mhx_comm_exchange_dev is handed: between any two ranks the send list of one and the receive list of the other
    have the same sizes in the same order (RCCL pairs them by order), and executing the runs on numpy buffers gives every
    rank the [its bands, N] matrix of the concatenated rows."""
    world, total = len(counts), sum(counts)
    rng = np.random.RandomState(1)
    local = [rng.randint(0, 2**63, (bands, c), dtype=np.uint64) for c in counts]
    want = np.concatenate(local, axis=1)
    runs = [dist._band_runs(counts, bands, r) for r in range(world)]
    part = dist.band_partition(bands, world)
    for q in range(world):
        lo, hi = part[q]
        out = np.zeros((hi - lo) * total, dtype=np.uint64).view(np.uint8)
        covered = np.zeros(out.size, dtype=np.int32)
        for p in range(world):
            sends = [(off, size) for peer, off, size in runs[p][0] if peer == q]
            recvs = [(off, size) for peer, off, size in runs[q][1] if peer == p]
            assert [s for _, s in sends] == [s for _, s in recvs]
            src = local[p].view(np.uint8).reshape(-1)
            for (so, size), (ro, _) in zip(sends, recvs):
                out[ro: ro + size] = src[so: so + size]
                covered[ro: ro + size] += 1
        assert np.all(covered == 1)
        assert np.array_equal(out.view(np.uint64).reshape(hi - lo, total), want[lo:hi])


@pytest.mark.parametrize("force_tcp", [False, True])
@pytest.mark.parametrize("counts,bands", [([5, 5, 5], 6), ([7, 0, 3], 5), ([1, 9, 4], 2)])
def test_host_staged_by_band_exchange_places_every_run(force_tcp, counts, bands, monkeypatch):
    """Three ranks (threads) over /dev/shm files and over the sockets in pieces smaller than a run: every rank ends up with
    [its bands, N] in rank order of the rows; no staging file is left behind.  bands = 2 < world leaves a rank without bands."""
    monkeypatch.setattr(dist, "_FORCE_TCP", force_tcp)
    monkeypatch.setattr(dist, "_HOST_PIECE", 40)
    world, port = 3, _free_port()
    local = [np.random.RandomState(50 + r).randint(0, 2**63, (bands, counts[r]), dtype=np.uint64) for r in range(world)]
    want = np.concatenate(local, axis=1)
    before = set(glob.glob("/dev/shm/mhx_gather_*"))
    res, errors = {}, []

    def run(rank):
        try:
            with rendezvous.Group(rank, world, "127.0.0.1", port, timeout=30) as g:
                ctx = _FakeContext()
                d_local = ctx.alloc(max(1, local[rank].nbytes)).upload(local[rank])
                got = dist.exchange_band_digests_dev(ctx, d_local, counts[rank], bands, counts, g, transport="host")
                res[rank] = (got.transport, got.lo_band, got.hi_band, got.bytes_received, got.to_host())
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errors, errors
    for r in range(world):
        transport, lo, hi, received, m = res[r]
        assert (lo, hi) == dist.band_partition(bands, world)[r]
        assert transport == ("host-tcp" if force_tcp else "host-shm")
        assert received == (hi - lo) * (sum(counts) - counts[r]) * 8
        assert np.array_equal(m, want[lo:hi])
    assert set(glob.glob("/dev/shm/mhx_gather_*")) == before


def test_a_rank_that_fails_inside_the_host_transport_fails_every_rank_instead_of_hanging_them(monkeypatch):
    """an upload error on one rank used to skip that rank's closing barrier only.  Now every phase ends in a status
    all-gather: the failing rank's message reaches all of them, nobody waits for the rendezvous timeout, nothing is left in /dev/shm."""
    world, port, counts, k = 3, _free_port(), [4, 4, 4], 3
    before = set(glob.glob("/dev/shm/mhx_gather_*"))
    outcome = {}

    class _Broken(_FakeBuffer):
        def upload(self, arr, offset=0):
            raise RuntimeError("device upload failed (injected)")

    def run(rank):
        try:
            with rendezvous.Group(rank, world, "127.0.0.1", port, timeout=20) as g:
                ctx = _FakeContext()
                shard = np.full((counts[rank], k), rank, dtype=np.uint32)
                d_local = ctx.alloc(shard.nbytes).upload(shard)
                if rank == 1:
                    ctx.alloc = lambda n: _Broken(ctx, n)  # this rank's gathered buffer cannot be written
                dist.allgather_signatures_dev(ctx, d_local, counts[rank], k, counts, g, transport="host")
                outcome[rank] = "returned"
        except OSError as e:
            outcome[rank] = str(e)
        except Exception as e:  # noqa: BLE001
            outcome[rank] = "other: " + repr(e)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(15)
    assert all(not t.is_alive() for t in threads), "a rank is still waiting"
    assert all("rank 1" in outcome[r] and "injected" in outcome[r] for r in range(world)), outcome
    assert set(glob.glob("/dev/shm/mhx_gather_*")) == before


def test_shard_csr_balances_a_heavy_tailed_corpus_by_tokens():
    rng = np.random.RandomState(3)
    lens = np.minimum(5000, (rng.pareto(1.2, 4000) * 20).astype(np.int64))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.randint(0, 2**32, int(offsets[-1]), dtype=np.uint64)
    seen_rows, tokens = 0, []
    for rank in range(4):
        v, o, (b, e) = dist.shard_csr(values, offsets, 4, rank)
        assert b == seen_rows and o[0] == 0 and o[-1] == v.size and np.array_equal(np.diff(o), lens[b:e])
        assert np.array_equal(v, values[offsets[b]: offsets[e]])
        seen_rows = e
        tokens.append(v.size)
    assert seen_rows == 4000
    assert max(tokens) - min(tokens) <= 2 * 5000  # within one longest row of each other, where equal ROW counts would not be
    by_rows = [int(offsets[dist.shard_rows(4000, 4, r)[1]] - offsets[dist.shard_rows(4000, 4, r)[0]]) for r in range(4)]
    assert max(tokens) - min(tokens) <= max(by_rows) - min(by_rows)
    with pytest.raises(ValueError):
        dist.shard_csr(values, offsets, 4, 0, balance="weight")


_RANK_BODY = r"""
import os, sys, pickle
import numpy as np
sys.path.insert(0, {root!r})
from datasketch_amd import dist, lsh_bulk
import torch.distributed as tdist
tdist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
class GlooGroup:                 # the three members dist.py needs, on top of a gloo process group
    rank, world = tdist.get_rank(), tdist.get_world_size()
    def allgather(self, payload):
        box = [None] * self.world
        tdist.all_gather_object(box, bytes(payload))
        return box
group = GlooGroup()
rng = np.random.RandomState(11)
lens = np.minimum(400, (rng.pareto(1.1, 301) * 6).astype(np.int64))     # heavy-tailed, with empty sets
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
values = rng.randint(0, 2**32, int(offsets[-1]), dtype=np.uint64)
K, BANDS, R = 24, 6, 4
v, o, (b, e) = dist.shard_csr(values, offsets, group.world, group.rank)
counts = dist.gather_counts(e - b, group)
full = dist.bulk_signatures_sharded(lambda: (v, o), num_perm=K, seed=3, gpu_mode="disable", group=group)
# the index partitioned by band: digests of the local rows, band-major, exchanged by band, bucketed per band
mine = lsh_bulk.band_digests(full[b:e], BANDS, R, gpu_mode="disable")
shard = dist.exchange_band_digests(np.ascontiguousarray(mine.T), group=group, counts=counts)
lo, hi = dist.band_partition(BANDS, group.world)[group.rank]
order = np.stack([np.argsort(shard[j], kind="stable") for j in range(hi - lo)]) if hi > lo else np.zeros((0, full.shape[0]), dtype=np.int64)
with open({out!r} + str(group.rank), "wb") as f:
    pickle.dump((full, counts, (lo, hi), shard, order), f)
tdist.destroy_process_group()
"""


def test_two_rank_gloo_ragged_shards_and_the_by_band_exchange_equal_the_single_process_results(tmp_path):
    """World 2 over gloo: a heavy-tailed ragged corpus cut by token count, hashed per rank through the CSR form, all-gathered;
    then the by-band exchange of the band digests.  Equal to the single-process CSR call and to the digests / stable order of the
    whole matrix."""
    pytest.importorskip("torch")
    import pickle

    from datasketch_amd import MinHash, lsh_bulk, prehashed

    out = str(tmp_path / "rank")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", _RANK_BODY.format(root=ROOT, out=out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        text, _ = p.communicate(timeout=240)
        assert p.returncode == 0, text.decode()
    rng = np.random.RandomState(11)
    lens = np.minimum(400, (rng.pareto(1.1, 301) * 6).astype(np.int64))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.randint(0, 2**32, int(offsets[-1]), dtype=np.uint64)
    assert (lens == 0).any() and lens.max() > 20 * np.median(lens)
    want = MinHash.bulk_signatures((values, offsets), num_perm=24, seed=3, hashfunc=prehashed, gpu_mode="disable")
    dig = lsh_bulk.band_digests(want, 6, 4, gpu_mode="disable")  # [N, bands]
    tokens_per_rank = []
    for rank in range(2):
        with open(out + str(rank), "rb") as f:
            full, counts, (lo, hi), shard, order = pickle.load(f)
        assert np.array_equal(full, want)
        assert sum(counts) == 301 and (lo, hi) == (3 * rank, 3 * rank + 3)
        assert np.array_equal(shard, dig[:, lo:hi].T)
        for j in range(lo, hi):
            assert np.array_equal(order[j - lo], np.argsort(dig[:, j], kind="stable"))
        b = sum(counts[:rank])
        tokens_per_rank.append(int(offsets[b + counts[rank]] - offsets[b]))
    assert abs(tokens_per_rank[0] - tokens_per_rank[1]) <= 400  # balanced by tokens, within one longest row


# ---- the inverse wire formats and packed byte input, host (numpy) side; the device side is tests/test_gpu_round6.py ----
def _golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f), np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))


def test_deserialize_matrix_reads_what_the_reference_wrote_in_every_byte_order():
    """ref: lean_minhash.py:177-214 -- the records come from the reference's own serialize, the expected values from its
    own deserialize (tests/golden, oracle/gen_golden.py section 10b)."""
    import struct

    from datasketch_amd import LeanMinHash
    from datasketch_amd.lean_minhash import deserialize_matrix, serialize_matrix

    meta, arrays = _golden()
    for bo, name in (("<", "le"), (">", "be"), ("@", "native"), ("!", "network")):
        g = meta[f"lean_deserialize_{name}"]
        rec = bytes.fromhex(g["bytes"])
        seeds, sig = deserialize_matrix(rec * 3, byteorder=bo, gpu_mode="disable")
        assert seeds.tolist() == [g["seed"]] * 3 and sig.dtype == np.uint64
        assert sig.tolist() == [g["hashvalues"]] * 3
        one = LeanMinHash.deserialize(rec, bo)  # the per-object form agrees
        assert one.seed == g["seed"] and one.hashvalues.tolist() == g["hashvalues"]
    # round trip of a matrix, and the error cases
    sig = np.random.RandomState(2).randint(0, 2**32, (17, 24), dtype=np.uint64)
    raw = serialize_matrix(sig, 99, gpu_mode="disable")
    seeds, back = deserialize_matrix(raw, gpu_mode="disable")
    assert np.array_equal(back, sig) and np.all(seeds == 99)
    assert np.array_equal(deserialize_matrix(raw, num_perm=24, byteorder="<", gpu_mode="disable")[1], sig)
    bad = raw.copy()
    bad[5, 8] ^= 1  # one record claims another length
    with pytest.raises(ValueError):
        deserialize_matrix(bad, gpu_mode="disable")
    with pytest.raises(ValueError):
        deserialize_matrix(raw.reshape(-1)[:-4], gpu_mode="disable")
    with pytest.raises(struct.error):
        deserialize_matrix(raw, byteorder="?", gpu_mode="disable")
    assert deserialize_matrix(b"", num_perm=8, gpu_mode="disable")[1].shape == (0, 8)


def test_unpack_matrix_restores_what_the_reference_restores():
    """ref: b_bit_minhash.py:103-125 -- the packed states and the restored hashvalues both come from the reference."""
    from datasketch_amd.b_bit_minhash import pack_matrix, unpack_matrix

    meta, arrays = _golden()
    for b, state in meta["bbit_states_k48"].items():
        raw = bytes.fromhex(state)
        blocks = np.frombuffer(raw[21:], dtype="<u8").reshape(1, -1)  # behind the 21-byte "<qBdi" header
        got = unpack_matrix(blocks, 48, int(b), gpu_mode="disable")
        assert got.dtype == np.uint32 and got[0].tolist() == meta["bbit_restored_k48"][b]
    sig = np.random.RandomState(4).randint(0, 2**32, (9, 70), dtype=np.uint64)
    for b in (0, 1, 2, 3, 4, 6, 8, 11, 16, 25, 32):
        assert np.array_equal(unpack_matrix(pack_matrix(sig, b, gpu_mode="disable"), 70, b, gpu_mode="disable"), (sig & np.uint64((1 << b) - 1)).astype(np.uint32))
    with pytest.raises(ValueError):
        unpack_matrix(np.zeros((2, 3), dtype=np.uint64), 70, 8, gpu_mode="disable")
    with pytest.raises(ValueError):
        unpack_matrix(np.zeros((2, 3), dtype=np.uint64), 70, 33, gpu_mode="disable")


def test_bulk_signatures_from_packed_byte_tokens_equals_the_per_object_corpus():
    """packed=(buf, byte_offsets, set_offsets): the same signatures as the list-of-lists corpus (ref: minhash.py:491-522 with the
    default hashfunc), on the host path here; chunk boundaries inside the corpus; empty sets and empty tokens."""
    from datasketch_amd import MinHash, minhash as mh_mod, sha1_hash64

    rng = np.random.RandomState(8)
    sets = [[bytes(rng.randint(0, 256, rng.randint(0, 12), dtype=np.uint8)) for _ in range(rng.randint(0, 9))] for _ in range(40)]
    sets[3], sets[-1] = [], []
    flat = [t for s in sets for t in s]
    buf = np.frombuffer(b"".join(flat), dtype=np.uint8)
    byte_offsets = np.concatenate([[0], np.cumsum([len(t) for t in flat])]).astype(np.int64)
    set_offsets = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int64)
    want = MinHash.bulk_signatures(sets, num_perm=16, seed=5, gpu_mode="disable")
    got = MinHash.bulk_signatures(packed=(buf, byte_offsets, set_offsets), num_perm=16, seed=5, gpu_mode="disable")
    assert np.array_equal(got, want)
    old = mh_mod._BULK_CHUNK_SETS, mh_mod._BULK_CHUNK_TOKENS
    try:
        mh_mod._BULK_CHUNK_SETS, mh_mod._BULK_CHUNK_TOKENS = 7, 11  # many chunks, some cut by the token budget
        assert np.array_equal(MinHash.bulk_signatures(packed=(bytes(buf), byte_offsets, set_offsets), num_perm=16, seed=5, gpu_mode="disable"), want)
    finally:
        mh_mod._BULK_CHUNK_SETS, mh_mod._BULK_CHUNK_TOKENS = old
    want64 = MinHash.bulk_signatures(sets, num_perm=16, seed=5, hashfunc=sha1_hash64, gpu_mode="disable")
    assert np.array_equal(MinHash.bulk_signatures(packed=(buf, byte_offsets, set_offsets), num_perm=16, seed=5, hashfunc=sha1_hash64, gpu_mode="disable"), want64)
    assert MinHash.bulk_signatures(packed=(buf, byte_offsets, set_offsets), num_perm=16, seed=5, out_dtype=np.uint32, gpu_mode="disable").dtype == np.uint32
    with pytest.raises(ValueError):
        MinHash.bulk_signatures(sets, packed=(buf, byte_offsets, set_offsets), num_perm=16)
    with pytest.raises(ValueError):
        MinHash.bulk_signatures(packed=(buf, byte_offsets, set_offsets + 1), num_perm=16, gpu_mode="disable")
    with pytest.raises(ValueError):
        MinHash.bulk_signatures(packed=(buf[:-1], byte_offsets, set_offsets), num_perm=16, gpu_mode="disable")


def test_a_second_hello_cannot_evict_a_live_rank_and_needs_a_nonce_to_replace_a_dead_one():
    """rank 0 replaces a registered rank's socket only when the hello carried the group's nonce AND the registered
    socket is dead.  An impostor claiming rank 1 while the real rank 1 is connected is turned away (nonce or not), and the group works."""
    import struct
    import time

    for nonce in ("n6", None):
        port = _free_port()
        res = {}

        def run(rank, delay=0.0):
            time.sleep(delay)
            try:
                with rendezvous.Group(rank, 3, "127.0.0.1", port, timeout=30, nonce=nonce) as g:
                    res[rank] = g.allgather(bytes([rank]))
            except Exception as e:  # noqa: BLE001
                res[rank] = e

        threads = [threading.Thread(target=run, args=(0,)), threading.Thread(target=run, args=(1, 0.1))]
        for t in threads:
            t.start()
        time.sleep(0.6)  # rank 1 is registered and alive, rank 2 still missing
        c = socket.create_connection(("127.0.0.1", port), timeout=2.0)
        c.sendall(struct.pack("<4sIQ", b"MHXR", 1, len(nonce or "")) + (nonce or "").encode())
        c.settimeout(3.0)
        try:
            got = c.recv(64)
        except OSError:
            got = b""
        assert got == b"", "the impostor was acknowledged"  # dropped: EOF, no ACK
        c.close()
        t2 = threading.Thread(target=run, args=(2,))
        t2.start()
        for t in threads + [t2]:
            t.join(40)
        assert all(res.get(r) == [b"\x00", b"\x01", b"\x02"] for r in range(3)), (nonce, res)
