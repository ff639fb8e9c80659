"""N>1 path on CPU: two processes shard the corpus, hash their own rows, all-gather -- once over the
package's own TCP rendezvous (what bench.py and dist.py use), once with a torch.distributed gloo group
wrapped as the same three-member protocol (rank, world, allgather)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from datasketch_amd.dist import shard_by_tokens, shard_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_partition():
    for n in (0, 1, 7, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            blocks = [shard_rows(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_rows(10, 2, 2)


def test_shard_by_tokens_balances_ragged():
    rng = np.random.RandomState(0)
    lens = rng.randint(0, 500, size=1000)
    offsets = np.concatenate([[0], np.cumsum(lens)])
    blocks = shard_by_tokens(offsets, 4)
    assert blocks[0][0] == 0 and blocks[-1][1] == 1000
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(3))
    tok = [offsets[e] - offsets[b] for b, e in blocks]
    assert max(tok) - min(tok) < 2 * 500


def _corpus():
    return np.random.RandomState(5).randint(0, 2**32, (101, 33), dtype=np.uint64)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RANK_BODY = r"""
import os, sys, pickle
import numpy as np
sys.path.insert(0, {root!r})
from datasketch_amd import dist, rendezvous
from datasketch_amd.dist import allgather_signatures, bulk_signatures_sharded, shard_rows
{make_group}
tokens = np.random.RandomState(5).randint(0, 2**32, (101, 33), dtype=np.uint64)
b, e = shard_rows(tokens.shape[0], group.world, group.rank)
calls = []
def mine():                      # a rank only ever materialises its own rows
    calls.append(1)
    return tokens[b:e]
full = bulk_signatures_sharded(mine, num_perm=24, seed=3, gpu_mode="disable", group=group)
glued = allgather_signatures(full[: 10 + 5 * group.rank], group=group)   # unequal shards, counts not given
group.barrier() if hasattr(group, "barrier") else None
with open({out!r} + str(group.rank), "wb") as f:
    pickle.dump((full, glued.shape[0], len(calls)), f)
{close}
"""


def _run_two_ranks(tmp_path, make_group, close, extra_env):
    out = str(tmp_path / "rank")
    body = _RANK_BODY.format(root=ROOT, make_group=make_group, close=close, out=out)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-c", body], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        text, _ = p.communicate(timeout=240)
        assert p.returncode == 0, text.decode()
    import pickle

    from datasketch_amd import MinHash, prehashed

    want = MinHash.bulk_signatures(_corpus(), num_perm=24, seed=3, hashfunc=prehashed, gpu_mode="disable")
    for rank in range(2):
        with open(out + str(rank), "rb") as f:
            full, glued_rows, calls = pickle.load(f)
        assert np.array_equal(full, want)
        assert glued_rows == 10 + 15
        assert calls == 1


def test_two_ranks_over_own_rendezvous_sharded_bulk_equals_single_process(tmp_path):
    _run_two_ranks(tmp_path, "group = rendezvous.from_env()", "group.close()",
                   {"MHX_RDZV_ADDR": f"127.0.0.1:{_free_port()}"})


def test_two_ranks_found_through_the_launcher_environment(tmp_path):
    # torch.distributed.run exports MASTER_ADDR / MASTER_PORT (the port belongs to the launcher's own store):
    # rank 0 publishes a free port in a file named after the launcher's pid
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "LOCAL_WORLD_SIZE": "2"}
    assert "MHX_RDZV_ADDR" not in os.environ
    _run_two_ranks(tmp_path, "group = rendezvous.from_env()", "group.close()", env)


_GLOO_GROUP = r"""
import torch.distributed as tdist
tdist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
class GlooGroup:                 # the three members dist.py needs, on top of a gloo process group
    rank, world = tdist.get_rank(), tdist.get_world_size()
    def allgather(self, payload):
        box = [None] * self.world
        tdist.all_gather_object(box, bytes(payload))
        return box
group = GlooGroup()
"""


def test_two_rank_gloo_sharded_bulk_equals_single_process(tmp_path):
    pytest.importorskip("torch")
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())}
    _run_two_ranks(tmp_path, _GLOO_GROUP, "tdist.destroy_process_group()", env)


def test_group_collectives_in_threads():
    import threading

    from datasketch_amd import rendezvous

    port = _free_port()
    world, res = 3, {}

    def run(rank):
        with rendezvous.Group(rank, world, "127.0.0.1", port, timeout=30) as g:
            got = g.allgather(bytes([rank]) * (rank + 1))
            mx = g.allreduce_max(10.0 * rank)
            bc = g.broadcast(b"id-from-0" if rank == 0 else None)
            ints = g.allgather_ints([rank, 100 + rank])
            g.barrier()
            res[rank] = (got, mx, bc, ints)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    for r in range(world):
        got, mx, bc, ints = res[r]
        assert got == [b"\x00", b"\x01\x01", b"\x02\x02\x02"]
        assert mx == 20.0 and bc == b"id-from-0"
        assert ints == [[0, 100], [1, 101], [2, 102]]


def test_group_with_eight_ranks_late_joiners_and_large_frames():
    """The shape of the first 8-GPU run: eight ranks, some arriving seconds after rank 0 listens, a 128-byte id broadcast
    (the RCCL unique id), per-rank timings reduced with max, and frames of a few MB (host-side gathers of signature
    shards) repeated over many rounds without the ranks drifting apart."""
    import threading
    import time

    from datasketch_amd import rendezvous

    port = _free_port()
    world, res, errors = 8, {}, []

    def run(rank):
        try:
            time.sleep(0.15 * (rank % 4))  # ranks 1..7 trickle in; rank 0 must wait for all of them
            with rendezvous.Group(rank, world, "127.0.0.1", port, timeout=60) as g:
                uid = g.broadcast(bytes(range(128)) if rank == 0 else None)
                shard = bytes([rank]) * (2_000_000 + rank)
                sizes = []
                for rnd in range(5):
                    got = g.allgather(shard if rnd == 0 else bytes([rank, rnd]))
                    sizes.append([len(b) for b in got])
                    assert all(b[0] == r for r, b in enumerate(got))
                    g.barrier()
                slow = g.allreduce_max(1.0 + rank / 10.0)
                seen = g.allgather_ints([rank])
                res[rank] = (uid, sizes, slow, seen)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    assert sorted(res) == list(range(world))
    for r in range(world):
        uid, sizes, slow, seen = res[r]
        assert uid == bytes(range(128))
        assert sizes[0] == [2_000_000 + q for q in range(world)] and all(sz == [2] * world for sz in sizes[1:])
        assert slow == 1.7 and seen == [[q] for q in range(world)]


def test_bench_self_launch_spawns_ranks_and_propagates_failure_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher environment spawns two ranks itself; on this device-less box
    both meet at the rendezvous, find no HIP device and exit non-zero -- and so does the parent."""
    from datasketch_amd import _native

    try:
        if _native.device_count() > 0:
            pytest.skip("a GPU is visible: the launch path is exercised by the gpu tests")
    except _native.MhxError:
        pytest.skip("libmhx.so not built")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MHX_RDZV_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sets", "1000"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0
    assert (p.stdout + p.stderr).count("no HIP device visible") == 2


# ---- the rendezvous itself: strangers are dropped, frames are capped, the published port is private ----
def _group_in_thread(rank, world, port, nonce, out, **kw):
    from datasketch_amd import rendezvous

    try:
        out[rank] = rendezvous.Group(rank, world, "127.0.0.1", port, timeout=20.0, nonce=nonce, **kw)
    except Exception as e:  # noqa: BLE001
        out[rank] = e


def test_rendezvous_drops_strangers_instead_of_aborting():
    """A connection that is not a rank of the group -- wrong magic, an oversized hello, the wrong nonce, a rank number
    twice -- is closed and the group still forms; rank 0 listens on the loopback interface only."""
    import struct
    import threading
    import time

    from datasketch_amd import rendezvous

    port = _free_port()
    res = {}
    t0 = threading.Thread(target=_group_in_thread, args=(0, 2, port, "s3cret", res))
    t0.start()
    hdr = struct.Struct("<4sIQ")
    deadline = time.time() + 10
    while True:  # wait for the listener
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            assert time.time() < deadline
            time.sleep(0.02)
    s.sendall(b"GET / HTTP/1.0\r\n\r\n" + b"\0" * 16)  # not our protocol
    s.close()
    for frame in (hdr.pack(b"MHXR", 1, 1 << 40),                        # a hello that claims a terabyte
                  hdr.pack(b"MHXR", 1, 5) + b"wrong",                   # the wrong nonce
                  hdr.pack(b"MHXR", 7, 6) + b"s3cret"):                 # a rank outside the group
        with socket.create_connection(("127.0.0.1", port), timeout=1.0) as c:
            c.sendall(frame)
            time.sleep(0.05)
    t1 = threading.Thread(target=_group_in_thread, args=(1, 2, port, "s3cret", res))
    t1.start()
    t0.join(30)
    t1.join(30)
    g0, g1 = res[0], res[1]
    assert isinstance(g0, rendezvous.Group) and isinstance(g1, rendezvous.Group), (g0, g1)
    assert g0._listener.getsockname()[0] == "127.0.0.1"
    got = {}
    th = threading.Thread(target=lambda: got.update(a=g1.allgather(b"one")))
    th.start()
    assert g0.allgather(b"zero") == [b"zero", b"one"]
    th.join(10)
    assert got["a"] == [b"zero", b"one"]
    g0.close()
    g1.close()


def test_rendezvous_frame_cap_and_cleanup_on_failure(tmp_path, monkeypatch):
    """_recv_frame refuses a length above its limit before allocating; a rank 0 whose peers never come closes its
    listener and removes the published file; the file is private (0600, O_EXCL) inside a 0700 directory."""
    import stat
    import struct

    from datasketch_amd import rendezvous

    a, b = socket.socketpair()
    a.sendall(struct.pack("<4sIQ", b"MHXR", 0, 10_000))
    with pytest.raises(ConnectionError):
        rendezvous._recv_frame(b, limit=1000)
    a.close()
    b.close()
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile

    monkeypatch.setattr(tempfile, "tempdir", None)
    d = rendezvous._publish_dir()
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700
    path = os.path.join(d, "29500_1")
    seen = {}

    def peek():  # what a joining rank would read while rank 0 waits
        import time

        for _ in range(200):
            if os.path.exists(path):
                seen["mode"] = stat.S_IMODE(os.stat(path).st_mode)
                seen["info"] = open(path).read()
                return
            time.sleep(0.01)

    import threading

    th = threading.Thread(target=peek)
    th.start()
    with pytest.raises(TimeoutError):
        rendezvous.Group(0, 2, "127.0.0.1", 0, timeout=1.0, publish=path)
    th.join()
    assert seen["mode"] == 0o600 and '"nonce"' in seen["info"]
    assert not os.path.exists(path)  # removed with the failed group
    os.chmod(d, 0o755)
    with pytest.raises(PermissionError):
        rendezvous._publish_dir()


# ---- the host-staged all-gather transport (dist._allgather_host) on stand-in device buffers ----------------------
class _FakeBuffer:
    """The four members of _native.DeviceBuffer the transport uses, over one bytearray 'device memory' per context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr = len(ctx.mem)
        ctx.mem.extend(b"\xEE" * self.nbytes)

    def upload(self, arr, offset=0):
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        assert offset + raw.size <= self.nbytes
        self.ctx.mem[self.ptr + offset: self.ptr + offset + raw.size] = raw.tobytes()
        return self

    def download(self, shape, dtype, offset=0):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert offset + n <= self.nbytes
        return np.frombuffer(bytes(self.ctx.mem[self.ptr + offset: self.ptr + offset + n]), dtype=dtype).reshape(shape).copy()

    def download_into(self, out, offset=0):
        out[...] = self.download(out.shape, out.dtype, offset)
        return out


class _FakeContext:
    def __init__(self):
        self.mem = bytearray()

    def alloc(self, nbytes):
        return _FakeBuffer(self, nbytes)

    def copy_dev(self, dst, src, nbytes):
        self.mem[dst: dst + nbytes] = self.mem[src: src + nbytes]

    def synchronize(self):
        pass


@pytest.mark.parametrize("force_tcp", [False, True])
@pytest.mark.parametrize("counts", [[5, 5, 5], [7, 0, 3], [1, 9, 4]])
def test_host_staged_allgather_places_every_shard(force_tcp, counts, monkeypatch):
    """Three ranks (threads), equal and unequal shards (one empty), over /dev/shm files and over the sockets in pieces
    smaller than a shard: every rank ends up with the [sum(counts), k] uint32 matrix in rank order, and no staging
    file is left behind."""
    import glob
    import threading

    from datasketch_amd import dist, rendezvous

    monkeypatch.setattr(dist, "_FORCE_TCP", force_tcp)
    monkeypatch.setattr(dist, "_HOST_PIECE", 64)  # several pieces per shard on the socket path
    k, world, port = 6, 3, _free_port()
    shards = [np.random.RandomState(90 + r).randint(0, 2**32, (counts[r], k), dtype=np.uint64).astype(np.uint32) for r in range(world)]
    want = np.concatenate(shards)
    before = set(glob.glob("/dev/shm/mhx_gather_*"))
    res, errors = {}, []

    def run(rank):
        try:
            with rendezvous.Group(rank, world, "127.0.0.1", port, timeout=30) as g:
                ctx = _FakeContext()
                d_local = ctx.alloc(max(1, shards[rank].nbytes)).upload(shards[rank])
                got = dist.allgather_signatures_dev(ctx, d_local, counts[rank], k, counts, g, transport="host")
                res[rank] = (got.transport, got.to_host(np.uint32))
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errors, errors
    for r in range(world):
        transport, m = res[r]
        assert transport == ("host-tcp" if force_tcp else "host-shm")
        assert np.array_equal(m, want)
    assert set(glob.glob("/dev/shm/mhx_gather_*")) == before


def test_allgather_transport_is_an_explicit_choice(monkeypatch):
    from datasketch_amd import dist

    monkeypatch.delenv("MHX_ALLGATHER_TRANSPORT", raising=False)
    assert dist.allgather_transport() == "rccl"            # never host by default
    monkeypatch.setenv("MHX_ALLGATHER_TRANSPORT", "host")
    assert dist.allgather_transport() == "host"
    assert dist.allgather_transport("rccl") == "rccl"      # the argument wins
    with pytest.raises(ValueError):
        dist.allgather_transport("carrier-pigeon")
