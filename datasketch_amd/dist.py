"""Multi-GPU sharding of the bulk MinHash path: one process per GPU, corpus split by rows.

The path is embarrassingly parallel over sets (every rank needs only the 2*K permutation
parameters, regenerated from the seed), so the compute phase has NO collective.  The only
exchange step is the optional assembly of the full ``[N, K]`` signature matrix on every rank
(an all-gather of row shards) when one consumer -- ``MinHashLSH`` insertion, b-bit packing and
band hashing of the whole corpus -- needs it whole.

On the GPU the collective is RCCL over xGMI through libmhx's own binding (``mhx_comm_*`` in
include/mhx.h, :class:`datasketch_amd._native.Communicator`): device buffers in, device buffer
out, enqueued on the kernel's stream, uint32 on the wire (signature values are < 2**32: half the
bytes).  Host-side plumbing -- the 128-byte RCCL id, shard sizes, barriers -- goes over
:mod:`datasketch_amd.rendezvous` (plain TCP, no PyTorch); on a host without GPUs the same group
carries the shards themselves, which is what the CPU tests exercise.  Any object with ``rank``,
``world`` and ``allgather(bytes) -> list[bytes]`` can stand in for the group (the tests wrap a
``torch.distributed`` gloo group that way).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os
import secrets

import numpy as np

from datasketch_amd import rendezvous


def shard_rows(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block ``[begin, end)`` of rank ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_rows), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_by_tokens(offsets: Sequence[int], world_size: int) -> List[Tuple[int, int]]:
    """Contiguous row blocks balanced by token count (ragged corpora): block r ends at the first
    row boundary at or after r/world of the total tokens.  Returns ``[(begin, end)] * world``."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    total = int(offsets[-1] - offsets[0])
    cuts = [0]
    for r in range(1, world_size):
        target = offsets[0] + (total * r) // world_size
        cut = int(np.searchsorted(offsets, target, side="left"))
        cuts.append(min(max(cut, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def _group(group):
    return group if group is not None else rendezvous.Group(0, 1)


def gather_counts(n_local: int, group=None) -> List[int]:
    """Rows held by every rank, in rank order."""
    g = _group(group)
    return [int(np.frombuffer(p, dtype=np.int64)[0]) for p in g.allgather(np.int64(n_local).tobytes())]


def allgather_signatures(local: np.ndarray, group=None, counts: Optional[Sequence[int]] = None) -> np.ndarray:
    """Assemble the full signature matrix from per-rank row shards (host arrays in, host array out):
    the CPU stand-in for the RCCL all-gather.  ``local`` is this rank's ``[n_r, K]`` uint64 shard;
    values are < 2**32, so shards travel as uint32 and are widened on arrival.  Shards may be unequal."""
    g = _group(group)
    local = np.ascontiguousarray(local, dtype=np.uint64)
    if g.world == 1:
        return local
    if local.ndim != 2:
        raise ValueError("a signature shard is a 2-D array")
    if np.any(local > np.uint64(0xFFFFFFFF)):
        raise ValueError("signature values >= 2**32 cannot use the uint32 wire format")
    k = local.shape[1]
    parts = g.allgather(local.astype(np.uint32).tobytes())
    mats = [np.frombuffer(p, dtype=np.uint32).reshape(-1, k) for p in parts]
    if counts is not None and [m.shape[0] for m in mats] != [int(c) for c in counts]:
        raise ValueError("shard sizes differ from the counts given")
    return np.concatenate(mats, axis=0).astype(np.uint64)


def communicator(ctx, group):
    """The RCCL communicator of this rank's context for ``group`` (created once: rank 0 makes the
    128-byte id, the group's broadcast hands it out).  It lives on the group object itself -- a table keyed by
    ``id()`` would hand a recycled address the communicator of a dead group -- and ``group.close()`` destroys it."""
    from datasketch_amd import _native

    comms = getattr(group, "_mhx_comms", None)
    if comms is None:
        # id(context) -> communicator.  The communicator holds its context (so the id cannot be recycled while the entry
        # lives) and is destroyed by whichever closes first: group.close() walks this table, Context.close() its own list
        # of communicators -- a closed one has handle None and is made afresh here.
        comms = {}
        try:
            group._mhx_comms = comms
        except AttributeError:  # a foreign group object without a __dict__: no caching
            pass
    comm = comms.get(id(ctx))
    if comm is not None and comm.ctx is not ctx:
        comm = None
    if comm is None or comm.handle is None:
        uid = _native.Communicator.unique_id() if group.rank == 0 else b""
        uid = group.allgather(uid)[0]
        comm = _native.Communicator(ctx, uid, group.rank, group.world)
        comms[id(ctx)] = comm
    return comm


class GatheredSignatures:
    """The all-gathered signature matrix, resident on this rank's GPU: ``buffer`` holds ``[rows, k]`` uint32
    (row-major, rank order).  Feed ``buffer.ptr`` to the ``*_dev`` entry points that take ``sig_dtype =
    MHX_U32`` (b-bit packing, band digests, LSH sort), or :meth:`to_host` for a numpy matrix.
    ``transport`` names how the shards travelled: ``"rccl"``, ``"host-shm"``, ``"host-tcp"`` or ``"none"`` (one rank)."""

    def __init__(self, ctx, buffer, rows: int, k: int, transport: str = "none"):
        self.ctx, self.buffer, self.rows, self.k, self.transport = ctx, buffer, int(rows), int(k), transport

    def to_host(self, dtype=np.uint64) -> np.ndarray:
        self.ctx.synchronize()
        m = self.buffer.download((self.rows, self.k), np.uint32)
        return m if np.dtype(dtype) == np.uint32 else m.astype(dtype)


TRANSPORTS = ("rccl", "host")


def allgather_transport(transport: Optional[str] = None) -> str:
    """``"rccl"`` (the default: RCCL over xGMI, one GPU per rank) or ``"host"`` -- an explicit opt-in, by argument or
    ``MHX_ALLGATHER_TRANSPORT=host``, never chosen silently: the shards are staged through host memory (D2H, shared
    memory or the TCP rendezvous, H2D), which is the one transport that lets several ranks share a GPU (RCCL refuses
    two ranks on one device) and so lets a 1-GPU box run every line of the N > 1 path."""
    t = (transport or os.environ.get("MHX_ALLGATHER_TRANSPORT") or "rccl").strip().lower()
    if t not in TRANSPORTS:
        raise ValueError(f"all-gather transport {t!r}: one of {TRANSPORTS}")
    return t


_HOST_PIECE = 256 << 20  # bytes per rank and TCP frame of the host transport (frames are bounded: rendezvous.MAX_FRAME)
_FORCE_TCP = False        # tests: take the socket path although every rank is on this node


def _node_identity() -> bytes:
    boot = ""
    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            boot = f.read().strip()
    except OSError:
        pass
    import socket

    return f"{socket.gethostname()}|{boot}|{os.stat('/dev/shm').st_dev if os.path.isdir('/dev/shm') else -1}".encode()


def _host_exchange(ctx, d_local, local_bytes: int, wanted, d_dst, group, what: str = "all-gather") -> str:
    """The host-staged transport, for any exchange in which every rank publishes ONE block (``local_bytes`` bytes at
    ``d_local``) and takes ranges of the other ranks' blocks: ``wanted`` = ``[(rank q, offset in q's block, bytes, offset
    in d_dst)]``.  The block leaves the device once -- into a /dev/shm file when every rank is on this node, else over the
    rendezvous sockets in bounded pieces -- and every wanted range is uploaded straight to its place in ``d_dst`` (a range
    of this rank's own block is a device copy).  Blocking; returns ``"host-shm"`` or ``"host-tcp"``.

    A rank that fails -- staging its block, opening a peer's file, uploading -- says so in the status all-gather that
    closes each phase, so every rank raises instead of some waiting at a barrier the failed one never reaches."""
    import mmap

    world, rank = group.world, group.rank
    mine = int(local_bytes)
    ctx.synchronize()  # the block is complete before it is read
    hello = group.allgather(_node_identity() + b"|" + str(mine).encode())
    sizes = [int(h.rsplit(b"|", 1)[1]) for h in hello]
    same_node = len({h.rsplit(b"|", 1)[0] for h in hello}) == 1 and os.path.isdir("/dev/shm") and not _FORCE_TCP
    for q, src, size, dst in wanted:
        if size < 0 or src < 0 or src + size > sizes[q]:
            raise ValueError(f"host {what}: bytes [{src}, {src + size}) of rank {q}'s block of {sizes[q]} bytes")
        if q == rank and size:
            ctx.copy_dev(d_dst.ptr + int(dst), d_local.ptr + int(src), int(size))

    def agree(err: bytes, payload: bytes = b""):
        """Status all-gather: (payloads of all ranks) when nobody failed, else OSError on EVERY rank."""
        got = group.allgather(b"E" + err if err else b"P" + payload)
        bad = [f"rank {q}: " + g[1:].decode("utf-8", "replace") for q, g in enumerate(got) if g[:1] == b"E"]
        if bad:
            raise OSError(f"host {what}: " + "; ".join(bad))
        return [g[1:] for g in got]

    if same_node:
        path, fd, mem, err = b"", -1, None, b""
        try:
            if mine:
                name = f"/dev/shm/mhx_gather_{os.getuid()}_{os.getpid()}_{secrets.token_hex(8)}"
                fd = os.open(name, os.O_RDWR | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
                path = name.encode()
                os.posix_fallocate(fd, 0, mine)  # reserves the pages: a full tmpfs fails HERE (ENOSPC), not with SIGBUS in the copy
                mem = mmap.mmap(fd, mine)
                staged = np.frombuffer(mem, dtype=np.uint8)
                try:
                    d_local.download_into(staged)
                finally:
                    del staged  # (an exported buffer would keep mem.close() from working)
        except Exception as e:  # noqa: BLE001 -- reported to every rank below
            err = repr(e).encode()
        try:
            names = agree(err, path)  # (also the barrier: every file is complete)
            err = b""
            views = {}
            try:
                for q, src, size, dst in wanted:
                    if q == rank or size == 0:
                        continue
                    if q not in views:
                        qfd = os.open(names[q].decode(), os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                        try:
                            views[q] = mmap.mmap(qfd, sizes[q], prot=mmap.PROT_READ)
                        finally:
                            os.close(qfd)
                    piece = np.frombuffer(views[q], dtype=np.uint8, count=int(size), offset=int(src))
                    try:
                        d_dst.upload(piece, offset=int(dst))
                    finally:
                        del piece  # before the view closes (BufferError otherwise, which would hide the real error)
                ctx.synchronize()
            except Exception as e:  # noqa: BLE001
                err = repr(e).encode()
            finally:
                for v in views.values():
                    v.close()
            agree(err)  # every rank has read every file -- or every rank learns that one could not
        finally:
            if mem is not None:
                mem.close()
            if fd >= 0:
                os.close(fd)
            if path:
                try:
                    os.unlink(path.decode())
                except OSError:
                    pass
        return "host-shm"
    step = max(1, min(_HOST_PIECE, (rendezvous.MAX_FRAME // 2) // world))  # rank 0 answers with all the pieces in one frame
    for lo in range(0, max(sizes) if sizes else 0, step):
        n_mine = max(0, min(step, mine - lo))
        piece = d_local.download((n_mine,), np.uint8, offset=lo).tobytes() if n_mine else b""
        parts = group.allgather(piece)
        for q, src, size, dst in wanted:
            if q == rank or size == 0:
                continue
            a, b = max(src, lo), min(src + size, lo + len(parts[q]))  # the part of this range inside the piece
            if a < b:
                d_dst.upload(np.frombuffer(parts[q], dtype=np.uint8, count=b - a, offset=a - lo), offset=int(dst) + (a - src))
    ctx.synchronize()
    return "host-tcp"


def _allgather_host(ctx, d_local, d_all, counts: Sequence[int], row_bytes: int, group) -> str:
    """All-gather of row shards over the host-staged transport: rank q's whole block lands at its rows of ``d_all``
    (unequal shards need no padding).  Blocking; returns ``"host-shm"`` or ``"host-tcp"``."""
    sizes = [int(c) * row_bytes for c in counts]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    wanted = [(q, 0, sizes[q], int(offs[q])) for q in range(group.world)]
    return _host_exchange(ctx, d_local, sizes[group.rank], wanted, d_all, group, "all-gather")


def allgather_signatures_dev(ctx, d_local, rows: int, k: int, counts: Sequence[int], group,
                             transport: Optional[str] = None) -> GatheredSignatures:
    """Device path of :func:`allgather_signatures`: ``d_local`` is a DeviceBuffer holding this rank's
    ``[rows, k]`` **uint32** shard (the compact output type of ``mhx_minhash_bulk_dev``).  The result is the
    ``[sum(counts), k]`` matrix in rank order on this rank's GPU; nothing comes back to the host.

    ``transport="rccl"`` (default): one ``ncclAllGather`` when the shards are equal, else one grouped launch of
    per-root broadcasts (``mhx_comm_allgatherv_dev``) that writes every shard at its final place -- no padded
    buffer, no squeeze copies, no host synchronisation; enqueued on the context's stream.
    ``transport="host"``: see :func:`allgather_transport` (explicit opt-in; blocking)."""
    world = len(counts)
    counts = [int(c) for c in counts]
    total = int(sum(counts))
    if world != group.world:
        raise ValueError("counts has one entry per rank")
    if rows != counts[group.rank]:
        raise ValueError("rows differs from this rank's entry of counts")
    transport = allgather_transport(transport)
    row_bytes = k * 4
    d_all = ctx.alloc(max(1, total * row_bytes))
    if transport == "host":
        used = _allgather_host(ctx, d_local, d_all, counts, row_bytes, group)
        return GatheredSignatures(ctx, d_all, total, k, used)
    comm = communicator(ctx, group)
    if all(c == counts[0] for c in counts):
        comm.allgather_dev(d_local.ptr, d_all.ptr, counts[0] * row_bytes)
    else:
        sizes = [c * row_bytes for c in counts]
        comm.allgatherv_dev(d_local.ptr, d_all.ptr, [sum(sizes[:q]) for q in range(world)], sizes)
    return GatheredSignatures(ctx, d_all, total, k, "rccl")


def shard_csr(values, offsets, world_size: int, rank: int, balance: str = "tokens"):
    """This rank's part of a ragged corpus held as one CSR pair (``values`` 1-D, ``offsets`` int64 ``[n+1]``): contiguous
    rows, balanced by token count (``balance="tokens"``: :func:`shard_by_tokens`, SURVEY 8e "balance by nnz") or by row
    count (``"rows"``).  Returns ``(local_values, local_offsets, (begin, end))`` with the offsets rebased to 0."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    if balance == "tokens":
        begin, end = shard_by_tokens(offsets, world_size)[rank]
    elif balance == "rows":
        begin, end = shard_rows(n, world_size, rank)
    else:
        raise ValueError("balance is 'tokens' or 'rows'")
    lo, hi = int(offsets[begin]), int(offsets[end])
    return np.asarray(values)[lo:hi], offsets[begin:end + 1] - offsets[begin], (begin, end)


def bulk_signatures_sharded(local_tokens, *, num_perm: int, seed: int = 1, gpu_mode: str = "always", group=None,
                            counts: Optional[Sequence[int]] = None, keep_on_device: bool = False,
                            transport: Optional[str] = None):
    """Config-3 shape: every rank hashes ITS OWN rows and the shards are all-gathered; every rank gets the full
    ``[N, K]`` matrix in rank order.  ``local_tokens`` is this rank's shard of pre-hashed tokens (uint32 or uint64):

    * a dense ``[n_r, T]`` array, or
    * a ragged shard as a CSR pair ``(values, offsets)`` -- ``values`` 1-D, ``offsets`` int64 ``[n_r + 1]`` starting at
      0 (the generator takes arbitrary iterables, ref: minhash.py:491-522; :func:`shard_csr` cuts a corpus into such
      shards balanced by token count), or
    * a callable returning either (so that a rank only ever materialises its own shard).

    ``counts`` = rows per rank (gathered over the group when not given).  With a GPU the shard stays on the device
    from the kernel to the RCCL all-gather (uint32 on the wire).  ``keep_on_device=True`` returns a
    :class:`GatheredSignatures` (for the pack / digest / sort chain) instead of a host uint64 matrix.  ``transport``:
    see :func:`allgather_transport` (``"host"`` is the explicit opt-in that lets several ranks share one GPU).
    ``gpu_mode='disable'`` is the numpy path with the host stand-in collective."""
    from datasketch_amd import _native
    from datasketch_amd.hashfunc import prehashed
    from datasketch_amd.minhash import MinHash

    g = _group(group)
    shard = local_tokens() if callable(local_tokens) else local_tokens
    offsets = None
    if isinstance(shard, tuple):
        if len(shard) != 2:
            raise ValueError("a ragged shard is a (values, offsets) pair")
        values, offsets = np.asarray(shard[0]), np.ascontiguousarray(shard[1], dtype=np.int64)
        if values.ndim != 1 or offsets.ndim != 1 or offsets.size < 1:
            raise ValueError("a ragged shard is (1-D values, 1-D offsets of rows + 1 entries)")
        if offsets[0] != 0 or offsets[-1] != values.size or np.any(np.diff(offsets) < 0):
            raise ValueError("offsets must start at 0, end at len(values) and never decrease")
        shard, n_local = values, offsets.size - 1
    else:
        shard = np.asarray(shard)
        if shard.ndim != 2:
            raise ValueError("local_tokens must be a dense [rows, tokens] array or a (values, offsets) pair")
        n_local = shard.shape[0]
    if counts is None:
        counts = gather_counts(n_local, g)
    counts = [int(c) for c in counts]
    if counts[g.rank] != n_local:
        raise ValueError("counts[rank] differs from the number of local rows")
    use_gpu = gpu_mode == "always" or (gpu_mode == "detect" and _native.gpu_detected())
    if not use_gpu:
        if keep_on_device:
            raise ValueError("keep_on_device needs the GPU path")
        tokens = shard if offsets is None else (shard, offsets)
        local = MinHash.bulk_signatures(tokens, num_perm=num_perm, seed=seed, hashfunc=prehashed, gpu_mode=gpu_mode)
        return allgather_signatures(local, group=g, counts=counts)
    ctx = _native.context()
    proto = MinHash(num_perm=num_perm, seed=seed, hashfunc=prehashed, gpu_mode=gpu_mode)
    if shard.dtype != np.uint32:
        shard = np.ascontiguousarray(shard, dtype=np.uint64)
    shard = np.ascontiguousarray(shard)
    tok_code = _native.MHX_U32 if shard.dtype == np.uint32 else _native.MHX_U64
    d_tok = ctx.to_device(shard)
    d_out = ctx.alloc(max(1, n_local * num_perm * 4))
    if offsets is None:
        ctx.minhash_bulk_dev(proto.permutations, d_tok.ptr, tok_code, None, shard.shape[1], n_local, shard.size, None, 0, d_out.ptr, _native.MHX_U32)
    elif n_local:
        d_off = ctx.to_device(offsets)
        ctx.minhash_bulk_dev(proto.permutations, d_tok.ptr, tok_code, d_off.ptr, 0, n_local, shard.size, None, 0, d_out.ptr, _native.MHX_U32)
        ctx.synchronize()  # d_off may go out of scope
    if g.world == 1:
        gathered = GatheredSignatures(ctx, d_out, n_local, num_perm)
    else:
        gathered = allgather_signatures_dev(ctx, d_out, n_local, num_perm, counts, g, transport=transport)
    return gathered if keep_on_device else gathered.to_host(np.uint64)


# ---- the index partitioned by band: exchange digests, not signatures -------------------------------------------------
# The reference's MinHashLSH is one independent hashtable per band (ref: datasketch/lsh.py:199 hashtables, :326-347
# _insert: band i's key goes into table i and nowhere else).  So the natural shard of the INDEX is by band, while the
# shard of the HASHING is by row: rank p digests its own rows band-major ([bands, n_p] uint64) and rank q, which owns the
# tables of bands [lo_q, hi_q), receives those bands' digests of every rank's rows -- [hi_q - lo_q, N] uint64, 8 bytes
# per (row, band) it buckets -- instead of the whole uint32 signature matrix (4*K bytes per row from every peer).  At
# 10M rows x 256 permutations x 32 bands on 8 ranks: 0.28 GB received per GPU instead of 8.96 GB.


def band_partition(bands: int, world_size: int) -> List[Tuple[int, int]]:
    """Bands ``[lo, hi)`` whose hashtables rank q builds; contiguous, sizes differ by at most one (a rank beyond the
    number of bands gets none)."""
    return [(q * bands // world_size, (q + 1) * bands // world_size) for q in range(world_size)]


class BandShard:
    """This rank's part of the index after the by-band exchange: ``buffer`` holds ``[hi_band - lo_band, rows]`` uint64
    band digests, band-major, of ALL ranks' rows in rank order (row numbers are global) -- what
    ``mhx_lsh_sort_digests_layout_dev(..., bands = hi - lo, MHX_BAND_MAJOR)`` buckets.  ``transport`` as in
    :class:`GatheredSignatures`; ``bytes_received`` = what came from other ranks."""

    def __init__(self, ctx, buffer, rows: int, lo_band: int, hi_band: int, transport: str, bytes_received: int):
        self.ctx, self.buffer, self.rows, self.lo_band, self.hi_band = ctx, buffer, int(rows), int(lo_band), int(hi_band)
        self.transport, self.bytes_received = transport, int(bytes_received)

    @property
    def bands(self) -> int:
        return self.hi_band - self.lo_band

    def to_host(self) -> np.ndarray:
        self.ctx.synchronize()
        return self.buffer.download((self.bands, self.rows), np.uint64)


def _band_runs(counts: Sequence[int], bands: int, rank: int):
    """The runs of the by-band exchange seen from ``rank``: (sends, recvs), each ``[(peer, offset, bytes)]``.  Rank p's
    block is ``[bands, n_p]`` uint64; the run (p -> q, band j) is row j of it and lands at ``[j - lo_q][begin_p ...)`` of
    q's ``[hi_q - lo_q, N]`` matrix.  Both ends list the runs of a pair in band order."""
    world = len(counts)
    part = band_partition(bands, world)
    begins = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    total = int(begins[-1])
    n_mine = int(counts[rank])
    sends = [(q, j * n_mine * 8, n_mine * 8) for q in range(world) for j in range(*part[q])]
    lo, hi = part[rank]
    recvs = [(p, ((j - lo) * total + int(begins[p])) * 8, int(counts[p]) * 8) for p in range(world) for j in range(lo, hi)]
    return sends, recvs


def exchange_band_digests(local: np.ndarray, group=None, counts: Optional[Sequence[int]] = None) -> np.ndarray:
    """Host stand-in of :func:`exchange_band_digests_dev` (numpy in, numpy out; what the CPU tests run): ``local`` is
    this rank's ``[bands, n_r]`` uint64 band-major digests; returns ``[hi - lo, N]`` for this rank's bands."""
    g = _group(group)
    local = np.ascontiguousarray(local, dtype=np.uint64)
    if local.ndim != 2:
        raise ValueError("band-major digests are a 2-D [bands, rows] array")
    bands = local.shape[0]
    if counts is None:
        counts = gather_counts(local.shape[1], g)
    counts = [int(c) for c in counts]
    if len(counts) != g.world or counts[g.rank] != local.shape[1]:
        raise ValueError("counts has one entry per rank and counts[rank] is the number of local rows")
    lo, hi = band_partition(bands, g.world)[g.rank]
    parts = g.allgather(local.tobytes())  # (the stand-in ships everything; the device path sends each peer its bands only)
    mats = [np.frombuffer(p, dtype=np.uint64).reshape(bands, c) for p, c in zip(parts, counts)]
    return np.ascontiguousarray(np.concatenate([m[lo:hi] for m in mats], axis=1))


def exchange_band_digests_dev(ctx, d_digests, n_local: int, bands: int, counts: Sequence[int], group,
                              transport: Optional[str] = None) -> BandShard:
    """The by-band exchange on the device.  ``d_digests``: DeviceBuffer with this rank's ``[bands, n_local]`` uint64
    band-major digests (``mhx_band_digests_layout_dev`` / ``mhx_bbit_pack_band_digests_dev`` with ``MHX_BAND_MAJOR``).
    Returns the :class:`BandShard` of this rank's bands over all ``sum(counts)`` rows.

    ``transport="rccl"`` (default): ONE grouped launch of ``ncclSend`` / ``ncclRecv`` (``mhx_comm_exchange_dev``) -- a run
    per (peer, band), every run written at its final place, enqueued on the context's stream, no host synchronisation.
    ``transport="host"``: the explicit opt-in of :func:`allgather_transport` (ranks sharing one GPU; blocking)."""
    counts = [int(c) for c in counts]
    world, rank = group.world, group.rank
    if len(counts) != world:
        raise ValueError("counts has one entry per rank")
    if int(n_local) != counts[rank]:
        raise ValueError("n_local differs from this rank's entry of counts")
    transport = allgather_transport(transport)
    total = int(sum(counts))
    lo, hi = band_partition(bands, world)[rank]
    d_out = ctx.alloc(max(1, (hi - lo) * total * 8))
    sends, recvs = _band_runs(counts, bands, rank)
    received = sum(size for p, _, size in recvs if p != rank)
    if world == 1:
        ctx.copy_dev(d_out.ptr, d_digests.ptr, bands * total * 8)
        return BandShard(ctx, d_out, total, lo, hi, "none", 0)
    if transport == "host":
        begins = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        wanted = [(p, j * counts[p] * 8, counts[p] * 8, ((j - lo) * total + int(begins[p])) * 8) for p in range(world) for j in range(lo, hi)]
        used = _host_exchange(ctx, d_digests, bands * counts[rank] * 8, wanted, d_out, group, "by-band exchange")
        return BandShard(ctx, d_out, total, lo, hi, used, received)
    communicator(ctx, group).exchange_dev(d_digests.ptr, d_out.ptr, sends, recvs)
    return BandShard(ctx, d_out, total, lo, hi, "rccl", received)


class ShardedIndex:
    """What :func:`lsh_index_sharded` leaves on this rank's GPU: ``blocks`` -- the b-bit blocks ``[n_local, num_blocks]``
    uint64 of this rank's OWN rows (``None`` without ``b``); ``digests`` -- the :class:`BandShard` of this rank's bands;
    ``sorted_digests`` / ``sorted_rows`` -- ``[bands_here, N]`` uint64 / uint32, every bucket of a band a run of equal
    digests, rows (global numbers) ascending inside a bucket; ``fused`` -- whether blocks and digests came from one read."""

    def __init__(self, blocks, digests: BandShard, sorted_digests, sorted_rows, fused: bool):
        self.blocks, self.digests, self.sorted_digests, self.sorted_rows, self.fused = blocks, digests, sorted_digests, sorted_rows, fused

    def to_host(self):
        s = self.digests
        s.ctx.synchronize()
        return (self.sorted_digests.download((s.bands, s.rows), np.uint64), self.sorted_rows.download((s.bands, s.rows), np.uint32))


def lsh_index_sharded(ctx, d_sig, sig_dtype: int, n_local: int, num_perm: int, bands: int, r: int, counts: Sequence[int], group,
                      b: Optional[int] = None, transport: Optional[str] = None, sort: bool = True) -> ShardedIndex:
    """Configs 3 and 5 across ranks without assembling the signature matrix: this rank's ``[n_local, num_perm]``
    signatures (device pointer ``d_sig``, ``MHX_U32`` or ``MHX_U64``) -> band digests of its rows, band-major -- with
    ``b`` given, in the same read as the b-bit blocks (ref: b_bit_minhash.py:78-101; one fused launch when the shape
    allows) -> by-band exchange -> bucketing of this rank's bands over all rows (ref: lsh.py:326-347)."""
    from datasketch_amd import _native

    lib = ctx.lib
    n_local = int(n_local)
    d_dig = ctx.alloc(max(1, bands * n_local * 8))
    d_blk, fused = None, False
    if b is not None:
        nb = ctypes_int32_blocks(lib, num_perm, b)
        d_blk = ctx.alloc(max(1, n_local * nb * 8))
        if n_local:
            fused = ctx.bbit_pack_band_digests_dev(d_sig, sig_dtype, n_local, num_perm, b, bands, r, d_blk.ptr, d_dig.ptr, _native.BAND_MAJOR)
    elif n_local:
        _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, d_sig, sig_dtype, n_local, num_perm, bands, r, _native.BAND_MAJOR, d_dig.ptr))
    shard = exchange_band_digests_dev(ctx, d_dig, n_local, bands, counts, group, transport=transport)
    d_sd = d_sr = None
    if sort and shard.bands > 0 and shard.rows > 0:
        d_sd, d_sr = ctx.alloc(shard.bands * shard.rows * 8), ctx.alloc(shard.bands * shard.rows * 4)
        _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, shard.buffer.ptr, shard.rows, shard.bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr))
    return ShardedIndex(d_blk, shard, d_sd, d_sr, fused)


def ctypes_int32_blocks(lib, num_perm: int, b: int) -> int:
    import ctypes

    from datasketch_amd import _native

    nb = ctypes.c_int32(0)
    _native.check(lib.mhx_bbit_num_blocks(int(num_perm), int(b), ctypes.byref(nb)))
    return int(nb.value)
