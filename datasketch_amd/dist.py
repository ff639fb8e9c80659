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


def _allgather_host(ctx, d_local, d_all, counts: Sequence[int], row_bytes: int, group) -> str:
    """The host-staged exchange: this rank's shard leaves the device once (into a /dev/shm file when every rank is on
    this node, else over the rendezvous sockets in bounded pieces) and every other rank's shard is uploaded straight
    to its place in ``d_all`` -- unequal shards need no padding.  Blocking; returns ``"host-shm"`` or ``"host-tcp"``."""
    import mmap

    world, rank = group.world, group.rank
    offs = np.concatenate([[0], np.cumsum([int(c) * row_bytes for c in counts])]).astype(np.int64)
    mine = int(counts[rank]) * row_bytes
    ctx.synchronize()  # the shard is complete before it is read
    same_node = len(set(group.allgather(_node_identity()))) == 1 and os.path.isdir("/dev/shm") and not _FORCE_TCP
    if mine:
        ctx.copy_dev(d_all.ptr + int(offs[rank]), d_local.ptr, mine)
    if same_node:
        path, fd, mem = b"", -1, None
        err = b""
        try:
            if mine:
                name = f"/dev/shm/mhx_gather_{os.getuid()}_{os.getpid()}_{secrets.token_hex(8)}"
                fd = os.open(name, os.O_RDWR | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
                path = name.encode()
                os.posix_fallocate(fd, 0, mine)  # reserves the pages: a full tmpfs fails HERE (ENOSPC), not with SIGBUS in the copy
                mem = mmap.mmap(fd, mine)
                d_local.download_into(np.frombuffer(mem, dtype=np.uint8))
        except OSError as e:
            err = repr(e).encode()
        try:
            names = group.allgather(b"E" + err if err else b"P" + path)  # (also the barrier: every file is complete)
            bad = [n[1:].decode("utf-8", "replace") for n in names if n[:1] == b"E"]
            if bad:
                raise OSError(f"host all-gather: a rank could not stage its shard in /dev/shm: {bad}")
            for q in range(world):
                size = int(counts[q]) * row_bytes
                if q == rank or size == 0:
                    continue
                qfd = os.open(names[q][1:].decode(), os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                try:
                    with mmap.mmap(qfd, size, prot=mmap.PROT_READ) as view:
                        d_all.upload(np.frombuffer(view, dtype=np.uint8), offset=int(offs[q]))
                finally:
                    os.close(qfd)
            ctx.synchronize()
            group.barrier()  # every rank has read every file
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
    sizes = [int(c) * row_bytes for c in counts]
    step = max(1, min(_HOST_PIECE, (rendezvous.MAX_FRAME // 2) // world))  # rank 0 answers with all the pieces in one frame
    for lo in range(0, max(sizes), step):
        n_mine = max(0, min(step, mine - lo))
        piece = d_local.download((n_mine,), np.uint8, offset=lo).tobytes() if n_mine else b""
        parts = group.allgather(piece)
        for q in range(world):
            if q != rank and parts[q]:
                d_all.upload(np.frombuffer(parts[q], dtype=np.uint8), offset=int(offs[q]) + lo)
    ctx.synchronize()
    return "host-tcp"


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


def bulk_signatures_sharded(local_tokens, *, num_perm: int, seed: int = 1, gpu_mode: str = "always", group=None,
                            counts: Optional[Sequence[int]] = None, keep_on_device: bool = False,
                            transport: Optional[str] = None):
    """Config-3 shape: every rank hashes ITS OWN rows -- ``local_tokens`` is this rank's dense ``[n_r, T]``
    array of pre-hashed tokens (uint32 or uint64), or a callable returning it (so that a rank only ever
    materialises its own shard) -- and the shards are all-gathered; every rank gets the full ``[N, K]``
    matrix in rank order.  ``counts`` = rows per rank (gathered over the group when not given).

    With a GPU the shard stays on the device from the kernel to the RCCL all-gather (uint32 on the wire).
    ``keep_on_device=True`` returns a :class:`GatheredSignatures` (for the pack / digest / sort chain) instead
    of a host uint64 matrix.  ``transport``: see :func:`allgather_transport` (``"host"`` is the explicit opt-in
    that lets several ranks share one GPU).  ``gpu_mode='disable'`` is the numpy path with the host stand-in collective."""
    from datasketch_amd import _native
    from datasketch_amd.hashfunc import prehashed
    from datasketch_amd.minhash import MinHash

    g = _group(group)
    shard = local_tokens() if callable(local_tokens) else local_tokens
    shard = np.asarray(shard)
    if shard.ndim != 2:
        raise ValueError("local_tokens must be a dense [rows, tokens] array")
    if counts is None:
        counts = gather_counts(shard.shape[0], g)
    counts = [int(c) for c in counts]
    if counts[g.rank] != shard.shape[0]:
        raise ValueError("counts[rank] differs from the number of local rows")
    use_gpu = gpu_mode == "always" or (gpu_mode == "detect" and _native.gpu_detected())
    if not use_gpu:
        if keep_on_device:
            raise ValueError("keep_on_device needs the GPU path")
        local = MinHash.bulk_signatures(shard, num_perm=num_perm, seed=seed, hashfunc=prehashed, gpu_mode=gpu_mode)
        return allgather_signatures(local, group=g, counts=counts)
    ctx = _native.context()
    proto = MinHash(num_perm=num_perm, seed=seed, hashfunc=prehashed, gpu_mode=gpu_mode)
    if shard.dtype != np.uint32:
        shard = np.ascontiguousarray(shard, dtype=np.uint64)
    shard = np.ascontiguousarray(shard)
    tok_code = _native.MHX_U32 if shard.dtype == np.uint32 else _native.MHX_U64
    n_local, t = shard.shape
    d_tok = ctx.to_device(shard)
    d_out = ctx.alloc(max(1, n_local * num_perm * 4))
    ctx.minhash_bulk_dev(proto.permutations, d_tok.ptr, tok_code, None, t, n_local, shard.size, None, 0, d_out.ptr, _native.MHX_U32)
    if g.world == 1:
        gathered = GatheredSignatures(ctx, d_out, n_local, num_perm)
    else:
        gathered = allgather_signatures_dev(ctx, d_out, n_local, num_perm, counts, g, transport=transport)
    return gathered if keep_on_device else gathered.to_host(np.uint64)
