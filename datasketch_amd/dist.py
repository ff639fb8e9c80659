"""Multi-GPU sharding of the bulk MinHash path: one process per GPU, corpus split by rows.

The path is embarrassingly parallel over sets (every rank needs only the 2*K permutation
parameters, regenerated from the seed), so the compute phase has NO collective.  The only
exchange step is the optional assembly of the full ``[N, K]`` signature matrix on every rank
(an all-gather of row shards) when one consumer -- e.g. ``MinHashLSH.insert`` -- needs it whole.

On the GPU the collective is RCCL over xGMI through libmhx's own binding (``mhx_comm_*`` in
include/mhx.h, :class:`datasketch_amd._native.Communicator`): device buffers in, device buffer
out, enqueued on the kernel's stream.  ``torch.distributed`` is plumbing only -- a ``gloo`` group
carries the 128-byte RCCL id and the shard sizes, and is the CPU stand-in for the collective in
the tests (PyTorch never touches the GPU in this package).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os

import numpy as np

# A program that imports this module means to use the RCCL path: libmhx then loads RCCL when the context
# is created (before a PyTorch-ROCm wheel brings its own ROCm runtime into the process) instead of lazily.
os.environ.setdefault("MHX_PRELOAD_RCCL", "1")


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


def allgather_signatures(local: np.ndarray, group=None, counts: Optional[Sequence[int]] = None) -> np.ndarray:
    """Assemble the full signature matrix from per-rank row shards (host arrays in, host array out).

    ``local`` is this rank's ``[n_r, K]`` uint64 shard.  Values are < 2**32, so shards travel as
    uint32 (half the bytes on the wire) and are widened on arrival.  Unequal shard sizes are
    padded to the largest (``counts`` = rows per rank; gathered first when not given).
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return np.asarray(local, dtype=np.uint64)
    world = dist.get_world_size(group)
    local = np.ascontiguousarray(local, dtype=np.uint64)
    k = local.shape[1]
    dev = torch.device("cpu")  # host arrays travel over gloo; device shards use allgather_signatures_dev
    if counts is None:
        c = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        all_c = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_c, c, group=group)
        counts = [int(x) for x in all_c.cpu().tolist()]
    if np.any(local > np.uint64(0xFFFFFFFF)):
        raise ValueError("signature values >= 2**32 cannot use the uint32 wire format")
    width = max(counts) if counts else 0
    send = np.zeros((width, k), dtype=np.int32)
    send[: local.shape[0]] = local.astype(np.uint32).view(np.int32)
    t_send = torch.from_numpy(send).to(dev)
    t_recv = torch.empty((world * width, k), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(t_recv, t_send, group=group)
    recv = t_recv.cpu().numpy().view(np.uint32).reshape(world, width, k)
    parts = [recv[r, : counts[r]] for r in range(world)]
    return np.concatenate(parts, axis=0).astype(np.uint64)


_COMM_CACHE = {}


def communicator(ctx, group=None):
    """The RCCL communicator of this rank's context for ``group`` (created once: rank 0 makes the
    id, a gloo broadcast hands it out)."""
    from datasketch_amd import _native

    _native.check(_native.load().mhx_comm_preload())  # RCCL before torch's own ROCm runtime enters the process
    import torch.distributed as dist

    key = (id(ctx), id(group))
    if key not in _COMM_CACHE:
        from datasketch_amd import _native

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [_native.Communicator.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        _COMM_CACHE[key] = _native.Communicator(ctx, box[0], rank, world)
    return _COMM_CACHE[key]


def allgather_signatures_dev(ctx, d_local, rows: int, k: int, counts: Sequence[int], group=None) -> np.ndarray:
    """Device path of :func:`allgather_signatures`: ``d_local`` is a DeviceBuffer holding this rank's
    ``[rows, k]`` **uint32** shard (the compact output type of ``mhx_minhash_bulk_dev``), padded
    capacity ``max(counts)`` rows.  RCCL gathers the padded shards; the host trims and widens."""
    world = len(counts)
    width = max(counts)
    comm = communicator(ctx, group)
    d_all = ctx.alloc(max(1, world * width * k * 4))
    comm.allgather_dev(d_local.ptr, d_all.ptr, width * k * 4)
    ctx.synchronize()
    recv = d_all.download((world, width, k), np.uint32)
    return np.concatenate([recv[r, : counts[r]] for r in range(world)], axis=0).astype(np.uint64)


def bulk_signatures_sharded(tokens, *, num_perm: int, seed: int = 1, gpu_mode: str = "always", group=None) -> np.ndarray:
    """Config-3 shape: every rank hashes its row block of ``tokens`` (a dense ``[N, T]`` array of
    pre-hashed tokens, identical on every rank) and the shards are all-gathered; every rank
    returns the full ``[N, K]`` matrix.  With a GPU the shard stays on the device from the kernel
    to the RCCL all-gather (uint32 on the wire); ``gpu_mode='disable'`` is the numpy + gloo path."""
    import torch.distributed as dist

    from datasketch_amd import _native
    from datasketch_amd.hashfunc import prehashed
    from datasketch_amd.minhash import MinHash

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, end = shard_rows(tokens.shape[0], world, rank)
    counts = [e - b for b, e in (shard_rows(tokens.shape[0], world, r) for r in range(world))]
    use_gpu = gpu_mode == "always" or (gpu_mode == "detect" and _native.gpu_available())
    if not use_gpu or world == 1:
        local = MinHash.bulk_signatures(tokens[begin:end], num_perm=num_perm, seed=seed, hashfunc=prehashed, gpu_mode=gpu_mode)
        return allgather_signatures(local, group=group, counts=counts)
    ctx = _native.context()
    proto = MinHash(num_perm=num_perm, seed=seed, hashfunc=prehashed, gpu_mode=gpu_mode)
    shard = np.ascontiguousarray(tokens[begin:end], dtype=np.uint64)
    t = shard.shape[1]
    d_tok = ctx.to_device(shard)
    d_out = ctx.alloc(max(1, max(counts) * num_perm * 4))
    ctx.minhash_bulk_dev(proto.permutations, d_tok.ptr, _native.MHX_U64, None, t, shard.shape[0], shard.size, None, 0,
                         d_out.ptr, _native.MHX_U32)
    return allgather_signatures_dev(ctx, d_out, shard.shape[0], num_perm, counts, group=group)
