"""N>1 path on CPU: world_size-2 gloo processes shard the corpus, hash their rows, all-gather."""
import os
import socket
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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from datasketch_amd.dist import allgather_signatures, bulk_signatures_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tokens = np.random.RandomState(5).randint(0, 2**32, (101, 33), dtype=np.uint64)
        full = bulk_signatures_sharded(tokens, num_perm=24, seed=3, gpu_mode="disable")
        # unequal shards without precomputed counts
        mine = full[: 10 + 5 * rank]
        glued = allgather_signatures(mine)
        q.put((rank, full, glued.shape[0]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_bulk_equals_single_process():
    import torch.multiprocessing as mp

    from datasketch_amd import MinHash, prehashed

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tokens = np.random.RandomState(5).randint(0, 2**32, (101, 33), dtype=np.uint64)
    want = MinHash.bulk_signatures(tokens, num_perm=24, seed=3, hashfunc=prehashed)
    for _rank, full, glued_rows in results:
        assert np.array_equal(full, want)
        assert glued_rows == 10 + 15
