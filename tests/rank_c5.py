"""One rank of round 6's N > 1 chains, run as a subprocess by tests/test_gpu_round6.py (N ranks may share one GPU: the exchange
then takes the explicit host-staged transport, or the RCCL binding over the stand-in library):

  byband  shard of the corpus -> K signatures (uint32, on the device, never gathered) -> b = 1 blocks + band-major digests of the
          rank's OWN rows in one read -> by-band exchange (rank q receives [its bands, N] digests) -> bucketing of its bands
          (configs 3 and 5 with the index partitioned by band: dist.lsh_index_sharded)
  ragged  a heavy-tailed ragged corpus cut by token count -> dist.bulk_signatures_sharded((values, offsets)) -> gathered matrix

Every rank writes its arrays; nothing here checks anything: the parent compares with the single-process results and the oracle."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ragged_corpus(n, seed=91, cap=3000):
    """Heavy-tailed set lengths (Pareto, capped), some empty sets; tokens 32-bit with a few wide ones."""
    rng = np.random.RandomState(seed)
    lens = np.minimum(cap, (rng.pareto(1.1, n) * 12).astype(np.int64))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.randint(0, 2**32, int(offsets[-1]), dtype=np.uint64)
    values[:: 997] |= np.uint64(1) << np.uint64(40)
    return values, offsets


def main():
    mode, out = sys.argv[1], sys.argv[-1]
    from datasketch_amd import MinHash, _native, dist, rendezvous
    import rank_c3

    group = rendezvous.from_env(timeout=180)
    ctx = _native.context(0)
    transport = os.environ.get("MHX_TEST_TRANSPORT", "host")
    if mode == "byband":
        n, t, k, bands, r = (int(v) for v in sys.argv[2:7])
        scheme = sys.argv[7]
        counts = rank_c3.split(n, group.world, scheme)
        begin = sum(counts[: group.rank])
        n_local = counts[group.rank]
        tokens = rank_c3.corpus(n, t)[begin: begin + n_local]
        perms = MinHash(num_perm=k, seed=3, hashfunc=lambda x: x).permutations
        d_tok, d_sig = ctx.to_device(tokens), ctx.alloc(max(1, n_local * k * 4))
        ctx.minhash_bulk_dev(perms, d_tok.ptr, _native.MHX_U64, None, t, n_local, tokens.size, None, 0, d_sig.ptr, _native.MHX_U32)
        idx = dist.lsh_index_sharded(ctx, d_sig.ptr, _native.MHX_U32, n_local, k, bands, r, counts, group, b=1, transport=transport)
        ctx.synchronize()
        shard = idx.digests
        arrays = {"blocks": idx.blocks.download((n_local, -(-k // 64)), np.uint64), "digests": shard.to_host()}
        if idx.sorted_digests is not None:
            arrays["sorted_digests"], arrays["sorted_rows"] = idx.to_host()
        rec = {"rank": group.rank, "world": group.world, "counts": counts, "transport": shard.transport, "fused": bool(idx.fused),
               "lo_band": shard.lo_band, "hi_band": shard.hi_band, "bytes_received": shard.bytes_received, "rows": shard.rows}
    else:
        n, k = int(sys.argv[2]), int(sys.argv[3])
        values, offsets = ragged_corpus(n)
        v, o, (b, e) = dist.shard_csr(values, offsets, group.world, group.rank)
        got = dist.bulk_signatures_sharded((v, o), num_perm=k, seed=3, gpu_mode="always", group=group, keep_on_device=True, transport=transport)
        arrays = {"sig": got.to_host(np.uint32)} if group.rank in (0, group.world - 1) else {}
        rec = {"rank": group.rank, "world": group.world, "rows": [int(b), int(e)], "tokens": int(v.size), "transport": got.transport}
    with open(f"{out}.{group.rank}.json", "w") as f:
        json.dump(rec, f)
    np.savez(f"{out}.{group.rank}.npz", **arrays)
    group.barrier()
    group.close()
    os._exit(0)  # (no interpreter teardown with N HIP runtimes on one device: nothing left to do)


if __name__ == "__main__":
    main()
