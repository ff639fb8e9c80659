"""One rank of the config-3 chain, run as a subprocess by tests/test_gpu_round5.py (N ranks may share one GPU: the
all-gather then takes the explicit host-staged transport, datasketch_amd.dist.allgather_transport):

    shard of the corpus -> K signatures (uint32, on the device) -> all-gather -> b=1 blocks + band digests (one read)
    -> bucketing sort of the digests

Every rank writes the sha256 of each stage's result; rank 0 also writes the arrays themselves.  Nothing here checks
anything: the parent compares with the single-process results and the oracle."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def corpus(n, t, seed=77):
    return np.random.RandomState(seed).randint(0, 2**32, (n, t), dtype=np.uint64)


def split(n, world, scheme):
    """Rows per rank: 'equal' (n divisible by world) or 'unequal' (a ragged split, one rank with a single row)."""
    if scheme == "equal":
        assert n % world == 0
        return [n // world] * world
    w = np.arange(1, world + 1, dtype=np.float64) ** 1.5
    counts = np.maximum(1, np.floor(w / w.sum() * (n - world))).astype(np.int64)
    counts[0] = 1
    counts[-1] += n - counts.sum()
    return [int(c) for c in counts]


def main():
    n, t, k, bands, r = (int(v) for v in sys.argv[1:6])
    scheme, out = sys.argv[6], sys.argv[7]
    from datasketch_amd import _native, dist, rendezvous

    group = rendezvous.from_env(timeout=180)
    counts = split(n, group.world, scheme)
    begin = sum(counts[: group.rank])
    tokens = corpus(n, t)[begin: begin + counts[group.rank]]
    ctx = _native.context(0)
    lib = ctx.lib
    got = dist.bulk_signatures_sharded(tokens, num_perm=k, seed=3, gpu_mode="always", group=group, counts=counts,
                                       keep_on_device=True, transport=os.environ.get("MHX_TEST_TRANSPORT", "host"))
    nb = k // 64
    d_blk, d_dig = ctx.alloc(n * nb * 8), ctx.alloc(n * bands * 8)
    # the digests band-major ([bands, n]: what the bucketing reads with unit stride), the layout the chain runs on
    fused = ctx.bbit_pack_band_digests_dev(got.buffer.ptr, _native.MHX_U32, n, k, 1, bands, r, d_blk.ptr, d_dig.ptr, _native.BAND_MAJOR)
    d_sd, d_sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
    _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig.ptr, n, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr))
    ctx.synchronize()
    arrays = {"sig": got.to_host(np.uint32), "blocks": d_blk.download((n, nb), np.uint64),
              "digests": np.ascontiguousarray(d_dig.download((bands, n), np.uint64).T),
              "sorted_digests": d_sd.download((bands, n), np.uint64), "sorted_rows": d_sr.download((bands, n), np.uint32)}
    rec = {"rank": group.rank, "world": group.world, "counts": counts, "transport": got.transport, "fused": bool(fused),
           "sha": {name: hashlib.sha256(a.tobytes()).hexdigest() for name, a in arrays.items()}}
    with open(f"{out}.{group.rank}.json", "w") as f:
        json.dump(rec, f)
    if group.rank == 0:
        np.savez(f"{out}.0.npz", **arrays)
    group.barrier()
    group.close()
    os._exit(0)  # (no interpreter teardown with N HIP runtimes on one device: nothing left to do)


if __name__ == "__main__":
    main()
