"""bench.py at N > 1: the exchange steps (all-gather of signatures, by-band exchange of digests) and configs 3 / 5 across ranks."""
from __future__ import annotations

import time

import numpy as np

from benchmarks.common import XGMI_GBPS_PER_LINK, XGMI_LINKS, _download_rows, _timed

class AllGather:
    """All-gather of this rank's uint32 [n, k] signature shard into a [world, n, k] device buffer.  transport "rccl": RCCL
    through libmhx's own binding (mhx_comm_*), enqueued on the kernel's stream, the 128-byte id travelling over the
    rendezvous group; "host": the explicit host-staged stand-in (datasketch_amd.dist.allgather_transport), blocking."""

    def __init__(self, ctx, group, n, k, transport="rccl"):
        from datasketch_amd import dist

        self.ctx, self.group, self.n, self.k, self.transport = ctx, group, n, k, transport
        self.comm = dist.communicator(ctx, group) if transport == "rccl" else None
        self.shard_bytes = n * k * 4
        self.d_shard = ctx.alloc(self.shard_bytes)
        self.d_all = ctx.alloc(self.shard_bytes * group.world)
        self.used = transport

    def step(self, perms, d_tok, tok_dtype, t):
        from datasketch_amd import _native

        self.ctx.minhash_bulk_dev(perms, d_tok.ptr, tok_dtype, None, t, self.n, self.n * t, None, 0, self.d_shard.ptr, _native.MHX_U32)
        self.gather_only()

    def gather_only(self):
        if self.comm is not None:
            self.comm.allgather_dev(self.d_shard.ptr, self.d_all.ptr, self.shard_bytes)
        else:
            from datasketch_amd import dist

            self.used = dist._allgather_host(self.ctx, self.d_shard, self.d_all, [self.n] * self.group.world, self.k * 4, self.group)


def allgather_probe(ctx, group, gather, perms, d_tok, tok_dtype, n, t, k, reps=5, transport="rccl"):
    """The exchange step alone: every rank's uint32 shard to every rank.  Reports what RCCL itself says about the
    communicator and checks the gathered matrix: row 0 of every rank's block must be that rank's row 0."""
    from datasketch_amd import _native

    res = {"wire_dtype": "uint32", "bytes_per_rank": n * k * 4, "bytes_received_per_gpu": n * k * 4 * (group.world - 1)}
    err = None
    try:
        if gather is None:
            gather = AllGather(ctx, group, n, k, transport)
        ctx.minhash_bulk_dev(perms, d_tok.ptr, tok_dtype, None, t, n, n * t, None, 0, gather.d_shard.ptr, _native.MHX_U32)
        gather.gather_only()  # warm-up (RCCL builds its rings / channels on first use)
        ctx.synchronize()
    except Exception as e:  # noqa: BLE001
        err = repr(e)
    flags = group.allgather(b"probe-init:" + (err or "").encode())
    if any(f != b"probe-init:" for f in flags):
        res["error"] = [f.decode("utf-8", "replace") for f in flags]
        return res
    group.barrier()
    evs = [ctx.event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        gather.gather_only()
        evs[i + 1].record()
    ctx.synchronize()
    ms = [evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]
    all_ms = [float(np.frombuffer(p, dtype=np.float64)[0]) for p in group.allgather(np.float64(np.mean(ms)).tobytes())]
    res["transport"] = gather.used
    # what RCCL itself says about the communicator; the host-staged stand-in has none: its ranks are the rendezvous group's
    info = gather.comm.info() if gather.comm is not None else {"ranks_seen": -1, "rank": group.rank, "device": ctx.device, "rccl_version": None}
    seen = group.allgather_ints([info["ranks_seen"], info["rank"], info["device"]])
    my_row0 = gather.d_shard.download((k,), np.uint32)
    rows0 = [np.frombuffer(p, dtype=np.uint32) for p in group.allgather(my_row0.tobytes())]
    ok = True
    for r in range(group.world):
        got = gather.d_all.download((k,), np.uint32, offset=r * gather.shard_bytes)
        ok = ok and np.array_equal(got, rows0[r])
    oks = group.allgather(b"\x01" if ok else b"\x00")
    if not all(o == b"\x01" for o in oks):
        raise SystemExit("PARITY FAILURE: the all-gathered matrix does not hold every rank's shard")
    worst = max(all_ms)
    res.update({
        "ms": worst,
        "ms_per_rank": all_ms,
        "rccl_ranks_seen": [s[0] for s in seen] if gather.comm is not None else None,
        "rccl_rank_device": [[s[1], s[2]] for s in seen],
        "rccl_version": info["rccl_version"],
        "received_GBps_per_gpu": res["bytes_received_per_gpu"] / (worst * 1e-3) / 1e9,
        "xgmi_bound_GBps_per_gpu": XGMI_LINKS * XGMI_GBPS_PER_LINK,
        "signatures_per_s_with_allgather_after_compute": None,
        "checked": "row 0 of every rank's block on every rank",
    })
    return res


# ------------------------------------------------------------------------------------------------


def c3_sharded(ctx, group, args, d_tok, n_head, t, check_rows_idx, check_tokens, k=256, bands=32, r=8):
    """BASELINE.json configs[2] end to end across the ranks: every rank hashes ITS shard (num_perm = 256, uint32 out), the
    shards are all-gathered (RCCL over xGMI; `--allgather-transport host` is the labelled stand-in that lets ranks share
    a GPU), and the LSH index is built PARTITIONED BY BAND -- the reference keeps one independent hashtable per band
    (lsh.py:199,326-347), so rank q digests and buckets bands [q*bands/world, (q+1)*bands/world) of ALL rows.  Per-stage
    HIP-event / wall times on every rank; parity on every rank: its own rows of the gathered matrix and row 0 of every
    other rank's block against the numpy path, its bands' digests of those rows, its first band's order."""
    from datasketch_amd import _native, dist, lsh_bulk
    from datasketch_amd.hashfunc import prehashed
    from datasketch_amd.minhash import MinHash

    lib, world, rank = ctx.lib, group.world, group.rank
    n3 = args.c3_rows
    res = {"workload": f"config 3 sharded: {world} ranks x {n3} sets x {t} tokens, num_perm={k} -> all-gather (uint32) -> band-partitioned LSH bucketing ({bands} x {r})"}
    perms = MinHash(num_perm=k, seed=args.seed, hashfunc=lambda x: x).permutations
    if n3 > n_head:  # the headline's corpus + more rows of the same kind
        d3 = ctx.alloc(n3 * t * 8)
        ctx.copy_dev(d3.ptr, d_tok.ptr, n_head * t * 8)
        rng = np.random.RandomState(4242 + rank)
        for lo in range(n_head, n3, 50_000):
            d3.upload(rng.randint(0, 2**32, size=(min(50_000, n3 - lo), t), dtype=np.uint64), offset=lo * t * 8)
    else:
        d3 = d_tok
    d_shard = ctx.alloc(n3 * k * 4)
    sig_call = lambda: ctx.minhash_bulk_dev(perms, d3.ptr, _native.MHX_U64, None, t, n3, n3 * t, None, 0, d_shard.ptr, _native.MHX_U32)
    ms_sig = _timed(ctx, sig_call, reps=3, ramp=0.1)
    counts = [n3] * world
    transport = dist.allgather_transport(args.allgather_transport)
    err = b""
    gathered = None
    try:
        gathered = dist.allgather_signatures_dev(ctx, d_shard, n3, k, counts, group, transport=transport)  # warm-up: communicator, rings
        ctx.synchronize()
    except Exception as e:  # noqa: BLE001 -- e.g. RCCL refusing ranks that share a device
        err = repr(e).encode()
    flags = group.allgather(err)
    if any(flags):
        res["error"] = [f.decode("utf-8", "replace") for f in flags]
        return res
    del gathered
    group.barrier()
    w0 = time.perf_counter()
    gathered = dist.allgather_signatures_dev(ctx, d_shard, n3, k, counts, group, transport=transport)
    ctx.synchronize()
    ms_gather = 1e3 * (time.perf_counter() - w0)
    total = world * n3
    lo_band = rank * bands // world
    hi_band = (rank + 1) * bands // world
    nbl = hi_band - lo_band
    ms_dig = ms_sort = 0.0
    if nbl > 0:
        d_dig, d_sd, d_sr = ctx.alloc(total * nbl * 8), ctx.alloc(total * nbl * 8), ctx.alloc(total * nbl * 4)
        sig_at = gathered.buffer.ptr + lo_band * r * 4  # the band subset: same rows, same stride, first band of this rank
        BM = _native.BAND_MAJOR
        ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, sig_at, _native.MHX_U32, total, k, nbl, r, BM, d_dig.ptr)), reps=3, ramp=0.05)
        ms_sort = _timed(ctx, lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig.ptr, total, nbl, BM, d_sd.ptr, d_sr.ptr)), reps=3, ramp=0.05)
    # ---- parity on every rank
    ok, why = True, ""
    sel = check_rows_idx[check_rows_idx < min(n3, n_head)][:512]
    tok = check_tokens[: len(sel)]
    want = MinHash.bulk_signatures(tok, num_perm=k, seed=args.seed, hashfunc=prehashed, gpu_mode="disable")
    mine = _download_rows(gathered.buffer, rank * n3 + sel, k, np.uint32).astype(np.uint64)
    if not np.array_equal(mine, want):
        ok, why = False, "own rows of the gathered matrix differ from the numpy path"
    row0 = group.allgather(want[0].astype(np.uint32).tobytes() if len(sel) and sel[0] == 0 else b"")
    for q in range(world):
        if row0[q] and not np.array_equal(gathered.buffer.download((k,), np.uint32, offset=q * n3 * k * 4), np.frombuffer(row0[q], dtype=np.uint32)):
            ok, why = False, f"row 0 of rank {q}'s block is not that rank's row 0"
    if nbl > 0 and ok:
        wd = lsh_bulk.band_digests(want, bands, r, gpu_mode="disable")[:, lo_band:hi_band]
        dig_local = d_dig.download((nbl, total), np.uint64)
        if not np.array_equal(dig_local[:, rank * n3 + sel].T, wd):
            ok, why = False, "band digests differ from FNV-1a-64 of the reference's key bytes"
        sd, sr = d_sd.download((total,), np.uint64), d_sr.download((total,), np.uint32)
        col = dig_local[0]
        tie = sd[1:] == sd[:-1]
        if np.any(sd[1:] < sd[:-1]) or not np.array_equal(col[sr.astype(np.int64)], sd) or np.any(sr[1:][tie] <= sr[:-1][tie]):
            ok, why = False, "the rank's first band is not in (digest, row) order"
    oks = group.allgather(b"" if ok else why.encode())
    if any(oks):
        raise SystemExit("PARITY FAILURE (extra.c3_sharded): " + "; ".join(f"rank {q}: {o.decode()}" for q, o in enumerate(oks) if o))
    stages = {"signatures": ms_sig, "allgather": ms_gather, "band_digests": ms_dig, "bucketing": ms_sort}
    per_rank = {name: [float(np.frombuffer(p, dtype=np.float64)[0]) for p in group.allgather(np.float64(v).tobytes())] for name, v in stages.items()}
    worst = {name: max(v) for name, v in per_rank.items()}
    received = (world - 1) * n3 * k * 4
    ag = {"transport": gathered.transport, "wire_dtype": "uint32", "bytes_received_per_gpu": received, "ms": worst["allgather"],
          "GBps_per_gpu": received / (worst["allgather"] * 1e-3) / 1e9,
          "xgmi_bound_GBps_per_gpu": XGMI_LINKS * XGMI_GBPS_PER_LINK,
          "note": "received bytes / slowest rank's wall time (one blocking gather after a warm-up one); the bound is 7 links x 153 GB/s into every GPU "
                  "(SURVEY.md section 5); a host-staged transport crosses PCIe twice and says nothing about xGMI"}
    if gathered.transport == "rccl":
        info = dist.communicator(ctx, group).info()
        ag["rccl_ranks_seen"] = [s[0] for s in group.allgather_ints([info["ranks_seen"]])]
    res.update({
        "rows_total": total,
        "bands_per_rank": [(q + 1) * bands // world - q * bands // world for q in range(world)],
        "per_rank_ms": per_rank,
        "ms": worst,
        "pipeline_ms": sum(worst.values()),
        "signatures_per_s_end_to_end": total / (sum(worst.values()) * 1e-3),
        "allgather": ag,
        "parity": "every rank: up to 512 of its own rows of the gathered matrix and row 0 of every other rank's block vs the numpy path; its bands' digests of "
                  "those rows vs FNV-1a-64 of the reference's key bytes; its first band ascending, equal to the gathered digest column, rows ascending inside buckets",
    })
    # ---- the same index WITHOUT assembling the signature matrix: digests of the own rows, by-band exchange, bucketing (and config 5)
    del gathered
    by_band, c5 = by_band_chain(ctx, group, d_shard, n3, k, bands, r, counts, transport, ms_sig,
                                (d_dig, d_sd, d_sr) if nbl > 0 else None, want, sel, rank * n3)
    res["by_band"] = by_band
    if "error" not in by_band:
        res["by_band"]["against_allgather"] = {
            "bytes_received_per_gpu": [by_band["exchange"]["bytes_received_per_gpu"], received],
            "exchange_ms": [by_band["ms"]["exchange"], worst["allgather"]],
            "pipeline_ms": [by_band["pipeline_ms"], res["pipeline_ms"]],
            "order": "[by band, all-gather]"}
    res["c5_sharded"] = c5
    return res


def by_band_chain(ctx, group, d_shard, n3, k, bands, r, counts, transport, ms_sig, allgather_result, want, sel, row0):
    """Configs 3 and 5 across the ranks with the index partitioned by band and NO signature matrix assembled:
    every rank digests ITS rows band-major -- for config 5 in the same read that packs their b = 1 blocks
    (ref: b_bit_minhash.py:78-101) --, sends every peer the runs of that peer's bands (dist.exchange_band_digests_dev: one grouped
    launch of ncclSend / ncclRecv, 8 bytes per (row, band) the receiver buckets) and buckets its own bands over all rows
    (ref: lsh.py:199,326-347).  Returns (config 3's by-band record, config 5's record).  Parity on every rank: the exchanged digest
    matrix and both sorted outputs are byte-identical to what the all-gather path produced on this rank (allgather_result), the blocks
    of `sel` rows equal the numpy packing of the numpy path's signatures (`want`)."""
    from datasketch_amd import _native, dist
    from datasketch_amd.b_bit_minhash import pack_matrix

    lib, world, rank = ctx.lib, group.world, group.rank
    total = int(sum(counts))
    BM, U32 = _native.BAND_MAJOR, _native.MHX_U32
    lo_band, hi_band = dist.band_partition(bands, world)[rank]
    nbl = hi_band - lo_band
    d_loc = ctx.alloc(max(1, bands * n3 * 8))
    ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, d_shard.ptr, U32, n3, k, bands, r, BM, d_loc.ptr)), reps=3, ramp=0.05)
    err, shard = b"", None
    try:
        shard = dist.exchange_band_digests_dev(ctx, d_loc, n3, bands, counts, group, transport=transport)  # warm-up (the communicator stands already)
        ctx.synchronize()
    except Exception as e:  # noqa: BLE001
        err = repr(e).encode()
    flags = group.allgather(err)
    if any(flags):
        bad = {"error": [f.decode("utf-8", "replace") for f in flags]}
        return bad, dict(bad)
    del shard
    group.barrier()
    w0 = time.perf_counter()
    shard = dist.exchange_band_digests_dev(ctx, d_loc, n3, bands, counts, group, transport=transport)
    ctx.synchronize()
    ms_x = 1e3 * (time.perf_counter() - w0)
    ms_sort = 0.0
    ok, why = True, ""
    if nbl > 0:
        d_sd, d_sr = ctx.alloc(total * nbl * 8), ctx.alloc(total * nbl * 4)
        ms_sort = _timed(ctx, lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, shard.buffer.ptr, total, nbl, BM, d_sd.ptr, d_sr.ptr)), reps=3, ramp=0.05)
        if allgather_result is not None:
            a_dig, a_sd, a_sr = allgather_result
            for name, mine, theirs, item in (("exchanged digests", shard.buffer, a_dig, 8), ("sorted digests", d_sd, a_sd, 8), ("sorted rows", d_sr, a_sr, 4)):
                if ok and mine.download((total * nbl * item,), np.uint8).tobytes() != theirs.download((total * nbl * item,), np.uint8).tobytes():
                    ok, why = False, f"{name} differ from the all-gather path's"
    # ---- config 5: blocks + digests of the own rows in one read
    nb = -(-k // 64)
    d_blk, d_loc5 = ctx.alloc(max(1, n3 * nb * 8)), ctx.alloc(max(1, bands * n3 * 8))
    fused = [False]

    def c5_call():
        fused[0] = ctx.bbit_pack_band_digests_dev(d_shard.ptr, U32, n3, k, 1, bands, r, d_blk.ptr, d_loc5.ptr, BM)

    ms_fused = _timed(ctx, c5_call, reps=3, ramp=0.05)
    if ok and d_loc5.download((bands * n3 * 8,), np.uint8).tobytes() != d_loc.download((bands * n3 * 8,), np.uint8).tobytes():
        ok, why = False, "config 5's digests differ from the digest kernel's"
    if ok and len(sel):
        blocks = _download_rows(d_blk, sel, nb, np.uint64)
        if not np.array_equal(blocks, pack_matrix(want, 1, gpu_mode="disable")):
            ok, why = False, "b = 1 blocks differ from the reference's packing (b_bit_minhash.py:82-101)"
    oks = group.allgather(b"" if ok else why.encode())
    if any(oks):
        raise SystemExit("PARITY FAILURE (extra.c3_sharded.by_band / c5_sharded): " + "; ".join(f"rank {q}: {o.decode()}" for q, o in enumerate(oks) if o))
    stages = {"signatures": ms_sig, "band_digests_own_rows": ms_dig, "exchange": ms_x, "bucketing": ms_sort, "c5_fused_own_rows": ms_fused}
    per_rank = {name: [float(np.frombuffer(p, dtype=np.float64)[0]) for p in group.allgather(np.float64(v).tobytes())] for name, v in stages.items()}
    worst = {name: max(v) for name, v in per_rank.items()}
    received = [int(np.frombuffer(p, dtype=np.int64)[0]) for p in group.allgather(np.int64(shard.bytes_received).tobytes())]
    x = {"transport": shard.transport, "wire": "uint64 band digests, one run per (peer, band), written in place", "bytes_received_per_gpu": max(received),
         "bytes_received_per_rank": received, "ms": worst["exchange"], "GBps_per_gpu": max(received) / (worst["exchange"] * 1e-3) / 1e9,
         "xgmi_bound_GBps_per_gpu": XGMI_LINKS * XGMI_GBPS_PER_LINK,
         "note": "one grouped launch of ncclSend / ncclRecv (mhx_comm_exchange_dev); received bytes / slowest rank's wall time after a warm-up exchange"}
    c3_keys = ("signatures", "band_digests_own_rows", "exchange", "bucketing")
    by_band = {
        "workload": f"config 3 with the index partitioned by band: {world} ranks x {n3} sets, num_perm={k}: own rows -> band-major digests ({bands} x {r}) -> "
                    f"by-band exchange -> bucketing of the rank's bands over all {total} rows; the signature matrix is never assembled",
        "per_rank_ms": {name: per_rank[name] for name in c3_keys}, "ms": {name: worst[name] for name in c3_keys},
        "pipeline_ms": sum(worst[name] for name in c3_keys),
        "signatures_per_s_end_to_end": total / (sum(worst[name] for name in c3_keys) * 1e-3),
        "exchange": x,
        "parity": "every rank: the exchanged [its bands, N] digest matrix, the sorted digests and the sorted rows byte-identical to the all-gather path's on this rank "
                  "(which is itself checked against the numpy path and FNV-1a-64 of the reference's key bytes)",
    }
    c5_keys = ("c5_fused_own_rows", "exchange", "bucketing")
    alg = n3 * (4 * k + 8 * nb + 8 * bands)
    c5 = {
        "workload": f"config 5 across {world} ranks: b=1 packing of the rank's own {n3} x {k} signatures (uint32) + band hashing ({bands} x {r}) in one read, "
                    f"by-band exchange of the digests, bucketing of the rank's bands over all {total} rows; the blocks stay with their rows (32 B per row)",
        "fused": {"one_read": bool(fused[0]), "kernel_ms": worst["c5_fused_own_rows"], "algorithmic_bytes_per_launch": int(alg),
                  "achieved_GBps": alg / (worst["c5_fused_own_rows"] * 1e-3) / 1e9, "frac_of_8TBps": alg / (worst["c5_fused_own_rows"] * 1e-3) / 8e12},
        "per_rank_ms": {name: per_rank[name] for name in c5_keys}, "ms": {name: worst[name] for name in c5_keys},
        "pipeline_ms": sum(worst[name] for name in c5_keys),
        "rows_per_s_end_to_end": total / (sum(worst[name] for name in c5_keys) * 1e-3),
        "exchange": x,
        "parity": "every rank: the fused kernel's band-major digests byte-identical to the digest kernel's on all its rows; the b = 1 blocks of up to 512 of its rows "
                  "equal the numpy packing (b_bit_minhash.py:82-101 bit order) of the numpy path's signatures; exchange and bucketing as in by_band",
    }
    return by_band, c5
