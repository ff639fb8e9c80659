#!/usr/bin/env python3
"""bench.py -- MinHash signatures/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: spawns its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1M sets x 256 tokens, num_perm=128 per GPU, synthetic
pre-hashed tokens ``RandomState(42+rank).randint(0, 2**32, (N, T), uint64)``, seed=1.  A "step"
is one pass of the hot path (``mhx_minhash_bulk_dev``) over the whole resident corpus, producing
the [N, K] uint64 signature matrix in HBM.  Inputs are in HBM before the timed region starts.
Weak scaling: every rank hashes its own 1M-set shard; there is no collective in the data path, so
``value`` is the compute-only rate.  At N > 1 the config-3 exchange step -- one RCCL all-gather of the
uint32 shards over xGMI -- is measured on its own after the timed region and reported under
``allgather`` (ms, bytes, ranks RCCL itself counts); ``--allgather`` puts it inside every step.

No PyTorch anywhere: ranks find each other over datasketch_amd.rendezvous (TCP; MASTER_ADDR /
MASTER_PORT of the launcher, or the port the self-spawning parent picked), which carries the
barriers, the max-over-ranks reduction and the 128-byte RCCL id.

Prints ONE JSON line on rank 0.  ``roofline.achieved`` = algorithmic bytes (8*T + 8*K per
signature) / average launch duration measured with HIP events on the
kernel's own stream.  ``cpu_baseline`` = the numpy restatement of the reference's CPU path
(oracle/, kind "port") timed on a bounded sample on this host, rank 0, N=1 only.  ``extra`` (N=1)
holds the other BASELINE configs at their per-GPU shapes, each HIP-event timed and parity-gated:
``c3`` (1.25M x 256, K=256 + LSH band digests + sort), ``c4`` (weighted, 100k x 4096, S=128),
``c5`` (b=1 packing + band digests of the 1.25M x 256 matrix).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchmarks.common import HBM_PEAK_GBS, kernel_stamp  # noqa: E402
from benchmarks.cpu import cpu_baseline  # noqa: E402
from benchmarks.sharded import AllGather, allgather_probe, c3_sharded  # noqa: E402
from benchmarks.single_gpu import extra_configs, extra_full, weighted_gap_gate  # noqa: E402,F401  (tests import it from here)

TRAFFIC_FILE = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_traffic_minhash_bulk.json") for r in (6, 5, 4, 3)) if os.path.exists(p)),
                    os.path.join(ROOT, "profiles", "r03_traffic_minhash_bulk.json"))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clock-warmup", type=float, default=0.4, help="seconds of untimed steps in front of the W warm-up steps (GPU clocks; 0 = none)")
    ap.add_argument("--sets", type=int, default=1_000_000, help="sets per GPU")
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--num-perm", type=int, default=128)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--allgather", action="store_true", help="RCCL all-gather of the uint32 shards inside every step")
    ap.add_argument("--no-allgather-probe", action="store_true", help="N > 1: skip the separate all-gather measurement")
    ap.add_argument("--probe-timeout", type=float, default=240.0, help="seconds the all-gather probe may take before it is given up (a first ncclCommInitRank over 8 GPUs can take tens of seconds)")
    ap.add_argument("--check-rows", type=int, default=4096, help="rows verified against the numpy path")
    ap.add_argument("--no-check-all", action="store_true", help="N=1: skip the comparison of EVERY row with the C oracle (keeps the --check-rows sample)")
    ap.add_argument("--cpu-sample", type=int, default=160_000, help="sets timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive host->host measurement")
    ap.add_argument("--no-extra", action="store_true", help="skip the config 3/4/5 entries (N=1 only anyway)")
    ap.add_argument("--extra-only", default="", help="comma list out of c3,c4,c5 (default: all)")
    ap.add_argument("--full-only", action="store_true", help="N=1: only configs 3 and 5 at their stated size (10M x 256, num_perm=256) with every check, one JSON line")
    ap.add_argument("--full-rows", type=int, default=10_000_000, help="rows of the full-size configs 3 / 5 (tests shrink it)")
    ap.add_argument("--allgather-transport", default=None, choices=["rccl", "host"],
                    help="N > 1: how the uint32 shards travel (default rccl over xGMI; host = staged through host memory, the explicit opt-in "
                         "that lets ranks share one GPU -- labelled in the JSON line, never chosen silently)")
    ap.add_argument("--no-c3-sharded", action="store_true", help="N > 1: skip extra.c3_sharded (config 3 end to end across the ranks)")
    ap.add_argument("--c3-rows", type=int, default=1_250_000, help="N > 1: rows per rank of extra.c3_sharded")
    ap.add_argument("--u32", action="store_true", help="compact variant: uint32 tokens in, uint32 signatures out")
    ap.add_argument("--share-devices", action="store_true", help="testing only: let several ranks use one GPU (no RCCL then)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="libmhx tuning knob (mhx_ctx_set_option), e.g. blocks_per_cu=4")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process spawns the ranks (plain subprocesses, LOCAL_RANK -> device,
# HIP_VISIBLE_DEVICES untouched) and hands them a rendezvous port; rank 0 prints the JSON line.
def spawn_ranks(args) -> int:
    from datasketch_amd import rendezvous

    import secrets

    port = rendezvous.free_port()
    nonce = secrets.token_hex(16)  # only this job's ranks may join the group
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MHX_RDZV_ADDR=f"127.0.0.1:{port}", MHX_RDZV_NONCE=nonce)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("MHX_BENCH_TIMEOUT", "1500"))
    for p in procs:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or p.returncode
    return rc


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

    from datasketch_amd import _native, rendezvous
    from datasketch_amd.minhash import MinHash

    group = rendezvous.from_env(timeout=float(os.environ.get("MHX_RDZV_TIMEOUT", "300")))  # eight ranks start eight HIP runtimes at once
    world, rank = group.world, group.rank
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    visible = _native.device_count()
    if visible < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # one GPU per rank: LOCAL_RANK indexes the visible devices; a launcher that already narrowed the
    # visibility to one device per process (HIP_VISIBLE_DEVICES) leaves device 0
    # (MHX_ONE_DEVICE_PER_PROCESS=1 says so)
    shared = local_world > visible and os.environ.get("MHX_ONE_DEVICE_PER_PROCESS", "0") != "1"
    if shared and not args.share_devices:
        raise SystemExit(f"{local_world} ranks on this node but only {visible} visible GPU(s): one GPU per rank "
                         "(--share-devices lets ranks share a GPU for plumbing tests; the numbers then mean nothing)")
    ctx = _native.Context(local_rank % visible)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        ctx.set_option(key, int(val))

    def sync():
        ctx.synchronize()

    if args.full_only:
        if world != 1:
            raise SystemExit("--full-only is a single-GPU run")
        res = extra_full(ctx, args.full_rows, args.seed, checks="all")
        print(json.dumps({"metric": "configs 3 and 5 at their stated size on one GPU (every check)", "n_gpus": 1, "data": "synthetic",
                          "guard_alloc": os.environ.get("MHX_GUARD_ALLOC"), "extra": res}), flush=True)
        return

    n, t, k = args.sets, args.tokens, args.num_perm
    proto = MinHash(num_perm=k, seed=args.seed, hashfunc=lambda x: x)
    perms = proto.permutations

    # ---- synthetic corpus, resident in HBM before timing.  One rank keeps the whole numpy array (the CPU baseline, the
    # host-to-host figures and the extra configs read it); with several ranks on a node each generates its shard in
    # pieces of 50k sets (100 MB) that go up as they are made -- same RandomState stream, bounded host footprint --
    # and keeps only the rows the parity check will look at.
    rng = np.random.RandomState(42 + rank)
    tok_dtype, out_dtype, out_np, tok_bytes, out_bytes = ((_native.MHX_U32, _native.MHX_U32, np.uint32, 4, 4) if args.u32
                                                          else (_native.MHX_U64, _native.MHX_U64, np.uint64, 8, 8))
    tok_np = np.uint32 if args.u32 else np.uint64
    check = max(0, min(args.check_rows, n))
    check_rows_idx = np.unique(np.linspace(0, n - 1, check).astype(np.int64)) if check else np.empty(0, dtype=np.int64)
    if world == 1:
        tokens = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
        d_tok = ctx.to_device(tokens.astype(tok_np) if args.u32 else tokens)
        check_tokens = tokens[check_rows_idx]
    else:
        tokens = None
        d_tok = ctx.alloc(n * t * tok_bytes)
        kept, piece = [], 50_000
        for lo in range(0, n, piece):
            part = rng.randint(0, 2**32, size=(min(piece, n - lo), t), dtype=np.uint64)
            sel = check_rows_idx[(check_rows_idx >= lo) & (check_rows_idx < lo + len(part))] - lo
            kept.append(part[sel])
            d_tok.upload(part.astype(tok_np) if args.u32 else part, offset=lo * t * tok_bytes)
        check_tokens = np.concatenate(kept) if kept else np.empty((0, t), dtype=np.uint64)
        del part, kept
    d_out = ctx.alloc(n * k * out_bytes)
    ctx.perm_handle(perms)

    gather = None
    gather_error = None
    from datasketch_amd import dist as _dist

    transport = _dist.allgather_transport(args.allgather_transport)
    if args.allgather and world > 1:
        try:
            gather = AllGather(ctx, group, n, k, transport)
        except Exception as e:  # noqa: BLE001 -- RCCL trouble must not take the compute numbers down with it
            gather_error = repr(e)
        if any(group.allgather(b"\x01" if gather is None else b"\x00")[r] == b"\x01" for r in range(world)):
            gather = None  # all ranks or none

    def step():
        ctx.minhash_bulk_dev(perms, d_tok.ptr, tok_dtype, None, t, n, n * t, None, 0, d_out.ptr, out_dtype)
        if gather is not None:
            gather.step(perms, d_tok, tok_dtype, t)

    # An MI355X that idled while the host drew the corpus is in a low-power state and needs tens of milliseconds of work to
    # bring its clocks back (profiles/r04_clock_ramp.txt: the same launch 1.08 -> 0.92 ms over 30 launches); the W warm-up
    # steps the contract asks for are 10 ms.  So the same step first runs untimed for --clock-warmup seconds: what is timed
    # below is what a long-running job sees, and the number of those extra steps is reported in the JSON line.
    clock_steps = 0
    t_ramp = time.perf_counter()
    if gather is not None and args.clock_warmup > 0:
        # (a step with the all-gather inside is a collective: every rank must run the same number of them, so not by the clock)
        clock_steps = 96
        for _ in range(clock_steps):
            step()
        sync()
    while gather is None and args.clock_warmup > 0 and time.perf_counter() - t_ramp < args.clock_warmup:
        for _ in range(8):
            step()
        clock_steps += 8
        sync()
    for _ in range(args.warmup):
        step()
    sync()

    # ---- timed region: exactly K steps, barrier + sync on both sides, max over ranks
    evs = [ctx.event() for _ in range(args.steps + 1)]
    group.barrier()
    sync()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    sync()
    elapsed_rank = time.perf_counter() - t0  # this rank's K steps, device work complete
    group.barrier()                          # closing barrier; the job's time is the MAX over ranks below
    per_rank = [float(np.frombuffer(p, dtype=np.float64)[0]) for p in group.allgather(np.float64(elapsed_rank).tobytes())]
    elapsed = max(per_rank)
    launch_ms = [evs[i].elapsed_ms(evs[i + 1]) for i in range(args.steps)]
    kernel_ms = float(np.mean(launch_ms))
    kernel_ms_ranks = [float(np.frombuffer(p, dtype=np.float64)[0]) for p in group.allgather(np.float64(kernel_ms).tobytes())]

    # ---- one counted launch outside the timed region: how often the sieve's proof failed
    ctx.counters(True)
    ctx.minhash_bulk_dev(perms, d_tok.ptr, tok_dtype, None, t, n, n * t, None, 0, d_out.ptr, out_dtype)
    counters = ctx.counters(False)
    try:  # which sets left the first launch's proof (the launches' own hand-over bytes: mhx_ctx_minhash_flags)
        flags = ctx.minhash_flags(n)
    except ValueError:  # (a shape that keeps none: one huge set split over waves)
        flags = None

    # ---- parity: rows spread over the whole matrix against the package's numpy path (gpu_mode="disable": the
    # reference's arithmetic, minhash.py:293-297), bit-exact or fail.  The oracle itself is only used in the
    # cpu_baseline leg below, which also compares its rows with the GPU's.
    from datasketch_amd.hashfunc import prehashed

    a, b = perms
    sig_head = None
    all_rows = None
    if check or (args.cpu_sample > 0 and rank == 0 and world == 1):
        sig = d_out.download((n, k), out_np)
        if check:
            want = MinHash.bulk_signatures(check_tokens, num_perm=k, seed=args.seed, hashfunc=prehashed, gpu_mode="disable")
            if not np.array_equal(sig[check_rows_idx].astype(np.uint64), want):
                raise SystemExit("PARITY FAILURE: GPU signatures differ from the numpy path")
        sig_head = sig[: min(n, 40_000)].astype(np.uint64)
        if world == 1 and not args.no_check_all and tokens is not None:
            # EVERY row against the C oracle (threads; ~2 s for 1M x 256 x 128 on 16 cores), and the flagged sets -- the ones the
            # rare-event launches produced -- by id: the checker, never the thing measured
            from oracle import oracle as O

            w0 = time.perf_counter()
            want_all = O.c_minhash_bulk_dense_parallel(tokens, a, b)
            all_rows = {"rows": int(n), "oracle_seconds": time.perf_counter() - w0, "oracle_threads": O.usable_threads()}
            if not np.array_equal(sig if sig.dtype == np.uint64 else sig.astype(np.uint64), want_all):
                raise SystemExit("PARITY FAILURE: GPU signatures differ from the C oracle (all rows)")
            if flags is not None:
                fl = np.flatnonzero(flags)
                if not np.array_equal(sig[fl].astype(np.uint64), want_all[fl]):
                    raise SystemExit("PARITY FAILURE: a set the second / third launch produced differs from the C oracle")
                all_rows.update({"flagged_sets_checked_by_id": int(fl.size), "of_them_pairwise": int((flags == 2).sum())})
            del want_all
        del sig

    out = {
        # BASELINE.json's metric at the default shape; any other shape is named as what it is
        "metric": "MinHash signatures/sec (1M sets x 256 tokens, num_perm=128)" if (n, t, k) == (1_000_000, 256, 128)
        else f"MinHash signatures/sec ({n} sets x {t} tokens, num_perm={k})",
        "value": world * n * args.steps / elapsed,
        "unit": "signatures/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "clock_warmup": {"steps": clock_steps, "seconds": args.clock_warmup,
                         "note": "untimed steps in front of the W warm-up steps: the box idles in a low-power state while the host draws the corpus"},
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32" if args.u32 else "u64",
        "data": "synthetic",
        "config": {
            "workload": f"MinHash.bulk {n} sets x {t} tokens, num_perm={k}, per GPU (BASELINE.json configs[1])",
            "sets_per_gpu": n,
            "tokens_per_set": t,
            "num_perm": k,
            "token_dtype": "uint32" if args.u32 else "uint64",
            "signature_dtype": "uint32" if args.u32 else "uint64",
            "parallelism": f"shard{world}" + ("+allgather" if gather is not None else ""),
            "allgather_transport": transport if world > 1 else None,
            # (a library other than librccl.so behind the "rccl" transport -- tests/fake_rccl.c in the test suite -- is named, never silent)
            **({"rccl_library_override": os.environ["MHX_RCCL_LIBRARY"]} if os.environ.get("MHX_RCCL_LIBRARY") else {}),
            "launcher": "torch.distributed.run env" if "TORCHELASTIC_RUN_ID" in os.environ else ("self-spawned ranks" if world > 1 else "single process"),
            "rendezvous": "datasketch_amd.rendezvous (TCP, no PyTorch)",
            "parity_rows_checked": int(all_rows["rows"]) if all_rows else int(check),
            **({"parity_all_rows": all_rows} if all_rows else {}),
            "sets_redone_by_full_evaluation": counters["sieve_sets_redone"],
            "sets_redone_by_exact_fold": counters["exact_sets_redone"],
        },
        "per_rank": {
            "ms_per_step": [1e3 * x / args.steps for x in per_rank],
            "kernel_ms": kernel_ms_ranks,
            "devices": [p.decode() for p in group.allgather(("%d:%s" % (ctx.device, ctx.info()["name"])).encode())],
        },
    }
    if gather_error:
        out["allgather_error"] = gather_error
    alg_bytes = n * (tok_bytes * t + out_bytes * k)  # 8*T + 8*K per signature
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    pair_rate = n * t * k / (kernel_ms * 1e-3)
    out["roofline"] = {
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": measured_traffic(n, t, k, args),
        "valu_issue_frac": measured_traffic(n, t, k, args, "valu_issue_frac"),
        "traffic_source": traffic_profile()[1],
        "kernel_stamp": kernel_stamp(),
        "kernel": "minhash_bulk_kernel<MODE_SIEVE> (+ the MODE_FULL launch over the flagged sets)",
        "kernel_ms": kernel_ms,
        "algorithmic_bytes_per_launch": alg_bytes,
        "note": "integer-VALU-bound kernel: (token,perm) pair evaluations/s = %.3e" % pair_rate,
    }

    # ---- N > 1: the exchange step of config 3 on its own (uint32 shards, RCCL over xGMI)
    if world > 1 and not args.no_allgather_probe:
        # RCCL blocks inside the library when a peer is missing or a topology is refused on one rank only; the compute
        # numbers above are complete, so a probe that does not come back in time is reported as such instead of
        # hanging the job: the watchdog prints the line (rank 0) and leaves
        import threading

        def give_up():
            if rank == 0:
                out.setdefault("allgather", {"error": f"the all-gather probe did not finish within {args.probe_timeout:.0f} s"})
                if "allgather" in out and "error" not in out["allgather"]:
                    out.setdefault("extra", {})["c3_sharded"] = {"error": f"did not finish within {args.probe_timeout:.0f} s"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(args.probe_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            out["allgather"] = allgather_probe(ctx, group, gather, perms, d_tok, tok_dtype, n, t, k, transport=transport)
            out["per_rank"]["rccl_ranks_seen"] = out["allgather"].get("rccl_ranks_seen")  # ncclCommCount as every rank reports it
            if not args.no_c3_sharded and not args.u32 and "error" not in out["allgather"]:
                dog.cancel()
                dog = threading.Timer(args.probe_timeout, give_up)  # a fresh allowance for the second collective phase
                dog.daemon = True
                dog.start()
                d_out.free()
                c3s = c3_sharded(ctx, group, args, d_tok, n, t, check_rows_idx, check_tokens)
                c5s = c3s.pop("c5_sharded", None)  # config 5 across the ranks rides on the same shards (benchmarks/sharded.py: by_band_chain)
                out.setdefault("extra", {})["c3_sharded"] = c3s
                if c5s is not None:
                    out["extra"]["c5_sharded"] = c5s
        except (ConnectionError, TimeoutError, OSError) as e:  # a peer left (its own watchdog, or a crash inside RCCL)
            out["allgather"] = {"error": repr(e)}
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)
        except (Exception, SystemExit) as e:
            # anything else that goes wrong in the exchange phases -- an MhxError out of RCCL, a PARITY FAILURE of the sharded chain on this
            # rank -- is reported IN the line, loudly, instead of taking the complete headline measurement above down with it; the peers this
            # rank leaves inside a collective are released by their own watchdogs
            import traceback

            traceback.print_exc()
            where = "allgather" if "allgather" not in out else None
            if where:
                out["allgather"] = {"error": repr(e)}
            else:
                out.setdefault("extra", {})["c3_sharded"] = {"error": repr(e), "rank": rank}
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)  # (a non-zero status would make the launcher kill rank 0 before its watchdog prints the line; the traceback is on stderr)
        dog.cancel()

    if rank == 0 and world == 1:
        if not args.no_e2e:
            # host numpy in -> host numpy out through mhx_minhash_bulk (pageable memory): the first call
            # also allocates device scratch and first-touches a fresh result array; the second reuses both
            t1 = time.perf_counter()
            host_out = ctx.minhash_bulk(perms, tokens.reshape(-1), None, t, n, None)
            out["pcie_inclusive_first_call_value"] = n / (time.perf_counter() - t1)
            t1 = time.perf_counter()
            ctx.minhash_bulk(perms, tokens.reshape(-1), None, t, n, None, out=host_out)
            out["pcie_inclusive_value"] = n / (time.perf_counter() - t1)
            del host_out
            # uint32 tokens in, uint32 signatures out (the range sha1_hash32 produces): half the bytes on the link
            tok32 = tokens.astype(np.uint32).reshape(-1)
            host32 = np.empty((n, k), dtype=np.uint32)
            ctx.minhash_bulk(perms, tok32, None, t, n, None, out=host32)
            t1 = time.perf_counter()
            ctx.minhash_bulk(perms, tok32, None, t, n, None, out=host32)
            out["pcie_inclusive_u32_value"] = n / (time.perf_counter() - t1)
            if sig_head is not None and not np.array_equal(host32[: len(sig_head)], sig_head.astype(np.uint32)):
                raise SystemExit("PARITY FAILURE: uint32 host entry differs from the uint64 device path")
            del tok32, host32
            ctx.release_scratch()
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(tokens, a, b, min(args.cpu_sample, n), k, t, sig_head, seed=args.seed)
        if not args.no_extra:
            d_out.free()
            only = [x for x in args.extra_only.split(",") if x] or ["c3", "c5", "c4", "full"]
            out["extra"] = extra_configs(ctx, tokens, d_tok, args.seed, only)
            if "full" in only:  # configs 3 and 5 at their stated size (10M rows) on this one GPU
                d_tok.free()
                del tokens
                ctx.release_scratch()
                out["extra"].update(extra_full(ctx, args.full_rows, args.seed, checks="sample"))
    group.barrier()
    group.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # leave without the interpreter's teardown: a process that holds a (possibly half-built) RCCL communicator
        # can block in the library's exit handlers, and the launcher would wait for it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def measured_traffic(n, t, k, args, field="traffic_bytes_per_launch"):
    """HBM bytes per launch (or the share of VALU issue cycles the launch needs) from the committed rocprofv3 PMC passes of
    this same command (TRAFFIC_FILE) -- replayed, not measured in this process, and only when the profile was taken on THIS
    kernel: its stamp's kernel_source_sha256 must equal the hash of csrc/minhash_kernels.hip + mhx_internal.h in this tree
    (benchmarks.common.kernel_stamp).  None for any other shape, an unstamped profile or another kernel."""
    if (n, t, k) != (1_000_000, 256, 128) or args.u32:
        return None
    return traffic_profile()[0].get(field)


def traffic_profile():
    """(profile dict or {}, why): the committed counter profile if it belongs to this tree's kernel."""
    if not os.path.exists(TRAFFIC_FILE):
        return {}, "no counter profile committed"
    with open(TRAFFIC_FILE) as f:
        prof = json.load(f)
    have = (prof.get("stamp") or {}).get("kernel_source_sha256")
    want = kernel_stamp()["kernel_source_sha256"]
    name = os.path.basename(TRAFFIC_FILE)
    if have is None:
        return {}, f"profiles/{name} carries no stamp (taken before round 6): not replayed"
    if have != want:
        return {}, f"profiles/{name} was taken on another kernel source (sha256 {have[:12]}..., this tree {want[:12]}...): not replayed"
    commit = prof["stamp"].get("git_commit")
    return prof, (f"profiles/{name}: rocprofv3 --pmc passes over this same command (tools/traffic.sh) on the same kernel source (sha256 {want[:12]}..., "
                  f"commit {commit}), replayed here (traffic, valu_issue_frac) -- not measured in this process")


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:  # noqa: BLE001
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:  # a rank: report, then leave hard (see main)
            import traceback

            if not isinstance(exc, SystemExit) or exc.code not in (0, None):
                if isinstance(exc, SystemExit):
                    print(exc.code, file=sys.stderr)
                else:
                    traceback.print_exc()
                sys.stderr.flush()
                os._exit(1)
            os._exit(0)
        raise
