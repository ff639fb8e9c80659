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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
TRAFFIC_FILE = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_traffic_minhash_bulk.json") for r in (5, 4, 3)) if os.path.exists(p)),
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

    # ---- parity: rows spread over the whole matrix against the package's numpy path (gpu_mode="disable": the
    # reference's arithmetic, minhash.py:293-297), bit-exact or fail.  The oracle itself is only used in the
    # cpu_baseline leg below, which also compares its rows with the GPU's.
    from datasketch_amd.hashfunc import prehashed

    a, b = perms
    sig_head = None
    if check or (args.cpu_sample > 0 and rank == 0 and world == 1):
        sig = d_out.download((n, k), out_np)
        if check:
            want = MinHash.bulk_signatures(check_tokens, num_perm=k, seed=args.seed, hashfunc=prehashed, gpu_mode="disable")
            if not np.array_equal(sig[check_rows_idx].astype(np.uint64), want):
                raise SystemExit("PARITY FAILURE: GPU signatures differ from the numpy path")
        sig_head = sig[: min(n, 40_000)].astype(np.uint64)
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
            "parity_rows_checked": int(check),
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
        "traffic_source": "profiles/%s: rocprofv3 --pmc passes over this same command (tools/traffic.sh), replayed here (traffic, valu_issue_frac) -- not measured in this process" % os.path.basename(TRAFFIC_FILE),
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
                out.setdefault("extra", {})["c3_sharded"] = c3_sharded(ctx, group, args, d_tok, n, t, check_rows_idx, check_tokens)
        except (ConnectionError, TimeoutError, OSError) as e:  # a peer left (its own watchdog, or a crash inside RCCL)
            out["allgather"] = {"error": repr(e)}
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)
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
    """HBM bytes per launch (or the share of VALU issue cycles the launch needs) from the committed rocprofv3 PMC
    passes of this same command (TRAFFIC_FILE; replayed, not measured in this process); None for any other shape."""
    if (n, t, k) != (1_000_000, 256, 128) or args.u32 or not os.path.exists(TRAFFIC_FILE):
        return None
    with open(TRAFFIC_FILE) as f:
        return json.load(f).get(field)


# ------------------------------------------------------------------------------------------------
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
def _cpu_worker(args):
    """One host core of the all-cores baseline: the numpy per-set loop of MinHash.bulk on its own
    shard (generated in the worker: nothing but a checksum travels)."""
    seed, n, t, k, pseed, ref_path = args
    sys.path.insert(0, ROOT)
    tokens = np.random.RandomState(seed).randint(0, 2**32, size=(n, t), dtype=np.uint64)
    if ref_path:  # the reference itself (DATASKETCH_REFERENCE): MinHash.bulk with its per-set copy()
        sys.path.insert(0, ref_path)
        import datasketch as ref

        t0 = time.perf_counter()
        objs = ref.MinHash.bulk(tokens, num_perm=k, seed=pseed, hashfunc=_identity)
        return time.perf_counter() - t0, int(sum(int(m.hashvalues[0]) for m in objs))
    from oracle import oracle as O

    a, b = O.np_init_permutations(k, pseed)
    t0 = time.perf_counter()
    sig = O.np_minhash_bulk(list(tokens), a, b)
    return time.perf_counter() - t0, int(sig[:, 0].sum())


def _identity(x):
    return x


def reference_path():
    """Where the real reference (ekzhu/datasketch) can be imported from, or None.  Only when DATASKETCH_REFERENCE names it:
    the GPU box has no reference, and nothing here looks for one on its own."""
    path = os.environ.get("DATASKETCH_REFERENCE", "").strip()
    if not path or not os.path.isdir(os.path.join(path, "datasketch")):
        return None
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, sys.argv[1]); import datasketch; print(datasketch.MinHash.__module__)", path],
                           capture_output=True, text=True, timeout=120)
    return path if probe.returncode == 0 and "datasketch" in probe.stdout else None


def _usable_cores(cap=64, why=None):
    """Cores this process may really use: affinity mask, clipped by the cgroup CPU quota (a container
    can see 256 CPUs and own 8) and by `cap` (start-up of hundreds of interpreters is not the point).
    `why` (a dict) receives where the number came from, for the bench line."""
    seen = os.cpu_count() or 1
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else seen
    if why is not None:
        why.update({"os_cpu_count": seen, "sched_getaffinity": n, "cgroup_quota_cpus": None, "cap": cap})
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    q = max(1, int(int(parts[0]) / int(parts[1])))
                    n = min(n, q)
                    if why is not None:
                        why["cgroup_quota_cpus"] = q
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
                    if why is not None:
                        why["cgroup_quota_cpus"] = max(1, quota // period)
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


def cpu_model() -> str:
    """`lscpu`'s model name (SURVEY.md section 8d asks for it next to the CPU number)."""
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in txt.splitlines():
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(tokens, a, b, sample, k, t, gpu_rows, seed=1):
    """The reference's CPU path (numpy restatement, oracle/oracle.py:np_minhash_bulk = the per-set
    loop of MinHash.bulk).  numpy's uint64 ufuncs are single-threaded, so "the host's cores" means
    one process per core, each with its own shard: `value` is that
    all-cores rate, `single_core_value` the rate of one process.  The rows the oracle produces for the
    timed sample are compared with the GPU's (`gpu_rows`): the baseline times the same function."""
    import multiprocessing as mp

    from oracle import oracle as O

    single = min(sample, 40_000)
    sets = list(tokens[:single])
    t0 = time.perf_counter()
    got = O.np_minhash_bulk(sets, a, b)
    dt = time.perf_counter() - t0
    ref_path = reference_path()
    ref_dt = None
    if ref_path:  # the real reference on the same sample, same core; its rows must be the restatement's
        code = ("import sys, time, numpy as np; sys.path.insert(0, sys.argv[1]); import datasketch as ref\n"
                "tok = np.load(sys.argv[2]); t0 = time.perf_counter()\n"
                "objs = ref.MinHash.bulk(tok, num_perm=int(sys.argv[3]), seed=int(sys.argv[4]), hashfunc=lambda x: x); dt = time.perf_counter() - t0\n"
                "np.save(sys.argv[5], np.stack([m.hashvalues for m in objs])); print(dt)")
        import tempfile

        with tempfile.TemporaryDirectory() as tmp:
            np.save(os.path.join(tmp, "tok.npy"), tokens[:single])
            p = subprocess.run([sys.executable, "-c", code, ref_path, os.path.join(tmp, "tok.npy"), str(k), str(seed), os.path.join(tmp, "sig.npy")],
                               capture_output=True, text=True, timeout=600)
            if p.returncode == 0:
                ref_dt = float(p.stdout.strip().splitlines()[-1])
                if not np.array_equal(np.load(os.path.join(tmp, "sig.npy")), got):
                    raise SystemExit("PARITY FAILURE: the reference's MinHash.bulk differs from the numpy restatement on the cpu_baseline sample")
            else:
                ref_path = None
    c0 = time.perf_counter()
    want = O.c_minhash_bulk_dense(tokens[:single], a, b)
    cdt = time.perf_counter() - c0
    assert np.array_equal(got, want)
    m = 0 if gpu_rows is None else min(single, len(gpu_rows))
    if m and not np.array_equal(got[:m], gpu_rows[:m]):
        raise SystemExit("PARITY FAILURE: GPU signatures differ from the oracle on the cpu_baseline sample")
    cores_why = {}
    cores = _usable_cores(why=cores_why)
    per = max(2_000, sample // 8)  # sets per process: 1.5-3 s of numpy each, 41 MB of tokens
    is_ref = ref_path is not None and ref_dt is not None
    what = "the reference's MinHash.bulk (DATASKETCH_REFERENCE)" if is_ref else "numpy per-set loop as MinHash.bulk"
    out = {
        "value": single / (ref_dt if is_ref else dt),
        "unit": "signatures/s",
        "cores": 1,
        "kind": "reference" if is_ref else "port",
        "sample": f"first {single} sets of the benchmark corpus ({t} tokens, num_perm={k}), {what}; {(ref_dt if is_ref else dt):.1f} s",
    }
    try:
        w0 = time.perf_counter()
        with mp.get_context("spawn").Pool(cores) as pool:  # spawn: children never see the HIP runtime
            res = pool.map_async(_cpu_worker, [(1000 + i, per, t, k, seed, ref_path if is_ref else None) for i in range(cores)]).get(timeout=180 if is_ref else 90)
        wall = time.perf_counter() - w0
        busy = max(r[0] for r in res)
        out.update({
            "value": cores * per / busy,
            "cores": cores,
            "cores_source": dict(cores_why, used=cores, rule="min(affinity mask, cgroup CPU quota, cap)"),
            "sample": f"{cores} processes x {per} sets of the same shape ({t} tokens, num_perm={k}), {what}; "
                      f"slowest process {busy:.1f} s (pool wall {wall:.1f} s incl. start-up)",
        })
    except Exception as e:  # the all-cores leg is best effort; the single-core figure stands
        out["all_cores_error"] = repr(e)
    out.update({
        "single_core_value": single / (ref_dt if is_ref else dt),
        "single_core_sample": f"first {single} sets of the benchmark corpus, {(ref_dt if is_ref else dt):.1f} s",
        "host_cpus": os.cpu_count(),
        "cpu_model": cpu_model(),
        # the real reference (MinHash.bulk with its per-set copy()) timed beside this restatement in the build container,
        # same sample and core: tools/cpu_reference_vs_port.py -> profiles/r02_cpu_reference_vs_port.txt
        # the real reference timed beside the restatement: here when DATASKETCH_REFERENCE resolves, else the build container's
        # figure (tools/cpu_reference_vs_port.py -> profiles/r05_cpu_reference_vs_port.txt, 2026-09-22)
        "reference_over_port_time": (ref_dt / dt) if is_ref else 1.21,
        "reference_over_port_measured": "in this run" if is_ref else "2026-09-22, build container, profiles/r05_cpu_reference_vs_port.txt",
        "port_single_core_value": single / dt,
        "note": ("kind 'reference' = ekzhu/datasketch's own MinHash.bulk imported from DATASKETCH_REFERENCE, rows equal to the restatement's and the GPU's"
                 if is_ref else
                 "kind 'port' = numpy restatement pinned to the reference; the reference itself is not on this box (DATASKETCH_REFERENCE unset) and "
                 "runs 1.21x slower than the restatement (object churn), so the reference-equivalent rate is value / 1.21"),
        "c_oracle_single_core_value": single / cdt,
        "rows_equal_to_gpu": int(m),
    })
    return out


# ------------------------------------------------------------------------------------------------
def _timed(ctx, fn, reps=3, ramp=0.25):
    """Average HIP-event time of `fn` (enqueues on ctx's stream) over `reps` runs, in ms, after `ramp` seconds of the same
    call untimed (GPU clocks: see the headline's clock warm-up)."""
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp:
        for _ in range(4):
            fn()
        ctx.synchronize()
    evs = [ctx.event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    ctx.synchronize()
    return float(np.mean([evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]))


def _roof(alg_bytes, ms):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": ms}


def extra_configs(ctx, tokens, d_tok, seed, only):
    """BASELINE.json configs 3, 4, 5 at their per-GPU shapes (the 8-GPU configs divide by 8), each timed with HIP
    events and parity-gated on a sample: a wrong result aborts the bench."""
    import ctypes

    from datasketch_amd import _native
    from datasketch_amd.minhash import MinHash
    from oracle import oracle as O

    lib = ctx.lib
    res = {}
    n3, t, k3, bands, r = 1_250_000, tokens.shape[1], 256, 32, 8
    state = {}

    def c3_corpus():
        # the config-2 corpus (already resident) + 250k more rows = one rank's 1.25M-set shard of config 3
        if "d_tok3" not in state:
            more = np.random.RandomState(4242).randint(0, 2**32, size=(n3 - tokens.shape[0], t), dtype=np.uint64)
            d = ctx.alloc(n3 * t * 8)
            ctx.copy_dev(d.ptr, d_tok.ptr, tokens.size * 8)
            d.upload(more, offset=tokens.size * 8)
            state["d_tok3"], state["more"] = d, more
            p3 = MinHash(num_perm=k3, seed=seed, hashfunc=lambda x: x).permutations
            state["perms3"] = p3
            state["d_sig3"] = ctx.alloc(n3 * k3 * 4)
            ctx.minhash_bulk_dev(p3, d.ptr, _native.MHX_U64, None, t, n3, n3 * t, None, 0, state["d_sig3"].ptr, _native.MHX_U32)
        return state

    def sample_rows():
        rows = np.unique(np.concatenate([np.linspace(0, tokens.shape[0] - 1, 384).astype(np.int64),
                                         np.arange(tokens.shape[0], tokens.shape[0] + 128)]))
        tok = np.concatenate([tokens[rows[rows < tokens.shape[0]]], state["more"][: 128]])
        return rows, tok

    if "c3" in only:
        st = c3_corpus()
        p3, d3, dsig = st["perms3"], st["d_tok3"], st["d_sig3"]
        ms_sig = _timed(ctx, lambda: ctx.minhash_bulk_dev(p3, d3.ptr, _native.MHX_U64, None, t, n3, n3 * t, None, 0, dsig.ptr, _native.MHX_U32))
        d_dig = ctx.alloc(n3 * bands * 8)
        d_sd = ctx.alloc(n3 * bands * 8)
        d_sr = ctx.alloc(n3 * bands * 4)
        ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, d_dig.ptr)))
        sort = lambda: _native.check(lib.mhx_lsh_sort_bands_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, d_sd.ptr, d_sr.ptr))
        ctx.set_option("lsh.sort", 1)  # A/B: the library radix sort (round 2's path, now the fallback), same call
        try:
            ms_sort_radix = _timed(ctx, sort, reps=3)
            sd_radix, sr_radix = d_sd.download((bands, n3), np.uint64), d_sr.download((bands, n3), np.uint32)
        finally:
            ctx.set_option("lsh.sort", 0)
        ms_sort = _timed(ctx, sort)
        # config 3 as a chain computes the digests once: the bucketing takes the [n, bands] digest matrix that was just written
        sort_dig = lambda: _native.check(lib.mhx_lsh_sort_digests_dev(ctx.handle, d_dig.ptr, n3, bands, d_sd.ptr, d_sr.ptr))
        ms_sort_dig_rm = _timed(ctx, sort_dig)
        sd_dig, sr_dig = d_sd.download((bands, n3), np.uint64), d_sr.download((bands, n3), np.uint32)
        # ... and the layout the chain runs on: the digests band-major ([bands, n]), read by the bucketing with unit stride
        d_dig_bm = ctx.alloc(n3 * bands * 8)
        ms_dig_bm = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, _native.BAND_MAJOR, d_dig_bm.ptr)))
        ms_sort_dig = _timed(ctx, lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig_bm.ptr, n3, bands, _native.BAND_MAJOR, d_sd.ptr, d_sr.ptr)))
        if not (np.array_equal(d_sd.download((bands, n3), np.uint64), sd_dig) and np.array_equal(d_sr.download((bands, n3), np.uint32), sr_dig)):
            raise SystemExit("PARITY FAILURE (extra.c3): bucketing from the band-major digests differs from bucketing from the row-major ones")
        dig_bm = d_dig_bm.download((bands, n3), np.uint64)
        d_dig_bm.free()
        ms_sort = _timed(ctx, sort)  # (d_sd / d_sr hold the sort-from-signatures result again for the checks below)
        # parity: signature rows against the C oracle, digests against FNV-1a of the reference's key bytes, order of the sort
        rows, tok = sample_rows()
        a3, b3 = p3
        want = O.c_minhash_bulk_dense(tok, a3, b3)
        sig = dsig.download((n3, k3), np.uint32)
        if not np.array_equal(sig[rows].astype(np.uint64), want):
            raise SystemExit("PARITY FAILURE (extra.c3): K=256 signatures differ from the oracle")
        keys = O.c_band_keys(want[:64], bands, r)
        dig = d_dig.download((n3, bands), np.uint64)
        if not np.array_equal(dig_bm, dig.T):
            raise SystemExit("PARITY FAILURE (extra.c3): band-major digests differ from the row-major ones")
        del dig_bm
        for i in range(64):
            for j in range(bands):
                if int(dig[rows[i], j]) != _fnv1a64(keys[i, j * r:(j + 1) * r].tobytes()):
                    raise SystemExit("PARITY FAILURE (extra.c3): band digest differs from FNV-1a-64 of the reference's key bytes")
        sd = d_sd.download((bands, n3), np.uint64)
        sr = d_sr.download((bands, n3), np.uint32)
        for j in (0, bands - 1):
            if np.any(sd[j, 1:] < sd[j, :-1]) or not np.array_equal(dig[sr[j].astype(np.int64), j], sd[j]):
                raise SystemExit("PARITY FAILURE (extra.c3): sorted bands are not the digests in ascending order")
        if not (np.array_equal(sd, sd_radix) and np.array_equal(sr, sr_radix)):
            raise SystemExit("PARITY FAILURE (extra.c3): the bucketing passes and the stable radix sort disagree")
        if not (np.array_equal(sd, sd_dig) and np.array_equal(sr, sr_dig)):
            raise SystemExit("PARITY FAILURE (extra.c3): bucketing from the digest matrix differs from bucketing from the signatures")
        del sig, dig, sd, sr, sd_radix, sr_radix, sd_dig, sr_dig
        res["c3"] = {
            "workload": f"config 3 per-GPU shard: {n3} sets x {t} tokens, num_perm={k3} (uint64 tokens in, uint32 signatures out = the all-gather's wire format), then LSH band digests ({bands} bands x {r}) and the bucketing sort",
            "signatures": dict(_roof(n3 * (8 * t + 4 * k3), ms_sig), signatures_per_s=n3 / (ms_sig * 1e-3),
                               note="algorithmic bytes 8*T + 4*K per signature (uint32 out); SURVEY 8d's 4096 B/sig assumes uint64 out"),
            "band_digests": dict(_roof(n3 * (4 * k3 + 8 * bands), ms_dig_bm), layout="band-major [bands, n] (MHX_BAND_MAJOR), written through an LDS tile"),
            "band_digests_row_major": _roof(n3 * (4 * k3 + 8 * bands), ms_dig),
            "lsh_sort_bands": dict(_roof(n3 * (4 * k3 + 12 * bands), ms_sort), keys_per_s=n3 * bands / (ms_sort * 1e-3),
                                   kernels="band_digest_bm_kernel (band-major digests into scratch) + lsh_bin_scatter_kernel + lsh_bin_sort_kernel",
                                   note="digests computed (their own pass since round 5: 0.86 -> 0.77 ms), scattered to bins by their top bits, every bin ordered in LDS: exact (band, digest, row) "
                                        "order; bytes = signatures in, (digest, row) out"),
            "lsh_sort_bands_radix": dict(_roof(n3 * (4 * k3 + 12 * bands), ms_sort_radix), keys_per_s=n3 * bands / (ms_sort_radix * 1e-3),
                                         note="lsh.sort=1: digests + the library radix sort of (band, digest prefix, row) + exact clean-up (round 2's path, "
                                              "now the fallback), same call, same box"),
            "lsh_sort_digests": dict(_roof(n3 * (8 * bands + 12 * bands), ms_sort_dig), keys_per_s=n3 * bands / (ms_sort_dig * 1e-3),
                                     note="mhx_lsh_sort_digests_layout_dev on the band-major digest matrix band_digests has just written (8 B read per key with "
                                          "unit stride, no hashing); bytes = digests in, (digest, row) out"),
            "lsh_sort_digests_row_major": dict(_roof(n3 * (8 * bands + 12 * bands), ms_sort_dig_rm), keys_per_s=n3 * bands / (ms_sort_dig_rm * 1e-3),
                                               note="the same from an [n, bands] matrix: every 128-byte input line is fetched by the four XCDs whose bands share it "
                                                    "(profiles/r05_pmc_scatter_work_orders.txt)"),
            "pipeline_ms": ms_sig + ms_dig_bm + ms_sort_dig,
            "pipeline": "signatures -> band_digests (band-major, kept: one key array per hashtable) -> lsh_sort_digests; digests computed once",
            "pipeline_ms_digests_twice": ms_sig + ms_dig + ms_sort,
            "parity": f"{len(rows)} signature rows vs the C oracle, 64 x {bands} digests vs FNV-1a-64 of the reference's key bytes, 2 bands' order, all {bands} sorted bands equal to the stable radix sort's and to the sort from the digest matrix",
        }
        for d in (d_dig, d_sd, d_sr):
            d.free()

    if "c5" in only:
        st = c3_corpus()
        dsig = st["d_sig3"]
        nb = k3 // 64
        d_pack = ctx.alloc(n3 * nb * 8)
        d_dig = ctx.alloc(n3 * bands * 8)
        ms_pack = _timed(ctx, lambda: _native.check(lib.mhx_bbit_pack_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, 1, d_pack.ptr)))
        ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_dev_typed(ctx.handle, dsig.ptr, _native.MHX_U32, n3, k3, bands, r, d_dig.ptr)))
        rows, tok = sample_rows()
        a3, b3 = st["perms3"]
        want = O.c_minhash_bulk_dense(tok, a3, b3)
        pack = d_pack.download((n3, nb), np.uint64)
        dig = d_dig.download((n3, bands), np.uint64)
        if not np.array_equal(pack[rows], O.c_bbit_pack(want, 1)):
            raise SystemExit("PARITY FAILURE (extra.c5): b=1 blocks differ from the oracle's bBitMinHash packing")
        # the same two outputs from ONE read of the matrix (bbit_digest_fused_kernel), over buffers cleared in between
        import ctypes as _ct

        for d in (d_pack, d_dig):
            _native.check(lib.mhx_memset_dev(ctx.handle, _ct.c_void_p(d.ptr), 0, d.nbytes))
        one_read = []
        ms_fused = _timed(ctx, lambda: one_read.append(ctx.bbit_pack_band_digests_dev(dsig.ptr, _native.MHX_U32, n3, k3, 1, bands, r, d_pack.ptr, d_dig.ptr)))
        if not (np.array_equal(d_pack.download((n3, nb), np.uint64), pack) and np.array_equal(d_dig.download((n3, bands), np.uint64), dig)):
            raise SystemExit("PARITY FAILURE (extra.c5): the fused kernel's blocks / digests differ from the two kernels'")
        keys = O.c_band_keys(want[:64], bands, r)
        for i in range(64):
            for j in range(bands):
                if int(dig[rows[i], j]) != _fnv1a64(keys[i, j * r:(j + 1) * r].tobytes()):
                    raise SystemExit("PARITY FAILURE (extra.c5): band digest differs from FNV-1a-64 of the reference's key bytes")
        for d in (d_pack, d_dig):
            _native.check(lib.mhx_memset_dev(ctx.handle, _ct.c_void_p(d.ptr), 0, d.nbytes))
        ms_fused_bm = _timed(ctx, lambda: one_read.append(ctx.bbit_pack_band_digests_dev(dsig.ptr, _native.MHX_U32, n3, k3, 1, bands, r, d_pack.ptr, d_dig.ptr, _native.BAND_MAJOR)))
        if not (np.array_equal(d_pack.download((n3, nb), np.uint64), pack) and np.array_equal(d_dig.download((bands, n3), np.uint64), dig.T)):
            raise SystemExit("PARITY FAILURE (extra.c5): the fused kernel's band-major digests / blocks differ from the two kernels'")
        del pack, dig
        res["c5"] = {
            "workload": f"config 5 per-GPU shard: b=1 packing of {n3} x {k3} signatures (uint32, as all-gathered) + LSH band hashing ({bands} x {r})",
            "fused": dict(_roof(n3 * (4 * k3 + k3 // 8 + 8 * bands), ms_fused), one_read=bool(one_read and all(one_read)),
                          kernel="bbit_digest_fused_kernel: blocks and digests from one read of the matrix (algorithmic bytes: 4K in, K/8 + 8*bands out)"),
            "bbit_pack_b1": _roof(n3 * (4 * k3 + k3 // 8), ms_pack),
            "band_digests": _roof(n3 * (4 * k3 + 8 * bands), ms_dig),
            "fused_band_major": dict(_roof(n3 * (4 * k3 + k3 // 8 + 8 * bands), ms_fused_bm),
                                     note="the same kernel writing the digests [bands, n] through an LDS tile: the layout the bucketing reads with unit stride"),
            "pipeline_ms": ms_fused_bm,
            "pipeline_ms_two_kernels": ms_pack + ms_dig,
            "parity": f"{len(rows)} packed rows vs the C oracle (b_bit_minhash.py:82-101 bit order), 64 x {bands} digests vs FNV-1a-64 of the reference's key "
                      f"bytes, and the fused kernel's outputs equal to the two kernels' on all {n3} rows",
        }
        d_pack.free()
        d_dig.free()

    for key in ("d_tok3", "d_sig3"):
        if key in state:
            state[key].free()
    state.clear()

    if "c4" in only:
        res["c4"] = extra_c4(ctx)
    return res


# ------------------------------------------------------------------------------------------------
XGMI_LINKS, XGMI_GBPS_PER_LINK = 7, 153.0  # 7 point-to-point links per GPU


def full_corpus(ctx, n, t, shards=8, piece=50_000, sample=4096):
    """Config 3's corpus, resident in HBM: shard q = RandomState(42 + q).randint(0, 2**32, (rows_q, t), uint64) (SURVEY.md
    section 8d), drawn by one host thread per shard (numpy releases the GIL inside the draw) in pieces of 50k sets that
    go up as they are made -- 20.5 GB on the device, 100 MB per thread on the host.  Returns the device buffer, `sample`
    row numbers spread over the whole corpus and their tokens (what the oracle will be given)."""
    from concurrent.futures import ThreadPoolExecutor

    from datasketch_amd.dist import shard_rows

    d_tok = ctx.alloc(n * t * 8)
    rows = np.unique(np.concatenate([np.linspace(0, n - 1, sample).astype(np.int64), [0, n - 1]]))

    def make(q):
        b, e = shard_rows(n, shards, q)
        rng = np.random.RandomState(42 + q)
        kept = []
        for lo in range(b, e, piece):
            m = min(piece, e - lo)
            part = rng.randint(0, 2**32, size=(m, t), dtype=np.uint64)
            sel = rows[(rows >= lo) & (rows < lo + m)]
            kept.append(part[sel - lo].copy())
            d_tok.upload(part, offset=lo * t * 8)
        return np.concatenate(kept) if kept else np.empty((0, t), dtype=np.uint64)

    with ThreadPoolExecutor(max(1, min(shards, _usable_cores()))) as pool:
        sample_tokens = np.concatenate(list(pool.map(make, range(shards))))
    return d_tok, rows, sample_tokens


def _download_rows(buf, rows, width, dtype):
    """The given rows of a row-major device matrix (one small copy per row: a few thousand rows of a 10 GB matrix)."""
    item = np.dtype(dtype).itemsize
    return np.stack([buf.download((width,), dtype, offset=int(r) * width * item) for r in rows]) if len(rows) else np.empty((0, width), dtype)


def extra_full(ctx, n, seed, checks="sample", t=256, k=256, bands=32, r=8):
    """BASELINE.json configs[2] and [4] at their STATED size on one GPU (an 8-GPU job holds exactly this on every GPU after
    the all-gather): n = 10M sets x 256 tokens (2.56e9 tokens: past 2^31 elements in every kernel), num_perm = 256 ->
    uint32 signatures (10.2 GB) -> band digests (32 x 8) -> bucketing of 320M (band, digest) keys; b = 1 blocks + band
    digests of the same matrix from one read.  checks = "sample": the spread sample rows against the C oracle at every
    stage, four bands of the sorted output in full; "all": every band, the fused outputs against the two kernels'
    everywhere, the bucketing against the stable radix sort everywhere."""
    from datasketch_amd import _native, lsh_bulk
    from datasketch_amd.minhash import MinHash
    from oracle import oracle as O

    lib = ctx.lib
    t0 = time.perf_counter()
    d_tok, rows, tok = full_corpus(ctx, n, t)
    gen_s = time.perf_counter() - t0
    perms = MinHash(num_perm=k, seed=seed, hashfunc=lambda x: x).permutations
    nb = k // 64
    d_sig = ctx.alloc(n * k * 4)
    ms_sig = _timed(ctx, lambda: ctx.minhash_bulk_dev(perms, d_tok.ptr, _native.MHX_U64, None, t, n, n * t, None, 0, d_sig.ptr, _native.MHX_U32), reps=2, ramp=0.1)
    d_tok.free()
    d_dig, d_sd, d_sr = ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 8), ctx.alloc(n * bands * 4)
    BM = _native.BAND_MAJOR  # the digests [bands, n]: one array per hashtable, read by the bucketing with unit stride
    ms_dig = _timed(ctx, lambda: _native.check(lib.mhx_band_digests_layout_dev(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, bands, r, BM, d_dig.ptr)), reps=3, ramp=0.1)
    sort_dig = lambda: _native.check(lib.mhx_lsh_sort_digests_layout_dev(ctx.handle, d_dig.ptr, n, bands, BM, d_sd.ptr, d_sr.ptr))
    ms_sort = _timed(ctx, sort_dig, reps=3, ramp=0.1)
    # ---- parity, config 3
    a, b = perms
    want = O.c_minhash_bulk_dense(tok, a, b)
    if not np.array_equal(_download_rows(d_sig, rows, k, np.uint32).astype(np.uint64), want):
        raise SystemExit("PARITY FAILURE (extra.c3_full): signatures differ from the oracle")
    want_dig = lsh_bulk.band_digests(want, bands, r, gpu_mode="disable")  # FNV-1a-64 of the reference's key bytes (lsh.py:537-538), numpy
    for i in range(8):
        keys = O.c_band_keys(want[i: i + 1], bands, r)
        if int(want_dig[i, bands - 1]) != _fnv1a64(keys[0, (bands - 1) * r:].tobytes()):
            raise SystemExit("PARITY FAILURE (extra.c3_full): the numpy digests differ from FNV-1a-64 of the key bytes")
    dig = d_dig.download((bands, n), np.uint64)
    if not np.array_equal(dig[:, rows].T, want_dig):
        raise SystemExit("PARITY FAILURE (extra.c3_full): band digests differ from FNV-1a-64 of the reference's key bytes")
    check_bands = list(range(bands)) if checks == "all" else sorted({0, bands // 3, 2 * bands // 3, bands - 1})
    for j in check_bands:
        sd = d_sd.download((n,), np.uint64, offset=j * n * 8)
        sr = d_sr.download((n,), np.uint32, offset=j * n * 4)
        col = dig[j]
        if np.any(sd[1:] < sd[:-1]) or not np.array_equal(col[sr.astype(np.int64)], sd):
            raise SystemExit(f"PARITY FAILURE (extra.c3_full): band {j} is not the band's digests in ascending order")
        tie = sd[1:] == sd[:-1]
        if np.any(sr[1:][tie] <= sr[:-1][tie]) or np.unique(sr).size != n:
            raise SystemExit(f"PARITY FAILURE (extra.c3_full): band {j}: rows not ascending inside a bucket, or not a permutation")
    radix = None
    if checks == "all":  # the whole output against the stable radix sort (the fallback path), every band
        sd_all, sr_all = d_sd.download((bands, n), np.uint64), d_sr.download((bands, n), np.uint32)
        ctx.set_option("lsh.sort", 1)
        try:
            radix = _timed(ctx, sort_dig, reps=1, ramp=0.0)
            same = np.array_equal(d_sd.download((bands, n), np.uint64), sd_all) and np.array_equal(d_sr.download((bands, n), np.uint32), sr_all)
        finally:
            ctx.set_option("lsh.sort", 0)
        del sd_all, sr_all
        if not same:
            raise SystemExit("PARITY FAILURE (extra.c3_full): the bucketing passes and the stable radix sort disagree")
    for d in (d_sd, d_sr):
        d.free()
    ctx.release_scratch()
    c3 = {
        "workload": f"config 3 at its stated size on one GPU: {n} sets x {t} tokens ({n * t:.3e} tokens), num_perm={k} (uint64 tokens in, uint32 signatures out), "
                    f"band digests ({bands} x {r}, band-major), bucketing of {n * bands} (band, digest) keys",
        "signatures": dict(_roof(n * (8 * t + 4 * k), ms_sig), signatures_per_s=n / (ms_sig * 1e-3)),
        "band_digests": _roof(n * (4 * k + 8 * bands), ms_dig),
        "lsh_sort_digests": dict(_roof(n * (8 * bands + 12 * bands), ms_sort), keys_per_s=n * bands / (ms_sort * 1e-3)),
        "pipeline_ms": ms_sig + ms_dig + ms_sort,
        "corpus_seconds_on_host": gen_s,
        "parity": f"{len(rows)} rows spread over the corpus: signatures vs the C oracle, {bands} digests each vs FNV-1a-64 of the reference's key bytes; "
                  f"sorted bands {check_bands if checks != 'all' else 'all'}: ascending, equal to the digest column gathered by the sorted rows, rows ascending "
                  f"inside every bucket, a permutation" + ("; all bands equal to the stable radix sort's" if radix is not None else ""),
    }
    if radix is not None:
        c3["lsh_sort_digests_radix_ms"] = radix
    # ---- config 5: b = 1 blocks + band digests of the same 10M x 256 matrix
    d_blk, d_dig2 = ctx.alloc(n * nb * 8), ctx.alloc(n * bands * 8)
    fused_flag = []
    fused = lambda: fused_flag.append(ctx.bbit_pack_band_digests_dev(d_sig.ptr, _native.MHX_U32, n, k, 1, bands, r, d_blk.ptr, d_dig2.ptr, BM))
    ms_fused = _timed(ctx, fused, reps=3, ramp=0.1)
    blk_rows = _download_rows(d_blk, rows, nb, np.uint64)
    if not np.array_equal(blk_rows, O.c_bbit_pack(want, 1)):
        raise SystemExit("PARITY FAILURE (extra.c5_full): fused b=1 blocks differ from the oracle")
    if not np.array_equal(d_dig2.download((bands, n), np.uint64), dig):  # (dig: checked against the oracle on the sample rows above)
        raise SystemExit("PARITY FAILURE (extra.c5_full): the fused kernel's digests differ from band_digest_kernel's")
    del dig
    ms_pack = _timed(ctx, lambda: _native.check(lib.mhx_bbit_pack_dev_typed(ctx.handle, d_sig.ptr, _native.MHX_U32, n, k, 1, d_dig.ptr)), reps=3, ramp=0.1)  # (into d_dig: free by now)
    if checks == "all" and not np.array_equal(d_dig.download((n, nb), np.uint64), d_blk.download((n, nb), np.uint64)):
        raise SystemExit("PARITY FAILURE (extra.c5_full): the fused kernel's blocks differ from bbit1_wide_kernel's")
    c5 = {
        "workload": f"config 5 at its stated size on one GPU: b=1 packing of {n} x {k} signatures (uint32) + LSH band hashing ({bands} x {r})",
        "fused": dict(_roof(n * (4 * k + k // 8 + 8 * bands), ms_fused), one_read=bool(fused_flag and all(fused_flag)),
                      kernel="bbit_digest_fused_kernel: blocks and band-major digests from one read of the matrix"),
        "two_kernels_ms": ms_pack + ms_dig,
        "bbit_pack_b1_ms": ms_pack,
        "band_digests_ms": ms_dig,
        "pipeline_ms": ms_fused,
        "parity": f"{len(rows)} spread rows: blocks vs the C oracle (b_bit_minhash.py:82-101), digests vs FNV-1a-64 of the key bytes; all {n} x {bands} "
                  f"digests equal to band_digest_kernel's" + (f"; all {n} x {nb} blocks equal to bbit1_wide_kernel's" if checks == "all" else ""),
    }
    for d in (d_sig, d_dig, d_blk, d_dig2):
        d.free()
    return {"c3_full": c3, "c5_full": c5}


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
    return res



def _fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for byte in data:
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def extra_c4(ctx, n=100_000, dim=4096, s=128):
    """Config 4: WeightedMinHashGenerator(4096, 128, seed=1).minhash_many on X = RandomState(42).uniform(0, 100,
    (100k, 4096)) float32 -- kernel-only on resident input, and from Python (numpy in, numpy out) in parity mode
    (np.log on the host) and device-log mode; the device-log mode's (k, t) mismatches against parity mode are
    counted and gated as BASELINE.md section 3 prescribes."""
    import scipy.sparse as sp

    from datasketch_amd import WeightedMinHashGenerator, _native
    from oracle import oracle as O

    rs = np.random.RandomState(42)
    x = np.empty((n, dim), dtype=np.float32)
    for i in range(0, n, 10_000):  # the same stream as one call; bounds the float64 temporary
        x[i:i + 10_000] = rs.uniform(0, 100, (min(10_000, n - i), dim))
    g = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always", device_log=False)  # np.log on the host, whatever the device could do
    gl = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always", device_log=True)
    ga = WeightedMinHashGenerator(dim, s, seed=1, gpu_mode="always")  # the default: the log on the device where it is numpy's bit for bit
    log_matches = bool(ctx.device_log_matches_numpy()) if hasattr(ctx, "device_log_matches_numpy") else False
    out = {"workload": f"config 4: {n} dense vectors x dim {dim}, sample_size {s}, float32 (the reference's arithmetic type)"}
    # from Python, parity mode and device-log mode
    g.minhash_many_arrays(x[:2048])
    t0 = time.perf_counter()
    hv, ne = g.minhash_many_arrays(x)
    dt_par_first = time.perf_counter() - t0  # takes the page-locked log buffers (kept on the generator) on top
    t0 = time.perf_counter()
    hv, ne = g.minhash_many_arrays(x)
    dt_par = time.perf_counter() - t0
    gl.minhash_many_arrays(x[:2048])
    t0 = time.perf_counter()
    hv_l, ne_l = gl.minhash_many_arrays(x)
    dt_log = time.perf_counter() - t0
    ga.minhash_many_arrays(x[:2048])
    t0 = time.perf_counter()
    hv_a, ne_a = ga.minhash_many_arrays(x)
    dt_auto = time.perf_counter() - t0
    if not (np.array_equal(hv_a, hv) and np.array_equal(ne_a, ne)):
        raise SystemExit("PARITY FAILURE (extra.c4): the default mode (log on the device after the start-up check) differs from the host-log results")
    # kernel only: logs resident on the device (the generator lives on the process-wide context: its stream is the
    # one the events must be recorded on)
    wctx, handle = g._device_handle()
    lib = wctx.lib
    with np.errstate(invalid="ignore", divide="ignore"):
        logs = np.log(x)
    d_x = wctx.to_device(logs)
    d_o = wctx.alloc(n * s * 16)
    d_ne = wctx.alloc(n)
    ms = _timed(wctx, lambda: _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 1, n, d_o.ptr, d_ne.ptr)), reps=5)
    hv_dev = d_o.download((n, s, 2), np.int64)  # the timed entry point's own result
    wctx.set_option("weighted.path", 2)  # A/B: every element evaluated (round 2's kernels), same call
    try:
        ms_every = _timed(wctx, lambda: _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 1, n, d_o.ptr, d_ne.ptr)), reps=1)
        hv_every = d_o.download((n, s, 2), np.int64)
    finally:
        wctx.set_option("weighted.path", 0)
    d_x.upload(x)
    ms_log = _timed(wctx, lambda: _native.check(lib.mhx_weighted_minhash_many_dense_dev(handle, d_x.ptr, 0, n, d_o.ptr, d_ne.ptr)), reps=5)
    for d in (d_x, d_o, d_ne):
        d.free()
    # parity gates: 2 048 rows spread over the matrix against the C oracle (bit-exact (k, t) in parity mode), and EVERY
    # row of the walk against the kernels that evaluate every element
    rows = np.unique(np.linspace(0, n - 1, 2048).astype(np.int64))
    csr = sp.csr_matrix(x[rows])
    csr.sort_indices()
    t0 = time.perf_counter()
    wo, wn = O.c_weighted_minhash_many(csr.indptr, csr.indices, csr.data, g.rs, g.ln_cs, g.betas)
    oracle_s = time.perf_counter() - t0
    if not (np.array_equal(hv[rows], wo) and np.array_equal(ne[rows], wn) and np.array_equal(hv_dev[rows], wo)):
        raise SystemExit("PARITY FAILURE (extra.c4): weighted (k, t) differ from the oracle in parity mode")
    if not np.array_equal(hv_dev, hv_every):
        raise SystemExit("PARITY FAILURE (extra.c4): the walk and the evaluate-every-element kernels disagree")
    cpu = weighted_cpu_baseline(x, g, wo[:64] if np.array_equal(rows[:64], np.arange(64)) else None, oracle_rows=len(rows), oracle_s=oracle_s)
    # fast-mode acceptance gate: every (k, t) mismatch must come from two smallest ln_a
    # within 1e-6 relative of each other
    mism = np.argwhere(np.any(hv != hv_l, axis=2))
    gate = weighted_gap_gate(x, g, hv, hv_l, mism[:5000])
    if gate["unexplained"]:
        raise SystemExit(f"PARITY FAILURE (extra.c4): {gate['unexplained']} device-log mismatches outside the 1e-6 ln_a tolerance")
    alg = n * (4 * dim + 16 * s)
    out.update({
        # the reference's function takes VALUES (weighted_minhash.py:212 takes np.log itself): that is config 4's primary number
        "kernel": dict(_roof(alg, ms_log), vectors_per_s=n / (ms_log * 1e-3), element_evaluations_per_s=n * dim * s / (ms_log * 1e-3),
                       kernels="walk_plan_kernel + walk_build_kernel (no-op once the tables stand) + weighted_walk_wave_kernel<values in>",
                       note="VALUES in, as the reference's minhash_many takes them: the device takes numpy's float32 log (np_logf) of the entries a walk "
                            "meets; bound-ordered walk: ~2 exact evaluations per (row, sample) instead of 4096 (the element rate counts the "
                            "evaluations the reference makes); the matrix is read once; algorithmic bytes = 4*dim + 16*S per vector"),
        "kernel_logs_in": dict(_roof(alg, ms), vectors_per_s=n / (ms * 1e-3),
                               note="the same with np.log of the matrix precomputed and resident (what the host-log parity mode hands over)"),
        "kernel_every_element": dict(_roof(alg, ms_every), vectors_per_s=n / (ms_every * 1e-3),
                                     note="weighted.path=2: round 2's kernels (every element evaluated), same call, same box"),
        "from_python_parity_mode": {"seconds": dt_auto, "vectors_per_s": n / dt_auto, "log_taken_on": "device" if log_matches else "host",
                                    "note": "numpy in -> numpy out, the default mode: (k, t) bit-identical to the reference; the log is taken on the device when "
                                            "its float32 log reproduces this host's np.log on the start-up sentinels (device_log_matches_numpy), else on the host; "
                                            "equal to the host-log results on all rows (checked above)"},
        "from_python_host_log": {"seconds": dt_par, "vectors_per_s": n / dt_par, "first_call_seconds": dt_par_first,
                                 "note": "device_log=False: np.log on the host; first call = with the one-time allocation of the page-locked log buffers"},
        "device_log_matches_numpy": log_matches,
        "from_python_device_log": {"seconds": dt_log, "vectors_per_s": n / dt_log},
        "device_log_mismatch_rate": float(len(mism)) / (n * s),
        "device_log_mismatches": int(len(mism)),
        "device_log_gate": gate,
        "cpu_baseline": cpu,
        "parity": f"{len(rows)} rows bit-exact (k, t) vs the C oracle in parity mode; all {n} rows equal to the evaluate-every-element "
                  f"kernels; all-rows nonempty = {bool(ne.all())}",
    })
    return out


def weighted_cpu_baseline(x, g, want64, oracle_rows, oracle_s):
    """The reference's CPU paths for config 4 on a bounded sample of its input, one core (numpy's float32 ufuncs are
    single-threaded): `minhash_many` as the reference evaluates it (weighted_minhash.py:205-239: per row the (S, nnz)
    arrays) through this package's gpu_mode='disable' path (the same numpy statements), the per-vector `minhash` loop
    (weighted_minhash.py:123-159; SURVEY.md section 8d: the reference's faster CPU alternative), and the scalar C
    oracle.  kind 'port': the reference itself is not on the GPU box."""
    from datasketch_amd import WeightedMinHashGenerator
    from oracle import oracle as O
    import scipy.sparse as sp

    gd = WeightedMinHashGenerator(g.dim, g.sample_size, seed=g.seed, gpu_mode="disable")
    m = 64
    t0 = time.perf_counter()
    res = gd.minhash_many(x[:m])
    dt_many = time.perf_counter() - t0
    if want64 is None:
        c = sp.csr_matrix(x[:m])
        c.sort_indices()
        want64 = O.c_weighted_minhash_many(c.indptr, c.indices, c.data, g.rs, g.ln_cs, g.betas)[0]
    if not np.array_equal(np.stack([r.hashvalues for r in res]), want64):
        raise SystemExit("PARITY FAILURE (extra.c4): the numpy path differs from the oracle on the cpu_baseline sample")
    t0 = time.perf_counter()
    for v in x[:m]:
        gd.minhash(v)
    dt_each = time.perf_counter() - t0
    return {"value": m / dt_many, "unit": "vectors/s", "cores": 1, "kind": "port",
            "sample": f"first {m} rows of config 4's input (dim {g.dim}, sample_size {g.sample_size}): minhash_many, numpy, {dt_many:.1f} s",
            "per_vector_minhash_loop_value": m / dt_each,
            "per_vector_minhash_loop_sample": f"the same {m} rows through minhash() one at a time, {dt_each:.1f} s",
            "c_oracle_value": oracle_rows / oracle_s, "c_oracle_sample": f"{oracle_rows} rows, scalar C, {oracle_s:.1f} s",
            "cpu_model": cpu_model()}


def weighted_gap_gate(x, g, hv_par, hv_log, mism, tol=1e-6):
    """BASELINE.md section 3's acceptance rule for the device-log mode: a (k, t) pair may differ from parity mode
    only where the choice was within rounding -- the two competing columns' ln_a (float32, the reference's formula,
    weighted_minhash.py:212-218) within `tol` relative of each other, or (same effect one step earlier) a column's
    ln(x)/r + beta within `tol` relative of an integer, where one ulp of the log moves the floor.  Returns the
    largest relative gap seen among the accepted mismatches and the number of mismatches neither rule explains."""
    worst_gap, worst_edge, unexplained, edge_cases = None, None, 0, 0
    one = np.float32(1)
    for row, smp in mism:
        k0, k1 = int(hv_par[row, smp, 0]), int(hv_log[row, smp, 0])
        ln_a, edge = [], []
        for kk in (k0, k1):
            lg = np.log(np.float32(x[row, kk]))
            r, be, lc = g.rs[smp, kk], g.betas[smp, kk], g.ln_cs[smp, kk]
            y = np.float32(np.float32(lg / r) + be)
            tt = np.floor(y)
            ln_a.append(float(np.float32(lc - np.float32(np.float32(np.float32(tt - be) + one) * r))))
            edge.append(float(min(y - tt, tt + one - y)) / max(abs(float(y)), 1.0))
        if min(edge) <= tol:  # a floor boundary: t (and with it ln_a) flips with the last bit of the log
            edge_cases += 1
            worst_edge = min(edge) if worst_edge is None else max(worst_edge, min(edge))
            continue
        gap = abs(ln_a[0] - ln_a[1]) / max(abs(ln_a[0]), abs(ln_a[1]), 1e-30)
        if k0 != k1 and gap <= tol:
            worst_gap = gap if worst_gap is None else max(worst_gap, gap)
        else:
            unexplained += 1
    return {"mismatches_examined": int(len(mism)), "worst_relative_ln_a_gap": worst_gap, "floor_boundary_cases": edge_cases,
            "worst_floor_boundary_distance": worst_edge, "unexplained": unexplained, "tolerance": tol,
            "rule": "BASELINE.md section 3: (k,t) may differ from parity mode only where the two smallest ln_a "
                    "(or ln(x)/r+beta and an integer) are within 1e-6 relative"}


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
