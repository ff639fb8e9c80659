#!/usr/bin/env python3
"""bench.py -- MinHash signatures/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1M sets x 256 tokens, num_perm=128 per GPU, synthetic
pre-hashed tokens ``RandomState(42+rank).randint(0, 2**32, (N, T), uint64)``, seed=1.  A "step"
is one pass of the hot path (``mhx_minhash_bulk_dev``) over the whole resident corpus, producing
the [N, K] uint64 signature matrix in HBM.  Inputs are in HBM before the timed region starts.
Weak scaling: every rank hashes its own 1M-set shard; there is no collective in the data path
(``--allgather`` adds the RCCL all-gather of the shards after every step, the config-3 shape).

Prints ONE JSON line on rank 0.  ``roofline.achieved`` = algorithmic bytes (8*T + 8*K per
signature) / average launch duration measured with HIP events on the
kernel's own stream.  ``cpu_baseline`` = the numpy restatement of the reference's CPU path
(oracle/, kind "port") timed on a bounded sample on this host, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sets", type=int, default=1_000_000, help="sets per GPU")
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--num-perm", type=int, default=128)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--allgather", action="store_true", help="RCCL all-gather of the shards inside every step")
    ap.add_argument("--check-rows", type=int, default=4096, help="rows verified against the oracle")
    ap.add_argument("--cpu-sample", type=int, default=160_000, help="sets timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive host->host measurement")
    ap.add_argument("--u32", action="store_true", help="compact variant: uint32 tokens in, uint32 signatures out")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="libmhx tuning knob (mhx_ctx_set_option), e.g. blocks_per_cu=4")
    return ap.parse_args()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus N>1 must be launched with `python -m torch.distributed.run --nproc-per-node N`")

    # libmhx (system ROCm runtime) is loaded before torch so that both share one HIP runtime.
    from datasketch_amd import _native
    from datasketch_amd.minhash import MinHash

    if args.allgather:  # RCCL must enter the process before torch's own ROCm runtime does (mhx_ctx_create loads it)
        os.environ.setdefault("MHX_PRELOAD_RCCL", "1")
    visible = _native.device_count()
    if visible < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    # one GPU per rank: LOCAL_RANK indexes the visible devices; a launcher that already narrowed the
    # visibility to one device per process (HIP_VISIBLE_DEVICES) leaves device 0
    ctx = _native.Context(local_rank if local_rank < visible else local_rank % visible)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        ctx.set_option(key, int(val))

    dist = None
    torch = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        import torch  # noqa: F811  (plumbing only: rendezvous, barrier, max-over-ranks)
        import torch.distributed as dist  # noqa: F811

        # gloo: PyTorch never touches the GPU here; RCCL traffic (--allgather) goes through libmhx
        backend = os.environ.get("MHX_BENCH_BACKEND", "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    def sync():
        ctx.synchronize()
        if torch is not None and torch.cuda.is_initialized():
            torch.cuda.synchronize()

    n, t, k = args.sets, args.tokens, args.num_perm
    proto = MinHash(num_perm=k, seed=args.seed, hashfunc=lambda x: x)
    perms = proto.permutations

    # ---- synthetic corpus, resident in HBM before timing
    rng = np.random.RandomState(42 + rank)
    tokens = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
    if args.u32:
        d_tok = ctx.to_device(tokens.astype(np.uint32))
        tok_dtype, out_dtype, out_np, tok_bytes, out_bytes = _native.MHX_U32, _native.MHX_U32, np.uint32, 4, 4
    else:
        d_tok = ctx.to_device(tokens)
        tok_dtype, out_dtype, out_np, tok_bytes, out_bytes = _native.MHX_U64, _native.MHX_U64, np.uint64, 8, 8
    d_out = ctx.alloc(n * k * out_bytes)
    ctx.perm_handle(perms)

    gather = None
    if args.allgather:
        gather = setup_allgather(ctx, dist, torch, d_out, n * k * out_bytes, world, rank)

    def step():
        ctx.minhash_bulk_dev(perms, d_tok.ptr, tok_dtype, None, t, n, n * t, None, 0, d_out.ptr, out_dtype)
        if gather is not None:
            gather()

    for _ in range(args.warmup):
        step()
    sync()

    # ---- timed region: exactly K steps, barrier + sync on both sides, max over ranks
    evs = [ctx.event() for _ in range(args.steps + 1)]
    barrier()
    sync()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0  # this rank's K steps, device work complete
    barrier()                           # closing barrier; the job's time is the MAX over ranks below
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    launch_ms = [evs[i].elapsed_ms(evs[i + 1]) for i in range(args.steps)]
    kernel_ms = float(np.mean(launch_ms))

    # ---- one counted launch outside the timed region: how often the sieve's proof failed
    ctx.counters(True)
    ctx.minhash_bulk_dev(perms, d_tok.ptr, tok_dtype, None, t, n, n * t, None, 0, d_out.ptr, out_dtype)
    counters = ctx.counters(False)

    # ---- parity: rows spread over the whole matrix against the package's numpy path (gpu_mode="disable": the
    # reference's arithmetic, minhash.py:293-297), bit-exact or fail.  The oracle itself is only used in the
    # cpu_baseline leg below, which also compares its rows with the GPU's.
    from datasketch_amd.hashfunc import prehashed

    a, b = perms
    check = max(0, min(args.check_rows, n))
    sig_head = None
    if check or (args.cpu_sample > 0 and rank == 0 and world == 1):
        sig = d_out.download((n, k), out_np)
        if check:
            rows = np.unique(np.linspace(0, n - 1, check).astype(np.int64))
            want = MinHash.bulk_signatures(tokens[rows], num_perm=k, seed=args.seed, hashfunc=prehashed, gpu_mode="disable")
            if not np.array_equal(sig[rows].astype(np.uint64), want):
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
            "parallelism": f"shard{world}" + ("+allgather" if args.allgather else ""),
            "parity_rows_checked": int(check),
            "sets_redone_by_full_evaluation": counters["sieve_sets_redone"],
            "sets_redone_by_exact_fold": counters["exact_sets_redone"],
        },
    }
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
        "kernel": "minhash_bulk_kernel<MODE_SIEVE> (+ the MODE_FULL launch over the flagged sets)",
        "kernel_ms": kernel_ms,
        "algorithmic_bytes_per_launch": alg_bytes,
        "note": "integer-VALU-bound kernel: (token,perm) pair evaluations/s = %.3e" % pair_rate,
    }

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
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(tokens, a, b, min(args.cpu_sample, n), k, t, sig_head, seed=args.seed)
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def measured_traffic(n, t, k, args):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_traffic_minhash_bulk.json); None for any other shape."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_minhash_bulk.json")
    if (n, t, k) != (1_000_000, 256, 128) or args.u32 or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("traffic_bytes_per_launch")


def _cpu_worker(args):
    """One host core of the all-cores baseline: the numpy per-set loop of MinHash.bulk on its own
    shard (generated in the worker: nothing but a checksum travels)."""
    seed, n, t, k, pseed = args
    sys.path.insert(0, ROOT)
    from oracle import oracle as O

    a, b = O.np_init_permutations(k, pseed)
    tokens = np.random.RandomState(seed).randint(0, 2**32, size=(n, t), dtype=np.uint64)
    t0 = time.perf_counter()
    sig = O.np_minhash_bulk(list(tokens), a, b)
    return time.perf_counter() - t0, int(sig[:, 0].sum())


def _usable_cores(cap=64):
    """Cores this process may really use: affinity mask, clipped by the cgroup CPU quota (a container
    can see 256 CPUs and own 8) and by `cap` (start-up of hundreds of interpreters is not the point)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


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
    c0 = time.perf_counter()
    want = O.c_minhash_bulk_dense(tokens[:single], a, b)
    cdt = time.perf_counter() - c0
    assert np.array_equal(got, want)
    m = 0 if gpu_rows is None else min(single, len(gpu_rows))
    if m and not np.array_equal(got[:m], gpu_rows[:m]):
        raise SystemExit("PARITY FAILURE: GPU signatures differ from the oracle on the cpu_baseline sample")
    cores = _usable_cores()
    per = max(2_000, sample // 8)  # sets per process: 1.5-3 s of numpy each, 41 MB of tokens
    out = {
        "value": single / dt,
        "unit": "signatures/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {single} sets of the benchmark corpus ({t} tokens, num_perm={k}), numpy per-set loop as MinHash.bulk; {dt:.1f} s",
    }
    try:
        w0 = time.perf_counter()
        with mp.get_context("spawn").Pool(cores) as pool:  # spawn: children never see the HIP runtime
            res = pool.map_async(_cpu_worker, [(1000 + i, per, t, k, seed) for i in range(cores)]).get(timeout=90)
        wall = time.perf_counter() - w0
        busy = max(r[0] for r in res)
        out.update({
            "value": cores * per / busy,
            "cores": cores,
            "sample": f"{cores} processes x {per} sets of the same shape ({t} tokens, num_perm={k}), numpy per-set loop as "
                      f"MinHash.bulk; slowest process {busy:.1f} s (pool wall {wall:.1f} s incl. start-up)",
        })
    except Exception as e:  # the all-cores leg is best effort; the single-core figure stands
        out["all_cores_error"] = repr(e)
    out.update({
        "single_core_value": single / dt,
        "single_core_sample": f"first {single} sets of the benchmark corpus, {dt:.1f} s",
        "host_cpus": os.cpu_count(),
        "c_oracle_single_core_value": single / cdt,
        "rows_equal_to_gpu": int(m),
    })
    return out


def setup_allgather(ctx, dist, torch, d_out, shard_bytes, world, rank):
    """All-gather of the signature shards with RCCL through libmhx's own binding (mhx_comm_*,
    include/mhx.h): enqueued on the kernel's stream, so it starts the moment the shard is complete.
    torch.distributed (gloo, CPU) only carries the 128-byte RCCL id from rank 0 to the others."""
    if dist is None:
        raise SystemExit("--allgather needs a torch.distributed launch (torchrun), also for 1 GPU")
    from datasketch_amd import _native

    box = [_native.Communicator.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = _native.Communicator(ctx, box[0], rank, world)
    d_all = ctx.alloc(shard_bytes * world)

    def gather():
        comm.allgather_dev(d_out.ptr, d_all.ptr, shard_bytes)

    gather.keepalive = (comm, d_all)
    return gather


if __name__ == "__main__":
    main()
