"""bench.py's cpu_baseline leg: the reference's CPU path (or its pinned numpy restatement) timed on this host's cores."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np

from benchmarks.common import ROOT

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
        # the real reference timed beside the restatement: measured HERE when DATASKETCH_REFERENCE resolves; otherwise no ratio is
        # claimed in this line -- the build container's measurement is a file (tools/cpu_reference_vs_port.py), named, not replayed
        **({"reference_over_port_time": ref_dt / dt, "reference_over_port_measured": "in this run"} if is_ref else
           {"reference_over_port_file": "profiles/r05_cpu_reference_vs_port.txt"}),
        "port_single_core_value": single / dt,
        "note": ("kind 'reference' = ekzhu/datasketch's own MinHash.bulk imported from DATASKETCH_REFERENCE, rows equal to the restatement's and the GPU's"
                 if is_ref else
                 "kind 'port' = the numpy restatement of MinHash.bulk's per-set loop, pinned to the reference by tests/golden and row-compared with the "
                 "GPU here; the reference itself is not on this box (DATASKETCH_REFERENCE unset), so nothing is said about its own rate in this line"),
        "c_oracle_single_core_value": single / cdt,
        "rows_equal_to_gpu": int(m),
    })
    return out


# ------------------------------------------------------------------------------------------------
