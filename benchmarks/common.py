"""Shared by bench.py and the benchmarks/ modules: peaks, HIP-event timing, the roofline record, row downloads."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
XGMI_LINKS, XGMI_GBPS_PER_LINK = 7, 153.0  # 7 point-to-point links per GPU


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


def _download_rows(buf, rows, width, dtype):
    """The given rows of a row-major device matrix (one small copy per row: a few thousand rows of a 10 GB matrix)."""
    item = np.dtype(dtype).itemsize
    return np.stack([buf.download((width,), dtype, offset=int(r) * width * item) for r in rows]) if len(rows) else np.empty((0, width), dtype)


def _fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for byte in data:
        h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def kernel_stamp() -> dict:
    """What ties a committed counter profile to the binary a bench run times: sha256 of the headline kernel's translation unit
    (csrc/minhash_kernels.hip + mhx_internal.h) and of libmhx.so as loaded.  tools/traffic_summary.py writes it into
    profiles/r0N_traffic_minhash_bulk.json on the box that took the counters; bench.py replays `traffic` / `valu_issue_frac` only
    when the kernel-source hash of the file equals the one of the tree it runs from."""
    import hashlib

    def sha(paths):
        h = hashlib.sha256()
        for p in paths:
            with open(p, "rb") as f:
                h.update(f.read())
        return h.hexdigest()

    csrc = os.path.join(ROOT, "datasketch_amd", "csrc")
    out = {"kernel_source_sha256": sha([os.path.join(csrc, "minhash_kernels.hip"), os.path.join(csrc, "mhx_internal.h")]),
           "kernel_source_files": ["datasketch_amd/csrc/minhash_kernels.hip", "datasketch_amd/csrc/mhx_internal.h"]}
    lib = os.environ.get("MHX_LIBRARY") or os.path.join(ROOT, "datasketch_amd", "libmhx.so")
    out["libmhx_sha256"] = sha([lib]) if os.path.exists(lib) else None
    return out
