#!/usr/bin/env python3
"""tools/pcie_probe.py -- what the host<->device path can do on this box: pageable vs registered (pinned in
place) numpy buffers, H2D / D2H, 2 GiB.  Decides whether mhx_minhash_bulk should pin caller buffers."""
import ctypes
import time

import numpy as np

hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
H2D, D2H = 1, 2
n = 2 << 30
host = np.ones(n, dtype=np.uint8)
dev = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(dev), n) == 0
hip.hipDeviceSynchronize()


def timed(kind, label):
    for rep in range(2):
        t0 = time.perf_counter()
        a, b = (dev, host.ctypes.data) if kind == H2D else (host.ctypes.data, dev)
        assert hip.hipMemcpy(a, b, n, kind) == 0
        hip.hipDeviceSynchronize()
        dt = time.perf_counter() - t0
        print(f"{label:32s} rep {rep}: {dt*1e3:8.1f} ms  {n/dt/1e9:6.1f} GB/s", flush=True)


timed(H2D, "H2D pageable")
timed(D2H, "D2H pageable")
t0 = time.perf_counter()
rc = hip.hipHostRegister(host.ctypes.data, n, 0)
print(f"hipHostRegister rc={rc}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
if rc == 0:
    timed(H2D, "H2D registered")
    timed(D2H, "D2H registered")
    t0 = time.perf_counter()
    hip.hipHostUnregister(host.ctypes.data)
    print(f"hipHostUnregister: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
