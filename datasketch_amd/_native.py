"""ctypes binding of libmhx.so (include/mhx.h) -- the only door from Python to the HIP kernels.

No PyTorch, no CuPy: numpy arrays in, numpy arrays out, raw pointers across the C ABI.
Product code: this module never imports anything from ``oracle/``; when the shared library or
a device is missing it raises, it does not fall back to a CPU implementation.
"""
from __future__ import annotations

import ctypes
import weakref
import os
import threading
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MHX_LIBRARY") or os.path.join(_HERE, "libmhx.so")  # MHX_LIBRARY: A/B builds (tools/)

MHX_OK, MHX_ERR_NO_DEVICE, MHX_ERR_INVALID, MHX_ERR_HIP, MHX_ERR_OOM, MHX_ERR_UNSUPPORTED, MHX_ERR_COMM = range(7)
MHX_U64, MHX_U32 = 0, 1
COMM_ID_BYTES = 128
ROW_MAJOR, BAND_MAJOR = 0, 1  # layouts of a band-digest matrix on the device (include/mhx.h)

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_int = ctypes.c_int
_sz = ctypes.c_size_t

# name -> argtypes; every function returns int unless listed in _RESTYPE
_PROTOTYPES = {
    "mhx_device_count": [ctypes.POINTER(_int)],
    "mhx_ctx_create": [_int, ctypes.POINTER(_vp)],
    "mhx_ctx_destroy": [_vp],
    "mhx_ctx_synchronize": [_vp],
    "mhx_ctx_release_scratch": [_vp],
    "mhx_ctx_device_info": [_vp, ctypes.c_char_p, _int, ctypes.POINTER(_int), ctypes.POINTER(_i64)],
    "mhx_ctx_set_option": [_vp, ctypes.c_char_p, _i64],
    "mhx_ctx_counters": [_vp, _int, ctypes.POINTER(ctypes.c_uint64)],
    "mhx_ctx_minhash_mode": [_vp, _int, ctypes.POINTER(_int)],
    "mhx_ctx_minhash_flags": [_vp, _i64, _vp],
    "mhx_dev_alloc": [_vp, _sz, ctypes.POINTER(_vp)],
    "mhx_dev_free": [_vp, _vp],
    "mhx_debug_guard_alloc": [_int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "mhx_debug_poison_alloc": [_int],
    "mhx_host_alloc": [_vp, _sz, ctypes.POINTER(_vp)],
    "mhx_host_free": [_vp, _vp],
    "mhx_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "mhx_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "mhx_memcpy_d2d": [_vp, _vp, _vp, _sz],
    "mhx_memset_dev": [_vp, _vp, _int, _sz],
    "mhx_event_create": [_vp, ctypes.POINTER(_vp)],
    "mhx_event_record": [_vp],
    "mhx_event_synchronize": [_vp],
    "mhx_event_elapsed_ms": [_vp, _vp, ctypes.POINTER(ctypes.c_float)],
    "mhx_event_destroy": [_vp],
    "mhx_perm_create": [_vp, _vp, _vp, _i32, ctypes.POINTER(_vp)],
    "mhx_perm_destroy": [_vp],
    "mhx_minhash_bulk_dev": [_vp, _vp, _int, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _int],
    "mhx_minhash_bulk": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp],
    "mhx_minhash_bulk_typed": [_vp, _vp, _int, _vp, _i64, _i64, _vp, _i64, _vp, _int],
    "mhx_sha1_tokens_dev": [_vp, _vp, _vp, _i64, _int, _vp],
    "mhx_sha1_tokens": [_vp, _vp, _vp, _i64, _int, _vp],
    "mhx_minhash_bulk_bytes": [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp],
    "mhx_minhash_bulk_bytes_typed": [_vp, _vp, _vp, _i64, _int, _vp, _i64, _vp, _i64, _vp],
    "mhx_minhash_update_batch": [_vp, _vp, _i64, _vp],
    "mhx_minhash_merge_dev": [_vp, _vp, _vp, _i64, _vp],
    "mhx_minhash_merge": [_vp, _vp, _vp, _i64, _vp],
    "mhx_wgen_create": [_vp, _vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_vp)],
    "mhx_wgen_destroy": [_vp],
    "mhx_weighted_minhash_many": [_vp, _vp, _vp, _vp, _int, _i64, _vp, _vp],
    "mhx_weighted_minhash_many_dense": [_vp, _vp, _int, _i64, _vp, _vp],
    "mhx_weighted_minhash_many_dense_dev": [_vp, _vp, _int, _i64, _vp, _vp],
    "mhx_weighted_minhash_many_dev": [_vp, _vp, _vp, _vp, _int, _i64, _i64, _vp, _vp],
    "mhx_weighted_logf": [_vp, _vp, _i64, _vp],
    "mhx_weighted_dense_begin": [_vp, _int, _i64, ctypes.POINTER(_vp)],
    "mhx_weighted_dense_feed": [_vp, _vp, _i64, _vp, _vp],
    "mhx_weighted_dense_end": [_vp],
    "mhx_bbit_num_blocks": [_i32, _i32, ctypes.POINTER(_i32)],
    "mhx_bbit_pack_dev": [_vp, _vp, _i64, _i32, _i32, _vp],
    "mhx_bbit_pack": [_vp, _vp, _i64, _i32, _i32, _vp],
    "mhx_band_keys_dev": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "mhx_band_keys": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "mhx_band_digests_dev": [_vp, _vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp],
    "mhx_band_digests": [_vp, _vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp],
    "mhx_lsh_sort_bands_dev": [_vp, _vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp],
    "mhx_lsh_sort_bands": [_vp, _vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp],
    "mhx_lsh_sort_digests_dev": [_vp, _vp, _i64, ctypes.c_int32, _vp, _vp],
    "mhx_lsh_candidate_pairs_dev": [_vp, _vp, _vp, _i64, ctypes.c_int32, _vp, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "mhx_lsh_candidate_pairs": [_vp, _vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _i64,
                                ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "mhx_lsh_query_dev": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _int, _i32, _i64, _vp, _i64, ctypes.POINTER(_i64)],
    "mhx_jaccard_pairs_dev": [_vp, _vp, _vp, ctypes.c_int32, _vp, _i64, _vp],
    "mhx_jaccard_pairs": [_vp, _vp, _i64, ctypes.c_int32, _vp, _i64, _vp],
    "mhx_bbit_pack_dev_typed": [_vp, _vp, _int, _i64, _i32, _i32, _vp],
    "mhx_band_digests_dev_typed": [_vp, _vp, _int, _i64, _i32, _i32, _i32, _vp],
    "mhx_bbit_pack_band_digests_dev": [_vp, _vp, _int, _i64, _i32, _i32, _i32, _i32, _int, _vp, _vp, ctypes.POINTER(_int)],
    "mhx_band_digests_layout_dev": [_vp, _vp, _int, _i64, _i32, _i32, _i32, _int, _vp],
    "mhx_lsh_sort_digests_layout_dev": [_vp, _vp, _i64, ctypes.c_int32, _int, _vp, _vp],
    "mhx_lsh_sort_bands_dev_typed": [_vp, _vp, _int, _i64, _i32, _i32, _i32, _vp, _vp],
    "mhx_jaccard_pairs_dev_typed": [_vp, _vp, _vp, _int, _i32, _vp, _i64, _vp],
    "mhx_bbit_jaccard_pairs_dev": [_vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp],
    "mhx_bbit_jaccard_pairs": [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp],
    "mhx_lean_serialize_dev": [_vp, _vp, _i64, _i32, _i64, _vp],
    "mhx_lean_serialize": [_vp, _vp, _i64, _i32, _i64, _vp],
    "mhx_lean_serialize_dev_typed": [_vp, _vp, _int, _i64, _i32, _i64, _int, _vp],
    "mhx_lean_deserialize_dev": [_vp, _vp, _i64, _i32, _int, _int, _vp, _vp, _vp],
    "mhx_lean_deserialize": [_vp, _vp, _i64, _i32, _int, _vp, _vp],
    "mhx_bbit_unpack_dev": [_vp, _vp, _i64, _i32, _i32, _vp],
    "mhx_bbit_unpack": [_vp, _vp, _i64, _i32, _i32, _vp],
    "mhx_comm_unique_id": [_vp],
    "mhx_comm_create": [_vp, _vp, _int, _int, ctypes.POINTER(_vp)],
    "mhx_comm_destroy": [_vp],
    "mhx_comm_info": [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)],
    "mhx_comm_allgather_dev": [_vp, _vp, _vp, _sz],
    "mhx_comm_allgatherv_dev": [_vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)],
    "mhx_comm_exchange_dev": [_vp, _vp, _vp, _i32, ctypes.POINTER(_i32), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                              _i32, ctypes.POINTER(_i32), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)],
}
_RESTYPE = {"mhx_last_error": ctypes.c_char_p, "mhx_version": ctypes.c_char_p}

EXPORTED_SYMBOLS = sorted(list(_PROTOTYPES) + list(_RESTYPE))


try:  # CPython helper (csrc/pack_module.c): ~10 ns per token instead of ~180 in the interpreter
    from datasketch_amd import _mhxpack
except ImportError:  # not built: the pure-Python packer below does the same job
    _mhxpack = None


class MhxError(RuntimeError):
    """A libmhx call failed (HIP / RCCL / allocation)."""


_lib = None
_lib_error: Optional[str] = None
_lock = threading.Lock()


def gpu_node_present() -> bool:
    """True when this host exposes an AMD GPU compute node (so a missing library is a bug, not a CPU box)."""
    return os.path.exists("/dev/kfd")


def load():
    """Load libmhx.so (built in-tree by datasketch_amd/csrc/build.sh).  Raises if it is missing."""
    global _lib, _lib_error
    with _lock:
        if _lib is not None:
            return _lib
        if _lib_error is not None:
            raise MhxError(_lib_error)
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            _lib_error = (
                f"libmhx.so could not be loaded from {LIB_PATH}: {e}. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or datasketch_amd/csrc/build.sh"
            )
            raise MhxError(_lib_error) from e
        for name, argtypes in _PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _int
        for name, restype in _RESTYPE.items():
            fn = getattr(lib, name)
            fn.argtypes = []
            fn.restype = restype
        _lib = lib
        return lib


def last_error() -> str:
    return load().mhx_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a libmhx status to the exception the reference's API would raise."""
    if rc == MHX_OK:
        return
    msg = last_error()
    if rc == MHX_ERR_INVALID:
        raise ValueError(msg)
    if rc == MHX_ERR_OOM:
        raise MemoryError(msg)
    if rc == MHX_ERR_NO_DEVICE:
        raise RuntimeError(msg)
    raise MhxError(f"libmhx error {rc}: {msg}")


_device_count: Optional[int] = None


def device_count() -> int:
    global _device_count
    if _device_count is None:
        n = _int(0)
        check(load().mhx_device_count(ctypes.byref(n)))
        _device_count = n.value
    return _device_count


def gpu_available() -> bool:
    """Counterpart of the reference's ``_gpu_available`` (datasketch/minhash.py:38-48), strict: False on a host
    without an AMD GPU; on a GPU host a missing / unloadable libmhx.so RAISES.  This is what ``gpu_mode='always'``
    asks (the HIP path must never be skipped silently where it was demanded); ``'detect'`` asks :func:`gpu_detected`."""
    try:
        return device_count() > 0
    except MhxError:
        if gpu_node_present():
            raise
        return False


_detect_warned = False


def gpu_detected() -> bool:
    """``gpu_mode='detect'``: the reference's semantics (datasketch/minhash.py:272-279) -- use the device when there is
    one, else fall back to the numpy path without failing.  The one addition: on a host that HAS an AMD GPU
    (``/dev/kfd``) a library that does not load is a broken installation, so the fallback is announced with a
    ``RuntimeWarning`` naming the load error (once per process) instead of passing in silence."""
    global _detect_warned
    try:
        return device_count() > 0
    except MhxError as e:
        if gpu_node_present() and not _detect_warned:
            _detect_warned = True
            import warnings

            warnings.warn(f"datasketch_amd: this host has an AMD GPU but the HIP path is unavailable ({e}); gpu_mode='detect' falls back to "
                          "the numpy path.  gpu_mode='always' makes this an error.", RuntimeWarning, stacklevel=3)
        return False


def _ptr(arr: Optional[np.ndarray]):
    return None if arr is None else arr.ctypes.data


class WeightedFeed:
    """``with ctx.weighted_dense_feed(...) as feed: feed.feed(x_piece, out_piece, nonempty_piece)`` -- the upload of a
    piece overlaps the evaluation of the one before and the download of the one before that.  ``x_piece`` may be
    overwritten as soon as ``feed`` returns; ``out_piece`` / ``nonempty_piece`` are complete after the next ``feed`` or
    on leaving the ``with`` block."""

    def __init__(self, ctx: "Context", h: int, sample_size: int, dim: int, values_are_logs: bool, piece_rows: int):
        self.ctx, self.s, self.dim, self.piece_rows = ctx, int(sample_size), int(dim), int(piece_rows)
        p = _vp()
        check(ctx.lib.mhx_weighted_dense_begin(h, int(bool(values_are_logs)), self.piece_rows, ctypes.byref(p)))
        self.handle = p.value

    def feed(self, x: np.ndarray, out: np.ndarray, nonempty: np.ndarray) -> None:
        n = x.shape[0]
        if x.dtype != np.float32 or x.ndim != 2 or x.shape[1] != self.dim or not x.flags.c_contiguous:
            raise ValueError("x must be a C-contiguous float32 array of shape (rows, dim)")
        if out.shape != (n, self.s, 2) or out.dtype != np.int64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous int64 array of shape (rows, sample_size, 2)")
        if nonempty.shape != (n,) or nonempty.dtype != np.uint8 or not nonempty.flags.c_contiguous:
            raise ValueError("nonempty must be a C-contiguous uint8 array of shape (rows,)")
        if self.handle is None:
            raise MhxError("the feed has ended")
        check(self.ctx.lib.mhx_weighted_dense_feed(self.handle, _ptr(x), n, _ptr(out), _ptr(nonempty)))

    def end(self) -> None:
        h, self.handle = self.handle, None
        if h is not None:
            check(self.ctx.lib.mhx_weighted_dense_end(h))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.end()
        else:  # release the buffers, keep the original exception
            h, self.handle = self.handle, None
            if h is not None:
                self.ctx.lib.mhx_weighted_dense_end(h)
        return False

    def __del__(self):
        try:
            h, self.handle = self.handle, None
            if h is not None and self.ctx.handle is not None:  # (a closed context has taken its generators along)
                self.ctx.lib.mhx_weighted_dense_end(h)
        except Exception:
            pass


def guard_alloc(align: int):
    """Debugging (mhx_debug_guard_alloc): from now on every device allocation of the library in this process abuts an
    unmapped page -- behind its last byte for align > 0 (size rounded up to ``align`` bytes), in front of its first for
    align < 0; 0 switches back to hipMalloc.  Returns (mapping granule in bytes, guarded blocks alive)."""
    g, live = _i64(0), _i64(0)
    check(load().mhx_debug_guard_alloc(int(align), ctypes.byref(g), ctypes.byref(live)))
    return int(g.value), int(live.value)


def poison_alloc(byte_value: int) -> None:
    """Debugging (mhx_debug_poison_alloc): fresh device allocations of the library are filled with this byte (0..255;
    -1 = off) -- the dirty memory of a board that has been in use, made deterministic."""
    check(load().mhx_debug_poison_alloc(int(byte_value)))


class DeviceBuffer:
    """A device allocation owned by a Context (freed on close() / garbage collection)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = _vp()
        check(ctx.lib.mhx_dev_alloc(ctx.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray, offset: int = 0) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        if offset + arr.nbytes > self.nbytes:
            raise ValueError("upload exceeds the device buffer")
        check(self.ctx.lib.mhx_memcpy_h2d(self.ctx.handle, self.ptr + offset, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if offset + out.nbytes > self.nbytes:
            raise ValueError("download exceeds the device buffer")
        check(self.ctx.lib.mhx_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def download_into(self, out: np.ndarray, offset: int = 0) -> np.ndarray:
        """Device -> an existing contiguous host array (a shared-memory view, a slice of a larger result)."""
        if not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("download_into needs a writable C-contiguous array")
        if offset + out.nbytes > self.nbytes:
            raise ValueError("download exceeds the device buffer")
        check(self.ctx.lib.mhx_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr is not None and self.ctx.handle is not None:
            self.ctx.lib.mhx_dev_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self, ctx: "Context"):
        self.ctx = ctx
        p = _vp()
        check(ctx.lib.mhx_event_create(ctx.handle, ctypes.byref(p)))
        self.handle = p.value

    def record(self) -> "Event":
        check(self.ctx.lib.mhx_event_record(self.handle))
        return self

    def synchronize(self) -> None:
        check(self.ctx.lib.mhx_event_synchronize(self.handle))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = ctypes.c_float(0)
        check(self.ctx.lib.mhx_event_elapsed_ms(self.handle, stop.handle, ctypes.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle is not None and self.ctx.handle is not None:
                self.ctx.lib.mhx_event_destroy(self.handle)
        except Exception:
            pass
        self.handle = None


class Context:
    """One device + one HIP stream (mhx_ctx).  Not thread-safe; use one per thread/GPU."""

    def __init__(self, device: int = 0):
        self.lib = load()
        p = _vp()
        check(self.lib.mhx_ctx_create(int(device), ctypes.byref(p)))
        self.handle = p.value
        self.device = int(device)
        self._perms = {}  # (K, digest) -> perm handle
        self._wgens = {}
        self._comms = weakref.WeakSet()  # RCCL communicators made on this context: destroyed before it (close())

    # -- info / knobs
    def info(self) -> dict:
        name = ctypes.create_string_buffer(128)
        cus = _int(0)
        hbm = _i64(0)
        check(self.lib.mhx_ctx_device_info(self.handle, name, 128, ctypes.byref(cus), ctypes.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value, "device": self.device}

    def set_option(self, key: str, value: int) -> None:
        check(self.lib.mhx_ctx_set_option(self.handle, key.encode(), int(value)))

    def counters(self, enable: bool = True) -> dict:
        """Kernel event counters since the previous call (then reset); see mhx_ctx_counters."""
        out = (ctypes.c_uint64 * 4)()
        check(self.lib.mhx_ctx_counters(self.handle, 1 if enable else 0, out))
        return {"sieve_sets_redone": int(out[0]), "exact_sets_redone": int(out[1]), "sieve_blocks": int(out[2]),
                "pairwise_sets": int(out[3])}

    def minhash_mode(self, reset: bool = False) -> int:
        """What the last MinHash call on this context learned about the corpus (0 clean, 1 many sets defeat the one-candidate
        proof, 2 heavily repeated tokens): the next call's first launch follows it.  ``reset=True`` forgets it."""
        mode = _int(0)
        check(self.lib.mhx_ctx_minhash_mode(self.handle, 1 if reset else 0, ctypes.byref(mode)))
        return mode.value

    def minhash_flags(self, n_sets: int) -> np.ndarray:
        """uint8[n_sets]: which launch produced each set of the LAST ``minhash_bulk_dev`` call (0 first launch's proof held,
        1 second launch, 2 pairwise launch; mhx_ctx_minhash_flags).  Results are exact either way: a parity audit's index."""
        out = np.empty(int(n_sets), dtype=np.uint8)
        check(self.lib.mhx_ctx_minhash_flags(self.handle, int(n_sets), out.ctypes.data_as(_vp)))
        return out

    def synchronize(self) -> None:
        check(self.lib.mhx_ctx_synchronize(self.handle))

    def release_scratch(self) -> None:
        """Give back the device staging buffers the host entry points grew (re-created on demand)."""
        check(self.lib.mhx_ctx_release_scratch(self.handle))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr: np.ndarray) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 1)).upload(arr)

    def event(self) -> Event:
        return Event(self)

    def copy_dev(self, d_dst: int, d_src: int, nbytes: int) -> None:
        """Device-to-device copy on the context's stream (enqueued, not synchronised)."""
        check(self.lib.mhx_memcpy_d2d(self.handle, _vp(d_dst), _vp(d_src), int(nbytes)))

    # -- MinHash permutations (replaces the reference's _ensure_gpu_caches)
    def perm_handle(self, permutations) -> int:
        a = np.ascontiguousarray(permutations[0], dtype=np.uint64)
        b = np.ascontiguousarray(permutations[1], dtype=np.uint64)
        if a.ndim != 1 or a.shape != b.shape or a.size == 0:
            raise ValueError("permutations must be two equally long 1-D arrays")
        key = (a.size, hash(a.tobytes()), hash(b.tobytes()))
        h = self._perms.get(key)
        if h is None:
            if len(self._perms) >= 64:  # bounded cache of uploaded permutation sets
                _, old = self._perms.popitem()
                self.lib.mhx_perm_destroy(old)
            p = _vp()
            check(self.lib.mhx_perm_create(self.handle, a.ctypes.data, b.ctypes.data, a.size, ctypes.byref(p)))
            h = p.value
            self._perms[key] = h
        return h

    def wgen_create(self, rs, ln_cs, betas) -> int:
        """Upload WeightedMinHashGenerator tables (float32 [sample_size, dim]); returns an mhx_wgen handle."""
        rs = np.ascontiguousarray(rs, dtype=np.float32)
        ln_cs = np.ascontiguousarray(ln_cs, dtype=np.float32)
        betas = np.ascontiguousarray(betas, dtype=np.float32)
        if rs.ndim != 2 or rs.shape != ln_cs.shape or rs.shape != betas.shape:
            raise ValueError("rs, ln_cs, betas must share one 2-D shape")
        p = _vp()
        check(self.lib.mhx_wgen_create(self.handle, rs.ctypes.data, ln_cs.ctypes.data, betas.ctypes.data, rs.shape[0], rs.shape[1], ctypes.byref(p)))
        self._wgens[p.value] = True
        return p.value

    def wgen_destroy(self, handle: int) -> None:
        if self.handle is not None and self._wgens.pop(handle, None):
            self.lib.mhx_wgen_destroy(handle)

    # -- host-buffer entry points ------------------------------------------------------------
    def minhash_bulk(self, permutations, hv: np.ndarray, offsets: Optional[np.ndarray], fixed_len: int, n_sets: int,
                     init: Optional[np.ndarray] = None, out: Optional[np.ndarray] = None, out_dtype=np.uint64) -> np.ndarray:
        """CSR / fixed-length corpus of pre-hashed tokens -> [n_sets, K] signatures (host in, host out).
        ``hv`` travels as it is when it is a uint32 or uint64 array (uint32 = the range of ``sha1_hash32``: half
        the bytes over PCIe), anything else is converted to uint64.  ``out`` may name a C-contiguous
        [n_sets, K] array of ``out_dtype`` (uint64 = the reference's layout, uint32 = compact) to fill --
        reusing one avoids the page faults of a fresh gigabyte."""
        perm = self.perm_handle(permutations)
        k = len(permutations[0])
        hv = np.asarray(hv)
        if hv.dtype != np.uint32:
            hv = np.ascontiguousarray(hv, dtype=np.uint64)
        hv = np.ascontiguousarray(hv)
        hv_code = MHX_U32 if hv.dtype == np.uint32 else MHX_U64
        out_dtype = np.dtype(out_dtype if out is None else out.dtype)
        if out_dtype not in (np.dtype(np.uint64), np.dtype(np.uint32)):
            raise ValueError("signatures are uint64 or uint32")
        out_code = MHX_U32 if out_dtype == np.uint32 else MHX_U64
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.int64)
            if offsets.shape != (n_sets + 1,):
                raise ValueError("offsets must have n_sets+1 entries")
            if n_sets and int(offsets[-1]) > hv.size:
                raise ValueError("offsets run past the token array")
        elif fixed_len * n_sets > hv.size:
            raise ValueError("token array shorter than n_sets*fixed_len")
        stride = 0
        if init is not None:
            init = np.ascontiguousarray(init, dtype=np.uint64)
            if init.shape == (k,):
                stride = 0
            elif init.shape == (n_sets, k):
                stride = k
            else:
                raise ValueError("init must have shape (K,) or (n_sets, K)")
        if out is None:
            out = np.empty((n_sets, k), dtype=out_dtype)
        elif out.shape != (n_sets, k) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous array of shape (n_sets, K)")
        check(self.lib.mhx_minhash_bulk_typed(perm, _ptr(hv), hv_code, _ptr(offsets), int(fixed_len), int(n_sets), _ptr(init),
                                              stride, _ptr(out), out_code))
        return out

    @staticmethod
    def pack_tokens(tokens) -> tuple:
        """Byte tokens (bytes / bytearray / memoryview) -> (packed uint8 array, int64 offsets[n+1]).
        A str token raises the TypeError hashlib would raise for it."""
        if _mhxpack is not None:
            data, offs = _mhxpack.pack_tokens(tokens)
            return np.frombuffer(data, dtype=np.uint8), np.frombuffer(offs, dtype=np.int64)
        tokens = tokens if isinstance(tokens, (list, tuple)) else list(tokens)
        lens = np.fromiter(map(len, tokens), dtype=np.int64, count=len(tokens))
        offsets = np.zeros(len(tokens) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        buf = np.frombuffer(b"".join(tokens), dtype=np.uint8)
        if buf.size != int(offsets[-1]):
            raise TypeError("tokens must be bytes-like objects of single bytes")
        return buf, offsets

    @staticmethod
    def pack_sets(sets) -> tuple:
        """Sets of byte tokens -> (packed uint8 array, int64 byte offsets[T+1], int64 set offsets[N+1])."""
        if _mhxpack is not None:
            data, offs, set_offs = _mhxpack.pack_sets(sets)
            return np.frombuffer(data, dtype=np.uint8), np.frombuffer(offs, dtype=np.int64), np.frombuffer(set_offs, dtype=np.int64)
        sets = [s if isinstance(s, (list, tuple)) else list(s) for s in sets]
        set_offsets = np.zeros(len(sets) + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, sets), dtype=np.int64, count=len(sets)), out=set_offsets[1:])
        buf, byte_offsets = Context.pack_tokens([t for s in sets for t in s])
        return buf, byte_offsets, set_offsets

    @staticmethod
    def pack_int_sets(sets):
        """Lists / tuples of exact Python ints -> (uint64 values, int64 set offsets) by the C helper, or None when it
        does not apply (helper missing, other element types): the caller takes the numpy route then."""
        if _mhxpack is None or not hasattr(_mhxpack, "pack_int_sets"):
            return None
        packed = _mhxpack.pack_int_sets(sets)
        if packed is None:
            return None
        return np.frombuffer(packed[0], dtype=np.uint64), np.frombuffer(packed[1], dtype=np.int64)

    def sha1_tokens(self, buf: np.ndarray, byte_offsets: np.ndarray, bits: int = 32) -> np.ndarray:
        """sha1_hash32 / sha1_hash64 of every token of a packed byte corpus (host in, host out)."""
        n = byte_offsets.size - 1
        out = np.empty(n, dtype=np.uint32 if bits == 32 else np.uint64)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.int64)
        check(self.lib.mhx_sha1_tokens(self.handle, _ptr(buf), _ptr(byte_offsets), n, MHX_U32 if bits == 32 else MHX_U64, _ptr(out)))
        return out

    def minhash_bulk_bytes(self, permutations, buf: np.ndarray, byte_offsets: np.ndarray, set_offsets: np.ndarray,
                           init: Optional[np.ndarray] = None, bits: int = 32) -> np.ndarray:
        """Raw byte tokens -> sha1_hash32 (``bits=32``) or sha1_hash64 (``bits=64``) -> [n_sets, K] uint64
        signatures, all on the device."""
        if bits not in (32, 64):
            raise ValueError("bits must be 32 or 64")
        perm = self.perm_handle(permutations)
        k = len(permutations[0])
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        byte_offsets = np.ascontiguousarray(byte_offsets, dtype=np.int64)
        set_offsets = np.ascontiguousarray(set_offsets, dtype=np.int64)
        n_sets, n_tokens = set_offsets.size - 1, byte_offsets.size - 1
        stride = 0
        if init is not None:
            init = np.ascontiguousarray(init, dtype=np.uint64)
            if init.shape == (n_sets, k):
                stride = k
            elif init.shape != (k,):
                raise ValueError("init must have shape (K,) or (n_sets, K)")
        out = np.empty((n_sets, k), dtype=np.uint64)
        check(self.lib.mhx_minhash_bulk_bytes_typed(perm, _ptr(buf), _ptr(byte_offsets), n_tokens, MHX_U32 if bits == 32 else MHX_U64,
                                                    _ptr(set_offsets), n_sets, _ptr(init), stride, _ptr(out)))
        return out

    def minhash_update_batch(self, permutations, hv: np.ndarray, hashvalues: np.ndarray) -> np.ndarray:
        perm = self.perm_handle(permutations)
        hv = np.ascontiguousarray(hv, dtype=np.uint64)
        state = np.array(hashvalues, dtype=np.uint64, copy=True)
        check(self.lib.mhx_minhash_update_batch(perm, _ptr(hv), hv.size, _ptr(state)))
        return state

    def minhash_merge(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.uint64)
        y = np.ascontiguousarray(y, dtype=np.uint64)
        if x.shape != y.shape:
            raise ValueError("signature matrices must have the same shape")
        out = np.empty_like(x)
        check(self.lib.mhx_minhash_merge(self.handle, _ptr(x), _ptr(y), x.size, _ptr(out)))
        return out

    def weighted_minhash_many(self, h: int, sample_size: int, indptr, indices, values, values_are_logs: bool):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        n = indptr.size - 1
        out = np.zeros((n, int(sample_size), 2), dtype=np.int64)
        nonempty = np.zeros(n, dtype=np.uint8)
        check(self.lib.mhx_weighted_minhash_many(h, _ptr(indptr), _ptr(indices), _ptr(values), int(bool(values_are_logs)), n, _ptr(out), _ptr(nonempty)))
        return out, nonempty.astype(bool)

    def weighted_minhash_many_dense(self, h: int, sample_size: int, x: np.ndarray, values_are_logs: bool, out=None, nonempty=None):
        """Dense [N, dim] float32 rows (values, or logs with -inf for absent entries) -> (out, nonempty).
        ``out`` / ``nonempty``: optional C-contiguous int64 [N, S, 2] / uint8 [N] arrays (or leading-axis slices
        of such) to fill in place."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        if out is None:
            out = np.zeros((n, int(sample_size), 2), dtype=np.int64)
        if nonempty is None:
            nonempty = np.zeros(n, dtype=np.uint8)
        if out.shape != (n, int(sample_size), 2) or out.dtype != np.int64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous int64 array of shape (N, sample_size, 2)")
        if nonempty.shape != (n,) or nonempty.dtype != np.uint8 or not nonempty.flags.c_contiguous:
            raise ValueError("nonempty must be a C-contiguous uint8 array of shape (N,)")
        check(self.lib.mhx_weighted_minhash_many_dense(h, _ptr(x), int(bool(values_are_logs)), n, _ptr(out), _ptr(nonempty)))
        return out, nonempty.view(bool)

    def pinned_empty(self, shape, dtype) -> np.ndarray:
        """An uninitialised array in page-locked host memory (mhx_host_alloc): what is filled and uploaded again and
        again goes up by DMA straight from it.  The memory is released when the array (and every view of it) is gone."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = _vp()
        check(self.lib.mhx_host_alloc(self.handle, n, ctypes.byref(p)))
        buf = (ctypes.c_char * max(n, 1)).from_address(p.value)
        ptr, lib, ctx_ref = p.value, self.lib, weakref.ref(self)

        def release():  # with the context if it is still open (it waits for transfers in flight), without it otherwise
            ctx = ctx_ref()
            lib.mhx_host_free(ctx.handle if ctx is not None and ctx.handle else None, ptr)

        # (not at interpreter exit, when the library may be gone already: the process's pages go back anyway)
        weakref.finalize(buf, release).atexit = False
        return np.frombuffer(buf, dtype=dtype, count=n // dtype.itemsize).reshape(shape)

    def weighted_dense_feed(self, h: int, sample_size: int, dim: int, values_are_logs: bool, piece_rows: int) -> "WeightedFeed":
        """Dense rows in pieces (mhx_weighted_dense_begin/feed/end); use as a context manager."""
        return WeightedFeed(self, h, sample_size, dim, values_are_logs, piece_rows)

    def device_log_matches_numpy(self) -> bool:
        """True when the device's float32 log (np_logf, weighted_kernels.hip: numpy's AVX2 / AVX512F loop restated)
        reproduces THIS host's ``np.log`` bit for bit on ~14 000 sentinel values -- random patterns over the whole
        range, denormals, both sides of the 1/sqrt(2) split in every binade, powers of two, zeros, infinities, NaNs,
        negatives.  numpy's float32 log is not correctly rounded and depends on the CPU dispatch (a host without AVX2
        runs libm's); the weighted sketch's parity mode takes the log on the device only where this holds
        (ref: datasketch/weighted_minhash.py:212).  Checked once per context."""
        ok = getattr(self, "_log_matches", None)
        if ok is None:
            rng = np.random.RandomState(20240)
            near = np.arange(-2, 3, dtype=np.int64)
            split = int(np.float32(0.70710678).view(np.uint32)) & 0x007FFFFF
            pats = [rng.randint(1 << 23, 0x7F800000, size=8192), rng.randint(1, 1 << 23, size=1024),
                    np.array([0, 1, 2, (1 << 23) - 1, 1 << 23, 0x7F7FFFFF, 0x7F800000, 0x7FC00000, 0x7F800001, 0x80000000, 0x80000001,
                              0xBF800000, 0xFF800000, 0xFFC00000, 0x3F800000, 0x3F800001, 0x3F7FFFFF])]
            for e in range(1, 255):
                pats += [(e << 23) + near, (e << 23) + split + near]
            bits = np.clip(np.concatenate(pats), 0, 0xFFFFFFFF).astype(np.uint32)
            x = bits.view(np.float32)
            with np.errstate(all="ignore"):
                want = np.log(x)
            ok = bool(np.array_equal(self.weighted_logf(x).view(np.uint32), want.view(np.uint32)))
            self._log_matches = ok
        return ok

    def weighted_logf(self, x: np.ndarray) -> np.ndarray:
        """The float32 log the device-log mode of the weighted path takes (mhx_weighted_logf)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        check(self.lib.mhx_weighted_logf(self.handle, _ptr(x), x.size, _ptr(out)))
        return out

    def bbit_pack(self, sig: np.ndarray, b: int) -> np.ndarray:
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n, k = sig.shape
        nb = _i32(0)
        check(self.lib.mhx_bbit_num_blocks(k, int(b), ctypes.byref(nb)))
        out = np.zeros((n, nb.value), dtype=np.uint64)
        check(self.lib.mhx_bbit_pack(self.handle, _ptr(sig), n, k, int(b), _ptr(out)))
        return out

    def band_keys(self, sig: np.ndarray, bands: int, r: int) -> np.ndarray:
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n, k = sig.shape
        out = np.empty((n, bands * r), dtype=np.uint64)
        check(self.lib.mhx_band_keys(self.handle, _ptr(sig), n, k, int(bands), int(r), _ptr(out)))
        return out

    def band_digests(self, sig: np.ndarray, bands: int, r: int) -> np.ndarray:
        """[n, bands] uint64: FNV-1a-64 of every band key (mhx_band_digests)."""
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n, k = sig.shape
        out = np.empty((n, bands), dtype=np.uint64)
        check(self.lib.mhx_band_digests(self.handle, _ptr(sig), n, k, int(bands), int(r), _ptr(out)))
        return out

    def lsh_sort_bands(self, sig: np.ndarray, bands: int, r: int):
        """(sorted_digests [bands, n] uint64, sorted_rows [bands, n] uint32): per band, the band digests in
        ascending order and the rows in that order (mhx_lsh_sort_bands)."""
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n, k = sig.shape
        dig = np.empty((bands, n), dtype=np.uint64)
        rows = np.empty((bands, n), dtype=np.uint32)
        check(self.lib.mhx_lsh_sort_bands(self.handle, _ptr(sig), n, k, int(bands), int(r), _ptr(dig), _ptr(rows)))
        return dig, rows

    def lsh_candidate_pairs(self, sig: np.ndarray, bands: int, r: int, capacity: Optional[int] = None):
        """(pairs int64 [M, 2] ascending and unique, raw pair count before deduplication across bands):
        rows i < j sharing the key of at least one band (mhx_lsh_candidate_pairs).  ``capacity`` is the
        first guess of M; a larger answer costs one more call."""
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n, k = sig.shape
        cap = int(capacity) if capacity is not None else max(4 * n, 1 << 16)
        while True:
            pairs = np.empty((cap, 2), dtype=np.int64)
            found, raw = _i64(0), _i64(0)
            check(self.lib.mhx_lsh_candidate_pairs(self.handle, _ptr(sig), n, k, int(bands), int(r), _ptr(pairs), cap,
                                                   ctypes.byref(found), ctypes.byref(raw)))
            if found.value <= cap:
                return pairs[: found.value], int(raw.value)
            cap = int(found.value)

    def jaccard_pairs(self, sig: np.ndarray, pairs: np.ndarray) -> np.ndarray:
        """int32 counts of equal positions for rows (pairs[:,0], pairs[:,1]) of one signature matrix."""
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        n, k = sig.shape
        out = np.empty(pairs.shape[0], dtype=np.int32)
        check(self.lib.mhx_jaccard_pairs(self.handle, _ptr(sig), n, k, _ptr(pairs), pairs.shape[0], _ptr(out)))
        return out

    def bbit_jaccard_pairs(self, blocks: np.ndarray, num_perm: int, b: int, pairs: np.ndarray) -> np.ndarray:
        """int32 counts of agreeing b-bit positions for rows (pairs[:,0], pairs[:,1]) of a packed matrix
        (``bbit_pack`` output, [n, num_blocks] uint64)."""
        blocks = np.ascontiguousarray(blocks, dtype=np.uint64)
        pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        nb = _i32(0)
        check(self.lib.mhx_bbit_num_blocks(int(num_perm), int(b), ctypes.byref(nb)))
        if blocks.ndim != 2 or blocks.shape[1] != nb.value:
            raise ValueError("blocks must be [n, %d] for num_perm=%d, b=%d" % (nb.value, num_perm, b))
        out = np.empty(pairs.shape[0], dtype=np.int32)
        check(self.lib.mhx_bbit_jaccard_pairs(self.handle, _ptr(blocks), blocks.shape[0], int(num_perm), int(b), _ptr(pairs),
                                              pairs.shape[0], _ptr(out)))
        return out

    def bbit_pack_band_digests_dev(self, d_sig: int, sig_dtype: int, n: int, k: int, b: int, bands: int, r: int,
                                   d_blocks: int, d_digests: int, layout: int = 0) -> bool:
        """b-bit blocks and band digests of a device-resident matrix (mhx_bbit_pack_band_digests_dev); True when the
        fused kernel ran (the matrix was read once)."""
        fused = _int(0)
        check(self.lib.mhx_bbit_pack_band_digests_dev(self.handle, _vp(d_sig), int(sig_dtype), int(n), int(k), int(b), int(bands),
                                                      int(r), int(layout), _vp(d_blocks), _vp(d_digests), ctypes.byref(fused)))
        return bool(fused.value)

    def bbit_pack_band_digests(self, sig: np.ndarray, b: int, bands: int, r: int, layout: int = 0):
        """(blocks [n, num_blocks] uint64, digests [n, bands] uint64 -- [bands, n] with layout = BAND_MAJOR --, fused) of a
        host matrix (uint64 or uint32)."""
        sig = np.ascontiguousarray(sig)
        if sig.dtype != np.uint32:
            sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n, k = sig.shape
        nb = _i32(0)
        check(self.lib.mhx_bbit_num_blocks(k, int(b), ctypes.byref(nb)))
        d_sig = self.to_device(sig)
        d_blk, d_dig = self.alloc(max(1, n * nb.value * 8)), self.alloc(max(1, n * bands * 8))
        fused = self.bbit_pack_band_digests_dev(d_sig.ptr, MHX_U32 if sig.dtype == np.uint32 else MHX_U64, n, k, b, bands, r, d_blk.ptr, d_dig.ptr, layout)
        self.synchronize()
        out = d_blk.download((n, nb.value), np.uint64), d_dig.download((bands, n) if layout == BAND_MAJOR else (n, bands), np.uint64), fused
        for d in (d_sig, d_blk, d_dig):
            d.free()
        return out

    def lean_serialize(self, sig: np.ndarray, seed: int) -> np.ndarray:
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        n, k = sig.shape
        out = np.zeros((n, 12 + 4 * k), dtype=np.uint8)
        check(self.lib.mhx_lean_serialize(self.handle, _ptr(sig), n, k, int(seed), _ptr(out)))
        return out

    def lean_deserialize(self, records: np.ndarray, num_perm: int, big_endian: bool = False):
        """(seeds int64[n], hashvalues uint64[n, K]) of n LeanMinHash records laid back to back (mhx_lean_deserialize);
        ValueError when a record's length field is not ``num_perm``."""
        records = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1)
        rec = 12 + 4 * int(num_perm)
        if records.size % rec:
            raise ValueError("the buffer is not a whole number of %d-byte records" % rec)
        n = records.size // rec
        sig, seeds = np.empty((n, int(num_perm)), dtype=np.uint64), np.empty(n, dtype=np.int64)
        check(self.lib.mhx_lean_deserialize(self.handle, _ptr(records), n, int(num_perm), 1 if big_endian else 0, _ptr(sig), _ptr(seeds)))
        return seeds, sig

    def bbit_unpack(self, blocks: np.ndarray, num_perm: int, b: int) -> np.ndarray:
        """uint32[n, num_perm]: the b-bit values of packed rows (mhx_bbit_unpack, the inverse of :meth:`bbit_pack`)."""
        blocks = np.ascontiguousarray(blocks, dtype=np.uint64)
        out = np.empty((blocks.shape[0], int(num_perm)), dtype=np.uint32)
        check(self.lib.mhx_bbit_unpack(self.handle, _ptr(blocks), blocks.shape[0], int(num_perm), int(b), _ptr(out)))
        return out

    # -- device-resident entry points (pointers are DeviceBuffer.ptr + byte offsets) -----------
    def minhash_bulk_dev(self, permutations, d_hv: int, hv_dtype: int, d_offsets: Optional[int], fixed_len: int, n_sets: int,
                         total_tokens: int, d_init: Optional[int], init_stride: int, d_out: int, out_dtype: int) -> None:
        perm = self.perm_handle(permutations)
        check(self.lib.mhx_minhash_bulk_dev(perm, d_hv, hv_dtype, d_offsets, int(fixed_len), int(n_sets), int(total_tokens), d_init, int(init_stride), d_out, out_dtype))

    def close(self) -> None:
        if self.handle is None:
            return
        for comm in list(self._comms):  # (mhx_comm_destroy needs the context it was created on)
            try:
                comm.close()
            except Exception:  # noqa: BLE001
                pass
        for h in self._perms.values():
            self.lib.mhx_perm_destroy(h)
        for h in list(self._wgens):
            self.lib.mhx_wgen_destroy(h)
        self._perms.clear()
        self._wgens.clear()
        self.lib.mhx_ctx_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


class Communicator:
    """RCCL communicator of libmhx (mhx_comm_*): one rank per Context / GPU, all-gather of row shards
    over xGMI on the context's stream.  The 128-byte unique id is created on rank 0
    (``Communicator.unique_id()``) and handed to the other ranks by the caller -- any channel works
    (``datasketch_amd.dist`` broadcasts it over its own TCP rendezvous, ``datasketch_amd.rendezvous``)."""

    ID_BYTES = 128

    def __init__(self, ctx: "Context", unique_id: bytes, rank: int, world_size: int):
        if len(unique_id) != self.ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        self.ctx, self.rank, self.world_size = ctx, int(rank), int(world_size)
        buf = (ctypes.c_uint8 * self.ID_BYTES).from_buffer_copy(unique_id)
        h = _vp()
        check(ctx.lib.mhx_comm_create(ctx.handle, buf, self.rank, self.world_size, ctypes.byref(h)))
        self.handle = h
        ctx._comms.add(self)

    @staticmethod
    def unique_id() -> bytes:
        buf = (ctypes.c_uint8 * Communicator.ID_BYTES)()
        check(load().mhx_comm_unique_id(buf))
        return bytes(buf)

    def info(self) -> dict:
        """What RCCL itself reports: rank, ranks seen (ncclCommCount), HIP device, library version."""
        r, w, d, v = _int(-1), _int(-1), _int(-1), _int(0)
        check(self.ctx.lib.mhx_comm_info(self.handle, ctypes.byref(r), ctypes.byref(w), ctypes.byref(d), ctypes.byref(v)))
        return {"rank": r.value, "ranks_seen": w.value, "device": d.value, "rccl_version": v.value}

    def allgather_dev(self, d_send: int, d_recv: int, bytes_per_rank: int) -> None:
        """Enqueue the all-gather on the context's stream (device pointers; d_recv holds world*bytes)."""
        check(self.ctx.lib.mhx_comm_allgather_dev(self.handle, _vp(d_send), _vp(d_recv), int(bytes_per_rank)))

    def allgatherv_dev(self, d_send: int, d_recv: int, offsets, sizes) -> None:
        """Unequal shards in place (one grouped launch of broadcasts): rank q's sizes[q] bytes land at d_recv + offsets[q]."""
        n = self.world_size
        if len(offsets) != n or len(sizes) != n:
            raise ValueError("offsets and sizes have one entry per rank")
        off = (ctypes.c_uint64 * n)(*[int(v) for v in offsets])
        siz = (ctypes.c_uint64 * n)(*[int(v) for v in sizes])
        check(self.ctx.lib.mhx_comm_allgatherv_dev(self.handle, _vp(d_send), _vp(d_recv), off, siz))

    def exchange_dev(self, d_send: int, d_recv: int, sends, recvs) -> None:
        """One grouped launch of ncclSend / ncclRecv (mhx_comm_exchange_dev): ``sends`` = [(peer, offset, bytes)] out of
        ``d_send``, ``recvs`` = [(peer, offset, bytes)] into ``d_recv``; between two ranks messages match in list order."""
        def lists(msgs):
            n = len(msgs)
            return (n, (_i32 * max(n, 1))(*[int(m[0]) for m in msgs]), (ctypes.c_uint64 * max(n, 1))(*[int(m[1]) for m in msgs]),
                    (ctypes.c_uint64 * max(n, 1))(*[int(m[2]) for m in msgs]))

        ns, sp, so, sb = lists(sends)
        nr, rp, ro, rb = lists(recvs)
        check(self.ctx.lib.mhx_comm_exchange_dev(self.handle, _vp(d_send), _vp(d_recv), ns, sp, so, sb, nr, rp, ro, rb))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.ctx.lib.mhx_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_device() -> int:
    """Device for the implicit context: MHX_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
    for var in ("MHX_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v.strip().lstrip("-").isdigit():
            n = device_count()
            return int(v) % n if n else 0
    return 0


def context(device: Optional[int] = None) -> Context:
    """Process-wide context per device (created lazily; raises RuntimeError without a device)."""
    dev = default_device() if device is None else int(device)
    key = (os.getpid(), dev)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = Context(dev)
        _contexts[key] = ctx
    return ctx
