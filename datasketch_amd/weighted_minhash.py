"""Weighted MinHash with the reference's API (datasketch/weighted_minhash.py) and a HIP back end.

``WeightedMinHashGenerator`` draws its tables on the host from numpy's legacy RandomState
(identical streams, identical float32 casts); ``minhash_many`` evaluates Ioffe's consistent
weighted sampling for a whole matrix on the device when ``gpu_mode`` is 'always' / 'detect'.
Parity: the natural log of the data is taken on the host with numpy (numpy's float32 log is not
correctly rounded, so only the same binary reproduces it); everything downstream is IEEE
float32 without FMA fusion on the device and yields bit-identical ``(k, t)`` pairs.
"""
from __future__ import annotations

import collections.abc
import logging
import os
import copy
from typing import List, Optional, Union

import numpy as np
import scipy.sparse as sparse

from datasketch_amd import _native


class WeightedMinHash:
    """Value type: ``seed`` and ``hashvalues`` of shape (sample_size, 2) int64 = (k, t) pairs."""

    def __init__(self, seed: int, hashvalues: np.ndarray) -> None:
        self.seed = seed
        self.hashvalues = hashvalues

    def jaccard(self, other: "WeightedMinHash") -> float:
        if other.seed != self.seed:
            raise ValueError("Cannot compute Jaccard given WeightedMinHash objects with different seeds")
        if len(self) != len(other):
            raise ValueError("Cannot compute Jaccard given WeightedMinHash objects with different numbers of hash values")
        same = np.all(np.asarray(self.hashvalues) == np.asarray(other.hashvalues), axis=1)
        return float(np.count_nonzero(same)) / float(len(self))

    def digest(self) -> np.ndarray:
        return copy.copy(self.hashvalues)

    def copy(self) -> "WeightedMinHash":
        return WeightedMinHash(self.seed, self.digest())

    def __len__(self) -> int:
        return len(self.hashvalues)

    def __eq__(self, other) -> bool:
        return type(self) is type(other) and self.seed == other.seed and np.array_equal(self.hashvalues, other.hashvalues)

    __hash__ = None


class WeightedMinHashGenerator:
    """Drop-in for ``datasketch.WeightedMinHashGenerator``.

    Args:
        dim: number of dimensions of the input vectors.
        sample_size: number of samples.
        seed: random seed.
        gpu_mode: 'disable' (numpy, as the reference), 'detect' or 'always' (HIP); not in the
            reference, which has no device path for the weighted sketch.
        device_log: where ``np.log`` of the data (ref: weighted_minhash.py:212) is taken.  None (default): on the
            device when its float32 log reproduces this host's numpy bit for bit (the device function restates numpy's
            AVX2 / AVX512F loop and equals it for all 2^32 float32 patterns; a start-up check on sentinel values says
            whether this host's numpy runs that loop), else on the host -- the ``(k, t)`` pairs are the reference's
            either way.  True: always on the device (no host pass; identical wherever the check holds).  False: always
            on the host.
    """

    def __init__(self, dim: int, sample_size: int = 128, seed: int = 1, gpu_mode: str = "disable", device_log: Optional[bool] = None) -> None:
        self.dim = dim
        self.sample_size = sample_size
        self.seed = seed
        rng = np.random.RandomState(seed=seed)
        # order of the draws and the float32 casts follow datasketch/weighted_minhash.py:118-121
        self.rs = rng.gamma(2, 1, (sample_size, dim)).astype(np.float32)
        self.ln_cs = np.log(rng.gamma(2, 1, (sample_size, dim))).astype(np.float32)
        self.betas = rng.uniform(0, 1, (sample_size, dim)).astype(np.float32)
        self._gpu_mode = gpu_mode
        self._device_log = device_log
        self._dev = None  # (Context, mhx_wgen handle); never pickled

    # ------------------------------------------------------------------ device plumbing
    def _use_gpu(self) -> bool:
        if self._gpu_mode == "always":
            try:
                ok = _native.gpu_available()
            except _native.MhxError as e:
                raise RuntimeError("GPU mode 'always' requested but no HIP device (or libmhx.so) is available.") from e
            if not ok:
                raise RuntimeError("GPU mode 'always' requested but no HIP device (or libmhx.so) is available.")
            return True
        if self._gpu_mode == "detect":
            return _native.gpu_detected()
        return False

    def _log_on_device(self, ctx) -> bool:
        """Where this generator's ``np.log`` is taken.  ``device_log`` given: that.  Else the environment override
        ``MHX_WEIGHTED_DEVICE_LOG`` (0 = always on the host, 1 = always on the device: for hosts whose numpy or GPU is
        not the pair the exhaustive 2^32-pattern proof was run on).  Else the start-up check on sentinel values.  The
        side chosen is kept in ``log_taken_on`` ('device' / 'host') and logged once at DEBUG level."""
        if self._device_log is not None:
            on_device, why = bool(self._device_log), "device_log argument"
        else:
            env = os.environ.get("MHX_WEIGHTED_DEVICE_LOG", "").strip()
            if env in ("0", "1"):
                on_device, why = env == "1", "MHX_WEIGHTED_DEVICE_LOG"
            else:
                on_device, why = bool(ctx.device_log_matches_numpy()), "start-up check of the device log against this host's np.log"
        side = "device" if on_device else "host"
        if getattr(self, "log_taken_on", None) != side:
            self.log_taken_on = side
            logging.getLogger("datasketch_amd").debug("WeightedMinHashGenerator: np.log taken on the %s (%s)", side, why)
        return on_device

    def _device_handle(self):
        ctx = _native.context()
        if self._dev is None or self._dev[0] is not ctx:
            self._dev = (ctx, ctx.wgen_create(self.rs, self.ln_cs, self.betas))
        return self._dev

    def __del__(self):
        dev = getattr(self, "_dev", None)
        if dev is not None:
            try:
                dev[0].wgen_destroy(dev[1])
            except Exception:
                pass

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_dev"] = None
        state.pop("_log_ring", None)
        return state

    # ------------------------------------------------------------------ single vector
    def minhash(self, v) -> WeightedMinHash:
        """One vector, on the host (reference: weighted_minhash.py:123-159; note its formula
        ``ln_a = ln_c - (t-beta)*r - r`` rounds differently from ``minhash_many``'s)."""
        if not isinstance(v, collections.abc.Sized):
            raise TypeError("Input vector must be sized")
        if not len(v) == self.dim:
            raise ValueError("Input dimension mismatch, expecting %d" % self.dim)
        v = np.array(v, dtype=np.float32)  # always a private float32 copy
        zeros = v == 0
        if zeros.all():
            raise ValueError("Input is all zeros")
        v[zeros] = np.nan
        vlog = np.log(v)
        t = np.floor((vlog / self.rs) + self.betas)  # (S, dim), row i == the reference's loop body i
        ln_y = (t - self.betas) * self.rs
        ln_a = self.ln_cs - ln_y - self.rs
        k = np.nanargmin(ln_a, axis=1)
        hashvalues = np.zeros((self.sample_size, 2), dtype=int)
        hashvalues[:, 0] = k
        hashvalues[:, 1] = t[np.arange(self.sample_size), k].astype(int)
        return WeightedMinHash(self.seed, hashvalues)

    # ------------------------------------------------------------------ matrix
    def minhash_many(self, X) -> List[Optional[WeightedMinHash]]:
        """One WeightedMinHash per row of ``X`` (dense ndarray or scipy sparse matrix); rows
        without non-zero entries give ``None`` (reference: weighted_minhash.py:161-247)."""
        if not isinstance(X, (sparse.spmatrix, np.ndarray)):
            raise TypeError("Input X must be a sparse matrix or numpy matrix")
        if X.ndim != 2:
            raise ValueError("Input must have two dimensions")
        if X.shape[1] != self.dim:
            raise ValueError("Input dimension mismatch, expecting %d" % self.dim)
        out, nonempty = self.minhash_many_arrays(X)
        return [WeightedMinHash(self.seed, out[i]) if nonempty[i] else None for i in range(out.shape[0])]

    def minhash_many_arrays(self, X):
        """Like :meth:`minhash_many` but returns ``(hashvalues[N, S, 2] int64, nonempty[N] bool)``."""
        if isinstance(X, np.ndarray) and X.ndim == 2 and self._use_gpu():
            # dense rows go to the device as they are: the CSR form (scipy on one host core in the
            # reference: seconds for 10^5 x 4096) is built there.  ln(0) = -inf marks the absent entries.
            ctx, handle = self._device_handle()
            x32 = np.ascontiguousarray(X, dtype=np.float32)
            if self._log_on_device(ctx):
                return ctx.weighted_minhash_many_dense(handle, self.sample_size, x32, False)
            return self._dense_parity_pipelined(ctx, handle, x32)
        X = sparse.csr_matrix(X, dtype=np.float32, copy=True)
        X.sort_indices()
        X.eliminate_zeros()  # explicit zeros are not part of a row (the reference's nonzero() skips them too)
        indptr = X.indptr.astype(np.int64)
        indices = X.indices.astype(np.int32)
        if self._use_gpu():
            ctx, handle = self._device_handle()
            if self._log_on_device(ctx):
                return ctx.weighted_minhash_many(handle, self.sample_size, indptr, indices, X.data, False)
            with np.errstate(invalid="ignore", divide="ignore"):
                log_data = np.log(X.data)
            return ctx.weighted_minhash_many(handle, self.sample_size, indptr, indices, log_data, True)
        return self._minhash_many_host(indptr, indices, X.data)

    # rows per piece of the pipelined dense call (>= 4096: fewer leave compute units without a row block; about
    # 64 MiB of values + results), and the threads that take np.log of the next piece while this one goes up
    _PIPE_PIECE_BYTES = 64 << 20
    _PIPE_LOG_THREADS = 3  # measured for 100k x 4096: 3 threads 0.044-0.051 s, 4 threads 0.052-0.060, 8 threads 0.056-0.066 (they compete with the upload for memory)

    def _dense_parity_pipelined(self, ctx, handle, x32: np.ndarray):
        """Parity mode on a dense matrix: ``np.log`` stays on the host (numpy's float32 log is what the reference
        computes, weighted_minhash.py:212, and only the same binary reproduces it bit for bit), but it does not
        serialise with the device: the matrix is cut into pieces of rows; a few threads take the logs of piece
        ``i+1`` (numpy releases the GIL inside the ufunc loop) into one of three reused page-locked buffers (a fresh
        buffer per piece would cost its page faults every time; they are kept on the generator between calls) while this thread
        feeds piece ``i`` to the device (``mhx_weighted_dense_feed``: upload of ``i``, evaluation of ``i``, download of
        ``i-1`` side by side; ctypes releases the GIL too)."""
        n, dim = x32.shape
        s = self.sample_size
        out = np.zeros((n, s, 2), dtype=np.int64)
        nonempty = np.zeros(n, dtype=np.uint8)
        rows = max(4096, (self._PIPE_PIECE_BYTES // (4 * dim + 16 * s)) & ~7)
        if n < 2 * rows:
            with np.errstate(invalid="ignore", divide="ignore"):
                logs = np.log(x32)
            return ctx.weighted_minhash_many_dense(handle, s, logs, True, out=out, nonempty=nonempty)
        from concurrent.futures import ThreadPoolExecutor

        ring = self.__dict__.pop("_log_ring", None)  # taken out while in use: a concurrent call makes its own
        if ring is None or ring[0].shape != (rows, dim):  # page-locked: a piece goes up by DMA straight from it
            try:
                ring = [ctx.pinned_empty((rows, dim), np.float32) for _ in range(3)]
            except _native.MhxError:  # no page-locked memory to be had: ordinary buffers (the upload is staged then)
                ring = [np.empty((rows, dim), dtype=np.float32) for _ in range(3)]
        threads = self._PIPE_LOG_THREADS
        starts = list(range(0, n, rows))

        def take_log(lo, hi, buf):
            with np.errstate(invalid="ignore", divide="ignore"):
                np.log(x32[lo:hi], out=buf)

        def submit(pool, i):  # the logs of piece i, one slice of its rows per thread
            lo, hi = starts[i], min(n, starts[i] + rows)
            buf = ring[i % 3]
            step = -(-(hi - lo) // threads)
            return [pool.submit(take_log, a, min(hi, a + step), buf[a - lo : min(hi, a + step) - lo]) for a in range(lo, hi, step)]

        with ThreadPoolExecutor(threads) as pool, ctx.weighted_dense_feed(handle, s, dim, True, rows) as feed:
            pending = {i: submit(pool, i) for i in range(min(2, len(starts)))}
            for i, lo in enumerate(starts):
                for f in pending.pop(i):
                    f.result()
                hi = min(n, lo + rows)
                feed.feed(ring[i % 3][: hi - lo], out[lo:hi], nonempty[lo:hi])
                if i + 2 < len(starts):  # buffer (i + 2) % 3 held piece i - 1, which went up during the last feed
                    pending[i + 2] = submit(pool, i + 2)
        self._log_ring = ring
        return out, nonempty.view(bool)

    def release_buffers(self) -> None:
        """Drop the host buffers the pipelined dense call keeps between calls (3 pieces of about 64 MiB)."""
        self._log_ring = None

    def _minhash_many_host(self, indptr, indices, data):
        """gpu_mode='disable': the reference's vectorised numpy evaluation, row by row."""
        n = indptr.size - 1
        s = self.sample_size
        out = np.zeros((n, s, 2), dtype=np.int64)
        nonempty = np.diff(indptr) > 0
        with np.errstate(invalid="ignore", divide="ignore"):
            log_data = np.log(data)
        rows = np.arange(s)
        for d in np.flatnonzero(nonempty):
            cols = indices[indptr[d] : indptr[d + 1]]
            r, be = self.rs[:, cols], self.betas[:, cols]
            t = np.floor(log_data[indptr[d] : indptr[d + 1]][None, :] / r + be)
            ln_a = self.ln_cs[:, cols] - (t - be + 1) * r
            j = np.argmin(ln_a, axis=1)
            out[d, :, 0] = cols[j]
            out[d, :, 1] = t[rows, j]
        return out, nonempty
