"""Every device entry point of include/mhx.h on buffers that abut an unmapped page (test infrastructure, run as a
script in a process of its own by tests/test_guard_pages.py -- a kernel that over-reads kills the process).

    python tests/guard_cases.py <align>            all cases; prints "GUARD OK <n> cases" and exits 0
    python tests/guard_cases.py <align> overread   the positive control: a launch told to read past its buffer must die

<align> > 0: the LAST byte of every buffer (size rounded up to <align> bytes) is the last byte of its mapping;
<align> < 0: the FIRST byte is the first byte of its mapping.  mhx_debug_guard_alloc (include/mhx.h) does the mapping
with the HIP virtual-memory API; inputs and outputs of every call are separate exact-size allocations, so that the end
of what the argument list describes is the end of what is mapped.  Results are checked too (oracle / host entry points).
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from datasketch_amd import _native  # noqa: E402
from datasketch_amd._native import MHX_U32, MHX_U64, check  # noqa: E402
from oracle import oracle as O  # noqa: E402

_i64 = ctypes.c_int64
CASES = 0


def _done(name):
    global CASES
    CASES += 1
    if os.environ.get("GUARD_VERBOSE"):
        print("ok", name, flush=True)


_KEEP = []  # GUARD_KEEP=1: nothing is freed before the process ends (no virtual address is ever mapped twice)


def _dev(ctx, arr):
    buf = ctx.to_device(np.ascontiguousarray(arr))
    if os.environ.get("GUARD_KEEP"):
        _KEEP.append(buf)
    return buf


def _alloc(ctx, nbytes, fill=0xA5):
    """An output buffer, pre-filled with a pattern: an element the kernel does not write shows."""
    buf = ctx.alloc(nbytes)
    check(ctx.lib.mhx_memset_dev(ctx.handle, ctypes.c_void_p(buf.ptr), fill, nbytes))
    if os.environ.get("GUARD_KEEP"):
        _KEEP.append(buf)
    return buf


def _expect(got, want, what):
    if np.array_equal(got, want):
        return
    bad = np.argwhere(np.asarray(got) != np.asarray(want))
    print("MISMATCH", what, "elements", len(bad), "of", np.asarray(want).size, "first", bad[:6].tolist(),
          "got", [hex(int(np.asarray(got)[tuple(i)])) for i in bad[:4]], "want", [hex(int(np.asarray(want)[tuple(i)])) for i in bad[:4]], flush=True)
    global FAILED
    FAILED += 1
    if not os.environ.get("GUARD_CONTINUE"):
        raise AssertionError(what)


FAILED = 0


def _p(buf):
    return None if buf is None else ctypes.c_void_p(buf.ptr)


def minhash_cases(ctx):
    rng = np.random.RandomState(5)
    lib = ctx.lib
    shapes = []  # (n_sets, lengths or fixed_len, K)
    for k in (1, 16, 48, 64, 96, 128, 150, 200, 256, 320):
        for fixed in (1, 7, 16, 17, 100, 256, 300):
            shapes.append((11 if fixed > 16 else 67, fixed, k))
    for tok_dtype, out_dtype in ((np.uint64, np.uint64), (np.uint32, np.uint32), (np.uint64, np.uint32), (np.uint32, np.uint64)):
        for n, fixed, k in shapes:
            if tok_dtype == np.uint32 and k not in (16, 128, 256):
                continue
            perms = O.np_init_permutations(k, 3)
            hv = rng.randint(0, 2**32, size=(n, fixed), dtype=np.uint64)
            want = O.c_minhash_bulk_dense(hv, perms[0], perms[1])
            d_hv, d_out = _dev(ctx, hv.astype(tok_dtype)), _alloc(ctx, n * k * np.dtype(out_dtype).itemsize)
            ctx.minhash_bulk_dev(perms, d_hv.ptr, MHX_U32 if tok_dtype == np.uint32 else MHX_U64, None, fixed, n, n * fixed, None, 0,
                                 d_out.ptr, MHX_U32 if out_dtype == np.uint32 else MHX_U64)
            got = d_out.download((n, k), out_dtype)
            _expect(got.astype(np.uint64), want, ("dense", n, fixed, k, np.dtype(tok_dtype).name, np.dtype(out_dtype).name))
            _done(f"minhash dense {n}x{fixed} K={k} {np.dtype(tok_dtype).name}->{np.dtype(out_dtype).name}")
    # CSR, ragged: tails of 1..7 tokens behind the last whole chunk, the LAST set ending at the end of the token array
    for k in (8, 32, 64, 128, 192, 256):
        for lo, hi, n in ((0, 8, 300), (1, 40, 257), (1, 100, 131), (32, 480, 70), (250, 270, 40), (3, 4, 1), (4000, 9000, 3), (60000, 60010, 1)):
            for tok_dtype in (np.uint64, np.uint32):
                lens = rng.randint(lo, hi, size=n)
                lens[-1] = max(1, int(lens[-1]) | 1)  # an odd tail at the very end
                offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                hv = rng.randint(0, 2**32, size=int(offsets[-1]), dtype=np.uint64)
                if tok_dtype == np.uint64:
                    hv[rng.randint(0, hv.size, size=max(1, hv.size // 50))] += np.uint64(2**40)  # wide tokens too
                perms = O.np_init_permutations(k, 9)
                init = rng.randint(0, 2**32, size=(n, k), dtype=np.uint64)
                want = O.c_minhash_bulk(hv, offsets, perms[0], perms[1], init=init)
                d_hv, d_off, d_init, d_out = _dev(ctx, hv.astype(tok_dtype)), _dev(ctx, offsets), _dev(ctx, init), _alloc(ctx, n * k * 8)
                ctx.minhash_bulk_dev(perms, d_hv.ptr, MHX_U32 if tok_dtype == np.uint32 else MHX_U64, d_off.ptr, 0, n, hv.size, d_init.ptr, k,
                                     d_out.ptr, MHX_U64)
                _expect(d_out.download((n, k), np.uint64), want, ("csr", k, lo, hi, n, np.dtype(tok_dtype).name))
                _done(f"minhash csr K={k} lens {lo}..{hi} n={n} {np.dtype(tok_dtype).name}")
    # repeated tokens: the dedup and pairwise launches
    for k in (64, 128, 256):
        n, t = 97, 77
        hv = rng.randint(0, 2**32, size=(n, t), dtype=np.uint64)
        hv[:, 5] = hv[:, 60]
        hv[::3, 7] = hv[::3, 8]
        hv[::3, 9] = hv[::3, 8]
        hv[::5] = hv[::5, :1]  # constant sets
        perms = O.np_init_permutations(k, 2)
        d_hv, d_out = _dev(ctx, hv), _alloc(ctx, n * k * 8)
        ctx.minhash_bulk_dev(perms, d_hv.ptr, MHX_U64, None, t, n, n * t, None, 0, d_out.ptr, MHX_U64)
        _expect(d_out.download((n, k), np.uint64), O.c_minhash_bulk_dense(hv, perms[0], perms[1]), ("repeats", k))
        _done(f"minhash repeats K={k}")
    # merge: odd counts
    for count in (1, 2, 3, 127, 128 * 33 + 1):
        x, y = rng.randint(0, 2**32, size=count, dtype=np.uint64), rng.randint(0, 2**32, size=count, dtype=np.uint64)
        d_x, d_y, d_o = _dev(ctx, x), _dev(ctx, y), ctx.alloc(8 * count)
        check(lib.mhx_minhash_merge_dev(ctx.handle, _p(d_x), _p(d_y), count, _p(d_o)))
        assert np.array_equal(d_o.download(count, np.uint64), np.minimum(x, y))
        _done(f"merge {count}")


def sha1_cases(ctx):
    import hashlib
    import struct

    rng = np.random.RandomState(6)
    for n, max_len in ((1, 1), (1, 0), (50, 70), (1000, 13), (333, 130)):
        toks = [bytes(rng.randint(0, 256, size=rng.randint(0, max_len + 1), dtype=np.uint8)) for _ in range(n)]
        buf = np.frombuffer(b"".join(toks), dtype=np.uint8)
        off = np.concatenate([[0], np.cumsum([len(t) for t in toks])]).astype(np.int64)
        d_buf = _dev(ctx, buf) if buf.size else ctx.alloc(1)
        d_off = _dev(ctx, off)
        for code, dt, fmt, nb in ((MHX_U32, np.uint32, "<I", 4), (MHX_U64, np.uint64, "<Q", 8)):
            d_out = ctx.alloc(n * nb)
            check(ctx.lib.mhx_sha1_tokens_dev(ctx.handle, _p(d_buf), _p(d_off), n, code, _p(d_out)))
            want = np.array([struct.unpack(fmt, hashlib.sha1(t).digest()[:nb])[0] for t in toks], dtype=dt)
            assert np.array_equal(d_out.download(n, dt), want), ("sha1", n, max_len, code)
            _done(f"sha1 n={n} len<={max_len} {np.dtype(dt).name}")


def weighted_cases(ctx):
    import scipy.sparse as sp

    rng = np.random.RandomState(7)
    for dim, s, n in ((1, 1, 3), (7, 3, 20), (63, 64, 50), (301, 65, 1100), (513, 128, 40), (1024, 128, 2100), (4096, 128, 33), (4097, 129, 9), (5000, 64, 21)):
        rs_, ln_cs, betas = O.np_weighted_params(dim, s, 4)
        h = ctx.wgen_create(rs_, ln_cs, betas)
        x = rng.uniform(0, 50, (n, dim)).astype(np.float32)
        if dim > 4:
            x[rng.random_sample(x.shape) < rng.choice([0.0, 0.5, 0.95], size=(n, 1))] = 0
        x[n // 2] = 0
        csr = sp.csr_matrix(x)
        csr.sort_indices()
        indptr, indices = csr.indptr.astype(np.int64), csr.indices.astype(np.int32)
        with np.errstate(divide="ignore"):
            want, wn = O.c_weighted_minhash_many(indptr, indices, csr.data, rs_, ln_cs, betas)
            logs = np.log(x)
        # dense, logs
        d_x, d_out, d_ne = _dev(ctx, logs), ctx.alloc(n * s * 16), ctx.alloc(n)
        check(ctx.lib.mhx_weighted_minhash_many_dense_dev(h, _p(d_x), 1, n, _p(d_out), _p(d_ne)))
        got, ne = d_out.download((n, s, 2), np.int64), d_ne.download(n, np.uint8).astype(bool)
        assert np.array_equal(ne, wn) and np.array_equal(got[wn], want[wn]), ("weighted dense", dim, s, n)
        # dense, values (device log: the launch and its reads are what matters here)
        d_v = _dev(ctx, x)
        check(ctx.lib.mhx_weighted_minhash_many_dense_dev(h, _p(d_v), 0, n, _p(d_out), _p(d_ne)))
        assert np.array_equal(d_ne.download(n, np.uint8).astype(bool), wn)
        # CSR, logs
        if csr.nnz:
            d_ip, d_ix, d_lv = _dev(ctx, indptr), _dev(ctx, indices), _dev(ctx, np.log(csr.data))
            check(ctx.lib.mhx_weighted_minhash_many_dev(h, _p(d_ip), _p(d_ix), _p(d_lv), 1, n, csr.nnz, _p(d_out), _p(d_ne)))
            got, ne = d_out.download((n, s, 2), np.int64), d_ne.download(n, np.uint8).astype(bool)
            assert np.array_equal(ne, wn) and np.array_equal(got[wn], want[wn]), ("weighted csr", dim, s, n)
            d_cv = _dev(ctx, csr.data)
            check(ctx.lib.mhx_weighted_minhash_many_dev(h, _p(d_ip), _p(d_ix), _p(d_cv), 0, n, csr.nnz, _p(d_out), _p(d_ne)))
            ctx.synchronize()
        # the every-element kernels of round 2 as well
        ctx.set_option("weighted.path", 2)
        check(ctx.lib.mhx_weighted_minhash_many_dense_dev(h, _p(d_x), 1, n, _p(d_out), _p(d_ne)))
        got = d_out.download((n, s, 2), np.int64)
        ctx.set_option("weighted.path", 0)
        assert np.array_equal(got[wn], want[wn]), ("weighted path 2", dim, s, n)
        ctx.wgen_destroy(h)
        _done(f"weighted dim={dim} S={s} n={n}")
    xs = rng.uniform(0.1, 9, 1001).astype(np.float32)
    assert np.allclose(ctx.weighted_logf(xs), np.log(xs), rtol=1e-6)  # host entry (stages through exact-size scratch in guard mode)
    _done("weighted logf")


def pack_and_lsh_cases(ctx):
    rng = np.random.RandomState(8)
    lib = ctx.lib
    for n, k in ((1, 1), (3, 7), (5, 64), (1000, 100), (777, 128), (129, 256), (2500, 256)):
        sig = rng.randint(0, 2**32, size=(n, k), dtype=np.uint64)
        sig[n // 2 :] = sig[: n - n // 2]  # duplicates: buckets with more than one row
        for dt, code in ((np.uint64, MHX_U64), (np.uint32, MHX_U32)):
            d_sig = _dev(ctx, sig.astype(dt))
            for b in (1, 2, 3, 4, 8, 13, 16, 32):
                want = O.c_bbit_pack(sig, b)
                d_out = ctx.alloc(want.nbytes)
                check(lib.mhx_bbit_pack_dev_typed(ctx.handle, _p(d_sig), code, n, k, b, _p(d_out)))
                assert np.array_equal(d_out.download(want.shape, np.uint64), want), ("bbit", n, k, b, dt)
                pairs = rng.randint(0, n, size=(37, 2)).astype(np.int64)
                d_pairs, d_cnt = _dev(ctx, pairs), ctx.alloc(4 * len(pairs))
                check(lib.mhx_bbit_jaccard_pairs_dev(ctx.handle, _p(d_out), _p(d_out), k, b, _p(d_pairs), len(pairs), _p(d_cnt)))
                mask = np.uint64((1 << b) - 1)
                assert np.array_equal(d_cnt.download(len(pairs), np.int32), ((sig[pairs[:, 0]] & mask) == (sig[pairs[:, 1]] & mask)).sum(axis=1))
            _done(f"bbit pack + jaccard n={n} K={k} {np.dtype(dt).name}")
            for bands, r in ((1, 1), (k, 1), (1, k), (max(1, k // 8), min(k, 8)), (max(1, k // 5), min(k, 5)), (max(1, k // 4), min(k, 3))):
                if bands * r > k or bands > 128:
                    continue
                keys = O.c_band_keys(sig, bands, r)
                if dt == np.uint64:
                    d_keys = ctx.alloc(keys.nbytes)
                    check(lib.mhx_band_keys_dev(ctx.handle, _p(d_sig), n, k, bands, r, _p(d_keys)))
                    assert np.array_equal(d_keys.download(keys.shape, np.uint64), keys), ("band keys", n, k, bands, r)
                dig = ctx.band_digests(sig, bands, r)
                d_dig = ctx.alloc(dig.nbytes)
                check(lib.mhx_band_digests_dev_typed(ctx.handle, _p(d_sig), code, n, k, bands, r, _p(d_dig)))
                assert np.array_equal(d_dig.download(dig.shape, np.uint64), dig), ("digests", n, k, bands, r, dt)
                order = np.argsort(dig.T, axis=1, kind="stable")
                d_sd, d_sr = ctx.alloc(8 * bands * n), ctx.alloc(4 * bands * n)
                for sort_opt in (0, 1):
                    ctx.set_option("lsh.sort", sort_opt)
                    check(lib.mhx_lsh_sort_bands_dev_typed(ctx.handle, _p(d_sig), code, n, k, bands, r, _p(d_sd), _p(d_sr)))
                    assert np.array_equal(d_sr.download((bands, n), np.uint32), order.astype(np.uint32)), ("sort", n, k, bands, r, dt, sort_opt)
                    assert np.array_equal(d_sd.download((bands, n), np.uint64), np.take_along_axis(dig.T, order, axis=1))
                    check(lib.mhx_lsh_sort_digests_dev(ctx.handle, _p(d_dig), n, bands, _p(d_sd), _p(d_sr)))  # the same from the digest matrix
                    assert np.array_equal(d_sr.download((bands, n), np.uint32), order.astype(np.uint32)), ("sort digests", n, k, bands, r, sort_opt)
                    assert np.array_equal(d_sd.download((bands, n), np.uint64), np.take_along_axis(dig.T, order, axis=1))
                ctx.set_option("lsh.sort", 0)
                want_pairs, _ = ctx.lsh_candidate_pairs(sig, bands, r)
                cap = max(1, len(want_pairs))
                d_pairs = ctx.alloc(16 * cap)
                found, raw = _i64(0), _i64(0)
                check(lib.mhx_lsh_candidate_pairs_dev(ctx.handle, _p(d_sd), _p(d_sr), n, bands, _p(d_pairs), cap, ctypes.byref(found), ctypes.byref(raw)))
                assert found.value == len(want_pairs) and np.array_equal(d_pairs.download((len(want_pairs), 2), np.int64), want_pairs)
                # bulk query of the first rows against the index (with band verification)
                m = min(n, 41)
                d_q = _dev(ctx, sig[:m].astype(dt))
                cap = 64 * m + 16
                while True:
                    d_qp = ctx.alloc(16 * cap)
                    check(lib.mhx_lsh_query_dev(ctx.handle, _p(d_sd), _p(d_sr), n, bands, r, _p(d_q), _p(d_sig), code, k, m, _p(d_qp), cap, ctypes.byref(found)))
                    if found.value <= cap:
                        break
                    cap = int(found.value)
                qp = d_qp.download((found.value, 2), np.int64)
                assert all(qp[qp[:, 0] == i][:, 1].tolist().count(i) == 1 for i in range(m)), "a probe finds itself"
                if len(want_pairs):
                    d_cnt, d_wp = ctx.alloc(4 * len(want_pairs)), _dev(ctx, want_pairs)
                    check(lib.mhx_jaccard_pairs_dev_typed(ctx.handle, _p(d_sig), _p(d_sig), code, k, _p(d_wp), len(want_pairs), _p(d_cnt)))
                    assert np.array_equal(d_cnt.download(len(want_pairs), np.int32), (sig[want_pairs[:, 0]] == sig[want_pairs[:, 1]]).sum(axis=1))
                _done(f"lsh n={n} K={k} bands={bands} r={r} {np.dtype(dt).name}")
        want = O.c_lean_serialize(sig, 11)
        d_sig, d_out = _dev(ctx, sig), ctx.alloc(want.nbytes)
        check(lib.mhx_lean_serialize_dev(ctx.handle, _p(d_sig), n, k, 11, _p(d_out)))
        assert np.array_equal(d_out.download(want.shape, np.uint8), want)
        _done(f"lean serialize n={n} K={k}")


def inverse_format_cases(ctx):
    """Round 6: the inverse wire formats (mhx_bbit_unpack_dev, mhx_lean_deserialize_dev) and the typed / big-endian serialiser."""
    lib = ctx.lib
    rng = np.random.RandomState(606)
    for n, k in ((1, 1), (3, 7), (65, 64), (33, 130), (257, 256)):
        sig = rng.randint(0, 2**32, (n, k), dtype=np.uint64)
        for b in (0, 1, 2, 3, 5, 8, 13, 16, 24, 32):
            blocks = O.c_bbit_pack(sig, b)
            d_blk, d_out = _dev(ctx, blocks), _alloc(ctx, n * k * 4)
            check(lib.mhx_bbit_unpack_dev(ctx.handle, _p(d_blk), n, k, b, _p(d_out)))
            _expect(d_out.download((n, k), np.uint32), (sig & np.uint64((1 << b) - 1)).astype(np.uint32), f"bbit unpack n={n} K={k} b={b}")
            _done(f"bbit unpack n={n} K={k} b={b}")
        for big in (0, 1):
            for code, dt in ((MHX_U64, np.uint64), (MHX_U32, np.uint32)):
                want = O.c_lean_serialize(sig, -77 - n)  # little-endian records
                if big:
                    w = want.reshape(n, -1).copy()
                    w[:, :8] = w[:, 7::-1]
                    w[:, 8:] = w[:, 8:].reshape(n, -1, 4)[:, :, ::-1].reshape(n, -1)
                    want = w.reshape(want.shape)
                d_sig, d_rec = _dev(ctx, sig.astype(dt)), _alloc(ctx, n * (12 + 4 * k))
                check(lib.mhx_lean_serialize_dev_typed(ctx.handle, _p(d_sig), code, n, k, -77 - n, big, _p(d_rec)))
                _expect(d_rec.download(want.shape, np.uint8), want, f"lean serialize typed n={n} K={k} big={big} code={code}")
                d_back, d_seeds, d_bad = _alloc(ctx, n * k * np.dtype(dt).itemsize), _alloc(ctx, n * 8), _dev(ctx, np.zeros(1, dtype=np.uint32))
                check(lib.mhx_lean_deserialize_dev(ctx.handle, _p(d_rec), n, k, big, code, _p(d_back), _p(d_seeds), _p(d_bad)))
                _expect(d_back.download((n, k), dt), sig.astype(dt), f"lean deserialize n={n} K={k} big={big} code={code}")
                _expect(d_seeds.download((n,), np.int64), np.full(n, -77 - n, dtype=np.int64), "lean deserialize seeds")
                _expect(d_bad.download((1,), np.uint32), np.zeros(1, dtype=np.uint32), "lean deserialize bad count")
                _done(f"lean round trip n={n} K={k} big={big} code={code}")


def overread(ctx):
    """The positive control: mhx_minhash_merge_dev told that its inputs are one granule longer than they are."""
    granule, _ = _native.guard_alloc(int(sys.argv[1]))
    count = 1 << 10
    d_x, d_y = _dev(ctx, np.zeros(count, dtype=np.uint64)), _dev(ctx, np.zeros(count, dtype=np.uint64))
    d_o = ctx.alloc(8 * count + 2 * granule)
    print("launching an over-read of", granule, "bytes", flush=True)
    check(ctx.lib.mhx_minhash_merge_dev(ctx.handle, _p(d_x), _p(d_y), count + granule // 8, _p(d_o)))
    ctx.synchronize()
    print("OVERREAD SURVIVED", flush=True)


def main():
    align = int(sys.argv[1])
    granule, _ = _native.guard_alloc(align)  # before the first allocation of the process
    assert granule > 0
    ctx = _native.context()
    if len(sys.argv) > 2 and sys.argv[2] == "overread":
        overread(ctx)
        return
    minhash_cases(ctx)
    sha1_cases(ctx)
    weighted_cases(ctx)
    pack_and_lsh_cases(ctx)
    inverse_format_cases(ctx)
    ctx.synchronize()
    _, live = _native.guard_alloc(align)
    if FAILED:
        print(f"GUARD FAILED: {FAILED} mismatching cases of {CASES}", flush=True)
        sys.exit(1)
    print(f"GUARD OK {CASES} cases (align {align}, granule {granule} bytes, {live} guarded blocks alive)", flush=True)


if __name__ == "__main__":
    main()
