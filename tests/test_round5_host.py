"""Round 5's host-side decisions, on the CPU.

* `gpu_mode='detect'` keeps the reference's semantics (datasketch/minhash.py:272-279: use the device if there is one,
  else the numpy path, never an error) -- with one RuntimeWarning when the host has an AMD GPU but libmhx does not load;
  `gpu_mode='always'` stays strict (ref :272-275).
"""
import os
import warnings

import numpy as np
import pytest

from datasketch_amd import MinHash, WeightedMinHashGenerator, _native, lsh_bulk
from datasketch_amd.b_bit_minhash import pack_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def broken_library_on_a_gpu_host(monkeypatch):
    """A host with /dev/kfd whose libmhx.so cannot be loaded."""

    def boom():
        raise _native.MhxError("libmhx.so could not be loaded from /nowhere/libmhx.so: cannot open shared object file")

    monkeypatch.setattr(_native, "device_count", boom)
    monkeypatch.setattr(_native, "gpu_node_present", lambda: True)
    monkeypatch.setattr(_native, "_detect_warned", False)


def test_detect_falls_back_to_numpy_with_one_warning(broken_library_on_a_gpu_host):
    tokens = [f"t{i}".encode() for i in range(50)]
    ref = MinHash(num_perm=32, seed=5, gpu_mode="disable")
    ref.update_batch(tokens)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        m = MinHash(num_perm=32, seed=5, gpu_mode="detect")
        m.update_batch(tokens)                                              # ref minhash.py:276-277: no error
        sigs = MinHash.bulk_signatures(np.arange(60, dtype=np.uint64).reshape(6, 10), num_perm=16, seed=2, hashfunc=lambda x: x, gpu_mode="detect")
        blocks = pack_matrix(sigs, 1, gpu_mode="detect")
        dig = lsh_bulk.band_digests(sigs, 4, 4, gpu_mode="detect")
        g = WeightedMinHashGenerator(8, 4, seed=1, gpu_mode="detect")
        out = g.minhash_many(np.array([[1, 0, 3, 0, .5, 2, 0, 7], [0] * 8], dtype=np.float32))
    assert np.array_equal(m.hashvalues, ref.hashvalues)
    assert sigs.shape == (6, 16) and blocks.shape == (6, 1) and dig.shape == (6, 4)
    assert out[1] is None and out[0].hashvalues.tolist() == [[0, 0], [7, 3], [5, 0], [2, 1]]   # SURVEY.md 8c golden
    mine = [w for w in seen if issubclass(w.category, RuntimeWarning) and "datasketch_amd" in str(w.message)]
    assert len(mine) == 1 and "cannot open shared object file" in str(mine[0].message)  # announced once, names the load error


def test_always_stays_strict_on_the_same_host(broken_library_on_a_gpu_host):
    m = MinHash(num_perm=8, seed=1, gpu_mode="always")
    with pytest.raises(RuntimeError, match="GPU mode 'always' requested"):     # ref minhash.py:272-275
        m.update_batch([b"a"])
    with pytest.raises((RuntimeError, _native.MhxError)):
        lsh_bulk.band_digests(np.zeros((2, 8), dtype=np.uint64), 2, 4, gpu_mode="always")
    with pytest.raises(_native.MhxError):
        _native.gpu_available()                                                 # the strict probe still raises here


def test_detect_is_silent_on_a_host_without_a_gpu(monkeypatch):
    def boom():
        raise _native.MhxError("no library")

    monkeypatch.setattr(_native, "device_count", boom)
    monkeypatch.setattr(_native, "gpu_node_present", lambda: False)
    monkeypatch.setattr(_native, "_detect_warned", False)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        assert _native.gpu_detected() is False and _native.gpu_available() is False
    assert not seen


def test_device_log_side_can_be_overridden_and_is_recorded(monkeypatch):
    """the start-up check is a probabilistic stand-in for the exhaustive proof run on one numpy / GPU pair; an
    environment override and a record of the side chosen."""
    class Ctx:
        calls = 0

        def device_log_matches_numpy(self):
            Ctx.calls += 1
            return True

    g = WeightedMinHashGenerator(8, 4, seed=1, gpu_mode="disable")
    monkeypatch.delenv("MHX_WEIGHTED_DEVICE_LOG", raising=False)
    assert g._log_on_device(Ctx()) is True and g.log_taken_on == "device" and Ctx.calls == 1
    monkeypatch.setenv("MHX_WEIGHTED_DEVICE_LOG", "0")
    assert g._log_on_device(Ctx()) is False and g.log_taken_on == "host" and Ctx.calls == 1   # the check is not even asked
    monkeypatch.setenv("MHX_WEIGHTED_DEVICE_LOG", "1")
    assert g._log_on_device(Ctx()) is True
    g2 = WeightedMinHashGenerator(8, 4, seed=1, gpu_mode="disable", device_log=False)      # the argument wins over the environment
    assert g2._log_on_device(Ctx()) is False and g2.log_taken_on == "host"


def test_a_rank_whose_first_ack_came_too_late_can_join_again():
    """rank 0 used to turn a second hello of a registered rank away for good.  Here 'rank 1' says hello, gets its ACK
    and hangs up (what a joiner does whose ACK wait timed out); the real rank 1 then joins, and the group of three works."""
    import socket
    import struct
    import threading
    import time

    from datasketch_amd import rendezvous

    with socket.socket() as s0:
        s0.bind(("127.0.0.1", 0))
        port = s0.getsockname()[1]
    res = {}

    def run(rank, delay=0.0):
        time.sleep(delay)
        try:
            with rendezvous.Group(rank, 3, "127.0.0.1", port, timeout=30, nonce="n5") as g:
                res[rank] = g.allgather(bytes([rank]))
        except Exception as e:  # noqa: BLE001
            res[rank] = e

    t0 = threading.Thread(target=run, args=(0,))
    t0.start()
    deadline = time.time() + 10
    while True:
        try:
            c = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            assert time.time() < deadline
            time.sleep(0.02)
    c.sendall(struct.pack("<4sIQ", b"MHXR", 1, 2) + b"n5")
    assert c.recv(64)  # the ACK
    c.close()           # ... which this joiner believes it never got
    ts = [threading.Thread(target=run, args=(1, 0.1)), threading.Thread(target=run, args=(2, 0.2))]
    for t in ts:
        t.start()
    for t in [t0] + ts:
        t.join(40)
    assert all(res.get(r) == [b"\x00", b"\x01", b"\x02"] for r in range(3)), res


def test_cpu_baseline_times_the_real_reference_when_it_is_named(monkeypatch):
    """bench.py's cpu_baseline: kind 'reference' when DATASKETCH_REFERENCE names an importable ekzhu/datasketch (rows
    compared with the restatement's), kind 'port' with the dated ratio otherwise.  The reference exists in the build
    container only; without it the first half is skipped."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from oracle import oracle as O

    tokens = np.random.RandomState(42).randint(0, 2**32, (2000, 64), dtype=np.uint64)
    a, b = O.np_init_permutations(16, 1)
    monkeypatch.delenv("DATASKETCH_REFERENCE", raising=False)
    port = bench.cpu_baseline(tokens, a, b, 2000, 16, 64, None, seed=1)
    assert port["kind"] == "port" and "reference_over_port_time" not in port  # no scaled claim without the reference
    assert os.path.exists(os.path.join(root, port["reference_over_port_file"]))
    if not os.path.isdir("/root/reference/datasketch"):
        pytest.skip("the reference is not on this box")
    monkeypatch.setenv("DATASKETCH_REFERENCE", "/root/reference")
    ref = bench.cpu_baseline(tokens, a, b, 2000, 16, 64, None, seed=1)
    assert ref["kind"] == "reference" and ref["reference_over_port_measured"] == "in this run" and ref["cores"] >= 1
    assert 0.5 < ref["reference_over_port_time"] < 5 and "DATASKETCH_REFERENCE" in ref["sample"]


def test_the_rccl_stand_in_exports_every_entry_point_the_binding_resolves(tmp_path):
    """tests/fake_rccl.c (what the GPU tests load through MHX_RCCL_LIBRARY to run the RCCL binding with world > 1 on one device) must
    keep up with csrc/comm.hip: every "nccl..." name the binding hands to dlsym is exported by the stand-in.  Compiles without a GPU."""
    import re
    import shutil
    import subprocess

    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("no hipcc here")
    wanted = set(re.findall(r'"(nccl[A-Za-z]+)"', open(os.path.join(ROOT, "datasketch_amd", "csrc", "comm.hip")).read()))
    assert {"ncclAllGather", "ncclBroadcast", "ncclGroupStart", "ncclGroupEnd", "ncclCommInitRank"} <= wanted
    out = str(tmp_path / "libfake_rccl.so")
    p = subprocess.run([hipcc, "-x", "c", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.c"),
                        "-o", out, "-L/opt/rocm/lib", "-lamdhip64"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    exported = set(re.findall(r" T (nccl[A-Za-z]+)", subprocess.run(["nm", "-D", out], capture_output=True, text=True).stdout))
    assert wanted <= exported, wanted - exported
