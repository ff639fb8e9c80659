"""The C-ABI library loads on a CPU-only host and exports every symbol include/mhx.h declares."""
import ctypes
import os
import re

import pytest

from datasketch_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mhx.h")).read()
    return sorted(set(re.findall(r"MHX_API\s+[\w\s\*]+?\b(mhx_\w+)\s*\(", text)))


def test_header_declares_the_bound_symbols():
    assert declared_symbols() == _native.EXPORTED_SYMBOLS


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_string():
    lib = _native.load()
    assert lib.mhx_version().startswith(b"mhx ")
    assert isinstance(_native.last_error(), str)


def test_bbit_num_blocks_is_pure_host_logic():
    lib = _native.load()
    nb = ctypes.c_int32(0)
    for k, b, want in ((256, 1, 4), (128, 1, 2), (100, 1, 2), (8, 3, 1), (48, 7, 6), (16, 32, 8), (17, 16, 5)):
        _native.check(lib.mhx_bbit_num_blocks(k, b, ctypes.byref(nb)))
        assert nb.value == want, (k, b)
    with pytest.raises(ValueError):
        _native.check(lib.mhx_bbit_num_blocks(8, 33, ctypes.byref(nb)))


@pytest.mark.skipif(_native.gpu_node_present(), reason="host has a GPU")
def test_no_device_is_reported_not_crashed():
    assert _native.device_count() == 0
    assert _native.gpu_available() is False
    with pytest.raises(RuntimeError):
        _native.Context(0)
