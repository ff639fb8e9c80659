"""The reference's OWN unit tests for this path, run unmodified against datasketch_amd.

Build container only (needs /root/reference; skipped elsewhere): the modules the reference's tests
import -- ``datasketch``, ``datasketch.minhash``, ``datasketch.lean_minhash``,
``datasketch.weighted_minhash``, ``datasketch.b_bit_minhash``, ``datasketch.hashfunc`` -- are aliased
to this package, then ``test/test_minhash.py``, ``test_lean_minhash.py``, ``test_weighted_minhash.py``
and ``test_minhash_gpu.py`` run as they are.  This is the drop-in claim, checked by the reference's
test authors rather than by ours.  Nothing of the reference's product code is imported.
"""
import importlib
import os
import sys
import unittest

import pytest

REFERENCE = "/root/reference"
FILES = ["test_minhash", "test_lean_minhash", "test_weighted_minhash", "test_minhash_gpu"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "test")), reason="reference repository not mounted")


def _alias():
    import datasketch_amd
    from datasketch_amd import b_bit_minhash, hashfunc, lean_minhash, minhash, weighted_minhash

    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "datasketch" or k.startswith("datasketch.") or k == "test" or k.startswith("test.")}
    for k in saved:
        del sys.modules[k]
    sys.modules["datasketch"] = datasketch_amd
    sys.modules["datasketch.minhash"] = minhash
    sys.modules["datasketch.lean_minhash"] = lean_minhash
    sys.modules["datasketch.weighted_minhash"] = weighted_minhash
    sys.modules["datasketch.b_bit_minhash"] = b_bit_minhash
    sys.modules["datasketch.hashfunc"] = hashfunc
    return saved


def _restore(saved):
    for k in [k for k in sys.modules if k == "datasketch" or k.startswith("datasketch.") or k == "test" or k.startswith("test.")]:
        del sys.modules[k]
    sys.modules.update({k: v for k, v in saved.items() if v is not None})


@pytest.mark.parametrize("name", FILES)
def test_reference_test_file_passes_on_this_package(name):
    saved = _alias()
    sys.path.insert(0, REFERENCE)  # only for the `test` package (test/utils.py, the test files themselves)
    try:
        mod = importlib.import_module(f"test.{name}")
        assert "datasketch_amd" in sys.modules["datasketch"].__name__
        suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
        assert suite.countTestCases() > 0
        result = unittest.TextTestRunner(verbosity=0).run(suite)
        problems = [f"{t}: {tb.splitlines()[-1]}" for t, tb in result.failures + result.errors]
        assert not problems, problems
    finally:
        sys.path.remove(REFERENCE)
        _restore(saved)


CONSUMER_FILES = ["test_lsh", "test_lshforest", "test_lshensemble"]


@pytest.mark.parametrize("name", CONSUMER_FILES)
def test_reference_indexes_run_on_our_sketches(name):
    """"Drops in under MinHashLSH": the reference's index code (lsh.py, lshforest.py, lshensemble.py,
    storage.py -- all out of scope here, host-side control plane) is imported as it is, but with OUR
    MinHash / LeanMinHash / WeightedMinHash modules seeded under the names it imports, and the
    reference's own index tests run on top."""
    from datasketch_amd import b_bit_minhash, hashfunc, lean_minhash, minhash, weighted_minhash

    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "datasketch" or k.startswith("datasketch.") or k == "test" or k.startswith("test.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE)
    try:
        sys.modules["datasketch.minhash"] = minhash
        sys.modules["datasketch.lean_minhash"] = lean_minhash
        sys.modules["datasketch.weighted_minhash"] = weighted_minhash
        sys.modules["datasketch.b_bit_minhash"] = b_bit_minhash
        sys.modules["datasketch.hashfunc"] = hashfunc
        ref = importlib.import_module("datasketch")  # the reference package: indexes + storage are its own
        assert ref.__file__.startswith(REFERENCE) and ref.MinHash is minhash.MinHash
        if "mockredis" not in sys.modules:  # not installed here; only the Redis-storage tests use it
            import types

            sys.modules["mockredis"] = types.ModuleType("mockredis")
            stubbed = True
        else:
            stubbed = False
        try:
            mod = importlib.import_module(f"test.{name}")
            suite = unittest.TestSuite(
                t for group in unittest.defaultTestLoader.loadTestsFromModule(mod) for t in group
                if "redis" not in t.id().lower()  # storage back ends are out of scope and need a server mock
            )
            assert suite.countTestCases() > 0
            result = unittest.TextTestRunner(verbosity=0).run(suite)
        finally:
            if stubbed:
                del sys.modules["mockredis"]
        problems = [f"{t}: {tb.splitlines()[-1]}" for t, tb in result.failures + result.errors]
        assert not problems, problems
    finally:
        sys.path.remove(REFERENCE)
        _restore(saved)
