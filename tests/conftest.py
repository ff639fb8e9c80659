"""pytest configuration: the `gpu` marker, repo root on sys.path, golden-fixture loader."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden():
    """Fixtures produced by the real reference (oracle/gen_golden.py)."""
    d = os.path.join(ROOT, "tests", "golden")
    arrays = dict(np.load(os.path.join(d, "golden.npz")))
    with open(os.path.join(d, "golden.json")) as f:
        meta = json.load(f)
    return arrays, meta


def identity(x):
    """Same idiom as the reference's test/utils.py:4-6 fake_hash_func."""
    return x
