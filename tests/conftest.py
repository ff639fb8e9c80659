"""pytest configuration: the `gpu` marker, repo root on sys.path, golden-fixture loader."""
import json
import os
import sys

import numpy as np
import pytest

# the suite exercises the RCCL path in the same process as torch.distributed (gloo): RCCL is loaded with the
# context, before torch's ROCm libraries can get in (see mhx_comm_preload in include/mhx.h)
os.environ.setdefault("MHX_PRELOAD_RCCL", "1")

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
