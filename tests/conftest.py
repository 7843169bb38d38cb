import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "model")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(HERE, "golden", "bls12_381_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import orclib
    orclib.lib()
    return orclib


def _has_gpu():
    try:
        from ripp_amd._lib import lib
        return lib().ripp_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def engine():
    """The HIP engine; GPU tests FAIL (not skip) if the library is missing, and skip only when no device exists."""
    import ripp_amd as R
    if not _has_gpu():
        pytest.skip("no HIP device in this environment")
    R.init(0)
    return R
