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


@pytest.fixture(scope="session")
def sipp_2p20(engine, orc, tmp_path_factory):
    """The headline statement (bench.py's: a_i = (1000 + i) G1, b_i = (2000 + i) G2, r_i = SplitMix64(0), n = 2^20) with the ORACLE's
    proof of it -- one ~80 s oracle run per session, shared by the one-GPU and the two-rank tests (the latter load it from `path`)."""
    import numpy as np
    n = 1 << 20
    a, b, r = engine.synth_g1(1000, n), engine.synth_g2(2000, n), engine.synth_fr(0, n)
    value = engine.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(value, orc.product_of_pairings_with_coeffs(a, b, r))
    rc, eproof, ech = orc.sipp_prove(a, b, r, value)
    assert rc == 0
    path = str(tmp_path_factory.mktemp("sipp2p20") / "oracle_proof.npz")
    np.savez(path, value=value, proof=eproof, ch=ech)
    return {"n": n, "a": a, "b": b, "r": r, "value": value, "proof": eproof, "ch": ech, "path": path}
