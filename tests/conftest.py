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


def _oracle_proof_2p20(tag, o, a, b, r):
    """The CPU oracle's proof of the headline statement on curve `tag` ("381" / "377"): loaded from tests/golden/sipp_2p20_oracle_proofs.npz -- the
    committed output of one oracle run (tests/golden/gen_sipp_2p20_oracle_proofs.py; ~80-110 s of 16 CPUs per curve, 40 % of the GPU suite when computed
    live) -- or recomputed with RIPP_TEST_LIVE_ORACLE=1.  Either way the oracle's VERIFIER then checks the GPU's proof against the statement itself."""
    import numpy as np
    path = os.path.join(HERE, "golden", "sipp_2p20_oracle_proofs.npz")
    if os.environ.get("RIPP_TEST_LIVE_ORACLE") or not os.path.exists(path):
        value = o.product_of_pairings_with_coeffs(a, b, r)
        rc, proof, ch = o.sipp_prove(a, b, r, value)
        assert rc == 0
        return value, proof, ch
    z = np.load(path)
    # the fixture is a cache of THIS oracle's output: its canary (the same generators and prover at n = 2^10, tests/golden/gen_sipp_2p20_oracle_proofs.py) must still
    # come out of the oracle as loaded -- a stale file fails here instead of passing or failing for the wrong reason
    m = 1 << 10
    ca, cb, cr = o.gen_g1(1000, m), o.gen_g2(2000, m), o.gen_scalars(0, m)
    cv = o.product_of_pairings_with_coeffs(ca, cb, cr)
    rc, cproof, cch = o.sipp_prove(ca, cb, cr, cv)
    assert rc == 0 and np.array_equal(cv, z["canary_value_" + tag]) and np.array_equal(cproof, z["canary_proof_" + tag]) and np.array_equal(cch, z["canary_ch_" + tag]), \
        "tests/golden/sipp_2p20_oracle_proofs.npz is stale for BLS12-" + tag + ": regenerate it (tests/golden/gen_sipp_2p20_oracle_proofs.py)"
    return z["value_" + tag], z["proof_" + tag], z["ch_" + tag]


@pytest.fixture(scope="session")
def sipp_2p20(engine, orc, tmp_path_factory):
    """The headline statement (bench.py's: a_i = (1000 + i) G1, b_i = (2000 + i) G2, r_i = SplitMix64(0), n = 2^20) with the ORACLE's
    proof of it (_oracle_proof_2p20), shared by the one-GPU and the sharded tests (the latter load it from `path`).  The claimed value is computed by
    the engine AND by the oracle live and must agree with the fixture's."""
    import numpy as np
    n = 1 << 20
    a, b, r = engine.synth_g1(1000, n), engine.synth_g2(2000, n), engine.synth_fr(0, n)
    value = engine.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(value, orc.product_of_pairings_with_coeffs(a, b, r))
    evalue, eproof, ech = _oracle_proof_2p20("381", orc, a, b, r)
    assert np.array_equal(value, evalue)
    path = str(tmp_path_factory.mktemp("sipp2p20") / "oracle_proof.npz")
    np.savez(path, value=value, proof=eproof, ch=ech)
    return {"n": n, "a": a, "b": b, "r": r, "value": value, "proof": eproof, "ch": ech, "path": path}
