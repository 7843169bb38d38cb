"""GPU parity (-m gpu): the polynomial-commitment applications of ripp_amd/poly_commit (mirror of
ip_proofs/src/applications/poly_commit/mod.rs and transparent.rs) against the oracle-backed restatement tests/model/poly_commit_oracle.py:
commitments, every proof member and both verifiers' verdicts; each side's verifier accepts the other side's proof.  The shapes are the
reference's own tests: bivariate (7, 7) (mod.rs:405-443, transparent.rs:346-379) and the univariate sqrt split (mod.rs:447-472)."""
import os, random, sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "model"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(engine):
    import ripp_amd.poly_commit as pc
    return pc


def _same_g1(orc, a, b):
    a, b = np.asarray(a).reshape(-1, 18), np.asarray(b).reshape(-1, 18)
    return np.array_equal(orc.normalize_g1(np.ascontiguousarray(a)), orc.normalize_g1(np.ascontiguousarray(b)))


def _same_g2(orc, a, b):
    a, b = np.asarray(a).reshape(-1, 36), np.asarray(b).reshape(-1, 36)
    return np.array_equal(orc.normalize_g2(np.ascontiguousarray(a)), orc.normalize_g2(np.ascontiguousarray(b)))


def _poly(rng, orc, n): return [rng.randrange(orc.R) for _ in range(n)]


@pytest.mark.parametrize("degree", [0, 1, 6, 255])
def test_kzg(engine, orc, P, degree):
    """mod.rs:50-119"""
    import poly_commit_oracle as PC
    rng = random.Random(100 + degree)
    alpha, beta = rng.randrange(1, orc.R), rng.randrange(1, orc.R)
    size = max(degree, 7)
    powers, v = P.KZG.setup(alpha, beta, size); epowers, ev_ = PC.kzg_setup(alpha, beta, size)
    assert np.array_equal(powers, epowers)
    assert _same_g1(orc, v["g_beta"], ev_["g_beta"]) and _same_g2(orc, v["h_alpha"], ev_["h_alpha"]) and _same_g1(orc, v["g"], ev_["g"]) and _same_g2(orc, v["h"], ev_["h"])
    p = _poly(rng, orc, degree + 1); z = rng.randrange(orc.R)
    com, proof = P.KZG.commit(powers, p), P.KZG.open(powers, p, z)
    assert _same_g1(orc, com, PC.kzg_commit(epowers, p)) and _same_g1(orc, proof, PC.kzg_open(epowers, p, z))
    val = P.evaluate(p, z); assert val == PC.horner(p, z)
    assert P.KZG.verify(v, com, z, val, proof) and PC.kzg_verify(ev_, com, z, val, proof)
    assert not P.KZG.verify(v, com, z, (val + 1) % orc.R, proof)
    if degree:                                                            # a constant evaluates to `val` everywhere
        assert not P.KZG.verify(v, com, (z + 1) % orc.R, val, proof)
    # trailing zero coefficients do not change anything (DensePolynomial strips them)
    assert _same_g1(orc, P.KZG.commit(powers, p + [0, 0]), com)


def _cmp_ssm(orc, got, exp):
    assert np.array_equal(got["com_gt"], exp["com_gt"]) and np.array_equal(got["tr"], exp["tr"]) and np.array_equal(got["kzg_c"], exp["kzg_c"])
    assert np.array_equal(got["base_b"], exp["base_b"])
    assert _same_g1(orc, got["com_g1"], exp["com_g1"]) and _same_g1(orc, got["base_a"], exp["base_a"])
    assert _same_g2(orc, got["final_ck_a"], exp["final_ck_a"]) and _same_g2(orc, got["opening_a"], exp["opening_a"])


@pytest.mark.parametrize("x_degree,y_degree,n_rows", [(7, 7, 8), (1, 3, 2), (3, 15, 3)])
def test_bivariate_poly_commit(engine, orc, P, x_degree, y_degree, n_rows):
    """mod.rs:405-443; n_rows < x_degree + 1 exercises the zero-polynomial padding (mod.rs:183-187)"""
    import poly_commit_oracle as PC
    rng = random.Random(x_degree * 100 + y_degree)
    alpha, beta = rng.randrange(1, orc.R), rng.randrange(1, orc.R)
    srs = P.BivariatePolynomialCommitment.setup(alpha, beta, x_degree, y_degree); s = PC.bi_setup(alpha, beta, x_degree, y_degree)
    v_srs = srs[0].get_verifier_key()
    ys = [_poly(rng, orc, y_degree + 1) for _ in range(n_rows)]
    bp = P.BivariatePolynomial(ys)
    com, coms = P.BivariatePolynomialCommitment.commit(srs, bp); ecom, ecoms = PC.bi_commit(s, ys)
    assert np.array_equal(com, ecom) and _same_g1(orc, coms, ecoms)
    point = (rng.randrange(orc.R), rng.randrange(orc.R))
    proof = P.BivariatePolynomialCommitment.open(srs, bp, coms, point); eproof = PC.bi_open(s, ys, ecoms, point)
    _cmp_ssm(orc, proof["ip_proof"], eproof["ip_proof"])
    assert _same_g1(orc, proof["y_eval_comm"], eproof["y_eval_comm"]) and _same_g1(orc, proof["kzg_proof"], eproof["kzg_proof"])
    val = bp.evaluate(point); assert val == PC.bi_evaluate(ys, point)
    assert P.BivariatePolynomialCommitment.verify(v_srs, com, point, val, proof)
    assert PC.bi_verify(s["v"], com, point, val, proof)                                   # the oracle accepts the device proof
    assert P.BivariatePolynomialCommitment.verify(v_srs, ecom, point, val, eproof)        # and the device verifier the oracle's
    assert not P.BivariatePolynomialCommitment.verify(v_srs, com, point, (val + 1) % orc.R, proof)
    assert not P.BivariatePolynomialCommitment.verify(v_srs, com, ((point[0] + 1) % orc.R, point[1]), val, proof)
    bad = dict(proof); bad["y_eval_comm"] = proof["kzg_proof"]
    assert not P.BivariatePolynomialCommitment.verify(v_srs, com, point, val, bad)
    srs[0].close()


@pytest.mark.parametrize("degree", [56, 1023, 65535])
def test_univariate_poly_commit(engine, orc, P, degree):
    """mod.rs:447-472 (the reference runs 65535, #[ignore]d for its CPU cost); the oracle side is compared up to 1023"""
    import poly_commit_oracle as PC
    U = P.UnivariatePolynomialCommitment
    rng = random.Random(degree)
    alpha, beta = rng.randrange(1, orc.R), rng.randrange(1, orc.R)
    xd, yd = U.bivariate_degrees(degree)
    assert (xd + 1) * (yd + 1) >= degree + 1
    if degree == 65535: assert (xd, yd) == (15, 4095)
    srs = U.setup(alpha, beta, degree); v_srs = srs[0].get_verifier_key()
    assert U.parse_bivariate_degrees_from_srs(srs) == (xd, yd)
    p = _poly(rng, orc, degree + 1)
    com, coms = U.commit(srs, p)
    z = rng.randrange(orc.R)
    proof = U.open(srs, p, coms, z)
    val = P.evaluate(p, z)
    assert U.verify(v_srs, degree, com, z, val, proof)
    assert not U.verify(v_srs, degree, com, z, (val + 1) % orc.R, proof)
    if degree <= 1023:
        s = PC.bi_setup(alpha, beta, xd, yd); ys = PC.split(p, xd, yd)
        ecom, ecoms = PC.bi_commit(s, ys)
        assert np.array_equal(com, ecom) and _same_g1(orc, coms, ecoms)
        eproof = PC.bi_open(s, ys, ecoms, (pow(z, yd + 1, orc.R), z))
        _cmp_ssm(orc, proof["ip_proof"], eproof["ip_proof"])
        assert _same_g1(orc, proof["kzg_proof"], eproof["kzg_proof"])
        assert PC.bi_verify(s["v"], com, (pow(z, yd + 1, orc.R), z), val, proof)
    srs[0].close()


def _cmp_gipa(orc, inst, proof, model):
    """generic-GIPA proof (reversed round order) against the model's (steps, transcript, base, ck_base)"""
    import gipa_generic_oracle as M
    from test_gpu_gipa_generic import _eq_out
    steps, tr, base, _ = model
    ip, lmc, rmc, t = inst
    outs = (M.COMMIT[lmc][2], M.COMMIT[rmc][2], t)
    rounds = len(steps); assert len(proof["r_commitment_steps"]) == rounds
    for k in range(rounds):
        got = proof["r_commitment_steps"][rounds - 1 - k]
        for side in range(2):
            assert _eq_out(orc, outs[0], got[side][0], steps[k][side][0])
            assert _eq_out(orc, outs[2], got[side][2][0], steps[k][side][2])
    assert _eq_out(orc, M.COMMIT[lmc][0], proof["r_base"][0], base[0]) and _eq_out(orc, "FR", proof["r_base"][1], base[1])


@pytest.mark.parametrize("x_degree,y_degree", [(7, 7), (1, 3)])
def test_transparent_bivariate_poly_commit(engine, orc, P, x_degree, y_degree):
    """transparent.rs:346-379"""
    import poly_commit_oracle as PC
    T = P.transparent.BivariatePolynomialCommitment
    rng = random.Random(x_degree * 10 + y_degree)
    ck = T.setup(700, 900, x_degree, y_degree)
    eck1, eck2 = orc.to_jac_g1(orc.gen_g1(700, y_degree + 1)), orc.to_jac_g2(orc.gen_g2(900, x_degree + 1))
    assert np.array_equal(ck[0], eck1) and np.array_equal(ck[1], eck2)
    ys = [_poly(rng, orc, y_degree + 1) for _ in range(x_degree + 1)]
    bp = P.BivariatePolynomial(ys)
    com, coms = T.commit(ck, bp); ecom, ecoms = PC.tr_commit(eck1, eck2, ys)
    assert np.array_equal(com, ecom) and _same_g1(orc, coms, ecoms)
    point = (rng.randrange(orc.R), rng.randrange(orc.R))
    proof = T.open(ck, bp, coms, point); eproof = PC.tr_open(eck1, eck2, ys, ecoms, point)
    assert _same_g1(orc, proof["y_eval_comm"], eproof["y_eval_comm"])
    _cmp_gipa(orc, PC.SECOND_TIER, proof["second_tier_ip_proof"], eproof["second"])
    _cmp_gipa(orc, PC.FIRST_TIER, proof["first_tier_ip_proof"], eproof["first"])
    val = bp.evaluate(point)
    assert T.verify(ck, com, point, val, proof) and PC.tr_verify(eck1, eck2, ecom, point, val, eproof)
    assert not T.verify(ck, com, point, (val + 1) % orc.R, proof)
    assert not T.verify(ck, com, (point[0], (point[1] + 1) % orc.R), val, proof)


def test_transparent_univariate_poly_commit(engine, orc, P):
    """transparent.rs:383-413 at degree 255 (the reference's constant is 65535, #[ignore]d)"""
    U = P.transparent.UnivariatePolynomialCommitment
    degree = 255
    assert U.bivariate_degrees(degree) == (3, 63) and U.bivariate_degrees(65535) == (63, 1023)
    rng = random.Random(5)
    ck = U.setup(11, 13, degree)
    p = _poly(rng, orc, degree + 1)
    com, coms = U.commit(ck, p)
    z = rng.randrange(orc.R)
    proof = U.open(ck, p, coms, z)
    val = P.evaluate(p, z)
    assert U.verify(ck, com, z, val, proof)
    assert not U.verify(ck, com, z, (val + 1) % orc.R, proof)
