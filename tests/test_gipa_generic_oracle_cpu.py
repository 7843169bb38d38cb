"""CPU: the oracle-backed generic GIPA restatement (tests/model/gipa_generic_oracle.py) is self-consistent for every instantiation of the
reference's tests (ip_proofs/src/gipa.rs:470-561) and for GIPAWithSSM: prove -> verify accepts, a wrong commitment / tampered step / wrong
structured scalar is rejected, and the TIPP instantiation reproduces the C oracle's own (fused) GIPA prover."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def M(orc):
    import gipa_generic_oracle as m
    return m


def test_tipp_instantiation_equals_fused_oracle_prover(orc, M):
    n = 8
    m_a, m_b = orc.blind_g1(orc.gen_g1(11, n), 1), orc.blind_g2(orc.gen_g2(22, n), 2)
    ck_a, ck_b = orc.blind_g2(orc.gen_g2(33, n), 3), orc.blind_g1(orc.gen_g1(44, n), 4)
    inst = ("PAIR", "AFGHO1", "AFGHO2", "GT")
    steps, tr, base, ck_base = M.prove(inst, m_a, m_b, ck_a, ck_b)
    rc, fsteps, ftr, ba, bb, ka, kb = orc.gipa_tipp_prove(m_a, m_b, ck_a, ck_b)
    assert rc == 0
    flat = np.stack([x for s in steps for side in s for x in side])
    assert np.array_equal(flat, fsteps) and [orc.limbs_to_fr(c) for c in ftr] == tr
    assert M.same("G1", base[0], ba) and M.same("G2", base[1], bb) and M.same("G2", ck_base[0], ka) and M.same("G1", ck_base[1], kb)
    com = [orc.pairing_product_j(m_a, ck_a)[1], orc.pairing_product_j(ck_b, m_b)[1], orc.pairing_product_j(m_a, m_b)[1]]
    assert M.verify(inst, ck_a, ck_b, com, steps, base)
    assert not M.verify(inst, ck_a, ck_b, [com[0], com[1], com[0]], steps, base)


@pytest.mark.parametrize("n", [1, 4])
def test_multiexp_and_scalar_instantiations(orc, M, n):
    g1, g2 = orc.blind_g1(orc.gen_g1(11, n), 1), orc.blind_g2(orc.gen_g2(33, n), 3)
    k1, k2 = orc.blind_g1(orc.gen_g1(44, n), 4), orc.blind_g2(orc.gen_g2(55, n), 5)
    s1 = [orc.limbs_to_fr(x) for x in orc.gen_scalars(5, n)]; s2 = [orc.limbs_to_fr(x) for x in orc.gen_scalars(6, n)]
    inst = ("MEXP1", "AFGHO1", "PED1", "G1")
    steps, tr, base, _ = M.prove(inst, g1, s1, g2, k1)
    com = [M.COMMIT["AFGHO1"][3](g2, g1), M.COMMIT["PED1"][3](k1, s1), M.inner_product("MEXP1", g1, s1)]
    assert M.verify(inst, g2, k1, com, steps, base)
    if n > 1:
        bad = [(steps[0][1], steps[0][0])] + steps[1:]
        assert not M.verify(inst, g2, k1, com, bad, base)
    inst = ("SCAL", "PED2", "PED2", "FR")
    steps, tr, base, _ = M.prove(inst, s1, s2, g2, k2)
    com = [M.COMMIT["PED2"][3](g2, s1), M.COMMIT["PED2"][3](k2, s2), M.inner_product("SCAL", s1, s2)]
    assert M.verify(inst, g2, k2, com, steps, base)
    assert not M.verify(inst, g2, k2, [com[0], com[1], (com[2] + 1) % orc.R], steps, base)


def test_gipa_with_ssm(orc, M):
    n, b = 8, 0xABCDEF0123456789
    bs = [pow(b, i, orc.R) for i in range(n)]
    g1, g2 = orc.blind_g1(orc.gen_g1(11, n), 1), orc.blind_g2(orc.gen_g2(33, n), 3)
    inst = ("MEXP1", "AFGHO1", "SSM", "G1")
    steps, tr, base, _ = M.prove(inst, g1, bs, g2, [None] * n)
    com = [M.COMMIT["AFGHO1"][3](g2, g1), 0, M.inner_product("MEXP1", g1, bs)]
    assert M.verify(inst, g2, [None] * n, com, steps, base, scalar_b=b)
    assert not M.verify(inst, g2, [None] * n, com, steps, base, scalar_b=b + 1)
