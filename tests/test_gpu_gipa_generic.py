"""GPU parity (-m gpu): the generic GIPA of ripp_amd/gipa.py -- every instantiation of the reference's own GIPA tests
(ip_proofs/src/gipa.rs:470-561) plus GIPAWithSSM (tipa/structured_scalar_message.rs:56-128) -- against the oracle-backed restatement
tests/model/gipa_generic_oracle.py: commitments of every round, transcript and base case equal; both verifiers accept each other's
proof and reject a tampered one.  The TIPP instantiation is also compared with the FUSED device prover of the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(engine):
    import ripp_amd.gipa as g
    return g


def _fr_ints(orc, arr): return [orc.limbs_to_fr(x) for x in arr]


def _eq_out(orc, tag, got, exp):
    import gipa_generic_oracle as M
    if tag == "FR":
        return (got if isinstance(got, int) else orc.limbs_to_fr(got)) == exp % orc.R
    return M.same(tag, np.asarray(got), np.asarray(exp))


def _compare(orc, inst, proof, aux, steps, tr, base, ck_base):
    import gipa_generic_oracle as M
    ip, lmc, rmc, t = inst
    outs = (M.COMMIT[lmc][2], M.COMMIT[rmc][2], t)
    rounds = len(steps)
    assert len(proof["r_commitment_steps"]) == rounds
    for k in range(rounds):
        got = proof["r_commitment_steps"][rounds - 1 - k]                 # reference order is reversed
        for side in range(2):
            assert _eq_out(orc, outs[0], got[side][0], steps[k][side][0])
            assert _eq_out(orc, outs[1], got[side][1], steps[k][side][1])
            assert len(got[side][2]) == 1 and _eq_out(orc, outs[2], got[side][2][0], steps[k][side][2])
    assert [orc.limbs_to_fr(c) for c in aux["r_transcript"]] == tr[::-1]
    assert _eq_out(orc, M.COMMIT[lmc][0], proof["r_base"][0], base[0]) and _eq_out(orc, M.COMMIT[rmc][0], proof["r_base"][1], base[1])
    assert _eq_out(orc, M.COMMIT[lmc][1], aux["ck_base"][0], ck_base[0])
    if M.COMMIT[rmc][1] != "UNIT":
        assert _eq_out(orc, M.COMMIT[rmc][1], aux["ck_base"][1], ck_base[1])


@pytest.mark.parametrize("n", [1, 2, 8, 64])
def test_pairing_inner_product_gipa(engine, orc, G, n):
    """gipa.rs:470-497, and bit-equality with the fused ripp_gipa_tipp_prove."""
    import gipa_generic_oracle as M
    inst = ("PAIR", "AFGHO1", "AFGHO2", "GT")
    m_a, m_b = orc.blind_g1(orc.gen_g1(11, n), 1), orc.blind_g2(orc.gen_g2(22, n), 2)
    ck_a, ck_b = orc.blind_g2(orc.gen_g2(33, n), 3), orc.blind_g1(orc.gen_g1(44, n), 4)
    gipa = G.GIPA(G.PairingIP, G.AFGHOCommitmentG1, G.AFGHOCommitmentG2, G.IdentityCommitment(G.GT))
    t = G.PairingIP.inner_product(m_a, m_b)
    com = (G.AFGHOCommitmentG1.commit(ck_a, m_a), G.AFGHOCommitmentG2.commit(ck_b, m_b), [t])
    proof = gipa.prove((m_a, m_b, t), (ck_a, ck_b, None), com)
    proof2, aux = gipa.prove_with_aux((m_a, m_b), (ck_a, ck_b, [None]))
    _compare(orc, inst, proof2, aux, *M.prove(inst, m_a, m_b, ck_a, ck_b))
    assert gipa.verify((ck_a, ck_b, None), com, proof)
    steps, tr, base, _ = M.prove(inst, m_a, m_b, ck_a, ck_b)
    assert M.verify(inst, ck_a, ck_b, [com[0], com[1], t], [(tuple(s[0][:2]) + (s[0][2][0],), tuple(s[1][:2]) + (s[1][2][0],)) for s in proof["r_commitment_steps"][::-1]], proof["r_base"])
    # fused prover: same commitments (round order), same transcript, same base
    fp, faux, extra = engine.GIPA_TIPP.prove_with_aux(m_a, m_b, ck_a, ck_b)
    rounds = n.bit_length() - 1
    for k in range(rounds):
        got = proof["r_commitment_steps"][rounds - 1 - k]
        flat = [got[0][0], got[0][1], got[0][2][0], got[1][0], got[1][1], got[1][2][0]]
        assert np.array_equal(np.stack(flat), extra["round_order_steps"][6 * k:6 * k + 6])
    if rounds:
        assert np.array_equal(np.stack(aux["r_transcript"]), faux["r_transcript"])
        bad = dict(proof); st = list(bad["r_commitment_steps"]); st[0] = (st[0][1], st[0][0]); bad["r_commitment_steps"] = st
        assert not gipa.verify((ck_a, ck_b, None), com, bad)


@pytest.mark.parametrize("n", [1, 2, 8, 64])
def test_multiexponentiation_inner_product_gipa(engine, orc, G, n):
    """gipa.rs:499-530: IP = MSM over G1, LMC = AFGHO-G1, RMC = Pedersen over G1."""
    import gipa_generic_oracle as M
    inst = ("MEXP1", "AFGHO1", "PED1", "G1")
    m_a, m_b = orc.blind_g1(orc.gen_g1(11, n), 1), orc.gen_scalars(5, n)
    ck_a, ck_b = orc.blind_g2(orc.gen_g2(33, n), 3), orc.blind_g1(orc.gen_g1(44, n), 4)
    gipa = G.GIPA(G.MultiexpIPG1, G.AFGHOCommitmentG1, G.PedersenCommitmentG1, G.IdentityCommitment(G.G1))
    t = G.MultiexpIPG1.inner_product(m_a, m_b)
    com = (G.AFGHOCommitmentG1.commit(ck_a, m_a), G.PedersenCommitmentG1.commit(ck_b, m_b), [t])
    proof, aux = gipa.prove_with_aux((m_a, m_b), (ck_a, ck_b, [None]))
    steps, tr, base, ck_base = M.prove(inst, m_a, _fr_ints(orc, m_b), ck_a, ck_b)
    _compare(orc, inst, proof, aux, steps, tr, base, ck_base)
    assert gipa.verify((ck_a, ck_b, None), com, gipa.prove((m_a, m_b, t), (ck_a, ck_b, None), com))
    assert M.verify(inst, ck_a, ck_b, [com[0], com[1], t], steps, base)
    if n > 1:
        wrong = (com[0], com[1], [G.G1.add(t, t)])
        assert not gipa.verify((ck_a, ck_b, None), wrong, proof)


@pytest.mark.parametrize("n", [1, 2, 8, 64])
def test_scalar_inner_product_gipa(engine, orc, G, n):
    """gipa.rs:532-561: IP = scalar product, both commitments Pedersen over G2."""
    import gipa_generic_oracle as M
    inst = ("SCAL", "PED2", "PED2", "FR")
    m_a, m_b = orc.gen_scalars(5, n), orc.gen_scalars(6, n)
    ck_a, ck_b = orc.blind_g2(orc.gen_g2(33, n), 3), orc.blind_g2(orc.gen_g2(55, n), 4)
    gipa = G.GIPA(G.ScalarIP, G.PedersenCommitmentG2, G.PedersenCommitmentG2, G.IdentityCommitment(G.Fr))
    t = G.ScalarIP.inner_product(m_a, m_b)
    com = (G.PedersenCommitmentG2.commit(ck_a, m_a), G.PedersenCommitmentG2.commit(ck_b, m_b), [t])
    proof, aux = gipa.prove_with_aux((m_a, m_b), (ck_a, ck_b, [None]))
    steps, tr, base, ck_base = M.prove(inst, _fr_ints(orc, m_a), _fr_ints(orc, m_b), ck_a, ck_b)
    _compare(orc, inst, proof, aux, steps, tr, base, ck_base)
    assert gipa.verify((ck_a, ck_b, None), com, gipa.prove((m_a, m_b, t), (ck_a, ck_b, None), com))
    assert M.verify(inst, ck_a, ck_b, [com[0], com[1], orc.limbs_to_fr(t)], steps, base)


@pytest.mark.parametrize("n", [2, 8, 64])
def test_gipa_with_structured_scalar_message(engine, orc, G, n):
    """structured_scalar_message.rs:56-128: m_b = (1, b, b^2, ...), not committed to; the verifier rebuilds the final scalar."""
    import gipa_generic_oracle as M
    inst = ("MEXP1", "AFGHO1", "SSM", "G1")
    b = 0x1234567890ABCDEF1234567890ABCDEF % orc.R
    bs = [pow(b, i, orc.R) for i in range(n)]
    m_a, m_b = orc.blind_g1(orc.gen_g1(11, n), 1), orc.fr_array(bs)
    ck_a = orc.blind_g2(orc.gen_g2(33, n), 3)
    ssm = G.GIPAWithSSM(G.MultiexpIPG1, G.AFGHOCommitmentG1, G.IdentityCommitment(G.G1))
    proof = ssm.prove_with_structured_scalar_message((m_a, m_b), (ck_a, None))
    t = G.MultiexpIPG1.inner_product(m_a, m_b)
    com = (G.AFGHOCommitmentG1.commit(ck_a, m_a), [t])
    assert ssm.verify_with_structured_scalar_message((ck_a, None), com, G.fr_from_int(b), proof)
    steps, tr, base, ck_base = M.prove(inst, m_a, bs, ck_a, [None] * n)
    _, aux = ssm.gipa.prove_with_aux((m_a, m_b), (ck_a, [None] * n, [None]))
    _compare(orc, inst, proof, aux, steps, tr, base, ck_base)
    assert M.verify(inst, ck_a, [None] * n, [com[0], 0, t], steps, base, scalar_b=b)
    assert not ssm.verify_with_structured_scalar_message((ck_a, None), com, G.fr_from_int(b + 1), proof)
