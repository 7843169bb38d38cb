"""CPU checks of the oracle's TIPA / TIPAWithSSM / Groth16-aggregation restatement (oracle/tipa.h).

The prover builds its KZG openings from the coefficient form + synthetic division (tipa/mod.rs:304-337, 406-422) while the
verifier uses the product form and the pairing equations (tipa/mod.rs:340-404), so prove -> verify round trips check the two
independent derivations against each other; tampering must be rejected.
"""
import hashlib
import random

import numpy as np
import pytest

import helpers as h
import orclib as o


def test_fr_from_random_bytes_matches_integer_definition():
    rng = random.Random(7); seen_none = False
    for _ in range(200):
        d = bytes(rng.randrange(256) for _ in range(64))
        v = int.from_bytes(d[:32], "little") & ((1 << 255) - 1)
        got = o.fr_from_random_bytes(d)
        if v >= o.R:
            assert got is None; seen_none = True
        else:
            assert got is not None and o.limbs_to_fr(got) == v
    assert seen_none        # about 9 % of digests are rejected, the nonce-retry path is live


def _tipp_instance(n, seed):
    srs = h.make_srs(n, 0x1234567 + seed, 0x89abcdef + seed)
    ck_a, ck_b = h.commitment_keys(srs)
    m_a = o.blind_g1(o.gen_g1(3 + seed, n), 11); m_b = o.blind_g2(o.gen_g2(5 + seed, n), 12)
    return srs, m_a, m_b, ck_a, ck_b


@pytest.mark.parametrize("n", [2, 4, 16])
def test_tipa_tipp_round_trip(n):
    srs, m_a, m_b, ck_a, ck_b = _tipp_instance(n, n)
    one = o.fr_array([1])[0]
    rc, pf = o.tipa_tipp_prove(srs[0], srs[1], m_a, m_b, ck_a, ck_b, one); assert rc == 0
    com = [o.pairing_product_j(m_a, ck_a)[1], o.pairing_product_j(ck_b, m_b)[1], o.pairing_product_j(m_a, m_b)[1]]
    assert o.tipa_tipp_verify(*h.verifier_srs(srs), com, pf, one) == 1
    bad = dict(pf); bad["opening_a"] = pf["opening_b"].copy().repeat(2)[:36].copy()
    assert o.tipa_tipp_verify(*h.verifier_srs(srs), com, bad, one) == 0
    bad = dict(pf); bad["final_ck_b"] = pf["base_a"].copy()
    assert o.tipa_tipp_verify(*h.verifier_srs(srs), com, bad, one) == 0
    com2 = [com[0], com[1], o.gt_mul(com[2], com[2])]
    assert o.tipa_tipp_verify(*h.verifier_srs(srs), com2, pf, one) == 0


def test_tipa_tipp_srs_shift():
    """prove_with_srs_shift (tipa/mod.rs:176-231): ck_a pre-scaled by r^-i, opening shifted by r^-1."""
    n = 8; srs, m_a, m_b, ck_a, ck_b = _tipp_instance(n, 3)
    r = 0x1d2c3b4a59687766554433221100ffeeddccbbaa99887766554433221100 % o.R
    rinv = [pow(r, -i, o.R) for i in range(n)]
    ck_a_r = np.stack([o.to_jac_g2(o.g2_mul_a(o.g2_to_affine(ck_a[i]), o.fr_array([rinv[i]])[0]))[0] for i in range(n)])
    fr_r = o.fr_array([r])[0]
    rc, pf = o.tipa_tipp_prove(srs[0], srs[1], m_a, m_b, ck_a_r, ck_b, fr_r); assert rc == 0
    com = [o.pairing_product_j(m_a, ck_a_r)[1], o.pairing_product_j(ck_b, m_b)[1], o.pairing_product_j(m_a, m_b)[1]]
    assert o.tipa_tipp_verify(*h.verifier_srs(srs), com, pf, fr_r) == 1
    assert o.tipa_tipp_verify(*h.verifier_srs(srs), com, pf, o.fr_array([1])[0]) == 0


@pytest.mark.parametrize("n", [2, 8])
def test_tipa_ssm_round_trip(n):
    srs = h.make_srs(n, 77, 99); ck_a, _ = h.commitment_keys(srs)
    m_a = o.blind_g1(o.gen_g1(9, n), 5)
    b = 0x5eed5eed5eed5eed5eed % o.R
    m_b = o.fr_array([pow(b, i, o.R) for i in range(n)])
    rc, pf = o.tipa_ssm_prove(srs[1], m_a, m_b, ck_a); assert rc == 0
    com_a = o.pairing_product_j(m_a, ck_a)[1]; com_t = o.msm_g1_j(m_a, m_b)[1]
    g, hh, g_beta, _ = h.verifier_srs(srs)
    fb = o.fr_array([b])[0]
    assert o.tipa_ssm_verify(g, hh, g_beta, com_a, com_t, fb, pf) == 1
    assert o.tipa_ssm_verify(g, hh, g_beta, com_a, com_t, o.fr_array([b + 1])[0], pf) == 0
    bad = dict(pf); bad["base_a"] = pf["com_g1"][0].copy()
    assert o.tipa_ssm_verify(g, hh, g_beta, com_a, com_t, fb, bad) == 0


def test_aggregate_proofs_accepts_valid_and_rejects_invalid():
    n, m = 8, 3
    srs = h.make_srs(n, 0xa1fa, 0xbe7a)
    vk, pub, a, b, c = h.fake_groth16(n, m, seed=3)
    rc, pf = o.aggregate_proofs(srs[0], srs[1], a, b, c); assert rc == 0
    assert o.verify_aggregate_proof(h.verifier_srs(srs), vk, pub, pf) == 1
    pub2 = pub.copy(); pub2[5, 1] = pub[4, 1]
    assert o.verify_aggregate_proof(h.verifier_srs(srs), vk, pub2, pf) == 0
    # one invalid Groth16 proof among the n: aggregation still runs, the verifier must reject
    c2 = c.copy(); c2[2] = c[3]
    rc, pf2 = o.aggregate_proofs(srs[0], srs[1], a, b, c2); assert rc == 0
    assert o.verify_aggregate_proof(h.verifier_srs(srs), vk, pub, pf2) == 0
    # the aggregation challenge is the documented hash (groth16_aggregation.rs:105-116) fed to from_random_bytes
    nonce = 0
    while True:
        d = hashlib.blake2b(nonce.to_bytes(8, "big") + o.ser_gt(pf.field("com_a")) + o.ser_gt(pf.field("com_b")) + o.ser_gt(pf.field("com_c"))).digest()
        v = int.from_bytes(d[:32], "little") & ((1 << 255) - 1)
        if v < o.R:
            break
        nonce += 1
    assert o.limbs_to_fr(pf.field("r")) == v


def test_tipa_proof_accepted_by_the_independent_python_verifier():
    """A TIPA proof made by the oracle (n = 4, SRS shift r != 1) must be accepted by tests/model/tipa_model.py -- a verifier written from
    the protocol on Python integers with hashlib's BLAKE2b -- and a tampered one rejected.  This pins transcript layout,
    from_random_bytes, the KZG equations and the transcript order independently of the C/HIP code."""
    import tipa_model as T
    n = 4; srs = h.make_srs(n, 0xabc1, 0xdef2); ck_a, ck_b = h.commitment_keys(srs)
    r = 0x1d2c3b4a59687766554433221100ffeeddccbbaa998877 % o.R
    ck_a = np.stack([o.to_jac_g2(o.g2_mul_a(o.g2_to_affine(ck_a[i]), o.fr_array([pow(r, -i, o.R)])[0]))[0] for i in range(n)])
    m_a, m_b = o.blind_g1(o.gen_g1(3, n), 1), o.blind_g2(o.gen_g2(5, n), 2)
    rc, pf = o.tipa_tipp_prove(srs[0], srs[1], m_a, m_b, ck_a, ck_b, o.fr_array([r])[0]); assert rc == 0
    gt = lambda f: T.gt_from_tower([o.limbs_to_fp(f[6 * i:6 * i + 6]) for i in range(12)])
    p1 = lambda pj: o.g1_from_row(o.g1_to_affine(np.ascontiguousarray(pj))); p2 = lambda pj: o.g2_from_row(o.g2_to_affine(np.ascontiguousarray(pj)))
    com = [gt(o.pairing_product_j(m_a, ck_a)[1]), gt(o.pairing_product_j(ck_b, m_b)[1]), gt(o.pairing_product_j(m_a, m_b)[1])]
    rounds = len(pf["steps"]) // 6
    steps = [[gt(pf["steps"][6 * k + j]) for j in range(6)] for k in range(rounds)]
    g, hh, g_beta, h_alpha = h.verifier_srs(srs)
    vs = (p1(g), p2(hh), p1(g_beta), p2(h_alpha))
    args = (p1(pf["base_a"]), p2(pf["base_b"]), p2(pf["final_ck_a"]), p1(pf["final_ck_b"]), p2(pf["opening_a"]), p1(pf["opening_b"]))
    assert T.verify_tipa_tipp(vs, com, steps, *args, r)
    assert not T.verify_tipa_tipp(vs, com, steps, *args, 1)


def test_golden_tipa_vector_oracle_and_wire_format(vectors):
    """The `tipa4` golden vector was produced and verified by the big-integer model; the oracle must reproduce its transcript, KZG
    challenge and -- through the product's host-only serialiser -- both wire images byte for byte."""
    import ripp_amd
    v = vectors["tipa4"]
    srs, m_a, m_b, ck_a, ck_b, r_shift = h.tipa4_instance(v)
    rc, pf = o.tipa_tipp_prove(srs[0], srs[1], m_a, m_b, ck_a, ck_b, r_shift); assert rc == 0
    assert [hex(o.limbs_to_fr(t)) for t in pf["tr"]] == v["transcript"] and hex(o.limbs_to_fr(pf["kzg_c"])) == v["kzg_challenge"]
    assert [o.ser_gt(o.pairing_product_j(m_a, ck_a)[1]).hex(), o.ser_gt(o.pairing_product_j(ck_b, m_b)[1]).hex(), o.ser_gt(o.pairing_product_j(m_a, m_b)[1]).hex()] == v["com"]
    assert ripp_amd.ser_tipa_tipp_proof(pf, compress=False).hex() == v["proof_uncompressed"]
    assert ripp_amd.ser_tipa_tipp_proof(pf, compress=True).hex() == v["proof_compressed"]


def test_golden_aggregate_vector_oracle(vectors):
    """`aggregate4`: aggregate_proofs on four (A, B, C) triples, produced by the big-integer model (both sub-proofs accepted by the model's
    verifiers).  The oracle must reproduce r, the commitments, ip_ab, agg_c and both sub-proof wire images byte for byte."""
    import ripp_amd
    v = vectors["aggregate4"]
    srs = h.make_srs(4, int(v["alpha"], 16), int(v["beta"], 16))
    rc, pf = o.aggregate_proofs(srs[0], srs[1], h.g1arr(v["a"]), h.g2arr(v["b"]), h.g1arr(v["c"])); assert rc == 0
    h.check_aggregate_golden(v, pf, o.ser_gt, lambda j: o.ser_g1(o.g1_to_affine(j)), ripp_amd.ser_tipa_tipp_proof, ripp_amd.ser_tipa_ssm_proof)
