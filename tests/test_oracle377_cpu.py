"""CPU: the BLS12-377 build of the C oracle (the curve of the reference's own SIPP test and scaling-ipp example, sipp/src/lib.rs:229,
sipp/examples/scaling-ipp.rs:2) against the golden vectors of the independent big-integer model tests/model/bls377_model.py, against
algebraic laws, and on the reference's own test restated."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def o():
    import orclib377
    orclib377.lib()
    return orclib377


@pytest.fixture(scope="module")
def v():
    return json.load(open(os.path.join(HERE, "golden", "bls12_377_vectors.json")))


def _g1(o, pts): return o.g1_array([None if p is None else (int(p[0], 16), int(p[1], 16)) for p in pts])
def _g2(o, pts): return o.g2_array([None if q is None else ((int(q[0][0], 16), int(q[0][1], 16)), (int(q[1][0], 16), int(q[1][1], 16))) for q in pts])


def test_parameters_and_generators(o, v):
    x = 0x8508C00000000001
    assert o.R == x**4 - x**2 + 1 and o.P == (x - 1) ** 2 * o.R // 3 + x and o.P.bit_length() == 377 and o.R.bit_length() == 253
    g = v["generators"]
    g1, g2 = _g1(o, [g["g1"]]), _g2(o, [g["g2"]])
    assert np.array_equal(o.gen_g1(1, 1), g1) and np.array_equal(o.gen_g2(1, 1), g2)
    assert o.ser_g1(g1[0]).hex() == g["ser_g1"] and o.ser_g2(g2[0]).hex() == g["ser_g2"]
    assert o.ser_g1(o.u64(12)).hex() == g["ser_g1_inf"] and o.ser_g2(o.u64(24)).hex() == g["ser_g2_inf"]
    # the y-sign flag (bit 7 of the last byte) distinguishes P from -P; exactly one of the two carries it
    n1 = g1.copy(); n1[0, 6:] = o.fp_to_limbs((o.P - o.limbs_to_fp(g1[0, 6:])) % o.P)
    assert o.ser_g1(n1[0]).hex() == g["ser_g1_neg"]
    assert (o.ser_g1(g1[0])[-1] ^ o.ser_g1(n1[0])[-1]) & 0x80
    n2 = g2.copy()
    for k in (12, 18):
        n2[0, k:k + 6] = o.fp_to_limbs((o.P - o.limbs_to_fp(g2[0, k:k + 6])) % o.P)
    assert o.ser_g2(n2[0]).hex() == g["ser_g2_neg"]


def test_pairing_known_answers_and_laws(o, v):
    g = v["generators"]
    e = o.pairing_product_a(_g1(o, [g["g1"]]), _g2(o, [g["g2"]]))
    assert o.ser_gt(e).hex() == v["pairing_generators"]["gt"]
    bl = v["bilinearity"]; a, b = int(bl["a"], 16), int(bl["b"], 16)
    assert o.ser_gt(o.pairing_product_a(o.gen_g1(a, 1), o.gen_g2(b, 1))).hex() == bl["gt"] == bl["gt_pow"]
    assert np.array_equal(o.gt_pow(e, o.fr_array([a * b % o.R])[0]), o.pairing_product_a(o.gen_g1(a, 1), o.gen_g2(b, 1)))
    one = o.gt_one()
    assert not np.array_equal(e, one) and np.array_equal(o.gt_mul(o.gt_pow(e, o.fr_array([o.R - 1])[0]), e), one)      # e^r = 1
    p8 = v["product8"]
    assert o.ser_gt(o.pairing_product_a(_g1(o, p8["a"]), _g2(o, p8["b"]))).hex() == p8["gt"]
    aa, bb = o.gen_g1(50, 7), o.gen_g2(60, 7)
    acc = one.copy()
    for i in range(7):
        acc = o.gt_mul(acc, o.pairing_product_a(aa[i:i + 1], bb[i:i + 1]))
    assert np.array_equal(acc, o.pairing_product_a(aa, bb))


def test_msm_known_answer(o, v):
    m8 = v["msm8"]; sc = o.fr_array([int(s, 16) for s in m8["scalars"]])
    r1 = o.msm_g1_a(_g1(o, m8["g1_bases"]), sc); r2 = o.msm_g2_a(_g2(o, m8["g2_bases"]), sc)
    assert o.g1_from_row(o.g1_to_affine(r1)) == (int(m8["g1"][0], 16), int(m8["g1"][1], 16))
    assert o.g2_from_row(o.g2_to_affine(r2)) == ((int(m8["g2"][0][0], 16), int(m8["g2"][0][1], 16)), (int(m8["g2"][1][0], 16), int(m8["g2"][1][1], 16)))
    n = 100; bases, s = o.gen_g1(3, n), o.gen_scalars(11, n)
    assert np.array_equal(o.g1_to_affine(o.msm_g1_a(bases, s)), o.g1_to_affine(o.msm_g1_naive(bases, s)))


def test_sipp_known_answer(o, v):
    s4 = v["sipp4"]
    a, b, r = _g1(o, s4["a"]), _g2(o, s4["b"]), o.fr_array([int(x, 16) for x in s4["r"]])
    value = o.product_of_pairings_with_coeffs(a, b, r)
    assert o.ser_gt(value).hex() == s4["value"]
    assert o.sipp_seed_digest(a, b, r, value).hex() == s4["seed_digest"]
    rc, proof, ch = o.sipp_prove(a, b, r, value)
    assert rc == 0
    assert [[o.ser_gt(proof[2 * j]).hex(), o.ser_gt(proof[2 * j + 1]).hex()] for j in range(2)] == s4["proof"]
    assert [hex(o.limbs_to_fr(c)) for c in ch] == s4["challenges"]
    assert o.sipp_verify(a, b, r, value, proof) == 1


def test_reference_prove_and_verify_base_case(o):
    """sipp/src/lib.rs:232-254 on its NATIVE curve: 32 random-looking pairs, prove, verify; a tampered proof is rejected."""
    n = 32
    a, b, r = o.gen_g1(123, n), o.gen_g2(456, n), o.gen_scalars(7, n)
    z = o.product_of_pairings_with_coeffs(a, b, r)
    rc, proof, _ = o.sipp_prove(a, b, r, z)
    assert rc == 0 and o.sipp_verify(a, b, r, z, proof) == 1
    bad = proof.copy(); bad[3] = proof[2]
    assert o.sipp_verify(a, b, r, z, bad) == 0


def test_aggregate_proofs_round_trip_bls12_377(o):
    """the TIPA / aggregation restatement (oracle/tipa.h) on BLS12-377 -- the curve of the reference's aggregation bench: prove -> verify accepts,
    other public inputs are rejected; Fr::from_random_bytes clears the THREE bits above this field's 253-bit modulus [ark-mem]"""
    import helpers as h
    n = 8
    osrs = h.make_srs(n, 0xa1fa + n, 0xbe7a + n, o=o)
    vk, pub, a, b, c = h.fake_groth16(n, 2, seed=n, o=o)
    rc, pf = o.aggregate_proofs(osrs[0], osrs[1], a, b, c)
    assert rc == 0 and o.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub, pf) == 1
    pub2 = pub.copy(); pub2[0, 0] = pub[1, 0]
    assert o.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub2, pf) == 0
    import ctypes
    dig = (ctypes.c_uint8 * 64)(*([0xff] * 64)); out = np.zeros(4, dtype=np.uint64)
    assert o.lib().orc_fr_from_random_bytes(dig, out.ctypes.data_as(ctypes.c_void_p)) == 0        # 2^253 - 1 >= r: rejected
    dig = (ctypes.c_uint8 * 64)(*([0xff] * 31 + [0xe0] + [0] * 32))                                  # the three top bits set, the rest of the top byte clear
    assert o.lib().orc_fr_from_random_bytes(dig, out.ctypes.data_as(ctypes.c_void_p)) == 1 and o.limbs_to_fr(out) == (1 << 248) - 1
