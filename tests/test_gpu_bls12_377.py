"""GPU parity (-m gpu), BLS12-377 build of the engine (ripp_amd.bls12_377 -> libripp_hip_377.so) against the BLS12-377 build of the oracle
on identical inputs, bit-exact: pairing products (incl. infinities), MSMs, folds, the SIPP prover and verifier -- the reference's own SIPP
test (sipp/src/lib.rs:232-254) and its scaling-ipp harness run on THIS curve."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def E(engine):
    import ripp_amd.bls12_377 as R377
    R377.init(0)
    return R377


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


def test_golden_vectors(E, o, v):
    g = v["generators"]
    g1, g2 = _g1(o, [g["g1"]]), _g2(o, [g["g2"]])
    assert np.array_equal(E.synth_g1(1, 1), g1) and np.array_equal(E.synth_g2(1, 1), g2)
    assert E.ser_g1(g1[0]).hex() == g["ser_g1"] and E.ser_g2(g2[0]).hex() == g["ser_g2"]
    assert E.ser_g1(np.zeros(12, dtype=np.uint64)).hex() == g["ser_g1_inf"]
    assert E.ser_gt(E.product_of_pairings(g1, g2)).hex() == v["pairing_generators"]["gt"]
    p8 = v["product8"]
    assert E.ser_gt(E.product_of_pairings(_g1(o, p8["a"]), _g2(o, p8["b"]))).hex() == p8["gt"]
    s4 = v["sipp4"]
    a, b, r = _g1(o, s4["a"]), _g2(o, s4["b"]), o.fr_array([int(x, 16) for x in s4["r"]])
    value = E.product_of_pairings_with_coeffs(a, b, r)
    assert E.ser_gt(value).hex() == s4["value"] and E.sipp_seed_digest(a, b, r, value).hex() == s4["seed_digest"]
    proof = E.SIPP.prove(a, b, r, value)
    assert [[E.ser_gt(proof[2 * j]).hex(), E.ser_gt(proof[2 * j + 1]).hex()] for j in range(2)] == s4["proof"]
    assert E.SIPP.verify(a, b, r, value, proof)


def test_synthetic_inputs_match_oracle(E, o):
    n = 257
    assert np.array_equal(E.synth_g1(1000, n), o.gen_g1(1000, n)) and np.array_equal(E.synth_g2(2000, n), o.gen_g2(2000, n))
    assert np.array_equal(E.synth_fr(3, n), o.gen_scalars(3, n))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 33, 64, 1000, 1 << 13, (1 << 16) + 3])
def test_pairing_inner_product_vs_oracle(E, o, n):
    """(2^16 + 3: above the VM crossover -- the carry-free throughput kernels k_miller_lines_q / k_line_products_q with the D-type twist's lines.)"""
    a, b = o.gen_g1(50, n), o.gen_g2(60, n)
    if n > 5:
        a[2] = 0; b[4] = 0
    aj, bj = o.blind_g1(a, 1), o.blind_g2(b, 2)
    rc, exp = o.pairing_product_j(aj, bj)
    assert rc == 0 and np.array_equal(E.PairingInnerProduct.inner_product(aj, bj), exp)
    assert np.array_equal(E.product_of_pairings(a, b), o.pairing_product_a(a, b))
    if n == 3:
        with pytest.raises(E.InnerProductError):
            E.PairingInnerProduct.inner_product(aj, bj[:2])


@pytest.mark.parametrize("n", [1, 2, 33, 1 << 10, (1 << 14) + 5])
def test_msm_and_folds_vs_oracle(E, o, n):
    a, b, s = o.gen_g1(31, n), o.gen_g2(41, n), o.gen_scalars(9, n)
    assert np.array_equal(E.normalize_batch_g1(E.MultiexponentiationInnerProductG1.inner_product(o.to_jac_g1(a), s)), o.g1_to_affine(o.msm_g1_a(a, s)).reshape(1, 12))
    assert np.array_equal(E.normalize_batch_g2(E.MultiexponentiationInnerProductG2.inner_product(o.to_jac_g2(b), s)), o.g2_to_affine(o.msm_g2_a(b, s)).reshape(1, 24))
    if n >= 2:
        h = n // 2
        for sc in (s[0], o.fr_array([2**128 - 1])[0], o.fr_array([0])[0]):
            assert np.array_equal(E.fold_g1_affine(a[h:2 * h], a[:h], sc), o.fold_g1_a(a[h:2 * h], a[:h], sc))
            assert np.array_equal(E.fold_g2_affine(b[h:2 * h], b[:h], sc), o.fold_g2_a(b[h:2 * h], b[:h], sc))
        assert np.array_equal(E.scale_g1_affine(a, s), o.scale_g1_a(a, s))


@pytest.mark.parametrize("n", [1, 2, 32, 1 << 10, 1 << 14])
def test_sipp_prove_vs_oracle(E, o, n):
    """n = 32 is the reference's own test size (prove_and_verify_base_case), 2^10 the scaling-ipp plumbing config."""
    a, b, r = o.gen_g1(123, n), o.gen_g2(456, n), o.gen_scalars(7, n)
    value = o.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(E.product_of_pairings_with_coeffs(a, b, r), value)
    proof, ch, _ = E.SIPP.prove_with_stats(a, b, r, value)
    rc, eproof, ech = o.sipp_prove(a, b, r, value)
    assert rc == 0 and np.array_equal(proof, eproof) and np.array_equal(ch, ech)
    if n >= 2:
        assert E.SIPP.verify(a, b, r, value, proof) and o.sipp_verify(a, b, r, value, proof) == 1
        bad = proof.copy(); bad[1] = proof[0]
        assert not E.SIPP.verify(a, b, r, value, bad)


def test_sipp_degenerate_statement_vs_oracle(E, o):
    n = 64
    a, b, r = o.gen_g1(70, n), o.gen_g2(80, n), o.gen_scalars(9, n)
    r[1] = 0; a[2] = 0; b[3] = 0; a[n - 1] = 0; b[n - 1] = 0
    a[5] = a[4]; b[5] = b[4]; r[5] = r[4]
    q = 4 + n // 2
    a[q] = a[4]; b[q] = b[4]; r[q] = r[4]
    value = E.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(value, o.product_of_pairings_with_coeffs(a, b, r))
    proof = E.SIPP.prove(a, b, r, value)
    rc, eproof, _ = o.sipp_prove(a, b, r, value)
    assert rc == 0 and np.array_equal(proof, eproof) and E.SIPP.verify(a, b, r, value, proof)


def test_both_curves_side_by_side(E, engine, orc, o):
    """the two builds are separate libraries with separate engines: BLS12-381 results are unaffected by the 377 engine being live"""
    n = 64
    a, b = orc.gen_g1(5, n), orc.gen_g2(6, n)
    assert np.array_equal(engine.product_of_pairings(a, b), orc.pairing_product_a(a, b))
    a7, b7 = o.gen_g1(5, n), o.gen_g2(6, n)
    assert np.array_equal(E.product_of_pairings(a7, b7), o.pairing_product_a(a7, b7))


@pytest.mark.parametrize("n", [32, 1 << 10])
def test_scaling_ipp_inputs_verbatim(E, o, n):
    """sipp/examples/scaling-ipp.rs:41-51 verbatim: ONE point 2g / 2h and ONE scalar repeated n times, on the example's own curve.
    Every fold then adds equal operands (x P + P): the exceptional cases of the group law; proofs must still equal the oracle's."""
    a, b, r = np.repeat(o.gen_g1(2, 1), n, axis=0), np.repeat(o.gen_g2(2, 1), n, axis=0), np.repeat(o.gen_scalars(0, 1), n, axis=0)
    value = E.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(value, o.product_of_pairings_with_coeffs(a, b, r))
    proof = E.SIPP.prove(a, b, r, value)
    rc, eproof, _ = o.sipp_prove(a, b, r, value)
    assert rc == 0 and np.array_equal(proof, eproof) and E.SIPP.verify(a, b, r, value, proof)


def test_sipp_2p17_endomorphism_paths_vs_oracle(E, o):
    """The BLS12-377 build now carries the GLV (beta, lambda = x^2 - 1) and GLS (psi on the D-type twist, x > 0: no sign flip) constants of ITS
    curve (tools/gen_params.py derives and checks them): round-0 fold tables in the hash window, the x-scaled G2 vector, the look-ahead and
    the GLV / GLS scalar splits of the folds, the per-element scaling and the MSMs all run here.  n = 2^17 is the smallest statement that
    takes the table folds; degenerate rows included.  Every switch that selects another implementation of the same step stays pinned to
    the oracle: RIPP_NO_ENDO is the former plain double-and-add build."""
    n = 1 << 17
    a, b, r = E.synth_g1(1000, n), E.synth_g2(2000, n), E.synth_fr(0, n)
    h = n // 2
    a[h + 3] = 0; b[h + 5] = 0; b[7] = 0; r[h + 9] = 0
    a[h + 11] = a[11]; b[h + 11] = b[11]; r[h + 11] = r[11]
    value = E.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(value, o.product_of_pairings_with_coeffs(a, b, r))
    rc, eproof, ech = o.sipp_prove(a, b, r, value)
    assert rc == 0
    proof, ch, st = E.SIPP.prove_with_stats(a, b, r, value)
    assert np.array_equal(proof, eproof) and np.array_equal(ch, ech)
    assert E.SIPP.verify(a, b, r, value, proof) and o.sipp_verify(a, b, r, value, proof) == 1
    # RIPP_NO_FQ: the 12 x 32-bit throughput kernels this build ran until build round 4 (the default is now the carry-free forms with u^2 = -5 and the
    # D-type twist's line placement: fq_curve2.hpp FQ2_BETA, fq_miller.hpp, fq_line_products.hpp); RIPP_*_FQ_MIN switch single kernels back
    for env in ({"RIPP_NO_ENDO": "1"}, {"RIPP_NO_FOLD_TABLES": "1"}, {"RIPP_NO_XSCALE": "1"}, {"RIPP_NO_PRECOMPUTE": "1"}, {"RIPP_LOOK_EIGHTHS": "12"}, {"RIPP_NO_MSM_GLV": "1"},
                {"RIPP_NO_FQ": "1"}, {"RIPP_LP_FQ_MIN": "4294967295"}, {"RIPP_ML_FQ_MIN": "4294967295"}, {"RIPP_FQ_MIN": "4096", "RIPP_FQ_MIN_G1": "64"},
                {"RIPP_LOOK_EIGHTHS": "48"}, {"RIPP_LOOK_EIGHTHS": "48", "RIPP_NO_SHARE": "1"}):
        os.environ.update(env)
        try:
            assert np.array_equal(E.SIPP.prove(a, b, r, value), eproof), env
            assert E.SIPP.verify(a, b, r, value, eproof), env
        finally:
            for k in env: del os.environ[k]


@pytest.mark.parametrize("n", [64, 1 << 12, 1 << 16])
def test_msm_glv_gls_split_vs_oracle(E, o, n):
    """GLV (G1) / GLS (G2) scalar splits of the Pippenger MSM on BLS12-377 with adversarial scalars: 0, 1, r - 1, lambda, lambda +- 1, x, x^2, x^3,
    2^128 - 1 and values around the Barrett quotient's correction range; also against the unsplit 253-bit windows (RIPP_NO_MSM_GLV)."""
    x = 0x8508C00000000001; lam = x * x - 1
    a, b, s = o.gen_g1(31, n), o.gen_g2(41, n), o.gen_scalars(9, n)
    special = [0, 1, o.R - 1, lam, lam + 1, lam - 1, x, x * x, x ** 3, 2**128 - 1, 2**128, o.R - lam, (o.R - 1) // 2, lam * lam % o.R, x ** 3 - 1, 2 * lam]
    s[:len(special)] = o.fr_array([v % o.R for v in special])[: min(n, len(special))] if n >= len(special) else s[:len(special)]
    e1, e2 = o.g1_to_affine(o.msm_g1_a(a, s)).reshape(1, 12), o.g2_to_affine(o.msm_g2_a(b, s)).reshape(1, 24)
    for env in ({}, {"RIPP_NO_MSM_GLV": "1"}):
        os.environ.update(env)
        try:
            assert np.array_equal(E.normalize_batch_g1(E.MultiexponentiationInnerProductG1.inner_product(o.to_jac_g1(a), s)), e1), env
            assert np.array_equal(E.normalize_batch_g2(E.MultiexponentiationInnerProductG2.inner_product(o.to_jac_g2(b), s)), e2), env
        finally:
            for k in env: del os.environ[k]
    assert np.array_equal(E.scale_g1_affine(a, s), o.scale_g1_a(a, s))


@pytest.mark.parametrize("n", [2, 8, 256, 1 << 12])
def test_aggregate_proofs_bls12_377_vs_oracle(E, o, n):
    """aggregate_proofs (TIPA with SRS shift + TIPAWithSSM, KZG openings, Fr::from_random_bytes with THIS field's 253-bit mask) on the curve of the
    reference's aggregation bench (benches/benches/groth16_aggregation/bench.rs): every member of the AggregateProof equals the BLS12-377
    oracle's, both verifiers accept it, and both reject it against other public inputs."""
    import helpers as h
    osrs = h.make_srs(n, 0xa1fa + n, 0xbe7a + n, o=o); srs = E.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
    vk, pub, a, b, c = h.fake_groth16(n, 2, seed=n, o=o)
    got, _ = E.aggregate_proofs(srs, a, b, c)
    rc, exp = o.aggregate_proofs(osrs[0], osrs[1], a, b, c); assert rc == 0
    rounds = n.bit_length() - 1
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        assert np.array_equal(got.field(k), exp.field(k)), k
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        assert np.array_equal(getattr(got, k), getattr(exp, k)), k
    n1 = lambda p: E.normalize_batch_g1(np.ascontiguousarray(p).reshape(-1, 18)); n2 = lambda p: E.normalize_batch_g2(np.ascontiguousarray(p).reshape(-1, 36))
    for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
        assert np.array_equal(n1(got.field(k)), o.normalize_g1(np.ascontiguousarray(exp.field(k)).reshape(-1, 18))), k
    for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
        assert np.array_equal(n2(got.field(k)), o.normalize_g2(np.ascontiguousarray(exp.field(k)).reshape(-1, 36))), k
    assert np.array_equal(n1(got.c_com_g1[: 2 * rounds]), o.normalize_g1(exp.c_com_g1[: 2 * rounds]))
    assert o.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub, got) == 1
    vs = srs.get_verifier_key()
    assert E.verify_aggregate_proof(vs, vk, pub, got)
    pub2 = pub.copy(); pub2[0, 0] = pub[1, 0]
    assert o.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub2, got) == 0 and not E.verify_aggregate_proof(vs, vk, pub2, got)
    srs.close()


def test_wire_format_round_trip_bls12_377(E, o):
    """CanonicalSerialize / CanonicalDeserialize images of a TIPA proof and a TIPAWithSSM proof on BLS12-377, in ark-ec's generic SWFlags layout
    (flags in the last byte, compressed = x alone, square roots by Tonelli-Shanks): serialise -> deserialise gives the same members in both
    modes, the deserialised proof still verifies, and a flipped flag / an x that is not on the curve / a point outside the subgroup is rejected."""
    import helpers as h
    n = 8
    osrs = h.make_srs(n, 0xa1fa + n, 0xbe7a + n, o=o); srs = E.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
    vk, pub, a, b, c = h.fake_groth16(n, 2, seed=n, o=o)
    got, _ = E.aggregate_proofs(srs, a, b, c)
    ab, cc = h.aggregate_subproofs(got)
    n1 = lambda p: E.normalize_batch_g1(np.ascontiguousarray(p).reshape(-1, 18)); n2 = lambda p: E.normalize_batch_g2(np.ascontiguousarray(p).reshape(-1, 36))
    for compress in (False, True):
        img = E.ser_tipa_tipp_proof(ab, compress=compress)
        back = E.de_tipa_tipp_proof(img, compress=compress)
        assert np.array_equal(back["steps"], ab["steps"])
        for k in ("base_a", "final_ck_b", "opening_b"): assert np.array_equal(n1(back[k]), n1(ab[k])), k
        for k in ("base_b", "final_ck_a", "opening_a"): assert np.array_equal(n2(back[k]), n2(ab[k])), k
        assert E.ser_tipa_tipp_proof(back, compress=compress) == img
        img2 = E.ser_tipa_ssm_proof(cc, compress=compress)
        back2 = E.de_tipa_ssm_proof(img2, compress=compress)
        assert np.array_equal(back2["com_gt"], cc["com_gt"]) and np.array_equal(back2["base_b"], cc["base_b"])
        assert np.array_equal(n1(back2["com_g1"]), n1(cc["com_g1"])) and np.array_equal(n2(back2["opening_a"]), n2(cc["opening_a"]))
        assert E.ser_tipa_ssm_proof(back2, compress=compress) == img2
        # tampering: flip the y-sign flag of the last G1 member (another valid point: members differ), set both flags (invalid), break x
        bad = bytearray(img); bad[-1] ^= 0x80
        if compress:
            assert not np.array_equal(n1(E.de_tipa_tipp_proof(bytes(bad), compress=True)["opening_b"]), n1(ab["opening_b"]))
        bad = bytearray(img); bad[-1] |= 0xC0
        with pytest.raises(ValueError): E.de_tipa_tipp_proof(bytes(bad), compress=compress)
    # the last member of the compressed TIPP image is opening_b (G1, 48 bytes): an x with no point on the curve, and a point of the curve that is
    # not in the prime-order subgroup (x = 0: y^2 = 1, the point (0, 1) has order 3 on y^2 = x^3 + 1 -- G1's cofactor is divisible by 3)
    img = bytearray(E.ser_tipa_tipp_proof(ab, compress=True))
    x = 5
    while pow((x ** 3 + 1) % o.P, (o.P - 1) // 2, o.P) == 1: x += 1
    bad = bytearray(img); bad[-48:] = x.to_bytes(48, "little")
    with pytest.raises(ValueError): E.de_tipa_tipp_proof(bytes(bad), compress=True)
    bad = bytearray(img); bad[-48:] = (0).to_bytes(48, "little")
    with pytest.raises(ValueError): E.de_tipa_tipp_proof(bytes(bad), compress=True)
    srs.close()
