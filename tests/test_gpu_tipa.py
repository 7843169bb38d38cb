"""GPU parity tests (-m gpu) of the callers directly above the hot path: TIPA (GIPA + KZG key
openings, ip_proofs/src/tipa/mod.rs), TIPAWithSSM (tipa/structured_scalar_message.rs) and aggregate_proofs
(applications/groth16_aggregation.rs), through the C ABI against the CPU oracle on identical inputs.  GT and Fr members must be
bit-equal; projective members are compared as group elements (after normalisation).  The oracle's verifiers must accept the
GPU-made proofs."""
import numpy as np
import pytest

import helpers as h
from helpers import gt_from_bytes

pytestmark = pytest.mark.gpu


def same_g1(engine, orc, x, y):
    return np.array_equal(engine.normalize_batch_g1(np.asarray(x).reshape(-1, 18)), orc.normalize_g1(np.ascontiguousarray(np.asarray(y).reshape(-1, 18))))


def same_g2(engine, orc, x, y):
    return np.array_equal(engine.normalize_batch_g2(np.asarray(x).reshape(-1, 36)), orc.normalize_g2(np.ascontiguousarray(np.asarray(y).reshape(-1, 36))))


def test_srs_powers_and_commitment_keys(engine, orc):
    n = 16; alpha, beta = orc.fr_array([0x1234567]), orc.fr_array([0x89abcdef])
    srs = engine.SRS.from_trapdoors(alpha[0], beta[0], n)
    osrs = h.make_srs(n, 0x1234567, 0x89abcdef)
    assert same_g1(engine, orc, srs.g_alpha_powers, osrs[0]) and same_g2(engine, orc, srs.h_beta_powers, osrs[1])
    assert same_g1(engine, orc, srs.g_beta, osrs[2]) and same_g2(engine, orc, srs.h_alpha, osrs[3])
    ck_1, ck_2 = srs.get_commitment_keys(); ock_1, ock_2 = h.commitment_keys(osrs)
    assert same_g2(engine, orc, ck_1, ock_1) and same_g1(engine, orc, ck_2, ock_2)


@pytest.mark.parametrize("n,shift", [(2, 1), (8, 1), (8, 0x1d2c3b4a59687766554433221100ffee), (64, 3), (1 << 10, 1)])
def test_tipa_tipp_prove_vs_oracle(engine, orc, n, shift):
    osrs = h.make_srs(n, 0xa11ce + n, 0xb0b + n)
    srs = engine.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
    ck_a, ck_b = h.commitment_keys(osrs)
    if shift != 1:      # LMC key pre-shifted by r^-i as aggregate_proofs does (groth16_aggregation.rs:127-131)
        ck_a = np.stack([orc.to_jac_g2(orc.g2_mul_a(orc.g2_to_affine(ck_a[i]), orc.fr_array([pow(shift, -i, orc.R)])[0]))[0] for i in range(n)])
    m_a, m_b = orc.blind_g1(orc.gen_g1(7, n), 1), orc.blind_g2(orc.gen_g2(9, n), 2)
    r_shift = orc.fr_array([shift])[0]
    got = engine.TIPA_TIPP.prove_with_srs_shift(srs, (m_a, m_b), (ck_a, ck_b), r_shift)
    rc, exp = orc.tipa_tipp_prove(osrs[0], osrs[1], m_a, m_b, ck_a, ck_b, r_shift); assert rc == 0
    assert np.array_equal(got["steps"], exp["steps"]) and np.array_equal(got["tr"], exp["tr"]) and np.array_equal(got["kzg_c"], exp["kzg_c"])
    for k in ("base_a", "final_ck_b", "opening_b"):
        assert same_g1(engine, orc, got[k], exp[k]), k
    for k in ("base_b", "final_ck_a", "opening_a"):
        assert same_g2(engine, orc, got[k], exp[k]), k
    com = [engine.AFGHOCommitmentG1.commit(ck_a, m_a), engine.AFGHOCommitmentG2.commit(ck_b, m_b), engine.PairingInnerProduct.inner_product(m_a, m_b)]
    assert orc.tipa_tipp_verify(*h.verifier_srs(osrs), com, got, r_shift) == 1
    # the product's own verifier (tipa/mod.rs:242-301): accepts, and rejects what the oracle's verifier rejects
    vk = srs.get_verifier_key()
    assert engine.TIPA_TIPP.verify_with_srs_shift(vk, com, got, r_shift)
    for field, repl in (("opening_a", "final_ck_a"), ("final_ck_b", "base_a"), ("base_b", "opening_a")):
        bad = dict(got); bad[field] = got[repl]
        assert not engine.TIPA_TIPP.verify_with_srs_shift(vk, com, bad, r_shift) and orc.tipa_tipp_verify(*h.verifier_srs(osrs), com, bad, r_shift) == 0
    assert not engine.TIPA_TIPP.verify_with_srs_shift(vk, [com[0], com[1], com[0]], got, r_shift)
    assert not engine.TIPA_TIPP.verify_with_srs_shift(vk, com, got, orc.fr_array([shift + 1])[0])
    bad = dict(got); bad["steps"] = got["steps"].copy(); bad["steps"][0] = got["steps"][3]
    assert not engine.TIPA_TIPP.verify_with_srs_shift(vk, com, bad, r_shift)
    # an Fp12 value that is not a GT element (a Miller value before its final exponentiation) must be rejected, not exponentiated
    bad = dict(got); bad["steps"] = got["steps"].copy(); bad["steps"][1] = orc.miller_product_a(orc.gen_g1(3, 1), orc.gen_g2(4, 1))
    assert not engine.TIPA_TIPP.verify_with_srs_shift(vk, com, bad, r_shift)
    srs.close()


def test_golden_tipa_vector(engine, orc, vectors):
    """The engine's TIPA proof of the golden `tipa4` instance (made and verified by the big-integer model) serialises to the committed
    wire images byte for byte, compressed and uncompressed."""
    v = vectors["tipa4"]
    osrs, m_a, m_b, ck_a, ck_b, r_shift = h.tipa4_instance(v)
    srs = engine.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
    got = engine.TIPA_TIPP.prove_with_srs_shift(srs, (m_a, m_b), (ck_a, ck_b), r_shift)
    assert [hex(orc.limbs_to_fr(t)) for t in got["tr"]] == v["transcript"] and hex(orc.limbs_to_fr(got["kzg_c"])) == v["kzg_challenge"]
    assert engine.ser_tipa_tipp_proof(got, compress=False).hex() == v["proof_uncompressed"]
    assert engine.ser_tipa_tipp_proof(got, compress=True).hex() == v["proof_compressed"]
    com = [gt_from_bytes(x) for x in v["com"]]
    assert engine.TIPA_TIPP.verify_with_srs_shift(srs.get_verifier_key(), com, got, r_shift)
    srs.close()


def test_golden_aggregate_vector(engine, orc, vectors):
    """The engine's aggregate_proofs on the golden `aggregate4` instance (made by the big-integer model) reproduces every member."""
    v = vectors["aggregate4"]
    osrs = h.make_srs(4, int(v["alpha"], 16), int(v["beta"], 16))
    srs = engine.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
    pf, _ = engine.aggregate_proofs(srs, h.g1arr(v["a"]), h.g2arr(v["b"]), h.g1arr(v["c"]))
    h.check_aggregate_golden(v, pf, engine.ser_gt, lambda j: engine.ser_g1(engine.normalize_batch_g1(j.reshape(1, 18))[0]), engine.ser_tipa_tipp_proof, engine.ser_tipa_ssm_proof)
    srs.close()


def test_tipa_rejects_mismatched_srs(engine, orc):
    osrs = h.make_srs(8, 5, 6); srs = engine.SRS(osrs[0], osrs[1])
    m_a, m_b = orc.blind_g1(orc.gen_g1(7, 4), 1), orc.blind_g2(orc.gen_g2(9, 4), 2)
    with pytest.raises(ValueError):
        engine.TIPA_TIPP.prove(srs, (m_a, m_b), (m_b, m_a))
    with pytest.raises(AssertionError):
        engine.TIPA_TIPP.prove(srs, (m_a[:1], m_b[:1]), (m_b[:1], m_a[:1]))       # n = 1: no transcript to open (tipa/mod.rs:200-202)


@pytest.mark.parametrize("n", [2, 8, 64, 1 << 10])
def test_tipa_ssm_prove_vs_oracle(engine, orc, n):
    osrs = h.make_srs(n, 0x5eed + n, 0xfeed + n); srs = engine.SRS(osrs[0], osrs[1])
    ck_a, _ = h.commitment_keys(osrs)
    m_a = orc.blind_g1(orc.gen_g1(13, n), 3)
    b = 0x5eed5eed5eed5eed5eed5eed5eed5eed5eed % orc.R
    m_b = orc.fr_array([pow(b, i, orc.R) for i in range(n)])
    got = engine.TIPAWithSSM.prove_with_structured_scalar_message(srs, (m_a, m_b), (ck_a,))
    rc, exp = orc.tipa_ssm_prove(osrs[1], m_a, m_b, ck_a); assert rc == 0
    assert np.array_equal(got["com_gt"], exp["com_gt"]) and np.array_equal(got["tr"], exp["tr"]) and np.array_equal(got["kzg_c"], exp["kzg_c"])
    assert np.array_equal(got["base_b"], exp["base_b"])
    assert same_g1(engine, orc, got["com_g1"], exp["com_g1"]) and same_g1(engine, orc, got["base_a"], exp["base_a"])
    assert same_g2(engine, orc, got["final_ck_a"], exp["final_ck_a"]) and same_g2(engine, orc, got["opening_a"], exp["opening_a"])
    g, hh, g_beta, _ = h.verifier_srs(osrs)
    com_a = engine.AFGHOCommitmentG1.commit(ck_a, m_a); com_t = engine.MultiexponentiationInnerProductG1.inner_product(m_a, m_b)
    assert orc.tipa_ssm_verify(g, hh, g_beta, com_a, com_t, orc.fr_array([b])[0], got) == 1
    vk = {"g": g, "h": hh, "g_beta": g_beta, "h_alpha": osrs[3]}
    assert engine.TIPAWithSSM.verify_with_structured_scalar_message(vk, (com_a, com_t), orc.fr_array([b])[0], got)
    assert not engine.TIPAWithSSM.verify_with_structured_scalar_message(vk, (com_a, com_t), orc.fr_array([b + 1])[0], got)
    assert not engine.TIPAWithSSM.verify_with_structured_scalar_message(vk, (com_a, got["base_a"]), orc.fr_array([b])[0], got)
    bad = dict(got); bad["opening_a"] = got["final_ck_a"]
    assert not engine.TIPAWithSSM.verify_with_structured_scalar_message(vk, (com_a, com_t), orc.fr_array([b])[0], bad)
    srs.close()


@pytest.mark.parametrize("n", [2, 8, 256])
def test_aggregate_proofs_vs_oracle(engine, orc, n):
    """aggregate_proofs on n synthetic-but-VALID Groth16 proofs: every member of the AggregateProof equals the oracle's and the
    oracle's verify_aggregate_proof (groth16_aggregation.rs:162-231) accepts the GPU-made proof."""
    m = 2
    osrs = h.make_srs(n, 0xa1fa + n, 0xbe7a + n); srs = engine.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
    vk, pub, a, b, c = h.fake_groth16(n, m, seed=n)
    got, stats = engine.aggregate_proofs(srs, a, b, c)
    rc, exp = orc.aggregate_proofs(osrs[0], osrs[1], a, b, c); assert rc == 0
    rounds = n.bit_length() - 1
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        assert np.array_equal(got.field(k), exp.field(k)), k
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        assert np.array_equal(getattr(got, k), getattr(exp, k)), k
    for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
        assert same_g1(engine, orc, got.field(k), exp.field(k)), k
    for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
        assert same_g2(engine, orc, got.field(k), exp.field(k)), k
    assert same_g1(engine, orc, got.c_com_g1[: 2 * rounds], exp.c_com_g1[: 2 * rounds])
    assert orc.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub, got) == 1
    pub2 = pub.copy(); pub2[0, 0] = pub[1, 0]
    assert orc.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub2, got) == 0
    # the product's verify_aggregate_proof (groth16_aggregation.rs:162-231)
    vs = srs.get_verifier_key()
    assert engine.verify_aggregate_proof(vs, vk, pub, got)
    assert not engine.verify_aggregate_proof(vs, vk, pub2, got)
    c2 = c.copy(); c2[0] = c[1]                       # one invalid Groth16 proof among the n
    bad, _ = engine.aggregate_proofs(srs, a, b, c2)
    assert not engine.verify_aggregate_proof(vs, vk, pub, bad) and orc.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub, bad) == 0
    # by default the TIPP and TIPAWithSSM sub-proofs run side by side (second host thread, the engine's auxiliary streams and scratch);
    # RIPP_AGG_SEQUENTIAL=1 runs them one after the other on the main engine: the same AggregateProof, member for member
    import os
    os.environ["RIPP_AGG_SEQUENTIAL"] = "1"
    try:
        seq, _ = engine.aggregate_proofs(srs, a, b, c)
    finally:
        del os.environ["RIPP_AGG_SEQUENTIAL"]
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        assert np.array_equal(seq.field(k), got.field(k)), k
    for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):          # projective members: the same POINT (an MSM's bucket order, hence its Z, varies run to run)
        assert same_g1(engine, orc, seq.field(k), got.field(k)), k
    for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
        assert same_g2(engine, orc, seq.field(k), got.field(k)), k
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        assert np.array_equal(getattr(seq, k), getattr(got, k)), k
    srs.close()


def test_aggregate_proofs_degenerate_inputs_vs_oracle(engine, orc):
    """identities among the (A, B, C) members and repeated triples: every member of the aggregate still equals the oracle's"""
    n = 16
    osrs = h.make_srs(n, 0x51, 0x52); srs = engine.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
    a, b, c = orc.gen_g1(3, n), orc.gen_g2(5, n), orc.gen_g1(7, n)
    a[2] = 0; b[3] = 0; c[1] = 0; c[9] = 0; a[12] = a[4]; b[12] = b[4]; c[12] = c[4]; a[15] = 0; b[15] = 0; c[15] = 0
    got, _ = engine.aggregate_proofs(srs, a, b, c)
    rc, exp = orc.aggregate_proofs(osrs[0], osrs[1], a, b, c); assert rc == 0
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        assert np.array_equal(got.field(k), exp.field(k)), k
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        assert np.array_equal(getattr(got, k), getattr(exp, k)), k
    for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
        assert same_g1(engine, orc, got.field(k), exp.field(k)), k
    for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
        assert same_g2(engine, orc, got.field(k), exp.field(k)), k
    assert same_g1(engine, orc, got.c_com_g1, exp.c_com_g1)
    srs.close()


def test_aggregate_proofs_config5_size(engine, orc):
    """n = 2^14 synthetic (A, B, C) triples (random group elements, the prover never checks Groth16
    validity).  EVERY member of the aggregate equals the oracle's `aggregate_proofs` on the same SRS and triples (a few seconds of
    oracle time on the GPU box's host cores), and the proof passes the oracle's and the engine's TIPA / SSM verifiers."""
    n = 1 << 14
    alpha, beta = orc.fr_array([0xa1fa0001]), orc.fr_array([0xbe7a0001])
    srs = engine.SRS.from_trapdoors(alpha[0], beta[0], n)
    a, b, c = engine.synth_g1(101, n), engine.synth_g2(202, n), engine.synth_g1(303, n)
    got, stats = engine.aggregate_proofs(srs, a, b, c)
    rc, exp = orc.aggregate_proofs(srs.g_alpha_powers, srs.h_beta_powers, a, b, c)
    assert rc == 0
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        assert np.array_equal(got.field(k), exp.field(k)), k
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        assert np.array_equal(getattr(got, k), getattr(exp, k)), k
    for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
        assert same_g1(engine, orc, got.field(k), exp.field(k)), k
    for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
        assert same_g2(engine, orc, got.field(k), exp.field(k)), k
    assert same_g1(engine, orc, got.c_com_g1, exp.c_com_g1)
    vs = srs.get_verifier_key(); g, hh, g_beta, h_alpha = vs["g"], vs["h"], vs["g_beta"], vs["h_alpha"]
    r = got.field("r")
    tipp = dict(steps=got.ab_com_steps, base_a=got.field("ab_base_a"), base_b=got.field("ab_base_b"), final_ck_a=got.field("ab_final_ck_a"),
                final_ck_b=got.field("ab_final_ck_b"), opening_a=got.field("ab_opening_a"), opening_b=got.field("ab_opening_b"))
    tipp = {k: np.ascontiguousarray(v) for k, v in tipp.items()}
    assert orc.tipa_tipp_verify(g, hh, g_beta, h_alpha, [got.field("com_a"), got.field("com_b"), got.field("ip_ab")], tipp, np.ascontiguousarray(r)) == 1
    ssm = dict(com_gt=got.c_com_gt, com_g1=got.c_com_g1, base_a=np.ascontiguousarray(got.field("c_base_a")), final_ck_a=np.ascontiguousarray(got.field("c_final_ck_a")),
               opening_a=np.ascontiguousarray(got.field("c_opening_a")))
    assert orc.tipa_ssm_verify(g, hh, g_beta, np.ascontiguousarray(got.field("com_c")), np.ascontiguousarray(got.field("agg_c")), np.ascontiguousarray(r), ssm) == 1
    tipp["tr"] = got.ab_transcript
    assert engine.TIPA_TIPP.verify_with_srs_shift(vs, [got.field("com_a"), got.field("com_b"), got.field("ip_ab")], tipp, r)
    assert engine.TIPAWithSSM.verify_with_structured_scalar_message(vs, (got.field("com_c"), got.field("agg_c")), r, ssm)
    print("aggregate_proofs n=2^14:", {k: round(v, 1) for k, v in stats.items() if k.endswith("_ms") and v})
    srs.close()


def test_aggregate_proofs_table_fold_size(engine, orc):
    """n = 2^16: the first TIPP round folds 32768 G2 elements per vector, the size from which the folds build in-round odd-multiple
    tables (width-4 wNAF; engine.hip::fold_g2_table).  The aggregate must be byte-identical with and without the tables
    (RIPP_NO_FOLD_TABLES=1, read per call) and pass the oracle's TIPA verifier."""
    import os
    n = 1 << 16
    alpha, beta = orc.fr_array([0xa1fa0001]), orc.fr_array([0xbe7a0001])
    srs = engine.SRS.from_trapdoors(alpha[0], beta[0], n)
    a, b, c = engine.synth_g1(101, n), engine.synth_g2(202, n), engine.synth_g1(303, n)
    got, _ = engine.aggregate_proofs(srs, a, b, c)
    os.environ["RIPP_NO_FOLD_TABLES"] = "1"
    try:
        ref, _ = engine.aggregate_proofs(srs, a, b, c)
    finally:
        del os.environ["RIPP_NO_FOLD_TABLES"]
    def canon(x):                                   # group elements travel in Jacobian form; MSM outputs are not canonical there (atomic scatter order)
        x = np.ascontiguousarray(x).reshape(-1, x.shape[-1])
        return engine.normalize_batch_g1(x) if x.shape[1] == 18 else engine.normalize_batch_g2(x) if x.shape[1] == 36 else x
    for name in got.FIXED:
        assert np.array_equal(canon(got.field(name)), canon(ref.field(name))), name
    for name in got.STEPS:
        assert np.array_equal(canon(getattr(got, name)), canon(getattr(ref, name))), name
    vs = srs.get_verifier_key(); g, hh, g_beta, h_alpha = vs["g"], vs["h"], vs["g_beta"], vs["h_alpha"]
    r = got.field("r")
    tipp = dict(steps=got.ab_com_steps, base_a=got.field("ab_base_a"), base_b=got.field("ab_base_b"), final_ck_a=got.field("ab_final_ck_a"),
                final_ck_b=got.field("ab_final_ck_b"), opening_a=got.field("ab_opening_a"), opening_b=got.field("ab_opening_b"))
    tipp = {k: np.ascontiguousarray(v) for k, v in tipp.items()}
    assert orc.tipa_tipp_verify(g, hh, g_beta, h_alpha, [got.field("com_a"), got.field("com_b"), got.field("ip_ab")], tipp, np.ascontiguousarray(r)) == 1
    srs.close()
