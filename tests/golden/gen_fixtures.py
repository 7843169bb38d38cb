#!/usr/bin/env python3
"""Generate tests/golden/*.json from the independent big-integer model (tests/model/bls381_model.py).

The reference (arkworks-rs/ripp) can be neither built nor imported here and its tests hold no golden vectors
(SURVEY.md section 8c), so these fixtures are what pins the C oracle and -- through it -- the HIP engine.
They are DATA (inputs + expected outputs); regenerate with:  python tests/golden/gen_fixtures.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "model"))
import bls381_model as m  # noqa: E402


def h(x): return hex(x)
def pg1(p): return None if p is None else [h(p[0]), h(p[1])]
def pg2(q): return None if q is None else [[h(q[0][0]), h(q[0][1])], [h(q[1][0]), h(q[1][1])]]


def main():
    fx = {}
    fx["generators"] = {"g1": pg1(m.G1), "g2": pg2(m.G2), "ser_g1": m.ser_g1(m.G1).hex(), "ser_g2": m.ser_g2(m.G2).hex(),
                        "ser_g1_inf": m.ser_g1(None).hex(), "ser_g2_inf": m.ser_g2(None).hex()}
    e = m.pairing(m.G1, m.G2)
    fx["pairing_generators"] = {"gt": m.ser_gt(e).hex()}
    a, b = 0x1234567, 0xABCDEF987
    fx["bilinearity"] = {"a": h(a), "b": h(b), "gt": m.ser_gt(m.pairing(m.g1_mul(a), m.g2_mul(b))).hex(),
                         "gt_pow": m.ser_gt(m.f12pow(e, a * b % m.R)).hex()}
    # 8-pair product with an infinity on each side
    ks = [3, 5, 7, 11, 13, 17, 19, 23]; ls = [29, 31, 37, 41, 43, 47, 53, 59]
    A = [m.g1_mul(k) for k in ks]; B = [m.g2_mul(l) for l in ls]
    A[2] = None; B[5] = None
    fx["product8"] = {"a": [pg1(p) for p in A], "b": [pg2(q) for q in B], "gt": m.ser_gt(m.pairing_product(A, B)).hex()}
    # MSM n = 8 in both groups (scalars cover 0, 1, r-1, a 128-bit and full-width values)
    sc = [0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % m.R, m.R - 1, 0, 1, 2**200 + 12345, 2**128 - 1, 4, 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000000 % m.R]
    b1 = [m.g1_mul(9 + i) for i in range(8)]; b2 = [m.g2_mul(9 + i) for i in range(8)]
    fx["msm8"] = {"scalars": [h(s) for s in sc], "g1_bases": [pg1(p) for p in b1], "g2_bases": [pg2(q) for q in b2],
                  "g1": pg1(m.g1_msm(b1, sc)), "g2": pg2(m.g2_msm(b2, sc))}
    # fold: s*hi + lo
    s = 2**127 + 0xDEADBEEF
    fx["fold"] = {"s": h(s), "g1_hi": pg1(m.g1_mul(77)), "g1_lo": pg1(m.g1_mul(5)), "g1": pg1(m.g1_add(m.g1_mul(s, m.g1_mul(77)), m.g1_mul(5))),
                  "g2_hi": pg2(m.g2_mul(77)), "g2_lo": pg2(m.g2_mul(5)), "g2": pg2(m.g2_add(m.g2_mul(s, m.g2_mul(77)), m.g2_mul(5)))}
    # one SIPP proof, n = 4  (sipp/src/lib.rs:42-106 with D = Blake2s)
    a4 = [m.g1_mul(21 + i) for i in range(4)]; b4 = [m.g2_mul(31 + i) for i in range(4)]; r4 = [5, 2**130 + 7, m.R - 2, 123456789]
    value = m.pairing_product([m.g1_mul(ri, ai) for ai, ri in zip(a4, r4)], b4)
    proof, ch = m.sipp_prove(a4, b4, r4, value)
    seed = m.ser_vec(a4, m.ser_g1) + m.ser_vec(b4, m.ser_g2) + m.ser_vec(r4, m.ser_fr) + m.ser_gt(value)
    fx["sipp4"] = {"a": [pg1(p) for p in a4], "b": [pg2(q) for q in b4], "r": [h(x) for x in r4], "value": m.ser_gt(value).hex(),
                   "seed_digest": hashlib.blake2s(seed).hexdigest(),
                   "proof": [[m.ser_gt(zl).hex(), m.ser_gt(zr).hex()] for zl, zr in proof], "challenges": [h(c) for c in ch]}
    # hash / rng primitives
    key = bytes(range(32))
    fx["primitives"] = {"blake2s_abc": hashlib.blake2s(b"abc").hexdigest(), "blake2b_abc": hashlib.blake2b(b"abc").hexdigest(),
                        "blake2s_1000x": hashlib.blake2s(b"x" * 1000).hexdigest(),
                        "chacha20_key": key.hex(), "chacha20_block0": m.chacha20_block(key, 0).hex(), "chacha20_block5": m.chacha20_block(key, 5).hex()}
    rng = m.FiatShamirRng(b"falafel")      # the seed string of the reference's own SIPP test (sipp/src/lib.rs:234)
    fx["fsrng"] = {"seed": "falafel", "u128_0": h(rng.next_u128()), "absorb": "00ff", }
    rng.absorb(bytes.fromhex("00ff")); fx["fsrng"]["u128_after_absorb"] = h(rng.next_u128())
    # one TIPA proof, n = 4, with an SRS shift (ip_proofs/src/tipa/mod.rs:176-231; TIPP instantiation, D = Blake2b): produced AND
    # verified by the big-integer model (tests/model/tipa_model.py); the wire images come from oracle/wire_format.py (also integers)
    import tipa_model as T
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import wire_format as W
    n, alpha, beta, rs = 4, 0xA1FA0007, 0xBE7A0009, 0x1D2C3B4A59687766554433221100FFEEDDCCBBAA998877
    gap = [m.g1_mul(pow(alpha, i, m.R)) for i in range(2 * n - 1)]; hbp = [m.g2_mul(pow(beta, i, m.R)) for i in range(2 * n - 1)]
    g_beta, h_alpha = m.g1_mul(beta), m.g2_mul(alpha)
    ck_1 = [m.g2_mul(pow(rs, -i, m.R), hbp[2 * i]) for i in range(n)]; ck_2 = [gap[2 * i] for i in range(n)]     # ck_1 shifted by r^-i
    m_a = [m.g1_mul(101 + 7 * i) for i in range(n)]; m_b = [m.g2_mul(211 + 5 * i) for i in range(n)]
    steps, tr, ba, bb, ka, kb, oa, ob, kc = T.prove_tipa_tipp(gap, hbp, m_a, m_b, ck_1, ck_2, rs)
    com = [m.pairing_product(m_a, ck_1), m.pairing_product(ck_2, m_b), m.pairing_product(m_a, m_b)]
    assert T.verify_tipa_tipp((gap[0], hbp[0], g_beta, h_alpha), com, steps, ba, bb, ka, kb, oa, ob, rs)
    tower = lambda f: [c for pair in m.f12_to_tower(f) for c in pair]
    fx["tipa4"] = {"alpha": h(alpha), "beta": h(beta), "r_shift": h(rs), "m_a": [pg1(p) for p in m_a], "m_b": [pg2(q) for q in m_b],
                   "com": [m.ser_gt(x).hex() for x in com], "transcript": [h(c) for c in tr], "kzg_challenge": h(kc),
                   "proof_uncompressed": W.tipa_tipp_proof([[tower(x) for x in s] for s in steps], ba, bb, ka, kb, oa, ob, False).hex(),
                   "proof_compressed": W.tipa_tipp_proof([[tower(x) for x in s] for s in steps], ba, bb, ka, kb, oa, ob, True).hex()}
    # aggregate_proofs on n = 4 synthetic (A, B, C) triples (applications/groth16_aggregation.rs:77-160): produced by the model, both
    # sub-proofs verified by the model's verifiers
    A4 = [m.g1_mul(1001 + 3 * i) for i in range(n)]; B4 = [m.g2_mul(2003 + 11 * i) for i in range(n)]; C4 = [m.g1_mul(3001 + 13 * i) for i in range(n)]
    ag = T.aggregate_proofs(gap, hbp, A4, B4, C4)
    vs = (gap[0], hbp[0], g_beta, h_alpha)
    st2, tr2, ba2, bb2, ka2, kb2, oa2, ob2, kc2 = ag["tipp"]
    assert T.verify_tipa_tipp(vs, [ag["com_a"], ag["com_b"], ag["ip_ab"]], st2, ba2, bb2, ka2, kb2, oa2, ob2, ag["r"])
    gts, g1s, tr3, ba3, bb3, ka3, oa3, kc3 = ag["ssm"]
    assert T.verify_tipa_ssm(vs, ag["com_c"], ag["agg_c"], ag["r"], gts, g1s, ba3, ka3, oa3)
    fx["aggregate4"] = {"alpha": h(alpha), "beta": h(beta), "a": [pg1(p) for p in A4], "b": [pg2(q) for q in B4], "c": [pg1(p) for p in C4],
                        "r": h(ag["r"]), "com_a": m.ser_gt(ag["com_a"]).hex(), "com_b": m.ser_gt(ag["com_b"]).hex(), "com_c": m.ser_gt(ag["com_c"]).hex(),
                        "ip_ab": m.ser_gt(ag["ip_ab"]).hex(), "agg_c": m.ser_g1(ag["agg_c"]).hex(),
                        "tipa_proof_ab": W.tipa_tipp_proof([[tower(x) for x in s_] for s_ in st2], ba2, bb2, ka2, kb2, oa2, ob2, False).hex(),
                        "tipa_proof_c": W.tipa_ssm_proof([[tower(x) for x in g_] for g_ in gts], g1s, ba3, bb3, ka3, oa3, False).hex()}
    with open(os.path.join(HERE, "bls12_381_vectors.json"), "w") as f:
        json.dump(fx, f, indent=1)
    print("wrote", os.path.join(HERE, "bls12_381_vectors.json"))


if __name__ == "__main__":
    main()
