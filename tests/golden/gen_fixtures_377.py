#!/usr/bin/env python3
"""Generate tests/golden/bls12_377_vectors.json from the independent big-integer model tests/model/bls377_model.py: the known answers
that pin the BLS12-377 build of the oracle (and through it the BLS12-377 build of the HIP engine).  DATA only; regenerate with
    python tests/golden/gen_fixtures_377.py
The reference's own SIPP test runs on this curve (sipp/src/lib.rs:229-254) but holds no byte-level answer: parity stays UNPINNED by the
reference, exactly as for BLS12-381."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "model"))
import bls377_model as m  # noqa: E402


def h(x): return hex(x)
def pg1(p): return None if p is None else [h(p[0]), h(p[1])]
def pg2(q): return None if q is None else [[h(q[0][0]), h(q[0][1])], [h(q[1][0]), h(q[1][1])]]


def main():
    fx = {}
    neg1 = (m.G1[0], (-m.G1[1]) % m.P); neg2 = (m.G2[0], m.f2neg(m.G2[1]))
    fx["generators"] = {"g1": pg1(m.G1), "g2": pg2(m.G2), "ser_g1": m.ser_g1(m.G1).hex(), "ser_g2": m.ser_g2(m.G2).hex(),
                        "ser_g1_neg": m.ser_g1(neg1).hex(), "ser_g2_neg": m.ser_g2(neg2).hex(),
                        "ser_g1_inf": m.ser_g1(None).hex(), "ser_g2_inf": m.ser_g2(None).hex(),
                        "ser_g1_compressed": m.ser_g1_compressed(m.G1).hex(), "ser_g2_compressed": m.ser_g2_compressed(m.G2).hex(),
                        "ser_g1_neg_compressed": m.ser_g1_compressed(neg1).hex(), "ser_g2_neg_compressed": m.ser_g2_compressed(neg2).hex(),
                        "ser_g1_inf_compressed": m.ser_g1_compressed(None).hex(), "ser_g2_inf_compressed": m.ser_g2_compressed(None).hex()}
    e = m.pairing(m.G1, m.G2)
    fx["pairing_generators"] = {"gt": m.ser_gt(e).hex()}
    a, b = 0x1234567, 0xABCDEF987
    fx["bilinearity"] = {"a": h(a), "b": h(b), "gt": m.ser_gt(m.pairing(m.g1_mul(a), m.g2_mul(b))).hex(), "gt_pow": m.ser_gt(m.f12pow(e, a * b % m.R)).hex()}
    ks = [3, 5, 7, 11, 13, 17, 19, 23]; ls = [29, 31, 37, 41, 43, 47, 53, 59]
    A = [m.g1_mul(k) for k in ks]; B = [m.g2_mul(l) for l in ls]; A[2] = None; B[5] = None
    fx["product8"] = {"a": [pg1(p) for p in A], "b": [pg2(q) for q in B], "gt": m.ser_gt(m.pairing_product(A, B)).hex()}
    sc = [0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % m.R, m.R - 1, 0, 1, 2**200 + 12345, 2**128 - 1, 4, (m.R - 2)]
    b1 = [m.g1_mul(9 + i) for i in range(8)]; b2 = [m.g2_mul(9 + i) for i in range(8)]
    fx["msm8"] = {"scalars": [h(s) for s in sc], "g1_bases": [pg1(p) for p in b1], "g2_bases": [pg2(q) for q in b2],
                  "g1": pg1(m.g1_msm(b1, sc)), "g2": pg2(m.g2_msm(b2, sc))}
    a4 = [m.g1_mul(21 + i) for i in range(4)]; b4 = [m.g2_mul(31 + i) for i in range(4)]; r4 = [5, 2**130 + 7, m.R - 2, 123456789]
    value = m.pairing_product([m.g1_mul(ri, ai) for ai, ri in zip(a4, r4)], b4)
    proof, ch = m.sipp_prove(a4, b4, r4, value)
    seed = m.ser_vec(a4, m.ser_g1) + m.ser_vec(b4, m.ser_g2) + m.ser_vec(r4, m.ser_fr) + m.ser_gt(value)
    fx["sipp4"] = {"a": [pg1(p) for p in a4], "b": [pg2(q) for q in b4], "r": [h(x) for x in r4], "value": m.ser_gt(value).hex(),
                   "seed_digest": hashlib.blake2s(seed).hexdigest(),
                   "proof": [[m.ser_gt(zl).hex(), m.ser_gt(zr).hex()] for zl, zr in proof], "challenges": [h(c) for c in ch]}
    with open(os.path.join(HERE, "bls12_377_vectors.json"), "w") as f:
        json.dump(fx, f, indent=1)
    print("wrote", os.path.join(HERE, "bls12_377_vectors.json"))


if __name__ == "__main__":
    main()
