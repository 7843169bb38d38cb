"""Fixture decoding shared by CPU and GPU tests."""
import numpy as np
import orclib as o


def g1pt(v): return None if v is None else (int(v[0], 16), int(v[1], 16))
def g2pt(v): return None if v is None else ((int(v[0][0], 16), int(v[0][1], 16)), (int(v[1][0], 16), int(v[1][1], 16)))
def g1arr(vs): return o.g1_array([g1pt(v) for v in vs])
def g2arr(vs): return o.g2_array([g2pt(v) for v in vs])
def frarr(vs): return o.fr_array([int(v, 16) for v in vs])


def gt_from_bytes(hexstr):
    """576-byte serialize_uncompressed image -> (72,) Montgomery limbs."""
    b = bytes.fromhex(hexstr)
    out = np.zeros(72, dtype=np.uint64)
    for i in range(12):
        out[6 * i:6 * i + 6] = o.fp_to_limbs(int.from_bytes(b[48 * i:48 * i + 48], "little"))
    return out


# ---------------------------------------------------------------- TIPA SRS and synthetic Groth16 instances (oracle-side, test infrastructure)
def make_srs(n, alpha, beta, o=o):
    """SRS of TIPA::setup (tipa/mod.rs:150-165) for fixed trapdoors: (g_alpha_powers[2n-1], h_beta_powers[2n-1], g_beta, h_alpha), all Jacobian."""
    fa, fb = o.fr_array([alpha]), o.fr_array([beta])
    gap = o.srs_powers_g1(fa[0], 2 * n - 1); hbp = o.srs_powers_g2(fb[0], 2 * n - 1)
    g_beta = o.to_jac_g1(o.g1_mul_a(o.g1_generator(), fb[0]))[0]; h_alpha = o.to_jac_g2(o.g2_mul_a(o.g2_generator(), fa[0]))[0]
    return gap, hbp, g_beta, h_alpha


def verifier_srs(srs):
    gap, hbp, g_beta, h_alpha = srs
    return gap[0].copy(), hbp[0].copy(), g_beta, h_alpha


def commitment_keys(srs):
    """SRS::get_commitment_keys (tipa/mod.rs:114-118): even powers."""
    return np.ascontiguousarray(srs[1][::2]), np.ascontiguousarray(srs[0][::2])


def fake_groth16(n, m, seed, o=o):
    """n Groth16 (A, B, C) triples over m public inputs that SATISFY e(A,B) = e(alpha,beta) e(sum x_k abc_k, gamma) e(C, delta) for a
    verifying key with known discrete logs (no circuit needed).  Returns vk = (alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1[m+1]),
    public_inputs (n, m, 4), a (n,12), b (n,24), c (n,12)."""
    import random
    rng = random.Random(seed); R = o.R
    rnd = lambda: rng.randrange(1, R)
    al, be, ga, de = rnd(), rnd(), rnd(), rnd(); abc = [rnd() for _ in range(m + 1)]
    g1, g2 = o.g1_generator(), o.g2_generator()
    mul1 = lambda k: o.g1_mul_a(g1, o.fr_array([k])[0]); mul2 = lambda k: o.g2_mul_a(g2, o.fr_array([k])[0])
    vk = (mul1(al), mul2(be), mul2(ga), mul2(de), np.stack([mul1(k) for k in abc]))
    a = np.zeros((n, 12), dtype=np.uint64); b = np.zeros((n, 24), dtype=np.uint64); c = np.zeros((n, 12), dtype=np.uint64); pub = np.zeros((n, m, 4), dtype=np.uint64)
    for i in range(n):
        ai, bi = rnd(), rnd(); xs = [rnd() for _ in range(m)]
        s = (abc[0] + sum(x * k for x, k in zip(xs, abc[1:]))) % R
        ci = (ai * bi - al * be - ga * s) * pow(de, -1, R) % R
        a[i], b[i], c[i] = mul1(ai), mul2(bi), mul1(ci); pub[i] = o.fr_array(xs)
    return vk, pub, a, b, c


def tipa4_instance(v):
    """Inputs of the golden `tipa4` vector (tests/golden/gen_fixtures.py) in the flat limb layouts: (srs, m_a, m_b, ck_a shifted by r^-i, ck_b, r_shift)."""
    n = 4; alpha, beta, rs = int(v["alpha"], 16), int(v["beta"], 16), int(v["r_shift"], 16)
    srs = make_srs(n, alpha, beta); ck_a, ck_b = commitment_keys(srs)
    ck_a = np.stack([o.to_jac_g2(o.g2_mul_a(o.g2_to_affine(ck_a[i]), o.fr_array([pow(rs, -i, o.R)])[0]))[0] for i in range(n)])
    m_a, m_b = o.to_jac_g1(g1arr(v["m_a"])), o.to_jac_g2(g2arr(v["m_b"]))
    return srs, m_a, m_b, ck_a, ck_b, o.fr_array([rs])[0]


def aggregate_subproofs(pf):
    """AggregateProof (ripp_amd._lib) -> the dict forms the wire-format serialisers take: (tipa_proof_ab, tipa_proof_c)."""
    f = lambda k: np.ascontiguousarray(pf.field(k))
    ab = dict(steps=pf.ab_com_steps, tr=pf.ab_transcript, base_a=f("ab_base_a"), base_b=f("ab_base_b"), final_ck_a=f("ab_final_ck_a"), final_ck_b=f("ab_final_ck_b"),
              opening_a=f("ab_opening_a"), opening_b=f("ab_opening_b"))
    c = dict(com_gt=pf.c_com_gt, com_g1=pf.c_com_g1, tr=pf.c_transcript, base_a=f("c_base_a"), base_b=f("c_base_b"), final_ck_a=f("c_final_ck_a"), opening_a=f("c_opening_a"))
    return ab, c


def check_aggregate_golden(v, pf, ser_gt, ser_g1_of_jac, ser_tipp, ser_ssm):
    """compare an AggregateProof with the golden `aggregate4` vector (model-made): scalar r, the four GT values, agg_c, both sub-proof images"""
    assert hex(o.limbs_to_fr(pf.field("r"))) == v["r"]
    for k in ("com_a", "com_b", "com_c", "ip_ab"):
        assert ser_gt(np.ascontiguousarray(pf.field(k))).hex() == v[k], k
    assert ser_g1_of_jac(np.ascontiguousarray(pf.field("agg_c"))).hex() == v["agg_c"]
    ab, c = aggregate_subproofs(pf)
    assert ser_tipp(ab, compress=False).hex() == v["tipa_proof_ab"]
    assert ser_ssm(c, compress=False).hex() == v["tipa_proof_c"]
