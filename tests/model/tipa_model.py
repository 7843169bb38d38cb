"""Independent big-integer VERIFIER for TIPA proofs of the TIPP instantiation, written from the protocol
(ip_proofs/src/gipa.rs:322-363, tipa/mod.rs:242-301, 340-404) on top of tests/model/bls381_model.py: Python integers,
textbook affine group law, final exponentiation by one pow, hashlib's BLAKE2b.  It shares no code with oracle/ or the engine, so a
proof it accepts pins the Fiat-Shamir transcript layout, `Fr::from_random_bytes`, the KZG equations and the transcript order of
both of them.  Test infrastructure only.
"""
import hashlib

import bls381_model as M

R = M.R


def gt_from_tower(t):           # 12 ints in arkworks tower order -> model flat polynomial in w
    c = [(t[2 * i], t[2 * i + 1]) for i in range(6)]
    return [c[0], c[3], c[1], c[4], c[2], c[5]]


def fr_from_random_bytes(d):
    v = int.from_bytes(d[:32], "little") & ((1 << 255) - 1)
    return v if v < R else None


def gipa_challenge(prev, s):     # s: six GT (model flat form) of one round
    nonce = 0
    while True:
        h = nonce.to_bytes(8, "big") + M.ser_fr(prev)
        for k in range(6):
            if k in (2, 5):
                h += (1).to_bytes(8, "little")
            h += M.ser_gt(s[k])
        c128 = int.from_bytes(hashlib.blake2b(h).digest()[:16], "big")
        if c128 % R:
            return pow(c128, -1, R), c128 % R          # (c, c_inv): the reference swaps the names (gipa.rs:252-256)
        nonce += 1


def kzg_challenge(first, ck_a, ck_b):
    nonce = 0
    while True:
        h = nonce.to_bytes(8, "big") + M.ser_fr(first) + M.ser_g2(ck_a) + (M.ser_g1(ck_b) if ck_b is not False else b"")
        c = fr_from_random_bytes(hashlib.blake2b(h).digest())
        if c is not None:
            return c
        nonce += 1


def poly_eval(tr, z, r_shift):
    p, acc = z * z % R * r_shift % R, 1
    for x in tr:
        acc = acc * (1 + x * p) % R; p = p * p % R
    return acc


def verify_tipa_tipp(v_srs, com, steps, base_a, base_b, final_ck_a, final_ck_b, opening_a, opening_b, r_shift):
    """v_srs = (g, h, g_beta, h_alpha) affine; com = 3 GT; steps = per ROUND six GT; everything as model values. -> bool"""
    g, h, g_beta, h_alpha = v_srs
    ca, cb, ct = com
    tr_fwd, prev = [], 0
    for s in steps:
        c, c_inv = gipa_challenge(prev, s)
        ca = M.f12mul(ca, M.f12mul(M.f12pow(s[0], c), M.f12pow(s[3], c_inv)))
        cb = M.f12mul(cb, M.f12mul(M.f12pow(s[1], c), M.f12pow(s[4], c_inv)))
        ct = M.f12mul(ct, M.f12mul(M.f12pow(s[2], c), M.f12pow(s[5], c_inv)))
        tr_fwd.append(c); prev = c
    tr = tr_fwd[::-1]; tri = [pow(x, -1, R) for x in tr]
    c = kzg_challenge(tr[0], final_ck_a, final_ck_b)
    ok = True
    ev = poly_eval(tri, c, pow(r_shift, -1, R))                                     # e(g, ck_a - h*ev) == e(g_beta - g*c, opening_a)
    ok &= M.pairing(g, M.g2_add(final_ck_a, M.ec_neg(M._Fp2, M.g2_mul(ev, h)))) == M.pairing(M.g1_add(g_beta, M.ec_neg(M._Fp, M.g1_mul(c, g))), opening_a)
    ev = poly_eval(tr, c, 1)                                                        # e(ck_b - g*ev, h) == e(opening_b, h_alpha - h*c)
    ok &= M.pairing(M.g1_add(final_ck_b, M.ec_neg(M._Fp, M.g1_mul(ev, g))), h) == M.pairing(opening_b, M.g2_add(h_alpha, M.ec_neg(M._Fp2, M.g2_mul(c, h))))
    ok &= M.pairing(base_a, final_ck_a) == ca and M.pairing(final_ck_b, base_b) == cb and M.pairing(base_a, base_b) == ct
    return bool(ok)


# ------------------------------------------------------------------ prover (for golden vectors; n small)
def _fold(mul, add, hi, lo, s):
    return [add(mul(s, x), y) for x, y in zip(hi, lo)]


def ck_poly_coeffs(tr, r_shift):
    co, p2r = [1], r_shift % R
    for i, x in enumerate(tr):
        for j in range(2 ** i):
            co.append(co[j] * (x * p2r % R) % R)
        p2r = p2r * p2r % R
    out = []
    for k, v in enumerate(co):                         # interleave with zeros (tipa/mod.rs:416-421)
        out.append(v)
        if k + 1 < len(co):
            out.append(0)
    return out


def kzg_quotient(tr, r_shift, c):
    p = ck_poly_coeffs(tr, r_shift)
    q = [0] * len(p); carry = 0
    for k in range(len(p) - 1, 0, -1):                 # (p(X) - p(c)) / (X - c), padded to len(p) coefficients
        carry = (p[k] + c * carry) % R; q[k - 1] = carry
    return q


def prove_tipa_tipp(g_alpha_powers, h_beta_powers, m_a, m_b, ck_a, ck_b, r_shift):
    """TIPA::prove_with_srs_shift (tipa/mod.rs:176-231) for the TIPP instantiation on model values (affine points / None).
    Returns (steps per round [6 GT], transcript per round, base_a, base_b, final_ck_a, final_ck_b, opening_a, opening_b, kzg_c)."""
    steps, tr_fwd, prev = [], [], 0
    while len(m_a) > 1:
        sp = len(m_a) // 2
        ma1, ma2, ka1, ka2 = m_a[sp:], m_a[:sp], ck_a[:sp], ck_a[sp:]
        mb1, mb2, kb1, kb2 = m_b[:sp], m_b[sp:], ck_b[sp:], ck_b[:sp]
        s = [M.pairing_product(ma1, ka1), M.pairing_product(kb1, mb1), M.pairing_product(ma1, mb1),
             M.pairing_product(ma2, ka2), M.pairing_product(kb2, mb2), M.pairing_product(ma2, mb2)]
        c, c_inv = gipa_challenge(prev, s)
        m_a = _fold(M.g1_mul, M.g1_add, ma1, ma2, c); m_b = _fold(M.g2_mul, M.g2_add, mb2, mb1, c_inv)
        ck_a = _fold(M.g2_mul, M.g2_add, ka2, ka1, c_inv); ck_b = _fold(M.g1_mul, M.g1_add, kb1, kb2, c)
        steps.append(s); tr_fwd.append(c); prev = c
    tr = tr_fwd[::-1]; tri = [pow(x, -1, R) for x in tr]
    c = kzg_challenge(tr[0], ck_a[0], ck_b[0])
    opening_a = M.g2_msm(h_beta_powers, kzg_quotient(tri, pow(r_shift, -1, R), c))
    opening_b = M.g1_msm(g_alpha_powers, kzg_quotient(tr, 1, c))
    return steps, tr_fwd, m_a[0], m_b[0], ck_a[0], ck_b[0], opening_a, opening_b, c


# ------------------------------------------------------------------ TIPAWithSSM and aggregate_proofs (model prover / verifier)
def gipa_ssm_challenge(prev, gt, g1):      # per side: GT || Fr::zero() || u64 1 || G1   (structured_scalar_message.rs + gipa.rs:235-258)
    nonce = 0
    while True:
        h = nonce.to_bytes(8, "big") + M.ser_fr(prev)
        for k in range(2):
            h += M.ser_gt(gt[k]) + M.ser_fr(0) + (1).to_bytes(8, "little") + M.ser_g1(g1[k])
        c128 = int.from_bytes(hashlib.blake2b(h).digest()[:16], "big")
        if c128 % R:
            return pow(c128, -1, R), c128 % R
        nonce += 1


def prove_tipa_ssm(h_beta_powers, m_a, m_b, ck_a):
    """TIPAWithSSM::prove_with_structured_scalar_message (structured_scalar_message.rs:211-268); m_b: scalars."""
    gts, g1s, tr_fwd, prev = [], [], [], 0
    while len(m_a) > 1:
        sp = len(m_a) // 2
        ma1, ma2, ka1, ka2, mb1, mb2 = m_a[sp:], m_a[:sp], ck_a[:sp], ck_a[sp:], m_b[:sp], m_b[sp:]
        gt = [M.pairing_product(ma1, ka1), M.pairing_product(ma2, ka2)]; g1 = [M.g1_msm(ma1, mb1), M.g1_msm(ma2, mb2)]
        c, c_inv = gipa_ssm_challenge(prev, gt, g1)
        m_a = _fold(M.g1_mul, M.g1_add, ma1, ma2, c); ck_a = _fold(M.g2_mul, M.g2_add, ka2, ka1, c_inv)
        m_b = [(y * c_inv + x) % R for x, y in zip(mb1, mb2)]
        gts.append(gt); g1s.append(g1); tr_fwd.append(c); prev = c
    tr = tr_fwd[::-1]; tri = [pow(x, -1, R) for x in tr]
    c = kzg_challenge(tr[0], ck_a[0], False)
    return gts, g1s, tr_fwd, m_a[0], m_b[0], ck_a[0], M.g2_msm(h_beta_powers, kzg_quotient(tri, 1, c)), c


def verify_tipa_ssm(v_srs, com_a, com_t, scalar_b, gts, g1s, base_a, final_ck_a, opening_a):
    g, h, g_beta, _ = v_srs
    ca, ct, tr_fwd, prev = com_a, com_t, [], 0
    for gt, g1 in zip(gts, g1s):
        c, c_inv = gipa_ssm_challenge(prev, gt, g1)
        ca = M.f12mul(ca, M.f12mul(M.f12pow(gt[0], c), M.f12pow(gt[1], c_inv)))
        ct = M.g1_add(ct, M.g1_add(M.g1_mul(c, g1[0]), M.g1_mul(c_inv, g1[1])))
        tr_fwd.append(c); prev = c
    tr = tr_fwd[::-1]; tri = [pow(x, -1, R) for x in tr]
    c = kzg_challenge(tr[0], final_ck_a, False)
    ev = poly_eval(tri, c, 1)
    ok = M.pairing(g, M.g2_add(final_ck_a, M.ec_neg(M._Fp2, M.g2_mul(ev, h)))) == M.pairing(M.g1_add(g_beta, M.ec_neg(M._Fp, M.g1_mul(c, g))), opening_a)
    p2b, b_base = scalar_b % R, 1
    for xi in tri:
        b_base = b_base * (1 + xi * p2b) % R; p2b = p2b * p2b % R
    return bool(ok and M.pairing(base_a, final_ck_a) == ca and M.g1_mul(b_base, base_a) == ct)


def aggregation_challenge(com_a, com_b, com_c):
    nonce = 0
    while True:
        r = fr_from_random_bytes(hashlib.blake2b(nonce.to_bytes(8, "big") + M.ser_gt(com_a) + M.ser_gt(com_b) + M.ser_gt(com_c)).digest())
        if r is not None:
            return r
        nonce += 1


def aggregate_proofs(g_alpha_powers, h_beta_powers, a, b, c):
    """aggregate_proofs (applications/groth16_aggregation.rs:77-160) on model values."""
    n = len(a)
    ck_1, ck_2 = h_beta_powers[::2], g_alpha_powers[::2]
    com_a, com_b, com_c = M.pairing_product(a, ck_1), M.pairing_product(ck_2, b), M.pairing_product(c, ck_1)
    r = aggregation_challenge(com_a, com_b, com_c)
    r_vec = [pow(r, i, R) for i in range(n)]
    a_r = [M.g1_mul(ri, ai) for ai, ri in zip(a, r_vec)]
    ip_ab = M.pairing_product(a_r, b); agg_c = M.g1_msm(c, r_vec)
    ck_1_r = [M.g2_mul(pow(ri, -1, R), k) for k, ri in zip(ck_1, r_vec)]
    assert M.pairing_product(a_r, ck_1_r) == com_a
    tipp = prove_tipa_tipp(g_alpha_powers, h_beta_powers, a_r, b, ck_1_r, ck_2, r)
    ssm = prove_tipa_ssm(h_beta_powers, c, r_vec, ck_1)
    return dict(com_a=com_a, com_b=com_b, com_c=com_c, ip_ab=ip_ab, agg_c=agg_c, r=r, tipp=tipp, ssm=ssm)
