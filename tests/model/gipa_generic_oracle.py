"""CPU restatement of the GENERIC GIPA prover / verifier (ip_proofs/src/gipa.rs:97-415) and of GIPAWithSSM
(tipa/structured_scalar_message.rs:56-128) over the C oracle's primitives -- test infrastructure, the checker for
ripp_amd/gipa.py.  Written independently of that module: values travel as (tag, payload) pairs, scalars as Python integers, and the
control flow follows the reference's recursion literally (small sizes only: per-element Python loops around oracle calls).

Instantiations are given as a 4-tuple of tags:
    ("PAIR",  "AFGHO1", "AFGHO2", "GT")   pairing_inner_product_test             gipa.rs:470-497
    ("MEXP1", "AFGHO1", "PED1",   "G1")   multiexponentiation_inner_product_test gipa.rs:499-530
    ("SCAL",  "PED2",   "PED2",   "FR")   scalar_inner_product_test              gipa.rs:532-561
    ("MEXP1", "AFGHO1", "SSM",    "G1")   GIPAWithSSM as used by TIPAWithSSM     structured_scalar_message.rs:211-268
"""
import hashlib

import numpy as np
import orclib as o

R = o.R


def fr(v): return o.fr_array([v % R])[0]


# ---- per-type operations: serialise (uncompressed), scale by an integer, add, canonical form
def ser(tag, v):
    if tag == "GT": return o.ser_gt(v)
    if tag == "G1": return o.ser_g1(o.g1_to_affine(v))
    if tag == "G2": return o.ser_g2(o.g2_to_affine(v))
    if tag == "FR": return (v % R).to_bytes(32, "little")
    if tag == "UNIT": return b""
    raise KeyError(tag)


def scale(tag, v, k):
    if tag == "GT": return o.gt_pow(v, fr(k))
    if tag == "G1": return o.fold_g1_j(v[None], np.zeros((1, 18), dtype=np.uint64), fr(k))[0]
    if tag == "G2": return o.fold_g2_j(v[None], np.zeros((1, 36), dtype=np.uint64), fr(k))[0]
    if tag == "FR": return v * k % R
    return v


def plus(tag, a, b):
    if tag == "GT": return o.gt_mul(a, b)
    if tag == "G1": return o.fold_g1_j(a[None], b[None], fr(1))[0]
    if tag == "G2": return o.fold_g2_j(a[None], b[None], fr(1))[0]
    if tag == "FR": return (a + b) % R
    return a


def same(tag, a, b):
    if tag == "G1": return np.array_equal(o.g1_to_affine(a), o.g1_to_affine(b))
    if tag == "G2": return np.array_equal(o.g2_to_affine(a), o.g2_to_affine(b))
    if tag == "FR": return a % R == b % R
    if tag == "GT": return np.array_equal(a, b)
    return True


def fold(tag, hi, lo, k):
    """[hi_i * k + lo_i]"""
    if tag == "G1": return o.fold_g1_j(np.ascontiguousarray(hi), np.ascontiguousarray(lo), fr(k))
    if tag == "G2": return o.fold_g2_j(np.ascontiguousarray(hi), np.ascontiguousarray(lo), fr(k))
    if tag == "FR": return [(h * k + l) % R for h, l in zip(hi, lo)]
    return list(hi)


def frs(v): return o.fr_array([x % R for x in v]) if len(v) else np.zeros((0, 4), dtype=np.uint64)


def inner_product(ip, left, right):
    if ip == "PAIR": return o.pairing_product_j(np.ascontiguousarray(left), np.ascontiguousarray(right))[1]
    if ip == "MEXP1": return o.msm_g1_j(np.ascontiguousarray(left), frs(right))[1]
    if ip == "SCAL": return sum(a * b for a, b in zip(left, right)) % R
    raise KeyError(ip)


COMMIT = {  # name -> (message tag, key tag, output tag, commit(k, m))
    "AFGHO1": ("G1", "G2", "GT", lambda k, m: o.pairing_product_j(np.ascontiguousarray(m), np.ascontiguousarray(k))[1]),
    "AFGHO2": ("G2", "G1", "GT", lambda k, m: o.pairing_product_j(np.ascontiguousarray(k), np.ascontiguousarray(m))[1]),
    "PED1": ("FR", "G1", "G1", lambda k, m: o.msm_g1_j(np.ascontiguousarray(k), frs(m))[1]),
    "PED2": ("FR", "G2", "G2", lambda k, m: o.msm_g2_j(np.ascontiguousarray(k), frs(m))[1]),
    "SSM": ("FR", "UNIT", "FR", lambda k, m: 0),
}
IP_TYPES = {"PAIR": ("G1", "G2", "GT"), "MEXP1": ("G1", "FR", "G1"), "SCAL": ("FR", "FR", "FR")}


def challenge(inst, prev, com_1, com_2):
    ip, lmc, rmc, t = inst
    nonce = 0
    while True:
        h = nonce.to_bytes(8, "big") + (prev or 0).to_bytes(32, "little")
        for c in (com_1, com_2):
            h += ser(COMMIT[lmc][2], c[0]) + ser(COMMIT[rmc][2], c[1]) + (1).to_bytes(8, "little") + ser(t, c[2])
        c128 = int.from_bytes(hashlib.blake2b(h).digest()[:16], "big")
        if c128:
            return pow(c128, -1, R), c128
        nonce += 1


def prove(inst, m_a, m_b, ck_a, ck_b):
    """-> (steps [(com_1, com_2)] in ROUND order, transcript [c] in round order, (a_base, b_base), (ck_a_base, ck_b_base))"""
    ip, lmc, rmc, t = inst
    steps, tr = [], []
    while len(m_a) > 1:
        s = len(m_a) // 2
        com_1 = (COMMIT[lmc][3](ck_a[:s], m_a[s:]), COMMIT[rmc][3](ck_b[s:], m_b[:s]), inner_product(ip, m_a[s:], m_b[:s]))
        com_2 = (COMMIT[lmc][3](ck_a[s:], m_a[:s]), COMMIT[rmc][3](ck_b[:s], m_b[s:]), inner_product(ip, m_a[:s], m_b[s:]))
        c, c_inv = challenge(inst, tr[-1] if tr else None, com_1, com_2)
        m_a = fold(COMMIT[lmc][0], m_a[s:], m_a[:s], c)
        m_b = fold(COMMIT[rmc][0], m_b[s:], m_b[:s], c_inv)
        ck_a = fold(COMMIT[lmc][1], ck_a[s:], ck_a[:s], c_inv)
        ck_b = fold(COMMIT[rmc][1], ck_b[s:], ck_b[:s], c) if COMMIT[rmc][1] != "UNIT" else ck_b[:s]      # ck_b_1 = ck_b[split..] (gipa.rs:216, 286-290)
        steps.append((com_1, com_2)); tr.append(c)
    return steps, tr, (m_a[0], m_b[0]), (ck_a[0], ck_b[0])


def verify(inst, ck_a, ck_b, com, steps, base, scalar_b=None):
    """GIPA::verify (gipa.rs:135-160); with scalar_b: GIPAWithSSM::verify_with_structured_scalar_message (ssm.rs:85-128)."""
    ip, lmc, rmc, t = inst
    outs = (COMMIT[lmc][2], COMMIT[rmc][2], t)
    cur = list(com); tr = []
    for com_1, com_2 in steps:                                   # steps are in round order here
        c, c_inv = challenge(inst, tr[-1] if tr else None, com_1, com_2)
        cur = [plus(o_, plus(o_, scale(o_, x1, c), cu), scale(o_, x2, c_inv)) for o_, x1, cu, x2 in zip(outs, com_1, cur, com_2)]
        tr.append(c)
    rev = tr[::-1]                                               # the reference's r_transcript order
    ea, eb = [1], [1]
    for i, c in enumerate(rev):
        for j in range(1 << i):
            ea.append(ea[j] * pow(c, -1, R) % R); eb.append(eb[j] * c % R)
    def key_sum(tag, keys, ex):
        if tag == "UNIT": return None
        acc = scale(tag, keys[0], ex[0])
        for k, x in zip(keys[1:], ex[1:]):
            acc = plus(tag, acc, scale(tag, k, x))               # the reference's sequential fold (gipa.rs:385-396)
        return acc
    ka, kb = key_sum(COMMIT[lmc][1], ck_a, ea), key_sum(COMMIT[rmc][1], ck_b, eb)
    a_base, b_base = base
    def wrap(tag, x): return [x] if tag in ("FR", "UNIT") else x[None]
    ok = same(outs[0], COMMIT[lmc][3](wrap(COMMIT[lmc][1], ka), wrap(COMMIT[lmc][0], a_base)), cur[0])
    ok_b = same(outs[1], COMMIT[rmc][3](wrap(COMMIT[rmc][1], kb), wrap(COMMIT[rmc][0], b_base)), cur[1])
    ok_t = same(outs[2], inner_product(ip, wrap(IP_TYPES[ip][0], a_base), wrap(IP_TYPES[ip][1], b_base)), cur[2])
    if scalar_b is None:
        return bool(ok and ok_b and ok_t)
    p2b, bb = scalar_b % R, 1
    for x in rev:
        bb = bb * (1 + pow(x, -1, R) * p2b) % R; p2b = p2b * p2b % R
    ok_t2 = same(outs[2], inner_product(ip, wrap(IP_TYPES[ip][0], a_base), [bb]), cur[2])
    return bool(ok and ok_b and ok_t and ok_t2)
