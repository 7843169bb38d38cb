"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no Rust toolchain / ark-serialize here; see oracle/ripp_oracle.c).

Big-integer restatement of the `CanonicalSerialize` images ark-serialize 0.4 derives for the proof structs
  GIPAProof (ip_proofs/src/gipa.rs:24-51), TIPAProof (ip_proofs/src/tipa/mod.rs:41-65),
  TIPAWithSSMProof (ip_proofs/src/tipa/structured_scalar_message.rs:138-156),
with ark-bls12-381 0.4's zcash point encoding.  Inputs are plain Python integers / tuples:
  Fp, Fr: int;  Fp2: (c0, c1);  G1 affine: (x, y) or None;  G2 affine: ((x0, x1), (y0, y1)) or None;  GT: 12 ints in tower order
  (c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1).
"""
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def _fp_be(v): return int(v).to_bytes(48, "big")
def _fp_le(v): return int(v).to_bytes(48, "little")
def ser_fr(s): return int(s).to_bytes(32, "little")
def ser_gt(f): return b"".join(_fp_le(c) for c in f)
def ser_u64(v): return int(v).to_bytes(8, "little")


def _largest_fp(y): return y > (P - y) % P
def _largest_fp2(y): return _largest_fp(y[1]) if y[1] != 0 else _largest_fp(y[0])      # Ord of QuadExtField: c1, then c0


def ser_g1(pt, compress):
    n = 48 if compress else 96
    if pt is None:
        return bytes([0xC0 if compress else 0x40]) + bytes(n - 1)
    if not compress:
        return _fp_be(pt[0]) + _fp_be(pt[1])
    b = bytearray(_fp_be(pt[0])); b[0] |= 0x80 | (0x20 if _largest_fp(pt[1]) else 0); return bytes(b)


def ser_g2(pt, compress):
    n = 96 if compress else 192
    if pt is None:
        return bytes([0xC0 if compress else 0x40]) + bytes(n - 1)
    (x0, x1), (y0, y1) = pt
    if not compress:
        return _fp_be(x1) + _fp_be(x0) + _fp_be(y1) + _fp_be(y0)
    b = bytearray(_fp_be(x1) + _fp_be(x0)); b[0] |= 0x80 | (0x20 if _largest_fp2((y0, y1)) else 0); return bytes(b)


def gipa_tipp_proof(steps_round_order, base_a, base_b, compress):
    """steps_round_order: list of 6-tuples of GT (com_1.0, com_1.1, com_1.2[0], com_2.0, com_2.1, com_2.2[0]), first round first."""
    out = ser_u64(len(steps_round_order))
    for s in reversed(steps_round_order):                       # r_commitment_steps.reverse(), gipa.rs:299
        for side in (0, 3):
            out += ser_gt(s[side]) + ser_gt(s[side + 1]) + ser_u64(1) + ser_gt(s[side + 2])      # IdentityOutput(Vec<GT>)
    return out + ser_g1(base_a, compress) + ser_g2(base_b, compress)


def tipa_tipp_proof(steps_round_order, base_a, base_b, final_ck_a, final_ck_b, opening_a, opening_b, compress):
    return (gipa_tipp_proof(steps_round_order, base_a, base_b, compress) + ser_g2(final_ck_a, compress) + ser_g1(final_ck_b, compress)
            + ser_g2(opening_a, compress) + ser_g1(opening_b, compress))


def tipa_ssm_proof(com_gt_round_order, com_g1_round_order, base_a, base_b, final_ck_a, opening_a, compress):
    """com_gt / com_g1: per round (com_1.0, com_2.0) and (com_1.2[0], com_2.2[0])."""
    out = ser_u64(len(com_gt_round_order))
    for gt, g1 in zip(reversed(com_gt_round_order), reversed(com_g1_round_order)):
        for side in (0, 1):
            out += ser_gt(gt[side]) + ser_fr(0) + ser_u64(1) + ser_g1(g1[side], compress)
    return out + ser_g1(base_a, compress) + ser_fr(base_b) + ser_g2(final_ck_a, compress) + ser_g2(opening_a, compress)
