"""Wire format, host-only code of libripp_hip -- runs without a GPU.  The product's serialisers are
compared with the big-integer restatement in oracle/wire_format.py on proofs made by the CPU oracle, and the deserialisers must
round-trip and reject malformed images (range, curve equation, subgroup, flags, trailing bytes)."""
import os
import sys

import numpy as np
import pytest

import helpers as h
import orclib as o

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import wire_format as W  # noqa: E402


def gt_ints(f): return [o.limbs_to_fp(f[6 * i:6 * i + 6]) for i in range(12)]
def g1_pt(pj): return o.g1_from_row(o.g1_to_affine(np.ascontiguousarray(pj)))
def g2_pt(pj): return o.g2_from_row(o.g2_to_affine(np.ascontiguousarray(pj)))


@pytest.fixture(scope="module")
def api():
    import ripp_amd
    return ripp_amd


@pytest.fixture(scope="module")
def tipp_proof():
    n = 4; srs = h.make_srs(n, 0x77, 0x99); ck_a, ck_b = h.commitment_keys(srs)
    m_a, m_b = o.blind_g1(o.gen_g1(3, n), 1), o.blind_g2(o.gen_g2(5, n), 2)
    rc, pf = o.tipa_tipp_prove(srs[0], srs[1], m_a, m_b, ck_a, ck_b, o.fr_array([1])[0]); assert rc == 0
    return pf


@pytest.fixture(scope="module")
def ssm_proof():
    n = 4; srs = h.make_srs(n, 0x55, 0x33); ck_a, _ = h.commitment_keys(srs)
    m_a = o.blind_g1(o.gen_g1(9, n), 5); m_b = o.fr_array([pow(0xabcdef, i, o.R) for i in range(n)])
    rc, pf = o.tipa_ssm_prove(srs[1], m_a, m_b, ck_a); assert rc == 0
    return pf


@pytest.mark.parametrize("compress", [False, True])
def test_point_encodings(api, compress):
    pts = o.gen_g1(17, 6); qts = o.gen_g2(23, 6)
    for i in range(6):
        exp1 = W.ser_g1(o.g1_from_row(pts[i]), compress); exp2 = W.ser_g2(o.g2_from_row(qts[i]), compress)
        got1 = api.ser_g1_compressed(pts[i]) if compress else api.ser_g1(pts[i]); got2 = api.ser_g2_compressed(qts[i]) if compress else api.ser_g2(qts[i])
        assert got1 == exp1 and got2 == exp2
    z1, z2 = np.zeros(12, dtype=np.uint64), np.zeros(24, dtype=np.uint64)
    assert (api.ser_g1_compressed(z1) if compress else api.ser_g1(z1)) == W.ser_g1(None, compress)
    assert (api.ser_g2_compressed(z2) if compress else api.ser_g2(z2)) == W.ser_g2(None, compress)
    # both sign flags occur among a handful of points (the flag is live)
    if compress:
        assert len({api.ser_g1_compressed(pts[i])[0] & 0x20 for i in range(6)}) == 2


@pytest.mark.parametrize("compress", [False, True])
def test_tipa_proof_image_and_round_trip(api, tipp_proof, compress):
    pf = tipp_proof; rounds = len(pf["steps"]) // 6
    steps = [tuple(gt_ints(pf["steps"][6 * k + j]) for j in range(6)) for k in range(rounds)]
    exp = W.tipa_tipp_proof(steps, g1_pt(pf["base_a"]), g2_pt(pf["base_b"]), g2_pt(pf["final_ck_a"]), g1_pt(pf["final_ck_b"]), g2_pt(pf["opening_a"]), g1_pt(pf["opening_b"]), compress)
    got = api.ser_tipa_tipp_proof(pf, compress=compress)
    assert got == exp
    assert len(got) == 8 + rounds * 2 * (3 * 576 + 8) + (3 * 48 + 3 * 96 if compress else 3 * 96 + 3 * 192)
    assert api.ser_tipa_tipp_proof(pf, compress=compress, with_tipa=False) == W.gipa_tipp_proof(steps, g1_pt(pf["base_a"]), g2_pt(pf["base_b"]), compress)
    back = api.de_tipa_tipp_proof(got, compress=compress)
    assert np.array_equal(back["steps"], pf["steps"])
    for k in ("base_a", "final_ck_b", "opening_b"):
        assert g1_pt(back[k]) == g1_pt(pf[k])
    for k in ("base_b", "final_ck_a", "opening_a"):
        assert g2_pt(back[k]) == g2_pt(pf[k])
    assert api.ser_tipa_tipp_proof(back, compress=compress) == got


@pytest.mark.parametrize("compress", [False, True])
def test_ssm_proof_image_and_round_trip(api, ssm_proof, compress):
    pf = ssm_proof; rounds = len(pf["com_gt"]) // 2
    gts = [(gt_ints(pf["com_gt"][2 * k]), gt_ints(pf["com_gt"][2 * k + 1])) for k in range(rounds)]
    g1s = [(g1_pt(pf["com_g1"][2 * k]), g1_pt(pf["com_g1"][2 * k + 1])) for k in range(rounds)]
    exp = W.tipa_ssm_proof(gts, g1s, g1_pt(pf["base_a"]), o.limbs_to_fr(pf["base_b"]), g2_pt(pf["final_ck_a"]), g2_pt(pf["opening_a"]), compress)
    got = api.ser_tipa_ssm_proof(pf, compress=compress)
    assert got == exp
    back = api.de_tipa_ssm_proof(got, compress=compress)
    assert np.array_equal(back["com_gt"], pf["com_gt"]) and np.array_equal(back["base_b"], pf["base_b"])
    assert api.ser_tipa_ssm_proof(back, compress=compress) == got


def test_deserialisers_reject_malformed_images(api, tipp_proof):
    good = bytearray(api.ser_tipa_tipp_proof(tipp_proof, compress=True))
    rounds = len(tipp_proof["steps"]) // 6
    off_base_a = 8 + rounds * 2 * (3 * 576 + 8)
    def bad(mut):
        b = bytearray(good); mut(b)
        with pytest.raises(ValueError):
            api.de_tipa_tipp_proof(bytes(b), compress=True)
    bad(lambda b: b.append(0))                                         # trailing byte
    bad(lambda b: b.__delitem__(len(b) - 1))                           # truncated
    bad(lambda b: b.__setitem__(off_base_a, b[off_base_a] & 0x7f))     # compression flag cleared
    bad(lambda b: b.__setitem__(slice(8, 56), b"\xff" * 48))          # Fp coefficient >= p inside a GT
    bad(lambda b: b.__setitem__(8 + 2 * 576, 2))                      # IdentityOutput length != 1
    # PairingOutput::check (ark-ec 0.4): an Fq12 outside the order-r subgroup is rejected -- the constant 2, zero, and a
    # cyclotomic-looking value that is just another step's coefficient vector permuted
    bad(lambda b: b.__setitem__(slice(8, 8 + 576), (2).to_bytes(48, "little") + bytes(528)))
    bad(lambda b: b.__setitem__(slice(8, 8 + 576), bytes(576)))
    bad(lambda b: b.__setitem__(slice(8, 8 + 576), bytes(b[8 + 96:8 + 576]) + bytes(b[8:8 + 96])))
    # an x with no point on the curve / a point outside the prime-order subgroup
    x = 0
    while True:
        x += 1
        y2 = (x * x * x + 4) % W.P
        if pow(y2, (W.P - 1) // 2, W.P) != 1:
            break
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    bad(lambda b: b.__setitem__(slice(off_base_a, off_base_a + 48), enc))
    x = 0
    while True:       # on the curve but (with overwhelming probability) not in the order-r subgroup: cofactor of E(Fp) is ~2^126
        x += 1
        y2 = (x * x * x + 4) % W.P
        if pow(y2, (W.P - 1) // 2, W.P) == 1:
            break
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    bad(lambda b: b.__setitem__(slice(off_base_a, off_base_a + 48), enc))
    assert api.de_tipa_tipp_proof(bytes(good), compress=True)["steps"].shape[0] == rounds * 6
