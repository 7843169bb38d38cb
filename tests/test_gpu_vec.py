"""GPU (-m gpu): device-resident vectors (include/ripp_hip.h `ripp_vec_*`) -- every operation on a resident vector
or view equals the host-slice entry point on the same elements (which the other GPU tests pin to the oracle), and the generic GIPA gives
the SAME proof with resident vectors as with per-call uploads."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _aff1(engine, pj): return engine.normalize_batch_g1(np.asarray(pj).reshape(-1, 18))
def _aff2(engine, pj): return engine.normalize_batch_g2(np.asarray(pj).reshape(-1, 36))


@pytest.mark.parametrize("n", [1, 2, 16, 1000, 1 << 14])
def test_vec_round_trip_views_and_inner_products(engine, orc, n):
    a, b, r = orc.blind_g1(orc.gen_g1(7, n), 1), orc.blind_g2(orc.gen_g2(9, n), 2), orc.gen_scalars(3, n)
    va, vb, vr = engine.Vec.upload("G1", a), engine.Vec.upload("G2", b), engine.Vec.upload("Fr", r)
    assert (len(va), len(vb), len(vr)) == (n, n, n)
    assert np.array_equal(va.download(), _aff1(engine, a)) and np.array_equal(vb.download(), _aff2(engine, b)) and np.array_equal(vr.download(), r)
    # affine upload = projective upload
    assert np.array_equal(engine.Vec.upload("G1", _aff1(engine, a)).download(), va.download())
    # the three inner products on resident vectors = the host-slice forms
    assert np.array_equal(engine.PairingInnerProduct.inner_product(va, vb), engine.PairingInnerProduct.inner_product(a, b))
    assert np.array_equal(_aff1(engine, engine.MultiexponentiationInnerProductG1.inner_product(va, vr)), _aff1(engine, engine.MultiexponentiationInnerProductG1.inner_product(a, r)))
    assert np.array_equal(_aff2(engine, engine.MultiexponentiationInnerProductG2.inner_product(vb, vr)), _aff2(engine, engine.MultiexponentiationInnerProductG2.inner_product(b, r)))
    assert np.array_equal(engine.ScalarInnerProduct.inner_product(vr, vr), engine.ScalarInnerProduct.inner_product(r, r))
    # the pairing product is checked against the oracle directly as well
    rc, exp = orc.pairing_product_j(a, b); assert rc == 0 and np.array_equal(engine.PairingInnerProduct.inner_product(va, vb), exp)
    if n >= 2:
        s = n // 2
        lo, hi = va[:s], va[s:]
        assert np.array_equal(lo.download(), va.download()[:s]) and np.array_equal(hi.download(), va.download()[s:])
        assert np.array_equal(engine.PairingInnerProduct.inner_product(hi, vb[:s]), engine.PairingInnerProduct.inner_product(a[s:], b[:s]))   # the cross terms of a round
        # element access gives the host layout back (projective, Z = 1)
        assert np.array_equal(_aff1(engine, va[s]), _aff1(engine, a[s])) and np.array_equal(vr[n - 1], r[n - 1])
        # a view keeps the storage alive after its parent is gone
        tmp = engine.Vec.upload("G2", b); view = tmp[s:]; tmp.close()
        assert np.array_equal(view.download(), _aff2(engine, b)[s:])


@pytest.mark.parametrize("n", [2, 16, 512, 1 << 13, 1 << 16])
def test_vec_folds(engine, orc, n):
    """out[i] = s hi[i] + lo[i] on resident halves = ripp_fold_* on host slices (every size class: VM, split, GLS, table kernels)"""
    a, b, r = orc.blind_g1(orc.gen_g1(70, n), 5), orc.blind_g2(orc.gen_g2(90, n), 6), orc.gen_scalars(8, n)
    c = orc.fr_array([0x1234567890ABCDEF0FEDCBA098765432 << 64 | 0x1111])[0]
    c128 = orc.fr_array([0xFEDCBA9876543210FEDCBA9876543210])[0]
    s = n // 2
    va, vb, vr = engine.Vec.upload("G1", a), engine.Vec.upload("G2", b), engine.Vec.upload("Fr", r)
    for sc in (c, c128):
        assert np.array_equal(va[s:].fold(va[:s], sc).download(), _aff1(engine, engine.fold_g1(a[s:], a[:s], sc)))
        assert np.array_equal(vb[s:].fold(vb[:s], sc).download(), _aff2(engine, engine.fold_g2(b[s:], b[:s], sc)))
    import ripp_amd.gipa as G
    assert np.array_equal(vr[s:].fold(vr[:s], c).download(), G.Fr.fold(r[s:], r[:s], c))
    # exceptional cases inside a fold: hi == lo (s = 1 doubles, s = -1 cancels), infinity in either operand
    one, minus1 = orc.fr_array([1])[0], orc.fr_array([orc.R - 1])[0]
    assert np.array_equal(va[:s].fold(va[:s], one).download(), _aff1(engine, engine.fold_g1(a[:s], a[:s], one)))
    assert not va[:s].fold(va[:s], minus1).download().any()                                   # all points at infinity: (0, 0)
    z = np.zeros((s, 18), dtype=np.uint64); vz = engine.Vec.upload("G1", z)
    assert np.array_equal(vz.fold(va[:s], c).download(), _aff1(engine, a[:s])) and np.array_equal(va[:s].fold(vz, one).download(), _aff1(engine, a[:s]))


def test_vec_errors_and_lifecycle(engine, orc):
    a, b = orc.gen_g1(1, 8), orc.gen_g2(2, 4)
    va, vb = engine.Vec.upload("G1", a), engine.Vec.upload("G2", b)
    with pytest.raises(engine.InnerProductError) as ei:
        engine.PairingInnerProduct.inner_product(va, vb)
    assert (ei.value.left, ei.value.right) == (8, 4)
    with pytest.raises(ValueError):
        engine.PairingInnerProduct.inner_product(vb, va)                                        # kinds swapped
    with pytest.raises(ValueError):
        engine.Vec.upload("G1", np.zeros((2, 7), dtype=np.uint64))
    with pytest.raises(IndexError):
        va[::2]
    with pytest.raises(engine.InnerProductError):
        va[:4].fold(va[:2], orc.fr_array([3])[0])
    # a live vector pins the engine to its device like a job / SRS handle does
    from ripp_amd._lib import lib
    if lib().ripp_device_count() > 1:
        assert lib().ripp_init(1) == 4
    # empty vectors
    e1, e2 = engine.Vec.upload("G1", np.zeros((0, 12), dtype=np.uint64)), engine.Vec.upload("G2", np.zeros((0, 24), dtype=np.uint64))
    assert np.array_equal(engine.PairingInnerProduct.inner_product(e1, e2), orc.gt_one())


@pytest.mark.parametrize("n", [8, 256])
def test_generic_gipa_resident_equals_host_slices(engine, orc, n):
    """the same proof, commitment for commitment, whether the vectors live in HBM between rounds or travel with every call"""
    import ripp_amd.gipa as G
    m_a, m_b = orc.blind_g1(orc.gen_g1(11, n), 1), orc.blind_g2(orc.gen_g2(22, n), 2)
    ck_a, ck_b = orc.blind_g2(orc.gen_g2(33, n), 3), orc.blind_g1(orc.gen_g1(44, n), 4)
    args = (G.PairingIP, G.AFGHOCommitmentG1, G.AFGHOCommitmentG2, G.IdentityCommitment(G.GT))
    p1, x1 = G.GIPA(*args, resident=True).prove_with_aux((m_a, m_b), (ck_a, ck_b, [None]))
    p2, x2 = G.GIPA(*args, resident=False).prove_with_aux((m_a, m_b), (ck_a, ck_b, [None]))
    for s1, s2 in zip(p1["r_commitment_steps"], p2["r_commitment_steps"]):
        for side in range(2):
            assert np.array_equal(s1[side][0], s2[side][0]) and np.array_equal(s1[side][1], s2[side][1]) and np.array_equal(s1[side][2][0], s2[side][2][0])
    assert np.array_equal(np.stack(x1["r_transcript"]), np.stack(x2["r_transcript"]))
    assert G.G1.canon(p1["r_base"][0]) == G.G1.canon(p2["r_base"][0]) and G.G2.canon(p1["r_base"][1]) == G.G2.canon(p2["r_base"][1])
    assert G.G2.canon(x1["ck_base"][0]) == G.G2.canon(x2["ck_base"][0]) and G.G1.canon(x1["ck_base"][1]) == G.G1.canon(x2["ck_base"][1])
