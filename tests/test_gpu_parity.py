"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI, against the CPU oracle on identical seeded
inputs (bit-exact: integer arithmetic), against the committed golden vectors, and -- at BASELINE.json's full sizes --
through size-independent properties.  Mirrors the reference's own tests where they exist
(dh_commitments/src/afgho16/mod.rs:61-94, pedersen/mod.rs:39-55, sipp/src/lib.rs:232-254)."""
import numpy as np
import pytest

from helpers import g1arr, g2arr, frarr, gt_from_bytes, g1pt, g2pt

pytestmark = pytest.mark.gpu


def test_library_is_the_hip_engine(engine):
    from ripp_amd._lib import lib
    assert lib().ripp_device_count() >= 1


def test_synthetic_inputs_match_oracle(engine, orc):
    n = 257
    assert np.array_equal(engine.synth_g1(1000, n), orc.gen_g1(1000, n))
    assert np.array_equal(engine.synth_g2(2000, n), orc.gen_g2(2000, n))
    assert np.array_equal(engine.synth_fr(3, n), orc.gen_scalars(3, n))
    # strided shard == every world-th element
    assert np.array_equal(engine.synth_g1(1000, 64, first=1, stride=4), orc.gen_g1(1000, 256)[1::4])
    assert np.array_equal(engine.synth_fr(3, 64, first=3, stride=4), orc.gen_scalars(3, 256)[3::4])


def test_golden_pairing_vectors(engine, vectors):
    g = vectors["generators"]
    e = engine.product_of_pairings(g1arr([g["g1"]]), g2arr([g["g2"]]))
    assert engine.ser_gt(e).hex() == vectors["pairing_generators"]["gt"]
    v = vectors["product8"]
    assert engine.ser_gt(engine.product_of_pairings(g1arr(v["a"]), g2arr(v["b"]))).hex() == v["gt"]
    assert engine.ser_g1(g1arr([g["g1"]])[0]).hex() == g["ser_g1"] and engine.ser_g2(g2arr([g["g2"]])[0]).hex() == g["ser_g2"]


def test_golden_sipp_vector(engine, vectors):
    v = vectors["sipp4"]
    a, b, r = g1arr(v["a"]), g2arr(v["b"]), frarr(v["r"])
    value = engine.product_of_pairings_with_coeffs(a, b, r)
    assert engine.ser_gt(value).hex() == v["value"]
    assert engine.sipp_seed_digest(a, b, r, value).hex() == v["seed_digest"]
    proof = engine.SIPP.prove(a, b, r, value)
    assert [[engine.ser_gt(proof[2 * j]).hex(), engine.ser_gt(proof[2 * j + 1]).hex()] for j in range(2)] == v["proof"]
    assert engine.SIPP.verify(a, b, r, value, proof)


def test_golden_msm_and_fold(engine, orc, vectors):
    v = vectors["msm8"]; sc = frarr(v["scalars"])
    r1 = engine.MultiexponentiationInnerProductG1.inner_product(orc.blind_g1(g1arr(v["g1_bases"]), 5), sc)
    r2 = engine.MultiexponentiationInnerProductG2.inner_product(orc.blind_g2(g2arr(v["g2_bases"]), 6), sc)
    assert orc.g1_from_row(engine.normalize_batch_g1(r1)[0]) == g1pt(v["g1"])
    assert orc.g2_from_row(engine.normalize_batch_g2(r2)[0]) == g2pt(v["g2"])
    f = vectors["fold"]; s = frarr([f["s"]])[0]
    assert orc.g1_from_row(engine.fold_g1_affine(g1arr([f["g1_hi"]]), g1arr([f["g1_lo"]]), s)[0]) == g1pt(f["g1"])
    assert orc.g2_from_row(engine.fold_g2_affine(g2arr([f["g2_hi"]]), g2arr([f["g2_lo"]]), s)[0]) == g2pt(f["g2"])


@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 64, 1000])
def test_pairing_inner_product_vs_oracle(engine, orc, n):
    """PairingInnerProduct::inner_product on projective inputs with random Z and points at infinity (config 2 shape)."""
    a, b = orc.gen_g1(11, n), orc.gen_g2(17, n)
    aj, bj = orc.blind_g1(a, 1), orc.blind_g2(b, 2)
    if n > 4:
        aj[1] = 0; bj[3] = 0                      # infinity on either side contributes 1
    rc, exp = orc.pairing_product_j(aj, bj)
    assert rc == 0 and np.array_equal(engine.PairingInnerProduct.inner_product(aj, bj), exp)


def test_pairing_inner_product_length_error(engine, orc):
    aj, bj = orc.blind_g1(orc.gen_g1(1, 5), 1), orc.blind_g2(orc.gen_g2(1, 4), 2)
    with pytest.raises(engine.InnerProductError) as ei:
        engine.PairingInnerProduct.inner_product(aj, bj)
    assert str(ei.value) == "left length, right length: 5, 4"       # Display impl, inner_products/src/lib.rs:29-38


def test_afgho_commitments(engine, orc):
    n = 8
    for C, keys, mk in ((engine.AFGHOCommitmentG1, orc.blind_g2(orc.gen_g2(10, n), 1), lambda s, m: orc.blind_g1(orc.gen_g1(s, m), 2)),
                        (engine.AFGHOCommitmentG2, orc.blind_g1(orc.gen_g1(10, n), 1), lambda s, m: orc.blind_g2(orc.gen_g2(s, m), 2))):
        msg, wrong = mk(20, n), mk(30, n)
        com = C.commit(keys, msg)
        assert C.verify(keys, msg, com) and not C.verify(keys, wrong, com)
        with pytest.raises(engine.InnerProductError):
            C.verify(keys, mk(20, n + 1), com)
    # AFGHO-G1 commit(k, m) = e-product(m, k): same value as the oracle's PairingInnerProduct(m, k)
    keys, msg = orc.blind_g2(orc.gen_g2(10, n), 1), orc.blind_g1(orc.gen_g1(20, n), 2)
    assert np.array_equal(engine.AFGHOCommitmentG1.commit(keys, msg), orc.pairing_product_j(msg, keys)[1])


def test_pedersen_commitment(engine, orc):
    n = 8
    keys, msg, wrong = orc.blind_g1(orc.gen_g1(40, n), 4), orc.gen_scalars(5, n), orc.gen_scalars(6, n)
    C = engine.PedersenCommitmentG1
    com = C.commit(keys, msg)
    assert C.verify(keys, msg, com) and not C.verify(keys, wrong, com)
    with pytest.raises(engine.InnerProductError):
        C.verify(keys, orc.gen_scalars(5, n + 1), com)
    # the committed VALUE against the oracle (pedersen/mod.rs:24-26 = MultiexponentiationInnerProduct::inner_product(k, m)), both groups
    rc, exp = orc.msm_g1_j(keys, msg); assert rc == 0
    assert np.array_equal(engine.normalize_batch_g1(com), orc.normalize_g1(exp.reshape(1, 18)))
    keys2 = orc.blind_g2(orc.gen_g2(41, n), 5)
    rc, exp2 = orc.msm_g2_j(keys2, msg); assert rc == 0
    assert np.array_equal(engine.normalize_batch_g2(engine.PedersenCommitmentG2.commit(keys2, msg)), orc.normalize_g2(exp2.reshape(1, 36)))


def test_engine_lifecycle_guards(engine, orc):
    """ripp_init on another ordinal is refused while handles are alive; ripp_release_scratch frees the grow-only buffers and the next call
    simply re-allocates (same results)."""
    import ctypes
    from ripp_amd._lib import lib
    n = 64
    a, b, r = orc.gen_g1(5, n), orc.gen_g2(6, n), orc.gen_scalars(7, n)
    before = engine.product_of_pairings(a, b)
    job = engine.SippJob(a, b, r)
    assert lib().ripp_init(ctypes.c_int32(1)) == 4                     # RIPP_ERR_ARG: a live job pins the engine to its device
    job.close()
    assert lib().ripp_release_scratch() == 0
    assert np.array_equal(engine.product_of_pairings(a, b), before)
    assert lib().ripp_init(ctypes.c_int32(0)) == 0


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1 << 10, (1 << 12) + 5])
def test_msm_vs_oracle(engine, orc, n):
    s = orc.gen_scalars(21, n)
    b1, b2 = orc.gen_g1(5, n), orc.gen_g2(6, n)
    assert np.array_equal(engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.blind_g1(b1, 9), s)), orc.g1_to_affine(orc.msm_g1_a(b1, s)).reshape(1, 12))
    if n <= 1 << 10:
        assert np.array_equal(engine.normalize_batch_g2(engine.MultiexponentiationInnerProductG2.inner_product(orc.blind_g2(b2, 9), s)), orc.g2_to_affine(orc.msm_g2_a(b2, s)).reshape(1, 24))


def test_one_shot_proofs_reuse_their_buffers(engine, orc):
    """ripp_sipp_prove hands the device / pinned buffers of its job to the engine when it returns and the next one-shot call adopts them (Engine::job_cache:
    the hipMalloc / hipFree of ~1 GB per call were ~10 ms of every host-slice proof at n = 2^20).  A larger, a smaller and an equal statement after one
    another, a resident job's proof in between, and a proof after ripp_release_scratch (which frees the cache): every proof is the oracle's."""
    import ctypes
    from ripp_amd._lib import lib
    stmts = {}
    for lg in (12, 4, 15, 12, 1):
        n = 1 << lg
        if n not in stmts:
            a, b, r = engine.synth_g1(300 + lg, n), engine.synth_g2(400 + lg, n), engine.synth_fr(500 + lg, n)
            v = engine.product_of_pairings_with_coeffs(a, b, r)
            rc, ep, ech = orc.sipp_prove(a, b, r, v); assert rc == 0
            stmts[n] = (a, b, r, v, ep, ech)
        a, b, r, v, ep, ech = stmts[n]
        p, ch, _ = engine.SIPP.prove_one_shot(a, b, r, v)
        assert np.array_equal(p, ep) and np.array_equal(ch, ech), n
        if lg == 15:                                   # a resident job between two one-shot calls owns its own buffers
            job = engine.SippJob(*stmts[1 << 12][:3])
            try:
                p2, ch2, _ = job.prove(stmts[1 << 12][3]); assert np.array_equal(p2, stmts[1 << 12][4])
                assert lib().ripp_release_scratch() == 0             # frees the cached buffers; the job keeps its own
                p2, ch2, _ = job.prove(stmts[1 << 12][3]); assert np.array_equal(p2, stmts[1 << 12][4])
            finally:
                job.close()


@pytest.mark.parametrize("n", [1, 2, 3, 33, 1000, (1 << 12) + 5, 1 << 16])
def test_msm_sort_and_two_stream_forms_vs_oracle(engine, orc, n):
    """Both digit sorts and the two-stream form of large host-slice MSMs at sizes the oracle finishes in seconds: the lane-per-term sort (k_msm_digits'
    atomics + k_msm_scatter; RIPP_MSM_LDS_SORT_MIN above n -- the default is the sort through LDS tiles, msm.hpp k_msm_hist_lds / k_msm_scatter_lds, which
    every other MSM test runs) and the call in two halves on two streams, the second half's bases uploading beside the first half's additions
    (engine.hip msm_impl; RIPP_MSM_CHUNK_MIN; default from 96 MB of bases) -- affine and projective inputs, skewed scalars (every term in one bucket
    per window), an identity among the bases and a zero scalar."""
    import os
    s = orc.gen_scalars(23, n); b1, b2 = orc.gen_g1(7, n), orc.gen_g2(8, n)
    if n >= 33:
        b1[5] = 0; b2[7] = 0; s[9] = 0
    e1, e2 = orc.g1_to_affine(orc.msm_g1_a(b1, s)).reshape(1, 12), (orc.g2_to_affine(orc.msm_g2_a(b2, s)).reshape(1, 24) if n <= 1 << 12 else None)
    same = np.repeat(orc.fr_array([0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % orc.R]), n, axis=0)
    es = orc.g1_to_affine(orc.msm_g1_a(b1, same)).reshape(1, 12)
    for env in ({"RIPP_MSM_LDS_SORT_MIN": "4000000000"}, {"RIPP_MSM_CHUNK_MIN": "1"}, {"RIPP_MSM_LDS_SORT_MIN": "4000000000", "RIPP_MSM_CHUNK_MIN": "1"}):
        os.environ.update(env)
        try:
            assert np.array_equal(engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.blind_g1(b1, 9), s)), e1), env
            assert np.array_equal(engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.to_jac_g1(b1), same)), es), env
            if e2 is not None:
                assert np.array_equal(engine.normalize_batch_g2(engine.MultiexponentiationInnerProductG2.inner_product(orc.blind_g2(b2, 9), s)), e2), env
        finally:
            for k in env: del os.environ[k]


@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 100003])
def test_scalar_inner_product(engine, orc, n):
    """ScalarInnerProduct (inner_products/src/lib.rs:144-166) against Python integers; length mismatch raises as the other products do."""
    l, r = orc.gen_scalars(5, n), orc.gen_scalars(6, n)
    exp = sum(orc.limbs_to_fr(a) * orc.limbs_to_fr(b) for a, b in zip(l[:2000], r[:2000])) % orc.R if n <= 2000 else None
    got = orc.limbs_to_fr(engine.ScalarInnerProduct.inner_product(l, r))
    if exp is not None:
        assert got == exp
    else:   # linearity: <l, r> = <l[:k], r[:k]> + <l[k:], r[k:]>
        k = 777
        assert got == (orc.limbs_to_fr(engine.ScalarInnerProduct.inner_product(l[:k], r[:k])) + orc.limbs_to_fr(engine.ScalarInnerProduct.inner_product(l[k:], r[k:]))) % orc.R
        assert orc.limbs_to_fr(engine.ScalarInnerProduct.inner_product(l[:k], r[:k])) == sum(orc.limbs_to_fr(a) * orc.limbs_to_fr(b) for a, b in zip(l[:k], r[:k])) % orc.R
    if n:
        with pytest.raises(engine.InnerProductError):
            engine.ScalarInnerProduct.inner_product(l, r[:-1])


def test_msm_adversarial_scalars(engine, orc):
    """all-zero, all-one, all-(r-1), few distinct values (bucket skew) -- SURVEY.md section 8d config 3."""
    n = 1 << 11
    b1 = orc.gen_g1(5, n)
    for vals in ([0] * n, [1] * n, [orc.R - 1] * n, [(i % 7) * 0x1234567890ABCDEF1234567 + 3 for i in range(n)]):
        s = orc.fr_array(vals)
        got = engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.blind_g1(b1, 3), s))
        assert np.array_equal(got, orc.g1_to_affine(orc.msm_g1_a(b1, s)).reshape(1, 12))


def test_msm_endomorphism_edge_scalars(engine, orc):
    """The device splits every scalar before the bucket sort (msm.hpp: G1 k = k1 + lambda k2, G2 base-|x| digits): scalars that sit on
    the boundaries of those decompositions (multiples and neighbours of lambda and of u^j, all-ones halves, r - 1) must come out as the
    oracle's plain 255-bit sums, one scalar at a time (n = 1) and all together."""
    lam, u = 0xac45a4010001a40200000000ffffffff, 0xd201000000010000
    vals = [0, 1, 2, orc.R - 1, orc.R - 2, lam - 1, lam, lam + 1, 2 * lam, orc.R - lam, lam * lam % orc.R, (lam - 1) * lam + lam - 1,
            u - 1, u, u + 1, u * u - 1, u * u, u * u + 1, u ** 3 - 1, u ** 3, u ** 3 + u - 1, (u - 1) * (1 + u + u * u + u ** 3) % orc.R,
            (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128, (1 << 254), (1 << 254) + (1 << 127)]
    vals = [v % orc.R for v in vals]
    n = len(vals)
    s = orc.fr_array(vals); b1, b2 = orc.gen_g1(15, n), orc.gen_g2(16, n)
    j1, j2 = orc.blind_g1(b1, 5), orc.blind_g2(b2, 6)
    for i in range(n):
        assert np.array_equal(engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(j1[i:i + 1], s[i:i + 1])),
                              orc.g1_to_affine(orc.msm_g1_a(b1[i:i + 1], s[i:i + 1])).reshape(1, 12)), hex(vals[i])
        assert np.array_equal(engine.normalize_batch_g2(engine.MultiexponentiationInnerProductG2.inner_product(j2[i:i + 1], s[i:i + 1])),
                              orc.g2_to_affine(orc.msm_g2_a(b2[i:i + 1], s[i:i + 1])).reshape(1, 24)), hex(vals[i])
    assert np.array_equal(engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(j1, s)), orc.g1_to_affine(orc.msm_g1_a(b1, s)).reshape(1, 12))
    assert np.array_equal(engine.normalize_batch_g2(engine.MultiexponentiationInnerProductG2.inner_product(j2, s)), orc.g2_to_affine(orc.msm_g2_a(b2, s)).reshape(1, 24))


@pytest.mark.parametrize("n", [1 << 12, 1 << 17])
def test_msm_repeated_bases_vs_oracle(engine, orc, n):
    """Equal bases with equal scalars land in the same bucket slot, so the gathered mixed additions of k_msm_slot_sum_q meet T = Q (doubling) and
    T = -Q (the identity) -- the exceptional cases its low-liveness formulas only DETECT: those slots are flagged and summed again with the
    complete addition law by k_msm_slot_sum_fix_vm (one wave per flagged slot on the field VM; four 16-lane groups sum every fourth term, then
    group 0 adds the other partial sums; k_msm_slot_sum_fix under RIPP_NO_VM).  Identities among the bases and zero scalars ride along.
    n = 2^17: slots of 32 terms (eight per group), hundreds of flagged slots per window."""
    b1, b2 = orc.gen_g1(25, n), orc.gen_g2(26, n)
    vals = [(i % 5) * 0x1F2E3D4C5B6A79881726354453627180 + 7 for i in range(n)]
    for k in range(0, n, 16):                                               # runs of identical terms, one negated term per run
        for t in range(1, 6): b1[k + t] = b1[k]; b2[k + t] = b2[k]; vals[k + t] = vals[k]
        b1[k + 6, :6] = b1[k, :6]; b1[k + 6, 6:] = orc.fp_to_limbs((orc.P - orc.limbs_to_fp(b1[k, 6:])) % orc.P); vals[k + 6] = vals[k]
    b1[9] = 0; b2[10] = 0; vals[11] = 0
    s = orc.fr_array([v % orc.R for v in vals])
    assert np.array_equal(engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.blind_g1(b1, 3), s)),
                          orc.g1_to_affine(orc.msm_g1_a(b1, s)).reshape(1, 12))
    assert np.array_equal(engine.normalize_batch_g2(engine.MultiexponentiationInnerProductG2.inner_product(orc.blind_g2(b2, 4), s)),
                          orc.g2_to_affine(orc.msm_g2_a(b2, s)).reshape(1, 24))


@pytest.mark.parametrize("lg", [15, 17, 18])
def test_msm_mid_sizes_vs_oracle(engine, orc, lg):
    """Sizes where the window width leaves a SHORT top window (c = lg - 6: 255 mod 9/11/12 = 3/2/3 bits, i.e. 7/3/7 buckets holding
    n/8 .. n/4 terms each): the hierarchical slot grouping of msm.hpp must reproduce the oracle's sums; plus a skewed set at the
    same size (every term in one bucket per window)."""
    n = 1 << lg
    s = engine.synth_fr(77, n); b1 = engine.synth_g1(31, n)
    got = engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.to_jac_g1(b1), s))
    assert np.array_equal(got, orc.g1_to_affine(orc.msm_g1_a(b1, s)).reshape(1, 12))
    if lg <= 17:
        b2 = engine.synth_g2(41, n)
        got = engine.normalize_batch_g2(engine.MultiexponentiationInnerProductG2.inner_product(orc.to_jac_g2(b2), s))
        assert np.array_equal(got, orc.g2_to_affine(orc.msm_g2_a(b2, s)).reshape(1, 24))
        same = np.repeat(orc.fr_array([0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % orc.R]), n, axis=0)
        got = engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.to_jac_g1(b1), same))
        assert np.array_equal(got, orc.g1_to_affine(orc.msm_g1_a(b1, same)).reshape(1, 12))


def test_folds_and_normalise_vs_oracle(engine, orc):
    n = 300; half = n // 2
    a, b = orc.gen_g1(31, n), orc.gen_g2(37, n)
    aj, bj = orc.blind_g1(a, 4), orc.blind_g2(b, 5); aj[7] = 0; bj[9] = 0
    assert np.array_equal(engine.normalize_batch_g1(aj), orc.normalize_g1(aj))
    assert np.array_equal(engine.normalize_batch_g2(bj), orc.normalize_g2(bj))
    for s in (orc.gen_scalars(8, 1)[0], orc.fr_array([2**128 - 1])[0], orc.fr_array([1])[0], orc.fr_array([0])[0]):
        assert np.array_equal(engine.fold_g1_affine(a[half:], a[:half], s), orc.fold_g1_a(a[half:], a[:half], s))
        assert np.array_equal(engine.fold_g2_affine(b[half:], b[:half], s), orc.fold_g2_a(b[half:], b[:half], s))
    s = orc.gen_scalars(8, 1)[0]
    assert np.array_equal(engine.normalize_batch_g1(engine.fold_g1(aj[half:], aj[:half], s)), orc.normalize_g1(orc.fold_g1_j(aj[half:], aj[:half], s)))
    assert np.array_equal(engine.normalize_batch_g2(engine.fold_g2(bj[half:], bj[:half], s)), orc.normalize_g2(orc.fold_g2_j(bj[half:], bj[:half], s)))
    r = orc.gen_scalars(9, n)
    assert np.array_equal(engine.scale_g1_affine(a, r), orc.scale_g1_a(a, r))


@pytest.mark.parametrize("n", [1, 2, 32, 1 << 10])
def test_sipp_prove_vs_oracle(engine, orc, n):
    """Whole proof byte-identical to the CPU path; n = 32 is the reference's own test size, 2^10 is config 1."""
    a, b, r = orc.gen_g1(123, n), orc.gen_g2(456, n), orc.gen_scalars(7, n)
    value = orc.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(engine.product_of_pairings_with_coeffs(a, b, r), value)
    proof, ch, _ = engine.SIPP.prove_with_stats(a, b, r, value)
    rc, eproof, ech = orc.sipp_prove(a, b, r, value)
    assert rc == 0 and np.array_equal(proof, eproof) and np.array_equal(ch, ech)
    if n >= 2:
        assert engine.SIPP.verify(a, b, r, value, proof)
        bad = proof.copy(); bad[1] = proof[0]
        assert not engine.SIPP.verify(a, b, r, value, bad)


@pytest.mark.parametrize("n", [8, 64, 4096, 32768])
def test_sipp_degenerate_statement_vs_oracle(engine, orc, n):
    """(n = 32768: round 0 folds G2 with the carry-free 4-lane GLS kernel, whose flagged lanes go to k_fold_g2_gls_split_fix.)
    Zero coefficients (r_i = 0 -> the scaled a_i is the identity), identities on both sides, repeated and negated points (folds
    meet P + P and P - P), on the scalar kernels (n = 4096 in its first rounds) and on the VM kernels: the proof must still equal the
    oracle's byte for byte and verify."""
    a, b, r = orc.gen_g1(70, n), orc.gen_g2(80, n), orc.gen_scalars(9, n)
    r[1] = 0; r[n // 2] = 0
    a[2] = 0; b[3] = 0; a[n - 1] = 0; b[n - 1] = 0
    a[5] = a[4]; b[5] = b[4]; r[5] = r[4]                              # identical neighbours
    q = (4 + n // 2) % n
    a[q] = a[4]; b[q] = b[4]; r[q] = r[4]                              # identical partners of a halving round: x*P + P patterns
    a[6, 6:] = orc.fp_to_limbs((orc.P - orc.limbs_to_fp(a[7, 6:])) % orc.P); a[6, :6] = a[7, :6]   # a_6 = -a_7
    value = engine.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(value, orc.product_of_pairings_with_coeffs(a, b, r))
    proof = engine.SIPP.prove(a, b, r, value)
    rc, eproof, _ = orc.sipp_prove(a, b, r, value)
    assert rc == 0 and np.array_equal(proof, eproof)
    assert engine.SIPP.verify(a, b, r, value, proof)


def test_sipp_prove_2p17_vs_oracle_with_precomputed_round0(engine, orc):
    """n = 2^17 is the smallest statement for which round 0 uses the fold with a precomputed second base (2^64 a_r, 2^32 b_r prepared
    while the statement hash runs) and, by default, over tables of the odd multiples {1,3,5,7} of both bases with width-4 wNAF digit
    strings: the whole proof must still equal the oracle's byte for byte, with the tables (default) and with RIPP_NO_FOLD_TABLES=1
    (the two-base NAF kernels; the library reads the switch at every call)."""
    import os
    n = 1 << 17
    a, b, r = engine.synth_g1(1000, n), engine.synth_g2(2000, n), engine.synth_fr(0, n)
    # degenerate table rows: identities in the right halves (every multiple is the identity), a zero coefficient, a right-half element equal
    # to its left partner (x * P + P), a negated pair
    h = n // 2
    a[h + 3] = 0; b[h + 5] = 0; b[7] = 0; r[h + 9] = 0
    a[h + 11] = a[11]; b[h + 11] = b[11]; r[h + 11] = r[11]
    b[h + 13, :12] = b[13, :12]
    for k in (12, 18):                                                     # b_(h+13) = -b_13
        b[h + 13, k:k + 6] = orc.fp_to_limbs((orc.P - orc.limbs_to_fp(b[13, k:k + 6])) % orc.P)
    value = engine.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, _ = orc.sipp_prove(a, b, r, value)
    assert rc == 0
    assert np.array_equal(engine.SIPP.prove(a, b, r, value), eproof)
    # every implementation of the same folds stays pinned to the oracle (the library reads the switches at every call):
    #   RIPP_NO_FOLD_TABLES  two-base NAF kernels            RIPP_NO_XSCALE   G2 folds on the plain vector with the full-width x^-1
    #   RIPP_FQ_MIN=4096     the carry-free (14 x 28-bit) fold kernels and k_line_products_q already at this size -- with the degenerate
    #                        rows above, so their exceptional-case fallback runs         RIPP_NO_FQ   the 12 x 32-bit kernels everywhere
    #   RIPP_LP_FQ_MIN / RIPP_ML_FQ_MIN = 2^32 - 1: the 12 x 32-bit pairing kernels (k_line_products, k_miller_lines) under the carry-free folds, one at a time
    #   RIPP_LOOK_EIGHTHS    the hash-window look-ahead cut to a fraction of an item: (1,l) in full + 3/8 of (1,r); 5/8 of (1,l) alone
    big = str((1 << 32) - 1)
    for env in ({"RIPP_NO_FOLD_TABLES": "1"}, {"RIPP_NO_XSCALE": "1"}, {"RIPP_FQ_MIN": "4096", "RIPP_LP_FQ_MIN": "1"},
                {"RIPP_FQ_MIN": "4096", "RIPP_NO_XSCALE": "1"}, {"RIPP_NO_FQ": "1"}, {"RIPP_LP_FQ_MIN": big}, {"RIPP_ML_FQ_MIN": big},
                {"RIPP_NO_LP_KARA": "1"}, {"RIPP_NO_LP_KARA": "1", "RIPP_LOOK_EIGHTHS": "48"},     # stage 2 with the six-product sums (k_line_products_q) instead of the Karatsuba form (k_line_products_k, the default on BLS12-381)
                {"RIPP_LOOK_EIGHTHS": "11"}, {"RIPP_LOOK_EIGHTHS": "5"}, {"RIPP_LOOK_EIGHTHS": "20", "RIPP_NO_XSCALE": "1"},
                {"RIPP_LOOK_EIGHTHS": "48"}, {"RIPP_LOOK_EIGHTHS": "48", "RIPP_NO_SHARE": "1"}, {"RIPP_LOOK_EIGHTHS": "24", "RIPP_ML_FQ_MIN": big},      # shared G2 chains (fq_miller.hpp) / every product its own chain
                # >= 16 eighths: both values of round 1 come from the look-ahead, so rounds 0 and 1 fold in ONE pass over three-quarter tables (job_fold_fused:
                # the degenerate rows above sit in the quarters A2 / B1 / B2 and reach its fix-up kernels); RIPP_NO_FUSE: the two folds one after the other;
                # 16 + RIPP_NO_XSCALE: tables on the high half, no fusion
                {"RIPP_LOOK_EIGHTHS": "16"}, {"RIPP_LOOK_EIGHTHS": "16", "RIPP_NO_FUSE": "1"}, {"RIPP_LOOK_EIGHTHS": "16", "RIPP_NO_XSCALE": "1"}, {"RIPP_LOOK_EIGHTHS": "16", "RIPP_FQ_MIN": "4096"},
                # RIPP_FUSE_TABLES: the three-quarter tables although x1 will NOT be known with x0 (no / half a value of round 1 from the look-ahead): round 0
                # folds alone over them (element offset q on G1), round 1 with its in-round tables -- what a proof does when the hash beats the look-ahead
                # RIPP_NO_PREBUILD: the in-round G2 tables of rounds >= 1 after the challenge instead of in the host phase before it (job_prebuild_g2_tables)
                {"RIPP_NO_PREBUILD": "1"}, {"RIPP_NO_PREBUILD": "1", "RIPP_LOOK_EIGHTHS": "16", "RIPP_NO_FUSE": "1"}, {"RIPP_NO_PREBUILD": "1", "RIPP_NO_XSCALE": "1"},
                {"RIPP_FUSE_TABLES": "1", "RIPP_LOOK_EIGHTHS": "12"}, {"RIPP_FUSE_TABLES": "1", "RIPP_LOOK_EIGHTHS": "8"}, {"RIPP_FUSE_TABLES": "1", "RIPP_LOOK_EIGHTHS": "8", "RIPP_NO_FQ": "1"}):
        os.environ.update(env)
        try:
            assert np.array_equal(engine.SIPP.prove(a, b, r, value), eproof), env
        finally:
            for k in env: del os.environ[k]


def test_sipp_rejects_non_power_of_two(engine, orc):
    a, b, r = orc.gen_g1(1, 24), orc.gen_g2(1, 24), orc.gen_scalars(1, 24)
    with pytest.raises(AssertionError):
        engine.SIPP.prove(a, b, r, np.zeros(72, dtype=np.uint64))


def test_pairing_product_config2_size(engine, orc):
    """Config 2: PairingInnerProduct at n = 2^16, checked (a) bit-exact against the oracle on a 2^12 slice,
    (b) by multiplicativity over a split of the full vector, (c) by bilinearity e(2A, B) = e(A, B)^2 on the full size."""
    n = 1 << 16
    a, b = engine.synth_g1(1000, n), engine.synth_g2(2000, n)
    full = engine.product_of_pairings(a, b)
    lo, hi = engine.product_of_pairings(a[: n // 2], b[: n // 2]), engine.product_of_pairings(a[n // 2:], b[n // 2:])
    assert np.array_equal(engine.gt_mul(lo, hi), full)
    m = 1 << 12
    assert np.array_equal(engine.product_of_pairings(a[:m], b[:m]), orc.pairing_product_a(a[:m], b[:m]))
    two = orc.fr_array([2])[0]
    a2 = engine.fold_g1_affine(a, np.zeros_like(a), two)                        # 2*A_i + infinity
    assert np.array_equal(engine.product_of_pairings(a2, b), engine.gt_mul(full, full))


def test_sipp_full_size_round_trip(engine):
    """Config 4 shape on one GPU at 2^16: prove -> verify accepts, tampered proof rejected (size-independent property)."""
    n = 1 << 16
    a, b, r = engine.synth_g1(1000, n), engine.synth_g2(2000, n), engine.synth_fr(0, n)
    value = engine.product_of_pairings_with_coeffs(a, b, r)
    proof = engine.SIPP.prove(a, b, r, value)
    assert engine.SIPP.verify(a, b, r, value, proof)
    bad = proof.copy(); bad[5] = proof[4]
    assert not engine.SIPP.verify(a, b, r, value, bad)


@pytest.mark.parametrize("switch", ["RIPP_NO_VM", "RIPP_NO_PRECOMPUTE", "RIPP_NO_MSM_GLV"])
def test_scalar_kernels_without_vm(engine, orc, switch):
    """The latency-form (lane-parallel VM) kernels serve small launches, and the folds use second bases precomputed in the host phase;
    RIPP_NO_VM=1 forces the scalar kernels, RIPP_NO_PRECOMPUTE=1 the one-base folds and RIPP_NO_MSM_GLV=1 the plain 255-bit Pippenger
    windows, so every implementation of the same path stays pinned to the oracle."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np, orclib as o, ripp_amd as R
        R.init(0)
        n = 128
        a, b, r = o.gen_g1(123, n), o.gen_g2(456, n), o.gen_scalars(7, n)
        v = o.product_of_pairings_with_coeffs(a, b, r)
        proof = R.SIPP.prove(a, b, r, v)
        rc, eproof, _ = o.sipp_prove(a, b, r, v)
        assert rc == 0 and np.array_equal(proof, eproof)
        m = 1000
        s, b1, b2 = o.gen_scalars(21, m), o.gen_g1(5, m), o.gen_g2(6, m)
        assert np.array_equal(R.normalize_batch_g1(R.MultiexponentiationInnerProductG1.inner_product(o.blind_g1(b1, 9), s)), o.g1_to_affine(o.msm_g1_a(b1, s)).reshape(1, 12))
        assert np.array_equal(R.normalize_batch_g2(R.MultiexponentiationInnerProductG2.inner_product(o.blind_g2(b2, 9), s)), o.g2_to_affine(o.msm_g2_a(b2, s)).reshape(1, 24))
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **{switch: "1"})
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_exceptional_group_law_cases_fall_back(engine, orc):
    """(0, 2) is a point of order 3 on y^2 = x^3 + 4 (outside G1).  Multiplying it by any scalar walks through
    T = +-Q and T = infinity, which the VM's incomplete addition formulas flag; the engine must then redo the fold with
    the complete scalar kernels and still match the oracle bit for bit (the reference's group law is complete)."""
    n = 8
    a, b, r = orc.gen_g1(123, n), orc.gen_g2(456, n), orc.gen_scalars(7, n)
    a[5, :6] = orc.fp_to_limbs(0); a[5, 6:] = orc.fp_to_limbs(2)
    a[2] = 0                                                      # and a point at infinity
    value = orc.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(engine.product_of_pairings_with_coeffs(a, b, r), value)
    proof = engine.SIPP.prove(a, b, r, value)
    rc, eproof, _ = orc.sipp_prove(a, b, r, value)
    assert rc == 0 and np.array_equal(proof, eproof)


@pytest.mark.parametrize("n", [1, 2, 8, 64, 1 << 10])
def test_gipa_tipp_prove_vs_oracle(engine, orc, n):
    """GIPA::_prove round body (ip_proofs/src/gipa.rs:196-297) for the TIPP instantiation; n = 8 is the reference's
    TEST_SIZE (gipa.rs:468).  Commitments, challenges and the base case must equal the oracle's, and the oracle's
    GIPA verifier (gipa.rs:135-160) must accept the GPU-made proof."""
    m_a, m_b = orc.blind_g1(orc.gen_g1(11, n), 1), orc.blind_g2(orc.gen_g2(22, n), 2)
    ck_a, ck_b = orc.blind_g2(orc.gen_g2(33, n), 3), orc.blind_g1(orc.gen_g1(44, n), 4)
    proof, aux, raw = engine.GIPA_TIPP.prove_with_aux(m_a, m_b, ck_a, ck_b)
    rc, steps, tr, ba, bb, ka, kb = orc.gipa_tipp_prove(m_a, m_b, ck_a, ck_b)
    assert rc == 0
    assert np.array_equal(raw["round_order_steps"], steps) and np.array_equal(raw["round_order_transcript"], tr)
    assert np.array_equal(engine.normalize_batch_g1(proof["r_base"][0]), orc.g1_to_affine(ba).reshape(1, 12))
    assert np.array_equal(engine.normalize_batch_g2(proof["r_base"][1]), orc.g2_to_affine(bb).reshape(1, 24))
    assert np.array_equal(engine.normalize_batch_g2(aux["ck_base"][0]), orc.g2_to_affine(ka).reshape(1, 24))
    assert np.array_equal(engine.normalize_batch_g1(aux["ck_base"][1]), orc.g1_to_affine(kb).reshape(1, 12))
    if n >= 2:
        com = [engine.AFGHOCommitmentG1.commit(ck_a, m_a), engine.AFGHOCommitmentG2.commit(ck_b, m_b), engine.PairingInnerProduct.inner_product(m_a, m_b)]
        assert orc.gipa_tipp_verify(ck_a, ck_b, com, raw["round_order_steps"], proof["r_base"][0], proof["r_base"][1]) == 1
        bad = raw["round_order_steps"].copy(); bad[0] = bad[1]
        assert orc.gipa_tipp_verify(ck_a, ck_b, com, bad, proof["r_base"][0], proof["r_base"][1]) == 0
        # the product's GIPA::verify (gipa.rs:135-160), final keys by MSM on the device
        assert engine.GIPA_TIPP.verify(ck_a, ck_b, com, raw["round_order_steps"], proof["r_base"])
        assert not engine.GIPA_TIPP.verify(ck_a, ck_b, com, bad, proof["r_base"])
        assert not engine.GIPA_TIPP.verify(ck_a, ck_b, com, raw["round_order_steps"], (proof["r_base"][0], aux["ck_base"][0]))


def test_c_abi_demo_program(engine, tmp_path):
    """examples/c_abi_demo.c: a C99 program using only include/ripp_hip.h proves and verifies a 2^10 statement, rejects a tampered proof
    and sees the reference's length error as a status code."""
    import subprocess
    from test_abi_cpu import _build_c_demo
    p = subprocess.run([_build_c_demo(tmp_path), "10"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "verify accepts" in p.stdout and "tampered proof rejected" in p.stdout


def test_per_element_scaling_small_vectors(engine, orc):
    """a_i <- r_i a_i (sipp/src/lib.rs:61-65) for few elements runs on the field VM (one group per element, GLV halves as signed base-16
    digits over a table of the multiples 1..8): every digit value and sign, zero scalars, the identity and full-width scalars, checked
    through product_of_pairings_with_coeffs against the oracle; and the same sizes with the throughput kernel (RIPP_VM_SCALE_MAX=0)."""
    import os
    n = 40
    a, b = orc.gen_g1(321, n), orc.gen_g2(654, n)
    small = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 0x21, 0x78, 0x87, 0x88, 0x89, 0xff, 0x100, (1 << 64) + 1, (1 << 127) + 3, (1 << 128) - 1]
    r = orc.gen_scalars(99, n)
    r[:len(small)] = orc.fr_array(small)
    a[30] = 0                                                     # the identity, with a full-width scalar
    for m in (1, 2, 7, n):
        exp = orc.product_of_pairings_with_coeffs(a[:m], b[:m], r[:m])
        assert np.array_equal(engine.product_of_pairings_with_coeffs(a[:m], b[:m], r[:m]), exp), m
        os.environ["RIPP_VM_SCALE_MAX"] = "0"
        try:
            assert np.array_equal(engine.product_of_pairings_with_coeffs(a[:m], b[:m], r[:m]), exp), m
        finally:
            del os.environ["RIPP_VM_SCALE_MAX"]
    for k in range(len(small)):                                    # one element at a time: each digit pattern on its own
        assert np.array_equal(engine.product_of_pairings_with_coeffs(a[k:k + 1], b[k:k + 1], r[k:k + 1]), orc.product_of_pairings_with_coeffs(a[k:k + 1], b[k:k + 1], r[k:k + 1])), hex(small[k])


def test_per_element_scaling_mid_size_on_the_vm(engine, orc):
    """4 K < n <= 16 K elements are still scaled by k_vm_scale_g1 (two waves per SIMD of 16-lane groups; a ragged last block): the direct
    product with coefficients (sipp/src/lib.rs:184-219) at n = 2^13 + 37 against the oracle, and against the throughput kernel."""
    import os
    n = (1 << 13) + 37
    a, b, r = orc.gen_g1(77, n), orc.gen_g2(88, n), orc.gen_scalars(5, n)
    a[n - 1] = 0                                                   # the identity in the ragged tail
    exp = orc.product_of_pairings_with_coeffs(a, b, r)
    assert np.array_equal(engine.product_of_pairings_with_coeffs(a, b, r), exp)
    os.environ["RIPP_VM_SCALE_MAX"] = "0"
    try:
        assert np.array_equal(engine.product_of_pairings_with_coeffs(a, b, r), exp)
    finally:
        del os.environ["RIPP_VM_SCALE_MAX"]


@pytest.mark.parametrize("n", [2, 4, 8, 64, 1 << 12])
def test_pipelined_tail_rounds_and_job_reuse(engine, orc, n):
    """Rounds of <= 2^11 elements take their (z_l, z_r) from eight quarter products of the PREVIOUS round's unfolded vectors and the previous
    challenge (engine.hip job_tail_enqueue / job_tail_values), their folds are not waited for and the last fold is not computed
    (sipp/src/lib.rs:69-104): same proof bytes as the oracle, as the round-by-round form (RIPP_TAIL_PIPE_MAX=0), and again when the same
    resident job is proved a second and a third time (nothing prepared for one proof may leak into the next)."""
    import os
    a, b, r = orc.gen_g1(31, n), orc.gen_g2(32, n), orc.gen_scalars(33, n)
    v = orc.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, v)
    assert rc == 0
    job = engine.SippJob(a, b, r)
    try:
        for rep in range(3):
            if rep == 1: os.environ["RIPP_TAIL_PIPE_MAX"] = "0"
            try:
                proof, ch, _ = job.prove(v)
            finally:
                os.environ.pop("RIPP_TAIL_PIPE_MAX", None)
            assert np.array_equal(proof, eproof), (n, rep)
            assert np.array_equal(ch, ech), (n, rep)
    finally:
        job.close()


@pytest.mark.parametrize("n,eighths", [(1 << 13, 12), (1 << 13, 3), (1 << 13, 21), (1 << 14, 38), (1 << 12, 47)])
def test_partial_lookahead_items_vs_oracle(engine, orc, n, eighths):
    """A look-ahead item cut to f/8 of every block's pairs (what is left of the hash window after the whole items): pairs [0, f q / 8) of that
    round's product come from the round-0 blocks, the rest from the folded vectors on the device, and the two GT values are multiplied.
    RIPP_LOOK_EIGHTHS = 8 k + f: k whole items and f/8 of the next.  Same proof bytes as the oracle."""
    import os
    a, b, r = orc.gen_g1(51, n), orc.gen_g2(52, n), orc.gen_scalars(53, n)
    a[3] = 0; b[n // 2 + 70] = 0; a[n // 4 + 65] = 0
    v = orc.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, v)
    assert rc == 0
    os.environ["RIPP_LOOK_EIGHTHS"] = str(eighths)
    try:
        proof, ch, st = engine.SippJob(a, b, r).prove(v)
    finally:
        del os.environ["RIPP_LOOK_EIGHTHS"]
    assert np.array_equal(proof, eproof) and np.array_equal(ch, ech)
    assert st["look_items"] == (eighths + 7) // 8 and st["look_pairs"] > 0


@pytest.mark.parametrize("n,items", [(8, 6), (64, 6), (1 << 12, 2), (1 << 13, 6), (1 << 15, 3)])
def test_lookahead_rounds_vs_oracle(engine, orc, n, items):
    """Rounds 1..3 taken from the look-ahead (engine.hip job_lookahead: 4^R block products of the ROUND-0 vectors grouped into 3^R values,
    reduced level by level as the challenges arrive; folds of those rounds are not waited for) instead of from the folded vectors:
    RIPP_LOOK_ITEMS forces the plan a multi-GPU proof chooses at n = 2^20 onto small statements.  Same proof bytes as the oracle, for every
    prefix of the item order (1,l) (1,r) (2,l) (2,r) (3,l) (3,r), and again on the reused job without the look-ahead."""
    import os
    a, b, r = orc.gen_g1(41, n), orc.gen_g2(42, n), orc.gen_scalars(43, n)
    a[n // 2 + 1] = 0; b[n // 4] = 0                                 # identities inside the blocks
    v = orc.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, v)
    assert rc == 0
    job = engine.SippJob(a, b, r)
    try:
        for k in sorted({items, max(items - 1, 0), 1, 0}, reverse=True):
            os.environ["RIPP_LOOK_ITEMS"] = str(k)
            try:
                proof, ch, st = job.prove(v)
            finally:
                os.environ.pop("RIPP_LOOK_ITEMS", None)
            assert np.array_equal(proof, eproof) and np.array_equal(ch, ech), (n, k)
            lg = n.bit_length() - 1
            assert st["look_items"] == min(k, 2 * max(min(lg - 1, 3), 0)), (n, k, st["look_items"])
    finally:
        job.close()


def test_configure_api_selects_the_same_forms_as_the_environment(engine, orc):
    """ripp_configure (include/ripp_hip.h: ripp_config) is the API for what the RIPP_* variables select: a forced look-ahead plan, the
    round-by-round tail, the single-lane kernels -- each gives the oracle's proof; ripp_config_get shows what the next call runs with
    (defaults < ripp_configure < environment), and configure() returns to the defaults."""
    import os
    n = 1 << 12
    a, b, r = orc.gen_g1(61, n), orc.gen_g2(62, n), orc.gen_scalars(63, n)
    v = orc.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, v)
    assert rc == 0
    job = engine.SippJob(a, b, r)
    try:
        d = engine.config_get()
        assert d.look_eighths == -1 and d.tail_pipe_max == 1 << 11 and d.no_vm == 0
        engine.configure(look_eighths=29, tail_pipe_max=0)
        g = engine.config_get()
        assert g.look_eighths == 29 and g.tail_pipe_max == 0
        proof, ch, st = job.prove(v)
        assert np.array_equal(proof, eproof) and np.array_equal(ch, ech) and st["look_items"] == 4
        os.environ["RIPP_LOOK_EIGHTHS"] = "8"                     # the environment overrides the configured value
        try:
            assert engine.config_get().look_eighths == 8
            proof, ch, st = job.prove(v)
        finally:
            del os.environ["RIPP_LOOK_EIGHTHS"]
        assert np.array_equal(proof, eproof) and st["look_items"] == 1
        engine.configure(no_vm=1, no_precompute=1)
        assert engine.config_get().no_vm == 1 and engine.config_get().look_eighths == -1
        proof, ch, st = job.prove(v)
        assert np.array_equal(proof, eproof) and np.array_equal(ch, ech) and st["look_items"] == 0
    finally:
        engine.configure()
        job.close()
    d = engine.config_get()
    assert d.look_eighths == -1 and d.tail_pipe_max == 1 << 11 and d.no_vm == 0 and d.no_precompute == 0


@pytest.mark.parametrize("n,eighths", [(4, 16), (16, 48), (256, 48), (1 << 12, 29), (1 << 13, 16), (1 << 14, 45)])
def test_shared_g2_chains_vs_oracle(engine, orc, n, eighths):
    """Products over the SAME Q vector share one G2 chain in the carry-free stage-1 kernel (fq_miller.hpp: ChainSets, up to four P's per lane):
    round 0 is evaluated together with look-ahead item (1,l) -- B0 and B2 each meet three A blocks (engine.hip job_round0_shared) -- and the later
    items pair every B block with 2^R A blocks.  vm_lines_max = 0 sends even these small launches to the throughput kernel.  Identities among the
    P's and Q's of every quarter (the per-P skip masks).  Same proof bytes as the oracle, with and without sharing; the chain count shows the sharing."""
    a, b, r = orc.gen_g1(71, n), orc.gen_g2(72, n), orc.gen_scalars(73, n)
    if n >= 16:
        q = n // 4
        a[q + 1] = 0; a[3 * q + 2] = 0; a[2 * q] = 0; a[0] = 0              # identities in A1, A3, A2, A0
        b[1] = 0; b[2 * q + 3] = 0; b[q + 2] = 0; b[3 * q] = 0                # .. and in B0, B2, B1, B3
        b[q + 1] = 0                                                          # P and Q both the identity in one pair of (1,r)'s blocks
    v = orc.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, v)
    assert rc == 0
    job = engine.SippJob(a, b, r)
    try:
        engine.configure(vm_lines_max=0, look_eighths=eighths)
        proof, ch, st = job.prove(v)
        assert np.array_equal(proof, eproof) and np.array_equal(ch, ech)
        assert 0 < st["chains_lines"] < st["pairs_lines"], (st["chains_lines"], st["pairs_lines"])
        shared = st["chains_lines"]
        engine.configure(vm_lines_max=0, look_eighths=eighths, no_share=1)
        proof, ch, st = job.prove(v)
        assert np.array_equal(proof, eproof) and np.array_equal(ch, ech)
        assert st["chains_lines"] == st["pairs_lines"] > shared
    finally:
        engine.configure()
        job.close()


def test_in_process_multi_device_dispatch_on_virtual_devices(engine, orc):
    """ripp_config.n_devices (include/ripp_hip.h): the stateless trait calls on host slices -- PairingInnerProduct, the pairing products of the sipp crate, both
    MultiexponentiationInnerProducts -- cut their index range into D parts, one per device and host thread of THIS process, and combine the partial results on
    the host (one final exponentiation).  On a one-GPU box the D slots are mapped onto the bound device (RIPP_VIRTUAL_DEVICES); the values must be the
    oracle's, ripp_device_slots_used tells that the split really happened, and small inputs are not split."""
    import ctypes
    import os
    from ripp_amd._lib import lib
    n = 1 << 15
    a, b, r = orc.gen_g1(71, n), orc.gen_g2(72, n), orc.gen_scalars(73, n)
    aj, bj = orc.blind_g1(a, 5), orc.blind_g2(b, 6)
    a[17] = 0; b[n // 2 + 3] = 0                                     # identities in two different parts
    rc, exp_j = orc.pairing_product_j(aj, bj)
    assert rc == 0
    exp_a = orc.product_of_pairings_with_coeffs(a, b, np.ascontiguousarray(np.repeat(orc.fr_array([1]), n, axis=0)))      # (affine inputs with identities)
    exp_m1, exp_m2 = orc.g1_to_affine(orc.msm_g1_a(a, r)).reshape(1, 12), orc.g2_to_affine(orc.msm_g2_a(b, r)).reshape(1, 24)
    L = lib(); L.ripp_device_slots_used.restype = ctypes.c_int32
    for D in (1, 2, 3, 4):
        os.environ["RIPP_VIRTUAL_DEVICES"] = str(D)
        try:
            assert np.array_equal(engine.PairingInnerProduct.inner_product(aj, bj), exp_j) and L.ripp_device_slots_used() == D
            assert np.array_equal(engine.product_of_pairings(a, b), exp_a) and L.ripp_device_slots_used() == D
            assert np.array_equal(engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.to_jac_g1(a), r)), exp_m1) and L.ripp_device_slots_used() == 1   # 2^15 terms: below 2 x 32 768
            assert np.array_equal(engine.normalize_batch_g2(engine.MultiexponentiationInnerProductG2.inner_product(orc.to_jac_g2(b), r)), exp_m2)
            # a short vector is not split; the reference's length error comes before any dispatch
            assert np.array_equal(engine.PairingInnerProduct.inner_product(aj[:1000], bj[:1000]), orc.pairing_product_j(aj[:1000], bj[:1000])[1]) and L.ripp_device_slots_used() == 1
            with pytest.raises(engine.InnerProductError):
                engine.PairingInnerProduct.inner_product(aj[:10], bj[:9])
        finally:
            del os.environ["RIPP_VIRTUAL_DEVICES"]
    # the configured form (no environment): n_devices beyond the visible devices is an argument error naming the device, not a crash
    engine.configure(n_devices=64)
    try:
        with pytest.raises(ValueError, match="does not exist"):
            engine.PairingInnerProduct.inner_product(aj, bj)
    finally:
        engine.configure()
    n2 = 1 << 17                                                     # an MSM large enough to be cut in four
    a2, r2 = engine.synth_g1(500, n2), engine.synth_fr(3, n2)
    os.environ["RIPP_VIRTUAL_DEVICES"] = "4"
    try:
        got = engine.normalize_batch_g1(engine.MultiexponentiationInnerProductG1.inner_product(orc.to_jac_g1(a2), r2))
        assert L.ripp_device_slots_used() == 4
    finally:
        del os.environ["RIPP_VIRTUAL_DEVICES"]
    assert np.array_equal(got, orc.g1_to_affine(orc.msm_g1_a(a2, r2)).reshape(1, 12))


@pytest.mark.parametrize("n,D", [(1 << 12, 2), (1 << 12, 4), (1 << 13, 8), (1 << 12, 3)])
def test_gipa_tipp_prove_on_in_process_ranks(engine, orc, n, D):
    """The fused GIPA prover with ripp_config.n_devices = D: D in-process ranks (one host thread and engine per device, vectors sharded by index residue, the per-round
    exchange a copy through host memory -- no communicator) must give the single-device proof, i.e. the oracle's: commitments, challenges, base case.  On a one-GPU
    box the ranks share the bound device (RIPP_VIRTUAL_DEVICES); D = 3 rounds down to two ranks; a vector too short to be worth a device stays on one."""
    import ctypes
    import os
    from ripp_amd._lib import lib
    m_a, m_b = orc.blind_g1(orc.gen_g1(11, n), 1), orc.blind_g2(orc.gen_g2(22, n), 2)
    ck_a, ck_b = orc.blind_g2(orc.gen_g2(33, n), 3), orc.blind_g1(orc.gen_g1(44, n), 4)
    rc, steps, tr, ba, bb, ka, kb = orc.gipa_tipp_prove(m_a, m_b, ck_a, ck_b)
    assert rc == 0
    L = lib(); L.ripp_device_slots_used.restype = ctypes.c_int32
    os.environ["RIPP_VIRTUAL_DEVICES"] = str(D)
    try:
        proof, aux, raw = engine.GIPA_TIPP.prove_with_aux(m_a, m_b, ck_a, ck_b)
        assert L.ripp_device_slots_used() == (2 if D == 3 else min(D, n // 1024))
        small = engine.GIPA_TIPP.prove_with_aux(m_a[:64], m_b[:64], ck_a[:64], ck_b[:64])[2]
        assert L.ripp_device_slots_used() == 1
    finally:
        del os.environ["RIPP_VIRTUAL_DEVICES"]
    assert np.array_equal(raw["round_order_steps"], steps) and np.array_equal(raw["round_order_transcript"], tr)
    assert np.array_equal(engine.normalize_batch_g1(proof["r_base"][0]), orc.g1_to_affine(ba).reshape(1, 12))
    assert np.array_equal(engine.normalize_batch_g2(proof["r_base"][1]), orc.g2_to_affine(bb).reshape(1, 24))
    assert np.array_equal(engine.normalize_batch_g2(aux["ck_base"][0]), orc.g2_to_affine(ka).reshape(1, 24))
    assert np.array_equal(engine.normalize_batch_g1(aux["ck_base"][1]), orc.g1_to_affine(kb).reshape(1, 12))
    assert np.array_equal(small["round_order_transcript"], orc.gipa_tipp_prove(m_a[:64], m_b[:64], ck_a[:64], ck_b[:64])[2])


@pytest.mark.parametrize("n,D", [(1 << 15, 2), (1 << 16, 4)])
def test_sipp_prove_on_in_process_ranks(engine, orc, n, D):
    """SIPP::prove with ripp_config.n_devices = D: D in-process ranks (one host thread, engine and job per device; the statement sharded by index residue, rank 0 hashing
    the caller's whole statement, the per-round exchange a copy through host memory) must give the single-device proof -- all 2 log2 n GT values and every challenge
    equal to the oracle's.  The D slots share the bound device here (RIPP_VIRTUAL_DEVICES); a statement below 2^14 elements per device is not sharded."""
    import ctypes
    import os
    from ripp_amd._lib import lib
    a, b, r = engine.synth_g1(31, n), engine.synth_g2(32, n), engine.synth_fr(33, n)
    a[5] = 0; b[n - 3] = 0; r[n // 2] = 0                             # identities and a zero coefficient on different ranks
    value = engine.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, value)
    assert rc == 0
    L = lib(); L.ripp_device_slots_used.restype = ctypes.c_int32
    os.environ["RIPP_VIRTUAL_DEVICES"] = str(D)
    try:
        proof, ch, st = engine.SIPP.prove_one_shot(a, b, r, value)
        assert L.ripp_device_slots_used() == D
        proof2, ch2, _ = engine.SIPP.prove_one_shot(a, b, r, value)      # again: the ranks' parked job buffers are adopted
        small = engine.SIPP.prove_one_shot(a[:4096], b[:4096], r[:4096], engine.product_of_pairings_with_coeffs(a[:4096], b[:4096], r[:4096]))[0]
        assert L.ripp_device_slots_used() == 1 and small.shape == (24, 72)
    finally:
        del os.environ["RIPP_VIRTUAL_DEVICES"]
    assert np.array_equal(proof, eproof) and np.array_equal(ch, ech)
    assert np.array_equal(proof2, eproof) and np.array_equal(ch2, ech)
    assert np.array_equal(engine.SIPP.prove_one_shot(a, b, r, value)[0], eproof)      # and back on one device
