/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ripp_oracle.c header).
 *
 * CPU restatement of the callers that sit directly above the hot path:
 *   TIPA  = GIPA + KZG openings of the final commitment keys      ip_proofs/src/tipa/mod.rs:137-301
 *   TIPAWithSSM (MIPP with a structured scalar vector)             ip_proofs/src/tipa/structured_scalar_message.rs:181-331
 *   aggregate_proofs / verify_aggregate_proof                      ip_proofs/src/applications/groth16_aggregation.rs:77-231
 * Included at the end of ripp_oracle.c (it uses that file's static helpers).
 *
 * [ark-mem] `Fr::from_random_bytes` lives in ark-ff 0.4 (not vendored in /root/reference); it is restated here from the
 * published crate's algorithm: the first 32 digest bytes are read little-endian, the bits above MODULUS_BIT_SIZE = 255
 * are cleared, and the result is None (-> nonce retry, tipa/mod.rs:196-209) when the integer is >= r.
 */

/* outputs of aggregate_proofs; the same layout is declared as `ripp_aggregate_proof` in include/ripp_hip.h.
 * Step arrays are caller-allocated and filled in ROUND order (the reference stores them reversed, gipa.rs:298-299). */
typedef struct {
    fp12_t com_a, com_b, com_c, ip_ab;          /* groth16_aggregation.rs:62-66 */
    g1j_t agg_c;
    fr_t r;                                      /* the random-linear-combination challenge (:105-116) */
    /* tipa_proof_ab: TIPAProof<PairingInnerProduct, AFGHO-G1, AFGHO-G2, Identity<GT>> */
    fp12_t *ab_com_steps;                        /* [rounds][6] */
    fr_t *ab_transcript;                         /* [rounds] */
    g1j_t ab_base_a; g2j_t ab_base_b;            /* gipa_proof.r_base */
    g2j_t ab_final_ck_a; g1j_t ab_final_ck_b;    /* final_ck */
    g2j_t ab_opening_a; g1j_t ab_opening_b;      /* final_ck_proof */
    fr_t ab_kzg_c;
    /* tipa_proof_c: TIPAWithSSMProof<MultiexponentiationInnerProduct<G1>, AFGHO-G1, Identity<G1>> */
    fp12_t *c_com_gt;                            /* [rounds][2] = (com_1.0, com_2.0) */
    g1j_t *c_com_g1;                             /* [rounds][2] = (com_1.2[0], com_2.2[0]) */
    fr_t *c_transcript;
    g1j_t c_base_a; fr_t c_base_b;
    g2j_t c_final_ck_a; g2j_t c_opening_a;
    fr_t c_kzg_c;
} orc_aggregate_proof_t;

static int fr_from_random_bytes(const uint8_t dig[64], fr_t *out) {
    /* bits above MODULUS_BIT_SIZE are cleared (ark-ff 0.4 from_random_bytes_with_flags: `shave_bits` = 64 N - MODULUS_BIT_SIZE): 1 bit on BLS12-381 (255-bit r), 3 on BLS12-377 (253-bit r) */
    int top = 63; while (top > 0 && !((FR_R[3] >> top) & 1)) --top;
    fr_t t; memcpy(t.l, dig, 32); t.l[3] &= (top == 63) ? ~0ull : (((uint64_t)1 << (top + 1)) - 1);
    uint64_t bo = 0; for (int i = 0; i < 4; ++i) (void)sbb64(t.l[i], FR_R[i], &bo);
    if (!bo) return 0;
    fr_to_mont(out, &t); return 1;
}
ORC_API int orc_fr_from_random_bytes(const uint8_t dig[64], fr_t *out) { return fr_from_random_bytes(dig, out); }

/* structured_scalar_power (structured_scalar_message.rs:334-340) */
static void scalar_powers(fr_t *out, size_t n, const fr_t *s) { if (!n) return; out[0] = fr_one(); for (size_t i = 1; i < n; ++i) fr_mul(&out[i], &out[i - 1], s); }

/* SRS::setup without the RNG (tipa/mod.rs:150-165, 372-391): out[i] = s^i * G.  The fixed-base window table of the
 * reference is an implementation detail of the same group elements. */
ORC_API int orc_srs_powers_g1(const fr_t *s, size_t num, g1j_t *out) {
    fr_t *p = (fr_t *)malloc((num ? num : 1) * sizeof(fr_t)); scalar_powers(p, num, s); g1a_t g = g1_generator();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)num; ++i) g1a_mul(&out[i], &g, &p[i]);
    free(p); return 0;
}
ORC_API int orc_srs_powers_g2(const fr_t *s, size_t num, g2j_t *out) {
    fr_t *p = (fr_t *)malloc((num ? num : 1) * sizeof(fr_t)); scalar_powers(p, num, s); g2a_t g = g2_generator();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)num; ++i) g2a_mul(&out[i], &g, &p[i]);
    free(p); return 0;
}

/* KZG challenge point (tipa/mod.rs:194-209; structured_scalar_message.rs:231-246 without ck_b_final) */
static fr_t kzg_challenge(const fr_t *first, const g2j_t *ck_a_final, const g1j_t *ck_b_final) {
    g2a_t ka; g2_to_affine(&ka, ck_a_final); g1a_t kb; if (ck_b_final) g1_to_affine(&kb, ck_b_final);
    for (uint64_t nonce = 0;; ++nonce) {
        uint8_t buf[8 + 32 + 192 + 96], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        p += orc_ser_fr(first, p); p += orc_ser_g2(&ka, p); if (ck_b_final) p += orc_ser_g1(&kb, p);
        uint8_t dig[64]; blake2b(buf, (size_t)(p - buf), dig);
        fr_t c; if (fr_from_random_bytes(dig, &c)) return c;
    }
}

/* polynomial_coefficients_from_transcript (tipa/mod.rs:406-422) BEFORE the interleave: co[j] multiplies X^(2j) */
static void ck_poly_coeffs(const fr_t *tr, size_t rounds, const fr_t *r_shift, fr_t *co) {
    co[0] = fr_one(); fr_t p2r = *r_shift; size_t cnt = 1;
    for (size_t i = 0; i < rounds; ++i) {
        fr_t xr; fr_mul(&xr, &tr[i], &p2r);
        for (size_t j = 0; j < ((size_t)1 << i); ++j) fr_mul(&co[cnt + j], &co[j], &xr);
        cnt += (size_t)1 << i; fr_sqr(&p2r, &p2r);
    }
}
/* polynomial_evaluation_product_form_from_transcript (tipa/mod.rs:393-404) */
static fr_t ck_poly_eval(const fr_t *tr, size_t rounds, const fr_t *z, const fr_t *r_shift) {
    fr_t p, acc = fr_one(), one = fr_one(); fr_sqr(&p, z); fr_mul(&p, &p, r_shift);
    for (size_t i = 0; i < rounds; ++i) { fr_t t; fr_mul(&t, &tr[i], &p); fr_add(&t, &t, &one); fr_mul(&acc, &acc, &t); fr_sqr(&p, &p); }
    return acc;
}
/* quotient of (p(X) - p(c)) by (X - c), resized to 2n-1 coefficients (tipa/mod.rs:318-329) */
static fr_t *ck_quotient(const fr_t *tr, size_t rounds, const fr_t *r_shift, const fr_t *c) {
    const size_t n = (size_t)1 << rounds, m = 2 * n - 1;
    fr_t *co = (fr_t *)malloc(n * sizeof(fr_t)); ck_poly_coeffs(tr, rounds, r_shift, co);
    fr_t *q = (fr_t *)malloc(m * sizeof(fr_t));
    q[m - 1] = fr_zero();
    fr_t carry = fr_zero();                                   /* q[k-1] = p[k] + c*q[k], p[k] = co[k/2] for even k, 0 otherwise */
    for (size_t k = m - 1; k >= 1; --k) {
        fr_t t; fr_mul(&t, &carry, c);
        if (!(k & 1)) fr_add(&t, &t, &co[k >> 1]);
        q[k - 1] = t; carry = t;
    }
    free(co); return q;
}
/* prove_commitment_key_kzg_opening (tipa/mod.rs:304-337) */
static void kzg_opening_g1(const g1j_t *srs_powers, const fr_t *tr, size_t rounds, const fr_t *r_shift, const fr_t *c, g1j_t *out) {
    fr_t *q = ck_quotient(tr, rounds, r_shift, c); const size_t m = ((size_t)2 << rounds) - 1;
    orc_msm_g1_j(srs_powers, m, q, m, out); free(q);
}
static void kzg_opening_g2(const g2j_t *srs_powers, const fr_t *tr, size_t rounds, const fr_t *r_shift, const fr_t *c, g2j_t *out) {
    fr_t *q = ck_quotient(tr, rounds, r_shift, c); const size_t m = ((size_t)2 << rounds) - 1;
    orc_msm_g2_j(srs_powers, m, q, m, out); free(q);
}
static int pairing_eq(const g1j_t *a1, const g2j_t *b1, const g1j_t *a2, const g2j_t *b2) {
    fp12_t l, r; orc_pairing_product_j(a1, 1, b1, 1, &l); orc_pairing_product_j(a2, 1, b2, 1, &r); return fp12_eq(&l, &r);
}
/* verify_commitment_key_g2_kzg_opening (tipa/mod.rs:340-354): e(g, ck_final - h*eval) == e(g_beta - g*c, opening) */
static int kzg_verify_g2(const g1j_t *g, const g2j_t *h, const g1j_t *g_beta, const g2j_t *ck_final, const g2j_t *opening,
                         const fr_t *tr, size_t rounds, const fr_t *r_shift, const fr_t *c) {
    fr_t ev = ck_poly_eval(tr, rounds, c, r_shift);
    g2j_t t, l2; g2j_mul(&t, h, &ev); g2j_neg(&t, &t); g2j_add(&l2, ck_final, &t);
    g1j_t u, r1; g1j_mul(&u, g, c); g1j_neg(&u, &u); g1j_add(&r1, g_beta, &u);
    return pairing_eq(g, &l2, &r1, opening);
}
/* verify_commitment_key_g1_kzg_opening (tipa/mod.rs:356-370): e(ck_final - g*eval, h) == e(opening, h_alpha - h*c) */
static int kzg_verify_g1(const g1j_t *g, const g2j_t *h, const g2j_t *h_alpha, const g1j_t *ck_final, const g1j_t *opening,
                         const fr_t *tr, size_t rounds, const fr_t *r_shift, const fr_t *c) {
    fr_t ev = ck_poly_eval(tr, rounds, c, r_shift);
    g1j_t t, l1; g1j_mul(&t, g, &ev); g1j_neg(&t, &t); g1j_add(&l1, ck_final, &t);
    g2j_t u, r2; g2j_mul(&u, h, c); g2j_neg(&u, &u); g2j_add(&r2, h_alpha, &u);
    return pairing_eq(&l1, h, opening, &r2);
}

static size_t log2_exact(size_t n) { size_t r = 0; while (((size_t)1 << r) < n) ++r; return r; }

/* ================================================================== TIPA::prove_with_srs_shift (tipa/mod.rs:176-231), TIPP instantiation */
ORC_API int orc_tipa_tipp_prove(const g1j_t *g_alpha_powers, const g2j_t *h_beta_powers,
                                const g1j_t *m_a, const g2j_t *m_b, const g2j_t *ck_a, const g1j_t *ck_b, size_t n, const fr_t *r_shift,
                                fp12_t *com_steps, fr_t *transcript, g1j_t *base_a, g2j_t *base_b, g2j_t *final_ck_a, g1j_t *final_ck_b,
                                g2j_t *opening_a, g1j_t *opening_b, fr_t *kzg_c) {
    if (n < 2 || (n & (n - 1))) return 2;                       /* n = 1: `transcript.first().unwrap()` panics (:200-202) */
    int rc = orc_gipa_tipp_prove(m_a, m_b, ck_a, ck_b, n, com_steps, transcript, base_a, base_b, final_ck_a, final_ck_b); if (rc) return rc;
    const size_t rounds = log2_exact(n);
    fr_t *tr = (fr_t *)malloc(rounds * sizeof(fr_t)), *tri = (fr_t *)malloc(rounds * sizeof(fr_t));
    for (size_t i = 0; i < rounds; ++i) { tr[i] = transcript[rounds - 1 - i]; fr_inv(&tri[i], &tr[i]); }     /* aux.r_transcript, :190-191 */
    fr_t r_inv, one = fr_one(); fr_inv(&r_inv, r_shift);                                                     /* :192 */
    const fr_t c = kzg_challenge(&tr[0], final_ck_a, final_ck_b);
    kzg_opening_g2(h_beta_powers, tri, rounds, &r_inv, &c, opening_a);                                       /* :212-217 */
    kzg_opening_g1(g_alpha_powers, tr, rounds, &one, &c, opening_b);                                         /* :218-223 */
    *kzg_c = c; free(tr); free(tri); return 0;
}

/* _compute_recursive_challenges for the TIPP instantiation (gipa.rs:322-363); tr in ROUND order */
static void tipp_replay(const fp12_t com[3], const fp12_t *com_steps, size_t rounds, fr_t *tr, fp12_t out[3]) {
    fp12_t ca = com[0], cb = com[1], ct = com[2];
    for (size_t k = 0; k < rounds; ++k) {
        const fp12_t *s = com_steps + 6 * k;
        fr_t c_inv, c = gipa_challenge(k ? &tr[k - 1] : NULL, s, &c_inv);
        fp12_t t1, t2;
        gt_pow(&t1, &s[0], &c); gt_pow(&t2, &s[3], &c_inv); fp12_mul(&ca, &ca, &t1); fp12_mul(&ca, &ca, &t2);
        gt_pow(&t1, &s[1], &c); gt_pow(&t2, &s[4], &c_inv); fp12_mul(&cb, &cb, &t1); fp12_mul(&cb, &cb, &t2);
        gt_pow(&t1, &s[2], &c); gt_pow(&t2, &s[5], &c_inv); fp12_mul(&ct, &ct, &t1); fp12_mul(&ct, &ct, &t2);
        tr[k] = c;
    }
    out[0] = ca; out[1] = cb; out[2] = ct;
}

/* TIPA::verify_with_srs_shift (tipa/mod.rs:242-301).  v_srs = (g, h, g_beta, h_alpha).  1 accept / 0 reject */
ORC_API int orc_tipa_tipp_verify(const g1j_t *g, const g2j_t *h, const g1j_t *g_beta, const g2j_t *h_alpha, const fp12_t com[3],
                                 const fp12_t *com_steps, size_t rounds, const g1j_t *base_a, const g2j_t *base_b,
                                 const g2j_t *final_ck_a, const g1j_t *final_ck_b, const g2j_t *opening_a, const g1j_t *opening_b, const fr_t *r_shift) {
    if (rounds == 0) return -2;
    fr_t *trf = (fr_t *)malloc(rounds * sizeof(fr_t)), *tr = (fr_t *)malloc(rounds * sizeof(fr_t)), *tri = (fr_t *)malloc(rounds * sizeof(fr_t));
    fp12_t bc[3]; tipp_replay(com, com_steps, rounds, trf, bc);
    for (size_t i = 0; i < rounds; ++i) { tr[i] = trf[rounds - 1 - i]; fr_inv(&tri[i], &tr[i]); }
    const fr_t c = kzg_challenge(&tr[0], final_ck_a, final_ck_b);
    fr_t r_inv, one = fr_one(); fr_inv(&r_inv, r_shift);
    int ok = kzg_verify_g2(g, h, g_beta, final_ck_a, opening_a, tri, rounds, &r_inv, &c);                      /* :273-280 */
    ok &= kzg_verify_g1(g, h, h_alpha, final_ck_b, opening_b, tr, rounds, &one, &c);                          /* :281-288 */
    fp12_t e1, e2, e3;                                                                                         /* :291-298 */
    orc_pairing_product_j(base_a, 1, final_ck_a, 1, &e1); orc_pairing_product_j(final_ck_b, 1, base_b, 1, &e2); orc_pairing_product_j(base_a, 1, base_b, 1, &e3);
    ok &= fp12_eq(&e1, &bc[0]) && fp12_eq(&e2, &bc[1]) && fp12_eq(&e3, &bc[2]);
    free(trf); free(tr); free(tri); return ok;
}

/* ================================================================== GIPA with SSMPlaceholderCommitment on the right
 * GIPA<MultiexponentiationInnerProduct<G1>, AFGHO-G1, SSMPlaceholder<Fr>, Identity<G1, Fr>, Blake2b>
 * (structured_scalar_message.rs:211-228; groth16_aggregation.rs:42-48).  RMC::commit == Fr::zero() (ssm.rs:44-46). */
static fr_t gipa_ssm_challenge(const fr_t *prev, const fp12_t gt[2], const g1j_t g1[2], fr_t *c_inv_out) {
    g1a_t pa[2]; g1_to_affine(&pa[0], &g1[0]); g1_to_affine(&pa[1], &g1[1]);
    for (uint64_t nonce = 0;; ++nonce) {
        uint8_t buf[8 + 32 + 2 * (576 + 32 + 8 + 96)], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        fr_t zero = fr_zero(); p += orc_ser_fr(prev ? prev : &zero, p);
        for (int k = 0; k < 2; ++k) {
            p += orc_ser_gt(&gt[k], p);                                   /* com_k.0  (LMC output, GT) */
            p += orc_ser_fr(&zero, p);                                    /* com_k.1  (placeholder Fr::zero()) */
            uint64_t one = 1; memcpy(p, &one, 8); p += 8;                 /* com_k.2  IdentityOutput(Vec<G1>): u64 length + point */
            p += orc_ser_g1(&pa[k], p);
        }
        uint8_t dig[64]; blake2b(buf, (size_t)(p - buf), dig);
        uint64_t hi = 0, lo = 0; for (int i = 0; i < 8; ++i) { hi = (hi << 8) | dig[i]; lo = (lo << 8) | dig[8 + i]; }
        fr_t c128 = fr_from_u128(lo, hi);
        if (!fr_is_zero(&c128)) { fr_t inv; fr_inv(&inv, &c128); *c_inv_out = c128; return inv; }
    }
}

/* TIPAWithSSM::prove_with_structured_scalar_message (structured_scalar_message.rs:211-268) */
ORC_API int orc_tipa_ssm_prove(const g2j_t *h_beta_powers, const g1j_t *m_a_in, const fr_t *m_b_in, const g2j_t *ck_a_in, size_t n,
                               fp12_t *com_gt, g1j_t *com_g1, fr_t *transcript, g1j_t *base_a, fr_t *base_b, g2j_t *final_ck_a, g2j_t *opening_a, fr_t *kzg_c) {
    if (n < 2 || (n & (n - 1))) return 2;
    g1j_t *m_a = (g1j_t *)malloc(n * sizeof(g1j_t)); g2j_t *ck_a = (g2j_t *)malloc(n * sizeof(g2j_t)); fr_t *m_b = (fr_t *)malloc(n * sizeof(fr_t));
    memcpy(m_a, m_a_in, n * sizeof(g1j_t)); memcpy(ck_a, ck_a_in, n * sizeof(g2j_t)); memcpy(m_b, m_b_in, n * sizeof(fr_t));
    size_t len = n, round = 0;
    while (len > 1) {                                                       /* gipa.rs:196-293 */
        const size_t split = len / 2;
        const g1j_t *m_a_1 = m_a + split, *m_a_2 = m_a; const g2j_t *ck_a_1 = ck_a, *ck_a_2 = ck_a + split; const fr_t *m_b_1 = m_b, *m_b_2 = m_b + split;
        fp12_t *gt = com_gt + 2 * round; g1j_t *g1 = com_g1 + 2 * round;
        orc_pairing_product_j(m_a_1, split, ck_a_1, split, &gt[0]);         /* LMC::commit(ck_a_1, m_a_1) */
        orc_msm_g1_j(m_a_1, split, m_b_1, split, &g1[0]);                   /* IP::inner_product(m_a_1, m_b_1) */
        orc_pairing_product_j(m_a_2, split, ck_a_2, split, &gt[1]);
        orc_msm_g1_j(m_a_2, split, m_b_2, split, &g1[1]);
        fr_t c_inv, c = gipa_ssm_challenge(round ? &transcript[round - 1] : NULL, gt, g1, &c_inv);
        g1j_t *na = (g1j_t *)malloc(split * sizeof(g1j_t)); g2j_t *nka = (g2j_t *)malloc(split * sizeof(g2j_t));
        orc_fold_g1_j(m_a_1, m_a_2, split, &c, na);
        orc_fold_g2_j(ck_a_2, ck_a_1, split, &c_inv, nka);
        for (size_t i = 0; i < split; ++i) { fr_t t; fr_mul(&t, &m_b_2[i], &c_inv); fr_add(&m_b[i], &t, &m_b_1[i]); }   /* gipa.rs:270-274 */
        memcpy(m_a, na, split * sizeof(g1j_t)); memcpy(ck_a, nka, split * sizeof(g2j_t)); free(na); free(nka);
        transcript[round] = c; ++round; len = split;
    }
    *base_a = m_a[0]; *base_b = m_b[0]; *final_ck_a = ck_a[0];
    free(m_a); free(ck_a); free(m_b);
    const size_t rounds = round;
    fr_t *tri = (fr_t *)malloc(rounds * sizeof(fr_t));
    for (size_t i = 0; i < rounds; ++i) fr_inv(&tri[i], &transcript[rounds - 1 - i]);                            /* ssm.rs:227-229 */
    fr_t one = fr_one();
    const fr_t c = kzg_challenge(&transcript[rounds - 1], final_ck_a, NULL);                                     /* ssm.rs:231-246 */
    kzg_opening_g2(h_beta_powers, tri, rounds, &one, &c, opening_a);                                             /* ssm.rs:249-254 */
    *kzg_c = c; free(tri); return 0;
}

/* TIPAWithSSM::verify_with_structured_scalar_message (structured_scalar_message.rs:270-331); com = (com_a GT, com_t G1) */
ORC_API int orc_tipa_ssm_verify(const g1j_t *g, const g2j_t *h, const g1j_t *g_beta, const fp12_t *com_a_in, const g1j_t *com_t_in, const fr_t *scalar_b,
                                const fp12_t *com_gt, const g1j_t *com_g1, size_t rounds, const g1j_t *base_a, const g2j_t *final_ck_a, const g2j_t *opening_a) {
    if (rounds == 0) return -2;
    fp12_t ca = *com_a_in; g1j_t ct = *com_t_in;
    fr_t *trf = (fr_t *)malloc(rounds * sizeof(fr_t)), *tr = (fr_t *)malloc(rounds * sizeof(fr_t)), *tri = (fr_t *)malloc(rounds * sizeof(fr_t));
    for (size_t k = 0; k < rounds; ++k) {                                                                       /* gipa.rs:329-360 */
        fr_t c_inv, c = gipa_ssm_challenge(k ? &trf[k - 1] : NULL, com_gt + 2 * k, com_g1 + 2 * k, &c_inv);
        fp12_t t1, t2; gt_pow(&t1, &com_gt[2 * k], &c); gt_pow(&t2, &com_gt[2 * k + 1], &c_inv); fp12_mul(&ca, &ca, &t1); fp12_mul(&ca, &ca, &t2);
        g1j_t u1, u2; g1j_mul(&u1, &com_g1[2 * k], &c); g1j_mul(&u2, &com_g1[2 * k + 1], &c_inv); g1j_add(&ct, &ct, &u1); g1j_add(&ct, &ct, &u2);
        trf[k] = c;
    }
    for (size_t i = 0; i < rounds; ++i) { tr[i] = trf[rounds - 1 - i]; fr_inv(&tri[i], &tr[i]); }
    const fr_t c = kzg_challenge(&tr[0], final_ck_a, NULL);
    fr_t one = fr_one();
    int ok = kzg_verify_g2(g, h, g_beta, final_ck_a, opening_a, tri, rounds, &one, &c);                          /* ssm.rs:305-312 */
    fr_t p2b = *scalar_b, b_base = fr_one();                                                                     /* ssm.rs:315-321 */
    for (size_t i = 0; i < rounds; ++i) { fr_t t; fr_mul(&t, &tri[i], &p2b); fr_add(&t, &t, &one); fr_mul(&b_base, &b_base, &t); fr_sqr(&p2b, &p2b); }
    fp12_t e1; orc_pairing_product_j(base_a, 1, final_ck_a, 1, &e1);                                             /* ssm.rs:324-328 */
    g1j_t tb; g1j_mul(&tb, base_a, &b_base);
    ok &= fp12_eq(&e1, &ca) && g1j_eq(&tb, &ct);
    free(trf); free(tr); free(tri); return ok;
}

/* ================================================================== Groth16 aggregation (groth16_aggregation.rs) */
static fr_t aggregation_challenge(const fp12_t *com_a, const fp12_t *com_b, const fp12_t *com_c) {              /* :105-116, 173-184 */
    for (uint64_t nonce = 0;; ++nonce) {
        uint8_t buf[8 + 3 * 576], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        p += orc_ser_gt(com_a, p); p += orc_ser_gt(com_b, p); p += orc_ser_gt(com_c, p);
        uint8_t dig[64]; blake2b(buf, (size_t)(p - buf), dig);
        fr_t r; if (fr_from_random_bytes(dig, &r)) return r;
    }
}

/* aggregate_proofs (groth16_aggregation.rs:77-160); SRS powers have 2n-1 entries each.  returns 3 when the :133-136 assertion fails */
ORC_API int orc_aggregate_proofs(const g1j_t *g_alpha_powers, const g2j_t *h_beta_powers, const g1a_t *a_in, const g2a_t *b_in, const g1a_t *c_in, size_t n,
                                 orc_aggregate_proof_t *out) {
    if (n < 2 || (n & (n - 1))) return 2;
    g1j_t *a = (g1j_t *)malloc(n * sizeof(g1j_t)), *c = (g1j_t *)malloc(n * sizeof(g1j_t)), *ck_2 = (g1j_t *)malloc(n * sizeof(g1j_t)), *a_r = (g1j_t *)malloc(n * sizeof(g1j_t));
    g2j_t *b = (g2j_t *)malloc(n * sizeof(g2j_t)), *ck_1 = (g2j_t *)malloc(n * sizeof(g2j_t)), *ck_1_r = (g2j_t *)malloc(n * sizeof(g2j_t));
    for (size_t i = 0; i < n; ++i) { a[i] = g1_from_affine(&a_in[i]); b[i] = g2_from_affine(&b_in[i]); c[i] = g1_from_affine(&c_in[i]);
                                     ck_1[i] = h_beta_powers[2 * i]; ck_2[i] = g_alpha_powers[2 * i]; }         /* get_commitment_keys, tipa/mod.rs:114-118 */
    orc_pairing_product_j(a, n, ck_1, n, &out->com_a);                                                           /* :100-102 */
    orc_pairing_product_j(ck_2, n, b, n, &out->com_b);
    orc_pairing_product_j(c, n, ck_1, n, &out->com_c);
    const fr_t r = aggregation_challenge(&out->com_a, &out->com_b, &out->com_c); out->r = r;
    fr_t *r_vec = (fr_t *)malloc(n * sizeof(fr_t)), *r_inv_vec = (fr_t *)malloc(n * sizeof(fr_t));
    scalar_powers(r_vec, n, &r);                                                                                 /* :118 */
    for (size_t i = 0; i < n; ++i) fr_inv(&r_inv_vec[i], &r_vec[i]);                                             /* :130 */
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) { g1j_mul(&a_r[i], &a[i], &r_vec[i]); g2j_mul(&ck_1_r[i], &ck_1[i], &r_inv_vec[i]); }   /* :119-123, 127-131 */
    orc_pairing_product_j(a_r, n, b, n, &out->ip_ab);                                                            /* :124 */
    orc_msm_g1_j(c, n, r_vec, n, &out->agg_c);                                                                   /* :125 */
    fp12_t chk; orc_pairing_product_j(a_r, n, ck_1_r, n, &chk);                                                  /* :133-136 */
    int rc = fp12_eq(&chk, &out->com_a) ? 0 : 3;
    if (!rc) rc = orc_tipa_tipp_prove(g_alpha_powers, h_beta_powers, a_r, b, ck_1_r, ck_2, n, &r, out->ab_com_steps, out->ab_transcript,      /* :138-143 */
                                      &out->ab_base_a, &out->ab_base_b, &out->ab_final_ck_a, &out->ab_final_ck_b, &out->ab_opening_a, &out->ab_opening_b, &out->ab_kzg_c);
    if (!rc) rc = orc_tipa_ssm_prove(h_beta_powers, c, r_vec, ck_1, n, out->c_com_gt, out->c_com_g1, out->c_transcript,                       /* :145-149 */
                                     &out->c_base_a, &out->c_base_b, &out->c_final_ck_a, &out->c_opening_a, &out->c_kzg_c);
    free(a); free(b); free(c); free(ck_1); free(ck_2); free(a_r); free(ck_1_r); free(r_vec); free(r_inv_vec);
    return rc;
}

/* verify_aggregate_proof (groth16_aggregation.rs:162-231).  vk = (alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1[m+1]);
 * public_inputs[n][m] row-major.  1 accept / 0 reject */
ORC_API int orc_verify_aggregate_proof(const g1j_t *g, const g2j_t *h, const g1j_t *g_beta, const g2j_t *h_alpha,
                                       const g1a_t *alpha_g1, const g2a_t *beta_g2, const g2a_t *gamma_g2, const g2a_t *delta_g2, const g1a_t *gamma_abc_g1,
                                       const fr_t *public_inputs, size_t n, size_t m, const orc_aggregate_proof_t *pf) {
    if (n < 2 || (n & (n - 1))) return -2;
    const size_t rounds = log2_exact(n);
    const fr_t r = aggregation_challenge(&pf->com_a, &pf->com_b, &pf->com_c);
    const fp12_t com_ab[3] = { pf->com_a, pf->com_b, pf->ip_ab };
    int ok = orc_tipa_tipp_verify(g, h, g_beta, h_alpha, com_ab, pf->ab_com_steps, rounds, &pf->ab_base_a, &pf->ab_base_b,                    /* :187-198 */
                                  &pf->ab_final_ck_a, &pf->ab_final_ck_b, &pf->ab_opening_a, &pf->ab_opening_b, &r) == 1;
    ok &= orc_tipa_ssm_verify(g, h, g_beta, &pf->com_c, &pf->agg_c, &r, pf->c_com_gt, pf->c_com_g1, rounds, &pf->c_base_a, &pf->c_final_ck_a, &pf->c_opening_a) == 1;   /* :199-205 */
    /* r_sum = (r^n - 1) / (r - 1)  (:209-210) */
    fr_t one = fr_one(), rn = one, num, den, r_sum;
    for (size_t i = 0; i < n; ++i) fr_mul(&rn, &rn, &r);
    fr_sub(&num, &rn, &one); fr_sub(&den, &r, &one); fr_inv(&den, &den); fr_mul(&r_sum, &num, &den);
    fr_t *r_vec = (fr_t *)malloc(n * sizeof(fr_t)); scalar_powers(r_vec, n, &r);
    g1j_t ar; g1a_mul(&ar, alpha_g1, &r_sum);
    g1j_t g_ic; g1a_mul(&g_ic, &gamma_abc_g1[0], &r_sum);                                                         /* :215-226 */
    for (size_t i = 0; i < m; ++i) {
        fr_t ip = fr_zero(); for (size_t k = 0; k < n; ++k) { fr_t t; fr_mul(&t, &public_inputs[k * m + i], &r_vec[k]); fr_add(&ip, &ip, &t); }
        g1j_t t; g1a_mul(&t, &gamma_abc_g1[i + 1], &ip); g1j_add(&g_ic, &g_ic, &t);
    }
    g2j_t bj = g2_from_affine(beta_g2), gj = g2_from_affine(gamma_g2), dj = g2_from_affine(delta_g2);
    fp12_t p1, p2, p3, rhs;
    orc_pairing_product_j(&ar, 1, &bj, 1, &p1); orc_pairing_product_j(&g_ic, 1, &gj, 1, &p2); orc_pairing_product_j(&pf->agg_c, 1, &dj, 1, &p3);
    fp12_mul(&rhs, &p1, &p2); fp12_mul(&rhs, &rhs, &p3);
    ok &= fp12_eq(&pf->ip_ab, &rhs);                                                                             /* :229 */
    free(r_vec); return ok;
}
