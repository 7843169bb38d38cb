/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * CPU restatement (plain C + OpenMP) of the data-parallel hot path of arkworks-rs/ripp on BLS12-381.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (ripp_amd/, include/) never does.
 *
 * The reference cannot be compiled or imported in this environment (no Rust toolchain, un-vendored
 * `ark-* 0.4` crates) and its own tests hold no byte-level golden vector, hence
 * "parity unpinned": what pins this file is tests/model/bls381_model.py (independent big-integer model,
 * fixtures under tests/golden/) plus algebraic laws.
 *
 * Each function cites the reference lines whose control flow it follows.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "pairing.h"
#include "hash.h"

#define ORC_API __attribute__((visibility("default")))

static int orc_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API int orc_num_threads(void) { return orc_threads(); }
ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ================================================================== serialisation (ark-serialize 0.4, uncompressed) */
static void ser_fp_be(uint8_t *out, const fp_t *a) {    /* canonical big-endian, zcash layout of ark-bls12-381 0.4 */
    fp_t c; fp_from_mont(&c, a);
    for (int i = 0; i < 6; ++i) for (int b = 0; b < 8; ++b) out[47 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
}
static void ser_fp_le(uint8_t *out, const fp_t *a) { fp_t c; fp_from_mont(&c, a); memcpy(out, c.l, 48); }
ORC_API size_t orc_ser_fr(const fr_t *a, uint8_t out[32]) { fr_t c; fr_from_mont(&c, a); memcpy(out, c.l, 32); return 32; }
#ifdef ORC_BLS12_377
/* ark-bls12-377 has no zcash-style override: the GENERIC ark-ec 0.4 short-Weierstrass encoding applies [ark-mem] -- x then y,
 * little-endian canonical, SWFlags in the two top bits of the LAST byte: bit 7 = YIsNegative (y > -y as integers; Fq2 compares c1 first),
 * bit 6 = PointAtInfinity (x = y = 0 written).  serialize_with_mode writes the flags in the uncompressed form too. */
static int fp_canon_gt(const fp_t *a, const fp_t *b) {                      /* canonical integers: a > b */
    fp_t x, y; fp_from_mont(&x, a); fp_from_mont(&y, b);
    for (int i = 5; i >= 0; --i) if (x.l[i] != y.l[i]) return x.l[i] > y.l[i];
    return 0;
}
ORC_API size_t orc_ser_g1(const g1a_t *p, uint8_t out[96]) {
    if (g1a_is_inf(p)) { memset(out, 0, 96); out[95] = 0x40; return 96; }
    ser_fp_le(out, &p->x); ser_fp_le(out + 48, &p->y);
    fp_t ny; fp_neg(&ny, &p->y); if (fp_canon_gt(&p->y, &ny)) out[95] |= 0x80;
    return 96;
}
ORC_API size_t orc_ser_g2(const g2a_t *p, uint8_t out[192]) {
    if (g2a_is_inf(p)) { memset(out, 0, 192); out[191] = 0x40; return 192; }
    ser_fp_le(out, &p->x.c0); ser_fp_le(out + 48, &p->x.c1); ser_fp_le(out + 96, &p->y.c0); ser_fp_le(out + 144, &p->y.c1);
    fp2_t ny; fp2_neg(&ny, &p->y);
    const int neg = fp_is_zero(&p->y.c1) ? fp_canon_gt(&p->y.c0, &ny.c0) : fp_canon_gt(&p->y.c1, &ny.c1);
    if (neg) out[191] |= 0x80;
    return 192;
}
#else
ORC_API size_t orc_ser_g1(const g1a_t *p, uint8_t out[96]) {
    if (g1a_is_inf(p)) { memset(out, 0, 96); out[0] = 0x40; return 96; }
    ser_fp_be(out, &p->x); ser_fp_be(out + 48, &p->y); return 96;
}
ORC_API size_t orc_ser_g2(const g2a_t *p, uint8_t out[192]) {
    if (g2a_is_inf(p)) { memset(out, 0, 192); out[0] = 0x40; return 192; }
    ser_fp_be(out, &p->x.c1); ser_fp_be(out + 48, &p->x.c0); ser_fp_be(out + 96, &p->y.c1); ser_fp_be(out + 144, &p->y.c0); return 192;
}
#endif
ORC_API size_t orc_ser_gt(const fp12_t *f, uint8_t out[576]) {
    const fp2_t *c[6] = { &f->c0.c0, &f->c0.c1, &f->c0.c2, &f->c1.c0, &f->c1.c1, &f->c1.c2 };
    for (int i = 0; i < 6; ++i) { ser_fp_le(out + 96 * i, &c[i]->c0); ser_fp_le(out + 96 * i + 48, &c[i]->c1); }
    return 576;
}

/* ================================================================== P4 batch normalisation (CurveGroup::normalize_batch) */
ORC_API int orc_normalize_g1(const g1j_t *in, size_t n, g1a_t *out) {
    int T = orc_threads(); size_t chunk = (n + T - 1) / (T ? T : 1); if (chunk == 0) chunk = 1;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)((n + chunk - 1) / chunk); ++c) { size_t s = c * chunk, e = s + chunk > n ? n : s + chunk; g1_normalize_batch(out + s, in + s, e - s); }
    return 0;
}
ORC_API int orc_normalize_g2(const g2j_t *in, size_t n, g2a_t *out) {
    int T = orc_threads(); size_t chunk = (n + T - 1) / (T ? T : 1); if (chunk == 0) chunk = 1;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)((n + chunk - 1) / chunk); ++c) { size_t s = c * chunk, e = s + chunk > n ? n : s + chunk; g2_normalize_batch(out + s, in + s, e - s); }
    return 0;
}

/* ================================================================== P1 pairing product
 * inner_products/src/lib.rs:77-116 `cfg_multi_pairing` and sipp/src/lib.rs:196-216: prepare, split into
 * `num_threads` chunks (chunk_size = len / threads, or 1), per-chunk multi_miller_loop, product, ONE final exp.
 * G2 line coefficients are prepared per group of 4 inside the chunk (same arithmetic as the reference's
 * separate prepare pass, but 78 KB live per thread instead of 19.6 KB per point). */
static void miller_product_affine(fp12_t *out, const g1a_t *a, const g2a_t *b, size_t n) {
    fp12_t total = fp12_one();
    if (n == 0) { *out = total; return; }
    size_t num_chunks = (size_t)orc_threads();
    size_t chunk_size = num_chunks <= n ? n / num_chunks : 1;
    size_t nchunks = (n + chunk_size - 1) / chunk_size;
    fp12_t *parts = (fp12_t *)malloc(nchunks * sizeof(fp12_t));
#pragma omp parallel for schedule(dynamic, 1)
    for (long c = 0; c < (long)nchunks; ++c) {
        size_t s = (size_t)c * chunk_size, e = s + chunk_size > n ? n : s + chunk_size;
        g2prep_t *prep = (g2prep_t *)malloc(4 * sizeof(g2prep_t));
        fp12_t acc = fp12_one();
        for (size_t i = s; i < e; i += 4) {
            size_t k = e - i < 4 ? e - i : 4;
            for (size_t j = 0; j < k; ++j) g2_prepare(&prep[j], &b[i + j]);
            fp12_t f; multi_miller_loop(&f, a + i, prep, k);     /* returns the conjugated value */
            fp12_mul(&acc, &acc, &f);
        }
        parts[c] = acc; free(prep);
    }
    for (size_t c = 0; c < nchunks; ++c) fp12_mul(&total, &total, &parts[c]);
    free(parts);
    *out = total;
}
ORC_API int orc_miller_product_a(const g1a_t *a, const g2a_t *b, size_t n, fp12_t *out) { miller_product_affine(out, a, b, n); return 0; }
ORC_API int orc_final_exp(const fp12_t *f, fp12_t *out) { final_exponentiation(out, f); return 0; }
/* sipp::product_of_pairings (sipp/src/lib.rs:219-224): the all-ones scaling of :189-194 leaves `a` unchanged */
ORC_API int orc_pairing_product_a(const g1a_t *a, const g2a_t *b, size_t n, fp12_t *out) {
    fp12_t ml; miller_product_affine(&ml, a, b, n); final_exponentiation(out, &ml); return 0;
}
/* PairingInnerProduct::inner_product (inner_products/src/lib.rs:61-73) */
ORC_API int orc_pairing_product_j(const g1j_t *l, size_t nl, const g2j_t *r, size_t nr, fp12_t *out) {
    if (nl != nr) return 1;                                      /* InnerProductError::MessageLengthInvalid */
    g1a_t *a = (g1a_t *)malloc((nl ? nl : 1) * sizeof(g1a_t)); g2a_t *b = (g2a_t *)malloc((nl ? nl : 1) * sizeof(g2a_t));
    orc_normalize_g1(l, nl, a); orc_normalize_g2(r, nr, b);      /* :80-81 */
    orc_pairing_product_a(a, b, nl, out);
    free(a); free(b); return 0;
}

/* ================================================================== P2 MSM (VariableBaseMSM::msm, ark-ec 0.4 window rule) */
static int ln_without_floats(size_t a) { int lg = 0; while ((a >> lg) > 1) ++lg; return lg * 69 / 100; }
#define DEFINE_MSM(G, AT, JT)                                                                        \
static void G##_msm_affine(JT *out, const AT *bases, const fr_t *scalars_mont, size_t n) {           \
    if (n == 0) { *out = G##j_inf(); return; }                                                       \
    int c = n < 32 ? 3 : ln_without_floats(n) + 2;                                                   \
    int nwin = (255 + c - 1) / c;                                                                    \
    fr_t *sc = (fr_t *)malloc(n * sizeof(fr_t));                                                     \
    _Pragma("omp parallel for schedule(static)")                                                     \
    for (long i = 0; i < (long)n; ++i) fr_from_mont(&sc[i], &scalars_mont[i]);                       \
    JT *wsum = (JT *)malloc(nwin * sizeof(JT));                                                      \
    _Pragma("omp parallel for schedule(dynamic, 1)")                                                 \
    for (int w = 0; w < nwin; ++w) {                                                                 \
        size_t nb = ((size_t)1 << c) - 1; int start = w * c;                                         \
        JT *buckets = (JT *)malloc(nb * sizeof(JT));                                                 \
        for (size_t k = 0; k < nb; ++k) buckets[k] = G##j_inf();                                     \
        for (size_t i = 0; i < n; ++i) {                                                             \
            int limb = start >> 6, sh = start & 63;                                                  \
            uint64_t d = sc[i].l[limb] >> sh;                                                        \
            if (sh + c > 64 && limb + 1 < 4) d |= sc[i].l[limb + 1] << (64 - sh);                    \
            d &= ((uint64_t)1 << c) - 1;                                                             \
            if (d) G##j_add_affine(&buckets[d - 1], &buckets[d - 1], &bases[i]);                     \
        }                                                                                            \
        JT run = G##j_inf(), res = G##j_inf();                                                       \
        for (size_t k = nb; k-- > 0;) { G##j_add(&run, &run, &buckets[k]); G##j_add(&res, &res, &run); } \
        wsum[w] = res; free(buckets);                                                                \
    }                                                                                                \
    JT total = G##j_inf();                                                                           \
    for (int w = nwin - 1; w >= 1; --w) { G##j_add(&total, &total, &wsum[w]); for (int k = 0; k < c; ++k) G##j_dbl(&total, &total); } \
    G##j_add(&total, &total, &wsum[0]);                                                              \
    *out = total; free(wsum); free(sc);                                                              \
}
DEFINE_MSM(g1, g1a_t, g1j_t)
DEFINE_MSM(g2, g2a_t, g2j_t)

/* MultiexponentiationInnerProduct::inner_product (inner_products/src/lib.rs:128-141) */
ORC_API int orc_msm_g1_j(const g1j_t *bases, size_t nl, const fr_t *scalars, size_t nr, g1j_t *out) {
    if (nl != nr) return 1;
    g1a_t *a = (g1a_t *)malloc((nl ? nl : 1) * sizeof(g1a_t)); orc_normalize_g1(bases, nl, a);
    g1_msm_affine(out, a, scalars, nl); free(a); return 0;
}
ORC_API int orc_msm_g2_j(const g2j_t *bases, size_t nl, const fr_t *scalars, size_t nr, g2j_t *out) {
    if (nl != nr) return 1;
    g2a_t *a = (g2a_t *)malloc((nl ? nl : 1) * sizeof(g2a_t)); orc_normalize_g2(bases, nl, a);
    g2_msm_affine(out, a, scalars, nl); free(a); return 0;
}
ORC_API int orc_msm_g1_a(const g1a_t *bases, const fr_t *scalars, size_t n, g1j_t *out) { g1_msm_affine(out, bases, scalars, n); return 0; }
ORC_API int orc_msm_g2_a(const g2a_t *bases, const fr_t *scalars, size_t n, g2j_t *out) { g2_msm_affine(out, bases, scalars, n); return 0; }
/* naive sum of scalar multiples: the algorithm-independent check for the MSM */
ORC_API int orc_msm_g1_naive(const g1a_t *bases, const fr_t *scalars, size_t n, g1j_t *out) {
    g1j_t acc = g1j_inf(); for (size_t i = 0; i < n; ++i) { g1j_t t; g1a_mul(&t, &bases[i], &scalars[i]); g1j_add(&acc, &acc, &t); } *out = acc; return 0;
}
ORC_API int orc_msm_g2_naive(const g2a_t *bases, const fr_t *scalars, size_t n, g2j_t *out) {
    g2j_t acc = g2j_inf(); for (size_t i = 0; i < n; ++i) { g2j_t t; g2a_mul(&t, &bases[i], &scalars[i]); g2j_add(&acc, &acc, &t); } *out = acc; return 0;
}

/* ================================================================== P3 halving-round fold  out[i] = s*hi[i] + lo[i]
 * sipp/src/lib.rs:87-100 (affine in, `a_r * x + a_l`, then normalize_batch); ip_proofs/src/gipa.rs:262-290 (projective) */
ORC_API int orc_fold_g1_a(const g1a_t *hi, const g1a_t *lo, size_t half, const fr_t *s, g1a_t *out) {
    g1j_t *t = (g1j_t *)malloc((half ? half : 1) * sizeof(g1j_t));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)half; ++i) { g1a_mul(&t[i], &hi[i], s); g1j_add_affine(&t[i], &t[i], &lo[i]); }
    orc_normalize_g1(t, half, out); free(t); return 0;
}
ORC_API int orc_fold_g2_a(const g2a_t *hi, const g2a_t *lo, size_t half, const fr_t *s, g2a_t *out) {
    g2j_t *t = (g2j_t *)malloc((half ? half : 1) * sizeof(g2j_t));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)half; ++i) { g2a_mul(&t[i], &hi[i], s); g2j_add_affine(&t[i], &t[i], &lo[i]); }
    orc_normalize_g2(t, half, out); free(t); return 0;
}
ORC_API int orc_fold_g1_j(const g1j_t *hi, const g1j_t *lo, size_t half, const fr_t *s, g1j_t *out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)half; ++i) { g1j_t t; g1j_mul(&t, &hi[i], s); g1j_add(&out[i], &t, &lo[i]); }
    return 0;
}
ORC_API int orc_fold_g2_j(const g2j_t *hi, const g2j_t *lo, size_t half, const fr_t *s, g2j_t *out) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)half; ++i) { g2j_t t; g2j_mul(&t, &hi[i], s); g2j_add(&out[i], &t, &lo[i]); }
    return 0;
}
/* a_i <- r_i * a_i with per-element scalars (sipp/src/lib.rs:61-66, 189-194), normalised */
ORC_API int orc_scale_g1_a(const g1a_t *a, const fr_t *r, size_t n, g1a_t *out) {
    g1j_t *t = (g1j_t *)malloc((n ? n : 1) * sizeof(g1j_t));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) g1a_mul(&t[i], &a[i], &r[i]);
    orc_normalize_g1(t, n, out); free(t); return 0;
}

/* ================================================================== SIPP (sipp/src/lib.rs) */
/* product_of_pairings_with_coeffs (sipp/src/lib.rs:184-217) */
ORC_API int orc_product_of_pairings_with_coeffs(const g1a_t *a, const g2a_t *b, const fr_t *r, size_t n, fp12_t *out) {
    g1a_t *ar = (g1a_t *)malloc((n ? n : 1) * sizeof(g1a_t));
    orc_scale_g1_a(a, r, n, ar); orc_pairing_product_a(ar, b, n, out); free(ar); return 0;
}

static void sipp_seed_digest(const g1a_t *a, const g2a_t *b, const fr_t *r, size_t n, const fp12_t *value, uint8_t digest[32]) {
    /* (a, b, r, value).serialize_uncompressed: each slice = u64-LE length prefix + items (sipp/src/lib.rs:56-59) */
    blake2s_ctx c; blake2s_init(&c);
    uint64_t len = (uint64_t)n; uint8_t buf[576];
    const size_t BATCH = 4096;
    uint8_t *tmp = (uint8_t *)malloc(BATCH * 192);
    blake2s_update(&c, (const uint8_t *)&len, 8);
    for (size_t s = 0; s < n; s += BATCH) { size_t e = s + BATCH > n ? n : s + BATCH;
        _Pragma("omp parallel for schedule(static)") for (long i = (long)s; i < (long)e; ++i) orc_ser_g1(&a[i], tmp + (i - s) * 96);
        blake2s_update(&c, tmp, (e - s) * 96); }
    blake2s_update(&c, (const uint8_t *)&len, 8);
    for (size_t s = 0; s < n; s += BATCH) { size_t e = s + BATCH > n ? n : s + BATCH;
        _Pragma("omp parallel for schedule(static)") for (long i = (long)s; i < (long)e; ++i) orc_ser_g2(&b[i], tmp + (i - s) * 192);
        blake2s_update(&c, tmp, (e - s) * 192); }
    blake2s_update(&c, (const uint8_t *)&len, 8);
    for (size_t s = 0; s < n; s += BATCH) { size_t e = s + BATCH > n ? n : s + BATCH;
        for (size_t i = s; i < e; ++i) orc_ser_fr(&r[i], tmp + (i - s) * 32);
        blake2s_update(&c, tmp, (e - s) * 32); }
    orc_ser_gt(value, buf); blake2s_update(&c, buf, 576);
    blake2s_final(&c, digest); free(tmp);
}
ORC_API int orc_sipp_seed_digest(const g1a_t *a, const g2a_t *b, const fr_t *r, size_t n, const fp12_t *value, uint8_t digest[32]) {
    sipp_seed_digest(a, b, r, n, value, digest); return 0;
}
static fr_t sipp_challenge(fsrng_t *rng, const fp12_t *zl, const fp12_t *zr) {
    uint8_t buf[1152]; orc_ser_gt(zl, buf); orc_ser_gt(zr, buf + 576);     /* sipp/src/lib.rs:80-84 */
    fsrng_absorb(rng, buf, 1152);
    uint64_t lo, hi; fsrng_next_u128(rng, &lo, &hi);                          /* :85 */
    return fr_from_u128(lo, hi);
}

/* SIPP::prove (sipp/src/lib.rs:42-106).  proof: 2*log2(n) GT elements (z_l, z_r per round); challenges (optional): log2(n) Fr.
 * returns 0 ok, 1 length mismatch (caller), 2 not a power of two */
ORC_API int orc_sipp_prove(const g1a_t *a_in, const g2a_t *b_in, const fr_t *r, size_t n, const fp12_t *value, fp12_t *proof, fr_t *challenges) {
    if (n == 0 || (n & (n - 1))) return 2;
    uint8_t digest[32]; sipp_seed_digest(a_in, b_in, r, n, value, digest);
    fsrng_t rng; fsrng_from_digest(&rng, digest);
    g1a_t *a = (g1a_t *)malloc(n * sizeof(g1a_t)); g2a_t *b = (g2a_t *)malloc(n * sizeof(g2a_t));
    orc_scale_g1_a(a_in, r, n, a);                                   /* :61-66 */
    memcpy(b, b_in, n * sizeof(g2a_t));
    size_t length = n, round = 0;
    while (length != 1) {
        length /= 2;
        const g1a_t *a_l = a, *a_r = a + length; const g2a_t *b_l = b, *b_r = b + length;
        fp12_t z_l, z_r;
        orc_pairing_product_a(a_r, b_l, length, &z_l);                  /* :77 */
        orc_pairing_product_a(a_l, b_r, length, &z_r);                  /* :78 */
        proof[2 * round] = z_l; proof[2 * round + 1] = z_r;
        fr_t x = sipp_challenge(&rng, &z_l, &z_r);
        if (challenges) challenges[round] = x;
        fr_t x_inv; fr_inv(&x_inv, &x);                                  /* :94 */
        g1a_t *na = (g1a_t *)malloc(length * sizeof(g1a_t)); g2a_t *nb = (g2a_t *)malloc(length * sizeof(g2a_t));
        orc_fold_g1_a(a_r, a_l, length, &x, na);                         /* :87-92 */
        orc_fold_g2_a(b_r, b_l, length, &x_inv, nb);                     /* :95-100 */
        memcpy(a, na, length * sizeof(g1a_t)); memcpy(b, nb, length * sizeof(g2a_t)); free(na); free(nb);
        ++round;
    }
    free(a); free(b); return 0;
}

/* GT "scalar multiplication" = exponentiation in Fp12 (PairingOutput * Fr) */
static void gt_pow(fp12_t *r, const fp12_t *a, const fr_t *k_mont) {
    fr_t k; fr_from_mont(&k, k_mont); fp12_t acc = fp12_one();
    for (int i = 255; i >= 0; --i) { fp12_sqr(&acc, &acc); if ((k.l[i >> 6] >> (i & 63)) & 1) fp12_mul(&acc, &acc, a); }
    *r = acc;
}
ORC_API int orc_gt_pow(const fp12_t *a, const fr_t *k, fp12_t *out) { gt_pow(out, a, k); return 0; }
ORC_API int orc_gt_mul(const fp12_t *a, const fp12_t *b, fp12_t *out) { fp12_mul(out, a, b); return 0; }

/* SIPP::verify (sipp/src/lib.rs:109-180); returns 1 accept / 0 reject / <0 error */
ORC_API int orc_sipp_verify(const g1a_t *a, const g2a_t *b, const fr_t *r, size_t n, const fp12_t *claimed, const fp12_t *proof, size_t proof_len) {
    if (n < 2 || (n & (n - 1))) return -2;
    size_t lg = 0; while (((size_t)1 << lg) < n) ++lg;
    if (proof_len != lg) return -3;
    uint8_t digest[32]; sipp_seed_digest(a, b, r, n, claimed, digest);
    fsrng_t rng; fsrng_from_digest(&rng, digest);
    fr_t *xs = (fr_t *)malloc(lg * sizeof(fr_t)), *xinv = (fr_t *)malloc(lg * sizeof(fr_t));
    for (size_t j = 0; j < lg; ++j) { xs[j] = sipp_challenge(&rng, &proof[2 * j], &proof[2 * j + 1]); fr_inv(&xinv[j], &xs[j]); }
    fp12_t zp = *claimed;                                                 /* GT is written multiplicatively here */
    for (size_t j = 0; j < lg; ++j) { fp12_t t; gt_pow(&t, &proof[2 * j], &xs[j]); fp12_mul(&zp, &zp, &t); gt_pow(&t, &proof[2 * j + 1], &xinv[j]); fp12_mul(&zp, &zp, &t); }
    fr_t *s = (fr_t *)malloc(n * sizeof(fr_t)), *si = (fr_t *)malloc(n * sizeof(fr_t));
    for (size_t i = 0; i < n; ++i) { s[i] = fr_one(); si[i] = fr_one(); }
    for (size_t j = 0; j < lg; ++j) for (size_t i = 0; i < n; ++i) if (i & ((size_t)1 << (lg - j - 1))) { fr_mul(&s[i], &s[i], &xs[j]); fr_mul(&si[i], &si[i], &xinv[j]); }
    for (size_t i = 0; i < n; ++i) fr_mul(&s[i], &s[i], &r[i]);
    g1j_t ap; g2j_t bp; g1_msm_affine(&ap, a, s, n); g2_msm_affine(&bp, b, si, n);       /* :174-175 */
    g1a_t apa; g2a_t bpa; g1_to_affine(&apa, &ap); g2_to_affine(&bpa, &bp);
    fp12_t e; orc_pairing_product_a(&apa, &bpa, 1, &e);
    int ok = fp12_eq(&e, &zp);
    free(xs); free(xinv); free(s); free(si); return ok;
}

/* ================================================================== small utilities for tests and synthetic inputs */
ORC_API void orc_g1_generator(g1a_t *out) { *out = g1_generator(); }
ORC_API void orc_g2_generator(g2a_t *out) { *out = g2_generator(); }
ORC_API void orc_fr_from_u64x4(const uint64_t v[4], fr_t *out) { fr_t t; memcpy(t.l, v, 32); fr_to_mont(out, &t); }   /* v < r canonical */
ORC_API void orc_fr_to_u64x4(const fr_t *a, uint64_t v[4]) { fr_t t; fr_from_mont(&t, a); memcpy(v, t.l, 32); }
ORC_API void orc_fp_from_u64x6(const uint64_t v[6], fp_t *out) { fp_t t; memcpy(t.l, v, 48); fp_to_mont(out, &t); }
ORC_API void orc_fp_to_u64x6(const fp_t *a, uint64_t v[6]) { fp_t t; fp_from_mont(&t, a); memcpy(v, t.l, 48); }
ORC_API void orc_fr_inv(const fr_t *a, fr_t *out) { fr_inv(out, a); }
ORC_API void orc_fr_mul(const fr_t *a, const fr_t *b, fr_t *out) { fr_mul(out, a, b); }
ORC_API void orc_g1_mul_a(const g1a_t *p, const fr_t *k, g1a_t *out) { g1j_t t; g1a_mul(&t, p, k); g1_to_affine(out, &t); }
ORC_API void orc_g2_mul_a(const g2a_t *p, const fr_t *k, g2a_t *out) { g2j_t t; g2a_mul(&t, p, k); g2_to_affine(out, &t); }
ORC_API void orc_g1_to_affine(const g1j_t *p, g1a_t *out) { g1_to_affine(out, p); }
ORC_API void orc_g2_to_affine(const g2j_t *p, g2a_t *out) { g2_to_affine(out, p); }
ORC_API void orc_g1_add_j(const g1j_t *a, const g1j_t *b, g1j_t *out) { g1j_add(out, a, b); }
ORC_API void orc_g2_add_j(const g2j_t *a, const g2j_t *b, g2j_t *out) { g2j_add(out, a, b); }
ORC_API void orc_fp12_sqr(const fp12_t *a, fp12_t *out) { fp12_sqr(out, a); }
ORC_API void orc_fp12_cyclotomic_sqr(const fp12_t *a, fp12_t *out) { fp12_cyclotomic_sqr(out, a); }
ORC_API void orc_fp12_inv(const fp12_t *a, fp12_t *out) { fp12_inv(out, a); }
ORC_API void orc_fp12_frobenius(const fp12_t *a, int k, fp12_t *out) { fp12_frobenius(out, a, k); }
ORC_API void orc_fp12_one(fp12_t *out) { *out = fp12_one(); }
ORC_API void orc_blake2s(const uint8_t *in, size_t n, uint8_t out[32]) { blake2s(in, n, out); }
ORC_API void orc_blake2b(const uint8_t *in, size_t n, uint8_t out[64]) { blake2b(in, n, out); }
ORC_API void orc_chacha20_block(const uint8_t key[32], uint64_t counter, uint8_t out[64]) { chacha20_block(key, counter, out); }

/* SplitMix64 stream shared with the product's synthetic-input generator */
static uint64_t splitmix64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
/* scalars: 4 draws -> 254-bit canonical integer (top two bits cleared, always < r) -> Montgomery form */
ORC_API void orc_gen_scalars(uint64_t seed, size_t n, fr_t *out) {
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) { uint64_t v[4]; for (int j = 0; j < 4; ++j) v[j] = splitmix64(&s); v[3] &= 0x3fffffffffffffffull; orc_fr_from_u64x4(v, &out[i]); }
}
/* points: out[i] = (start + i) * G, start >= 1, by repeated mixed addition + one batch normalisation per block */
ORC_API void orc_gen_g1(uint64_t start, size_t n, g1a_t *out) {
    if (!n) return; g1a_t g = g1_generator();
    int T = orc_threads(); size_t chunk = (n + T - 1) / T;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)((n + chunk - 1) / chunk); ++c) {
        size_t s = c * chunk, e = s + chunk > n ? n : s + chunk;
        uint64_t k[4] = { start + s, 0, 0, 0 }; fr_t km; orc_fr_from_u64x4(k, &km);
        g1j_t *tmp = (g1j_t *)malloc((e - s) * sizeof(g1j_t));
        g1a_mul(&tmp[0], &g, &km);
        for (size_t i = 1; i < e - s; ++i) g1j_add_affine(&tmp[i], &tmp[i - 1], &g);
        g1_normalize_batch(out + s, tmp, e - s); free(tmp);
    }
}
ORC_API void orc_gen_g2(uint64_t start, size_t n, g2a_t *out) {
    if (!n) return; g2a_t g = g2_generator();
    int T = orc_threads(); size_t chunk = (n + T - 1) / T;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)((n + chunk - 1) / chunk); ++c) {
        size_t s = c * chunk, e = s + chunk > n ? n : s + chunk;
        uint64_t k[4] = { start + s, 0, 0, 0 }; fr_t km; orc_fr_from_u64x4(k, &km);
        g2j_t *tmp = (g2j_t *)malloc((e - s) * sizeof(g2j_t));
        g2a_mul(&tmp[0], &g, &km);
        for (size_t i = 1; i < e - s; ++i) g2j_add_affine(&tmp[i], &tmp[i - 1], &g);
        g2_normalize_batch(out + s, tmp, e - s); free(tmp);
    }
}
/* Jacobian re-randomisation: (x, y) -> (x z^2, y z^3, z) with z from SplitMix64 (exercises normalize_batch) */
ORC_API void orc_jacobian_blind_g1(const g1a_t *in, size_t n, uint64_t seed, g1j_t *out) {
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        if (g1a_is_inf(&in[i])) { out[i] = g1j_inf(); continue; }
        uint64_t v[6]; for (int j = 0; j < 6; ++j) v[j] = splitmix64(&s); v[5] &= 0x0fffffffffffffffull; if (!(v[0] | v[1])) v[0] = 1;
        fp_t z, z2, z3; orc_fp_from_u64x6(v, &z); fp_sqr(&z2, &z); fp_mul(&z3, &z2, &z);
        fp_mul(&out[i].x, &in[i].x, &z2); fp_mul(&out[i].y, &in[i].y, &z3); out[i].z = z;
    }
}
ORC_API void orc_jacobian_blind_g2(const g2a_t *in, size_t n, uint64_t seed, g2j_t *out) {
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        if (g2a_is_inf(&in[i])) { out[i] = g2j_inf(); continue; }
        uint64_t v[6]; fp2_t z, z2, z3;
        for (int j = 0; j < 6; ++j) v[j] = splitmix64(&s); v[5] &= 0x0fffffffffffffffull; orc_fp_from_u64x6(v, &z.c0);
        for (int j = 0; j < 6; ++j) v[j] = splitmix64(&s); v[5] &= 0x0fffffffffffffffull; if (!(v[0] | v[1])) v[0] = 1; orc_fp_from_u64x6(v, &z.c1);
        fp2_sqr(&z2, &z); fp2_mul(&z3, &z2, &z);
        fp2_mul(&out[i].x, &in[i].x, &z2); fp2_mul(&out[i].y, &in[i].y, &z3); out[i].z = z;
    }
}

/* ================================================================== GIPA, TIPP instantiation
 * GIPA<PairingInnerProduct, AFGHOCommitmentG1, AFGHOCommitmentG2, IdentityCommitment<GT, Fr>, Blake2b>
 * as in the reference's own test (ip_proofs/src/gipa.rs:470-497).  Vectors stay projective between rounds
 * (gipa.rs:262-290), every commitment re-normalises its inputs (inner_products/src/lib.rs:80-81). */
static fr_t gipa_challenge(const fr_t *prev /* NULL = Default */, const fp12_t com[6], fr_t *c_inv_out) {
    /* gipa.rs:235-258: nonce (usize, big-endian 8 B) || transcript || com_1.{0,1,2} || com_2.{0,1,2}; IdentityOutput(Vec<GT>) carries a u64 length */
    uint64_t nonce = 0;
    for (;;) {
        uint8_t buf[8 + 32 + 6 * 576 + 16], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        fr_t zero = fr_zero(); orc_ser_fr(prev ? prev : &zero, p); p += 32;
        for (int k = 0; k < 6; ++k) {
            if (k == 2 || k == 5) { uint64_t one = 1; memcpy(p, &one, 8); p += 8; }
            orc_ser_gt(&com[k], p); p += 576;
        }
        uint8_t dig[64]; blake2b(buf, (size_t)(p - buf), dig);
        uint64_t hi = 0, lo = 0;                                   /* u128::from_be_bytes(digest[0..16]) */
        for (int i = 0; i < 8; ++i) { hi = (hi << 8) | dig[i]; lo = (lo << 8) | dig[8 + i]; }
        fr_t c128 = fr_from_u128(lo, hi);
        if (!fr_is_zero(&c128)) { fr_t inv; fr_inv(&inv, &c128); *c_inv_out = c128; return inv; }   /* (c, c_inv) := (c128^-1, c128): names swapped, gipa.rs:252-256 */
        ++nonce;
    }
}

/* GIPA::_prove (gipa.rs:181-312).  Outputs in ROUND order (the proof stores them reversed, gipa.rs:298-299):
 * com_steps[round][6] = (com_1.0, com_1.1, com_1.2[0], com_2.0, com_2.1, com_2.2[0]); transcript[round] = c. */
ORC_API int orc_gipa_tipp_prove(const g1j_t *m_a_in, const g2j_t *m_b_in, const g2j_t *ck_a_in, const g1j_t *ck_b_in, size_t n,
                                fp12_t *com_steps, fr_t *transcript, g1j_t *base_a, g2j_t *base_b, g2j_t *ck_base_a, g1j_t *ck_base_b) {
    if (n == 0 || (n & (n - 1))) return 2;
    g1j_t *m_a = (g1j_t *)malloc(n * sizeof(g1j_t)), *ck_b = (g1j_t *)malloc(n * sizeof(g1j_t));
    g2j_t *m_b = (g2j_t *)malloc(n * sizeof(g2j_t)), *ck_a = (g2j_t *)malloc(n * sizeof(g2j_t));
    memcpy(m_a, m_a_in, n * sizeof(g1j_t)); memcpy(m_b, m_b_in, n * sizeof(g2j_t)); memcpy(ck_a, ck_a_in, n * sizeof(g2j_t)); memcpy(ck_b, ck_b_in, n * sizeof(g1j_t));
    size_t len = n, round = 0;
    while (len > 1) {
        size_t split = len / 2;
        const g1j_t *m_a_1 = m_a + split, *m_a_2 = m_a, *ck_b_1 = ck_b + split, *ck_b_2 = ck_b;
        const g2j_t *ck_a_1 = ck_a, *ck_a_2 = ck_a + split, *m_b_1 = m_b, *m_b_2 = m_b + split;
        fp12_t *com = com_steps + 6 * round;
        orc_pairing_product_j(m_a_1, split, ck_a_1, split, &com[0]);      /* LMC::commit(ck_a_1, m_a_1) = IP(m, k)   afgho16/mod.rs:31 */
        orc_pairing_product_j(ck_b_1, split, m_b_1, split, &com[1]);      /* RMC::commit(ck_b_1, m_b_1) = IP(k, m)   afgho16/mod.rs:46 */
        orc_pairing_product_j(m_a_1, split, m_b_1, split, &com[2]);       /* IP::inner_product(m_a_1, m_b_1) */
        orc_pairing_product_j(m_a_2, split, ck_a_2, split, &com[3]);
        orc_pairing_product_j(ck_b_2, split, m_b_2, split, &com[4]);
        orc_pairing_product_j(m_a_2, split, m_b_2, split, &com[5]);
        fr_t c_inv, c = gipa_challenge(round ? &transcript[round - 1] : NULL, com, &c_inv);
        g1j_t *na = (g1j_t *)malloc(split * sizeof(g1j_t)), *nkb = (g1j_t *)malloc(split * sizeof(g1j_t));
        g2j_t *nb = (g2j_t *)malloc(split * sizeof(g2j_t)), *nka = (g2j_t *)malloc(split * sizeof(g2j_t));
        orc_fold_g1_j(m_a_1, m_a_2, split, &c, na);                        /* gipa.rs:262-266 */
        orc_fold_g2_j(m_b_2, m_b_1, split, &c_inv, nb);                    /* :270-274 */
        orc_fold_g2_j(ck_a_2, ck_a_1, split, &c_inv, nka);                 /* :278-282 */
        orc_fold_g1_j(ck_b_1, ck_b_2, split, &c, nkb);                     /* :286-290 */
        memcpy(m_a, na, split * sizeof(g1j_t)); memcpy(m_b, nb, split * sizeof(g2j_t)); memcpy(ck_a, nka, split * sizeof(g2j_t)); memcpy(ck_b, nkb, split * sizeof(g1j_t));
        free(na); free(nb); free(nka); free(nkb);
        transcript[round] = c; ++round; len = split;
    }
    *base_a = m_a[0]; *base_b = m_b[0]; *ck_base_a = ck_a[0]; *ck_base_b = ck_b[0];
    free(m_a); free(m_b); free(ck_a); free(ck_b);
    return 0;
}

/* GIPA::verify (gipa.rs:135-160, 322-415) for the same instantiation; inputs in ROUND order as produced above.
 * com = (com_a, com_b, com_t[0]).  returns 1 accept / 0 reject. */
ORC_API int orc_gipa_tipp_verify(const g2j_t *ck_a, const g1j_t *ck_b, size_t n, const fp12_t com[3], const fp12_t *com_steps, size_t rounds,
                                 const g1j_t *base_a, const g2j_t *base_b) {
    if (n == 0 || (n & (n - 1)) || ((size_t)1 << rounds) != n) return -2;
    fp12_t ca = com[0], cb = com[1], ct = com[2];
    fr_t *tr = (fr_t *)malloc((rounds ? rounds : 1) * sizeof(fr_t));
    for (size_t k = 0; k < rounds; ++k) {                                   /* proof.r_commitment_steps.iter().rev() == round order */
        const fp12_t *s = com_steps + 6 * k;
        fr_t c_inv, c = gipa_challenge(k ? &tr[k - 1] : NULL, s, &c_inv);
        fp12_t t1, t2;
        gt_pow(&t1, &s[0], &c); gt_pow(&t2, &s[3], &c_inv); fp12_mul(&ca, &ca, &t1); fp12_mul(&ca, &ca, &t2);   /* gipa.rs:358-360 */
        gt_pow(&t1, &s[1], &c); gt_pow(&t2, &s[4], &c_inv); fp12_mul(&cb, &cb, &t1); fp12_mul(&cb, &cb, &t2);
        gt_pow(&t1, &s[2], &c); gt_pow(&t2, &s[5], &c_inv); fp12_mul(&ct, &ct, &t1); fp12_mul(&ct, &ct, &t2);
        tr[k] = c;
    }
    /* _compute_final_commitment_keys (gipa.rs:367-399) on the REVERSED transcript */
    fr_t *ea = (fr_t *)malloc(n * sizeof(fr_t)), *eb = (fr_t *)malloc(n * sizeof(fr_t));
    ea[0] = fr_one(); eb[0] = fr_one(); size_t cnt = 1;
    for (size_t i = 0; i < rounds; ++i) {
        const fr_t *c = &tr[rounds - 1 - i]; fr_t ci; fr_inv(&ci, c);
        for (size_t j = 0; j < ((size_t)1 << i); ++j) { fr_mul(&ea[cnt + j], &ea[j], &ci); fr_mul(&eb[cnt + j], &eb[j], c); }
        cnt += (size_t)1 << i;
    }
    g2j_t ka; g1j_t kb; orc_msm_g2_j(ck_a, n, ea, n, &ka); orc_msm_g1_j(ck_b, n, eb, n, &kb);
    fp12_t e1, e2, e3;
    orc_pairing_product_j(base_a, 1, &ka, 1, &e1);                          /* LMC::verify([ck_a_base], [a_base], com_a) */
    orc_pairing_product_j(&kb, 1, base_b, 1, &e2);                          /* RMC::verify([ck_b_base], [b_base], com_b) */
    orc_pairing_product_j(base_a, 1, base_b, 1, &e3);                       /* IPC: [IP(a_base, b_base)] == com_t */
    int ok = fp12_eq(&e1, &ca) && fp12_eq(&e2, &cb) && fp12_eq(&e3, &ct);
    free(tr); free(ea); free(eb);
    return ok;
}

#include "tipa.h"       /* TIPA / aggregation restatements (both curves: nothing in them depends on the curve beyond field.h / curve.h / pairing.h and the point encodings above) */
