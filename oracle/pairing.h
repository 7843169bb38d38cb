/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see field.h header).
 * Restates ark-ec 0.4 `models::bls12` (un-vendored dependency): G2Prepared line coefficients in
 * homogeneous projective coordinates, `ell` for the M-twist (mul_by_014), multi_miller_loop with one
 * shared squaring per chunk of 4 pairs, and the final exponentiation chain of eprint 2020/875
 * (exponent (p^6-1)(p^2+1)((x-1)^2(x+p)(x^2+p^2-1)+3), i.e. the CUBE of the textbook pairing).
 * Call sites in the reference: inner_products/src/lib.rs:83-88,112,115; sipp/src/lib.rs:196-216.
 */
#ifndef RIPP_ORACLE_PAIRING_H
#define RIPP_ORACLE_PAIRING_H
#include "curve.h"

#ifdef ORC_BLS12_377
#define BLS_X_ABS 0x8508c00000000001ull      /* x > 0: no conjugations (X_IS_NEGATIVE = false) */
#define BLS_X_NEG 0
#define N_ELL 69                              /* 63 doublings + 6 additions */
#else
#define BLS_X_ABS 0xd201000000010000ull      /* |x|, x < 0 */
#define BLS_X_NEG 1
#define N_ELL 68                              /* 63 doublings + 5 additions */
#endif

typedef struct { fp2_t c0, c1, c2; } ell_t;
typedef struct { ell_t ell[N_ELL]; int infinity; } g2prep_t;   /* 68 * 288 B = 19.6 KB per point, as arkworks */

static void g2_prepare(g2prep_t *out, const g2a_t *q) {
    out->infinity = g2a_is_inf(q);
    if (out->infinity) return;
    fp_t two_inv; memcpy(two_inv.l, FP_TWO_INV, 48);
#ifdef ORC_BLS12_377
    fp2_t bcoef; bcoef.c0 = fp_zero(); memcpy(bcoef.c1.l, FP_TWIST_B1, 48);                        /* 1/u = (0, -1/5): D-type twist */
#else
    fp2_t bcoef; { fp_t four; memcpy(four.l, FP_B_G1, 48); bcoef.c0 = four; bcoef.c1 = four; }   /* 4(1+u) */
#endif
    fp2_t X = q->x, Y = q->y, Z = fp2_one();
    int n = 0;
    for (int i = 62; i >= 0; --i) {
        /* doubling step (homogeneous projective) */
        fp2_t a, b, c, e, f, g, h, ii, j, e2, t;
        fp2_mul(&a, &X, &Y); fp2_mul_fp(&a, &a, &two_inv);
        fp2_sqr(&b, &Y); fp2_sqr(&c, &Z);
        fp2_dbl(&t, &c); fp2_add(&t, &t, &c); fp2_mul(&e, &bcoef, &t);
        fp2_dbl(&f, &e); fp2_add(&f, &f, &e);
        fp2_add(&g, &b, &f); fp2_mul_fp(&g, &g, &two_inv);
        fp2_add(&h, &Y, &Z); fp2_sqr(&h, &h); fp2_add(&t, &b, &c); fp2_sub(&h, &h, &t);
        fp2_sub(&ii, &e, &b);
        fp2_sqr(&j, &X);
        fp2_sqr(&e2, &e);
        fp2_sub(&t, &b, &f); fp2_mul(&X, &a, &t);
        fp2_sqr(&t, &g); fp2_t e3; fp2_dbl(&e3, &e2); fp2_add(&e3, &e3, &e2); fp2_sub(&Y, &t, &e3);
        fp2_mul(&Z, &b, &h);
#ifdef ORC_BLS12_377
        fp2_neg(&out->ell[n].c0, &h); fp2_dbl(&t, &j); fp2_add(&out->ell[n].c1, &t, &j); out->ell[n].c2 = ii; ++n;      /* TwistType::D => (-h, 3j, i) */
#else
        out->ell[n].c0 = ii; fp2_dbl(&t, &j); fp2_add(&out->ell[n].c1, &t, &j); fp2_neg(&out->ell[n].c2, &h); ++n;
#endif
        if ((BLS_X_ABS >> i) & 1) {
            /* addition step with the affine Q */
            fp2_t theta, lambda, cc, d, ee, ff, gg, hh, jj, t2;
            fp2_mul(&t, &q->y, &Z); fp2_sub(&theta, &Y, &t);
            fp2_mul(&t, &q->x, &Z); fp2_sub(&lambda, &X, &t);
            fp2_sqr(&cc, &theta); fp2_sqr(&d, &lambda); fp2_mul(&ee, &lambda, &d); fp2_mul(&ff, &Z, &cc); fp2_mul(&gg, &X, &d);
            fp2_dbl(&t, &gg); fp2_add(&hh, &ee, &ff); fp2_sub(&hh, &hh, &t);
            fp2_mul(&X, &lambda, &hh);
            fp2_sub(&t, &gg, &hh); fp2_mul(&t, &theta, &t); fp2_mul(&t2, &ee, &Y); fp2_sub(&Y, &t, &t2);
            fp2_mul(&Z, &Z, &ee);
            fp2_mul(&t, &theta, &q->x); fp2_mul(&t2, &lambda, &q->y); fp2_sub(&jj, &t, &t2);
#ifdef ORC_BLS12_377
            out->ell[n].c0 = lambda; fp2_neg(&out->ell[n].c1, &theta); out->ell[n].c2 = jj; ++n;                      /* TwistType::D => (lambda, -theta, j) */
#else
            out->ell[n].c0 = jj; fp2_neg(&out->ell[n].c1, &theta); out->ell[n].c2 = lambda; ++n;
#endif
        }
    }
}

/* ark-ec bls12 `ell`, TwistType::M: f *= (c0, c1 * px, c2 * py) via mul_by_014 */
ORC_INLINE void ell_apply(fp12_t *f, const ell_t *l, const g1a_t *p) {
#ifdef ORC_BLS12_377
    fp2_t c0, c1; fp2_mul_fp(&c0, &l->c0, &p->y); fp2_mul_fp(&c1, &l->c1, &p->x);       /* TwistType::D: c0 *= p.y, c1 *= p.x, mul_by_034 */
    fp12_mul_by_034(f, &c0, &c1, &l->c2);
#else
    fp2_t c1, c2; fp2_mul_fp(&c2, &l->c2, &p->y); fp2_mul_fp(&c1, &l->c1, &p->x);
    fp12_mul_by_014(f, &l->c0, &c1, &c2);
#endif
}

/* multi_miller_loop over `n` (P, prepared Q) pairs: chunks of 4 share the squaring; infinity pairs are skipped */
static void multi_miller_loop(fp12_t *out, const g1a_t *ps, const g2prep_t *qs, size_t n) {
    fp12_t total = fp12_one();
    size_t idx[4];
    size_t i = 0;
    while (i < n) {
        int k = 0;
        while (i < n && k < 4) { if (!g1a_is_inf(&ps[i]) && !qs[i].infinity) idx[k++] = i; ++i; }
        if (k == 0) break;
        fp12_t f = fp12_one();
        int c = 0;
        for (int b = 62; b >= 0; --b) {
            fp12_sqr(&f, &f);
            for (int j = 0; j < k; ++j) ell_apply(&f, &qs[idx[j]].ell[c], &ps[idx[j]]);
            ++c;
            if ((BLS_X_ABS >> b) & 1) { for (int j = 0; j < k; ++j) ell_apply(&f, &qs[idx[j]].ell[c], &ps[idx[j]]); ++c; }
        }
        fp12_mul(&total, &total, &f);
    }
    if (BLS_X_NEG) fp12_conj(out, &total); else *out = total;    /* x < 0: cyclotomic_inverse_in_place */
}

/* f^|x| by cyclotomic square-and-multiply, then conjugate because x < 0 (ark-ec `exp_by_x`) */
static void fp12_exp_by_x(fp12_t *r, const fp12_t *a) {
    fp12_t acc = *a;                       /* top bit */
    for (int i = 62; i >= 0; --i) { fp12_cyclotomic_sqr(&acc, &acc); if ((BLS_X_ABS >> i) & 1) fp12_mul(&acc, &acc, a); }
    if (BLS_X_NEG) fp12_conj(r, &acc); else *r = acc;
}

static void final_exponentiation(fp12_t *out, const fp12_t *f) {
    fp12_t f1, f2, r, y0, y1, y2;
    fp12_conj(&f1, f); fp12_inv(&f2, f);
    fp12_mul(&r, &f1, &f2);                        /* f^(p^6-1) */
    f2 = r; fp12_frobenius(&r, &r, 2); fp12_mul(&r, &r, &f2);   /* ^(p^2+1) */
    fp12_cyclotomic_sqr(&y0, &r);
    fp12_exp_by_x(&y1, &r);
    fp12_conj(&y2, &r);
    fp12_mul(&y1, &y1, &y2);
    fp12_exp_by_x(&y2, &y1);
    fp12_conj(&y1, &y1);
    fp12_mul(&y1, &y1, &y2);
    fp12_exp_by_x(&y2, &y1);
    fp12_frobenius(&y1, &y1, 1);
    fp12_mul(&y1, &y1, &y2);
    fp12_mul(&r, &r, &y0);
    fp12_exp_by_x(&y0, &y1);
    fp12_exp_by_x(&y2, &y0);
    y0 = y1; fp12_frobenius(&y0, &y0, 2);
    fp12_conj(&y1, &y1);
    fp12_mul(&y1, &y1, &y2);
    fp12_mul(&y1, &y1, &y0);
    fp12_mul(out, &r, &y1);
}
#endif
