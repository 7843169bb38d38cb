/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED: the reference
 * (arkworks-rs/ripp) cannot be built or imported here and its tests hold no golden vectors, so
 * this restatement is pinned only by tests/model (an independent big-integer model) and by
 * algebraic laws.  Nothing under ripp_amd/ may include, link or call this.
 *
 * BLS12-381 field tower, plain C, 6 x u64 / 4 x u64 little-endian Montgomery limbs -- the same
 * in-memory representation ark-ff 0.4 uses for Fp384 / Fp256 (un-vendored dependency of
 * /root/reference, `ark-ff = "0.4"`, inner_products/Cargo.toml:19-22).
 * Tower (ark-bls12-381 0.4): Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-(1+u)), Fp12 = Fp6[w]/(w^2-v).
 */
#ifndef RIPP_ORACLE_FIELD_H
#define RIPP_ORACLE_FIELD_H
#include <stdint.h>
#include <string.h>
#ifdef ORC_BLS12_377
/* BLS12-377 build (liboracle for the curve of the reference's own SIPP test, sipp/src/lib.rs:229): Fp2 = Fp[u]/(u^2+5),
 * Fp6 = Fp2[v]/(v^3-u), Fp12 = Fp6[w]/(w^2-v)  (ark-bls12-377 0.4) -- only fp2_mul / fp2_sqr / fp2_inv / fp2_mul_xi differ. */
#include "params_377.h"
#else
#include "params.h"
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp_t;
typedef struct { uint64_t l[4]; } fr_t;
typedef struct { fp_t c0, c1; } fp2_t;
typedef struct { fp2_t c0, c1, c2; } fp6_t;
typedef struct { fp6_t c0, c1; } fp12_t;

/* ------------------------------------------------------------------ generic N-limb Montgomery */
#define ORC_INLINE static inline __attribute__((always_inline))
#ifndef ORC_MULATTR
#define ORC_MULATTR static __attribute__((noinline))
#endif

ORC_INLINE uint64_t adc64(uint64_t a, uint64_t b, uint64_t *c) { u128 s = (u128)a + b + *c; *c = (uint64_t)(s >> 64); return (uint64_t)s; }
ORC_INLINE uint64_t sbb64(uint64_t a, uint64_t b, uint64_t *bo) { u128 d = (u128)a - b - *bo; *bo = (uint64_t)(d >> 64) & 1; return (uint64_t)d; }

#define DEFINE_MONT(NAME, T, N, MOD, INV)                                                         \
ORC_INLINE void NAME##_reduce_once(T *a) {                                                        \
    uint64_t d[N], bo = 0;                                                                        \
    for (int i = 0; i < N; ++i) d[i] = sbb64(a->l[i], MOD[i], &bo);                               \
    if (!bo) for (int i = 0; i < N; ++i) a->l[i] = d[i];                                          \
}                                                                                                 \
ORC_INLINE void NAME##_add(T *r, const T *a, const T *b) {                                        \
    uint64_t c = 0; for (int i = 0; i < N; ++i) r->l[i] = adc64(a->l[i], b->l[i], &c);            \
    NAME##_reduce_once(r);                                                                        \
}                                                                                                 \
ORC_INLINE void NAME##_sub(T *r, const T *a, const T *b) {                                        \
    uint64_t bo = 0; for (int i = 0; i < N; ++i) r->l[i] = sbb64(a->l[i], b->l[i], &bo);          \
    if (bo) { uint64_t c = 0; for (int i = 0; i < N; ++i) r->l[i] = adc64(r->l[i], MOD[i], &c); } \
}                                                                                                 \
ORC_INLINE int NAME##_is_zero(const T *a) { uint64_t o = 0; for (int i = 0; i < N; ++i) o |= a->l[i]; return o == 0; } \
ORC_INLINE int NAME##_eq(const T *a, const T *b) { uint64_t o = 0; for (int i = 0; i < N; ++i) o |= a->l[i] ^ b->l[i]; return o == 0; } \
ORC_INLINE void NAME##_neg(T *r, const T *a) {                                                    \
    if (NAME##_is_zero(a)) { *r = *a; return; }                                                   \
    uint64_t bo = 0; for (int i = 0; i < N; ++i) r->l[i] = sbb64(MOD[i], a->l[i], &bo);           \
}                                                                                                 \
ORC_INLINE void NAME##_dbl(T *r, const T *a) { NAME##_add(r, a, a); }                             \
ORC_MULATTR void NAME##_mul(T *r, const T *a, const T *b) {                                        \
    uint64_t t[N + 2]; memset(t, 0, sizeof t);                                                    \
    for (int i = 0; i < N; ++i) {                                                                 \
        uint64_t c = 0;                                                                           \
        for (int j = 0; j < N; ++j) { u128 s = (u128)a->l[j] * b->l[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); } \
        u128 s2 = (u128)t[N] + c; t[N] = (uint64_t)s2; t[N + 1] = (uint64_t)(s2 >> 64);           \
        uint64_t m = t[0] * INV;                                                                  \
        u128 s = (u128)m * MOD[0] + t[0]; c = (uint64_t)(s >> 64);                                \
        for (int j = 1; j < N; ++j) { s = (u128)m * MOD[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); } \
        s2 = (u128)t[N] + c; t[N - 1] = (uint64_t)s2; t[N] = t[N + 1] + (uint64_t)(s2 >> 64);     \
    }                                                                                             \
    for (int i = 0; i < N; ++i) r->l[i] = t[i];                                                   \
    NAME##_reduce_once(r);                                                                        \
}                                                                                                 \
ORC_INLINE void NAME##_sqr(T *r, const T *a) { NAME##_mul(r, a, a); }                             \
static void NAME##_pow(T *r, const T *a, const uint64_t *e, int elimbs, const T *one) {           \
    T acc = *one; int started = 0;                                                                \
    for (int i = elimbs * 64 - 1; i >= 0; --i) {                                                  \
        if (started) NAME##_sqr(&acc, &acc);                                                      \
        if ((e[i >> 6] >> (i & 63)) & 1) { if (started) NAME##_mul(&acc, &acc, a); else { acc = *a; started = 1; } } \
    }                                                                                             \
    *r = acc;                                                                                     \
}

DEFINE_MONT(fp, fp_t, 6, FP_P, FP_INV)
DEFINE_MONT(fr, fr_t, 4, FR_R, FR_INV)

ORC_INLINE fp_t fp_one(void) { fp_t r; memcpy(r.l, FP_R1, 48); return r; }
ORC_INLINE fp_t fp_zero(void) { fp_t r; memset(r.l, 0, 48); return r; }
ORC_INLINE fr_t fr_one(void) { fr_t r; memcpy(r.l, FR_R1, 32); return r; }
ORC_INLINE fr_t fr_zero(void) { fr_t r; memset(r.l, 0, 32); return r; }
static void fp_inv(fp_t *r, const fp_t *a) { fp_t one = fp_one(); fp_pow(r, a, FP_P_MINUS_2, 6, &one); }
static void fr_inv(fr_t *r, const fr_t *a) { fr_t one = fr_one(); fr_pow(r, a, FR_R_MINUS_2, 4, &one); }
ORC_INLINE void fp_to_mont(fp_t *r, const fp_t *a) { fp_t r2; memcpy(r2.l, FP_R2, 48); fp_mul(r, a, &r2); }
ORC_INLINE void fp_from_mont(fp_t *r, const fp_t *a) { fp_t one; memset(&one, 0, sizeof one); one.l[0] = 1; fp_mul(r, a, &one); }
ORC_INLINE void fr_to_mont(fr_t *r, const fr_t *a) { fr_t r2; memcpy(r2.l, FR_R2, 32); fr_mul(r, a, &r2); }
ORC_INLINE void fr_from_mont(fr_t *r, const fr_t *a) { fr_t one; memset(&one, 0, sizeof one); one.l[0] = 1; fr_mul(r, a, &one); }
/* Fr::from(u128) -- sipp/src/lib.rs:85 `u128::rand(&mut rng).into()` */
ORC_INLINE fr_t fr_from_u128(uint64_t lo, uint64_t hi) { fr_t t; t.l[0] = lo; t.l[1] = hi; t.l[2] = t.l[3] = 0; fr_t r; fr_to_mont(&r, &t); return r; }

/* ------------------------------------------------------------------ Fp2 */
ORC_INLINE void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
ORC_INLINE void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
ORC_INLINE void fp2_neg(fp2_t *r, const fp2_t *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
ORC_INLINE void fp2_dbl(fp2_t *r, const fp2_t *a) { fp_dbl(&r->c0, &a->c0); fp_dbl(&r->c1, &a->c1); }
ORC_INLINE void fp2_conj(fp2_t *r, const fp2_t *a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
ORC_INLINE int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
ORC_INLINE int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
ORC_INLINE fp2_t fp2_zero(void) { fp2_t r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
ORC_INLINE fp2_t fp2_one(void) { fp2_t r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
#ifdef ORC_BLS12_377
ORC_INLINE void fp_mul5(fp_t *r, const fp_t *a) { fp_t t; fp_dbl(&t, a); fp_dbl(&t, &t); fp_add(r, &t, a); }
ORC_INLINE void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b) {      /* u^2 = -5: c0 = a0 b0 - 5 a1 b1 */
    fp_t t0, t1, s0, s1, m, t5;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&s0, &a->c0, &a->c1); fp_add(&s1, &b->c0, &b->c1); fp_mul(&m, &s0, &s1);
    fp_mul5(&t5, &t1); fp_sub(&r->c0, &t0, &t5); fp_sub(&m, &m, &t0); fp_sub(&r->c1, &m, &t1);
}
ORC_INLINE void fp2_sqr(fp2_t *r, const fp2_t *a) {                       /* c0 = a0^2 - 5 a1^2 = (a0+a1)(a0-5a1) + 4 a0 a1 */
    fp_t s, d, m, a5, m4;
    fp_mul5(&a5, &a->c1); fp_add(&s, &a->c0, &a->c1); fp_sub(&d, &a->c0, &a5); fp_mul(&m, &a->c0, &a->c1);
    fp_mul(&s, &s, &d); fp_dbl(&m4, &m); fp_dbl(&m4, &m4); fp_add(&r->c0, &s, &m4); fp_dbl(&r->c1, &m);
}
#else
ORC_INLINE void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b) {
    fp_t t0, t1, s0, s1, m;
    fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&s0, &a->c0, &a->c1); fp_add(&s1, &b->c0, &b->c1); fp_mul(&m, &s0, &s1);
    fp_sub(&r->c0, &t0, &t1); fp_sub(&m, &m, &t0); fp_sub(&r->c1, &m, &t1);
}
ORC_INLINE void fp2_sqr(fp2_t *r, const fp2_t *a) {
    fp_t s, d, m;
    fp_add(&s, &a->c0, &a->c1); fp_sub(&d, &a->c0, &a->c1); fp_mul(&m, &a->c0, &a->c1);
    fp_mul(&r->c0, &s, &d); fp_dbl(&r->c1, &m);
}
#endif
ORC_INLINE void fp2_mul_fp(fp2_t *r, const fp2_t *a, const fp_t *s) { fp_mul(&r->c0, &a->c0, s); fp_mul(&r->c1, &a->c1, s); }
#ifdef ORC_BLS12_377
/* multiply by the Fp6 non-residue xi = u:  (a0 + a1 u) u = -5 a1 + a0 u */
ORC_INLINE void fp2_mul_xi(fp2_t *r, const fp2_t *a) { fp_t t0, t1; fp_mul5(&t0, &a->c1); fp_neg(&t0, &t0); t1 = a->c0; r->c0 = t0; r->c1 = t1; }
static void fp2_inv(fp2_t *r, const fp2_t *a) {                            /* norm = a0^2 + 5 a1^2 */
    fp_t n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_mul5(&t, &t); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&r->c1, &t);
}
#else
/* multiply by the Fp6 non-residue xi = 1 + u */
ORC_INLINE void fp2_mul_xi(fp2_t *r, const fp2_t *a) { fp_t t0, t1; fp_sub(&t0, &a->c0, &a->c1); fp_add(&t1, &a->c0, &a->c1); r->c0 = t0; r->c1 = t1; }
static void fp2_inv(fp2_t *r, const fp2_t *a) {
    fp_t n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&r->c1, &t);
}
#endif

/* ------------------------------------------------------------------ Fp6 */
ORC_INLINE void fp6_add(fp6_t *r, const fp6_t *a, const fp6_t *b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
ORC_INLINE void fp6_sub(fp6_t *r, const fp6_t *a, const fp6_t *b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
ORC_INLINE void fp6_neg(fp6_t *r, const fp6_t *a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
ORC_INLINE fp6_t fp6_zero(void) { fp6_t r; r.c0 = r.c1 = r.c2 = fp2_zero(); return r; }
ORC_INLINE fp6_t fp6_one(void) { fp6_t r = fp6_zero(); r.c0 = fp2_one(); return r; }
ORC_INLINE int fp6_eq(const fp6_t *a, const fp6_t *b) { return fp2_eq(&a->c0, &b->c0) && fp2_eq(&a->c1, &b->c1) && fp2_eq(&a->c2, &b->c2); }
/* multiply by v:  (c0, c1, c2) -> (xi*c2, c0, c1) */
ORC_INLINE void fp6_mul_v(fp6_t *r, const fp6_t *a) { fp2_t t; fp2_mul_xi(&t, &a->c2); r->c2 = a->c1; r->c1 = a->c0; r->c0 = t; }
static void fp6_mul(fp6_t *r, const fp6_t *a, const fp6_t *b) {
    fp2_t v0, v1, v2, t0, t1, t2, x, y;
    fp2_mul(&v0, &a->c0, &b->c0); fp2_mul(&v1, &a->c1, &b->c1); fp2_mul(&v2, &a->c2, &b->c2);
    /* c0 = v0 + xi((a1+a2)(b1+b2) - v1 - v2) */
    fp2_add(&x, &a->c1, &a->c2); fp2_add(&y, &b->c1, &b->c2); fp2_mul(&t0, &x, &y); fp2_sub(&t0, &t0, &v1); fp2_sub(&t0, &t0, &v2); fp2_mul_xi(&t0, &t0); fp2_add(&t0, &t0, &v0);
    /* c1 = (a0+a1)(b0+b1) - v0 - v1 + xi v2 */
    fp2_add(&x, &a->c0, &a->c1); fp2_add(&y, &b->c0, &b->c1); fp2_mul(&t1, &x, &y); fp2_sub(&t1, &t1, &v0); fp2_sub(&t1, &t1, &v1); fp2_mul_xi(&x, &v2); fp2_add(&t1, &t1, &x);
    /* c2 = (a0+a2)(b0+b2) - v0 - v2 + v1 */
    fp2_add(&x, &a->c0, &a->c2); fp2_add(&y, &b->c0, &b->c2); fp2_mul(&t2, &x, &y); fp2_sub(&t2, &t2, &v0); fp2_sub(&t2, &t2, &v2); fp2_add(&t2, &t2, &v1);
    r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
static void fp6_sqr(fp6_t *r, const fp6_t *a) { fp6_mul(r, a, a); }
/* a * (b0 + b1 v)   -- ark-ff Fp6::mul_by_01 */
static void fp6_mul_by_01(fp6_t *r, const fp6_t *a, const fp2_t *b0, const fp2_t *b1) {
    fp2_t v0, v1, t0, t1, t2, x, y;
    fp2_mul(&v0, &a->c0, b0); fp2_mul(&v1, &a->c1, b1);
    fp2_add(&x, &a->c1, &a->c2); fp2_mul(&t0, &x, b1); fp2_sub(&t0, &t0, &v1); fp2_mul_xi(&t0, &t0); fp2_add(&t0, &t0, &v0);
    fp2_add(&x, &a->c0, &a->c1); fp2_add(&y, b0, b1); fp2_mul(&t1, &x, &y); fp2_sub(&t1, &t1, &v0); fp2_sub(&t1, &t1, &v1);
    fp2_add(&x, &a->c0, &a->c2); fp2_mul(&t2, &x, b0); fp2_sub(&t2, &t2, &v0); fp2_add(&t2, &t2, &v1);
    r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
/* a * (b1 v) */
static void fp6_mul_by_1(fp6_t *r, const fp6_t *a, const fp2_t *b1) {
    fp2_t t0, t1, t2;
    fp2_mul(&t0, &a->c2, b1); fp2_mul_xi(&t0, &t0); fp2_mul(&t1, &a->c0, b1); fp2_mul(&t2, &a->c1, b1);
    r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
static void fp6_inv(fp6_t *r, const fp6_t *a) {
    fp2_t A, B, C, t, F;
    fp2_sqr(&A, &a->c0); fp2_mul(&t, &a->c1, &a->c2); fp2_mul_xi(&t, &t); fp2_sub(&A, &A, &t);          /* A = a0^2 - xi a1 a2 */
    fp2_sqr(&B, &a->c2); fp2_mul_xi(&B, &B); fp2_mul(&t, &a->c0, &a->c1); fp2_sub(&B, &B, &t);          /* B = xi a2^2 - a0 a1 */
    fp2_sqr(&C, &a->c1); fp2_mul(&t, &a->c0, &a->c2); fp2_sub(&C, &C, &t);                              /* C = a1^2 - a0 a2   */
    fp2_mul(&F, &a->c2, &B); fp2_mul(&t, &a->c1, &C); fp2_add(&F, &F, &t); fp2_mul_xi(&F, &F);
    fp2_mul(&t, &a->c0, &A); fp2_add(&F, &F, &t); fp2_inv(&F, &F);
    fp2_mul(&r->c0, &A, &F); fp2_mul(&r->c1, &B, &F); fp2_mul(&r->c2, &C, &F);
}

/* ------------------------------------------------------------------ Fp12 */
ORC_INLINE fp12_t fp12_one(void) { fp12_t r; r.c0 = fp6_one(); r.c1 = fp6_zero(); return r; }
ORC_INLINE int fp12_eq(const fp12_t *a, const fp12_t *b) { return fp6_eq(&a->c0, &b->c0) && fp6_eq(&a->c1, &b->c1); }
ORC_INLINE void fp12_conj(fp12_t *r, const fp12_t *a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
static void fp12_mul(fp12_t *r, const fp12_t *a, const fp12_t *b) {
    fp6_t v0, v1, s, t, x;
    fp6_mul(&v0, &a->c0, &b->c0); fp6_mul(&v1, &a->c1, &b->c1);
    fp6_add(&s, &a->c0, &a->c1); fp6_add(&t, &b->c0, &b->c1); fp6_mul(&x, &s, &t); fp6_sub(&x, &x, &v0); fp6_sub(&x, &x, &v1);
    fp6_mul_v(&t, &v1); fp6_add(&r->c0, &v0, &t); r->c1 = x;
}
static void fp12_sqr(fp12_t *r, const fp12_t *a) {
    /* complex squaring: c0 = (a0+a1)(a0+v a1) - v0 - v v0, c1 = 2 v0, v0 = a0 a1 */
    fp6_t v0, s, t, va1;
    fp6_mul(&v0, &a->c0, &a->c1); fp6_mul_v(&va1, &a->c1);
    fp6_add(&s, &a->c0, &a->c1); fp6_add(&t, &a->c0, &va1); fp6_mul(&s, &s, &t);
    fp6_sub(&s, &s, &v0); fp6_mul_v(&t, &v0); fp6_sub(&s, &s, &t);
    fp6_add(&r->c1, &v0, &v0); r->c0 = s;
}
/* f * (c0 + c1 v + c4 v w)  -- ark-ff Fp12::mul_by_014, used by the M-twist line (ark-ec bls12 `ell`) */
static void fp12_mul_by_014(fp12_t *f, const fp2_t *c0, const fp2_t *c1, const fp2_t *c4) {
    fp6_t aa, bb, s, t; fp2_t o;
    fp6_mul_by_01(&aa, &f->c0, c0, c1);
    fp6_mul_by_1(&bb, &f->c1, c4);
    fp2_add(&o, c1, c4);
    fp6_add(&s, &f->c1, &f->c0); fp6_mul_by_01(&s, &s, c0, &o); fp6_sub(&s, &s, &aa); fp6_sub(&s, &s, &bb);
    fp6_mul_v(&t, &bb); fp6_add(&f->c0, &t, &aa); f->c1 = s;
}
/* f * (c0 + (d0 + d1 v) w)  -- ark-ff Fp12::mul_by_034, used by the D-twist line (ark-ec bls12 `ell`, TwistType::D) */
static void fp12_mul_by_034(fp12_t *f, const fp2_t *c0, const fp2_t *d0, const fp2_t *d1) {
    fp6_t a, b, e, t; fp2_t c0d0;
    a.c0 = f->c0.c0; a.c1 = f->c0.c1; a.c2 = f->c0.c2;
    fp2_mul(&a.c0, &f->c0.c0, c0); fp2_mul(&a.c1, &f->c0.c1, c0); fp2_mul(&a.c2, &f->c0.c2, c0);     /* a = f.c0 * c0 */
    fp6_mul_by_01(&b, &f->c1, d0, d1);                                                                /* b = f.c1 * (d0 + d1 v) */
    fp2_add(&c0d0, c0, d0);
    fp6_add(&e, &f->c0, &f->c1); fp6_mul_by_01(&e, &e, &c0d0, d1);                                   /* e = (f.c0 + f.c1)(c0 + d0 + d1 v) */
    fp6_sub(&e, &e, &a); fp6_sub(&e, &e, &b);
    fp6_mul_v(&t, &b); fp6_add(&f->c0, &a, &t); f->c1 = e;
}
static void fp12_inv(fp12_t *r, const fp12_t *a) {
    fp6_t t0, t1; fp6_sqr(&t0, &a->c0); fp6_sqr(&t1, &a->c1); fp6_mul_v(&t1, &t1); fp6_sub(&t0, &t0, &t1); fp6_inv(&t0, &t0);
    fp6_mul(&r->c0, &a->c0, &t0); fp6_mul(&t1, &a->c1, &t0); fp6_neg(&r->c1, &t1);
}
/* p^k-Frobenius, k in {1,2,3}: conjugate each Fp2 coefficient (k odd) and scale the w^i coefficient by xi^(i(p^k-1)/6) */
static void fp12_frobenius(fp12_t *r, const fp12_t *a, int k) {
    /* flat order w^0..w^5 = c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2 */
    fp2_t g[6] = { a->c0.c0, a->c1.c0, a->c0.c1, a->c1.c1, a->c0.c2, a->c1.c2 };
    const uint64_t *C0[3][6] = {
        { FROB1_W0_C0, FROB1_W1_C0, FROB1_W2_C0, FROB1_W3_C0, FROB1_W4_C0, FROB1_W5_C0 },
        { FROB2_W0_C0, FROB2_W1_C0, FROB2_W2_C0, FROB2_W3_C0, FROB2_W4_C0, FROB2_W5_C0 },
        { FROB3_W0_C0, FROB3_W1_C0, FROB3_W2_C0, FROB3_W3_C0, FROB3_W4_C0, FROB3_W5_C0 } };
    const uint64_t *C1[3][6] = {
        { FROB1_W0_C1, FROB1_W1_C1, FROB1_W2_C1, FROB1_W3_C1, FROB1_W4_C1, FROB1_W5_C1 },
        { FROB2_W0_C1, FROB2_W1_C1, FROB2_W2_C1, FROB2_W3_C1, FROB2_W4_C1, FROB2_W5_C1 },
        { FROB3_W0_C1, FROB3_W1_C1, FROB3_W2_C1, FROB3_W3_C1, FROB3_W4_C1, FROB3_W5_C1 } };
    for (int i = 0; i < 6; ++i) {
        fp2_t c, coef;
        if (k & 1) fp2_conj(&c, &g[i]); else c = g[i];
        memcpy(coef.c0.l, C0[k - 1][i], 48); memcpy(coef.c1.l, C1[k - 1][i], 48);
        fp2_mul(&g[i], &c, &coef);
    }
    r->c0.c0 = g[0]; r->c1.c0 = g[1]; r->c0.c1 = g[2]; r->c1.c1 = g[3]; r->c0.c2 = g[4]; r->c1.c2 = g[5];
}
/* Granger-Scott squaring in the cyclotomic subgroup (ark-ff `cyclotomic_square`); valid only after the easy part */
static void fp12_cyclotomic_sqr(fp12_t *r, const fp12_t *a) {
    /* Fp4 squaring helper on pairs (x, y): (x + y s)^2 with s^2 = xi  ->  (x^2 + xi y^2, 2xy) */
#define FP4_SQ(o0, o1, x, y) do { fp2_t t0_, t1_, s_; fp2_sqr(&t0_, (x)); fp2_sqr(&t1_, (y)); fp2_add(&s_, (x), (y)); fp2_sqr(&s_, &s_); \
        fp2_sub(&s_, &s_, &t0_); fp2_sub(&(o1), &s_, &t1_); fp2_mul_xi(&t1_, &t1_); fp2_add(&(o0), &t0_, &t1_); } while (0)
    const fp2_t *z0 = &a->c0.c0, *z4 = &a->c0.c1, *z3 = &a->c0.c2, *z2 = &a->c1.c0, *z1 = &a->c1.c1, *z5 = &a->c1.c2;
    fp2_t t0, t1, t2, t3, t4, t5, x;
    FP4_SQ(t0, t1, z0, z1);
    FP4_SQ(t2, t3, z2, z3);
    FP4_SQ(t4, t5, z4, z5);
    fp12_t o;
    /* z0' = 3 t0 - 2 z0 ; z1' = 3 t1 + 2 z1 */
    fp2_sub(&x, &t0, z0); fp2_dbl(&x, &x); fp2_add(&o.c0.c0, &x, &t0);
    fp2_add(&x, &t1, z1); fp2_dbl(&x, &x); fp2_add(&o.c1.c1, &x, &t1);
    /* z2' = 3 xi t5 + 2 z2 ; z3' = 3 t4 - 2 z3 */
    fp2_t xt5; fp2_mul_xi(&xt5, &t5);
    fp2_add(&x, &xt5, z2); fp2_dbl(&x, &x); fp2_add(&o.c1.c0, &x, &xt5);
    fp2_sub(&x, &t4, z3); fp2_dbl(&x, &x); fp2_add(&o.c0.c2, &x, &t4);
    /* z4' = 3 t2 - 2 z4 ; z5' = 3 t3 + 2 z5 */
    fp2_sub(&x, &t2, z4); fp2_dbl(&x, &x); fp2_add(&o.c0.c1, &x, &t2);
    fp2_add(&x, &t3, z5); fp2_dbl(&x, &x); fp2_add(&o.c1.c2, &x, &t3);
    *r = o;
#undef FP4_SQ
}
#endif
