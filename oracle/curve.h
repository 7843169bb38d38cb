/* ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see field.h header).
 * G1 = E(Fp): y^2 = x^3 + 4,  G2 = E'(Fp2): y^2 = x^3 + 4(1+u); Jacobian projective (x = X/Z^2, y = Y/Z^3)
 * as ark-ec 0.4 `short_weierstrass::Projective`.  Affine infinity is flagged (ark-ec `Affine{infinity}`),
 * in the flat C-ABI layout it is encoded as (0,0).
 */
#ifndef RIPP_ORACLE_CURVE_H
#define RIPP_ORACLE_CURVE_H
#include "field.h"
#include <stdlib.h>

typedef struct { fp_t x, y; } g1a_t;          /* (0,0) == infinity */
typedef struct { fp_t x, y, z; } g1j_t;       /* z == 0 == infinity */
typedef struct { fp2_t x, y; } g2a_t;
typedef struct { fp2_t x, y, z; } g2j_t;

#define DEFINE_CURVE(G, F, FT, AT, JT)                                                              \
ORC_INLINE int G##a_is_inf(const AT *p) { return F##_is_zero(&p->x) && F##_is_zero(&p->y); }        \
ORC_INLINE int G##j_is_inf(const JT *p) { return F##_is_zero(&p->z); }                              \
ORC_INLINE JT G##j_inf(void) { JT r; r.x = F##_one(); r.y = F##_one(); r.z = F##_zero(); return r; } \
ORC_INLINE AT G##a_inf(void) { AT r; r.x = F##_zero(); r.y = F##_zero(); return r; }                \
ORC_INLINE JT G##_from_affine(const AT *p) { JT r; if (G##a_is_inf(p)) return G##j_inf(); r.x = p->x; r.y = p->y; r.z = F##_one(); return r; } \
static void G##j_dbl(JT *r, const JT *p) {                                                          \
    if (G##j_is_inf(p)) { *r = *p; return; }                                                        \
    FT A, B, C, D, E, Fq, t, z3;                                                                    \
    F##_sqr(&A, &p->x); F##_sqr(&B, &p->y); F##_sqr(&C, &B);                                        \
    F##_add(&t, &p->x, &B); F##_sqr(&t, &t); F##_sub(&t, &t, &A); F##_sub(&t, &t, &C); F##_dbl(&D, &t); \
    F##_dbl(&E, &A); F##_add(&E, &E, &A); F##_sqr(&Fq, &E);                                         \
    F##_mul(&z3, &p->y, &p->z); F##_dbl(&z3, &z3);                                                  \
    F##_sub(&t, &Fq, &D); F##_sub(&r->x, &t, &D);                                                   \
    F##_sub(&t, &D, &r->x); F##_mul(&t, &E, &t); F##_dbl(&C, &C); F##_dbl(&C, &C); F##_dbl(&C, &C); \
    F##_sub(&r->y, &t, &C); r->z = z3;                                                              \
}                                                                                                   \
static void G##j_add_affine(JT *r, const JT *p, const AT *q) {                                      \
    if (G##a_is_inf(q)) { *r = *p; return; }                                                        \
    if (G##j_is_inf(p)) { *r = G##_from_affine(q); return; }                                        \
    FT Z1Z1, U2, S2, H, HH, I, J, rr, V, t, t2;                                                     \
    F##_sqr(&Z1Z1, &p->z); F##_mul(&U2, &q->x, &Z1Z1); F##_mul(&S2, &q->y, &p->z); F##_mul(&S2, &S2, &Z1Z1); \
    F##_sub(&H, &U2, &p->x); F##_sub(&rr, &S2, &p->y);                                              \
    if (F##_is_zero(&H)) { if (F##_is_zero(&rr)) { G##j_dbl(r, p); } else { *r = G##j_inf(); } return; } \
    F##_dbl(&rr, &rr); F##_sqr(&HH, &H); F##_dbl(&I, &HH); F##_dbl(&I, &I); F##_mul(&J, &H, &I); F##_mul(&V, &p->x, &I); \
    JT o;                                                                                           \
    F##_sqr(&t, &rr); F##_sub(&t, &t, &J); F##_sub(&t, &t, &V); F##_sub(&o.x, &t, &V);              \
    F##_sub(&t, &V, &o.x); F##_mul(&t, &rr, &t); F##_mul(&t2, &p->y, &J); F##_dbl(&t2, &t2); F##_sub(&o.y, &t, &t2); \
    F##_add(&t, &p->z, &H); F##_sqr(&t, &t); F##_sub(&t, &t, &Z1Z1); F##_sub(&o.z, &t, &HH);        \
    *r = o;                                                                                         \
}                                                                                                   \
static void G##j_add(JT *r, const JT *p, const JT *q) {                                             \
    if (G##j_is_inf(q)) { *r = *p; return; }                                                        \
    if (G##j_is_inf(p)) { *r = *q; return; }                                                        \
    FT Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, t2;                                           \
    F##_sqr(&Z1Z1, &p->z); F##_sqr(&Z2Z2, &q->z); F##_mul(&U1, &p->x, &Z2Z2); F##_mul(&U2, &q->x, &Z1Z1); \
    F##_mul(&S1, &p->y, &q->z); F##_mul(&S1, &S1, &Z2Z2); F##_mul(&S2, &q->y, &p->z); F##_mul(&S2, &S2, &Z1Z1); \
    F##_sub(&H, &U2, &U1); F##_sub(&rr, &S2, &S1);                                                  \
    if (F##_is_zero(&H)) { if (F##_is_zero(&rr)) { G##j_dbl(r, p); } else { *r = G##j_inf(); } return; } \
    F##_dbl(&rr, &rr); F##_dbl(&I, &H); F##_sqr(&I, &I); F##_mul(&J, &H, &I); F##_mul(&V, &U1, &I); \
    JT o;                                                                                           \
    F##_sqr(&t, &rr); F##_sub(&t, &t, &J); F##_sub(&t, &t, &V); F##_sub(&o.x, &t, &V);              \
    F##_sub(&t, &V, &o.x); F##_mul(&t, &rr, &t); F##_mul(&t2, &S1, &J); F##_dbl(&t2, &t2); F##_sub(&o.y, &t, &t2); \
    F##_add(&t, &p->z, &q->z); F##_sqr(&t, &t); F##_sub(&t, &t, &Z1Z1); F##_sub(&t, &t, &Z2Z2); F##_mul(&o.z, &t, &H); \
    *r = o;                                                                                         \
}                                                                                                   \
ORC_INLINE void G##j_neg(JT *r, const JT *p) { r->x = p->x; F##_neg(&r->y, &p->y); r->z = p->z; }   \
ORC_INLINE void G##a_neg(AT *r, const AT *p) { r->x = p->x; F##_neg(&r->y, &p->y); }                \
/* variable-base double-and-add over the canonical (non-Montgomery) scalar bits, MSB first */       \
static void G##j_mul(JT *r, const JT *p, const fr_t *k_mont) {                                      \
    fr_t k; fr_from_mont(&k, k_mont);                                                               \
    JT acc = G##j_inf();                                                                            \
    for (int i = 255; i >= 0; --i) { G##j_dbl(&acc, &acc); if ((k.l[i >> 6] >> (i & 63)) & 1) G##j_add(&acc, &acc, p); } \
    *r = acc;                                                                                       \
}                                                                                                   \
static void G##a_mul(JT *r, const AT *p, const fr_t *k_mont) {                                      \
    fr_t k; fr_from_mont(&k, k_mont);                                                               \
    JT acc = G##j_inf();                                                                            \
    for (int i = 255; i >= 0; --i) { G##j_dbl(&acc, &acc); if ((k.l[i >> 6] >> (i & 63)) & 1) G##j_add_affine(&acc, &acc, p); } \
    *r = acc;                                                                                       \
}                                                                                                   \
static void G##_to_affine(AT *r, const JT *p) {                                                     \
    if (G##j_is_inf(p)) { *r = G##a_inf(); return; }                                                \
    FT zi, zi2, zi3; F##_inv(&zi, &p->z); F##_sqr(&zi2, &zi); F##_mul(&zi3, &zi2, &zi);             \
    F##_mul(&r->x, &p->x, &zi2); F##_mul(&r->y, &p->y, &zi3);                                       \
}                                                                                                   \
/* CurveGroup::normalize_batch: one shared inversion (Montgomery's trick); z == 0 -> identity */    \
static void G##_normalize_batch(AT *out, const JT *in, size_t n) {                                  \
    if (n == 0) return;                                                                             \
    FT *pre = (FT *)malloc(n * sizeof(FT));                                                         \
    FT acc = F##_one();                                                                             \
    for (size_t i = 0; i < n; ++i) { pre[i] = acc; if (!G##j_is_inf(&in[i])) F##_mul(&acc, &acc, &in[i].z); } \
    FT inv; F##_inv(&inv, &acc);                                                                    \
    for (size_t i = n; i-- > 0;) {                                                                  \
        if (G##j_is_inf(&in[i])) { out[i] = G##a_inf(); continue; }                                 \
        FT zi, zi2, zi3; F##_mul(&zi, &inv, &pre[i]); F##_mul(&inv, &inv, &in[i].z);                \
        F##_sqr(&zi2, &zi); F##_mul(&zi3, &zi2, &zi);                                               \
        AT o; F##_mul(&o.x, &in[i].x, &zi2); F##_mul(&o.y, &in[i].y, &zi3); out[i] = o;             \
    }                                                                                               \
    free(pre);                                                                                      \
}                                                                                                   \
static int G##j_eq(const JT *a, const JT *b) {                                                      \
    if (G##j_is_inf(a) || G##j_is_inf(b)) return G##j_is_inf(a) && G##j_is_inf(b);                  \
    FT za2, zb2, l, r2; F##_sqr(&za2, &a->z); F##_sqr(&zb2, &b->z);                                 \
    F##_mul(&l, &a->x, &zb2); F##_mul(&r2, &b->x, &za2); if (!F##_eq(&l, &r2)) return 0;            \
    F##_mul(&za2, &za2, &a->z); F##_mul(&zb2, &zb2, &b->z);                                         \
    F##_mul(&l, &a->y, &zb2); F##_mul(&r2, &b->y, &za2); return F##_eq(&l, &r2);                    \
}

DEFINE_CURVE(g1, fp, fp_t, g1a_t, g1j_t)
DEFINE_CURVE(g2, fp2, fp2_t, g2a_t, g2j_t)

ORC_INLINE g1a_t g1_generator(void) { g1a_t g; memcpy(g.x.l, G1_GEN_X, 48); memcpy(g.y.l, G1_GEN_Y, 48); return g; }
ORC_INLINE g2a_t g2_generator(void) {
    g2a_t g; memcpy(g.x.c0.l, G2_GEN_X0, 48); memcpy(g.x.c1.l, G2_GEN_X1, 48); memcpy(g.y.c0.l, G2_GEN_Y0, 48); memcpy(g.y.c1.l, G2_GEN_Y1, 48); return g;
}
#endif
