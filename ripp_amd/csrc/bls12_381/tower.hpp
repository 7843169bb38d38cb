// BLS12-381 extension tower over the 32-bit-limb Fp of fp.hpp (host + device):
//   Fp2 = Fp[u]/(u^2+1),  Fp6 = Fp2[v]/(v^3-(1+u)),  Fp12 = Fp6[w]/(w^2-v)
// -- the tower ark-bls12-381 0.4 uses (un-vendored dependency of the reference), so that an Fp12 held here
// has the coefficient order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) of `ripp_gt` in include/ripp_hip.h and
// of ark-serialize's `serialize_uncompressed` for `PairingOutput` (sipp/src/lib.rs:80-84).
#pragma once
#include "fp.hpp"
#include "fp_inv.hpp"

namespace ripp {

// ------------------------------------------------------------------ Fp2
struct Fp2 {
    Fp c0, c1;
    RIPP_HD static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    RIPP_HD static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    RIPP_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    RIPP_HD bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
};
RIPP_HD Fp2 add(const Fp2& a, const Fp2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
RIPP_HD Fp2 sub(const Fp2& a, const Fp2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
RIPP_HD Fp2 neg(const Fp2& a) { return {neg(a.c0), neg(a.c1)}; }
RIPP_HD Fp2 dbl(const Fp2& a) { return {dbl(a.c0), dbl(a.c1)}; }
RIPP_HD Fp2 conj(const Fp2& a) { return {a.c0, neg(a.c1)}; }
#if defined(RIPP_BLS12_377)
// BLS12-377: Fp2 = Fp[u]/(u^2 + 5), Fp6 non-residue xi = u  (ark-bls12-377 0.4)
RIPP_HD Fp mul5(const Fp& a) { const Fp t = dbl(dbl(a)); return add(t, a); }
RIPP_MID Fp2 mul(const Fp2& a, const Fp2& b) {           // Karatsuba: c0 = a0 b0 - 5 a1 b1
    const Fp t0 = fmul(a.c0, b.c0), t1 = fmul(a.c1, b.c1);
    const Fp m = fmul(add(a.c0, a.c1), add(b.c0, b.c1));
    return {sub(t0, mul5(t1)), sub(sub(m, t0), t1)};
}
RIPP_MID Fp2 sqr(const Fp2& a) {                          // a0^2 - 5 a1^2 = (a0 + a1)(a0 - 5 a1) + 4 a0 a1
    const Fp m = fmul(a.c0, a.c1);
    return {add(fmul(add(a.c0, a.c1), sub(a.c0, mul5(a.c1))), dbl(dbl(m))), dbl(m)};
}
RIPP_MID Fp2 mul_fp(const Fp2& a, const Fp& s) { return {fmul(a.c0, s), fmul(a.c1, s)}; }
RIPP_HD Fp2 mul_xi(const Fp2& a) { return {neg(mul5(a.c1)), a.c0}; }                 // * u
RIPP_MID Fp2 inv(const Fp2& a) {                          // norm a0^2 + 5 a1^2
    const Fp n = finv(add(fsqr(a.c0), mul5(fsqr(a.c1))));
    return {fmul(a.c0, n), neg(fmul(a.c1, n))};
}
#else
RIPP_MID Fp2 mul(const Fp2& a, const Fp2& b) {           // Karatsuba: 3 Fp products
    const Fp t0 = fmul(a.c0, b.c0), t1 = fmul(a.c1, b.c1);
    const Fp m = fmul(add(a.c0, a.c1), add(b.c0, b.c1));
    return {sub(t0, t1), sub(sub(m, t0), t1)};
}
RIPP_MID Fp2 sqr(const Fp2& a) {                          // (a0+a1)(a0-a1), 2 a0 a1
    const Fp m = fmul(a.c0, a.c1);
    return {fmul(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(m)};
}
RIPP_MID Fp2 mul_fp(const Fp2& a, const Fp& s) { return {fmul(a.c0, s), fmul(a.c1, s)}; }
RIPP_HD Fp2 mul_xi(const Fp2& a) { return {sub(a.c0, a.c1), add(a.c0, a.c1)}; }     // * (1 + u)
RIPP_MID Fp2 inv(const Fp2& a) {
    const Fp n = finv(add(fsqr(a.c0), fsqr(a.c1)));
    return {fmul(a.c0, n), neg(fmul(a.c1, n))};
}
#endif

RIPP_HD Fp2 finv(const Fp2& a) { return inv(a); }
// fmul is what the group law (curve.hpp) and the MSM / fold kernels call.  On the device it is two sum-of-two-products with one
// Montgomery reduction each (fp.hpp::mul2_add): the same limb products as Karatsuba's three multiplications, none of its five
// additions and two calls instead of three -- measured 4-5 % on the G2 folds and MSM.  The pairing tower above keeps Karatsuba
// (`mul`): its kernels are register bound and the 48-register call makes k_line_products 5 % slower.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RIPP_NO_FP2_LAZY) && defined(RIPP_BLS12_377)
RIPP_MID Fp2 fmul(const Fp2& a, const Fp2& b) { return {fmul2_add(a.c0, b.c0, neg(mul5(a.c1)), b.c1), fmul2_add(a.c0, b.c1, a.c1, b.c0)}; }   // a0 b0 + (-5 a1) b1
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(RIPP_NO_FP2_LAZY)
RIPP_MID Fp2 fmul(const Fp2& a, const Fp2& b) { return {fmul2_add(a.c0, b.c0, neg(a.c1), b.c1), fmul2_add(a.c0, b.c1, a.c1, b.c0)}; }
#else
RIPP_HD Fp2 fmul(const Fp2& a, const Fp2& b) { return mul(a, b); }
#endif
RIPP_HD Fp2 fsqr(const Fp2& a) { return sqr(a); }

// ------------------------------------------------------------------ Fp6
struct Fp6 {
    Fp2 c0, c1, c2;
    RIPP_HD static Fp6 zero() { return {Fp2::zero(), Fp2::zero(), Fp2::zero()}; }
    RIPP_HD static Fp6 one() { return {Fp2::one(), Fp2::zero(), Fp2::zero()}; }
    RIPP_HD bool operator==(const Fp6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};
RIPP_HD Fp6 add(const Fp6& a, const Fp6& b) { return {add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2)}; }
RIPP_HD Fp6 sub(const Fp6& a, const Fp6& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2)}; }
RIPP_HD Fp6 neg(const Fp6& a) { return {neg(a.c0), neg(a.c1), neg(a.c2)}; }
RIPP_HD Fp6 mul_v(const Fp6& a) { return {mul_xi(a.c2), a.c0, a.c1}; }
RIPP_MID Fp6 mul(const Fp6& a, const Fp6& b) {            // Karatsuba-3: 6 Fp2 products
    const Fp2 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1), v2 = mul(a.c2, b.c2);
    const Fp2 t0 = add(mul_xi(sub(sub(mul(add(a.c1, a.c2), add(b.c1, b.c2)), v1), v2)), v0);
    const Fp2 t1 = add(sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1), mul_xi(v2));
    const Fp2 t2 = add(sub(sub(mul(add(a.c0, a.c2), add(b.c0, b.c2)), v0), v2), v1);
    return {t0, t1, t2};
}
// a * (b0 + b1 v)
RIPP_MID Fp6 mul_by_01(const Fp6& a, const Fp2& b0, const Fp2& b1) {
    const Fp2 v0 = mul(a.c0, b0), v1 = mul(a.c1, b1);
    const Fp2 t0 = add(mul_xi(sub(mul(add(a.c1, a.c2), b1), v1)), v0);
    const Fp2 t1 = sub(sub(mul(add(a.c0, a.c1), add(b0, b1)), v0), v1);
    const Fp2 t2 = add(sub(mul(add(a.c0, a.c2), b0), v0), v1);
    return {t0, t1, t2};
}
// a * (b1 v)
RIPP_MID Fp6 mul_by_1(const Fp6& a, const Fp2& b1) { return {mul_xi(mul(a.c2, b1)), mul(a.c0, b1), mul(a.c1, b1)}; }
RIPP_FN Fp6 inv(const Fp6& a) {
    const Fp2 A = sub(sqr(a.c0), mul_xi(mul(a.c1, a.c2)));
    const Fp2 B = sub(mul_xi(sqr(a.c2)), mul(a.c0, a.c1));
    const Fp2 C = sub(sqr(a.c1), mul(a.c0, a.c2));
    const Fp2 F = inv(add(mul_xi(add(mul(a.c2, B), mul(a.c1, C))), mul(a.c0, A)));
    return {mul(A, F), mul(B, F), mul(C, F)};
}

// ------------------------------------------------------------------ Fp12
struct Fp12 {
    Fp6 c0, c1;
    RIPP_HD static Fp12 one() { return {Fp6::one(), Fp6::zero()}; }
    RIPP_HD bool operator==(const Fp12& o) const { return c0 == o.c0 && c1 == o.c1; }
};
RIPP_HD Fp12 conj(const Fp12& a) { return {a.c0, neg(a.c1)}; }
RIPP_MID Fp12 mul(const Fp12& a, const Fp12& b) {
    const Fp6 v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
    const Fp6 x = sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), v0), v1);
    return {add(v0, mul_v(v1)), x};
}
RIPP_MID Fp12 sqr(const Fp12& a) {
    const Fp6 v0 = mul(a.c0, a.c1);
    const Fp6 s = sub(sub(mul(add(a.c0, a.c1), add(a.c0, mul_v(a.c1))), v0), mul_v(v0));
    return {s, add(v0, v0)};
}
// f * (c0 + c1 v + c4 v w): the M-twist line element (ark-ff Fp12::mul_by_014, called from ark-ec bls12 `ell`)
RIPP_MID Fp12 mul_by_014(const Fp12& f, const Fp2& c0, const Fp2& c1, const Fp2& c4) {
    const Fp6 aa = mul_by_01(f.c0, c0, c1);
    const Fp6 bb = mul_by_1(f.c1, c4);
    const Fp6 s = sub(sub(mul_by_01(add(f.c1, f.c0), c0, add(c1, c4)), aa), bb);
    return {add(mul_v(bb), aa), s};
}
// f * (c0 + (d0 + d1 v) w): the D-twist line element (ark-ff Fp12::mul_by_034)
RIPP_MID Fp12 mul_by_034(const Fp12& f, const Fp2& c0, const Fp2& d0, const Fp2& d1) {
    const Fp6 a = {mul(f.c0.c0, c0), mul(f.c0.c1, c0), mul(f.c0.c2, c0)};
    const Fp6 b = mul_by_01(f.c1, d0, d1);
    const Fp6 e = sub(sub(mul_by_01(add(f.c0, f.c1), add(c0, d0), d1), a), b);
    return {add(a, mul_v(b)), e};
}
RIPP_FN Fp12 inv(const Fp12& a) {
    const Fp6 t = inv(sub(mul(a.c0, a.c0), mul_v(mul(a.c1, a.c1))));
    return {mul(a.c0, t), neg(mul(a.c1, t))};
}

// Frobenius coefficient tables: FROBk[i] multiplies the w^i coefficient (flat index i = 0..5) after the
// p^k-conjugation.  Derived in tools/gen_params.py as xi^(i (p^k - 1) / 6).
RIPP_FN Fp12 frobenius(const Fp12& a, int k) {
    // flat order w^0..w^5 = c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2
    Fp2 g[6] = {a.c0.c0, a.c1.c0, a.c0.c1, a.c1.c1, a.c0.c2, a.c1.c2};
    for (int i = 0; i < 6; ++i) {
        constexpr uint32_t T[3][6][2][12] = {
            {{RIPP_FROB1_W0_C0, RIPP_FROB1_W0_C1}, {RIPP_FROB1_W1_C0, RIPP_FROB1_W1_C1}, {RIPP_FROB1_W2_C0, RIPP_FROB1_W2_C1},
             {RIPP_FROB1_W3_C0, RIPP_FROB1_W3_C1}, {RIPP_FROB1_W4_C0, RIPP_FROB1_W4_C1}, {RIPP_FROB1_W5_C0, RIPP_FROB1_W5_C1}},
            {{RIPP_FROB2_W0_C0, RIPP_FROB2_W0_C1}, {RIPP_FROB2_W1_C0, RIPP_FROB2_W1_C1}, {RIPP_FROB2_W2_C0, RIPP_FROB2_W2_C1},
             {RIPP_FROB2_W3_C0, RIPP_FROB2_W3_C1}, {RIPP_FROB2_W4_C0, RIPP_FROB2_W4_C1}, {RIPP_FROB2_W5_C0, RIPP_FROB2_W5_C1}},
            {{RIPP_FROB3_W0_C0, RIPP_FROB3_W0_C1}, {RIPP_FROB3_W1_C0, RIPP_FROB3_W1_C1}, {RIPP_FROB3_W2_C0, RIPP_FROB3_W2_C1},
             {RIPP_FROB3_W3_C0, RIPP_FROB3_W3_C1}, {RIPP_FROB3_W4_C0, RIPP_FROB3_W4_C1}, {RIPP_FROB3_W5_C0, RIPP_FROB3_W5_C1}}};
        Fp2 coef;
        for (int j = 0; j < 12; ++j) { coef.c0.l[j] = T[k - 1][i][0][j]; coef.c1.l[j] = T[k - 1][i][1][j]; }
        g[i] = mul((k & 1) ? conj(g[i]) : g[i], coef);
    }
    return {{g[0], g[2], g[4]}, {g[1], g[3], g[5]}};
}

// Granger-Scott squaring, valid in the cyclotomic subgroup (after the easy part of the final exponentiation)
RIPP_HD void fp4_sqr(Fp2& o0, Fp2& o1, const Fp2& x, const Fp2& y) {
    const Fp2 t0 = sqr(x), t1 = sqr(y);
    o1 = sub(sub(sqr(add(x, y)), t0), t1);
    o0 = add(t0, mul_xi(t1));
}
RIPP_FN Fp12 cyclotomic_sqr(const Fp12& a) {
    const Fp2 &z0 = a.c0.c0, &z4 = a.c0.c1, &z3 = a.c0.c2, &z2 = a.c1.c0, &z1 = a.c1.c1, &z5 = a.c1.c2;
    Fp2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(t0, t1, z0, z1); fp4_sqr(t2, t3, z2, z3); fp4_sqr(t4, t5, z4, z5);
    const Fp2 xt5 = mul_xi(t5);
    Fp12 o;
    o.c0.c0 = add(dbl(sub(t0, z0)), t0);
    o.c1.c1 = add(dbl(add(t1, z1)), t1);
    o.c1.c0 = add(dbl(add(xt5, z2)), xt5);
    o.c0.c2 = add(dbl(sub(t4, z3)), t4);
    o.c0.c1 = add(dbl(sub(t2, z4)), t2);
    o.c1.c2 = add(dbl(add(t3, z5)), t3);
    return o;
}

}  // namespace ripp
