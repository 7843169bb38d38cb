// BLS12-381 optimal-ate pairing pieces (host + device).
//
// The pairing product  prod_i e(A_i, B_i)  (inner_products/src/lib.rs:77-116, sipp/src/lib.rs:184-217) is
// evaluated MI355X-style in three stages instead of arkworks' per-thread multi_miller_loop:
//   (1) LINES     one lane per pair walks T <- 2T / T+Q over the 68 steps of |x| and emits the sparse line
//                 element (c0, c1*xP, c2*yP) of every step to HBM                         [kernels/miller.hip]
//   (2) PRODUCTS  for every step s independently, L_s = prod_i line_{s,i}  (sparse mul_by_014 accumulation,
//                 then a dense Fp12 tree)                                                  [kernels/miller.hip]
//   (3) COMBINE   f = fold_s (f^2 * L_s)  -- 63 squarings TOTAL instead of 63 per pair (or per 4 pairs as in
//                 ark-ec's multi_miller_loop), conjugate (x < 0), ONE final exponentiation  [this file, host]
// Since prod_i (f_i) with f_i = prod_s l_{s,i}^{2^(..)} regroups exactly into stage (3)'s recurrence, the value
// after the final exponentiation is bit-identical to the reference's.
#pragma once
#include "curve.hpp"

namespace ripp {

#if defined(RIPP_BLS12_377)
constexpr uint64_t BLS_X_ABS = 0x8508c00000000001ull;   // x > 0 (no conjugations)
constexpr bool BLS_X_NEG = false;
constexpr int N_LINES = 69;                              // 63 doubling + 6 addition steps
#else
constexpr uint64_t BLS_X_ABS = 0xd201000000010000ull;   // |x|, x < 0
constexpr bool BLS_X_NEG = true;
constexpr int N_LINES = 68;                              // 63 doubling + 5 addition steps
#endif

// a / 2 mod p without a multiplication: (a + (a odd ? p : 0)) >> 1
RIPP_HD Fp half(const Fp& a) {
    const uint32_t mask = 0u - (a.l[0] & 1u);
    uint32_t t[12], c = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) t[i] = addc32(a.l[i], FpParams::mod(i) & mask, c);   // < 2^382: no carry out
    Fp r;
#pragma unroll
    for (int i = 0; i < 11; ++i) r.l[i] = (t[i] >> 1) | (t[i + 1] << 31);
    r.l[11] = t[11] >> 1;
    return r;
}
RIPP_HD Fp2 half(const Fp2& a) { return {half(a.c0), half(a.c1)}; }

struct LineCoeffs { Fp2 c0, c1, c2; };   // line element c0 + c1 v + c2 v w  (indices 0,1,4 of mul_by_014)

#if defined(RIPP_BLS12_377)
// b' = 1/u = (0, -1/5):  (a0 + a1 u) * b' = a1 + (a0 * (-1/5)) u   -- one Fp product by a constant
RIPP_MID Fp2 mul_by_b_twist(const Fp2& a) { return {a.c1, fmul(a.c0, fp_const(RIPP_FP_TWIST_B1))}; }
#else
// 4(1+u) * a  -- multiplication by the twist coefficient b' with additions only
RIPP_HD Fp2 mul_by_b_twist(const Fp2& a) { return mul_xi(dbl(dbl(a))); }
#endif

// Doubling step in homogeneous projective coordinates (the formulas ark-ec 0.4 bls12::g2 `double_in_place` uses,
// with the two multiplications by 1/2 replaced by shifts): T <- 2T, returns the tangent line at T scaled for P.
RIPP_MID LineCoeffs line_double(Fp2& X, Fp2& Y, Fp2& Z, const Fp& xP, const Fp& yP) {
    const Fp2 a = half(mul(X, Y));
    const Fp2 b = sqr(Y), c = sqr(Z);
    const Fp2 e = mul_by_b_twist(add(dbl(c), c));
    const Fp2 f = add(dbl(e), e);
    const Fp2 g = half(add(b, f));
    const Fp2 h = sub(sqr(add(Y, Z)), add(b, c));
    const Fp2 i = sub(e, b);
    const Fp2 j = sqr(X);
    const Fp2 e2 = sqr(e);
    X = mul(a, sub(b, f));
    Y = sub(sqr(g), add(dbl(e2), e2));
    Z = mul(b, h);
#if defined(RIPP_BLS12_377)
    return {mul_fp(neg(h), yP), mul_fp(add(dbl(j), j), xP), i};            // TwistType::D: (-h yP, 3j xP, i) at w^0, w^1, w^3
#else
    return {i, mul_fp(add(dbl(j), j), xP), mul_fp(neg(h), yP)};
#endif
}
// Addition step T <- T + Q (Q affine), returns the chord line scaled for P.
RIPP_MID LineCoeffs line_add(Fp2& X, Fp2& Y, Fp2& Z, const Fp2& qx, const Fp2& qy, const Fp& xP, const Fp& yP) {
    const Fp2 theta = sub(Y, mul(qy, Z));
    const Fp2 lambda = sub(X, mul(qx, Z));
    const Fp2 c = sqr(theta), d = sqr(lambda);
    const Fp2 e = mul(lambda, d), f = mul(Z, c), g = mul(X, d);
    const Fp2 h = sub(add(e, f), dbl(g));
    X = mul(lambda, h);
    Y = sub(mul(theta, sub(g, h)), mul(e, Y));
    Z = mul(Z, e);
    const Fp2 j = sub(mul(theta, qx), mul(lambda, qy));
#if defined(RIPP_BLS12_377)
    return {mul_fp(lambda, yP), mul_fp(neg(theta), xP), j};
#else
    return {j, mul_fp(neg(theta), xP), mul_fp(lambda, yP)};
#endif
}

// Stage (3): fold the per-step products L[0..67] into the Miller-loop value (conjugated because x < 0).
RIPP_FN Fp12 miller_combine(const Fp12* L) {
    Fp12 f = Fp12::one();
    int s = 0;
    for (int b = 62; b >= 0; --b) {
        f = mul(sqr(f), L[s++]);
        if ((BLS_X_ABS >> b) & 1) f = mul(f, L[s++]);
    }
    return BLS_X_NEG ? conj(f) : f;
}

// The same recurrence over the bit range [hi, lo] only, started from 1 and NOT conjugated: with the bits cut into ranges R_1 > R_2 > ... the
// whole value is  conj?( prod_j g_j^(2^(bits below R_j)) ),  so the ranges -- and their final exponentiations, a homomorphism -- can run on
// different host threads and be joined with cheap cyclotomic squarings (engine.hip: pairing_values).
RIPP_FN Fp12 miller_combine_range(const Fp12* L, int hi, int lo) {
    int s = 0;
    for (int b = 62; b > hi; --b) s += 1 + (int)((BLS_X_ABS >> b) & 1);      // rows consumed by the bits above the range
    Fp12 f = Fp12::one(); bool first = true;
    for (int b = hi; b >= lo; --b) {
        if (first) { f = L[s++]; first = false; } else f = mul(sqr(f), L[s++]);
        if ((BLS_X_ABS >> b) & 1) f = mul(f, L[s++]);
    }
    return f;
}

// f^|x| then conjugate (x < 0), on cyclotomic-subgroup elements
RIPP_FN Fp12 exp_by_x(const Fp12& a) {
    Fp12 acc = a;
    for (int i = 62; i >= 0; --i) { acc = cyclotomic_sqr(acc); if ((BLS_X_ABS >> i) & 1) acc = mul(acc, a); }
    return BLS_X_NEG ? conj(acc) : acc;
}

// Final exponentiation with the exponent (p^6-1)(p^2+1) * ((x-1)^2 (x+p)(x^2+p^2-1) + 3): the value arkworks'
// `Bls12::final_exponentiation` returns (inner_products/src/lib.rs:115, sipp/src/lib.rs:216) -- the cube of the
// textbook optimal-ate pairing.  Any addition chain gives the same field element; this one is eprint 2020/875's.
RIPP_FN Fp12 final_exponentiation(const Fp12& f) {
    Fp12 r = mul(conj(f), inv(f));            // f^(p^6-1)
    r = mul(frobenius(r, 2), r);              // ^(p^2+1)
    Fp12 y0 = cyclotomic_sqr(r);
    Fp12 y1 = mul(exp_by_x(r), conj(r));
    Fp12 y2 = exp_by_x(y1);
    y1 = mul(conj(y1), y2);
    y2 = exp_by_x(y1);
    y1 = mul(frobenius(y1, 1), y2);
    r = mul(r, y0);
    y0 = exp_by_x(y1);
    y2 = exp_by_x(y0);
    y0 = frobenius(y1, 2);
    y1 = mul(mul(conj(y1), y2), y0);
    return mul(r, y1);
}

}  // namespace ripp
