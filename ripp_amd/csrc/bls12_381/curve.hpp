// Short-Weierstrass a = 0 group law for G1 = E(Fp) and G2 = E'(Fp2) of BLS12-381 (host + device).
// Jacobian coordinates (x = X/Z^2, y = Y/Z^3) as ark-ec 0.4 `short_weierstrass::Projective`; the flat layouts
// (ripp_g1a/g1j/g2a/g2j in include/ripp_hip.h) encode affine infinity as (0,0) and Jacobian infinity as Z = 0.
#pragma once
#include "tower.hpp"

namespace ripp {

template <class F> struct Affine { F x, y; };
template <class F> struct Jac { F x, y, z; };
using G1A = Affine<Fp>;  using G1J = Jac<Fp>;
using G2A = Affine<Fp2>; using G2J = Jac<Fp2>;

template <class F> RIPP_HD bool is_inf(const Affine<F>& p) { return p.x.is_zero() && p.y.is_zero(); }
template <class F> RIPP_HD bool is_inf(const Jac<F>& p) { return p.z.is_zero(); }
template <class F> RIPP_HD Jac<F> jac_inf() { return {F::one(), F::one(), F::zero()}; }
template <class F> RIPP_HD Affine<F> aff_inf() { return {F::zero(), F::zero()}; }
template <class F> RIPP_HD Jac<F> to_jac(const Affine<F>& p) { return is_inf(p) ? jac_inf<F>() : Jac<F>{p.x, p.y, F::one()}; }
template <class F> RIPP_HD Jac<F> neg(const Jac<F>& p) { return {p.x, neg(p.y), p.z}; }
template <class F> RIPP_HD Affine<F> neg(const Affine<F>& p) { return {p.x, neg(p.y)}; }

// dbl-2009-l (a = 0): 2M + 5S.  Z = 0 maps to Z3 = 0 without a branch.
template <class F> RIPP_MID Jac<F> dbl(const Jac<F>& p) {
    const F A = fsqr(p.x), B = fsqr(p.y), C = fsqr(B);
    const F D = dbl(sub(sub(fsqr(add(p.x, B)), A), C));
    const F E = add(dbl(A), A), Fq = fsqr(E);
    Jac<F> r;
    r.x = sub(sub(Fq, D), D);
    r.y = sub(fmul(E, sub(D, r.x)), dbl(dbl(dbl(C))));
    r.z = dbl(fmul(p.y, p.z));
    return r;
}

// madd-2007-bl mixed addition p + q (q affine), all special cases handled.
template <class F> RIPP_MID Jac<F> add_mixed(const Jac<F>& p, const Affine<F>& q) {
    if (is_inf(q)) return p;
    if (is_inf(p)) return Jac<F>{q.x, q.y, F::one()};
    const F Z1Z1 = fsqr(p.z), U2 = fmul(q.x, Z1Z1), S2 = fmul(fmul(q.y, p.z), Z1Z1);
    const F H = sub(U2, p.x);
    F rr = sub(S2, p.y);
    if (H.is_zero()) return rr.is_zero() ? dbl(p) : jac_inf<F>();
    rr = dbl(rr);
    const F HH = fsqr(H), I = dbl(dbl(HH)), J = fmul(H, I), V = fmul(p.x, I);
    Jac<F> r;
    r.x = sub(sub(sub(fsqr(rr), J), V), V);
    r.y = sub(fmul(rr, sub(V, r.x)), dbl(fmul(p.y, J)));
    r.z = sub(sub(fsqr(add(p.z, H)), Z1Z1), HH);
    return r;
}

// add-2007-bl general addition.
template <class F> RIPP_MID Jac<F> add(const Jac<F>& p, const Jac<F>& q) {
    if (is_inf(q)) return p;
    if (is_inf(p)) return q;
    const F Z1Z1 = fsqr(p.z), Z2Z2 = fsqr(q.z);
    const F U1 = fmul(p.x, Z2Z2), U2 = fmul(q.x, Z1Z1);
    const F S1 = fmul(fmul(p.y, q.z), Z2Z2), S2 = fmul(fmul(q.y, p.z), Z1Z1);
    const F H = sub(U2, U1);
    F rr = sub(S2, S1);
    if (H.is_zero()) return rr.is_zero() ? dbl(p) : jac_inf<F>();
    rr = dbl(rr);
    const F I = fsqr(dbl(H)), J = fmul(H, I), V = fmul(U1, I);
    Jac<F> r;
    r.x = sub(sub(sub(fsqr(rr), J), V), V);
    r.y = sub(fmul(rr, sub(V, r.x)), dbl(fmul(S1, J)));
    r.z = fmul(sub(sub(fsqr(add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}

template <class F> RIPP_FN Affine<F> to_affine(const Jac<F>& p) {
    if (is_inf(p)) return aff_inf<F>();
    const F zi = finv(p.z), zi2 = fsqr(zi);
    return {fmul(p.x, zi2), fmul(p.y, fmul(zi2, zi))};
}
template <class F> RIPP_HD bool eq(const Jac<F>& a, const Jac<F>& b) {
    if (is_inf(a) || is_inf(b)) return is_inf(a) && is_inf(b);
    const F za2 = fsqr(a.z), zb2 = fsqr(b.z);
    if (!(fmul(a.x, zb2) == fmul(b.x, za2))) return false;
    return fmul(a.y, fmul(zb2, b.z)) == fmul(b.y, fmul(za2, a.z));
}

// Plain MSB-first double-and-add over a canonical (non-Montgomery) little-endian scalar of `nbits` bits.
template <class F> RIPP_FN Jac<F> scalar_mul_bits(const Affine<F>& p, const uint32_t* k, int nbits) {
    Jac<F> acc = jac_inf<F>();
    for (int i = nbits - 1; i >= 0; --i) {
        acc = dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = add_mixed(acc, p);
    }
    return acc;
}

RIPP_HD G1A g1_generator() { return {fp_const(RIPP_G1_GEN_X), fp_const(RIPP_G1_GEN_Y)}; }
RIPP_HD G2A g2_generator() {
    return {{fp_const(RIPP_G2_GEN_X0), fp_const(RIPP_G2_GEN_X1)}, {fp_const(RIPP_G2_GEN_Y0), fp_const(RIPP_G2_GEN_Y1)}};
}

}  // namespace ripp
