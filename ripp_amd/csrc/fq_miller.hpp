// Stage 1 of the pairing product on the carry-free field form (fq28.hpp): the throughput twin of kernels.hpp's k_miller_lines
// (G2Prepared::from + the line side of multi_miller_loop, inner_products/src/lib.rs:86-88,112).
//
// k_miller_lines runs AT the VALU issue roof (2^19 pairs x ~1.3 M instructions / 34.4 T lane-instructions/s = 19.8 ms; measured 19.7): only
// fewer instructions make it faster.  Here
//   * an Fp2 product is two lazily reduced sums of two products  c0 = a0 b0 + (K - a1) b1,  c1 = a0 b1 + a1 b0  (fq_dot<2>: 4 x 196 + 2 x 196
//     multiply-adds -- the Karatsuba count -- with NO Karatsuba sums or temporaries: the live state of a product is its operands), a square
//     is (a0 + a1)(a0 - a1), (2 a0) a1; additions / subtractions are limb-wise on lazily bounded values (bounds checked by the compiler);
//   * the doubling step is rescaled so that it has no halvings and no negations: with a' = XY, g' = b + 3e it produces
//         (X3, Y3, Z3) = (2 a' (f - b),  12 e^2 - g'^2,  (2b)(2 nh)),   nh = b + c - (Y + Z)^2 = -h
//     = -4 x the textbook point -- the same projective point -- and the line (nh yP, 3 X^2 xP, e - b) of the step is the textbook one;
//   * P's coordinates are converted once per pair and rest in LDS; Q is re-loaded and converted where an addition step uses it.
// All values are true Montgomery-392 values (P and Q are converted with the exact x 2^8 re-slicing + one quotient estimate).  A line coefficient is
// stored as the 12 words of its canonical Montgomery-392 integer: read as a Montgomery-384 value (k_line_products) that is the coefficient times
// 2^8 -- every line is scaled by the same element of Fp, which the final exponentiation removes; k_line_products_q reads the words as they are.
// Both twists: BLS12-381 (M-type, b' = 4 (1 + u), line slots (free, xP, yP)) and BLS12-377 (D-type, b' = 1 / u = (0, -1/5), line slots
// (yP, xP, free): kernels.hpp LINE_SLOT_*); the tower's non-residue is in the Fp2 products (fq_curve2.hpp FQ2_BETA).
#pragma once
#include "fq_curve2.hpp"

namespace ripp {

#if defined(__HIP_DEVICE_COMPILE__)
// a line coefficient (value < 2p) -> the canonical integer, 12 words, chunked SoA (kernels.hpp store_chunks layout)
__device__ __forceinline__ void store_line_q(uint4* lines, size_t row, size_t stride, size_t i, const Fq2n& v) {
    uint32_t w[24];
    { const Fqn c = fq_canon(v.c0); uint32_t t[12]; fq_pack(c, t);
#pragma unroll
      for (int k = 0; k < 12; ++k) w[k] = t[k]; }
    { const Fqn c = fq_canon(v.c1); uint32_t t[12]; fq_pack(c, t);
#pragma unroll
      for (int k = 0; k < 12; ++k) w[12 + k] = t[k]; }
#pragma unroll
    for (int q = 0; q < 6; ++q) lines[(row * 6 + q) * stride + i] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}
__device__ __forceinline__ void store_line_raw(uint4* lines, size_t row, size_t stride, size_t i, const Fp2& v) {
    const uint4* src = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (int q = 0; q < 6; ++q) lines[(row * 6 + q) * stride + i] = src[q];
}
// P's coordinates in LDS: 4 chunks each (14 limbs + 2), lane-strided
template <class FQ = Fqn> __device__ __forceinline__ FQ ld_park_fq(const uint4* park, int slot) {
    uint4 q[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = park[(slot * 4 + c) * 256];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
    FQ v;
#pragma unroll
    for (int k = 0; k < fq28::NL; ++k) v.l[k] = w[k];
    return v;
}
template <class FQ> __device__ __forceinline__ void st_park_fq(uint4* park, int slot, const FQ& v) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < fq28::NL; ++k) w[k] = v.l[k];
    w[14] = 0; w[15] = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) park[(slot * 4 + c) * 256] = uint4{w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]};
}

// ---- one G2 chain, several P's (ChainSets, kernels.hpp).  The doubling / addition steps on Q do not depend on P: of a line only the two
// coefficients scaled by xP / yP do.  When several products of one launch pair the SAME Q vector with different P vectors -- round 0 of a SIPP proof
// and the look-ahead of the following rounds: every B block meets up to 2^R + 1 A blocks (engine.hip job_round0_shared / job_lookahead) -- ONE lane
// walks Q's chain and emits the lines of all of them: per extra P and step two Fp2 x Fp products (1 568 multiply-adds) + the conversion of its two
// coordinates (re-loaded from global memory in the engine's form: no LDS for them) against ~9 800 / ~16 000 for the step itself.
struct ExtraP { const G1A* const* a; int np; bool sk1, sk2, sk3; };       // a[t], t = 1 .. np - 1: the other P vectors of the group (already offset).  Wave-uniform (SGPRs).
// Skip flags ("the pair (P_t[i], Q[i]) has a point at infinity") are BOOLS, i.e. lane masks in scalar registers: the kernel sits at 256 vector registers
// the lines of the extra P's for one step: (yP-scaled from cy, xP-scaled from cx, free coefficient cf), rows of product t at s + t * N_LINES
// the free coefficient (the same value for every P of the group) of one step, rows of product t at s + t * N_LINES: stored where it is formed
__device__ __forceinline__ void extra_free_q(const ExtraP& xp, const Fq2n& cf, uint4* lines, size_t s, size_t stride, size_t i) {
#pragma unroll 1
    for (int t = 1; t < xp.np; ++t) {
        const size_t st = s + (size_t)t * N_LINES;
        const bool sk = t == 1 ? xp.sk1 : t == 2 ? xp.sk2 : xp.sk3;      // (its OWN P or the shared Q at infinity; the group's first P being the identity does not concern it)
        if (sk) store_line_raw(lines, st * 3 + LINE_SLOT_FREE, stride, i, LINE_UNIT_FREE); else store_line_q(lines, st * 3 + LINE_SLOT_FREE, stride, i, cf);
    }
}
// the two P-scaled coefficients: (yP-scaled from cy, xP-scaled from cx)
template <class CY, class CX>
__device__ __forceinline__ void extra_scaled_q(const ExtraP& xp, const CY& cy, const CX& cx, uint4* lines, size_t s, size_t stride, size_t i) {
#pragma unroll 1
    for (int t = 1; t < xp.np; ++t) {
        const size_t st = s + (size_t)t * N_LINES;
        const bool sk = t == 1 ? xp.sk1 : t == 2 ? xp.sk2 : xp.sk3;
        auto addr = [&]() { uint32_t il = (uint32_t)i; asm volatile("" : "+v"(il)); return xp.a[t] + il; };      // P_t's address is re-formed per use, not held across a product
        { Fq2n l2 = f2_mul_fq(cy, fq_from_fp_fast(addr()->y)); f2_pin(l2);
          if (sk) store_line_raw(lines, st * 3 + LINE_SLOT_YP, stride, i, LINE_UNIT_YP); else store_line_q(lines, st * 3 + LINE_SLOT_YP, stride, i, l2); }
        { Fq2n l1 = f2_mul_fq(cx, fq_from_fp_fast(addr()->x)); f2_pin(l1);
          if (sk) store_line_raw(lines, st * 3 + 1, stride, i, Fp2::zero()); else store_line_q(lines, st * 3 + 1, stride, i, l1); }
    }
}

// doubling step + its line; (X, Y, Z) <- -4 x the doubled point.  park: slots 0 (xP), 1 (yP).  Every product is PINNED where it is written
// (fq28.hpp fq_pin): the compiler otherwise carries un-reduced column sums of one step across the loop's branch into the next.
__device__ __forceinline__ void line_double_store_q(Fq2C& X, Fq2C& Y, Fq2C& Z, const uint4* park, uint4* lines, size_t s, size_t stride, size_t i, bool skip, const ExtraP& xp) {
    Fq2n t1 = f2_sqrd(f2_norm(f2_add(Y, Z))); f2_pin(t1);
    Fq2n c = f2_sqrd(Z); f2_pin(c);
    Fq2n b = f2_sqrd(Y); f2_pin(b);
    auto nh = f2_norm(f2_sub(f2_add(b, c), t1)); f2_pin(nh);                      // -h, value < 7p
    {
        Fq2n l2 = f2_mul_fq(nh, ld_park_fq(park, 1)); f2_pin(l2);
        if (skip) store_line_raw(lines, s * 3 + LINE_SLOT_YP, stride, i, LINE_UNIT_YP); else store_line_q(lines, s * 3 + LINE_SLOT_YP, stride, i, l2);
    }
    Fq2n e;                                                                       // b' * 3c
    {
        const auto c3 = f2_add(f2_add(c, c), c);
#if defined(RIPP_BLS12_377)
        // b' = 1 / u = (0, B1), B1 = -1/5:  (c0 + c1 u) B1 u = -5 B1 c1 + B1 c0 u = (c1, B1 c0): one product by a constant
        e.c0 = fq_reduce(c3.c1);
        { constexpr uint32_t b1w[12] = RIPP_FP_TWIST_B1; constexpr fq28::Limbs B1 = fq28::from_mont384(b1w);
          e.c1 = fq_mul(fq_norm(c3.c0), fq_const<FQ_LN, 1>(B1)); }
#else
        // b' = 4 (1 + u): 4 (c0 - c1) + 4 (c0 + c1) u
        e.c0 = fq_reduce(fq_dbl(fq_dbl(fq_norm(fq_sub(c3.c0, c3.c1)))));
        e.c1 = fq_reduce(fq_dbl(fq_dbl(fq_norm(fq_add(c3.c0, c3.c1)))));
#endif
    } f2_pin(e);
    { const Fq2n fr = f2_reduce(f2_sub(e, b));
      if (skip) store_line_raw(lines, s * 3 + LINE_SLOT_FREE, stride, i, LINE_UNIT_FREE); else store_line_q(lines, s * 3 + LINE_SLOT_FREE, stride, i, fr);
      if (xp.np > 1) extra_free_q(xp, fr, lines, s, stride, i); }                  // (wave-uniform branch)
    Fq2n a2 = f2_muld(X, Y); f2_pin(a2);                                          // a' = X Y
    {
        Fq2n j = f2_sqrd(X); f2_pin(j);
        { Fq2n l1 = f2_mul_fq(f2_add(f2_add(j, j), j), ld_park_fq(park, 0)); f2_pin(l1);
          if (skip) store_line_raw(lines, s * 3 + 1, stride, i, Fp2::zero()); else store_line_q(lines, s * 3 + 1, stride, i, l1); }
        if (xp.np > 1) extra_scaled_q(xp, nh, f2_add(f2_add(j, j), j), lines, s, stride, i);      // (3 j is re-formed, not held)
    }
    Z = f2_to_coord(f2_muld(f2_dbl(b), f2_dbl(nh))); f2_pin(Z);                   // (2b)(2 nh) = -4 b h
    const auto f = f2_add(f2_add(e, e), e);                                       // 3e, lazy
    X = f2_to_coord(f2_muld(f2_dbl(a2), f2_norm(f2_sub(f, b)))); f2_pin(X);       // 2 a' (f - b) = -4 a (b - f)
#if defined(RIPP_BLS12_377)
    Fq2n g2 = f2_sqrd(f2_norm(f2_add(b, f))); f2_pin(g2);                         // g'^2 = 4 g^2
    Fq2n e2 = f2_sqrd(e); f2_pin(e2);
    const auto e12 = f2_dbl(f2_dbl(f2_add(f2_add(e2, e2), e2)));                  // 12 e^2, lazy
    Y = f2_to_coord(f2_sub(e12, g2)); f2_pin(Y);                                  // 12 e^2 - g'^2 = -4 (g^2 - 3 e^2)
#else
    // Y3 = 12 e^2 - g'^2 (g' = b + 3e) as a difference of two SQUARES with one reduction per part (fq_curve.hpp fq_mul_sub):
    //   re = 3 (e0 + e1) . 4 (e0 - e1) - (g0 + g1)(g0 - g1),   im = 8 e0 . 3 e1 - g0 . 2 g1
    // four limb products + two reductions where the two squares took four + four and their difference two more quotient estimates; the result is a reduced value
    {
        const auto g = f2_norm(f2_add(b, f));
        Fq2n y3;
        { const auto se = fq_add(e.c0, e.c1); const auto a3 = fq_add(fq_dbl(se), se);
          y3.c0 = fq_mul_sub(a3, fq_norm(fq_dbl(fq_dbl(fq_sub(e.c0, e.c1)))), fq_add(g.c0, g.c1), fq_norm(fq_sub(g.c0, g.c1))); } fq_pin(y3.c0);
        { const auto e3 = fq_add(fq_dbl(e.c1), e.c1);
          y3.c1 = fq_mul_sub(fq_norm(fq_dbl(fq_dbl(fq_dbl(e.c0)))), e3, g.c0, fq_dbl(g.c1)); }
        Y = f2_to_coord(y3); f2_pin(Y);
    }
#endif
}
__device__ __forceinline__ Fq2n f2_load_conv(const Fp2* p) { const Fp2 v = *p; return f2_from(v); }
// mixed addition step + its line (-j, theta xP, -lambda yP) = -1 x the textbook line
__device__ __forceinline__ void line_add_store_q(Fq2C& X, Fq2C& Y, Fq2C& Z, const G2A* q, const uint4* park, uint4* lines, size_t s, size_t stride, size_t i, bool skip, const ExtraP& xp) {
    Fq2n theta, lambda;
    { const Fq2n qy = f2_load_conv(&opaque(q)->y); theta = f2_reduce(f2_sub(Y, f2_muld(qy, Z))); } f2_pin(theta);
    st_park_fq(const_cast<uint4*>(park), 2, Y.c0); st_park_fq(const_cast<uint4*>(park), 3, Y.c1);      // Y rests in LDS until the step's last product
    { const Fq2n qx = f2_load_conv(&opaque(q)->x); lambda = f2_reduce(f2_sub(X, f2_muld(qx, Z))); } f2_pin(lambda);
    {
        Fq2n t; { const Fq2n qy = f2_load_conv(&opaque(q)->y); t = f2_muld(lambda, qy); } f2_pin(t);
        const Fq2n qx = f2_load_conv(&opaque(q)->x);
        Fq2n nj = f2_reduce(f2_sub(t, f2_muld(theta, qx))); f2_pin(nj);             // lambda qy - theta qx = -j
        if (skip) store_line_raw(lines, s * 3 + LINE_SLOT_FREE, stride, i, LINE_UNIT_FREE); else store_line_q(lines, s * 3 + LINE_SLOT_FREE, stride, i, nj);
        if (xp.np > 1) extra_free_q(xp, nj, lines, s, stride, i);                  // the other P's of the group: (-lambda yP_t, theta xP_t, -j)
    }
    { Fq2n l1 = f2_mul_fq(theta, ld_park_fq(park, 0)); f2_pin(l1);
      if (skip) store_line_raw(lines, s * 3 + 1, stride, i, Fp2::zero()); else store_line_q(lines, s * 3 + 1, stride, i, l1); }
    { const auto nl = Fq2T<fq28::sub_lm(1, FQ_LN), 4>{fq_neg(lambda.c0), fq_neg(lambda.c1)};
      Fq2n l2 = f2_mul_fq(nl, ld_park_fq(park, 1)); f2_pin(l2);
      if (skip) store_line_raw(lines, s * 3 + LINE_SLOT_YP, stride, i, LINE_UNIT_YP); else store_line_q(lines, s * 3 + LINE_SLOT_YP, stride, i, l2);
      if (xp.np > 1) extra_scaled_q(xp, nl, theta, lines, s, stride, i); }
    Fq2n f;
    { Fq2n c = f2_sqrd(theta); f2_pin(c); f = f2_muld(c, Z); } f2_pin(f);
    Fq2n d = f2_sqrd(lambda); f2_pin(d);
    Fq2n g = f2_muld(d, X); f2_pin(g);                                            // (before e: X dies here, d right after -- one coordinate less alive at the peak)
    Fq2n e = f2_muld(lambda, d); f2_pin(e);
    Fq2n h = f2_reduce(f2_sub(f2_add(e, f), f2_dbl(g))); f2_pin(h);
    X = f2_to_coord(f2_muld(lambda, h)); f2_pin(X);
    Z = f2_to_coord(f2_muld(e, Z)); f2_pin(Z);
    Fq2n t = f2_muld(theta, f2_norm(f2_sub(g, h))); f2_pin(t);
    { Fq2C y1; y1.c0 = ld_park_fq<decltype(y1.c0)>(park, 2); y1.c1 = ld_park_fq<decltype(y1.c1)>(park, 3);
      Y = f2_to_coord(f2_sub(t, f2_muld(e, y1))); } f2_pin(Y);
}
#endif

// same line-buffer layout as k_miller_lines; grid.y = CHAIN (a group of cs.np[g] consecutive products first[g] .. that pair ONE Q vector with different P vectors)
__global__ void __launch_bounds__(256, RIPP_OCC) k_miller_lines_q(ChainSets cs, uint32_t M, uint4* __restrict__ lines, size_t stride) {
    __shared__ uint4 park_[16 * 256];                            // per lane: xP, yP (the group's first P) and (addition steps) the two halves of Y
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const int p0 = cs.first[blockIdx.y];
    const G1A* __restrict__ a = cs.a[p0];
    const G2A* __restrict__ b = cs.b[blockIdx.y];
    uint4* park = park_ + threadIdx.x;
    Fq2C X, Y, Z;
    bool skip;
    ExtraP xp{cs.a + p0, cs.np[blockIdx.y], false, false, false};
    {
        const G2A Q = b[i]; const G1A P = a[i];
        const bool qinf = Q.x.is_zero() && Q.y.is_zero();
        skip = is_inf(P) || qinf;
        if (xp.np > 1) xp.sk1 = is_inf(xp.a[1][i]) || qinf;
        if (xp.np > 2) xp.sk2 = is_inf(xp.a[2][i]) || qinf;
        if (xp.np > 3) xp.sk3 = is_inf(xp.a[3][i]) || qinf;
        X = f2_to_coord(f2_from(Q.x)); Y = f2_to_coord(f2_from(Q.y)); Z = f2_to_coord(Fq2n{fq_one(), fq_zero()});
        st_park_fq(park, 0, fq_from_fp_fast(P.x)); st_park_fq(park, 1, fq_from_fp_fast(P.y));
        f2_pin(X); f2_pin(Y); f2_pin(Z);                             // (Z = 1 is not to be folded into a peeled first iteration: that copy of the loop body spilled)
    }
    size_t s = (size_t)p0 * N_LINES;
#pragma unroll 1
    for (int bit = 62; bit >= 0; --bit) {
        line_double_store_q(X, Y, Z, park, lines, s, stride, i, skip, xp);
        ++s;
        if ((BLS_X_ABS >> bit) & 1ull) {
            uint32_t iq = i; asm volatile("" : "+v"(iq));                  // Q's address is re-formed here (kept alive across the loop it was the kernel's last spilled register pair)
            line_add_store_q(X, Y, Z, b + iq, park, lines, s, stride, i, skip, xp);
            ++s;
        }
    }
#endif
}

}  // namespace ripp
