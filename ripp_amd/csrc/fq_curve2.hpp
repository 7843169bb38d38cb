// G2 group law and the table fold on the carry-free field form: the throughput twin of kernels.hpp's k_fold_g2_tab (dbl-2009-l / madd-2007-bl in
// the same low-liveness order, Y1 parked in LDS while it is idle).  An Fp2 product is two lazily reduced sums of two products
// (a0 b0 + (K - a1) b1 and a0 b1 + a1 b0: 4 x 196 limb products + 2 reductions, no Karatsuba additions), a square is (a0 + a1)(a0 - a1) and
// (2 a0) a1 on lazy operands.  Every bound is checked by the compiler (fq28.hpp).
// Both towers: Fp2 = Fp[u] / (u^2 + FQ2_BETA) with FQ2_BETA = 1 (BLS12-381) or 5 (BLS12-377) -- the factor is a limb-wise lazy multiple of the
// operand that is negated anyway (K - 5 a1), so a BLS12-377 product costs the same multiply-adds; its square is a0 a0 + (K - 5 a1) a1 and (2 a0) a1
// (588 + 392 multiply-adds against 392 + 392: the (a0 + a1)(a0 - a1) trick needs u^2 = -1).
#pragma once
#include "fq_curve.hpp"

namespace ripp {

#if defined(__HIP_DEVICE_COMPILE__)
#if defined(RIPP_BLS12_377)
constexpr int FQ2_BETA = 5;               // u^2 = -5
#else
constexpr int FQ2_BETA = 1;               // u^2 = -1
#endif
// FQ2_BETA * a, limb-wise (lazy)
template <uint64_t L1, int V1> __device__ __forceinline__ auto fq_mul_beta(const Fq<L1, V1>& a) {
    if constexpr (FQ2_BETA == 1) return a;
    else { static_assert(FQ2_BETA == 5, "FQ2_BETA"); return fq_add(fq_dbl(fq_dbl(a)), a); }
}
template <uint64_t LM = FQ_LN, int VB = 2> struct Fq2T { Fq<LM, VB> c0, c1; };
using Fq2n = Fq2T<FQ_LN, 2>;          // a product
using Fq2C = Fq2T<FQ_LN, 4>;          // a coordinate between two group operations

template <uint64_t L1, int V1, uint64_t L2, int V2> __device__ __forceinline__ auto f2_add(const Fq2T<L1, V1>& a, const Fq2T<L2, V2>& b) { return Fq2T<L1 + L2 - 1, V1 + V2>{fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
template <uint64_t L1, int V1, uint64_t L2, int V2> __device__ __forceinline__ auto f2_sub(const Fq2T<L1, V1>& a, const Fq2T<L2, V2>& b) { return Fq2T<fq28::sub_lm(L1, L2), V1 + V2 + 1>{fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ auto f2_dbl(const Fq2T<L1, V1>& a) { return Fq2T<2 * L1 - 1, 2 * V1>{fq_dbl(a.c0), fq_dbl(a.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ auto f2_norm(const Fq2T<L1, V1>& a) { return Fq2T<FQ_LN, V1>{fq_norm(a.c0), fq_norm(a.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ Fq2n f2_reduce(const Fq2T<L1, V1>& a) { return {fq_reduce(a.c0), fq_reduce(a.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ Fq2C f2_coord(const Fq2T<L1, V1>& a) { return {fq_coord(a.c0), fq_coord(a.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ void f2_pin(Fq2T<L1, V1>& a) { fq_pin(a.c0); fq_pin(a.c1); }
__device__ __forceinline__ bool f2_is_zero(const Fq2n& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
// ---- the inlined forms (no call, no Karatsuba temporaries): what the throughput kernels use.  The first half of a product is PINNED (fq28.hpp
// fq_pin) so that the two column sets are never alive together.
// (a0 + a1 u)(b0 + b1 u), u^2 = -FQ2_BETA: two lazily reduced sums of two products, inlined; the result is reduced (< 2p, normalised limbs).
// BLS12-377: the operand that carries the factor 5 is normalised first when its limbs are lazy (the column sums would leave 64 bits otherwise).
template <uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ Fq2n f2_muld(const Fq2T<L1, V1>& a_, const Fq2T<L2, V2>& b_) {
    constexpr bool fits = fq28::dot_fits(2, fq28::sub_lm(1, (uint64_t)FQ2_BETA * (L1 - 1) + 1), L2);
    const auto a = [&]() { if constexpr (fits) return a_; else return Fq2T<FQ_LN, V1>{fq_norm(a_.c0), fq_norm(a_.c1)}; }();
    const auto b = [&]() { if constexpr (fits || L2 <= FQ_LN) return b_; else return Fq2T<FQ_LN, V2>{fq_norm(b_.c0), fq_norm(b_.c1)}; }();
    using TA0 = decltype(a.c0); using TB0 = decltype(b.c0);
    const auto na1 = fq_neg(fq_mul_beta(a.c1));                        // K - beta a1, K a multiple of p with dominating limbs
    using TA = decltype(na1);
    Fq2n r;
    { const TA aa[2] = {fq_widen<TA::LMAX, TA::VMAXB>(a.c0), na1}; const TB0 bb[2] = {b.c0, b.c1}; r.c0 = fq_dot<2>(aa, bb); } fq_pin(r.c0);
    { const TA0 aa[2] = {a.c0, a.c1}; const TB0 bb[2] = {b.c1, b.c0}; r.c1 = fq_dot<2>(aa, bb); }
    return r;
}
template <uint64_t L1, int V1>
__device__ __forceinline__ Fq2n f2_sqrd(const Fq2T<L1, V1>& a) {
    Fq2n r;
    if constexpr (FQ2_BETA == 1) { r.c0 = fq_mul(fq_add(a.c0, a.c1), fq_sub(a.c0, a.c1)); fq_pin(r.c0); }
    else {                                                              // a0 a0 + (K - beta a1) a1
        const auto na1 = fq_neg(fq_mul_beta(a.c1)); using TA = decltype(na1);
        const TA aa[2] = {fq_widen<TA::LMAX, TA::VMAXB>(a.c0), na1}; const Fq<L1, V1> bb[2] = {a.c0, a.c1};
        r.c0 = fq_dot<2>(aa, bb); fq_pin(r.c0);
    }
    r.c1 = fq_mul(fq_dbl(a.c0), a.c1);
    return r;
}
template <uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ Fq2n f2_mul_fq(const Fq2T<L1, V1>& a, const Fq<L2, V2>& s) { Fq2n r; r.c0 = fq_mul(a.c0, s); fq_pin(r.c0); r.c1 = fq_mul(a.c1, s); return r; }
template <uint64_t L1, int V1> __device__ __forceinline__ Fq2C f2_to_coord(const Fq2T<L1, V1>& a) { return {fq_coord(a.c0), fq_coord(a.c1)}; }

// engine Fp2 (Mont-384) -> carry-free form: times 2^8 is a re-slicing, then one quotient estimate (fq_curve.hpp fq_from_fp_fast: no Montgomery product)
__device__ __forceinline__ Fq2n f2_from(const Fp2& x) { return {fq_from_fp_fast(x.c0), fq_from_fp_fast(x.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ Fp2 f2_to(const Fq2T<L1, V1>& a) { return {fq_to_fp(fq_reduce(a.c0)), fq_to_fp(fq_reduce(a.c1))}; }

// A Jacobian point of G2 between two group operations.  BLS12-381 (u^2 = -1): a mixed addition leaves (X, Y, Z) < (11, 7, 8) p and these bounds pass every
// static check of the next operation (an Fp2 square needs 2V (2V + 1) <= 2 500: the widest square is (Z + H)^2 with Z + H < 22p), so its five values that
// used to be REDUCED (H, r, X3, Y3, Z3: ten quotient estimates + signed carry passes, a fifth of a field product each) are only carry-normalised
// (fq_curve.hpp JacQ has the G1 side and the measurements).  A doubling still reduces X3 and Y3 (E^2 - 2D < 36p would not fit the next square).
// BLS12-377 (u^2 = -5: a product's bound carries the factor 5) keeps every coordinate below 4p as before.
#if defined(RIPP_BLS12_377)
using Fq2X = Fq2C; using Fq2Y = Fq2C; using Fq2Z = Fq2C;
#else
using Fq2X = Fq2T<FQ_LN, 11>; using Fq2Y = Fq2T<FQ_LN, 7>; using Fq2Z = Fq2T<FQ_LN, 8>;
#endif
struct JacQ2 { Fq2X x; Fq2Y y; Fq2Z z; };
// table operands of a mixed addition (BLS12-381): 2^8 x < 256p re-sliced, -y2 limb-wise (fq_curve.hpp fq_tab / fq_tab_y)
__device__ __forceinline__ Fq2T<FQ_LN, 256> f2_tab(const Fp2& v) { return {fq_tab(v.c0), fq_tab(v.c1)}; }
__device__ __forceinline__ Fq2T<FqTabY::LMAX, 258> f2_tab_y(const Fp2& v, bool negate) { return {fq_tab_y(v.c0, negate), fq_tab_y(v.c1, negate)}; }
// a value into a coordinate slot: widened / carry-normalised when its bound fits the slot's, reduced otherwise
template <class T, uint64_t LM, int VB> __device__ __forceinline__ T fq_slot_r(const Fq<LM, VB>& a) {
    if constexpr (VB <= T::VMAXB && LM <= FQ_LN) return fq_widen<FQ_LN, T::VMAXB>(a);
    else if constexpr (VB <= T::VMAXB) return fq_widen<FQ_LN, T::VMAXB>(fq_norm(a));
    else return fq_widen<FQ_LN, T::VMAXB>(fq_reduce(a));
}
template <class T2, uint64_t L1, int V1> __device__ __forceinline__ T2 f2_slot(const Fq2T<L1, V1>& a) { return {fq_slot_r<decltype(T2::c0)>(a.c0), fq_slot_r<decltype(T2::c1)>(a.c1)}; }
// the values a mixed addition carries between its products: normalised where the next square's bound allows it (BLS12-381), reduced otherwise
template <uint64_t L1, int V1> __device__ __forceinline__ auto f2_lazy(const Fq2T<L1, V1>& a) {
#if defined(RIPP_BLS12_377)
    return f2_reduce(a);
#else
    return f2_norm(a);
#endif
}
#if !defined(RIPP_BLS12_377)
// a b - c d over Fp2 (u^2 = -1) with ONE reduction per part: two lazily reduced sums of four products (fq_curve.hpp fq_mul_sub has the Fp form)
template <class TA, class TB, class TC, class TD>
__device__ __forceinline__ Fq2n f2_muld_sub(const TA& a, const TB& b, const TC& c, const TD& d) {
    const auto na1 = fq_neg(a.c1); const auto nc0 = fq_neg(c.c0); const auto nc1 = fq_neg(c.c1);
    using A0 = decltype(a.c0); using NA = decltype(na1); using NC = decltype(nc0); using C1 = decltype(c.c1);
    using B0 = decltype(b.c0); using D0 = decltype(d.c0);
    constexpr uint64_t m1 = A0::LMAX > NA::LMAX ? A0::LMAX : NA::LMAX, m2 = NC::LMAX > C1::LMAX ? NC::LMAX : C1::LMAX, L1 = m1 > m2 ? m1 : m2;
    constexpr int v1 = A0::VMAXB > NA::VMAXB ? A0::VMAXB : NA::VMAXB, v2 = NC::VMAXB > C1::VMAXB ? NC::VMAXB : C1::VMAXB, V1 = v1 > v2 ? v1 : v2;
    constexpr uint64_t L2 = B0::LMAX > D0::LMAX ? B0::LMAX : D0::LMAX;
    constexpr int V2 = B0::VMAXB > D0::VMAXB ? B0::VMAXB : D0::VMAXB;
    using T1 = Fq<L1, V1>; using T2 = Fq<L2, V2>;
    auto w1 = [](const auto& x) { return fq_widen<L1, V1>(x); };
    auto w2 = [](const auto& x) { return fq_widen<L2, V2>(x); };
    Fq2n r;
    { const T1 aa[4] = {w1(a.c0), w1(na1), w1(nc0), w1(c.c1)}; const T2 bb[4] = {w2(b.c0), w2(b.c1), w2(d.c0), w2(d.c1)}; r.c0 = fq_dot<4>(aa, bb); } fq_pin(r.c0);
    { const T1 aa[4] = {w1(a.c0), w1(a.c1), w1(nc0), w1(nc1)}; const T2 bb[4] = {w2(b.c1), w2(b.c0), w2(d.c1), w2(d.c0)}; r.c1 = fq_dot<4>(aa, bb); }
    return r;
}
#endif
template <class TX, class TY, class TZ> __device__ __forceinline__ void j2_set(JacQ2& a, const TX& x, const TY& y, const TZ& z) { a.x = f2_slot<Fq2X>(x); a.y = f2_slot<Fq2Y>(y); a.z = f2_slot<Fq2Z>(z); }
__device__ __forceinline__ void j2_set_identity(JacQ2& a) { j2_set(a, Fq2n{fq_one(), fq_zero()}, Fq2n{fq_one(), fq_zero()}, Fq2n{fq_zero(), fq_zero()}); }
// dbl-2009-l / madd-2007-bl in the low-liveness order of kernels.hpp (jdbl_lo / jmadd_lo), every product inlined and PINNED where it is written:
// with the out-of-line Karatsuba products (f2_mul above) these two kept ~340 dwords per lane in scratch, with inlined products and
// sched_barriers between them 84 -- the barriers bind only the machine scheduler, the DAG had already interleaved the products.
__device__ __forceinline__ void jdbl2_q(JacQ2& p) {
    p.z = f2_slot<Fq2Z>(f2_dbl(f2_muld(p.y, p.z))); f2_pin(p.z);
    Fq2n A = f2_sqrd(p.x); f2_pin(A);
    Fq2n B = f2_sqrd(p.y); f2_pin(B);
    Fq2n t = f2_sqrd(f2_norm(f2_add(p.x, B))); f2_pin(t);
    Fq2n C = f2_sqrd(B); f2_pin(C);
    auto D = f2_norm(f2_dbl(f2_sub(f2_sub(t, A), C))); f2_pin(D);
    auto E = f2_norm(f2_add(f2_dbl(A), A)); f2_pin(E);
    Fq2n X3 = f2_reduce(f2_sub(f2_sub(f2_sqrd(E), D), D)); f2_pin(X3);
    p.y = f2_slot<Fq2Y>(f2_sub(f2_muld(E, f2_norm(f2_sub(D, X3))), f2_dbl(f2_dbl(f2_dbl(C))))); f2_pin(p.y);
    p.x = f2_slot<Fq2X>(X3);
}
// loadx / loady fetch the affine addend (Fq2n); park: this lane's LDS column (stride 64 lanes, 7 x 16 B).
// Returns true when the result is NOT valid (H = 0: T = +-Q).
template <class LOADX, class LOADY>
__device__ __forceinline__ bool jmadd2_q(JacQ2& p, LOADX loadx, LOADY loady, uint4* park) {
    Fq2n Z1Z1 = f2_sqrd(p.z); f2_pin(Z1Z1);
    auto H = f2_lazy(f2_sub(f2_muld(loadx(), Z1Z1), p.x)); f2_pin(H);
    Fq2n t = f2_muld(Z1Z1, p.z); f2_pin(t);
    auto r = f2_lazy(f2_dbl(f2_sub(f2_muld(loady(), t), p.y))); f2_pin(r);
    { uint32_t w[28];                                                      // Y1 rests in LDS until the last product
#pragma unroll
      for (int k = 0; k < 14; ++k) { w[k] = p.y.c0.l[k]; w[14 + k] = p.y.c1.l[k]; }
      const uint4* src = reinterpret_cast<const uint4*>(w);
#pragma unroll
      for (int k = 0; k < 7; ++k) park[k * 64] = src[k]; }
    Fq2n HH = f2_sqrd(H); f2_pin(HH);
    const bool special = f2_is_zero(HH);                                   // H = 0 iff H^2 = 0: HH is a reduced value, H need not be
    p.z = f2_slot<Fq2Z>(f2_sub(f2_sub(f2_sqrd(f2_norm(f2_add(p.z, H))), Z1Z1), HH)); f2_pin(p.z);
    const auto I = f2_dbl(f2_dbl(HH));                                     // 4 HH, lazy
    Fq2n J = f2_muld(H, I); f2_pin(J);
    Fq2n V = f2_muld(p.x, I); f2_pin(V);
    auto X3 = f2_lazy(f2_sub(f2_sub(f2_sub(f2_sqrd(r), J), V), V)); f2_pin(X3);
    { uint4 q[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) q[k] = park[k * 64];
      const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
      Fq2Y y1;
#pragma unroll
      for (int k = 0; k < 14; ++k) { y1.c0.l[k] = w[k]; y1.c1.l[k] = w[14 + k]; }
#if defined(RIPP_BLS12_377)
      Fq2n t2 = f2_muld(J, y1); f2_pin(t2);
      p.y = f2_slot<Fq2Y>(f2_sub(f2_muld(r, f2_norm(f2_sub(V, X3))), f2_dbl(t2))); f2_pin(p.y);
#else
      p.y = f2_slot<Fq2Y>(f2_muld_sub(r, f2_norm(f2_sub(V, X3)), f2_dbl(J), y1)); f2_pin(p.y);             // r (V - X3) - 2 J Y1: eight products, two reductions
#endif
    }
    p.x = f2_slot<Fq2X>(X3);
    return special;
}
#endif

// NS digit strings; string t works on table rows t M .. t M + M - 1 (kernels.hpp k_fold_g2_tab).  Exceptional lanes are FLAGGED (flag[i] = 1) and redone
// with the complete formulas by k_fold_g2_tab_fix, launched behind this kernel: the out-of-line fallback and its stack frame stay out of the
// throughput kernel (as a callee its frame was the kernel's whole private segment: 870 B per lane with no register spilled).  -DRIPP_INLINE_FALLBACK
// keeps the callee (A/B: the same round times -- a *_fix launch waits for a free SIMD while the G1 fold of the other stream fills the chip, but so does
// whatever kernel comes next on this stream).
template <class D, int NS>
__global__ void __launch_bounds__(64, 2) k_fold_g2_tab_q(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t half, D dg, G2J* __restrict__ out, uint8_t* __restrict__ flag) {
    __shared__ uint4 park_[7 * 64];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* park = park_ + threadIdx.x;
    JacQ2 acc; j2_set_identity(acc);
    bool inf = true, bad = false;                                        // inf is wave-uniform: the digit strings are shared by the launch
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl2_q(acc);
#pragma unroll 1
        for (int t = 0; t < NS; ++t) {
            const int d = dg.d[t][pos];
            if (d == 0) continue;
            uint32_t il = i; asm volatile("" : "+v"(il));                     // (the loop-invariant qtab + i is re-formed per addition instead of living in a spilled register pair)
            const uint4* base = qtab + ((size_t)t * M + ((d < 0 ? -d : d) >> 1)) * G2A_CHUNKS * stride + il;
            auto ldfp2 = [&](int q0) { Fp2 v; uint4* dd = reinterpret_cast<uint4*>(&v); const uint4* b2 = opaque(base);      // re-loaded where it is used (merged with the identity check below, x and y stayed alive from the top of the addition)
#pragma unroll
                for (int q = 0; q < 6; ++q) dd[q] = b2[(size_t)(q0 + q) * stride]; return v; };
#if defined(RIPP_BLS12_377)
            auto loadx = [&]() { return f2_from(ldfp2(0)); };
            auto loady = [&]() { const Fp2 y = ldfp2(6); return f2_from(d < 0 ? neg(y) : y); };
#else
            auto loadx = [&]() { return f2_tab(ldfp2(0)); };                   // table operands are not reduced (fq_curve.hpp fq_tab): the products absorb the factor 2^8
            auto loady = [&]() { return f2_tab_y(ldfp2(6), d < 0); };
#endif
            if (inf) {                                                    // first addition: acc <- +-Q
                const Fp2 x0 = ldfp2(0), y0 = ldfp2(6);
                bad |= x0.is_zero() && y0.is_zero();
                j2_set(acc, f2_from(x0), f2_from(d < 0 ? neg(y0) : y0), Fq2n{fq_one(), fq_zero()}); inf = false;
            } else {
                { const Fp2 x0 = ldfp2(0); bad |= x0.is_zero() && ldfp2(6).is_zero(); }      // a table point at infinity
                bad |= jmadd2_q(acc, loadx, loady, park);
            }
        }
    }
    const G2A* lp = lo + i;                                             // (re-read where it is used: held in registers, the 48 words of lo[i] were the kernel's only spills)
    if (inf) { out[i] = to_jac(*lp);
#if !defined(RIPP_INLINE_FALLBACK)
        flag[i] = 0;
#endif
        return; }
    bool linf; { const G2A l = *lp; linf = is_inf(l); }
    if (!linf) bad |= jmadd2_q(acc, [&]() { return f2_from(opaque(lp)->x); }, [&]() { return f2_from(opaque(lp)->y); }, park);
#if !defined(RIPP_INLINE_FALLBACK)
    flag[i] = bad;
    if (!bad) out[i] = G2J{f2_to(acc.x), f2_to(acc.y), f2_to(acc.z)};
#else
    if (bad) fold_g2_tab_complete<D, NS>(qtab, stride, M, lo, i, dg, out);
    else out[i] = G2J{f2_to(acc.x), f2_to(acc.y), f2_to(acc.z)};
#endif
#endif
}
template <class D, int NS>
__global__ void __launch_bounds__(64) k_fold_g2_tab_fix(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t half, D dg, G2J* __restrict__ out, const uint8_t* __restrict__ flag) {
    for_flagged(flag, half, [&](uint32_t i) { fold_g2_tab_complete<D, NS>(qtab, stride, M, lo, i, dg, out); });
}

// ---- the FUSED fold of rounds 0 and 1 on the x-scaled G2 vector (engine.hip job_fold_fused; fq_curve.hpp has the G1 side).  With the quarters
// B0 .. B3 of the round-0 vector and both challenges known,   bt''_i = (x0 x1) B0_i + x0 B1_i + x1 B2_i + B3_i   (bt'' = x0 x1 b'': the vector
// stays x-scaled).  Three digit sets over the round-0 kind of table (rows (4 b + j) M + m = psi^j((2 m + 1) 2^(16 b) Q)), built over B0 | B1 | B2:
// set 0: x0 x1 (full width: four GLS digits) on B0 (element i), set 1: x0 on B1 (q + i), set 2: x1 on B2 (2 q + i) -- 17 doublings + ~32 x 3.2
// additions per output where the two folds take 2 x (17 + ~45) for round 0 and 65 + ~52 plus its in-round tables for round 1.
struct Wnaf16x3 { Wnaf16 s[3]; int len; };
__device__ __noinline__ inline void fold_g2_fused_complete(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t q, uint32_t i, const Wnaf16x3& dg, G2J* __restrict__ out) {
    G2J acc = jac_inf<Fp2>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
#pragma unroll 1
        for (int u = 0; u < 3; ++u) {
#pragma unroll 1
            for (int t = 0; t < 16; ++t) {
                const int d = dg.s[u].d[t][pos];
                if (d != 0) {
                    G2A p = load_chunks<G2A_CHUNKS, G2A>(qtab, (size_t)t * M + ((d < 0 ? -d : d) >> 1), stride, (size_t)u * q + i);
                    if (d < 0) p.y = neg(p.y);
                    acc = add_mixed(acc, p);
                }
            }
        }
    }
    out[i] = add_mixed(acc, lo[i]);
}
__global__ void __launch_bounds__(64, 2) k_fold_g2_fused_q(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t q, Wnaf16x3 dg, G2J* __restrict__ out, uint8_t* __restrict__ flag) {
    __shared__ uint4 park_[7 * 64];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* park = park_ + threadIdx.x;
    JacQ2 acc; j2_set_identity(acc);
    bool inf = true, bad = false;                                        // inf is wave-uniform: the digit strings are shared by the launch
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl2_q(acc);
#pragma unroll 1
        for (int ut = 0; ut < 48; ++ut) {
            const int u = ut >> 4, t = ut & 15;
            const int d = dg.s[u].d[t][pos];
            if (d == 0) continue;
            uint32_t il = i; asm volatile("" : "+v"(il));                     // (re-formed per addition instead of living in a spilled register pair)
            const uint4* base = qtab + ((size_t)t * M + ((d < 0 ? -d : d) >> 1)) * G2A_CHUNKS * stride + (size_t)u * q + il;
            auto ldfp2 = [&](int q0) { Fp2 v; uint4* dd = reinterpret_cast<uint4*>(&v); const uint4* b2 = opaque(base);
#pragma unroll
                for (int c = 0; c < 6; ++c) dd[c] = b2[(size_t)(q0 + c) * stride]; return v; };
#if defined(RIPP_BLS12_377)
            auto loadx = [&]() { return f2_from(ldfp2(0)); };
            auto loady = [&]() { const Fp2 y = ldfp2(6); return f2_from(d < 0 ? neg(y) : y); };
#else
            auto loadx = [&]() { return f2_tab(ldfp2(0)); };                   // table operands are not reduced (fq_curve.hpp fq_tab): the products absorb the factor 2^8
            auto loady = [&]() { return f2_tab_y(ldfp2(6), d < 0); };
#endif
            if (inf) {                                                    // first addition: acc <- +-Q
                const Fp2 x0 = ldfp2(0), y0 = ldfp2(6);
                bad |= x0.is_zero() && y0.is_zero();
                j2_set(acc, f2_from(x0), f2_from(d < 0 ? neg(y0) : y0), Fq2n{fq_one(), fq_zero()}); inf = false;
            } else {
                { const Fp2 x0 = ldfp2(0); bad |= x0.is_zero() && ldfp2(6).is_zero(); }      // a table point at infinity
                bad |= jmadd2_q(acc, loadx, loady, park);
            }
        }
    }
    const G2A* lp = lo + i;
    if (inf) { out[i] = to_jac(*lp); flag[i] = 0; return; }
    bool linf; { const G2A l = *lp; linf = is_inf(l); }
    if (!linf) bad |= jmadd2_q(acc, [&]() { return f2_from(opaque(lp)->x); }, [&]() { return f2_from(opaque(lp)->y); }, park);
    flag[i] = bad;
    if (!bad) out[i] = G2J{f2_to(acc.x), f2_to(acc.y), f2_to(acc.z)};
#endif
}
__global__ void __launch_bounds__(64) k_fold_g2_fused_fix(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t q, Wnaf16x3 dg, G2J* __restrict__ out, const uint8_t* __restrict__ flag) {
    for_flagged(flag, q, [&](uint32_t i) { fold_g2_fused_complete(qtab, stride, M, lo, q, i, dg, out); });
}

// Odd multiples 3 Q, 5 Q, .., (2 M - 1) Q of the fold tables (kernels.hpp k_odd_multiples: out[m][i] = (2m + 3) base[i], Jacobian, batch-normalised
// by the caller) on the carry-free form.  k_odd_multiples<Fp2> chains M - 2 GENERAL Jacobian additions t += 2Q (16 Fp2 products each, out-of-line
// 12-word products: 1 448 B of scratch per lane, 63 % of its own issue roof).  Here the chain runs on the isomorphic curve on which 2Q = (X2, Y2, Z2)
// is AFFINE -- (x, y) -> (x Z2^2, y Z2^3) -- so every step is a mixed addition with the same affine addend (X2, Y2) (11 products), and a result
// (X', Y', Z') is the point (X', Y', Z' Z2) of the original curve (one more product).  (X2, Y2) rest in the element's LAST output slot until the
// last addition has read them, Z2 and the idle Y1 in LDS.  Exceptional elements (infinity, small order) are flagged and redone with the complete
// formulas by k_odd_multiples_fix.
template <class F>
__device__ __noinline__ void odd_multiples_complete(const Affine<F>* __restrict__ base, uint32_t n, int M, uint32_t i, Jac<F>* __restrict__ out) {
    const Affine<F> b = base[i];
    const Jac<F> b2 = dbl(to_jac(b));
    Jac<F> t = add_mixed(b2, b);
    out[i] = t;
#pragma unroll 1
    for (int m = 1; m < M - 1; ++m) { t = add(t, b2); out[(size_t)m * n + i] = t; }
}
__global__ void __launch_bounds__(64, 2) k_odd_multiples_q(const G2A* __restrict__ base, uint32_t n, int M, G2J* __restrict__ out, uint8_t* __restrict__ flag) {
    __shared__ uint4 park_[14 * 64];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || M < 2) return;
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* park = park_ + threadIdx.x;
    uint4* zpark = park + 7 * 64;
    auto st7 = [](uint4* dst, size_t stride, const Fq2n& v) { uint32_t w[28];
#pragma unroll
        for (int k = 0; k < 14; ++k) { w[k] = v.c0.l[k]; w[14 + k] = v.c1.l[k]; }
        const uint4* src = reinterpret_cast<const uint4*>(w);
#pragma unroll
        for (int k = 0; k < 7; ++k) dst[k * stride] = src[k]; };
    auto ld7 = [](const uint4* src, size_t stride) { uint4 q[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = src[k * stride];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
        Fq2n v;
#pragma unroll
        for (int k = 0; k < 14; ++k) { v.c0.l[k] = w[k]; v.c1.l[k] = w[14 + k]; }
        return v; };
    const G2A* bp = base + i;
    bool bad;
    { const G2A b = *bp; bad = is_inf(b); }
    uint4* tmp = reinterpret_cast<uint4*>(out + (size_t)(M - 2) * n + i);       // 14 of the slot's 18 chunks: X2, Y2
    JacQ2 t;
    {
        JacQ2 d; j2_set(d, f2_from(opaque(bp)->x), f2_from(opaque(bp)->y), Fq2n{fq_one(), fq_zero()});
        f2_pin(d.z);
        jdbl2_q(d);
        Fq2n Z2 = f2_reduce(d.z); f2_pin(Z2);
        bad |= f2_is_zero(Z2);
        st7(zpark, 64, Z2);
        st7(tmp, 1, f2_reduce(d.x)); st7(tmp + 7, 1, f2_reduce(d.y));
        Fq2n zz = f2_sqrd(Z2); f2_pin(zz);
        { Fq2n px = f2_muld(f2_from(opaque(bp)->x), zz); f2_pin(px); t.x = f2_slot<Fq2X>(px); }
        Fq2n zzz = f2_muld(zz, Z2); f2_pin(zzz);
        { Fq2n py = f2_muld(f2_from(opaque(bp)->y), zzz); f2_pin(py); t.y = f2_slot<Fq2Y>(py); }
        t.z = f2_slot<Fq2Z>(Fq2n{fq_one(), fq_zero()}); f2_pin(t.z);
    }
#pragma unroll 1
    for (int m = 0; m < M - 1; ++m) {
        bad |= jmadd2_q(t, [&]() { return ld7(opaque(tmp), 1); }, [&]() { return ld7(opaque(tmp) + 7, 1); }, park);
        Fq2n zo = f2_muld(t.z, ld7(zpark, 64)); f2_pin(zo);
        out[(size_t)m * n + i] = G2J{f2_to(t.x), f2_to(t.y), f2_to(zo)};
    }
#if !defined(RIPP_INLINE_FALLBACK)
    flag[i] = bad;
#else
    if (bad) odd_multiples_complete<Fp2>(base, n, M, i, out);
#endif
#endif
}
__global__ void __launch_bounds__(64) k_odd_multiples_fix(const G2A* __restrict__ base, uint32_t n, int M, G2J* __restrict__ out, const uint8_t* __restrict__ flag) {
    if (M < 2) return;
    for_flagged(flag, n, [&](uint32_t i) { odd_multiples_complete<Fp2>(base, n, M, i, out); });
}

// The 4-lane GLS fold of the latency-bound rounds (kernels.hpp k_fold_g2_gls_split: lane (i, j) multiplies psi^j(hi[i]) by digit string j; the
// combine kernel sums the four parts) on the carry-free form: the chain of 33 doublings + ~11 additions is ~1 150 Fp products per lane at one wave
// per SIMD, i.e. pure latency -- ~420 instead of ~600 instructions per product and no scratch traffic (the 12-word form spills 262 dwords).
// The first non-zero digit LOADS the point; an exceptional addition (acc = +-Q) or an identity input flags the lane for k_fold_g2_gls_split_fix.
__device__ __noinline__ inline G2J fold_g2_gls_split_complete(const G2A& q, const GlsDigits& dg, int j) {
    G2J acc = jac_inf<Fp2>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
        const int d = dg.d[j][pos];
        if (d != 0) { G2A t = q; if (d < 0) t.y = neg(t.y); acc = add_mixed(acc, t); }
    }
    return acc;
}
__global__ void __launch_bounds__(64, 2) k_fold_g2_gls_split_q(const G2A* __restrict__ hi, uint32_t half, GlsDigits dg, G2J* __restrict__ parts /* [4][half] */, uint8_t* __restrict__ flag /* [4][half] */) {
    __shared__ uint4 park_[21 * 64];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const int j = blockIdx.y;
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* park = park_ + threadIdx.x;
    uint4* qpark = park + 7 * 64;                                       // psi^j(Q) in the carry-free form: x at chunks 7..13, y at 14..20
    auto st7 = [](uint4* dst, const Fq2n& v) { uint32_t w[28];
#pragma unroll
        for (int k = 0; k < 14; ++k) { w[k] = v.c0.l[k]; w[14 + k] = v.c1.l[k]; }
        const uint4* src = reinterpret_cast<const uint4*>(w);
#pragma unroll
        for (int k = 0; k < 7; ++k) dst[k * 64] = src[k]; };
    auto ld7 = [](const uint4* src) { uint4 q[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = src[k * 64];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
        Fq2n v;
#pragma unroll
        for (int k = 0; k < 14; ++k) { v.c0.l[k] = w[k]; v.c1.l[k] = w[14 + k]; }
        return v; };
    bool bad;
    {
        const G2A q = gls_image(hi[i], j);
        bad = is_inf(q);
        st7(qpark, f2_from(q.x)); st7(qpark + 7 * 64, f2_from(q.y));
    }
    JacQ2 acc; j2_set_identity(acc);
    bool inf = true;                                                    // wave-uniform: one digit string per wave
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl2_q(acc);
        const int d = dg.d[j][pos];
        if (d == 0) continue;
        auto loadx = [&]() { return ld7(qpark); };
        auto loady = [&]() { const Fq2n y = ld7(qpark + 7 * 64); if (d > 0) return y; return Fq2n{fq_reduce(fq_neg(y.c0)), fq_reduce(fq_neg(y.c1))}; };
        if (inf) { j2_set(acc, loadx(), loady(), Fq2n{fq_one(), fq_zero()}); f2_pin(acc.z); inf = false; }
        else bad |= jmadd2_q(acc, loadx, loady, park);
    }
#if !defined(RIPP_INLINE_FALLBACK)
    flag[(size_t)j * half + i] = bad;
    if (bad) return;
#else
    if (bad) { parts[(size_t)j * half + i] = fold_g2_gls_split_complete(gls_image(hi[i], j), dg, j); return; }
#endif
    if (inf) parts[(size_t)j * half + i] = jac_inf<Fp2>();
    else parts[(size_t)j * half + i] = G2J{f2_to(acc.x), f2_to(acc.y), f2_to(acc.z)};
#endif
}
__global__ void __launch_bounds__(64) k_fold_g2_gls_split_fix(const G2A* __restrict__ hi, uint32_t half, GlsDigits dg, G2J* __restrict__ parts, const uint8_t* __restrict__ flag) {
    for_flagged(flag, 4 * half, [&](uint32_t f) { const int j = (int)(f / half); const uint32_t i = f - (uint32_t)j * half; parts[f] = fold_g2_gls_split_complete(gls_image(hi[i], j), dg, j); });
}

// 2^k * in[i] on G2 (Jacobian out): the carry-free twin of k_pow2_mul<Fp2> (16 doublings per table base, 32 for the pre-doubled second base)
__global__ void __launch_bounds__(64, 2) k_pow2_mul_g2_q(const G2A* __restrict__ in, uint32_t n, int k, G2J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const G2A* bp = in + i;
    bool inf; { const G2A b = *bp; inf = is_inf(b); }
    if (inf) { out[i] = jac_inf<Fp2>(); return; }
    JacQ2 d; j2_set(d, f2_from(opaque(bp)->x), f2_from(opaque(bp)->y), Fq2n{fq_one(), fq_zero()});
    f2_pin(d.z);
#pragma unroll 1
    for (int t = 0; t < k; ++t) jdbl2_q(d);
    out[i] = G2J{f2_to(d.x), f2_to(d.y), f2_to(d.z)};
#endif
}

}  // namespace ripp
