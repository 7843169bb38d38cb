// G1 group law and fold kernels on the carry-free field form (fq28.hpp): the throughput twins of kernels.hpp's k_fold_affine_naf<Fp> /
// k_fold_g1_tab.  Same formulas (dbl-2009-l, madd-2007-bl), written over lazily bounded values: additions and subtractions are limb-wise
// (no carry chains, no conditional subtractions), every multiplication is 196 + 196 multiply-adds instead of 288 + 288 with carries, and
// the compiler checks every bound (fq28.hpp).  Coordinates are carried between group operations carry-normalised and UNREDUCED (JacQ below).
// The low-liveness forms do not cover the exceptional cases (T = +-Q, an operand at infinity): they REPORT them and the kernel recomputes
// such a lane with the complete formulas of curve.hpp, exactly like k_fold_g2_tab.
#pragma once
#include "fq28.hpp"
#include "kernels.hpp"

namespace ripp {

#if defined(__HIP_DEVICE_COMPILE__)
using FqC = Fq<FQ_LN, 4>;                 // a coordinate between two group operations

// value < VB p (VB <= ~2000)  ->  the representative in [0, 2p) with normalised limbs: one quotient estimate from the top limb
// (never too large, at most one too small) and one signed carry pass
template <uint64_t LM, int VB>
__device__ __forceinline__ Fqn fq_reduce(const Fq<LM, VB>& a) {
    using namespace fq28;
    static_assert((uint64_t)VB * (P_TOP + 1) < ((uint64_t)1 << 31), "top limb must stay a positive int");
    const Fq<FQ_LN, VB> n = fq_norm(a);
    constexpr float INV = (1.0f - 1.0f / 1048576.0f) / (float)(P_TOP + 1);
    const int q = (int)((float)n.l[NL - 1] * INV);
    Fqn r; int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) { const int64_t t = (int64_t)n.l[i] - (int64_t)q * (int32_t)P28.l[i] + carry; r.l[i] = (uint32_t)t & MASK; carry = t >> W; }
    r.l[NL - 1] = (uint32_t)((int64_t)n.l[NL - 1] - (int64_t)q * (int32_t)P28.l[NL - 1] + carry);
    return r;
}
// a == 0 mod p, for a reduced value (< 2p, normalised): all limbs zero, or equal to p
__device__ __forceinline__ bool fq_is_zero(const Fqn& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < fq28::NL; ++i) { z |= a.l[i]; e |= a.l[i] ^ fq28::P28.l[i]; }
    return z == 0 || e == 0;
}
template <uint64_t LM, int VB> __device__ __forceinline__ FqC fq_coord(const Fq<LM, VB>& a) {
    if constexpr (LM <= FQ_LN && VB <= 4) return fq_widen<FQ_LN, 4>(a);
    else if constexpr (VB <= 4) return fq_widen<FQ_LN, 4>(fq_norm(a));
    else return fq_widen<FQ_LN, 4>(fq_reduce(a));
}

// A Jacobian point between two group operations.  Its coordinates are NOT reduced: they keep the bounds the formulas leave them with -- after a doubling
// (X, Y, Z) < (36, 19, 4) p, after a mixed addition < (11, 7, 8) p, whatever the bounds of the operands (every input first meets a product) -- and are only
// carry-normalised: a reduction (quotient estimate + a signed 64-bit carry pass) costs a fifth of a field product, a normalisation a few per cent
// (tools/ubench/fqgroup.hip), and the group law used four per addition and two per doubling.  The compiler checks every bound (fq28.hpp).
using JX = Fq<FQ_LN, 36>; using JY = Fq<FQ_LN, 19>; using JZ = Fq<FQ_LN, 8>;
struct JacQ { JX x; JY y; JZ z; };
// a b - c d with ONE Montgomery reduction: a lazily reduced sum of two products whose second term enters as (K - c) d (K a multiple of p above c, limb-wise).
// Y3 of the addition formulas is such a difference: one reduction (196 multiply-adds + its glue) less per addition, and the result is a reduced value.
template <class TA, class TB, class TC, class TD>
__device__ __forceinline__ Fqn fq_mul_sub(const TA& a, const TB& b, const TC& c, const TD& d) {
    const auto nc = fq_neg(c);
    using TN = decltype(nc);
    constexpr uint64_t L1 = TA::LMAX > TN::LMAX ? TA::LMAX : TN::LMAX, L2 = TB::LMAX > TD::LMAX ? TB::LMAX : TD::LMAX;
    constexpr int V1 = TA::VMAXB > TN::VMAXB ? TA::VMAXB : TN::VMAXB, V2 = TB::VMAXB > TD::VMAXB ? TB::VMAXB : TD::VMAXB;
    const Fq<L1, V1> aa[2] = {fq_widen<L1, V1>(a), fq_widen<L1, V1>(nc)};
    const Fq<L2, V2> bb[2] = {fq_widen<L2, V2>(b), fq_widen<L2, V2>(d)};
    return fq_dot<2>(aa, bb);
}
struct AffQ { Fqn x, y; };
// a value into a coordinate slot: carry-normalised where it is lazy; its bound must fit the slot's (a compile error otherwise, never a silent reduction)
template <class T, uint64_t LM, int VB> __device__ __forceinline__ T fq_slot(const Fq<LM, VB>& a) {
    static_assert(VB <= T::VMAXB, "coordinate bound exceeded: reduce the value first");
    if constexpr (LM <= FQ_LN) return fq_widen<FQ_LN, T::VMAXB>(a);
    else return fq_widen<FQ_LN, T::VMAXB>(fq_norm(a));
}
template <class TX, class TY, class TZ> __device__ __forceinline__ void jq_set(JacQ& a, const TX& x, const TY& y, const TZ& z) { a.x = fq_slot<JX>(x); a.y = fq_slot<JY>(y); a.z = fq_slot<JZ>(z); }
__device__ __forceinline__ void jq_set_identity(JacQ& a) { jq_set(a, fq_one(), fq_one(), fq_zero()); }
// engine value (Mont-384) -> carry-free form: times 2^8 is a re-slicing of the words, then one quotient estimate -- a fifth of the Montgomery product by 2^400 that
// fq_from_fp spends (a table fold converts two coordinates per addition: 2 of its 12 products)
__device__ __forceinline__ Fqn fq_from_fp_fast(const Fp& x) { return fq_reduce(fq_unpack_shl8(x.l)); }
__device__ __forceinline__ AffQ affq_from(const G1A& p) { return {fq_from_fp_fast(p.x), fq_from_fp_fast(p.y)}; }
// A TABLE operand of a mixed addition is not even reduced: the re-sliced words are the value 2^8 x < 256p with normalised limbs, and both products it enters
// (x2 Z1Z1, y2 Z1) keep V1 V2 <= 2 500 (fq28.hpp VMAX = R' / p), so the Montgomery product absorbs the factor; -y2 is the limb-wise K - y2 (K = 257p).
using FqTab = Fq<FQ_LN, 256>;
using FqTabY = Fq<fq28::sub_lm(1, FQ_LN), 258>;                          // y2 or -y2
__device__ __forceinline__ FqTab fq_tab(const Fp& x) { return fq_unpack_shl8(x.l); }
__device__ __forceinline__ FqTabY fq_tab_y(const Fp& y, bool negate) {
    const FqTab v = fq_tab(y);
    FqTabY r;
    if (negate) r = fq_neg(v); else r = fq_widen<FqTabY::LMAX, 258>(v);
    return r;
}
__device__ __forceinline__ G1J jacq_to_g1j(const JacQ& p) { return {fq_to_fp(fq_reduce(p.x)), fq_to_fp(fq_reduce(p.y)), fq_to_fp(fq_reduce(p.z))}; }

// dbl-2009-l (a = 0): 2M + 5S.  Z = 0 maps to Z3 = 0.  Every product is PINNED where it is written (fq28.hpp fq_pin): the multiply-adds of a product are tied
// into chains (FQ_CHAIN), and without the pins the compiler starts several products before it finishes one and spills their operands.
// Bounds out: X3 = E^2 - 2D < (2 + 17 + 17) p, Y3 = E (D - X3) - 8C < (2 + 16 + 1) p, Z3 = 2 Y Z < 4p.
__device__ __forceinline__ void jdbl_q(JacQ& p) {
    Fqn A = fq_sqr(p.x); fq_pin(A);
    Fqn B = fq_sqr(p.y); fq_pin(B);
    Fqn C = fq_sqr(B); fq_pin(C);
    Fqn t = fq_sqr(fq_add(p.x, B)); fq_pin(t);
    auto D = fq_norm(fq_dbl(fq_sub(fq_sub(t, A), C))); fq_pin(D);                   // < 16p
    const auto E = fq_add(fq_dbl(A), A);
    auto X3 = fq_norm(fq_sub(fq_sub(fq_sqr(E), D), D)); fq_pin(X3);
    Fqn yz = fq_mul(p.y, p.z); fq_pin(yz);
    p.y = fq_slot<JY>(fq_sub(fq_mul(E, fq_sub(D, X3)), fq_dbl(fq_dbl(fq_dbl(C))))); fq_pin(p.y);
    p.x = fq_slot<JX>(X3);
    p.z = fq_slot<JZ>(fq_dbl(yz));
}
// madd-2007-bl, q affine and NOT the identity, p NOT the identity.  Returns true when the result is not valid (H = 0: p = +-q) -- read off H^2, which the
// formula needs anyway and which is a reduced value (H itself is only normalised; H = 0 mod p iff H^2 = 0 mod p).
// Bounds out: X3 = r^2 - J - 2V < 11p, Y3 = r (V - X3) - 2 Y1 J < 2p (a two-product sum with its own reduction), Z3 = (Z1 + H)^2 - Z1Z1 - HH < 8p.
template <class TX2, class TY2>
__device__ __forceinline__ bool jmadd_q(JacQ& p, const TX2& x2, const TY2& y2) {
    Fqn Z1Z1 = fq_sqr(p.z); fq_pin(Z1Z1);
    auto H = fq_norm(fq_sub(fq_mul(x2, Z1Z1), p.x)); fq_pin(H);                     // < 39p
    Fqn yz = fq_mul(y2, p.z); fq_pin(yz);
    auto rr = fq_norm(fq_dbl(fq_sub(fq_mul(yz, Z1Z1), p.y))); fq_pin(rr);            // < 44p
    Fqn HH = fq_sqr(H); fq_pin(HH);
    const bool special = fq_is_zero(HH);
    const auto I = fq_dbl(fq_dbl(HH));
    Fqn J = fq_mul(H, I); fq_pin(J);
    Fqn V = fq_mul(p.x, I); fq_pin(V);
    auto X3 = fq_norm(fq_sub(fq_sub(fq_sub(fq_sqr(rr), J), V), V)); fq_pin(X3);
    Fqn zh = fq_sqr(fq_add(p.z, H)); fq_pin(zh);
    p.z = fq_slot<JZ>(fq_sub(fq_sub(zh, Z1Z1), HH)); fq_pin(p.z);
    p.y = fq_slot<JY>(fq_mul_sub(rr, fq_norm(fq_sub(V, X3)), fq_dbl(p.y), J)); fq_pin(p.y);       // r (V - X3) - 2 Y1 J, one reduction
    p.x = fq_slot<JX>(X3);
    return special;
}
#endif

// out[i] = s * hi[i] + lo[i], NAF digit string shared by the launch (the G1 fold of a SIPP round, sipp/src/lib.rs:87-91): the carry-free twin of
// k_fold_affine_naf<Fp>.  The first non-zero digit loads the point instead of adding it, so the accumulator is never the identity inside the loop.
__global__ void __launch_bounds__(256, 2) k_fold_g1_naf_q(const G1A* __restrict__ hi, const G1A* __restrict__ lo, uint32_t half, NafDigits dg, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const G1A p = hi[i], l = lo[i];
    int pos = dg.len - 1;
    while (pos >= 0 && dg.d[pos] == 0) --pos;                    // uniform
    if (pos < 0 || is_inf(p)) { out[i] = to_jac(l); return; }   // s * hi = identity
    const AffQ q = affq_from(p);
    const Fqn ny = fq_reduce(fq_neg(q.y));
    JacQ acc; jq_set(acc, q.x, dg.d[pos] < 0 ? ny : q.y, fq_one());
    bool bad = false;
#pragma unroll 1
    for (--pos; pos >= 0; --pos) {
        jdbl_q(acc);
        const int d = dg.d[pos];
        if (d != 0) bad |= jmadd_q(acc, q.x, d < 0 ? ny : q.y);
    }
    if (!is_inf(l)) { const AffQ lq = affq_from(l); bad |= jmadd_q(acc, lq.x, lq.y); }
    if (bad) {                                                   // an addition met T = +-Q: the complete formulas, for this lane only
        G1J a2 = jac_inf<Fp>();
#pragma unroll 1
        for (int k = dg.len - 1; k >= 0; --k) { a2 = dbl(a2); const int d = dg.d[k]; if (d != 0) { G1A t = p; if (d < 0) t.y = neg(t.y); a2 = add_mixed(a2, t); } }
        out[i] = add_mixed(a2, l);
    } else out[i] = jacq_to_g1j(acc);
#endif
}

// out[i] = s * hi[i] + lo[i] with the FULL-WIDTH challenge of a GIPA round split through the GLV endomorphism (kernels.hpp k_fold_g1_glv: two NAF strings
// of ~128 digits on P and phi(P) = (beta x, y), shared by the launch): the carry-free twin, one 128-step chain per lane at ~420 instead of ~600
// instructions per Fp product.  The first non-zero digit LOADS its point; an exceptional addition redoes the lane with the complete formulas.
__global__ void __launch_bounds__(256, 2) k_fold_g1_glv_q(const G1A* __restrict__ hi, const G1A* __restrict__ lo, uint32_t half, GlvDigits dg, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const G1A* hp = hi + i; const G1A* lp = lo + i;                         // hi[i] / lo[i] are re-read where they are needed, not held for 128 steps
    AffQ q; Fqn bx, ny;
    {
        const G1A p = *hp;
        if (is_inf(p)) { out[i] = to_jac(*lp); return; }
        q = affq_from(p);
        bx = fq_mul(q.x, fq_from_fp(fp_const(RIPP_GLV_BETA)));             // phi(P) = (beta x, y)
        ny = fq_reduce(fq_neg(q.y));
    }
    JacQ acc; jq_set_identity(acc);
    bool inf = true, bad = false;                                         // inf is wave-uniform (shared digit strings)
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl_q(acc);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int d = h ? dg.d2[pos] : dg.d1[pos];
            if (d == 0) continue;
            const Fqn& x = h ? bx : q.x;
            if (inf) { jq_set(acc, x, d < 0 ? ny : q.y, fq_one()); inf = false; }
            else bad |= jmadd_q(acc, x, d < 0 ? ny : q.y);
        }
    }
    if (inf) { out[i] = to_jac(*lp); return; }
    { const G1A l = *opaque(lp); if (!is_inf(l)) { const AffQ lq = affq_from(l); bad |= jmadd_q(acc, lq.x, lq.y); } }
    if (bad) {
        const G1A p = *opaque(hp);
        G1A q2 = p; q2.x = fmul(p.x, fp_const(RIPP_GLV_BETA));
        G1J a2 = jac_inf<Fp>();
#pragma unroll 1
        for (int pos = dg.len - 1; pos >= 0; --pos) {
            a2 = dbl(a2);
            const int d1 = dg.d1[pos], d2 = dg.d2[pos];
            if (d1 != 0) { G1A t = p; if (d1 < 0) t.y = neg(t.y); a2 = add_mixed(a2, t); }
            if (d2 != 0) { G1A t = q2; if (d2 < 0) t.y = neg(t.y); a2 = add_mixed(a2, t); }
        }
        out[i] = add_mixed(a2, *opaque(lp));
    } else out[i] = jacq_to_g1j(acc);
#endif
}

// the table fold of round 0 (kernels.hpp k_fold_g1_tab): tab[e][i], e = M b + m: (2m + 1) * (base b of element i); four wNAF strings
// (tstride: elements per table row -- `half` for a table of a_r alone, 3 q for the table over A1 | A2 | A3 of the fused fold, passed with tab + q)
__global__ void __launch_bounds__(256, 2) k_fold_g1_tab_q(const G1A* __restrict__ tab, size_t tstride, int M, const G1A* __restrict__ lo, uint32_t half, Wnaf4 dg, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
#if defined(__HIP_DEVICE_COMPILE__)
    JacQ acc; jq_set_identity(acc);
    bool inf = true, bad = false;                               // inf is wave-uniform (shared digit strings); a table point at infinity makes the lane `bad`
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl_q(acc);
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const int d = dg.d[t][pos];
            if (d == 0) continue;
            const G1A qa = tab[(size_t)(M * t + ((d < 0 ? -d : d) >> 1)) * tstride + i];
            bad |= is_inf(qa);
            if (inf) { AffQ q = affq_from(qa); if (d < 0) q.y = fq_reduce(fq_neg(q.y)); jq_set(acc, q.x, q.y, fq_one()); inf = false; }
            else bad |= jmadd_q(acc, fq_tab(qa.x), fq_tab_y(qa.y, d < 0));
        }
    }
    const G1A l = lo[i];
    if (inf) { out[i] = to_jac(l); return; }
    if (!is_inf(l)) { const AffQ lq = affq_from(l); bad |= jmadd_q(acc, lq.x, lq.y); }
    if (bad) {
        G1J a2 = jac_inf<Fp>();
#pragma unroll 1
        for (int pos = dg.len - 1; pos >= 0; --pos) {
            a2 = dbl(a2);
#pragma unroll 1
            for (int t = 0; t < 4; ++t) { const int d = dg.d[t][pos]; if (d != 0) { G1A q = tab[(size_t)(M * t + ((d < 0 ? -d : d) >> 1)) * tstride + i]; if (d < 0) q.y = neg(q.y); a2 = add_mixed(a2, q); } }
        }
        out[i] = add_mixed(a2, l);
    } else out[i] = jacq_to_g1j(acc);
#endif
}

// ---- the FUSED fold of rounds 0 and 1 (engine.hip job_fold_fused).  With both challenges known at once -- the look-ahead delivers round 1's two
// values in the hash window -- the quarter-length vector of round 2 is, per element i < q (A0 .. A3 the quarters of the round-0 vector),
//     a''_i = A0_i + x0 A2_i + x1 A1_i + (x0 x1) A3_i,        x0 x1 = k1 + k2 lambda (GLV: both halves < 2^128, phi(P) = (beta x, y)),
// i.e. SIXTEEN 32-bit strings over the same kind of table the round-0 fold uses (bases 2^(32 b) P, odd multiples), now built over A1 | A2 | A3:
// 33 doublings + ~16 x 6.4 additions per output where the two folds take 2 x (33 + ~26) for round 0 and 128 + ~43 for round 1.
// String 4 u + b = 32-bit word b of scalar u:  u = 0: x0 on A2 (table element q + i)   1: k1 on A3 (2 q + i)   2: k2 on phi(A3)   3: x1 on A1 (i).
struct WnafG1x4 { int8_t d[16][36]; int len; };
__device__ __noinline__ inline G1J fold_g1_fused_complete(const G1A* __restrict__ tab, size_t tstride, int M, const G1A& l, uint32_t q, uint32_t i, const WnafG1x4& dg) {
    G1J a2 = jac_inf<Fp>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        a2 = dbl(a2);
#pragma unroll 1
        for (int t = 0; t < 16; ++t) {
            const int d = dg.d[t][pos]; if (d == 0) continue;
            const int u = t >> 2; const uint32_t el = (u == 0 ? q : u == 3 ? 0u : 2 * q) + i;
            G1A p = tab[(size_t)(M * (t & 3) + ((d < 0 ? -d : d) >> 1)) * tstride + el];
            if (u == 2) p.x = fmul(p.x, fp_const(RIPP_GLV_BETA));
            if (d < 0) p.y = neg(p.y);
            a2 = add_mixed(a2, p);
        }
    }
    return add_mixed(a2, l);
}
__global__ void __launch_bounds__(256, 2) k_fold_g1_fused_q(const G1A* __restrict__ tab, size_t tstride, int M, const G1A* __restrict__ lo, uint32_t q, WnafG1x4 dg, G1J* __restrict__ out, uint8_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr uint32_t betaw[12] = RIPP_GLV_BETA; constexpr fq28::Limbs BETA = fq28::from_mont384(betaw);
    JacQ acc; jq_set_identity(acc);
    bool inf = true, bad = false;                               // inf is wave-uniform (shared digit strings); a table point at infinity makes the lane `bad`
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl_q(acc);
#pragma unroll 1
        for (int t = 0; t < 16; ++t) {
            const int d = dg.d[t][pos];
            if (d == 0) continue;
            const int u = t >> 2;
            uint32_t el = i; asm volatile("" : "+v"(el));
            el += u == 0 ? q : u == 3 ? 0u : 2 * q;
            const G1A qa = tab[(size_t)(M * (t & 3) + ((d < 0 ? -d : d) >> 1)) * tstride + el];
            bad |= is_inf(qa);
            if (inf) {
                AffQ p = affq_from(qa);
                if (u == 2) p.x = fq_mul(p.x, fq_const<FQ_LN, 1>(BETA));
                if (d < 0) p.y = fq_reduce(fq_neg(p.y));
                jq_set(acc, p.x, p.y, fq_one()); inf = false;
            } else {
                FqTab px = fq_tab(qa.x);
                if (u == 2) px = fq_widen<FQ_LN, 256>(fq_mul(px, fq_const<FQ_LN, 1>(BETA)));
                bad |= jmadd_q(acc, px, fq_tab_y(qa.y, d < 0));
            }
        }
    }
    const G1A l = lo[i];
    if (inf) { out[i] = to_jac(l); flag[i] = 0; return; }
    if (!is_inf(l)) { const AffQ lq = affq_from(l); bad |= jmadd_q(acc, lq.x, lq.y); }
    flag[i] = bad;
    if (!bad) out[i] = jacq_to_g1j(acc);
#endif
}
__global__ void __launch_bounds__(64) k_fold_g1_fused_fix(const G1A* __restrict__ tab, size_t tstride, int M, const G1A* __restrict__ lo, uint32_t q, WnafG1x4 dg, G1J* __restrict__ out, const uint8_t* __restrict__ flag) {
    for_flagged(flag, q, [&](uint32_t i) { out[i] = fold_g1_fused_complete(tab, tstride, M, lo[i], q, i, dg); });
}

// 2^k * in[i] (Jacobian out): the carry-free twin of kernels.hpp's k_pow2_mul<Fp> -- the pre-doubled bases of the round-0 fold tables (engine.hip:
// job_precompute_round0, three launches of 32 doublings over three quarters of the vector at n = 2^20: 10 ms of the hash window on the 12 x 32-bit form).
// A doubling has no exceptional case on a group of odd order; the identity stays the identity (Z = 0).
__global__ void __launch_bounds__(256, 2) k_pow2_mul_g1_q(const G1A* __restrict__ in, uint32_t n, int k, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const G1A p = in[i];
    if (is_inf(p)) { out[i] = jac_inf<Fp>(); return; }
    const AffQ q = affq_from(p);
    JacQ acc; jq_set(acc, q.x, q.y, fq_one());
#pragma unroll 1
    for (int t = 0; t < k; ++t) jdbl_q(acc);
    out[i] = jacq_to_g1j(acc);
#endif
}

}  // namespace ripp
