// G2 group law and the table fold on the carry-free field form: the throughput twin of kernels.hpp's k_fold_g2_tab (dbl-2009-l / madd-2007-bl in
// the same low-liveness order, Y1 parked in LDS while it is idle).  An Fp2 product is two lazily reduced sums of two products
// (a0 b0 + (K - a1) b1 and a0 b1 + a1 b0: 4 x 196 limb products + 2 reductions, no Karatsuba additions), a square is (a0 + a1)(a0 - a1) and
// (2 a0) a1 on lazy operands.  Every bound is checked by the compiler (fq28.hpp).
#pragma once
#include "fq_curve.hpp"

namespace ripp {

#if defined(__HIP_DEVICE_COMPILE__)
template <uint64_t LM = FQ_LN, int VB = 2> struct Fq2T { Fq<LM, VB> c0, c1; };
using Fq2n = Fq2T<FQ_LN, 2>;          // a product
using Fq2C = Fq2T<FQ_LN, 4>;          // a coordinate between two group operations

template <uint64_t L1, int V1, uint64_t L2, int V2> __device__ __forceinline__ auto f2_add(const Fq2T<L1, V1>& a, const Fq2T<L2, V2>& b) { return Fq2T<L1 + L2 - 1, V1 + V2>{fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
template <uint64_t L1, int V1, uint64_t L2, int V2> __device__ __forceinline__ auto f2_sub(const Fq2T<L1, V1>& a, const Fq2T<L2, V2>& b) { return Fq2T<fq28::sub_lm(L1, L2), V1 + V2 + 1>{fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ auto f2_dbl(const Fq2T<L1, V1>& a) { return Fq2T<2 * L1 - 1, 2 * V1>{fq_dbl(a.c0), fq_dbl(a.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ auto f2_norm(const Fq2T<L1, V1>& a) { return Fq2T<FQ_LN, V1>{fq_norm(a.c0), fq_norm(a.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ Fq2n f2_reduce(const Fq2T<L1, V1>& a) { return {fq_reduce(a.c0), fq_reduce(a.c1)}; }
template <uint64_t L1, int V1> __device__ __forceinline__ Fq2C f2_coord(const Fq2T<L1, V1>& a) { return {fq_coord(a.c0), fq_coord(a.c1)}; }
__device__ __forceinline__ bool f2_is_zero(const Fq2n& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
// THE call boundary of the G2 kernels: one out-of-line Fq product with vector-typed register arguments (28 VGPRs in, 14 out -- within the
// 32 argument registers of the AMDGPU convention, nothing on the stack).  Inlining the ~550-instruction product 20 times per group
// operation lets the scheduler interleave them and spills ~340 dwords; behind a call every product is a liveness barrier.
typedef uint32_t fq_v4u __attribute__((ext_vector_type(4)));
typedef uint32_t fq_v2u __attribute__((ext_vector_type(2)));
struct FqRegs { fq_v4u v0, v1, v2; fq_v2u v3; };
using FqW = Fq<((uint64_t)1 << 30), 50>;              // what the out-of-line product is compiled for: limbs < 2^30, V1 * V2 <= 2500 (checked by the wrapper)
__device__ __noinline__ inline FqRegs fq_mul_call(fq_v4u a0, fq_v4u a1, fq_v4u a2, fq_v2u a3, fq_v4u b0, fq_v4u b1, fq_v4u b2, fq_v2u b3) {
    FqW a, b;
    a.l[0] = a0.x; a.l[1] = a0.y; a.l[2] = a0.z; a.l[3] = a0.w; a.l[4] = a1.x; a.l[5] = a1.y; a.l[6] = a1.z; a.l[7] = a1.w; a.l[8] = a2.x; a.l[9] = a2.y; a.l[10] = a2.z; a.l[11] = a2.w; a.l[12] = a3.x; a.l[13] = a3.y;
    b.l[0] = b0.x; b.l[1] = b0.y; b.l[2] = b0.z; b.l[3] = b0.w; b.l[4] = b1.x; b.l[5] = b1.y; b.l[6] = b1.z; b.l[7] = b1.w; b.l[8] = b2.x; b.l[9] = b2.y; b.l[10] = b2.z; b.l[11] = b2.w; b.l[12] = b3.x; b.l[13] = b3.y;
    const Fqn r = fq_mul(a, b);
    FqRegs o;
    o.v0 = fq_v4u{r.l[0], r.l[1], r.l[2], r.l[3]}; o.v1 = fq_v4u{r.l[4], r.l[5], r.l[6], r.l[7]}; o.v2 = fq_v4u{r.l[8], r.l[9], r.l[10], r.l[11]}; o.v3 = fq_v2u{r.l[12], r.l[13]};
    return o;
}
template <uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ Fqn fq_mulc(const Fq<L1, V1>& a, const Fq<L2, V2>& b) {
    static_assert(L1 <= ((uint64_t)1 << 30) && L2 <= ((uint64_t)1 << 30), "normalise the operand first");
    static_assert((long)V1 * V2 <= fq28::VMAX, "value bound");
    const FqRegs r = fq_mul_call(fq_v4u{a.l[0], a.l[1], a.l[2], a.l[3]}, fq_v4u{a.l[4], a.l[5], a.l[6], a.l[7]}, fq_v4u{a.l[8], a.l[9], a.l[10], a.l[11]}, fq_v2u{a.l[12], a.l[13]},
                                 fq_v4u{b.l[0], b.l[1], b.l[2], b.l[3]}, fq_v4u{b.l[4], b.l[5], b.l[6], b.l[7]}, fq_v4u{b.l[8], b.l[9], b.l[10], b.l[11]}, fq_v2u{b.l[12], b.l[13]});
    Fqn o;
    o.l[0] = r.v0.x; o.l[1] = r.v0.y; o.l[2] = r.v0.z; o.l[3] = r.v0.w; o.l[4] = r.v1.x; o.l[5] = r.v1.y; o.l[6] = r.v1.z; o.l[7] = r.v1.w; o.l[8] = r.v2.x; o.l[9] = r.v2.y; o.l[10] = r.v2.z; o.l[11] = r.v2.w; o.l[12] = r.v3.x; o.l[13] = r.v3.y;
    return o;
}
// operands of a product are normalised (one carry pass per coefficient) when their limbs could exceed LIM
template <uint64_t LIM, uint64_t L1, int V1> __device__ __forceinline__ auto f2_n(const Fq2T<L1, V1>& a) { if constexpr (L1 <= LIM) return a; else return f2_norm(a); }
// (a0 + a1 u)(b0 + b1 u), u^2 = -1: Karatsuba on three out-of-line products; the three additions and five subtractions are limb-wise.
// The result is LAZY (c0 < 5p, c1 < 8p, limbs < 5 * 2^28).
template <uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ auto f2_mul(const Fq2T<L1, V1>& a_, const Fq2T<L2, V2>& b_) {
    const auto a = f2_n<((uint64_t)1 << 29)>(a_); const auto b = f2_n<((uint64_t)1 << 29)>(b_);
    const Fqn t0 = fq_mulc(a.c0, b.c0), t1 = fq_mulc(a.c1, b.c1);
    const Fqn m = fq_mulc(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1));
    const auto c0 = fq_sub(t0, t1); const auto c1 = fq_sub(fq_sub(m, t0), t1);
    constexpr uint64_t LO = fq28::sub_lm(fq28::sub_lm(FQ_LN, FQ_LN), FQ_LN);
    return Fq2T<LO, 8>{fq_widen<LO, 8>(c0), fq_widen<LO, 8>(c1)};
}
// (a0 + a1)(a0 - a1), (2 a0) a1: two products; the result is reduced
template <uint64_t L1, int V1>
__device__ __forceinline__ Fq2n f2_sqr(const Fq2T<L1, V1>& a_) {
    const auto a = f2_n<FQ_LN>(a_);
    return {fq_mulc(fq_add(a.c0, a.c1), fq_sub(a.c0, a.c1)), fq_mulc(fq_dbl(a.c0), a.c1)};
}

// engine Fp2 (Mont-384) -> carry-free form: times 2^8 is a re-slicing, then one quotient estimate (no Montgomery product)
__device__ __forceinline__ Fqn fq_from_fp_fast(const Fp& x) { return fq_reduce(fq_unpack_shl8(x.l)); }
__device__ __forceinline__ Fq2n f2_from(const Fp2& x) { return {fq_from_fp_fast(x.c0), fq_from_fp_fast(x.c1)}; }
__device__ __forceinline__ Fp2 f2_to(const Fq2C& a) { return {fq_to_fp(fq_reduce(a.c0)), fq_to_fp(fq_reduce(a.c1))}; }

struct JacQ2 { Fq2C x, y, z; };
#define SBQ() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void jdbl2_q(JacQ2& p) {            // dbl-2009-l in the order of kernels.hpp::jdbl_lo
    p.z = f2_coord(f2_dbl(f2_mul(p.y, p.z))); SBQ();
    const Fq2n A = f2_sqr(p.x); SBQ();
    const Fq2n B = f2_sqr(p.y); SBQ();
    const Fq2n t = f2_sqr(f2_add(p.x, B)); SBQ();
    const Fq2n C = f2_sqr(B); SBQ();
    const auto D = f2_norm(f2_dbl(f2_sub(f2_sub(t, A), C))); SBQ();
    const auto E = f2_add(f2_dbl(A), A); SBQ();
    const Fq2n X3 = f2_reduce(f2_sub(f2_sub(f2_sqr(E), D), D)); SBQ();
    p.y = f2_coord(f2_sub(f2_mul(E, f2_sub(D, X3)), f2_dbl(f2_dbl(f2_dbl(C)))));
    p.x = f2_coord(X3);
}
// madd-2007-bl; loadx / loady fetch the affine addend (Fq2n); park: this lane's LDS column (stride 64 lanes, 7 x 16 B).
// Returns true when the result is NOT valid (H = 0: T = +-Q).
template <class LOADX, class LOADY>
__device__ __forceinline__ bool jmadd2_q(JacQ2& p, LOADX loadx, LOADY loady, uint4* park) {
    const Fq2n Z1Z1 = f2_sqr(p.z); SBQ();
    const Fq2n H = f2_reduce(f2_sub(f2_mul(loadx(), Z1Z1), p.x)); SBQ();
    const auto t = f2_mul(p.z, Z1Z1); SBQ();
    const Fq2n r = f2_reduce(f2_dbl(f2_sub(f2_mul(loady(), t), p.y))); SBQ();
    const bool special = f2_is_zero(H);
    { uint32_t w[28];                                                      // Y1 rests in LDS until the last product
#pragma unroll
      for (int k = 0; k < 14; ++k) { w[k] = p.y.c0.l[k]; w[14 + k] = p.y.c1.l[k]; }
      const uint4* src = reinterpret_cast<const uint4*>(w);
#pragma unroll
      for (int k = 0; k < 7; ++k) park[k * 64] = src[k]; } SBQ();
    const Fq2n HH = f2_sqr(H); SBQ();
    p.z = f2_coord(f2_sub(f2_sub(f2_sqr(f2_add(p.z, H)), Z1Z1), HH)); SBQ();
    const auto I = f2_dbl(f2_dbl(HH));
    const auto J = f2_norm(f2_mul(H, I)); SBQ();
    const auto V = f2_norm(f2_mul(p.x, I)); SBQ();
    const Fq2n X3 = f2_reduce(f2_sub(f2_sub(f2_sub(f2_sqr(r), J), V), V)); SBQ();
    Fq2n t2;
    { uint4 q[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) q[k] = park[k * 64];
      const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
      Fq2C y1;
#pragma unroll
      for (int k = 0; k < 14; ++k) { y1.c0.l[k] = w[k]; y1.c1.l[k] = w[14 + k]; }
      t2 = f2_reduce(f2_mul(y1, J)); } SBQ();
    p.y = f2_coord(f2_sub(f2_mul(r, f2_sub(V, X3)), f2_dbl(t2)));
    p.x = f2_coord(X3);
    return special;
}
#undef SBQ
#endif

// NS digit strings; string t works on table rows t M .. t M + M - 1 (kernels.hpp k_fold_g2_tab).  Exceptional lanes are recomputed with the
// complete formulas (fold_g2_tab_complete).
template <class D, int NS>
__global__ void __launch_bounds__(64, 2) k_fold_g2_tab_q(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t half, D dg, G2J* __restrict__ out) {
    __shared__ uint4 park_[7 * 64];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* park = park_ + threadIdx.x;
    JacQ2 acc; acc.x = acc.y = f2_coord(Fq2n{fq_one(), fq_zero()}); acc.z = f2_coord(Fq2n{fq_zero(), fq_zero()});
    bool inf = true, bad = false;                                        // inf is wave-uniform: the digit strings are shared by the launch
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl2_q(acc);
#pragma unroll 1
        for (int t = 0; t < NS; ++t) {
            const int d = dg.d[t][pos];
            if (d == 0) continue;
            const uint4* base = qtab + ((size_t)t * M + ((d < 0 ? -d : d) >> 1)) * G2A_CHUNKS * stride + i;
            auto ldfp2 = [&](int q0) { Fp2 v; uint4* dd = reinterpret_cast<uint4*>(&v);
#pragma unroll
                for (int q = 0; q < 6; ++q) dd[q] = base[(size_t)(q0 + q) * stride]; return v; };
            auto loadx = [&]() { return f2_from(ldfp2(0)); };
            auto loady = [&]() { const Fp2 y = ldfp2(6); return f2_from(d < 0 ? neg(y) : y); };
            if (inf) {                                                    // first addition: acc <- +-Q
                const Fp2 x0 = ldfp2(0), y0 = ldfp2(6);
                bad |= x0.is_zero() && y0.is_zero();
                acc.x = f2_coord(f2_from(x0)); acc.y = f2_coord(f2_from(d < 0 ? neg(y0) : y0)); acc.z = f2_coord(Fq2n{fq_one(), fq_zero()}); inf = false;
            } else {
                { const Fp2 x0 = ldfp2(0); bad |= x0.is_zero() && ldfp2(6).is_zero(); }      // a table point at infinity
                bad |= jmadd2_q(acc, loadx, loady, park);
            }
        }
    }
    const G2A l = lo[i];
    if (inf) { out[i] = to_jac(l); return; }
    if (!is_inf(l)) bad |= jmadd2_q(acc, [&]() { return f2_from(l.x); }, [&]() { return f2_from(l.y); }, park);
    if (bad) fold_g2_tab_complete<D, NS>(qtab, stride, M, lo, i, dg, out);
    else out[i] = G2J{f2_to(acc.x), f2_to(acc.y), f2_to(acc.z)};
#endif
}

}  // namespace ripp
