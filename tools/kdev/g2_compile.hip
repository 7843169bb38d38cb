// compile-only probe: low-register G2 doubling / mixed addition for the fold kernels
#include "../../ripp_amd/csrc/kernels.hpp"
namespace ripp {
#define SB() __builtin_amdgcn_sched_barrier(0)
// dbl-2009-l in low-liveness order (at most 5 Fp2 live)
__device__ __forceinline__ void jdbl_lo(Fp2& X, Fp2& Y, Fp2& Z) {
    Z = dbl(fmul(Y, Z)); SB();
    const Fp2 A = fsqr(X); SB();
    Fp2 B = fsqr(Y); SB();
    Fp2 t = fsqr(add(X, B)); SB();
    const Fp2 C = fsqr(B); SB();
    const Fp2 D = dbl(sub(sub(t, A), C)); SB();
    const Fp2 E = add(dbl(A), A); SB();
    X = sub(sub(fsqr(E), D), D); SB();
    Y = sub(fmul(E, sub(D, X)), dbl(dbl(dbl(C))));
}
// madd-2007-bl with the table point read from chunked memory where it is used; special cases flagged, not handled (caller falls back)
template <class LOADX, class LOADY>
__device__ __forceinline__ bool jmadd_lo(Fp2& X, Fp2& Y, Fp2& Z, LOADX loadx, LOADY loady, bool negy, uint4* park) {
    const Fp2 Z1Z1 = fsqr(Z); SB();
    Fp2 H; { const Fp2 x2 = loadx(); H = sub(fmul(x2, Z1Z1), X); } SB();
    Fp2 r; { Fp2 y2 = loady(); if (negy) y2 = neg(y2); const Fp2 t = fmul(Z, Z1Z1); SB(); r = sub(fmul(y2, t), Y); } SB();
    const bool special = H.is_zero();
    { const uint4* src = reinterpret_cast<const uint4*>(&Y);              // Y rests in LDS until the last product
#pragma unroll
      for (int k = 0; k < 6; ++k) park[k * 64] = src[k]; } SB();
    r = dbl(r);
    const Fp2 HH = fsqr(H); SB();
    Z = sub(sub(fsqr(add(Z, H)), Z1Z1), HH); SB();
    const Fp2 I = dbl(dbl(HH)); 
    const Fp2 J = fmul(H, I); SB();
    const Fp2 V = fmul(X, I); SB();
    X = sub(sub(sub(fsqr(r), J), V), V); SB();
    Fp2 t2; { Fp2 y1; uint4* dst = reinterpret_cast<uint4*>(&y1);
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = park[k * 64];
      t2 = fmul(y1, J); } SB();
    Y = sub(fmul(r, sub(V, X)), dbl(t2));
    return special;
}
template <class D, int NS>
__global__ void __launch_bounds__(64, RIPP_OCC) k_fold_g2_tab_v2(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t half, D dg, G2J* __restrict__ out, uint32_t* __restrict__ flag) {
    __shared__ uint4 park[6 * 64];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    Fp2 X = Fp2::one(), Y = Fp2::one(), Z = Fp2::zero();
    bool inf = true, bad = false;
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl_lo(X, Y, Z);
#pragma unroll 1
        for (int t = 0; t < NS; ++t) {
            const int d = dg.d[t][pos];
            if (d != 0) {
                const size_t row = (size_t)t * M + ((d < 0 ? -d : d) >> 1);
                if (inf) {   // first addition: acc <- +-Q  (uniform: the digit strings are shared by the launch)
                    G2A q = load_chunks<G2A_CHUNKS, G2A>(qtab, row, stride, i);
                    if (d < 0) q.y = neg(q.y);
                    X = q.x; Y = q.y; Z = Fp2::one(); inf = false;
                    if (is_inf(q)) bad = true;
                } else {
                    const uint4* base = qtab + row * G2A_CHUNKS * stride + i;
                    bad |= jmadd_lo(X, Y, Z,
                        [&]() { Fp2 v; uint4* dd = reinterpret_cast<uint4*>(&v);
#pragma unroll
                                for (int q = 0; q < 6; ++q) dd[q] = base[(size_t)q * stride]; return v; },
                        [&]() { Fp2 v; uint4* dd = reinterpret_cast<uint4*>(&v);
#pragma unroll
                                for (int q = 0; q < 6; ++q) dd[q] = base[(size_t)(6 + q) * stride]; return v; }, d < 0, park + threadIdx.x);
                }
            }
        }
    }
    {
        const G2A* lp = lo + i;
        if (inf) { const G2A q = *lp; X = q.x; Y = q.y; Z = is_inf(q) ? Fp2::zero() : Fp2::one(); }
        else {
            bool linf; { const G2A q = *lp; linf = is_inf(q); }
            if (!linf) bad |= jmadd_lo(X, Y, Z, [&]() { return opaque(lp)->x; }, [&]() { return opaque(lp)->y; }, false, park + threadIdx.x);
        }
    }
    if (bad) atomicOr(flag, 1u);
    out[i] = G2J{X, Y, Z};
}
template __global__ void k_fold_g2_tab_v2<GlsDigits, 4>(const uint4*, size_t, int, const G2A*, uint32_t, GlsDigits, G2J*, uint32_t*);
}
