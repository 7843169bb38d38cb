// compile-only probe: variants of the Miller-line kernel
#include "../../ripp_amd/csrc/kernels.hpp"
namespace ripp {
#define SB() __builtin_amdgcn_sched_barrier(0)
template <class T> __device__ __forceinline__ const T* opaque(const T* p);
// low-liveness order; lines are stored as soon as they are complete
__device__ __forceinline__ void line_double_store(Fp2& X, Fp2& Y, Fp2& Z, const G1A* p, uint4* lines, size_t s, size_t stride, size_t i, bool skip) {
    const Fp2 t1 = sqr(add(Y, Z)); SB();
    const Fp2 c = sqr(Z); SB();
    const Fp2 b = sqr(Y); SB();
    const Fp2 h = sub(t1, add(b, c)); SB();
    {
        const Fp yP = opaque(p)->y; const Fp2 l2 = mul_fp(neg(h), yP);
        store_chunks<6>(lines, s * 3 + 2, stride, i, skip ? Fp2::zero() : l2);
    } SB();
    const Fp2 e = mul_by_b_twist(add(dbl(c), c)); SB();
    store_chunks<6>(lines, s * 3 + 0, stride, i, skip ? Fp2::one() : sub(e, b)); SB();
    const Fp2 a = half(mul(X, Y)); SB();
    {
        const Fp2 j = sqr(X);
        const Fp xP = opaque(p)->x; const Fp2 l1 = mul_fp(add(dbl(j), j), xP);
        store_chunks<6>(lines, s * 3 + 1, stride, i, skip ? Fp2::zero() : l1);
    } SB();
    Z = mul(b, h); SB();
    const Fp2 f = add(dbl(e), e);
    X = mul(a, sub(b, f)); SB();
    const Fp2 g = half(add(b, f));
    const Fp2 e2 = sqr(e); SB();
    Y = sub(sqr(g), add(dbl(e2), e2));
}

template <class T> __device__ __forceinline__ const T* opaque(const T* p) { asm volatile("" : "+v"(p)); return p; }   // defeats hoisting: operands are RE-LOADED where used
__device__ __forceinline__ void line_add_store(Fp2& X, Fp2& Y, Fp2& Z, const G2A* q, const G1A* p, uint4* lines, size_t s, size_t stride, size_t i, bool skip, uint4* park) {
    Fp2 theta, lambda;
    { const Fp2 qy = opaque(q)->y; theta = sub(Y, mul(qy, Z)); } SB();
    { const uint4* src = reinterpret_cast<const uint4*>(&Y);          // Y is not needed again until the last product: park it in LDS
#pragma unroll
      for (int k = 0; k < 6; ++k) park[k * 256] = src[k]; } SB();
    { const Fp2 qx = opaque(q)->x; lambda = sub(X, mul(qx, Z)); } SB();
    { const Fp2 qx = opaque(q)->x; const Fp2 t = mul(theta, qx); SB(); const Fp2 qy = opaque(q)->y; const Fp2 j = sub(t, mul(lambda, qy));
      store_chunks<6>(lines, s * 3 + 0, stride, i, skip ? Fp2::one() : j); } SB();
    { const Fp xP = opaque(p)->x; store_chunks<6>(lines, s * 3 + 1, stride, i, skip ? Fp2::zero() : mul_fp(neg(theta), xP)); } SB();
    { const Fp yP = opaque(p)->y; store_chunks<6>(lines, s * 3 + 2, stride, i, skip ? Fp2::zero() : mul_fp(lambda, yP)); } SB();
    Fp2 f;
    { const Fp2 c = sqr(theta); SB(); f = mul(Z, c); } SB();
    { const uint4* src = reinterpret_cast<const uint4*>(&theta);      // theta rests in LDS until the last-but-one product
#pragma unroll
      for (int k = 0; k < 6; ++k) park[(6 + k) * 256] = src[k]; } SB();
    Fp2 e, g;
    { const Fp2 d = sqr(lambda); SB(); e = mul(lambda, d); SB(); g = mul(X, d); } SB();
    const Fp2 h = sub(add(e, f), dbl(g)); SB();
    X = mul(lambda, h); SB();
    Z = mul(Z, e); SB();
    Fp2 t;
    { Fp2 th; uint4* dst = reinterpret_cast<uint4*>(&th);
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = park[(6 + k) * 256];
      t = mul(th, sub(g, h)); } SB();
    { Fp2 y0; uint4* dst = reinterpret_cast<uint4*>(&y0);
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = park[k * 256];
      Y = sub(t, mul(e, y0)); }
}
__global__ void __launch_bounds__(256, RIPP_OCC) k_miller_lines_v2(PairSets ps, uint32_t M, uint4* __restrict__ lines, size_t stride) {
    __shared__ uint4 park[12 * 256];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const G1A* __restrict__ a = ps.a[blockIdx.y];
    const G2A* __restrict__ b = ps.b[blockIdx.y];
    Fp2 X, Y, Z = Fp2::one();
    { const G2A Q = b[i]; X = Q.x; Y = Q.y; }
    bool skip; { const G1A P = a[i]; skip = is_inf(P) || (X.is_zero() && Y.is_zero()); }
    const LineCoeffs unit = {Fp2::one(), Fp2::zero(), Fp2::zero()};
    size_t s = (size_t)blockIdx.y * N_LINES;
#pragma unroll 1
    for (int bit = 62; bit >= 0; --bit) {
        line_double_store(X, Y, Z, a + i, lines, s, stride, i, skip);
        ++s;
        if ((BLS_X_ABS >> bit) & 1ull) {
            line_add_store(X, Y, Z, b + i, a + i, lines, s, stride, i, skip, park + threadIdx.x);
            ++s;
        }
    }
}
}
