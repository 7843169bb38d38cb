// HIP kernels of the inner-pairing-product engine (gfx950 / MI355X).  Included once, by engine.hip.
//
// Work decomposition (see bls12_381/pairing.hpp for the algebra):
//   k_miller_lines    one LANE per (P,Q) pair; 68 sequential steps; writes the sparse line element of every step
//   k_line_products   grid.y = step-rows; each lane multiplies a strided subset of one row's lines (mul_by_014)
//   k_fp12_tree       dense Fp12 product tree over the per-lane partials
//   k_scale / k_fold  per-lane scalar multiplication (+ add) for the SIPP scaling and halving-round folds
//   k_normalize       Jacobian -> affine with one inversion per lane-batch (Montgomery's trick)
// No MFMA anywhere: this is 381-bit modular integer arithmetic on the VALU (v_mad_u64_u32).
//
// HBM layout of the line / partial buffers: 16-byte chunks, chunk-major then lane:  buf[(row*Q + q)*stride + i]
// (Q = 18 chunks per sparse line, 36 per dense Fp12), so that the 64 lanes of a wave read/write 1 KiB contiguous
// per instruction (global_load/store_dwordx4, fully coalesced).  Points are read as AoS structs (96/192 B).
#pragma once
#include <hip/hip_runtime.h>
#include "bls12_381/pairing.hpp"

namespace ripp {

#ifndef RIPP_OCC_PROD
#define RIPP_OCC_PROD 2
#endif
#ifndef RIPP_OCC
#define RIPP_OCC 2      // min waves per SIMD requested for the register-heavy kernels (caps them at 256 VGPR+AGPR)
#endif
constexpr int LINE_CHUNKS = 18;    // 3 Fp2 = 72 dwords
constexpr int FP12_CHUNKS = 36;    // 144 dwords

struct ScalarBits { uint32_t w[8]; int nbits; };   // canonical little-endian scalar shared by all lanes of a launch

// ---- chunked SoA accessors -------------------------------------------------------------------------------
template <int NCH, class T>
__device__ __forceinline__ void store_chunks(uint4* buf, size_t row, size_t stride, size_t i, const T& v) {
    static_assert(sizeof(T) == NCH * 16, "size");
    const uint4* src = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (int q = 0; q < NCH; ++q) buf[(row * NCH + q) * stride + i] = src[q];
}
template <int NCH, class T>
__device__ __forceinline__ T load_chunks(const uint4* buf, size_t row, size_t stride, size_t i) {
    static_assert(sizeof(T) == NCH * 16, "size");
    T v;
    uint4* dst = reinterpret_cast<uint4*>(&v);
#pragma unroll
    for (int q = 0; q < NCH; ++q) dst[q] = buf[(row * NCH + q) * stride + i];
    return v;
}

// ---- stage 1: line elements --------------------------------------------------------------------------------
// grid.y = product index p: pairs (a[p][i], b[p][i]), i < M, rows p*68 .. p*68+67 of lines[rows][18][stride].
// Pairs with a point at infinity emit the unit line.  Both pairing products of a SIPP round go in ONE launch.
constexpr int MAX_PRODUCTS = 64;    // pairing products sharing one stage-1 launch (2 per SIPP round, 6 per GIPA/TIPP round, 8 quarter products of a pipelined SIPP tail round;
                                    // up to the 64 block products of a round-3 look-ahead item: build round 5 -- eight launches of 8 left stage 1 at a lone wave's latency)
struct PairSets { const G1A* a[MAX_PRODUCTS]; const G2A* b[MAX_PRODUCTS]; };
// The same products grouped into CHAINS (fq_miller.hpp k_miller_lines_q): chain g walks b[g] once and emits the lines of the np[g] consecutive
// products first[g] .. first[g] + np[g] - 1, whose P vectors are a[first[g]] ..  (a chain of one product is the plain form).
constexpr int MAX_SHARE = 4;
struct ChainSets { const G1A* a[MAX_PRODUCTS]; const G2A* b[MAX_PRODUCTS]; uint8_t first[MAX_PRODUCTS], np[MAX_PRODUCTS]; };
// Register discipline of this kernel (it used to spill 273 dwords, ~35 GB of scratch traffic per launch): the steps below are written in
// a LOW-LIVENESS order pinned with scheduling barriers -- every line coefficient is stored the moment it is complete, P and Q are
// re-loaded from L2 where they are used instead of being held for 68 steps, and the addition step (5 of 68) parks Y and theta in LDS
// while they are idle.  The doubling step then needs 249 registers and no scratch at all.
// Slots of the stored sparse line (c[0], c[1], c[2]) and where they sit in Fp12 (line_products.hpp):
//   M-type twist (BLS12-381, ark-ec `ell` -> mul_by_014): (free, xP-scaled, yP-scaled) at w^0, w^2, w^3
//   D-type twist (BLS12-377,               -> mul_by_034): (yP-scaled, xP-scaled, free) at w^0, w^1, w^3
// The unit line (pairs with a point at infinity) is 1 at w^0 in both.
#if defined(RIPP_BLS12_377)
#define LINE_SLOT_YP 0
#define LINE_SLOT_FREE 2
#define LINE_UNIT_YP Fp2::one()
#define LINE_UNIT_FREE Fp2::zero()
#else
#define LINE_SLOT_YP 2
#define LINE_SLOT_FREE 0
#define LINE_UNIT_YP Fp2::zero()
#define LINE_UNIT_FREE Fp2::one()
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
template <class T> __device__ __forceinline__ const T* opaque(const T* p) { asm volatile("" : "+v"(p)); return p; }   // defeats hoisting: operands are RE-LOADED where used
// The *_fix kernels behind the carry-free throughput kernels: a SMALL fixed grid (FIX_GRID x 64 lanes) walks the flag bytes 16 at a time and redoes the
// flagged lanes with the complete formulas.  (One lane per element, as first written, cost 2.4 ms per 2^19-lane launch with NOTHING flagged: these
// kernels hold 256 + 110 registers and a 400-1 300 B frame per lane, and dispatching 8 192 such waves is not free.)  flag: 16-byte aligned, readable
// up to the next multiple of 16; bytes at or beyond n, and bytes the throughput kernel did not write, are ignored by the caller's own range checks.
constexpr int FIX_GRID = 128;
template <class FN> __device__ __forceinline__ void for_flagged(const uint8_t* __restrict__ flag, uint32_t n, FN fn) {
    const uint32_t nth = gridDim.x * blockDim.x;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; (uint64_t)c * 16 < n; c += nth) {
        const uint4 f = reinterpret_cast<const uint4*>(flag)[c];
        if ((f.x | f.y | f.z | f.w) == 0) continue;
        const uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll 1
        for (int k = 0; k < 16; ++k) { const uint32_t i = c * 16 + (uint32_t)k; if (i < n && ((w[k >> 2] >> (8 * (k & 3))) & 0xFFu)) fn(i); }
    }
}
// low-liveness order; lines are stored as soon as they are complete
__device__ __forceinline__ void line_double_store(Fp2& X, Fp2& Y, Fp2& Z, const G1A* p, uint4* lines, size_t s, size_t stride, size_t i, bool skip) {
    const Fp2 t1 = sqr(add(Y, Z)); SB();
    const Fp2 c = sqr(Z); SB();
    const Fp2 b = sqr(Y); SB();
    const Fp2 h = sub(t1, add(b, c)); SB();
    {
        const Fp yP = opaque(p)->y; const Fp2 l2 = mul_fp(neg(h), yP);
        store_chunks<6>(lines, s * 3 + LINE_SLOT_YP, stride, i, skip ? LINE_UNIT_YP : l2);
    } SB();
    const Fp2 e = mul_by_b_twist(add(dbl(c), c)); SB();
    store_chunks<6>(lines, s * 3 + LINE_SLOT_FREE, stride, i, skip ? LINE_UNIT_FREE : sub(e, b)); SB();
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

__device__ __forceinline__ void line_add_store(Fp2& X, Fp2& Y, Fp2& Z, const G2A* q, const G1A* p, uint4* lines, size_t s, size_t stride, size_t i, bool skip, uint4* park) {
    Fp2 theta, lambda;
    { const Fp2 qy = opaque(q)->y; theta = sub(Y, mul(qy, Z)); } SB();
    { const uint4* src = reinterpret_cast<const uint4*>(&Y);          // Y is not needed again until the last product: park it in LDS
#pragma unroll
      for (int k = 0; k < 6; ++k) park[k * 256] = src[k]; } SB();
    { const Fp2 qx = opaque(q)->x; lambda = sub(X, mul(qx, Z)); } SB();
    { const uint4* src = reinterpret_cast<const uint4*>(&X);          // X idles until g = X lambda^2
#pragma unroll
      for (int k = 0; k < 6; ++k) park[(12 + k) * 256] = src[k]; } SB();
    { const Fp2 qx = opaque(q)->x; const Fp2 t = mul(theta, qx); SB(); const Fp2 qy = opaque(q)->y; const Fp2 j = sub(t, mul(lambda, qy));
      store_chunks<6>(lines, s * 3 + LINE_SLOT_FREE, stride, i, skip ? LINE_UNIT_FREE : j); } SB();
    { const Fp xP = opaque(p)->x; store_chunks<6>(lines, s * 3 + 1, stride, i, skip ? Fp2::zero() : mul_fp(neg(theta), xP)); } SB();
    { const Fp yP = opaque(p)->y; store_chunks<6>(lines, s * 3 + LINE_SLOT_YP, stride, i, skip ? LINE_UNIT_YP : mul_fp(lambda, yP)); } SB();
    Fp2 f;
    { const Fp2 c = sqr(theta); SB(); f = mul(Z, c); } SB();
    { const uint4* src = reinterpret_cast<const uint4*>(&theta);      // theta rests in LDS until the last-but-one product
#pragma unroll
      for (int k = 0; k < 6; ++k) park[(6 + k) * 256] = src[k]; } SB();
    Fp2 e, g;
    { const Fp2 d = sqr(lambda); SB(); e = mul(lambda, d); SB();
      Fp2 x0; uint4* dst = reinterpret_cast<uint4*>(&x0);
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = park[(12 + k) * 256];
      g = mul(x0, d); } SB();
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
__global__ void __launch_bounds__(256, RIPP_OCC) k_miller_lines(PairSets ps, uint32_t M, uint4* __restrict__ lines, size_t stride) {
    __shared__ uint4 park[18 * 256];
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

#undef SB

// ---- stage 2a: sparse accumulation -------------------------------------------------------------------------
// grid = (T / block, rows).  Lane t of row r multiplies lines r[t], r[t+T], ... (< M) and writes one dense partial.
#if defined(RIPP_AB_KERNELS) && !defined(RIPP_BLS12_377)       // A/B reference form (mul_by_014): BLS12-381 builds with -DRIPP_AB_KERNELS only
__global__ void __launch_bounds__(64, RIPP_OCC_PROD) k_line_products1(const uint4* __restrict__ lines, size_t stride, uint32_t M,
                                                        uint4* __restrict__ partials, uint32_t T) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const size_t row = blockIdx.y;
    Fp12 acc = Fp12::one();
    if (t < M) {
        const LineCoeffs l0 = load_chunks<LINE_CHUNKS, LineCoeffs>(lines, row, stride, t);
        acc.c0.c0 = l0.c0; acc.c0.c1 = l0.c1; acc.c1.c1 = l0.c2;       // 1 * line, without the multiplication
#pragma unroll 1
        for (uint32_t i = t + T; i < M; i += T) {
            const LineCoeffs l = load_chunks<LINE_CHUNKS, LineCoeffs>(lines, row, stride, i);
            acc = mul_by_014(acc, l.c0, l.c1, l.c2);
        }
    }
    store_chunks<FP12_CHUNKS>(partials, row, T, t, acc);
}
#endif

// ---- stage 2b: dense product tree --------------------------------------------------------------------------
// in: [rows][36][Tin] -> out: [rows][36][Tout];  out[j] = prod_{k<R} in[j + k*Tout]
__global__ void __launch_bounds__(64, RIPP_OCC) k_fp12_tree(const uint4* __restrict__ in, uint32_t Tin, uint4* __restrict__ out, uint32_t Tout, int R) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Tout) return;
    const size_t row = blockIdx.y;
    Fp12 acc = load_chunks<FP12_CHUNKS, Fp12>(in, row, Tin, j);
#pragma unroll 1
    for (int k = 1; k < R; ++k) {
        const uint32_t idx = j + k * Tout;
        if (idx < Tin) acc = mul(acc, load_chunks<FP12_CHUNKS, Fp12>(in, row, Tin, idx));
    }
    store_chunks<FP12_CHUNKS>(out, row, Tout, j, acc);
}

// ---- scalar multiplication kernels -------------------------------------------------------------------------
// out[i] = r[i] * a[i]  (per-element 255-bit scalars, Montgomery form in memory)   -- sipp/src/lib.rs:61-65
__global__ void __launch_bounds__(256) k_scale_g1(const G1A* __restrict__ a, const Fr* __restrict__ r, uint32_t n, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr k = from_mont(r[i]);
    const G1A p = a[i];
    G1J acc = jac_inf<Fp>();
#pragma unroll 1
    for (int bit = 254; bit >= 0; --bit) {
        acc = dbl(acc);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1u) acc = add_mixed(acc, p);
    }
    out[i] = acc;
}

// out[i] = s * hi[i] + lo[i]  with ONE scalar for the whole launch (uniform control flow)
//   -- sipp/src/lib.rs:87-91, 95-99 (affine operands); ip_proofs/src/gipa.rs:262-290 (projective operands)
template <class F>
__global__ void __launch_bounds__(256) k_fold_affine(const Affine<F>* __restrict__ hi, const Affine<F>* __restrict__ lo, uint32_t half,
                                                      ScalarBits s, Jac<F>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Affine<F> p = hi[i];
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (int bit = s.nbits - 1; bit >= 0; --bit) {
        acc = dbl(acc);
        if ((s.w[bit >> 5] >> (bit & 31)) & 1u) acc = add_mixed(acc, p);
    }
    out[i] = add_mixed(acc, lo[i]);
}
template <class F>
__global__ void __launch_bounds__(256) k_fold_jac(const Jac<F>* __restrict__ hi, const Jac<F>* __restrict__ lo, uint32_t half,
                                                   ScalarBits s, Jac<F>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Jac<F> p = hi[i];
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (int bit = s.nbits - 1; bit >= 0; --bit) {
        acc = dbl(acc);
        if ((s.w[bit >> 5] >> (bit & 31)) & 1u) acc = add(acc, p);
    }
    out[i] = add(acc, lo[i]);
}

// ---- G2 fold with the GLS endomorphism --------------------------------------------------------------------
// psi = twist o Frobenius o untwist acts on G2 as multiplication by x (p = x mod r), so with u = |x| (64 bit)
//   [u^j] Q = (conj^j(x_Q) * PSIj_CX, conj^j(y_Q) * PSIj_CY)          (constants from tools/gen_params.py)
// costs two Fp2 multiplications.  The shared 255-bit scalar s is written in base u on the host,
// s = d0 + d1 u + d2 u^2 + d3 u^3 (0 <= d_j < u), each d_j NAF-recoded, and every lane runs the SAME joint
// double-and-add over <= 65 digit positions: 65 doublings + ~87 mixed additions instead of 255 + ~128.
// The four affine points [u^j]Q of a lane are parked in HBM (chunked SoA, L2-resident) so only T stays in VGPRs.
struct GlsDigits { int8_t d[4][68]; int len; };
constexpr int G2A_CHUNKS = 12;     // 192 B

RIPP_HD G2A gls_image(const G2A& q, int j) {
    if (j == 0) return q;
    Fp2 cx, cy;
    if (j == 1) { cx = {fp_const(RIPP_PSI1_CX0), fp_const(RIPP_PSI1_CX1)}; cy = {fp_const(RIPP_PSI1_CY0), fp_const(RIPP_PSI1_CY1)}; }
    else if (j == 2) { cx = {fp_const(RIPP_PSI2_CX0), fp_const(RIPP_PSI2_CX1)}; cy = {fp_const(RIPP_PSI2_CY0), fp_const(RIPP_PSI2_CY1)}; }
    else { cx = {fp_const(RIPP_PSI3_CX0), fp_const(RIPP_PSI3_CX1)}; cy = {fp_const(RIPP_PSI3_CY0), fp_const(RIPP_PSI3_CY1)}; }
    const bool odd = (j & 1) != 0;
    return {mul(odd ? conj(q.x) : q.x, cx), mul(odd ? conj(q.y) : q.y, cy)};
}

// the two coordinates of gls_image separately (the low-liveness addition fetches them where it uses them)
RIPP_HD Fp2 gls_image_x(const Fp2& x, int j) {
    if (j == 0) return x;
    const Fp2 c = j == 1 ? Fp2{fp_const(RIPP_PSI1_CX0), fp_const(RIPP_PSI1_CX1)} : j == 2 ? Fp2{fp_const(RIPP_PSI2_CX0), fp_const(RIPP_PSI2_CX1)} : Fp2{fp_const(RIPP_PSI3_CX0), fp_const(RIPP_PSI3_CX1)};
    return mul((j & 1) ? conj(x) : x, c);
}
RIPP_HD Fp2 gls_image_y(const Fp2& y, int j) {
    if (j == 0) return y;
    const Fp2 c = j == 1 ? Fp2{fp_const(RIPP_PSI1_CY0), fp_const(RIPP_PSI1_CY1)} : j == 2 ? Fp2{fp_const(RIPP_PSI2_CY0), fp_const(RIPP_PSI2_CY1)} : Fp2{fp_const(RIPP_PSI3_CY0), fp_const(RIPP_PSI3_CY1)};
    return mul((j & 1) ? conj(y) : y, c);
}

__global__ void __launch_bounds__(64, RIPP_OCC) k_fold_g2_gls(const G2A* __restrict__ hi, const G2A* __restrict__ lo, uint32_t half, GlsDigits dg,
                                                     uint4* __restrict__ qtab, size_t stride, G2J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    {
        const G2A q = hi[i];
#pragma unroll 1
        for (int j = 0; j < 4; ++j) store_chunks<G2A_CHUNKS>(qtab, j, stride, i, gls_image(q, j));
    }
    G2J acc = jac_inf<Fp2>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int d = dg.d[j][pos];
            if (d != 0) {
                G2A q = load_chunks<G2A_CHUNKS, G2A>(qtab, j, stride, i);
                if (d < 0) q.y = neg(q.y);
                acc = add_mixed(acc, q);
            }
        }
    }
    out[i] = add_mixed(acc, lo[i]);
}

// Latency form of the GLS fold for SMALL rounds (the chip is mostly idle, a lone lane needs ~2.4 us per dependent
// Fp product): the four sub-scalar multiplications d_j * [u^j]Q run on four different lanes.  blockIdx.y = j, so
// every wave still has uniform control flow (one digit string per wave); 65 dbl + ~22 adds per lane instead of
// 65 + ~87.  k_fold_g2_combine then sums the four partial points and adds lo.  1.9x the total work of the
// single-lane form, so the engine uses it only below a size threshold.
__global__ void __launch_bounds__(64, 2) k_fold_g2_gls_split(const G2A* __restrict__ hi, uint32_t half, GlsDigits dg, G2J* __restrict__ parts /* [4][half] */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const int j = blockIdx.y;
    const G2A q = gls_image(hi[i], j);
    G2J acc = jac_inf<Fp2>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
        const int d = dg.d[j][pos];
        if (d != 0) { G2A t = q; if (d < 0) t.y = neg(t.y); acc = add_mixed(acc, t); }
    }
    parts[(size_t)j * half + i] = acc;
}
__global__ void __launch_bounds__(64) k_fold_g2_combine(const G2J* __restrict__ parts, const G2A* __restrict__ lo, uint32_t half, G2J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    G2J acc = add(add(parts[i], parts[(size_t)half + i]), add(parts[2 * (size_t)half + i], parts[3 * (size_t)half + i]));
    out[i] = add_mixed(acc, lo[i]);
}

// out[i] = sum of the 8 partial points parts[t][i] (two 4-string GLS multiplications of one element: the fold that returns an x-scaled
// vector to the plain one, engine.hip job_fold)
__global__ void __launch_bounds__(64) k_fold_g2_combine8(const G2J* __restrict__ parts, uint32_t half, G2J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    G2J acc = add(parts[i], parts[(size_t)half + i]);
#pragma unroll 1
    for (int t = 2; t < 8; ++t) acc = add(acc, parts[(size_t)t * half + i]);
    out[i] = acc;
}

// Single-scalar NAF fold (G1 with the 128-bit SIPP challenge): out[i] = s*hi[i] + lo[i]
struct NafDigits { int8_t d[260]; int len; };
template <class F>
__global__ void __launch_bounds__(256) k_fold_affine_naf(const Affine<F>* __restrict__ hi, const Affine<F>* __restrict__ lo, uint32_t half,
                                                          NafDigits dg, Jac<F>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Affine<F> p = hi[i];
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
        const int d = dg.d[pos];
        if (d != 0) { Affine<F> q = p; if (d < 0) q.y = neg(q.y); acc = add_mixed(acc, q); }
    }
    out[i] = add_mixed(acc, lo[i]);
}

// G1 fold with a FULL-WIDTH scalar (GIPA's c, ip_proofs/src/gipa.rs:262-266, 286-290) through the GLV endomorphism
// phi(x, y) = (beta x, y) = [lambda](x, y), lambda = u^2 - 1 ~ sqrt(r):  s = s1 + s2 lambda with s1, s2 <= 128 bits, so
//   out[i] = s1 * P + s2 * phi(P) + lo[i]
// costs 128 doublings + ~86 additions instead of 255 + ~85.  d1 / d2: NAF digit strings of s1 / s2 (shared by the launch).
struct GlvDigits { int8_t d1[132]; int8_t d2[132]; int len; };
__global__ void __launch_bounds__(256) k_fold_g1_glv(const G1A* __restrict__ hi, const G1A* __restrict__ lo, uint32_t half, GlvDigits dg, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const G1A p = hi[i];
    G1A q = p; q.x = fmul(p.x, fp_const(RIPP_GLV_BETA));          // phi(P); the identity (0,0) maps to itself
    G1J acc = jac_inf<Fp>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
        const int d1 = dg.d1[pos], d2 = dg.d2[pos];
        if (d1 != 0) { G1A t = p; if (d1 < 0) t.y = neg(t.y); acc = add_mixed(acc, t); }
        if (d2 != 0) { G1A t = q; if (d2 < 0) t.y = neg(t.y); acc = add_mixed(acc, t); }
    }
    out[i] = add_mixed(acc, lo[i]);
}

// ---- round-0 folds with a PRECOMPUTED second base ---------------------------------------------------------------------
// In SIPP::prove the first challenge needs the Blake2s digest of the whole statement, which the host finishes ~90 ms after the GPU
// has finished the first round's pairing products (n = 2^20).  That window is used to compute hi2[i] = 2^K * hi[i] for both
// vectors (K = 64 on G1, K = 32 on G2): the fold scalar is then split at bit K, s = s_lo + 2^K s_hi, and
//     out[i] = s_lo * hi[i] + s_hi * hi2[i] + lo[i]
// needs K doublings instead of 2K -- the other K were done before the challenge was known.
template <class F>
__global__ void __launch_bounds__(256, 2) k_pow2_mul(const Affine<F>* __restrict__ in, uint32_t n, int k, Jac<F>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jac<F> acc = to_jac(in[i]);
#pragma unroll 1
    for (int t = 0; t < k; ++t) acc = dbl(acc);
    out[i] = acc;
}
// G1: s_lo, s_hi are the 64-bit halves of the 128-bit challenge (sipp/src/lib.rs:85-91), NAF digit strings in dg.d1 / dg.d2
__global__ void __launch_bounds__(256) k_fold_g1_two(const G1A* __restrict__ hi, const G1A* __restrict__ hi2, const G1A* __restrict__ lo, uint32_t half,
                                                      GlvDigits dg, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const G1A p = hi[i], q = hi2[i];
    G1J acc = jac_inf<Fp>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
        const int d1 = dg.d1[pos], d2 = dg.d2[pos];
        if (d1 != 0) { G1A t = p; if (d1 < 0) t.y = neg(t.y); acc = add_mixed(acc, t); }
        if (d2 != 0) { G1A t = q; if (d2 < 0) t.y = neg(t.y); acc = add_mixed(acc, t); }
    }
    out[i] = add_mixed(acc, lo[i]);
}
// G2: every base-u digit of the GLS decomposition is split at bit 32: eight digit strings of <= 33 NAF digits over the eight bases
// psi^j(Q), psi^j(2^32 Q) (psi commutes with doubling): 33 doublings + ~88 additions instead of 65 + ~87.
struct Gls8Digits { int8_t d[8][36]; int len; };
__global__ void __launch_bounds__(64, RIPP_OCC) k_fold_g2_gls8(const G2A* __restrict__ hi, const G2A* __restrict__ hi2, const G2A* __restrict__ lo, uint32_t half,
                                                      Gls8Digits dg, uint4* __restrict__ qtab, size_t stride, G2J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    {
        const G2A q = hi[i], q2 = hi2[i];
#pragma unroll 1
        for (int j = 0; j < 4; ++j) { store_chunks<G2A_CHUNKS>(qtab, j, stride, i, gls_image(q, j)); store_chunks<G2A_CHUNKS>(qtab, 4 + j, stride, i, gls_image(q2, j)); }
    }
    G2J acc = jac_inf<Fp2>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
#pragma unroll 1
        for (int t = 0; t < 8; ++t) {
            const int d = dg.d[t][pos];
            if (d != 0) {
                G2A q = load_chunks<G2A_CHUNKS, G2A>(qtab, t, stride, i);
                if (d < 0) q.y = neg(q.y);
                acc = add_mixed(acc, q);
            }
        }
    }
    out[i] = add_mixed(acc, lo[i]);
}

// ---- round-0 folds over PRECOMPUTED odd multiples (width-4 wNAF) -----------------------------------------------------------------
// The same hash window also fits two more bases per element and tables of the odd multiples {1, 3, .., 2^(w-1) - 1} of all four (w = RIPP_FOLD_W; k_odd_multiples,
// then the batch normalisation; for G2 the psi images of all of them, k_g2_tab_images), so the digit strings of the fold become
// width-w wNAF strings of 32 (G1) / 16 (G2) bits -- one addition per w + 1 digit positions instead of one per 3, and half the doublings
// again.  w = 5: G1 33 doublings + ~22 additions (two-base NAF form: 65 + ~43); G2: 17 doublings + ~45 additions (33 + ~88).
// Same group elements, hence the same proof bytes.
#ifndef RIPP_FOLD_W
#define RIPP_FOLD_W 5                                   // wNAF width of the table folds
#endif
constexpr int FOLD_TAB_M = 1 << (RIPP_FOLD_W - 2);      // odd multiples 1, 3, .., 2 M - 1 per base
template <class F>
__global__ void __launch_bounds__(64, 2) k_odd_multiples(const Affine<F>* __restrict__ base, uint32_t n, int M, Jac<F>* __restrict__ out) {   // out[m][i] = (2m + 3) base[i], m < M - 1
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<F> b = base[i];
    const Jac<F> b2 = dbl(to_jac(b));
    Jac<F> t = add_mixed(b2, b);
    out[i] = t;
#pragma unroll 1
    for (int m = 1; m < M - 1; ++m) { t = add(t, b2); out[(size_t)m * n + i] = t; }
}
// Four bases per element (G1: 2^(32 b) hi[i], G2: 2^(16 b) hi[i], b < 4), so the chain has 32 / 16 doublings.
struct Wnaf4 { int8_t d[4][36]; int len; };            // G1: string b = 32-bit word b of the 128-bit challenge
struct Wnaf16 { int8_t d[16][20]; int len; };          // G2: string 4 b + j = 16-bit piece b of GLS digit j
// tab[e][i], e = M b + m: (2m + 1) * (base b of element i)
__global__ void __launch_bounds__(256) k_fold_g1_tab(const G1A* __restrict__ tab, size_t tstride, int M, const G1A* __restrict__ lo, uint32_t half, Wnaf4 dg, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    G1J acc = jac_inf<Fp>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const int d = dg.d[t][pos];
            if (d != 0) { G1A q = tab[(size_t)(M * t + ((d < 0 ? -d : d) >> 1)) * tstride + i]; if (d < 0) q.y = neg(q.y); acc = add_mixed(acc, q); }
        }
    }
    out[i] = add_mixed(acc, lo[i]);
}
// qtab row (4 b + j) M + m = psi^j((2m + 1) * base b), chunked like the other G2 tables; mult[e][i], e = M b + m
__global__ void __launch_bounds__(64, RIPP_OCC) k_g2_tab_images(const G2A* __restrict__ mult, uint32_t half, int M, uint4* __restrict__ qtab, size_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const int e = blockIdx.y, b = e / M, m = e % M;
    const G2A q = mult[(size_t)e * half + i];
#pragma unroll 1
    for (int j = 0; j < 4; ++j) store_chunks<G2A_CHUNKS>(qtab, (size_t)(4 * b + j) * M + m, stride, i, gls_image(q, j));
}
// ---- low-register G2 group law for the fold kernels ----------------------------------------------------------------------------------
// With Fp2 coordinates the textbook formulas keep ~9 temporaries (216 dwords) beside the accumulator: the fold kernels spilled 200-250
// dwords.  These forms compute the same values in a LOW-LIVENESS order pinned with scheduling barriers (at most 5-6 Fp2 live), read the
// table point where it is used and park Y1 in LDS while it is idle.  They do NOT handle the exceptional cases (accumulator at infinity,
// T = +-Q, table point at infinity): they REPORT them, and the kernel recomputes such a lane with the complete formulas of curve.hpp.
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void jdbl_lo(Fp2& X, Fp2& Y, Fp2& Z) {          // dbl-2009-l
    Z = dbl(fmul(Y, Z)); SB();
    const Fp2 A = fsqr(X); SB();
    const Fp2 B = fsqr(Y); SB();
    const Fp2 t = fsqr(add(X, B)); SB();
    const Fp2 C = fsqr(B); SB();
    const Fp2 D = dbl(sub(sub(t, A), C)); SB();
    const Fp2 E = add(dbl(A), A); SB();
    X = sub(sub(fsqr(E), D), D); SB();
    Y = sub(fmul(E, sub(D, X)), dbl(dbl(dbl(C))));
}
// madd-2007-bl; loadx / loady fetch the affine addend's coordinates; park: this lane's 6 x 16 B LDS column (stride 64 lanes).
// Returns true when the result is NOT valid (H = 0, i.e. T = +-Q, or Q at infinity).
template <class LOADX, class LOADY>
__device__ __forceinline__ bool jmadd_lo(Fp2& X, Fp2& Y, Fp2& Z, LOADX loadx, LOADY loady, bool negy, uint4* park) {
    const Fp2 Z1Z1 = fsqr(Z); SB();
    Fp2 H; bool qinf; { const Fp2 x2 = loadx(); qinf = x2.is_zero(); H = sub(fmul(x2, Z1Z1), X); } SB();
    Fp2 r; { Fp2 y2 = loady(); qinf = qinf && y2.is_zero(); if (negy) y2 = neg(y2); const Fp2 t = fmul(Z, Z1Z1); SB(); r = sub(fmul(y2, t), Y); } SB();
    const bool special = H.is_zero() || qinf;
    { const uint4* src = reinterpret_cast<const uint4*>(&Y);              // Y1 rests in LDS until the last product
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
#undef SB
// NS digit strings; string t = 4 b + j works on table rows t M .. t M + M - 1.  Complete formulas: the reference form and the per-lane fallback.
template <class D, int NS>
__device__ __noinline__ void fold_g2_tab_complete(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t i, const D& dg, G2J* __restrict__ out) {
    G2J acc = jac_inf<Fp2>();
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        acc = dbl(acc);
#pragma unroll 1
        for (int t = 0; t < NS; ++t) {
            const int d = dg.d[t][pos];
            if (d != 0) {
                G2A q = load_chunks<G2A_CHUNKS, G2A>(qtab, (size_t)t * M + ((d < 0 ? -d : d) >> 1), stride, i);
                if (d < 0) q.y = neg(q.y);
                acc = add_mixed(acc, q);
            }
        }
    }
    out[i] = add_mixed(acc, lo[i]);
}
template <class D, int NS>
__global__ void __launch_bounds__(64, RIPP_OCC) k_fold_g2_tab(const uint4* __restrict__ qtab, size_t stride, int M, const G2A* __restrict__ lo, uint32_t half, D dg, G2J* __restrict__ out) {
    __shared__ uint4 park_[6 * 64];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    uint4* park = park_ + threadIdx.x;
    Fp2 X = Fp2::one(), Y = Fp2::one(), Z = Fp2::zero();
    bool inf = true, bad = false;                                        // inf is wave-uniform: the digit strings are shared by the launch
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        if (!inf) jdbl_lo(X, Y, Z);
#pragma unroll 1
        for (int t = 0; t < NS; ++t) {
            const int d = dg.d[t][pos];
            if (d == 0) continue;
            const uint4* base = qtab + ((size_t)t * M + ((d < 0 ? -d : d) >> 1)) * G2A_CHUNKS * stride + i;
            auto loadx = [&]() { Fp2 v; uint4* dd = reinterpret_cast<uint4*>(&v);
#pragma unroll
                for (int q = 0; q < 6; ++q) dd[q] = base[(size_t)q * stride]; return v; };
            auto loady = [&]() { Fp2 v; uint4* dd = reinterpret_cast<uint4*>(&v);
#pragma unroll
                for (int q = 0; q < 6; ++q) dd[q] = base[(size_t)(6 + q) * stride]; return v; };
            if (inf) {                                                    // first addition: acc <- +-Q
                X = loadx(); Y = loady(); if (d < 0) Y = neg(Y); Z = Fp2::one(); inf = false;
                bad |= X.is_zero() && Y.is_zero();
            } else bad |= jmadd_lo(X, Y, Z, loadx, loady, d < 0, park);
        }
    }
    {
        const G2A* lp = lo + i;
        bool linf; { const G2A q = *lp; linf = is_inf(q); if (inf) { X = q.x; Y = q.y; Z = linf ? Fp2::zero() : Fp2::one(); } }
        if (!inf && !linf) bad |= jmadd_lo(X, Y, Z, [&]() { return opaque(lp)->x; }, [&]() { return opaque(lp)->y; }, false, park);
    }
    if (bad) fold_g2_tab_complete<D, NS>(qtab, stride, M, lo, i, dg, out);
    else out[i] = G2J{X, Y, Z};
}

// ---- batch normalisation (CurveGroup::normalize_batch) ------------------------------------------------------
// Lane t handles points t, t+T, ..., one inversion per lane (Montgomery's trick over its K points).  The running
// prefix products are parked in out[i].x, so `in` and `out` must not alias.
template <class F>
__global__ void __launch_bounds__(256) k_normalize(const Jac<F>* __restrict__ in, uint32_t n, Affine<F>* __restrict__ out, uint32_t T) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    F acc = F::one();
#pragma unroll 1
    for (uint32_t i = t; i < n; i += T) {
        const F z = in[i].z;
        if (!z.is_zero()) { out[i].x = acc; acc = fmul(acc, z); }
    }
    F ainv = finv(acc);
    uint32_t last = t + ((n - 1 - t) / T) * T;       // largest index of this lane (t < n guaranteed by T <= n)
#pragma unroll 1
    for (uint32_t i = last;; i -= T) {
        const Jac<F> p = in[i];
        if (p.z.is_zero()) { out[i] = aff_inf<F>(); }
        else {
            const F zi = fmul(ainv, out[i].x);
            ainv = fmul(ainv, p.z);
            const F zi2 = fsqr(zi);
            Affine<F> o; o.x = fmul(p.x, zi2); o.y = fmul(p.y, fmul(zi2, zi));
            out[i] = o;
        }
        if (i == t) break;
    }
}

// (Build round 5 tried an LDS-transposing twin -- a wave copies 64 consecutive structs as one contiguous run of 16-byte chunks, lanes pick theirs out of
// LDS at an odd chunk pitch, results leave the same way.  Measured on the proof's launch mix: 1.17 ms per average G2 launch against 0.51 ms for this
// kernel -- six barriers per step and one wave per 19 KB of LDS cost more than the uncoalesced struct accesses, which the L2 absorbs.  Not kept.)

// ---- per-element scalar multiplication, either group: out[i] = k[i] * base[i * base_stride]
// (a_r = a_i * r^i and ck_1_r = ck_i * r^-i of groth16_aggregation.rs:119-131; base_stride = 0 broadcasts one base,
//  which is structured_generators_scalar_power of tipa/mod.rs:372-391).  Plain MSB-first double-and-add: every lane
// has its own scalar, so the add is data-dependent and executes under the wave's EXEC mask.
template <class F>
__global__ void __launch_bounds__(256, 2) k_scale_pts(const Affine<F>* __restrict__ base, uint32_t base_stride, const Fr* __restrict__ k_mont, uint32_t n, Jac<F>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr k = from_mont(k_mont[i]);
    const Affine<F> p = base[(size_t)i * base_stride];
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (int bit = 254; bit >= 0; --bit) {
        acc = dbl(acc);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1u) acc = add_mixed(acc, p);
    }
    out[i] = acc;
}

// scalar-vector fold of GIPA with a structured scalar message: out[i] = hi[i] * s + lo[i]  (gipa.rs:270-274 with Message = Fr)
__global__ void __launch_bounds__(256) k_fold_fr(const Fr* __restrict__ hi, const Fr* __restrict__ lo, uint32_t half, Fr s, Fr* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    out[i] = add(mul(hi[i], s), lo[i]);
}

// ScalarInnerProduct::inner_product (inner_products/src/lib.rs:149-166): sum_i l_i * r_i in Fr.  Each lane sums a strided subset, one
// LDS tree per block, one partial per block (the host adds the <= 1024 partials).
__global__ void __launch_bounds__(256) k_fr_dot(const Fr* __restrict__ l, const Fr* __restrict__ r, uint32_t n, Fr* __restrict__ partials) {
    __shared__ Fr sh[256];
    Fr acc = Fr::zero();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc = add(acc, mul(l[i], r[i]));
    sh[threadIdx.x] = acc; __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] = add(sh[threadIdx.x], sh[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// ---- synthetic inputs (bench harness) ---------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_synth_points(Affine<F> g, uint64_t start, uint64_t first, uint64_t stride_, uint32_t n, Jac<F>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = start + first + (uint64_t)i * stride_;
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (int bit = 63; bit >= 0; --bit) { acc = dbl(acc); if ((k >> bit) & 1ull) acc = add_mixed(acc, g); }
    out[i] = acc;
}
__device__ __forceinline__ uint64_t splitmix_at(uint64_t seed, uint64_t draw /* 1-based */) {
    uint64_t z = seed + draw * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_synth_fr(uint64_t seed, uint64_t first, uint64_t stride_, uint32_t n, Fr* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t e = first + (uint64_t)i * stride_;
    Fr v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint64_t d = splitmix_at(seed, 4 * e + j + 1); v.l[2 * j] = (uint32_t)d; v.l[2 * j + 1] = (uint32_t)(d >> 32); }
    v.l[7] &= 0x3fffffffu;                      // 254-bit canonical integer, always < r
    out[i] = to_mont(v);
}

}  // namespace ripp
