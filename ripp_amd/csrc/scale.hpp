// Per-element scalar multiplication  out[i] = k[i] * base[i]  on G1 with the GLV endomorphism and signed fixed windows
//   -- `a.mul(r)` of sipp/src/lib.rs:61-65 and :189-194, and a_r = a_i * r^i of groth16_aggregation.rs:119-123.
//
// The plain form (k_scale_pts in kernels.hpp) walks 255 bits MSB-first; every lane has its OWN scalar, so in a wave of 64 the
// addition under `if (bit)` executes in practically every iteration: 255 doublings + 255 additions.  Here every lane
//   * splits k = k1 + k2 * lambda  (lambda = z^2 - 1, phi(x, y) = (beta x, y) = [lambda](x, y); both halves < 2^128; Barrett division
//     of msm.hpp),
//   * writes the multiples 1..8 of its base to a table in HBM (chunked, lane-coalesced; 1.2 GB at n = 2^20 -- this is what 288 GB are for),
//   * walks 33 signed base-16 digit positions of both halves jointly: 4 doublings, acc += +-T[|d1|], acc += +-phi(T[|d2|]).
// 132 doublings + <= 66 additions with UNIFORM control flow (a zero digit, probability 1/16, idles its lane for one addition).
// Group elements are assumed to lie in G1 proper, like everywhere the endomorphisms are used (include/ripp_hip.h).
#pragma once
#include "msm.hpp"

namespace ripp {

constexpr int SCALE_TAB = 8;                 // multiples 1..8
constexpr int G1J_CHUNKS = 9;                // 144 B

// v (4 limbs, < 2^128) -> v + 0x88..8 (33 nibbles): signed digit j = nibble_j - 8, in [-8, 7]
__device__ __forceinline__ void scale_bias(const uint32_t* v, uint32_t* o) {
    uint64_t c = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) { const uint64_t s = (uint64_t)v[t] + 0x88888888u + c; o[t] = (uint32_t)s; c = s >> 32; }
    o[4] = (uint32_t)c + 8u;
}
__device__ __forceinline__ int scale_digit(const uint32_t* o, int j) { return (int)((o[j >> 3] >> ((j & 7) * 4)) & 15u) - 8; }

// tab: [(e * 9 + q) * n + i] 16-byte chunks, e < 8 (multiple e + 1), written and read by lane i only
__global__ void __launch_bounds__(256, 2) k_scale_g1_glv(const G1A* __restrict__ base, uint32_t base_stride, const Fr* __restrict__ k_mont, uint32_t n,
                                                         uint4* __restrict__ tab, G1J* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d1[5], d2[5];
    {
        Fr k = from_mont(k_mont[i]);
        const uint32_t lam[8] = RIPP_GLV_LAMBDA;
        const uint32_t lam_mu[5] = RIPP_GLV_LAMBDA_MU;                  // floor(2^256 / lambda)
        uint32_t rem[5];
        msm_divmod<4, 5>(k.l, lam, lam_mu, rem);                      // k = q * lambda + rem
        scale_bias(rem, d1); scale_bias(k.l, d2);
    }
    {   // multiples 1..8 of the base, Jacobian (no inversion): 1 doubling + 6 mixed additions
        const G1A p = base[(size_t)i * base_stride];
        G1J t = to_jac(p);
        store_chunks<G1J_CHUNKS>(tab, 0, n, i, t);
        t = dbl(t);
        store_chunks<G1J_CHUNKS>(tab, 1, n, i, t);
#pragma unroll 1
        for (int e = 2; e < SCALE_TAB; ++e) { t = add_mixed(t, p); store_chunks<G1J_CHUNKS>(tab, e, n, i, t); }
    }
    const Fp beta = fp_const(RIPP_GLV_BETA);
    G1J acc = jac_inf<Fp>();
#pragma unroll 1
    for (int j = 32; j >= 0; --j) {
        if (j != 32) { acc = dbl(acc); acc = dbl(acc); acc = dbl(acc); acc = dbl(acc); }
        const int a = scale_digit(d1, j), b = scale_digit(d2, j);
        if (a != 0) {
            G1J t = load_chunks<G1J_CHUNKS, G1J>(tab, (a < 0 ? -a : a) - 1, n, i);
            if (a < 0) t.y = neg(t.y);
            acc = add(acc, t);
        }
        if (b != 0) {
            G1J t = load_chunks<G1J_CHUNKS, G1J>(tab, (b < 0 ? -b : b) - 1, n, i);
            if (b < 0) t.y = neg(t.y);
            t.x = fmul(t.x, beta);                                     // phi in Jacobian coordinates: (beta X, Y, Z)
            acc = add(acc, t);
        }
    }
    out[i] = acc;
}

}  // namespace ripp
