// Fp inversion on the device by Kaliski's almost-Montgomery inverse (binary extended Euclid) instead of Fermat's
// a^(p-2): ~760 iterations of 384-bit add/sub/shift (~100 VALU ops each, fully predicated so a wave stays uniform)
// versus 610 dependent Montgomery products (~700 ops each) -- about 5x fewer instructions on the critical path of
// every batch normalisation (one inversion per lane).
//   phase 1:  u = p, v = a, r = 0, s = 1, k = 0;  while v > 0: the four classic cases, k += 1
//             => r = -a^-1 * 2^k (mod p),  381 <= k <= 762
//   phase 2:  the input is a~ = a R (Montgomery form), so x = p - r = a^-1 R^-1 2^k; one Montgomery product with
//             KALISKI_FIX[k] = R^3 2^-k gives a^-1 R, the Montgomery form of the inverse.
#pragma once
#include "fp.hpp"

namespace ripp {
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ const uint32_t KALISKI_FIX[769][12] = {
#if defined(RIPP_BLS12_377)
#include "../bls12_377/inv_table.inc"
#else
#include "inv_table.inc"
#endif
};

__device__ __noinline__ inline Fp fp_inv_kaliski(const Fp& a) {
    uint32_t u[12], v[12], r[13], s[13];
#pragma unroll
    for (int i = 0; i < 12; ++i) { u[i] = FpParams::mod(i); v[i] = a.l[i]; r[i] = 0; s[i] = 0; }
    r[12] = 0; s[12] = 0; s[0] = 1;
    uint32_t k = 0;
    bool live = !a.is_zero();
#pragma unroll 1
    for (int it = 0; it < 768; ++it) {
        if (!__any(live)) break;
        // case selection (u is odd whenever v is odd at the comparison, standard invariant)
        const bool u_even = (u[0] & 1u) == 0, v_even = (v[0] & 1u) == 0;
        uint32_t d[12], bo = 0;                          // d = u - v, bo = (u < v)
#pragma unroll
        for (int i = 0; i < 12; ++i) d[i] = subb32(u[i], v[i], bo);
        uint32_t e[12], bo2 = 0;                         // e = v - u, bo2 = (v < u)
#pragma unroll
        for (int i = 0; i < 12; ++i) e[i] = subb32(v[i], u[i], bo2);
        const bool u_gt_v = bo2 != 0;                     // STRICT: at u == v (== 1, the last step) the v-branch must run
        (void)bo;
        const bool cA = live && u_even, cB = live && !u_even && v_even, cC = live && !u_even && !v_even && u_gt_v, cD = live && !u_even && !v_even && !u_gt_v;
        uint32_t rs[13], c = 0;                          // rs = r + s
#pragma unroll
        for (int i = 0; i < 13; ++i) rs[i] = addc32(r[i], s[i], c);
        // new u: A: u/2, C: (u-v)/2, else u          new v: B: v/2, D: (v-u)/2, else v
        uint32_t nu[12], nv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) { nu[i] = cC ? d[i] : u[i]; nv[i] = cD ? e[i] : v[i]; }
        const bool su = cA || cC, sv = cB || cD;
#pragma unroll
        for (int i = 0; i < 11; ++i) { u[i] = su ? ((nu[i] >> 1) | (nu[i + 1] << 31)) : nu[i]; v[i] = sv ? ((nv[i] >> 1) | (nv[i + 1] << 31)) : nv[i]; }
        u[11] = su ? (nu[11] >> 1) : nu[11]; v[11] = sv ? (nv[11] >> 1) : nv[11];
        // r, s:  A: s *= 2;  B: r *= 2;  C: r += s, s *= 2;  D: s += r, r *= 2
        uint32_t nr[13], ns[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) { nr[i] = cC ? rs[i] : r[i]; ns[i] = cD ? rs[i] : s[i]; }
        const bool dr = cB || cD, ds = cA || cC;
#pragma unroll
        for (int i = 12; i >= 1; --i) { r[i] = dr ? ((nr[i] << 1) | (nr[i - 1] >> 31)) : nr[i]; s[i] = ds ? ((ns[i] << 1) | (ns[i - 1] >> 31)) : ns[i]; }
        r[0] = dr ? (nr[0] << 1) : nr[0]; s[0] = ds ? (ns[0] << 1) : ns[0];
        k += live ? 1u : 0u;
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) nz |= v[i];
        live = live && (nz != 0);
    }
    // r < 2p (13 limbs): bring into [0, p), then x = p - r
    Fp x;
    {
        uint32_t t[13], bo = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) t[i] = subb32(r[i], FpParams::mod(i), bo);
        t[12] = subb32(r[12], 0u, bo);
        const bool ge = bo == 0;
        uint32_t w[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) w[i] = ge ? t[i] : r[i];
        uint32_t b2 = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) x.l[i] = subb32(FpParams::mod(i), w[i], b2);
    }
    Fp fix;
    const uint32_t kk = k > 768u ? 768u : k;
#pragma unroll
    for (int i = 0; i < 12; ++i) fix.l[i] = KALISKI_FIX[kk][i];
    Fp out = mul(x, fix);
    if (a.is_zero()) out = Fp::zero();
    return out;
}
#endif
// finv: the inversion the kernels use -- Kaliski on the device, Fermat (fp.hpp) on the host
#if defined(__HIP_DEVICE_COMPILE__)
RIPP_HD Fp finv(const Fp& a) { return fp_inv_kaliski(a); }
#else
RIPP_HD Fp finv(const Fp& a) { return inv(a); }
#endif
}  // namespace ripp
