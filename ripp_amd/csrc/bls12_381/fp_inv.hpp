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
#if defined(__HIP_DEVICE_COMPILE__)
}  // namespace ripp
#include "../fq28.hpp"
namespace ripp {
// Binary GCD with 30-step inner loops on 64-bit approximations (Pornin, "Optimized binary GCD for modular inversion", eprint 2020/972, Alg. 2):
// a = y, b = p, u = 1, v = 0; 26 times: take the low 30 and the top 34 bits of (a, b), run 30 steps of the binary GCD on those 64-bit values
// while collecting the update factors (f0 g0; f1 g1), |f|, |g| <= 2^30; then (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^30 exactly and
// (u, v) <- (f0 u + g0 v, f1 u + g1 v) / 2^30 MOD p (one Montgomery-style correction k p), which keeps a = u y, b = v y (mod p): after
// 26 x 30 >= 2 * 381 - 1 steps a = 0, b = 1 and v = y^-1.  All long values are 14 limbs of 28 bits (fq28.hpp): every update is one pass of
// signed multiply-adds into 64-bit columns + one signed carry pass.  ~36 K instructions against ~95 K for the bit-serial Kaliski form --
// and this IS the latency of every batch normalisation of a small round (one inversion per lane).  Model: tools/inv_model.py.
__device__ __noinline__ inline Fp fp_inv_bingcd(const Fp& y) {
    using namespace fq28;
    constexpr int K = 30, ITER = (2 * FpParams::BITS - 1 + K - 1) / K;
    constexpr uint32_t KMASK = (1u << K) - 1u, MINV = RIPP_FP_INV & KMASK;
    int32_t a[NL], b[NL], u[NL], v[NL];
    { const Fqn t = fq_unpack(y.l);
#pragma unroll
      for (int i = 0; i < NL; ++i) { a[i] = (int32_t)t.l[i]; b[i] = (int32_t)P28.l[i]; u[i] = 0; v[i] = 0; }
      u[0] = 1; }
    // signed carry pass over columns (the top word keeps the rest, signed) followed by the exact division by 2^30
    auto carry_shr = [&](const int64_t (&col)[NL], int32_t (&out)[NL]) {
        int64_t c = 0; uint32_t l[NL - 1];
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) { const int64_t t = col[i] + c; l[i] = (uint32_t)t & MASK; c = t >> W; }
        const int64_t T = col[NL - 1] + c;                      // weight 2^364, up to ~48 bits
#pragma unroll
        for (int j = 0; j <= NL - 4; ++j) out[j] = (int32_t)((l[j + 1] >> 2) | ((l[j + 2] & 3u) << 26));
        out[NL - 3] = (int32_t)((l[NL - 2] >> 2) | (((uint32_t)T & 3u) << 26));
        out[NL - 2] = (int32_t)((uint32_t)(T >> 2) & MASK);
        out[NL - 1] = (int32_t)(T >> 30);
    };
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        // ---- approximations: low 30 bits + the 34 bits below the top bit of (a | b)
        uint32_t ah = 0, am = 0, al = 0, bh = 0, bm = 0, bl = 0; bool found = false, at2 = false;
#pragma unroll
        for (int i = NL - 1; i >= 2; --i) {
            const bool take = !found && ((a[i] | b[i]) != 0);
            ah = take ? (uint32_t)a[i] : ah; am = take ? (uint32_t)a[i - 1] : am; al = take ? (uint32_t)a[i - 2] : al;
            bh = take ? (uint32_t)b[i] : bh; bm = take ? (uint32_t)b[i - 1] : bm; bl = take ? (uint32_t)b[i - 2] : bl;
            at2 = take ? (i == 2) : at2;
            found = found || take;
        }
        const uint64_t lo_a = (uint64_t)(uint32_t)a[0] | ((uint64_t)(uint32_t)a[1] << W), lo_b = (uint64_t)(uint32_t)b[0] | ((uint64_t)(uint32_t)b[1] << W);
        const int nb = 32 - __clz((int)(ah | bh | 1u));           // bit length of the top limb, 1..28 (only used when found)
        const uint64_t ta = ((uint64_t)am << W) | al, tb = ((uint64_t)bm << W) | bl;
        const uint64_t ha = ((uint64_t)ah << (34 - nb)) + (ta >> (nb + 22)), hb = ((uint64_t)bh << (34 - nb)) + (tb >> (nb + 22));
        uint64_t a_ = (lo_a & KMASK) | (ha << K), b_ = (lo_b & KMASK) | (hb << K);
        const bool exact64 = !found || (at2 && nb <= 8);          // bit length <= 64: the values themselves
        a_ = exact64 ? (lo_a | ((uint64_t)ah << 56 & (found ? ~0ull : 0ull))) : a_;
        b_ = exact64 ? (lo_b | ((uint64_t)bh << 56 & (found ? ~0ull : 0ull))) : b_;
        // ---- 30 steps on the approximations
        int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 2
        for (int i = 0; i < K; ++i) {
            const bool odd = (a_ & 1ull) != 0, sw = odd && a_ < b_;
            const uint64_t ta_ = sw ? b_ : a_, tb_ = sw ? a_ : b_;
            const int32_t tf0 = sw ? f1 : f0, tf1 = sw ? f0 : f1, tg0 = sw ? g1 : g0, tg1 = sw ? g0 : g1;
            a_ = (ta_ - (odd ? tb_ : 0ull)) >> 1; b_ = tb_;
            f0 = tf0 - (odd ? tf1 : 0); g0 = tg0 - (odd ? tg1 : 0);
            f1 = tf1 << 1; g1 = tg1 << 1;
        }
        // ---- (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^30, made non-negative
        int32_t na[NL], nbv[NL];
        { int64_t c0[NL], c1[NL];
#pragma unroll
          for (int i = 0; i < NL; ++i) { c0[i] = (int64_t)f0 * a[i] + (int64_t)g0 * b[i]; c1[i] = (int64_t)f1 * a[i] + (int64_t)g1 * b[i]; }
          carry_shr(c0, na); carry_shr(c1, nbv); }
        const bool nega = na[NL - 1] < 0, negb = nbv[NL - 1] < 0;
        { int64_t c = 0;
#pragma unroll
          for (int i = 0; i < NL - 1; ++i) { const int64_t t = (nega ? -(int64_t)na[i] : (int64_t)na[i]) + c; a[i] = (int32_t)((uint32_t)t & MASK); c = t >> W; }
          a[NL - 1] = (int32_t)((nega ? -(int64_t)na[NL - 1] : (int64_t)na[NL - 1]) + c); }
        { int64_t c = 0;
#pragma unroll
          for (int i = 0; i < NL - 1; ++i) { const int64_t t = (negb ? -(int64_t)nbv[i] : (int64_t)nbv[i]) + c; b[i] = (int32_t)((uint32_t)t & MASK); c = t >> W; }
          b[NL - 1] = (int32_t)((negb ? -(int64_t)nbv[NL - 1] : (int64_t)nbv[NL - 1]) + c); }
        f0 = nega ? -f0 : f0; g0 = nega ? -g0 : g0; f1 = negb ? -f1 : f1; g1 = negb ? -g1 : g1;
        // ---- (u, v) <- (f0 u + g0 v, f1 u + g1 v) / 2^30 mod p
        int32_t nu[NL], nv[NL];
        { int64_t c0[NL], c1[NL];
#pragma unroll
          for (int i = 0; i < NL; ++i) { c0[i] = (int64_t)f0 * u[i] + (int64_t)g0 * v[i]; c1[i] = (int64_t)f1 * u[i] + (int64_t)g1 * v[i]; }
          const uint32_t k0 = (((uint32_t)c0[0] + ((uint32_t)c0[1] << W)) * MINV) & KMASK, k1 = (((uint32_t)c1[0] + ((uint32_t)c1[1] << W)) * MINV) & KMASK;
#pragma unroll
          for (int i = 0; i < NL; ++i) { c0[i] += (int64_t)k0 * (int32_t)P28.l[i]; c1[i] += (int64_t)k1 * (int32_t)P28.l[i]; }
          carry_shr(c0, nu); carry_shr(c1, nv); }
#pragma unroll
        for (int i = 0; i < NL; ++i) { u[i] = nu[i]; v[i] = nv[i]; }
    }
    // v = y^-1 with |v| < 64 p: add 64 p, bring into [0, 2p) with one quotient estimate, then canonical
    Fqn r;
    { constexpr Limbs B64 = times_p(64);
      int64_t c = 0; uint32_t l[NL];
#pragma unroll
      for (int i = 0; i < NL - 1; ++i) { const int64_t t = (int64_t)v[i] + (int32_t)B64.l[i] + c; l[i] = (uint32_t)t & MASK; c = t >> W; }
      l[NL - 1] = (uint32_t)((int64_t)v[NL - 1] + (int32_t)B64.l[NL - 1] + c);
      constexpr float INV = (1.0f - 1.0f / 1048576.0f) / (float)(P_TOP + 1);
      const int q = (int)((float)l[NL - 1] * INV);
      c = 0;
#pragma unroll
      for (int i = 0; i < NL - 1; ++i) { const int64_t t = (int64_t)l[i] - (int64_t)q * (int32_t)P28.l[i] + c; r.l[i] = (uint32_t)t & MASK; c = t >> W; }
      r.l[NL - 1] = (uint32_t)((int64_t)l[NL - 1] - (int64_t)q * (int32_t)P28.l[NL - 1] + c); }
    const Fqn cn = fq_canon(r);
    Fp x; fq_pack(cn, x.l);
    Fp fix;
#pragma unroll
    for (int i = 0; i < 12; ++i) fix.l[i] = KALISKI_FIX[0][i];           // R^3: plain inverse of a R  ->  Montgomery form of a^-1
    Fp out = mul(x, fix);
    if (y.is_zero()) out = Fp::zero();
    return out;
}
#endif
// finv: the inversion the kernels use -- the binary GCD above on the device (RIPP_INV_KALISKI: the bit-serial form), Fermat (fp.hpp) on the host
#if defined(__HIP_DEVICE_COMPILE__) && defined(RIPP_INV_KALISKI)
RIPP_HD Fp finv(const Fp& a) { return fp_inv_kaliski(a); }
#elif defined(__HIP_DEVICE_COMPILE__)
RIPP_HD Fp finv(const Fp& a) { return fp_inv_bingcd(a); }
#else
RIPP_HD Fp finv(const Fp& a) { return inv(a); }
#endif
}  // namespace ripp
