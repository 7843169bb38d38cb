// BLS12-381 prime fields Fp (381 bit, 12 x u32) and Fr (255 bit, 8 x u32), Montgomery form.
//
// Shared by the HIP kernels (device) and the host-side driver (final exponentiation,
// Fiat-Shamir scalars).  Limbs are little-endian 32-bit words, so the in-memory image is
// identical to the 6 x u64 / 4 x u64 little-endian Montgomery limbs that arkworks' Fp384 /
// Fp256 hold (ark-ff 0.4 `BigInt<N>([u64; N])`) and that the C ABI (include/ripp_hip.h) uses.
//
// gfx950 notes: 32 x 32 -> 64 multiply-add is `v_mad_u64_u32` (the only wide integer multiply on
// the VALU); everything here is written so that hipcc emits exactly one of those per limb product
// and add-with-carry chains (`v_add_co_u32` / `v_addc_co_u32`) for the rest.
#pragma once
#include <stdint.h>
#if defined(RIPP_BLS12_377)
// BLS12-377 build (libripp_hip_377.so; the curve of the reference's own SIPP test, sipp/src/lib.rs:229): the same 12 x u32 / 8 x u32 limb
// layouts and code, other constants (GLV / GLS endomorphism constants included: tools/gen_params.py derives and checks them).
#include "../bls12_377/params.hpp"
#else
#include "params.hpp"
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RIPP_HD __host__ __device__ __forceinline__
// Heavy primitives (Fp2 multiply/square, Fp inversion, ...) are REAL functions on host and device: a 12x12-limb
// Montgomery product is ~600 VALU instructions, so inlining the tower makes kernels of 10^5 instructions that
// neither fit the instruction cache nor compile in reasonable time.  Everything above them inlines into
// sequences of calls + carry-chain adds.
#define RIPP_FN __host__ __device__ inline __attribute__((noinline))
// Mid-level compositions (Fp6/Fp12 products, group law): inlined into kernels on the device so operands stay in
// VGPRs, but ordinary out-of-line functions in the host pass (keeps host compile time and code size sane).
#if defined(__HIP_DEVICE_COMPILE__)
#define RIPP_MID __host__ __device__ __forceinline__
#else
#define RIPP_MID __host__ __device__ inline __attribute__((noinline))
#endif
#else
#define RIPP_HD inline __attribute__((always_inline))
#define RIPP_FN inline __attribute__((noinline))
#define RIPP_MID inline __attribute__((noinline))
#endif

namespace ripp {

// ---------------------------------------------------------------- limb helpers
RIPP_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) {
    uint32_t co;
    uint32_t r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
RIPP_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) {
    uint32_t bo;
    uint32_t r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}
// (carry, lo) = a*b + t + c   -- never overflows 64 bits
RIPP_HD uint32_t mac32(uint32_t a, uint32_t b, uint32_t t, uint32_t& c) {
    uint64_t s = (uint64_t)a * b + t + c;
    c = (uint32_t)(s >> 32);
    return (uint32_t)s;
}

// ---------------------------------------------------------------- field parameter packs
struct FpParams {
    static constexpr int N = 12;
    static constexpr uint32_t INV = RIPP_FP_INV;
    RIPP_HD static constexpr uint32_t mod(int i)  { constexpr uint32_t v[12] = RIPP_FP_P;  return v[i]; }
    RIPP_HD static constexpr uint32_t one(int i)  { constexpr uint32_t v[12] = RIPP_FP_R1; return v[i]; }
    RIPP_HD static constexpr uint32_t r2(int i)   { constexpr uint32_t v[12] = RIPP_FP_R2; return v[i]; }
    RIPP_HD static constexpr uint32_t pm2(int i)  { constexpr uint32_t v[12] = RIPP_FP_P_MINUS_2; return v[i]; }
#if defined(RIPP_BLS12_377)
    static constexpr int BITS = 377;
#else
    static constexpr int BITS = 381;
#endif
};
struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t INV = RIPP_FR_INV;
    RIPP_HD static constexpr uint32_t mod(int i)  { constexpr uint32_t v[8] = RIPP_FR_R;  return v[i]; }
    RIPP_HD static constexpr uint32_t one(int i)  { constexpr uint32_t v[8] = RIPP_FR_R1; return v[i]; }
    RIPP_HD static constexpr uint32_t r2(int i)   { constexpr uint32_t v[8] = RIPP_FR_R2; return v[i]; }
    RIPP_HD static constexpr uint32_t pm2(int i)  { constexpr uint32_t v[8] = RIPP_FR_R_MINUS_2; return v[i]; }
#if defined(RIPP_BLS12_377)
    static constexpr int BITS = 253;
#else
    static constexpr int BITS = 255;
#endif
};

// ---------------------------------------------------------------- generic Montgomery field
template <class P>
struct alignas(16) Mont {   // 16-byte aligned so device loads/stores are dwordx4
    static constexpr int N = P::N;
    uint32_t l[N];

    RIPP_HD static Mont zero() { Mont r; for (int i = 0; i < N; ++i) r.l[i] = 0; return r; }
    RIPP_HD static Mont one()  { Mont r; for (int i = 0; i < N; ++i) r.l[i] = P::one(i); return r; }
    RIPP_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < N; ++i) o |= l[i]; return o == 0; }
    RIPP_HD bool operator==(const Mont& b) const { uint32_t o = 0; for (int i = 0; i < N; ++i) o |= l[i] ^ b.l[i]; return o == 0; }
    RIPP_HD bool operator!=(const Mont& b) const { return !(*this == b); }
};

#if !defined(__HIP_DEVICE_COMPILE__)
// Host forms: the same little-endian image read as N/2 64-bit limbs (the per-round final exponentiations and the verifiers'
// GT / group exponentiations run on the host and sit on the prover's critical path).
template <class P> RIPP_HD void h_load(const Mont<P>& a, uint64_t* o) { for (int i = 0; i < P::N / 2; ++i) o[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32); }
template <class P> RIPP_HD void h_store(Mont<P>& r, const uint64_t* t) { for (int i = 0; i < P::N / 2; ++i) { r.l[2 * i] = (uint32_t)t[i]; r.l[2 * i + 1] = (uint32_t)(t[i] >> 32); } }
template <class P> RIPP_HD uint64_t h_mod(int i) { return (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32); }
template <class P> RIPP_HD void h_reduce_once(uint64_t* t) {       // t < 2p  ->  t mod p
    constexpr int M = P::N / 2; typedef unsigned __int128 u128;
    uint64_t d[M], bo = 0;
    for (int i = 0; i < M; ++i) { const u128 x = (u128)t[i] - h_mod<P>(i) - bo; d[i] = (uint64_t)x; bo = (uint64_t)(x >> 64) & 1; }
    for (int i = 0; i < M; ++i) t[i] = bo ? t[i] : d[i];
}
#endif

// r = (a >= p) ? a - p : a        (a < 2p)
template <class P>
RIPP_HD void reduce_once(Mont<P>& a) {
    constexpr int N = P::N;
    uint32_t d[N], bo = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = subb32(a.l[i], P::mod(i), bo);
#pragma unroll
    for (int i = 0; i < N; ++i) a.l[i] = bo ? a.l[i] : d[i];
}

template <class P>
RIPP_HD Mont<P> add(const Mont<P>& a, const Mont<P>& b) {
    constexpr int N = P::N;
#if !defined(__HIP_DEVICE_COMPILE__)
    { constexpr int M = N / 2; typedef unsigned __int128 u128; uint64_t x[M], y[M], c = 0; h_load(a, x); h_load(b, y);
      for (int i = 0; i < M; ++i) { const u128 z = (u128)x[i] + y[i] + c; x[i] = (uint64_t)z; c = (uint64_t)(z >> 64); }
      h_reduce_once<P>(x); Mont<P> r; h_store(r, x); return r; }
#endif
    Mont<P> r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = addc32(a.l[i], b.l[i], c);
    reduce_once(r);   // 2p < 2^(32N): no carry out of the top limb for both fields
    return r;
}
template <class P>
RIPP_HD Mont<P> sub(const Mont<P>& a, const Mont<P>& b) {
    constexpr int N = P::N;
#if !defined(__HIP_DEVICE_COMPILE__)
    { constexpr int M = N / 2; typedef unsigned __int128 u128; uint64_t x[M], y[M], bo = 0; h_load(a, x); h_load(b, y);
      for (int i = 0; i < M; ++i) { const u128 z = (u128)x[i] - y[i] - bo; x[i] = (uint64_t)z; bo = (uint64_t)(z >> 64) & 1; }
      const uint64_t mask = 0ull - bo; uint64_t c = 0;
      for (int i = 0; i < M; ++i) { const u128 z = (u128)x[i] + (h_mod<P>(i) & mask) + c; x[i] = (uint64_t)z; c = (uint64_t)(z >> 64); }
      Mont<P> r; h_store(r, x); return r; }
#endif
    Mont<P> r; uint32_t bo = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = subb32(a.l[i], b.l[i], bo);
    uint32_t mask = 0u - bo, c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = addc32(r.l[i], P::mod(i) & mask, c);
    return r;
}
template <class P>
RIPP_HD Mont<P> neg(const Mont<P>& a) {
    constexpr int N = P::N;
    Mont<P> r; uint32_t bo = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) { r.l[i] = subb32(P::mod(i), a.l[i], bo); nz |= a.l[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = nz ? r.l[i] : 0u;
    return r;
}
template <class P>
RIPP_HD Mont<P> dbl(const Mont<P>& a) { return add(a, a); }

// Montgomery product, coarsely-integrated operand scanning (CIOS).  Inputs < p, output < p.
// Portable form (host, and the reference the device form is tested against).
template <class P>
RIPP_HD Mont<P> mul_cios(const Mont<P>& a, const Mont<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint32_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < N; ++j) t[j] = mac32(a.l[j], bi, t[j], c);
        t[N] += c;                           // running value < 2^33 * p < 2^(32(N+1)): no overflow
        const uint32_t m = t[0] * P::INV;
        uint32_t cc = 0;
        (void)mac32(m, P::mod(0), t[0], cc);
#pragma unroll
        for (int j = 1; j < N; ++j) t[j - 1] = mac32(m, P::mod(j), t[j], cc);
        t[N - 1] = t[N] + cc;                // shifted value < 2p < 2^(32N): fits N limbs
        t[N] = 0;
    }
    Mont<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = t[i];
    reduce_once(r);                            // p < 2^(32N-2) for both fields => t < 2p
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form: finely-integrated product scanning.  Measured on MI355X (profiles/r01_ubench_valu_rates.txt):
// v_mad_u64_u32 3.85 cyc/wave, v_addc_co_u32 3.85, v_lshl_add_u64 3.6, v_mov/v_add_u32 2.  hipcc's CIOS spends
// 2/3 of its cycles zero-extending limbs into 64-bit pairs (v_mov + v_lshl_add_u64).  Here every limb product is
// ONE v_mad_u64_u32 accumulating into a 96-bit column accumulator {c2:acc}, with the MAD's carry-out (VCC)
// captured by ONE v_addc_co_u32 -- 2 VALU ops per limb product, no zero-extension.
RIPP_HD void madc96(uint64_t& acc, uint32_t& c2, uint32_t x, uint32_t y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(x), "v"(y) : "vcc");
}
RIPP_HD void madc96_s(uint64_t& acc, uint32_t& c2, uint32_t x, uint32_t y_const) {   // y in an SGPR (modulus limb)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(x), "s"(y_const) : "vcc");
}
// two limb products per asm statement: hipcc puts one s_nop behind EVERY inline-asm statement, so pairing them halves those issue slots
RIPP_HD void madc96_2(uint64_t& acc, uint32_t& c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
// sum_{i = LO}^{HI} a_i b_(K-i), pairwise
template <int LO, int HI, int K, class P>
RIPP_HD void madc96_row(uint64_t& acc, uint32_t& c2, const Mont<P>& a, const Mont<P>& b) {
#pragma unroll
    for (int i = LO; i + 1 <= HI; i += 2) madc96_2(acc, c2, a.l[i], b.l[K - i], a.l[i + 1], b.l[K - i - 1]);
    if constexpr (((HI - LO + 1) & 1) != 0) madc96(acc, c2, a.l[HI], b.l[K - HI]);
}
template <int K, int N, class P> RIPP_HD void mul_cols_lo(uint64_t& acc, uint32_t& c2, const Mont<P>& a, const Mont<P>& b, uint32_t (&m)[N]) {
    if constexpr (K < N) {
        madc96_row<0, K, K>(acc, c2, a, b);
#pragma unroll
        for (int i = 0; i < K; ++i) madc96_s(acc, c2, m[i], P::mod(K - i));
        m[K] = (uint32_t)acc * P::INV;
        madc96_s(acc, c2, m[K], P::mod(0));                 // low word becomes 0
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
        mul_cols_lo<K + 1, N>(acc, c2, a, b, m);
    }
}
template <int K, int N, class P> RIPP_HD void mul_cols_hi(uint64_t& acc, uint32_t& c2, const Mont<P>& a, const Mont<P>& b, const uint32_t (&m)[N], Mont<P>& r) {
    if constexpr (K < 2 * N - 1) {
        madc96_row<K - N + 1, N - 1, K>(acc, c2, a, b);
#pragma unroll
        for (int i = K - N + 1; i < N; ++i) madc96_s(acc, c2, m[i], P::mod(K - i));
        r.l[K - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
        mul_cols_hi<K + 1, N>(acc, c2, a, b, m, r);
    }
}
template <class P>
RIPP_HD Mont<P> mul(const Mont<P>& a, const Mont<P>& b) {
    constexpr int N = P::N;
    uint32_t m[N];
    Mont<P> r;
    uint64_t acc = 0; uint32_t c2 = 0;
    mul_cols_lo<0, N>(acc, c2, a, b, m);
    mul_cols_hi<N, N>(acc, c2, a, b, m, r);
    r.l[N - 1] = (uint32_t)acc;                             // result < 2p < 2^(32N): acc >> 32 == 0
    reduce_once(r);
    return r;
}
template <class P>
RIPP_HD Mont<P> mul_single(const Mont<P>& a, const Mont<P>& b) {      // one asm statement per limb product (the form before the pairing; A/B reference)
    constexpr int N = P::N;
    uint32_t m[N];
    Mont<P> r;
    uint64_t acc = 0; uint32_t c2 = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) madc96(acc, c2, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) madc96_s(acc, c2, m[i], P::mod(k - i));
        m[k] = (uint32_t)acc * P::INV;
        madc96_s(acc, c2, m[k], P::mod(0));                 // low word becomes 0
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) madc96(acc, c2, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) madc96_s(acc, c2, m[i], P::mod(k - i));
        r.l[k - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
    r.l[N - 1] = (uint32_t)acc;                             // result < 2p < 2^(32N): acc >> 32 == 0
    reduce_once(r);
    return r;
}
// (a b + c d) R^-1 mod p with ONE Montgomery reduction: the limb products of both pairs go into the same column accumulator
// (a b + c d < 2 p^2 < p R, so the reduced value is < 2p and one conditional subtraction finishes it).  3/4 of the work of two
// multiplications; Fp2 products are two of these (tower.hpp) instead of three multiplications and five additions.
template <class P>
RIPP_HD Mont<P> mul2_add(const Mont<P>& a, const Mont<P>& b, const Mont<P>& c, const Mont<P>& d) {
    constexpr int N = P::N;
    uint32_t m[N];
    Mont<P> r;
    uint64_t acc = 0; uint32_t c2 = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) madc96_2(acc, c2, a.l[i], b.l[k - i], c.l[i], d.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) madc96_s(acc, c2, m[i], P::mod(k - i));
        m[k] = (uint32_t)acc * P::INV;
        madc96_s(acc, c2, m[k], P::mod(0));
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) madc96_2(acc, c2, a.l[i], b.l[k - i], c.l[i], d.l[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) madc96_s(acc, c2, m[i], P::mod(k - i));
        r.l[k - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
    r.l[N - 1] = (uint32_t)acc;
    reduce_once(r);
    return r;
}
#else
// Host: the same little-endian image read as N/2 64-bit limbs, CIOS with unsigned __int128 (mulx/adx) -- ~3x the
// speed of the 32-bit portable form; used by the per-round final exponentiations on the critical path.
template <class P>
RIPP_HD Mont<P> mul(const Mont<P>& a, const Mont<P>& b) {
    // CIOS with the two carry chains (a_j b_i and m p_j) kept separate: they are independent, so the out-of-order core overlaps
    // them; valid without an extra carry word because the top modulus limb is < 2^63 for both fields (p: 0x1a01..., r: 0x73ed...).
    constexpr int M = P::N / 2;
    typedef unsigned __int128 u128;
    uint64_t al[M], bl[M], pl[M], t[M];
    h_load(a, al); h_load(b, bl);
    for (int i = 0; i < M; ++i) { pl[i] = h_mod<P>(i); t[i] = 0; }
    static_assert((P::mod(P::N - 1) >> 31) == 0, "no-carry CIOS needs the top bit of the modulus clear");
    const uint64_t ninv32 = P::INV;
    const uint64_t inv64 = ninv32 * (2 + pl[0] * ninv32);       // -p^-1 mod 2^64 from the 32-bit constant (one Newton step)
    for (int i = 0; i < M; ++i) {
        u128 z = (u128)al[0] * bl[i] + t[0];
        uint64_t C = (uint64_t)(z >> 64); const uint64_t t0 = (uint64_t)z;
        const uint64_t m = t0 * inv64;
        z = (u128)m * pl[0] + t0;
        uint64_t C2 = (uint64_t)(z >> 64);
        for (int j = 1; j < M; ++j) {
            z = (u128)al[j] * bl[i] + t[j] + C; C = (uint64_t)(z >> 64);
            z = (u128)m * pl[j] + (uint64_t)z + C2; C2 = (uint64_t)(z >> 64); t[j - 1] = (uint64_t)z;
        }
        t[M - 1] = C + C2;
    }
    h_reduce_once<P>(t);
    Mont<P> r; h_store(r, t);
    return r;
}
#endif

template <class P>
RIPP_HD Mont<P> sqr(const Mont<P>& a) { return mul(a, a); }

// a^e for a plain-integer exponent given as little-endian limbs (square-and-multiply, MSB first)
template <class P, int EN>
RIPP_FN Mont<P> pow_limbs(const Mont<P>& a, const uint32_t (&e)[EN]) {
    Mont<P> r = Mont<P>::one();
    bool started = false;
    for (int i = EN * 32 - 1; i >= 0; --i) {
        if (started) r = sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1u) { r = started ? mul(r, a) : a; started = true; }
    }
    return r;
}
// Fermat inversion a^(p-2); returns 0 for a == 0 (callers check).
template <class P>
RIPP_FN Mont<P> inv(const Mont<P>& a) {
    uint32_t e[P::N];
    for (int i = 0; i < P::N; ++i) e[i] = P::pm2(i);
    return pow_limbs<P, P::N>(a, e);
}

// to / from Montgomery form
template <class P>
RIPP_HD Mont<P> to_mont(const Mont<P>& a) { Mont<P> r2; for (int i = 0; i < P::N; ++i) r2.l[i] = P::r2(i); return mul(a, r2); }
template <class P>
RIPP_HD Mont<P> from_mont(const Mont<P>& a) { Mont<P> o = Mont<P>::zero(); o.l[0] = 1; return mul(a, o); }

using Fp = Mont<FpParams>;
using Fr = Mont<FrParams>;

// Out-of-line Fp multiply / square: THE call boundary of the device code.  hipcc passes aggregates of more than
// 16 dwords through scratch memory, so the device entry takes the two operands as six 4-dword vectors (24 VGPRs
// in, 12 VGPRs out, nothing on the stack); everything above (Fp2/Fp6/Fp12, group law) inlines into call sequences.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t ripp_v4u __attribute__((ext_vector_type(4)));
struct FpRegs { ripp_v4u v0, v1, v2; };
static_assert(sizeof(FpRegs) == sizeof(Fp), "layout");
__device__ __noinline__ inline FpRegs fp_mul_call(ripp_v4u a0, ripp_v4u a1, ripp_v4u a2, ripp_v4u b0, ripp_v4u b1, ripp_v4u b2) {
    Fp a, b;
    ripp_v4u* pa = reinterpret_cast<ripp_v4u*>(a.l); pa[0] = a0; pa[1] = a1; pa[2] = a2;
    ripp_v4u* pb = reinterpret_cast<ripp_v4u*>(b.l); pb[0] = b0; pb[1] = b1; pb[2] = b2;
    const Fp r = mul(a, b);
    const ripp_v4u* pr = reinterpret_cast<const ripp_v4u*>(r.l);
    return {pr[0], pr[1], pr[2]};
}
RIPP_HD Fp fmul(const Fp& a, const Fp& b) {
    const ripp_v4u* pa = reinterpret_cast<const ripp_v4u*>(a.l);
    const ripp_v4u* pb = reinterpret_cast<const ripp_v4u*>(b.l);
    const FpRegs r = fp_mul_call(pa[0], pa[1], pa[2], pb[0], pb[1], pb[2]);
    Fp o; ripp_v4u* po = reinterpret_cast<ripp_v4u*>(o.l); po[0] = r.v0; po[1] = r.v1; po[2] = r.v2;
    return o;
}
RIPP_HD Fp fsqr(const Fp& a) { return fmul(a, a); }
__device__ __noinline__ inline FpRegs fp_mul2_call(ripp_v4u a0, ripp_v4u a1, ripp_v4u a2, ripp_v4u b0, ripp_v4u b1, ripp_v4u b2,
                                                   ripp_v4u c0, ripp_v4u c1, ripp_v4u c2, ripp_v4u d0, ripp_v4u d1, ripp_v4u d2) {
    Fp a, b, c, d;
    ripp_v4u* pa = reinterpret_cast<ripp_v4u*>(a.l); pa[0] = a0; pa[1] = a1; pa[2] = a2;
    ripp_v4u* pb = reinterpret_cast<ripp_v4u*>(b.l); pb[0] = b0; pb[1] = b1; pb[2] = b2;
    ripp_v4u* pc = reinterpret_cast<ripp_v4u*>(c.l); pc[0] = c0; pc[1] = c1; pc[2] = c2;
    ripp_v4u* pd = reinterpret_cast<ripp_v4u*>(d.l); pd[0] = d0; pd[1] = d1; pd[2] = d2;
    const Fp r = mul2_add(a, b, c, d);
    const ripp_v4u* pr = reinterpret_cast<const ripp_v4u*>(r.l);
    return {pr[0], pr[1], pr[2]};
}
RIPP_HD Fp fmul2_add(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {          // a b + c d
    const ripp_v4u* pa = reinterpret_cast<const ripp_v4u*>(a.l); const ripp_v4u* pb = reinterpret_cast<const ripp_v4u*>(b.l);
    const ripp_v4u* pc = reinterpret_cast<const ripp_v4u*>(c.l); const ripp_v4u* pd = reinterpret_cast<const ripp_v4u*>(d.l);
    const FpRegs r = fp_mul2_call(pa[0], pa[1], pa[2], pb[0], pb[1], pb[2], pc[0], pc[1], pc[2], pd[0], pd[1], pd[2]);
    Fp o; ripp_v4u* po = reinterpret_cast<ripp_v4u*>(o.l); po[0] = r.v0; po[1] = r.v1; po[2] = r.v2;
    return o;
}
#else
RIPP_FN Fp fmul(const Fp& a, const Fp& b) { return mul(a, b); }
RIPP_FN Fp fsqr(const Fp& a) { return mul(a, a); }
#endif
RIPP_FN Fr fmul(const Fr& a, const Fr& b) { return mul(a, b); }

RIPP_HD Fp fp_const(const uint32_t (&v)[12]) { Fp r; for (int i = 0; i < 12; ++i) r.l[i] = v[i]; return r; }

}  // namespace ripp
