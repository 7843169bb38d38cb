// Fq28: the base field in a CARRY-FREE radix for gfx950 -- 14 limbs of 28 bits, Montgomery form with R' = 2^392.
//
// Why: the only wide integer multiply of the CDNA VALU is v_mad_u64_u32 (32 x 32 + 64 -> 64), and there is no multiply-add WITH carry.
// With 32-bit limbs (fp.hpp) every limb product therefore costs TWO half-rate instructions: the MAD and a v_addc_co_u32 that captures
// its carry-out into a third accumulator word (578 instructions per Montgomery product).  With 28-bit limbs a whole column
//     col_k = sum_{i+j=k} a_i b_j  (+ sum m_i p_j)     <=  (NT + 1) * 14 * 2^56  <  2^64   for NT <= 16 products
// fits the MAD's own 64-bit accumulator: ONE instruction per limb product, 196 + 196 per Montgomery product instead of 288 + 288,
// and a lazily reduced sum of NT products (line_products.hpp) costs 196 NT + 196 instead of 288 NT + 288.  Additions and subtractions
// need no carry chain at all (limb-wise, full-rate v_add_u32; limbs may grow to 32 bits before they are normalised).
//
// Laziness is checked AT COMPILE TIME: Fq<LB, VB> carries the bound of its limbs (< 2^LB) and of its value (< VB * p) in its type;
// mul / dot static_assert that the column sums fit 64 bits and that the Montgomery result stays below 2p.  A bound violation is a
// compile error, not a wrong result on rare inputs.
//
// Storage format ("q-form"): 12 x u32 holding the INTEGER of a Montgomery-392 value < 2^384 (results of mul are < 2p < 2^382).
// Conversion from / to the engine's Mont-384 values (fp.hpp, the C ABI's limbs) is one Montgomery product each way.
#pragma once
#include "bls12_381/fp.hpp"

namespace ripp {

#if defined(__HIP_DEVICE_COMPILE__)

namespace fq28 {
constexpr int NL = 14, W = 28;
constexpr uint32_t MASK = (1u << W) - 1u;

struct Big12 { uint32_t l[12]; };
constexpr Big12 big_p() { Big12 r{}; constexpr uint32_t v[12] = RIPP_FP_P; for (int i = 0; i < 12; ++i) r.l[i] = v[i]; return r; }
constexpr Big12 big_r1() { Big12 r{}; constexpr uint32_t v[12] = RIPP_FP_R1; for (int i = 0; i < 12; ++i) r.l[i] = v[i]; return r; }
constexpr bool big_geq(const Big12& a, const Big12& b) { for (int i = 11; i >= 0; --i) { if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; } return true; }
constexpr Big12 big_dbl_mod(const Big12& a) {       // 2a mod p for a < p (2p < 2^384)
    Big12 r{}; uint32_t c = 0;
    for (int i = 0; i < 12; ++i) { r.l[i] = (a.l[i] << 1) | c; c = a.l[i] >> 31; }
    const Big12 p = big_p();
    if (big_geq(r, p)) { uint64_t bo = 0; for (int i = 0; i < 12; ++i) { const uint64_t d = (uint64_t)r.l[i] - p.l[i] - bo; r.l[i] = (uint32_t)d; bo = (d >> 63) & 1u; } }
    return r;
}
constexpr Big12 big_pow2_mod(int e) { Big12 r = big_r1(); for (int i = 384; i < e; ++i) r = big_dbl_mod(r); return r; }     // 2^e mod p, e >= 384
// c * p as 14 limbs of 28 bits (normalised), c < 2^11
struct Limbs { uint32_t l[NL]; };
constexpr Limbs slice28(const Big12& a) {
    Limbs r{};
    for (int k = 0; k < NL; ++k) {
        const int bit = W * k, w = bit >> 5, s = bit & 31;
        uint64_t v = w < 12 ? a.l[w] : 0u; if (w + 1 < 12) v |= (uint64_t)a.l[w + 1] << 32;
        r.l[k] = (uint32_t)(v >> s) & MASK;
    }
    return r;
}
constexpr Limbs P28 = slice28(big_p());
constexpr uint32_t INV28 = RIPP_FP_INV & MASK;          // -p^-1 mod 2^28
constexpr Limbs times_p(uint32_t c) {
    Limbs r{}; uint64_t carry = 0;
    for (int k = 0; k < NL; ++k) { const uint64_t t = (uint64_t)P28.l[k] * c + carry; r.l[k] = (uint32_t)t & MASK; carry = t >> W; }
    r.l[NL - 1] += (uint32_t)(carry << W);          // c p < 2^392 for c < 2^11: nothing left
    return r;
}
constexpr Limbs ONE_M392 = slice28(big_pow2_mod(392));      // Montgomery one
constexpr Limbs C_IN = slice28(big_pow2_mod(400));          // mont(x, 2^400) = x * 2^8      : Mont-384 -> Mont-392
constexpr Limbs C_OUT = slice28(big_r1());                  // mont(x~, 2^384) = x~ * 2^-8   : Mont-392 -> Mont-384
// a Montgomery-384 constant of the engine (12 words, canonical) -> its Montgomery-392 limbs, at compile time: times 2^8 mod p
constexpr Limbs from_mont384(const uint32_t (&v)[12]) { Big12 r{}; for (int i = 0; i < 12; ++i) r.l[i] = v[i]; for (int i = 0; i < 8; ++i) r = big_dbl_mod(r); return slice28(r); }
constexpr uint32_t P_TOP = P28.l[NL - 1];                   // p >> 364
constexpr int VMAX = 2500;                                  // R' / p = 2519.6: products with V1 * V2 <= VMAX reduce to < 2p
}  // namespace fq28

// every limb < LM (an exact exclusive bound, not a bit count), value < VB * p
constexpr uint64_t FQ_LN = (uint64_t)1 << 28;      // the limb bound of a normalised value
template <uint64_t LM = FQ_LN, int VB = 2>
struct Fq {
    static_assert(LM >= 1 && LM <= ((uint64_t)1 << 32), "limb bound");
    static_assert(VB >= 1 && VB <= fq28::VMAX, "value bound");
    static constexpr uint64_t LMAX = LM;
    static constexpr int VMAXB = VB;
    uint32_t l[fq28::NL];
};
using Fqn = Fq<FQ_LN, 2>;         // what every multiplication returns

__device__ __forceinline__ void mad64(uint64_t& acc, uint32_t x, uint32_t y) { acc += (uint64_t)x * y; }
// A column is ONE chain of multiply-adds seeded with the previous column's carry.  Left to itself LLVM re-associates the sum: it starts the chain at zero and adds the
// carry with a 64-bit add behind it (26 v_lshl_add_u64 per product) -- one more instruction per column for a dependence that costs nothing here: a dependent
// v_mad_u64_u32 issues at the rate of an independent one (profiles/r01_ubench_valu_rates.txt).  An empty asm on the running sum keeps the chain as written.
#ifndef FQ_CHAIN_TIES
#define FQ_CHAIN_TIES 2
#endif
#if FQ_CHAIN_TIES == 2
#define FQ_CHAIN(x) asm volatile("" :: "v"(x))
#elif FQ_CHAIN_TIES
#define FQ_CHAIN(x) asm("" : "+v"(x))
#else
#define FQ_CHAIN(x)
#endif
// a square as 105 limb products (the 91 cross terms once, against a doubled operand) instead of 196
#ifndef FQ_DEDICATED_SQR
#define FQ_DEDICATED_SQR 1
#endif

template <uint64_t LM, int VB> __device__ __forceinline__ Fq<LM, VB> fq_const(const fq28::Limbs& c) { Fq<LM, VB> r; for (int i = 0; i < fq28::NL; ++i) r.l[i] = c.l[i]; return r; }
__device__ __forceinline__ Fqn fq_zero() { Fqn r; for (int i = 0; i < fq28::NL; ++i) r.l[i] = 0; return r; }
__device__ __forceinline__ Fqn fq_one() { return fq_const<FQ_LN, 2>(fq28::ONE_M392); }
// forget precision of the bounds (e.g. to give the two arms of a select, or a loop-carried value, one type)
template <uint64_t LM2, int VB2, uint64_t LM, int VB>
__device__ __forceinline__ Fq<LM2, VB2> fq_widen(const Fq<LM, VB>& a) {
    static_assert(LM2 >= LM && VB2 >= VB, "widening only");
    Fq<LM2, VB2> r;
#pragma unroll
    for (int i = 0; i < fq28::NL; ++i) r.l[i] = a.l[i];
    return r;
}

// Pins a value where it is written: an empty asm over its 14 limbs.  The compiler is otherwise free to sink a product below later loads (or hoist
// loads above it) and then keeps BOTH alive -- __builtin_amdgcn_sched_barrier only binds the machine scheduler, the order is already lost in the
// DAG.  A data dependence through a volatile asm is honoured everywhere (fq_line_products.hpp: 48 spilled dwords -> 0 with this alone).
template <uint64_t LM, int VB> __device__ __forceinline__ void fq_pin(Fq<LM, VB>& a) {
    asm volatile("" : "+v"(a.l[0]), "+v"(a.l[1]), "+v"(a.l[2]), "+v"(a.l[3]), "+v"(a.l[4]), "+v"(a.l[5]), "+v"(a.l[6]), "+v"(a.l[7]), "+v"(a.l[8]), "+v"(a.l[9]), "+v"(a.l[10]), "+v"(a.l[11]), "+v"(a.l[12]), "+v"(a.l[13]));
}

namespace fq28 {
// column bound of a lazily reduced sum of NT products: NT * 14 * (L1-1)(L2-1) + 14 * 2^56 (the m p terms) + carry-in (< 2^37) < 2^64
constexpr bool dot_fits(int NT, uint64_t L1, uint64_t L2) {
    const long double col = (long double)NT * 14.0L * (long double)(L1 - 1) * (long double)(L2 - 1) + 14.0L * 72057594037927936.0L + 137438953472.0L;
    return col < 18446744073709551615.0L;
}
}  // namespace fq28

// The Montgomery product over columns: terms(k, acc) adds the limb products of column k (0 <= k < 27) of the integer to be reduced to acc; the m p terms of the
// reduction follow in the same chain.  Result limbs < 2^28, value < 2p when the integer is < VMAX p^2.
template <class TERMS>
__device__ __forceinline__ Fqn fq_montgomery(TERMS&& terms) {
    using namespace fq28;
    uint32_t m[NL];
    Fqn r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        uint64_t s = carry;
        terms(k, s);
#pragma unroll
        for (int i = 0; i < k; ++i) { mad64(s, m[i], P28.l[k - i]); FQ_CHAIN(s); }
        m[k] = ((uint32_t)s * INV28) & MASK;
        mad64(s, m[k], P28.l[0]);
        carry = s >> W;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
        uint64_t s = carry;
        terms(k, s);
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) { mad64(s, m[i], P28.l[k - i]); FQ_CHAIN(s); }
        r.l[k - NL] = (uint32_t)s & MASK;
        carry = s >> W;
    }
    r.l[NL - 1] = (uint32_t)carry;        // value < 2p < 2^382: the top limb is < 2^18
    return r;
}

// sum_t a[t] * b[t] * R'^-1 mod p with ONE Montgomery reduction; result limbs < 2^28, value < 2p
template <int NT, uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ Fqn fq_dot(const Fq<L1, V1> (&a)[NT], const Fq<L2, V2> (&b)[NT]) {
    using namespace fq28;
    static_assert(dot_fits(NT, L1, L2), "column sum overflows 64 bits: normalise an operand");
    static_assert((long)NT * V1 * V2 <= VMAX, "value bound: the sum of products must stay below p R'");
    return fq_montgomery([&](int k, uint64_t& s) __attribute__((always_inline)) {
        const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int i = lo; i <= hi; ++i) { mad64(s, a[t].l[i], b[t].l[k - i]); FQ_CHAIN(s); }
        }
    });
}
template <uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ Fqn fq_mul(const Fq<L1, V1>& a, const Fq<L2, V2>& b) {
    const Fq<L1, V1> aa[1] = {a}; const Fq<L2, V2> bb[1] = {b};
    return fq_dot<1>(aa, bb);
}
// a^2: column k = sum_(i < j, i + j = k) (2 a_i) a_j + [k even] a_(k/2)^2 -- the same integer as a * a, 105 limb products
template <uint64_t L1, int V1> __device__ __forceinline__ Fqn fq_sqr(const Fq<L1, V1>& a) {
#if FQ_DEDICATED_SQR
    using namespace fq28;
    static_assert(dot_fits(1, L1, L1), "column sum overflows 64 bits: normalise the operand");
    static_assert(L1 <= ((uint64_t)1 << 31), "the doubled limbs must fit 32 bits");
    static_assert((long)V1 * V1 <= VMAX, "value bound");
    uint32_t d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = a.l[i] << 1;
    return fq_montgomery([&](int k, uint64_t& s) __attribute__((always_inline)) {
        const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
#pragma unroll
        for (int i = lo; i <= hi; ++i) {
            const int j = k - i;
            if (i < j) { mad64(s, d[i], a.l[j]); FQ_CHAIN(s); }
            else if (i == j) { mad64(s, a.l[i], a.l[i]); FQ_CHAIN(s); }
        }
    });
#else
    return fq_mul(a, a);
#endif
}

// limb-wise, no carries
template <uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ Fq<L1 + L2 - 1, V1 + V2> fq_add(const Fq<L1, V1>& a, const Fq<L2, V2>& b) {
    Fq<L1 + L2 - 1, V1 + V2> r;
#pragma unroll
    for (int i = 0; i < fq28::NL; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <uint64_t L1, int V1> __device__ __forceinline__ Fq<2 * L1 - 1, 2 * V1> fq_dbl(const Fq<L1, V1>& a) { return fq_add(a, a); }
// a - b = a + (K - b), K = (V2 + 1) p written with every limb >= the corresponding limb of b:
//     K_k = n_k + B [k < 13] - (B >> 28) [k > 0],   B = the multiple of 2^28 above b's limb bound;
// b's TOP limb is bounded by its value (b_13 <= V2 p / 2^364), so the top limb of (V2 + 1) p minus the borrow covers it.
namespace fq28 {
constexpr uint64_t sub_B(uint64_t L2) { uint64_t B = ((L2 + MASK) >> W) << W; while (B - (B >> W) + 1 < L2) B += (uint64_t)1 << W; return B; }      // B - (B >> 28) >= L2 - 1
template <uint64_t L2, int V2> constexpr Limbs sub_bias() {
    Limbs n = times_p((uint32_t)V2 + 1u), r{};
    for (int k = 0; k < NL; ++k) {
        int64_t v = n.l[k];
        if (k < NL - 1) v += (int64_t)sub_B(L2);
        if (k > 0) v -= (int64_t)(sub_B(L2) >> W);
        r.l[k] = (uint32_t)v;
    }
    return r;
}
template <uint64_t L2, int V2> constexpr bool sub_bias_ok() {
    const Limbs n = times_p((uint32_t)V2 + 1u);
    return (int64_t)n.l[NL - 1] - (int64_t)(sub_B(L2) >> W) >= (int64_t)V2 * (P_TOP + 1);
}
constexpr uint64_t sub_lm(uint64_t L1, uint64_t L2) { return L1 + MASK + sub_B(L2); }      // a_k + (n_k + B - b_k)
}  // namespace fq28
template <uint64_t L1, int V1, uint64_t L2, int V2>
__device__ __forceinline__ Fq<fq28::sub_lm(L1, L2), V1 + V2 + 1> fq_sub(const Fq<L1, V1>& a, const Fq<L2, V2>& b) {
    static_assert(fq28::sub_bias_ok<L2, V2>(), "subtrahend too lazy");
    constexpr fq28::Limbs K = fq28::sub_bias<L2, V2>();
    Fq<fq28::sub_lm(L1, L2), V1 + V2 + 1> r;
#pragma unroll
    for (int i = 0; i < fq28::NL; ++i) r.l[i] = a.l[i] + (K.l[i] - b.l[i]);
    return r;
}
template <uint64_t L2, int V2> __device__ __forceinline__ auto fq_neg(const Fq<L2, V2>& b) { return fq_sub(fq_widen<1, 1>(Fq<1, 1>{}), b); }
// carry propagation: limbs < 2^28 again (the top limb keeps the rest), same value
template <uint64_t LM, int VB>
__device__ __forceinline__ Fq<FQ_LN, VB> fq_norm(const Fq<LM, VB>& a) {
    static_assert(LM <= ((uint64_t)1 << 32) - 16, "normalise before the limbs reach 32 bits");
    Fq<FQ_LN, VB> r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < fq28::NL - 1; ++i) { const uint32_t t = a.l[i] + c; r.l[i] = t & fq28::MASK; c = t >> fq28::W; }
    r.l[fq28::NL - 1] = a.l[fq28::NL - 1] + c;
    return r;
}

// ---- storage: 12 x u32 holding the integer of a normalised value < 2^384 ---------------------------------------------------
__device__ __forceinline__ Fqn fq_unpack(const uint32_t (&x)[12]) {
    using namespace fq28;
    Fqn r;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int bit = W * k, w = bit >> 5, s = bit & 31;
        uint32_t v;
        if (s + W <= 32 || w + 1 >= 12) v = x[w] >> s;
        else v = __builtin_amdgcn_alignbit(x[w + 1], x[w], s);
        r.l[k] = v & MASK;
    }
    return r;
}
// value must be < 2^384 with normalised limbs (any mul result is)
__device__ __forceinline__ void fq_pack(const Fqn& a, uint32_t (&x)[12]) {
    using namespace fq28;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int bit = 32 * j, k = bit / W, o = bit - k * W;       // word j starts o bits into limb k
        uint32_t v = a.l[k] >> o;
        if (k + 1 < NL) v |= a.l[k + 1] << (W - o);
        x[j] = v;
    }
}
// canonical representative (< p) of a value < 2p with normalised limbs
__device__ __forceinline__ Fqn fq_canon(const Fqn& a) {
    using namespace fq28;
    uint32_t d[NL]; uint32_t bo = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) { const uint32_t t = a.l[i] - P28.l[i] - bo; bo = t >> 31; d[i] = (i < NL - 1) ? (t & MASK) : t; }
    Fqn r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = bo ? a.l[i] : d[i];
    return r;
}
// x * 2^8 as limbs: the re-slicing of fq_unpack with an 8-bit offset (value < 256 p)
__device__ __forceinline__ Fq<FQ_LN, 256> fq_unpack_shl8(const uint32_t (&x)[12]) {
    using namespace fq28;
    Fq<FQ_LN, 256> r;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int bit = W * k - 8;                                   // limb k = bits [28k - 8, 28k + 20) of x
        uint32_t v;
        if (bit < 0) v = x[0] << 8;
        else { const int w = bit >> 5, s = bit & 31;
               if (s + W <= 32 || w + 1 >= 12) v = x[w] >> s; else v = __builtin_amdgcn_alignbit(x[w + 1], x[w], s); }
        r.l[k] = v & MASK;
    }
    return r;
}
// engine value (Mont-384, canonical) <-> Fq (Mont-392)
__device__ __forceinline__ Fqn fq_from_fp(const Fp& x) { const Fqn t = fq_unpack(x.l); return fq_mul(t, fq_const<FQ_LN, 1>(fq28::C_IN)); }
__device__ __forceinline__ Fp fq_to_fp(const Fqn& a) { const Fqn t = fq_canon(fq_mul(a, fq_const<FQ_LN, 1>(fq28::C_OUT))); Fp r; fq_pack(t, r.l); return r; }

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace ripp
