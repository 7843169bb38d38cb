// Wire format of the proof structs: `CanonicalSerialize` / `CanonicalDeserialize` images as
// ark-serialize 0.4 derives them for
//   GIPAProof        ip_proofs/src/gipa.rs:24-51        { r_commitment_steps: Vec<((L,R,I),(L,R,I))>, r_base: (LMsg, RMsg) }
//   TIPAProof        ip_proofs/src/tipa/mod.rs:41-65    { gipa_proof, final_ck: (G2, G1), final_ck_proof: (G2, G1) }
//   TIPAWithSSMProof tipa/structured_scalar_message.rs:138-156  { gipa_proof, final_ck: G2, final_ck_proof: G2 }
// Field order = declaration order; Vec<T> = u64-LE length + items; tuples = items in order; PhantomData = nothing;
// IdentityOutput<T>(Vec<T>) (dh_commitments/src/identity/mod.rs:32-35) = a Vec; group elements are written as AFFINE points in the
// zcash layout of ark-bls12-381 0.4 (big-endian coordinates, c1 before c0, flag bits 7/6/5 = compressed / infinity / y is the
// lexicographically larger root) -- [ark-mem]: restated from the published crate, not vendored in /root/reference; the BLS12-377 build
// (RIPP_BLS12_377) writes ark-ec's GENERIC short-Weierstrass layout instead (little-endian x [, y], SWFlags in the two top bits of the LAST
// byte: bit 7 = y is the larger root, bit 6 = infinity; compressed = x alone) and takes its square roots by Tonelli-Shanks (p = 1 mod 2^46);
// GT (Fq12) and Fr have no compressed form.  The step vector is stored REVERSED (gipa.rs:298-299); the C ABI takes ROUND order.
// Host-only code: proofs are a few KB.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "host_fs.hpp"

namespace ripp { namespace wire {

inline bool fp_gt(const Fp& a_mont, const Fp& b_mont) {          // canonical integers a > b
    const Fp a = from_mont(a_mont), b = from_mont(b_mont);
    for (int i = 11; i >= 0; --i) { if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; }
    return false;
}
inline bool lex_largest(const Fp& y) { return fp_gt(y, neg(y)); }                                   // `p.y > -p.y`
inline bool lex_largest(const Fp2& y) { return y.c1.is_zero() ? lex_largest(y.c0) : lex_largest(y.c1); }   // QuadExtField orders by c1, then c0

inline size_t g1_size(bool compress) { return compress ? 48 : 96; }
inline size_t g2_size(bool compress) { return compress ? 96 : 192; }
#if defined(RIPP_BLS12_377)
inline void put_g1(const G1A& p, bool compress, uint8_t* out) {
    if (!compress) { fs::ser_g1(p, out); return; }
    if (is_inf(p)) { std::memset(out, 0, 48); out[47] = 0x40; return; }
    fs::ser_fp_le(p.x, out); if (lex_largest(p.y)) out[47] |= 0x80;
}
inline void put_g2(const G2A& p, bool compress, uint8_t* out) {
    if (!compress) { fs::ser_g2(p, out); return; }
    if (is_inf(p)) { std::memset(out, 0, 96); out[95] = 0x40; return; }
    fs::ser_fp_le(p.x.c0, out); fs::ser_fp_le(p.x.c1, out + 48); if (lex_largest(p.y)) out[95] |= 0x80;
}
#else
inline void put_g1(const G1A& p, bool compress, uint8_t* out) {
    if (!compress) { fs::ser_g1(p, out); return; }
    if (is_inf(p)) { std::memset(out, 0, 48); out[0] = 0xC0; return; }
    fs::ser_fp_be(p.x, out); out[0] |= 0x80 | (lex_largest(p.y) ? 0x20 : 0);
}
inline void put_g2(const G2A& p, bool compress, uint8_t* out) {
    if (!compress) { fs::ser_g2(p, out); return; }
    if (is_inf(p)) { std::memset(out, 0, 96); out[0] = 0xC0; return; }
    fs::ser_fp_be(p.x.c1, out); fs::ser_fp_be(p.x.c0, out + 48); out[0] |= 0x80 | (lex_largest(p.y) ? 0x20 : 0);
}
#endif

// ---- reading ---------------------------------------------------------------------------------------------------------
inline bool get_fp_be(const uint8_t* in, Fp& out, bool mask_flags) {        // false when the integer is >= p
    Fp c;
    for (int i = 0; i < 12; ++i) c.l[i] = (uint32_t)in[47 - 4 * i] | ((uint32_t)in[46 - 4 * i] << 8) | ((uint32_t)in[45 - 4 * i] << 16) | ((uint32_t)in[44 - 4 * i] << 24);
    if (mask_flags) c.l[11] &= 0x1fffffffu;
    uint32_t borrow = 0; for (int i = 0; i < 12; ++i) (void)subb32(c.l[i], FpParams::mod(i), borrow);
    if (!borrow) return false;
    out = to_mont(c); return true;
}
inline bool get_fp_le(const uint8_t* in, Fp& out) {
    Fp c; std::memcpy(c.l, in, 48);
    uint32_t borrow = 0; for (int i = 0; i < 12; ++i) (void)subb32(c.l[i], FpParams::mod(i), borrow);
    if (!borrow) return false;
    out = to_mont(c); return true;
}
inline bool get_fr(const uint8_t* in, Fr& out) {
    Fr c; std::memcpy(c.l, in, 32);
    uint32_t borrow = 0; for (int i = 0; i < 8; ++i) (void)subb32(c.l[i], FrParams::mod(i), borrow);
    if (!borrow) return false;
    out = to_mont(c); return true;
}
// `PairingOutput<P>::check` (ark-ec 0.4 pairing.rs, Valid impl; [ark-mem]): `self.0.pow(P::ScalarField::characteristic()).is_one()` --
// an Fq12 that is not in the order-r subgroup (zero included: 0^r = 0) is InvalidData.  Plain square-and-multiply: the value is untrusted,
// so the cyclotomic squaring formulas do not apply.  ~0.5 ms per element on the host; arkworks pays the same exponentiation.
inline bool gt_in_subgroup(const Fp12& f) {
    int top = 255; while (top > 0 && !((FrParams::mod(top >> 5) >> (top & 31)) & 1u)) --top;      // bit length of r from the parameters (255 bits on BLS12-381, 253 on BLS12-377)
    Fp12 acc = f;
    for (int i = top - 1; i >= 0; --i) { acc = sqr(acc); if ((FrParams::mod(i >> 5) >> (i & 31)) & 1u) acc = mul(acc, f); }
    return acc == Fp12::one();
}
inline bool get_gt(const uint8_t* in, Fp12& f) {
    Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
    for (int i = 0; i < 6; ++i) if (!get_fp_le(in + 96 * i, c[i]->c0) || !get_fp_le(in + 96 * i + 48, c[i]->c1)) return false;
    return gt_in_subgroup(f);
}
#if defined(RIPP_BLS12_377)
// square roots by Tonelli-Shanks: p - 1 = 2^S t with S = 46 on BLS12-377 (S and t are read off the modulus, the non-residue is searched once)
inline bool fp_sqrt(const Fp& a, Fp& r) {
    if (a.is_zero()) { r = a; return true; }
    static int S = 0; static uint32_t t[12], t1h[12]; static Fp z;          // t, (t + 1) / 2, z = g^t for a non-residue g
    static const bool init = []() {
        uint32_t m[12]; for (int i = 0; i < 12; ++i) m[i] = FpParams::mod(i);
        m[0] -= 1u;                                                      // p - 1 (p is odd)
        S = 0; while (!((m[S >> 5] >> (S & 31)) & 1u)) ++S;
        for (int i = 0; i < 12; ++i) { const int w = i + (S >> 5); const uint64_t lo = w < 12 ? m[w] : 0u, hi = w + 1 < 12 ? m[w + 1] : 0u; t[i] = (uint32_t)(((hi << 32) | lo) >> (S & 31)); }
        uint32_t c = 1; for (int i = 0; i < 12; ++i) { const uint64_t v = (uint64_t)t[i] + c; t1h[i] = (uint32_t)v; c = (uint32_t)(v >> 32); }
        for (int i = 0; i < 12; ++i) t1h[i] = (t1h[i] >> 1) | (i + 1 < 12 ? t1h[i + 1] << 31 : 0);
        uint32_t h[12]; for (int i = 0; i < 12; ++i) h[i] = FpParams::mod(i);       // (p - 1) / 2: Euler's criterion
        h[0] -= 1u; for (int i = 0; i < 12; ++i) h[i] = (h[i] >> 1) | (i + 1 < 12 ? h[i + 1] << 31 : 0);
        Fp g = Fp::one();
        for (;;) { g = add(g, Fp::one()); if (!(pow_limbs(g, h) == Fp::one())) break; }
        z = pow_limbs(g, t);
        return true; }();
    (void)init;
    Fp c = z, x = pow_limbs(a, t1h), b = pow_limbs(a, t);               // x^2 = a b
    int M = S;
    while (!(b == Fp::one())) {
        int i = 0; Fp b2 = b; while (!(b2 == Fp::one())) { b2 = mul(b2, b2); if (++i == M) return false; }      // a is a non-residue
        Fp e = c; for (int k = 0; k < M - i - 1; ++k) e = mul(e, e);
        x = mul(x, e); c = mul(e, e); b = mul(b, c); M = i;
    }
    r = x;
    return mul(r, r) == a;
}
#else
// square roots (p = 3 mod 4)
inline bool fp_sqrt(const Fp& a, Fp& r) {
    uint32_t e[12]; uint32_t carry = 1;                             // (p + 1) / 4
    for (int i = 0; i < 12; ++i) { const uint64_t s = (uint64_t)FpParams::mod(i) + carry; e[i] = (uint32_t)s; carry = (uint32_t)(s >> 32); }
    for (int i = 0; i < 12; ++i) e[i] = (e[i] >> 2) | (i + 1 < 12 ? e[i + 1] << 30 : 0);
    r = pow_limbs(a, e);
    return mul(r, r) == a;
}
#endif
inline bool fp2_sqrt(const Fp2& a, Fp2& r) {
#if defined(RIPP_BLS12_377)
    const Fp nb = add(add(add(Fp::one(), Fp::one()), add(Fp::one(), Fp::one())), Fp::one());      // -beta = 5: norm(a0 + a1 u) = a0^2 + 5 a1^2
    if (a.c1.is_zero()) {                                           // a in Fp: sqrt(a0), or u * sqrt(a0 / u^2) = u * sqrt(-a0 / 5)
        Fp s; if (fp_sqrt(a.c0, s)) { r = {s, Fp::zero()}; return true; }
        if (fp_sqrt(mul(neg(a.c0), inv(nb)), s)) { r = {Fp::zero(), s}; return true; }
        return false;
    }
    Fp n, s; n = add(mul(a.c0, a.c0), mul(nb, mul(a.c1, a.c1)));
#else
    if (a.c1.is_zero()) {                                           // a in Fp: sqrt(a0) or u * sqrt(-a0)
        Fp s; if (fp_sqrt(a.c0, s)) { r = {s, Fp::zero()}; return true; }
        if (fp_sqrt(neg(a.c0), s)) { r = {Fp::zero(), s}; return true; }
        return false;
    }
    Fp n, s; n = add(mul(a.c0, a.c0), mul(a.c1, a.c1));
#endif
    if (!fp_sqrt(n, s)) return false;
    const Fp half_ = inv(add(Fp::one(), Fp::one()));
    Fp t = mul(add(a.c0, s), half_), x0;
    if (!fp_sqrt(t, x0)) { t = mul(sub(a.c0, s), half_); if (!fp_sqrt(t, x0)) return false; }
    const Fp x1 = mul(a.c1, inv(add(x0, x0)));
    r = {x0, x1};
    return sqr(r) == a;
}
#if defined(RIPP_BLS12_377)
inline Fp curve_b(const Fp*) { return Fp::one(); }                                                   // y^2 = x^3 + 1
inline Fp2 curve_b(const Fp2*) { return {Fp::zero(), fp_const(RIPP_FP_TWIST_B1)}; }                  // D-type twist: 1 / u = (0, -1/5)
#else
inline Fp curve_b(const Fp*) { Fp four = Fp::one(); four = add(four, four); return add(four, four); }
inline Fp2 curve_b(const Fp2*) { const Fp b = curve_b((const Fp*)nullptr); return {b, b}; }          // 4(1 + u)
#endif
template <class F> bool on_curve(const Affine<F>& p) { return fsqr(p.y) == add(fmul(fsqr(p.x), p.x), curve_b((const F*)nullptr)); }
template <class F> bool in_subgroup(const Affine<F>& p) {
    uint32_t r[8]; for (int i = 0; i < 8; ++i) r[i] = FrParams::mod(i);
    return is_inf(scalar_mul_bits(p, r, 255));                     // (255 bits cover r on both curves: leading zero bits only double the identity)
}
// deserialize_{un,}compressed with Validate::Yes: flags, range, curve equation, prime-order subgroup
#if defined(RIPP_BLS12_377)
// generic SWFlags layout: the flags sit in the LAST byte of the image (of x when compressed, of y otherwise); `both flags set` is invalid
inline bool get_fp_le_flags(const uint8_t* in, Fp& out, uint8_t& flags) {
    uint8_t b[48]; std::memcpy(b, in, 48); flags = b[47] & 0xC0; b[47] &= 0x3F;
    return get_fp_le(b, out);
}
inline bool get_g1(const uint8_t* in, bool compress, G1A& p) {
    uint8_t fl; Fp y;
    if (compress) { if (!get_fp_le_flags(in, p.x, fl)) return false; }
    else { if (!get_fp_le(in, p.x)) return false; if (!get_fp_le_flags(in + 48, y, fl)) return false; }
    if (fl == 0xC0) return false;
    if (fl & 0x40) { p = aff_inf<Fp>(); return true; }            // arkworks returns zero whatever the coordinate bytes hold
    if (compress) {
        if (!fp_sqrt(add(mul(mul(p.x, p.x), p.x), curve_b((const Fp*)nullptr)), y)) return false;
        if (lex_largest(y) != ((fl & 0x80) != 0)) y = neg(y);
        p.y = y;
    } else { p.y = y; if (!on_curve(p)) return false; }
    return in_subgroup(p);
}
inline bool get_g2(const uint8_t* in, bool compress, G2A& p) {
    uint8_t fl, f0; Fp2 y;
    if (compress) { if (!get_fp_le(in, p.x.c0) || !get_fp_le_flags(in + 48, p.x.c1, fl)) return false; }
    else { if (!get_fp_le(in, p.x.c0) || !get_fp_le(in + 48, p.x.c1) || !get_fp_le(in + 96, y.c0) || !get_fp_le_flags(in + 144, y.c1, fl)) return false; }
    (void)f0;
    if (fl == 0xC0) return false;
    if (fl & 0x40) { p = aff_inf<Fp2>(); return true; }
    if (compress) {
        if (!fp2_sqrt(add(mul(sqr(p.x), p.x), curve_b((const Fp2*)nullptr)), y)) return false;
        if (lex_largest(y) != ((fl & 0x80) != 0)) y = neg(y);
        p.y = y;
    } else { p.y = y; if (!on_curve(p)) return false; }
    return in_subgroup(p);
}
#else
inline bool get_g1(const uint8_t* in, bool compress, G1A& p) {
    const uint8_t fl = in[0];
    if (((fl & 0x80) != 0) != compress) return false;
    if (fl & 0x40) { for (size_t i = 0; i < g1_size(compress); ++i) if ((i ? in[i] : (in[0] & 0x1f)) != 0) return false; if (!compress && (fl & 0x20)) return false; p = aff_inf<Fp>(); return true; }
    if (!get_fp_be(in, p.x, true)) return false;
    if (compress) {
        Fp y; if (!fp_sqrt(add(mul(mul(p.x, p.x), p.x), curve_b((const Fp*)nullptr)), y)) return false;
        if (lex_largest(y) != ((fl & 0x20) != 0)) y = neg(y);
        p.y = y;
    } else { if (fl & 0x20) return false; if (!get_fp_be(in + 48, p.y, false)) return false; if (!on_curve(p)) return false; }
    return in_subgroup(p);
}
inline bool get_g2(const uint8_t* in, bool compress, G2A& p) {
    const uint8_t fl = in[0];
    if (((fl & 0x80) != 0) != compress) return false;
    if (fl & 0x40) { for (size_t i = 0; i < g2_size(compress); ++i) if ((i ? in[i] : (in[0] & 0x1f)) != 0) return false; if (!compress && (fl & 0x20)) return false; p = aff_inf<Fp2>(); return true; }
    if (!get_fp_be(in, p.x.c1, true) || !get_fp_be(in + 48, p.x.c0, false)) return false;
    if (compress) {
        Fp2 y; if (!fp2_sqrt(add(mul(sqr(p.x), p.x), curve_b((const Fp2*)nullptr)), y)) return false;
        if (lex_largest(y) != ((fl & 0x20) != 0)) y = neg(y);
        p.y = y;
    } else { if (fl & 0x20) return false; if (!get_fp_be(in + 96, p.y.c1, false) || !get_fp_be(in + 144, p.y.c0, false)) return false; if (!on_curve(p)) return false; }
    return in_subgroup(p);
}
#endif

struct Writer {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { uint8_t t[8]; std::memcpy(t, &v, 8); b.insert(b.end(), t, t + 8); }
    void gt(const Fp12& f) { const size_t o = b.size(); b.resize(o + 576); fs::ser_gt(f, b.data() + o); }
    void fr(const Fr& s) { const size_t o = b.size(); b.resize(o + 32); fs::ser_fr(s, b.data() + o); }
    void g1(const G1A& p, bool c) { const size_t o = b.size(); b.resize(o + g1_size(c)); put_g1(p, c, b.data() + o); }
    void g2(const G2A& p, bool c) { const size_t o = b.size(); b.resize(o + g2_size(c)); put_g2(p, c, b.data() + o); }
};
struct Reader {
    const uint8_t* p; size_t left; bool ok = true;
    const uint8_t* take(size_t n) { if (!ok || left < n) { ok = false; return nullptr; } const uint8_t* r = p; p += n; left -= n; return r; }
    uint64_t u64() { const uint8_t* q = take(8); uint64_t v = 0; if (q) std::memcpy(&v, q, 8); return v; }
    void gt(Fp12& f) { const uint8_t* q = take(576); if (q && !get_gt(q, f)) ok = false; }
    void fr(Fr& s) { const uint8_t* q = take(32); if (q && !get_fr(q, s)) ok = false; }
    void g1(G1A& a, bool c) { const uint8_t* q = take(g1_size(c)); if (q && !get_g1(q, c, a)) ok = false; }
    void g2(G2A& a, bool c) { const uint8_t* q = take(g2_size(c)); if (q && !get_g2(q, c, a)) ok = false; }
};

}}  // namespace ripp::wire
