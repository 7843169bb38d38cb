// Host-side Fiat-Shamir plumbing of the SIPP prover (product code; NOT the oracle):
//   * ark-serialize 0.4 `serialize_uncompressed` byte images of Fr / G1Affine / G2Affine / Fq12
//     (ark-bls12-381 0.4 uses the zcash big-endian point layout), as hashed at sipp/src/lib.rs:56-59,80-84
//   * BLAKE2s-256 (RFC 7693) = `blake2::Blake2s` of sipp/src/lib.rs:230
//   * ChaCha20 keystream = `rand_chacha::ChaChaRng`, and `FiatShamirRng` of sipp/src/rng.rs:47-72
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>
#include "bls12_381/pairing.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace ripp { namespace fs {

// ------------------------------------------------------------------ BLAKE2s
#if defined(__x86_64__) && !defined(RIPP_NO_B2S_ASM)
extern "C" void ripp_blake2s_blocks_x64(uint32_t h[8], const uint8_t* in, size_t nblocks, uint64_t t);      // t: bytes hashed before the first block
#endif
struct Blake2s {
    uint32_t h[8]; uint64_t t = 0; uint8_t buf[64]; size_t buflen = 0;
    static constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    Blake2s() { for (int i = 0; i < 8; ++i) h[i] = IV[i]; h[0] ^= 0x01010000u ^ 32u; }
    static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    static const uint8_t* sigma(int r) {
        static const uint8_t S[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        return S[r];
    }
#if defined(__x86_64__) && defined(RIPP_BLAKE2S_AVX512)
    // MEASURED on the GPU pool's EPYC 9575F (Zen 5): scalar 870 MB/s, this AVX-512VL form 487 MB/s, an SSSE3 form
    // 373 MB/s -- the wide out-of-order core already runs the four column G's in parallel, so the vector forms only
    // lengthen the dependent chain.  Kept for CPUs where it wins; OFF by default.
    // The statement hash (336 MB at n = 2^20) sits on the prover's critical path and BLAKE2s is inherently sequential.
    // AVX-512VL form: the 16 message words live in one zmm, each round's schedule is ONE vpermd, rotations are vprord.
    // (An SSSE3 form that gathered message words with scalar inserts measured SLOWER than scalar code on Zen 5.)
    static bool have_avx512() { static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"); return ok; }
    __attribute__((target("avx512f,avx512vl"))) void compress_avx512(const uint8_t* blk, bool last) {
        alignas(64) static const int32_t IDX[10][16] = {
#define RIPP_S(a0,a1,a2,a3,a4,a5,a6,a7,a8,a9,a10,a11,a12,a13,a14,a15) {a0,a2,a4,a6, a1,a3,a5,a7, a8,a10,a12,a14, a9,a11,a13,a15}
            RIPP_S(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), RIPP_S(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3),
            RIPP_S(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4), RIPP_S(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8),
            RIPP_S(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13), RIPP_S(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9),
            RIPP_S(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11), RIPP_S(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10),
            RIPP_S(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5), RIPP_S(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)};
#undef RIPP_S
        const __m512i m = _mm512_loadu_si512((const void*)blk);
        __m128i row1 = _mm_loadu_si128((const __m128i*)&h[0]), row2 = _mm_loadu_si128((const __m128i*)&h[4]);
        __m128i row3 = _mm_loadu_si128((const __m128i*)&IV[0]);
        __m128i row4 = _mm_xor_si128(_mm_loadu_si128((const __m128i*)&IV[4]), _mm_set_epi32(0, last ? -1 : 0, (int)(uint32_t)(t >> 32), (int)(uint32_t)t));
        const __m128i f1 = row1, f2 = row2;
#define RIPP_B2S_G(b1, b2) \
        row1 = _mm_add_epi32(_mm_add_epi32(row1, b1), row2); row4 = _mm_ror_epi32(_mm_xor_si128(row4, row1), 16); \
        row3 = _mm_add_epi32(row3, row4); row2 = _mm_ror_epi32(_mm_xor_si128(row2, row3), 12); \
        row1 = _mm_add_epi32(_mm_add_epi32(row1, b2), row2); row4 = _mm_ror_epi32(_mm_xor_si128(row4, row1), 8); \
        row3 = _mm_add_epi32(row3, row4); row2 = _mm_ror_epi32(_mm_xor_si128(row2, row3), 7);
        for (int r = 0; r < 10; ++r) {
            const __m512i pm = _mm512_permutexvar_epi32(_mm512_load_si512((const void*)IDX[r]), m);
            const __m128i b1 = _mm512_castsi512_si128(pm), b2 = _mm512_extracti32x4_epi32(pm, 1);
            const __m128i b3 = _mm512_extracti32x4_epi32(pm, 2), b4 = _mm512_extracti32x4_epi32(pm, 3);
            RIPP_B2S_G(b1, b2)
            row4 = _mm_shuffle_epi32(row4, _MM_SHUFFLE(2, 1, 0, 3)); row3 = _mm_shuffle_epi32(row3, _MM_SHUFFLE(1, 0, 3, 2)); row2 = _mm_shuffle_epi32(row2, _MM_SHUFFLE(0, 3, 2, 1));
            RIPP_B2S_G(b3, b4)
            row4 = _mm_shuffle_epi32(row4, _MM_SHUFFLE(0, 3, 2, 1)); row3 = _mm_shuffle_epi32(row3, _MM_SHUFFLE(1, 0, 3, 2)); row2 = _mm_shuffle_epi32(row2, _MM_SHUFFLE(2, 1, 0, 3));
        }
#undef RIPP_B2S_G
        _mm_storeu_si128((__m128i*)&h[0], _mm_xor_si128(f1, _mm_xor_si128(row1, row3)));
        _mm_storeu_si128((__m128i*)&h[4], _mm_xor_si128(f2, _mm_xor_si128(row2, row4)));
    }
    void compress(const uint8_t* blk, bool last) { if (have_avx512()) compress_avx512(blk, last); else compress_scalar(blk, last); }
    void compress_scalar(const uint8_t* blk, bool last) {
#else
    void compress(const uint8_t* blk, bool last) {
#endif
        // Sixteen named state words, the ten rounds unrolled with the message schedule as compile-time constants, and the four
        // independent G's of every half-round interleaved statement by statement.  MEASURED on the GPU pool's EPYC 9575F (Zen 5,
        // tools/ubench/blake2s_bench.cpp, clang -O3): 1.14 GB/s against 0.87 GB/s for the round-loop + schedule-table form; the
        // statement hash (336 MB at n = 2^20) is the serial floor of SIPP::prove.  Do NOT build this file with -march=native /
        // vector ISA flags: Zen 5's 2-cycle SIMD integer adds make the auto-vectorised G 2-4x slower (0.3-0.6 GB/s).
        uint32_t m[16]; std::memcpy(m, blk, 64);
        uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
        uint32_t v8 = IV[0], v9 = IV[1], v10 = IV[2], v11 = IV[3], v12 = IV[4] ^ (uint32_t)t, v13 = IV[5] ^ (uint32_t)(t >> 32), v14 = last ? ~IV[6] : IV[6], v15 = IV[7];
#define RIPP_Q1(a, b, x) a += m[x]; a += b;
#define RIPP_Q2(d, a, n) d = rotr(d ^ a, n);
#define RIPP_Q3(c, d) c += d;
#define RIPP_HALF(a0,b0,c0,d0,a1,b1,c1,d1,a2,b2,c2,d2,a3,b3,c3,d3,x0,y0,x1,y1,x2,y2,x3,y3) \
        RIPP_Q1(a0,b0,x0) RIPP_Q1(a1,b1,x1) RIPP_Q1(a2,b2,x2) RIPP_Q1(a3,b3,x3) RIPP_Q2(d0,a0,16) RIPP_Q2(d1,a1,16) RIPP_Q2(d2,a2,16) RIPP_Q2(d3,a3,16) \
        RIPP_Q3(c0,d0) RIPP_Q3(c1,d1) RIPP_Q3(c2,d2) RIPP_Q3(c3,d3) RIPP_Q2(b0,c0,12) RIPP_Q2(b1,c1,12) RIPP_Q2(b2,c2,12) RIPP_Q2(b3,c3,12) \
        RIPP_Q1(a0,b0,y0) RIPP_Q1(a1,b1,y1) RIPP_Q1(a2,b2,y2) RIPP_Q1(a3,b3,y3) RIPP_Q2(d0,a0,8) RIPP_Q2(d1,a1,8) RIPP_Q2(d2,a2,8) RIPP_Q2(d3,a3,8) \
        RIPP_Q3(c0,d0) RIPP_Q3(c1,d1) RIPP_Q3(c2,d2) RIPP_Q3(c3,d3) RIPP_Q2(b0,c0,7) RIPP_Q2(b1,c1,7) RIPP_Q2(b2,c2,7) RIPP_Q2(b3,c3,7)
#define RIPP_ROUND(s0,s1,s2,s3,s4,s5,s6,s7,s8,s9,s10,s11,s12,s13,s14,s15) \
        RIPP_HALF(v0,v4,v8,v12, v1,v5,v9,v13, v2,v6,v10,v14, v3,v7,v11,v15, s0,s1,s2,s3,s4,s5,s6,s7) \
        RIPP_HALF(v0,v5,v10,v15, v1,v6,v11,v12, v2,v7,v8,v13, v3,v4,v9,v14, s8,s9,s10,s11,s12,s13,s14,s15)
        RIPP_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15) RIPP_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
        RIPP_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4) RIPP_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
        RIPP_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13) RIPP_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
        RIPP_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11) RIPP_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
        RIPP_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5) RIPP_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef RIPP_ROUND
#undef RIPP_HALF
#undef RIPP_Q3
#undef RIPP_Q2
#undef RIPP_Q1
        h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
    }
    void update(const uint8_t* in, size_t n) {
        while (n) {
            if (buflen == 64) { t += 64; compress(buf, false); buflen = 0; }
            if (buflen == 0 && n > 64) {   // bulk path: compress straight from the input
#if defined(__x86_64__) && !defined(RIPP_NO_B2S_ASM) && !defined(__HIP_DEVICE_COMPILE__)
                // Hand-allocated x86-64 form (blake2s_x64.S, generated by tools/ubench/gen_blake2s_x64.py): 4.23 cycles per byte on the pool's EPYC 9575F
                // against 4.40 for the compiled form below -- the statement hash is 73 % of a single-GPU proof.  It stages the NEXT block's message
                // while the current one runs, i.e. reads 64 bytes beyond the blocks it processes: it gets n / 64 - 1 blocks, the rest goes below.
                if (n >= 192) { const size_t k = n / 64 - 1; ripp_blake2s_blocks_x64(h, in, k, t); t += 64 * (uint64_t)k; in += 64 * k; n -= 64 * k; }
#endif
                while (n > 64) { t += 64; compress(in, false); in += 64; n -= 64; }
                continue;
            }
            size_t k = 64 - buflen; if (k > n) k = n;
            std::memcpy(buf + buflen, in, k); buflen += k; in += k; n -= k;
        }
    }
    void finish(uint8_t out[32]) { t += buflen; std::memset(buf + buflen, 0, 64 - buflen); compress(buf, true); std::memcpy(out, h, 32); }
};

// ------------------------------------------------------------------ BLAKE2b-512 (the digest of GIPA's challenges, ip_proofs/src/gipa.rs:248-251)
struct Blake2b {
    uint64_t h[8]; uint64_t t = 0; uint8_t buf[128]; size_t buflen = 0;
    static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                       0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    Blake2b() { for (int i = 0; i < 8; ++i) h[i] = IV[i]; h[0] ^= 0x01010000ull ^ 64ull; }
    static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(const uint8_t* blk, bool last) {
        uint64_t m[16], v[16];
        std::memcpy(m, blk, 128);
        for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= t; if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] += v[b] + x; v[d] = rotr(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 24);
            v[a] += v[b] + y; v[d] = rotr(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 63); };
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = Blake2s::sigma(r % 10);
            G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const uint8_t* in, size_t n) {
        while (n) {
            if (buflen == 128) { t += 128; compress(buf, false); buflen = 0; }
            size_t k = 128 - buflen; if (k > n) k = n;
            std::memcpy(buf + buflen, in, k); buflen += k; in += k; n -= k;
        }
    }
    void finish(uint8_t out[64]) { t += buflen; std::memset(buf + buflen, 0, 128 - buflen); compress(buf, true); std::memcpy(out, h, 64); }
};

// ------------------------------------------------------------------ ChaCha20 (64-bit counter, stream id 0)
inline void chacha20_block(const uint8_t key[32], uint64_t counter, uint8_t out[64]) {
    auto rotl = [](uint32_t x, int n) { return (x << n) | (x >> (32 - n)); };
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u}, w[16];
    std::memcpy(&s[4], key, 32);
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;
    std::memcpy(w, s, 64);
    auto QR = [&](int a, int b, int c, int d) {
        w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12);
        w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8); w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7); };
    for (int i = 0; i < 10; ++i) { QR(0, 4, 8, 12); QR(1, 5, 9, 13); QR(2, 6, 10, 14); QR(3, 7, 11, 15); QR(0, 5, 10, 15); QR(1, 6, 11, 12); QR(2, 7, 8, 13); QR(3, 4, 9, 14); }
    for (int i = 0; i < 16; ++i) w[i] += s[i];
    std::memcpy(out, w, 64);
}

// ------------------------------------------------------------------ serialisation
inline void ser_fp_le(const Fp& a, uint8_t* out) { const Fp c = from_mont(a); std::memcpy(out, c.l, 48); }
inline void ser_fp_be(const Fp& a, uint8_t* out) {
    const Fp c = from_mont(a);
    for (int i = 0; i < 12; ++i) { const uint32_t w = c.l[i]; out[47 - 4 * i] = (uint8_t)w; out[46 - 4 * i] = (uint8_t)(w >> 8); out[45 - 4 * i] = (uint8_t)(w >> 16); out[44 - 4 * i] = (uint8_t)(w >> 24); }
}
inline void ser_fr(const Fr& a, uint8_t* out) { const Fr c = from_mont(a); std::memcpy(out, c.l, 32); }
#if defined(RIPP_BLS12_377)
// ark-bls12-377 keeps the GENERIC ark-ec 0.4 short-Weierstrass encoding [ark-mem]: x then y, little-endian canonical, SWFlags in the two
// top bits of the LAST byte -- bit 7 YIsNegative (y > -y as integers; Fq2 compares c1 first, then c0), bit 6 PointAtInfinity -- also in
// the uncompressed form (SWCurveConfig::serialize_with_mode writes y.serialize_with_flags).
inline bool canon_gt(const Fp& a, const Fp& b) { const Fp x = from_mont(a), y = from_mont(b); for (int i = 11; i >= 0; --i) if (x.l[i] != y.l[i]) return x.l[i] > y.l[i]; return false; }
inline void ser_g1(const G1A& p, uint8_t* out) {
    if (is_inf(p)) { std::memset(out, 0, 96); out[95] = 0x40; return; }
    ser_fp_le(p.x, out); ser_fp_le(p.y, out + 48);
    if (canon_gt(p.y, neg(p.y))) out[95] |= 0x80;
}
inline void ser_g2(const G2A& p, uint8_t* out) {
    if (is_inf(p)) { std::memset(out, 0, 192); out[191] = 0x40; return; }
    ser_fp_le(p.x.c0, out); ser_fp_le(p.x.c1, out + 48); ser_fp_le(p.y.c0, out + 96); ser_fp_le(p.y.c1, out + 144);
    const bool ng = p.y.c1.is_zero() ? canon_gt(p.y.c0, neg(p.y.c0)) : canon_gt(p.y.c1, neg(p.y.c1));
    if (ng) out[191] |= 0x80;
}
#else
inline void ser_g1(const G1A& p, uint8_t* out) {
    if (is_inf(p)) { std::memset(out, 0, 96); out[0] = 0x40; return; }
    ser_fp_be(p.x, out); ser_fp_be(p.y, out + 48);
}
inline void ser_g2(const G2A& p, uint8_t* out) {
    if (is_inf(p)) { std::memset(out, 0, 192); out[0] = 0x40; return; }
    ser_fp_be(p.x.c1, out); ser_fp_be(p.x.c0, out + 48); ser_fp_be(p.y.c1, out + 96); ser_fp_be(p.y.c0, out + 144);
}
#endif
inline void ser_gt(const Fp12& f, uint8_t* out) {
    const Fp2* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
    for (int i = 0; i < 6; ++i) { ser_fp_le(c[i]->c0, out + 96 * i); ser_fp_le(c[i]->c1, out + 96 * i + 48); }
}

// ------------------------------------------------------------------ FiatShamirRng<Blake2s>
struct FiatShamirRng {
    uint8_t seed[32];
    void from_digest(const uint8_t d[32]) { std::memcpy(seed, d, 32); }
    // seed <- H(new || seed); the ChaCha stream restarts (sipp/src/rng.rs:67-72)
    void absorb(const uint8_t* bytes, size_t n) { Blake2s h; h.update(bytes, n); h.update(seed, 32); h.finish(seed); }
    // first `u128::rand` after a reseed: rand 0.8 Standard = two next_u64, low half first = 16 LE keystream bytes
    void next_u128(uint64_t& lo, uint64_t& hi) const { uint8_t blk[64]; chacha20_block(seed, 0, blk); std::memcpy(&lo, blk, 8); std::memcpy(&hi, blk + 8, 8); }
};
inline Fr fr_from_u128(uint64_t lo, uint64_t hi) {
    Fr t = Fr::zero(); t.l[0] = (uint32_t)lo; t.l[1] = (uint32_t)(lo >> 32); t.l[2] = (uint32_t)hi; t.l[3] = (uint32_t)(hi >> 32);
    return to_mont(t);
}
// sipp/src/lib.rs:80-85
inline Fr sipp_challenge(FiatShamirRng& rng, const Fp12& z_l, const Fp12& z_r) {
    uint8_t buf[1152]; ser_gt(z_l, buf); ser_gt(z_r, buf + 576);
    rng.absorb(buf, 1152);
    uint64_t lo, hi; rng.next_u128(lo, hi);
    return fr_from_u128(lo, hi);
}

// GIPA challenge (ip_proofs/src/gipa.rs:235-258) for the TIPP instantiation: Blake2b over
//   nonce (usize, big-endian 8 B) || previous challenge (Fr, 32 B LE; Default = 0) || com_1.{0,1,2} || com_2.{0,1,2}
// where com_x.2 is IdentityOutput(Vec<GT>) => u64-LE length prefix (1) before the element.
// Returns c = c128^-1 and c_inv = c128 (the reference swaps the names on purpose, :252-256).
inline Fr gipa_tipp_challenge(const Fr* prev, const Fp12 com[6], Fr& c_inv) {
    for (uint64_t nonce = 0;; ++nonce) {
        uint8_t buf[8 + 32 + 6 * 576 + 16], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        ser_fr(prev ? *prev : Fr::zero(), p); p += 32;
        for (int k = 0; k < 6; ++k) {
            if (k == 2 || k == 5) { const uint64_t one = 1; std::memcpy(p, &one, 8); p += 8; }
            ser_gt(com[k], p); p += 576;
        }
        uint8_t dig[64]; Blake2b h; h.update(buf, (size_t)(p - buf)); h.finish(dig);
        uint64_t hi = 0, lo = 0;
        for (int i = 0; i < 8; ++i) { hi = (hi << 8) | dig[i]; lo = (lo << 8) | dig[8 + i]; }
        const Fr c128 = fr_from_u128(lo, hi);
        if (!c128.is_zero()) { c_inv = c128; return inv(c128); }
    }
}

// Fr::from_random_bytes(&digest) of ark-ff 0.4 (Fp::from_random_bytes_with_flags::<EmptyFlags>): the first 32 bytes are
// read little-endian, the bits above MODULUS_BIT_SIZE (255 on BLS12-381, 253 on BLS12-377) are cleared, and the value is rejected when it is >= r.
inline bool fr_from_random_bytes(const uint8_t dig[64], Fr& out) {
    int top = 31; while (top > 0 && !((FrParams::mod(7) >> top) & 1u)) --top;                 // MODULUS_BIT_SIZE - 225: 30 on BLS12-381 (255-bit r), 28 on BLS12-377 (253-bit r)
    Fr t; std::memcpy(t.l, dig, 32); t.l[7] &= (top == 31) ? ~0u : ((1u << (top + 1)) - 1u);
    uint32_t borrow = 0; for (int i = 0; i < 8; ++i) (void)subb32(t.l[i], FrParams::mod(i), borrow);
    if (!borrow) return false;
    out = to_mont(t); return true;
}

// KZG challenge point of TIPA (ip_proofs/src/tipa/mod.rs:194-209) and TIPAWithSSM (structured_scalar_message.rs:231-246):
//   nonce (usize BE) || r_transcript.first() || ck_a_final (G2, uncompressed) [|| ck_b_final (G1)]  -> Blake2b -> from_random_bytes
inline Fr kzg_challenge(const Fr& first, const G2A& ck_a_final, const G1A* ck_b_final) {
    for (uint64_t nonce = 0;; ++nonce) {
        uint8_t buf[8 + 32 + 192 + 96], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        ser_fr(first, p); p += 32; ser_g2(ck_a_final, p); p += 192;
        if (ck_b_final) { ser_g1(*ck_b_final, p); p += 96; }
        uint8_t dig[64]; Blake2b h; h.update(buf, (size_t)(p - buf)); h.finish(dig);
        Fr c; if (fr_from_random_bytes(dig, c)) return c;
    }
}

// GIPA challenge with SSMPlaceholderCommitment on the right and IdentityCommitment<G1> for the inner product
// (gipa.rs:235-258 instantiated as in groth16_aggregation.rs:42-48): per side  GT (576) || Fr::zero() (32) || u64 1 || G1 (96).
inline Fr gipa_ssm_challenge(const Fr* prev, const Fp12 gt[2], const G1A g1[2], Fr& c_inv) {
    for (uint64_t nonce = 0;; ++nonce) {
        uint8_t buf[8 + 32 + 2 * (576 + 32 + 8 + 96)], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        ser_fr(prev ? *prev : Fr::zero(), p); p += 32;
        for (int k = 0; k < 2; ++k) {
            ser_gt(gt[k], p); p += 576;
            std::memset(p, 0, 32); p += 32;
            const uint64_t one = 1; std::memcpy(p, &one, 8); p += 8;
            ser_g1(g1[k], p); p += 96;
        }
        uint8_t dig[64]; Blake2b h; h.update(buf, (size_t)(p - buf)); h.finish(dig);
        uint64_t hi = 0, lo = 0;
        for (int i = 0; i < 8; ++i) { hi = (hi << 8) | dig[i]; lo = (lo << 8) | dig[8 + i]; }
        const Fr c128 = fr_from_u128(lo, hi);
        if (!c128.is_zero()) { c_inv = c128; return inv(c128); }
    }
}

// random-linear-combination challenge of aggregate_proofs (groth16_aggregation.rs:105-116)
inline Fr aggregation_challenge(const Fp12& com_a, const Fp12& com_b, const Fp12& com_c) {
    for (uint64_t nonce = 0;; ++nonce) {
        uint8_t buf[8 + 3 * 576], *p = buf;
        for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(nonce >> (56 - 8 * i));
        ser_gt(com_a, p); p += 576; ser_gt(com_b, p); p += 576; ser_gt(com_c, p); p += 576;
        uint8_t dig[64]; Blake2b h; h.update(buf, (size_t)(p - buf)); h.finish(dig);
        Fr r; if (fr_from_random_bytes(dig, r)) return r;
    }
}

}}  // namespace ripp::fs
