// BLAKE2s compression variants timed on the host CPU (the statement hash of SIPP::prove is the serial floor of the prover).
//   g++/clang++ -O3 -march=native blake2s_bench.cpp -o blake2s_bench && ./blake2s_bench
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t S[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// V0: the product's current form (round loop, table-driven schedule)
static void compress_v0(uint32_t h[8], const uint8_t* blk, uint64_t t, bool last) {
    uint32_t m[16], v[16];
    memcpy(m, blk, 64);
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32); if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
        v[a] += v[b] + x; v[d] = rotr(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 12);
        v[a] += v[b] + y; v[d] = rotr(v[d] ^ v[a], 8); v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 7); };
    for (int r = 0; r < 10; ++r) {
        const uint8_t* s = S[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

// V1: sixteen named state words, rounds fully unrolled with the schedule as compile-time constants, message words read from the block
#define G1(a, b, c, d, x, y) a += b + m[x]; d = rotr(d ^ a, 16); c += d; b = rotr(b ^ c, 12); a += b + m[y]; d = rotr(d ^ a, 8); c += d; b = rotr(b ^ c, 7);
#define ROUND1(s0,s1,s2,s3,s4,s5,s6,s7,s8,s9,s10,s11,s12,s13,s14,s15) \
    G1(v0, v4, v8, v12, s0, s1) G1(v1, v5, v9, v13, s2, s3) G1(v2, v6, v10, v14, s4, s5) G1(v3, v7, v11, v15, s6, s7) \
    G1(v0, v5, v10, v15, s8, s9) G1(v1, v6, v11, v12, s10, s11) G1(v2, v7, v8, v13, s12, s13) G1(v3, v4, v9, v14, s14, s15)
static void compress_v1(uint32_t h[8], const uint8_t* blk, uint64_t t, bool last) {
    uint32_t m[16]; memcpy(m, blk, 64);
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = IV[0], v9 = IV[1], v10 = IV[2], v11 = IV[3], v12 = IV[4] ^ (uint32_t)t, v13 = IV[5] ^ (uint32_t)(t >> 32), v14 = last ? ~IV[6] : IV[6], v15 = IV[7];
    ROUND1(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15) ROUND1(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    ROUND1(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4) ROUND1(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    ROUND1(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13) ROUND1(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    ROUND1(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11) ROUND1(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    ROUND1(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5) ROUND1(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// V2: as V1 but the four G's of a half-round are interleaved statement by statement (explicit ILP for in-order-ish schedulers)
#define Q1(a, b, x) a += b + m[x];
#define Q2(d, a, n) d = rotr(d ^ a, n);
#define Q3(c, d) c += d;
#define HALF(a0,b0,c0,d0,a1,b1,c1,d1,a2,b2,c2,d2,a3,b3,c3,d3,x0,y0,x1,y1,x2,y2,x3,y3) \
    Q1(a0,b0,x0) Q1(a1,b1,x1) Q1(a2,b2,x2) Q1(a3,b3,x3) Q2(d0,a0,16) Q2(d1,a1,16) Q2(d2,a2,16) Q2(d3,a3,16) Q3(c0,d0) Q3(c1,d1) Q3(c2,d2) Q3(c3,d3) \
    Q2(b0,c0,12) Q2(b1,c1,12) Q2(b2,c2,12) Q2(b3,c3,12) Q1(a0,b0,y0) Q1(a1,b1,y1) Q1(a2,b2,y2) Q1(a3,b3,y3) Q2(d0,a0,8) Q2(d1,a1,8) Q2(d2,a2,8) Q2(d3,a3,8) \
    Q3(c0,d0) Q3(c1,d1) Q3(c2,d2) Q3(c3,d3) Q2(b0,c0,7) Q2(b1,c1,7) Q2(b2,c2,7) Q2(b3,c3,7)
#define ROUND2(s0,s1,s2,s3,s4,s5,s6,s7,s8,s9,s10,s11,s12,s13,s14,s15) \
    HALF(v0,v4,v8,v12, v1,v5,v9,v13, v2,v6,v10,v14, v3,v7,v11,v15, s0,s1,s2,s3,s4,s5,s6,s7) \
    HALF(v0,v5,v10,v15, v1,v6,v11,v12, v2,v7,v8,v13, v3,v4,v9,v14, s8,s9,s10,s11,s12,s13,s14,s15)
static void compress_v2(uint32_t h[8], const uint8_t* blk, uint64_t t, bool last) {
    uint32_t m[16]; memcpy(m, blk, 64);
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = IV[0], v9 = IV[1], v10 = IV[2], v11 = IV[3], v12 = IV[4] ^ (uint32_t)t, v13 = IV[5] ^ (uint32_t)(t >> 32), v14 = last ? ~IV[6] : IV[6], v15 = IV[7];
    ROUND2(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15) ROUND2(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    ROUND2(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4) ROUND2(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    ROUND2(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13) ROUND2(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    ROUND2(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11) ROUND2(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    ROUND2(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5) ROUND2(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// V4: as V2, with the message word added to `a` BEFORE `b` ((a + m) + b: one dependent add after b instead of two)
#define P1(a, b, x) a += m[x]; a += b;
#define HALF4(a0,b0,c0,d0,a1,b1,c1,d1,a2,b2,c2,d2,a3,b3,c3,d3,x0,y0,x1,y1,x2,y2,x3,y3) \
    P1(a0,b0,x0) P1(a1,b1,x1) P1(a2,b2,x2) P1(a3,b3,x3) Q2(d0,a0,16) Q2(d1,a1,16) Q2(d2,a2,16) Q2(d3,a3,16) Q3(c0,d0) Q3(c1,d1) Q3(c2,d2) Q3(c3,d3) \
    Q2(b0,c0,12) Q2(b1,c1,12) Q2(b2,c2,12) Q2(b3,c3,12) P1(a0,b0,y0) P1(a1,b1,y1) P1(a2,b2,y2) P1(a3,b3,y3) Q2(d0,a0,8) Q2(d1,a1,8) Q2(d2,a2,8) Q2(d3,a3,8) \
    Q3(c0,d0) Q3(c1,d1) Q3(c2,d2) Q3(c3,d3) Q2(b0,c0,7) Q2(b1,c1,7) Q2(b2,c2,7) Q2(b3,c3,7)
#define ROUND4(s0,s1,s2,s3,s4,s5,s6,s7,s8,s9,s10,s11,s12,s13,s14,s15) \
    HALF4(v0,v4,v8,v12, v1,v5,v9,v13, v2,v6,v10,v14, v3,v7,v11,v15, s0,s1,s2,s3,s4,s5,s6,s7) \
    HALF4(v0,v5,v10,v15, v1,v6,v11,v12, v2,v7,v8,v13, v3,v4,v9,v14, s8,s9,s10,s11,s12,s13,s14,s15)
static void compress_v4(uint32_t h[8], const uint8_t* blk, uint64_t t, bool last) {
    uint32_t m[16]; memcpy(m, blk, 64);
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = IV[0], v9 = IV[1], v10 = IV[2], v11 = IV[3], v12 = IV[4] ^ (uint32_t)t, v13 = IV[5] ^ (uint32_t)(t >> 32), v14 = last ? ~IV[6] : IV[6], v15 = IV[7];
    ROUND4(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15) ROUND4(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    ROUND4(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4) ROUND4(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    ROUND4(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13) ROUND4(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    ROUND4(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11) ROUND4(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    ROUND4(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5) ROUND4(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}
// V5: as V4 but only TWO G's interleaved at a time (8 live state words + the rest parked by the compiler)
#define PAIR5(a0,b0,c0,d0,a1,b1,c1,d1,x0,y0,x1,y1) \
    P1(a0,b0,x0) P1(a1,b1,x1) Q2(d0,a0,16) Q2(d1,a1,16) Q3(c0,d0) Q3(c1,d1) Q2(b0,c0,12) Q2(b1,c1,12) P1(a0,b0,y0) P1(a1,b1,y1) Q2(d0,a0,8) Q2(d1,a1,8) Q3(c0,d0) Q3(c1,d1) Q2(b0,c0,7) Q2(b1,c1,7)
#define ROUND5(s0,s1,s2,s3,s4,s5,s6,s7,s8,s9,s10,s11,s12,s13,s14,s15) \
    PAIR5(v0,v4,v8,v12, v1,v5,v9,v13, s0,s1,s2,s3) PAIR5(v2,v6,v10,v14, v3,v7,v11,v15, s4,s5,s6,s7) \
    PAIR5(v0,v5,v10,v15, v1,v6,v11,v12, s8,s9,s10,s11) PAIR5(v2,v7,v8,v13, v3,v4,v9,v14, s12,s13,s14,s15)
static void compress_v5(uint32_t h[8], const uint8_t* blk, uint64_t t, bool last) {
    uint32_t m[16]; memcpy(m, blk, 64);
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = IV[0], v9 = IV[1], v10 = IV[2], v11 = IV[3], v12 = IV[4] ^ (uint32_t)t, v13 = IV[5] ^ (uint32_t)(t >> 32), v14 = last ? ~IV[6] : IV[6], v15 = IV[7];
    ROUND5(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15) ROUND5(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    ROUND5(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4) ROUND5(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    ROUND5(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13) ROUND5(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    ROUND5(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11) ROUND5(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    ROUND5(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5) ROUND5(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    h[0] ^= v0 ^ v8; h[1] ^= v1 ^ v9; h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11; h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// (A hand-allocated x86-64 assembly form -- whole state in GPRs, v12/v15 sharing a register, in-place two-operand G -- measured
//  1.11 GB/s on the EPYC 9575F against 1.14 GB/s for V2 as compiled by clang -O3: V2 already sits at the dependency-chain floor.)

// V6: hand-allocated x86-64 assembly over MANY blocks per call (gen_blake2s_x64.py: fifteen registers of state, a2 / a3 in rsp-relative slots)
#include "build/b2s_list.h"      // generated: one entry per build/b2s_*.S (gen_blake2s_x64.py <order> [keep_h] [early_copy])
typedef void (*fn_t)(uint32_t*, const uint8_t*, uint64_t, bool);
typedef void (*bulk_t)(uint32_t*, const uint8_t*, size_t, uint64_t);
static double run_bulk(bulk_t f, const std::vector<uint8_t>& buf, uint32_t out[8]) {
    uint32_t h[8]; for (int i = 0; i < 8; ++i) h[i] = IV[i]; h[0] ^= 0x01010020u;
    auto t0 = std::chrono::steady_clock::now();
    const size_t nb = buf.size() / 64;
    f(h, buf.data(), nb - 1, 0);
    compress_v2(h, buf.data() + 64 * (nb - 1), 64 * nb, true);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    memcpy(out, h, 32); return s;
}
static double run(fn_t f, const std::vector<uint8_t>& buf, uint32_t out[8]) {
    uint32_t h[8]; for (int i = 0; i < 8; ++i) h[i] = IV[i]; h[0] ^= 0x01010020u;
    auto t0 = std::chrono::steady_clock::now();
    uint64_t t = 0; const size_t nb = buf.size() / 64;
    for (size_t i = 0; i < nb; ++i) { t += 64; f(h, buf.data() + 64 * i, t, i + 1 == nb); }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    memcpy(out, h, 32); return s;
}
// core clock under this kind of load: a chain of dependent 1-cycle adds (16 per iteration), timed
static double core_ghz() {
    uint64_t x = 1; const uint64_t n = 200000000ull;
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n; ++i)
        asm volatile("add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n"
                     "add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0\n" : "+r"(x));
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 16.0 * (double)n / s / 1e9;
}
int main() {
    const double ghz = core_ghz();
    printf("core clock (dependent-add chain): %.2f GHz -> cycles per byte below = GHz / (GB/s)\n", ghz);
    std::vector<uint8_t> buf((size_t)64 << 20); buf.reserve(buf.size() + 64);      // (early_copy forms stage 64 bytes beyond the last block they process: the final block follows in this buffer)
    uint64_t x = 88172645463325252ull; for (auto& b : buf) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; b = (uint8_t)x; }
    struct { const char* name; fn_t f; } vs[] = {{"v0 loop+table", compress_v0}, {"v1 unrolled", compress_v1}, {"v2 unrolled interleaved", compress_v2}, {"v4 v2 + (a+m)+b", compress_v4}, {"v5 pairs + (a+m)+b", compress_v5}};
    uint32_t ref[8];
    for (auto& v : vs) {
        uint32_t out[8]; double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) { double s = run(v.f, buf, out); if (s < best) best = s; }
        if (v.f == compress_v0) memcpy(ref, out, 32);
        printf("%-28s %8.1f MB/s  %5.2f c/B  %s\n", v.name, buf.size() / best / 1e6, ghz * 1e9 * best / buf.size(), memcmp(ref, out, 32) ? "MISMATCH" : "ok");
    }
    for (auto& v : B2S_ASM) {
        uint32_t out[8]; double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) { double s = run_bulk(v.f, buf, out); if (s < best) best = s; }
        printf("%-28s %8.1f MB/s  %5.2f c/B  %s\n", v.name, buf.size() / best / 1e6, ghz * 1e9 * best / buf.size(), memcmp(ref, out, 32) ? "MISMATCH" : "ok");
    }
    printf("core clock after: %.2f GHz\n", core_ghz());
    return 0;
}
