/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Self-contained BLAKE2s-256 / BLAKE2b-512 (RFC 7693, unkeyed) and
 * ChaCha20 (64-bit counter, 64-bit stream id = 0) as `rand_chacha 0.3::ChaCha20Rng` (un-vendored dependency,
 * sipp/Cargo.toml:29-31).  Call sites: sipp/src/rng.rs:54-72 (FiatShamirRng), ip_proofs/src/gipa.rs:235-258.
 */
#ifndef RIPP_ORACLE_HASH_H
#define RIPP_ORACLE_HASH_H
#include <stdint.h>
#include <string.h>
#include <stddef.h>

static const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

/* ---------------------------------------------------------------- BLAKE2s */
typedef struct { uint32_t h[8]; uint64_t t; uint8_t buf[64]; size_t buflen; } blake2s_ctx;
static const uint32_t B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void blake2s_compress(blake2s_ctx *c, const uint8_t *blk, int last) {
    uint32_t m[16], v[16];
    memcpy(m, blk, 64);
    for (int i = 0; i < 8; ++i) { v[i] = c->h[i]; v[i + 8] = B2S_IV[i]; }
    v[12] ^= (uint32_t)c->t; v[13] ^= (uint32_t)(c->t >> 32); if (last) v[14] = ~v[14];
#define G2S(a, b, cc, d, x, y) do { v[a] += v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16); v[cc] += v[d]; v[b] = rotr32(v[b] ^ v[cc], 12); \
        v[a] += v[b] + (y); v[d] = rotr32(v[d] ^ v[a], 8); v[cc] += v[d]; v[b] = rotr32(v[b] ^ v[cc], 7); } while (0)
    for (int r = 0; r < 10; ++r) {
        const uint8_t *s = B2_SIGMA[r];
        G2S(0, 4, 8, 12, m[s[0]], m[s[1]]); G2S(1, 5, 9, 13, m[s[2]], m[s[3]]); G2S(2, 6, 10, 14, m[s[4]], m[s[5]]); G2S(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G2S(0, 5, 10, 15, m[s[8]], m[s[9]]); G2S(1, 6, 11, 12, m[s[10]], m[s[11]]); G2S(2, 7, 8, 13, m[s[12]], m[s[13]]); G2S(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef G2S
    for (int i = 0; i < 8; ++i) c->h[i] ^= v[i] ^ v[i + 8];
}
static void blake2s_init(blake2s_ctx *c) { memcpy(c->h, B2S_IV, 32); c->h[0] ^= 0x01010000u ^ 32u; c->t = 0; c->buflen = 0; }
static void blake2s_update(blake2s_ctx *c, const uint8_t *in, size_t n) {
    while (n) {
        if (c->buflen == 64) { c->t += 64; blake2s_compress(c, c->buf, 0); c->buflen = 0; }
        size_t k = 64 - c->buflen; if (k > n) k = n;
        memcpy(c->buf + c->buflen, in, k); c->buflen += k; in += k; n -= k;
    }
}
static void blake2s_final(blake2s_ctx *c, uint8_t out[32]) {
    c->t += c->buflen; memset(c->buf + c->buflen, 0, 64 - c->buflen); blake2s_compress(c, c->buf, 1); memcpy(out, c->h, 32);
}
static void blake2s(const uint8_t *in, size_t n, uint8_t out[32]) { blake2s_ctx c; blake2s_init(&c); blake2s_update(&c, in, n); blake2s_final(&c, out); }

/* ---------------------------------------------------------------- BLAKE2b */
typedef struct { uint64_t h[8]; uint64_t t; uint8_t buf[128]; size_t buflen; } blake2b_ctx;
static const uint64_t B2B_IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static void blake2b_compress(blake2b_ctx *c, const uint8_t *blk, int last) {
    uint64_t m[16], v[16];
    memcpy(m, blk, 128);
    for (int i = 0; i < 8; ++i) { v[i] = c->h[i]; v[i + 8] = B2B_IV[i]; }
    v[12] ^= c->t; if (last) v[14] = ~v[14];
#define G2B(a, b, cc, d, x, y) do { v[a] += v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[cc] += v[d]; v[b] = rotr64(v[b] ^ v[cc], 24); \
        v[a] += v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[cc] += v[d]; v[b] = rotr64(v[b] ^ v[cc], 63); } while (0)
    for (int r = 0; r < 12; ++r) {
        const uint8_t *s = B2_SIGMA[r];
        G2B(0, 4, 8, 12, m[s[0]], m[s[1]]); G2B(1, 5, 9, 13, m[s[2]], m[s[3]]); G2B(2, 6, 10, 14, m[s[4]], m[s[5]]); G2B(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G2B(0, 5, 10, 15, m[s[8]], m[s[9]]); G2B(1, 6, 11, 12, m[s[10]], m[s[11]]); G2B(2, 7, 8, 13, m[s[12]], m[s[13]]); G2B(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef G2B
    for (int i = 0; i < 8; ++i) c->h[i] ^= v[i] ^ v[i + 8];
}
static void blake2b_init(blake2b_ctx *c) { memcpy(c->h, B2B_IV, 64); c->h[0] ^= 0x01010000ull ^ 64ull; c->t = 0; c->buflen = 0; }
static void blake2b_update(blake2b_ctx *c, const uint8_t *in, size_t n) {
    while (n) {
        if (c->buflen == 128) { c->t += 128; blake2b_compress(c, c->buf, 0); c->buflen = 0; }
        size_t k = 128 - c->buflen; if (k > n) k = n;
        memcpy(c->buf + c->buflen, in, k); c->buflen += k; in += k; n -= k;
    }
}
static void blake2b_final(blake2b_ctx *c, uint8_t out[64]) {
    c->t += c->buflen; memset(c->buf + c->buflen, 0, 128 - c->buflen); blake2b_compress(c, c->buf, 1); memcpy(out, c->h, 64);
}
static void blake2b(const uint8_t *in, size_t n, uint8_t out[64]) { blake2b_ctx c; blake2b_init(&c); blake2b_update(&c, in, n); blake2b_final(&c, out); }

/* ---------------------------------------------------------------- ChaCha20 block */
static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static void chacha20_block(const uint8_t key[32], uint64_t counter, uint8_t out[64]) {
    uint32_t s[16], w[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    memcpy(&s[4], key, 32);
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;
    memcpy(w, s, 64);
#define QR(a, b, c, d) do { w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 12); \
        w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 8); w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 7); } while (0)
    for (int i = 0; i < 10; ++i) { QR(0, 4, 8, 12); QR(1, 5, 9, 13); QR(2, 6, 10, 14); QR(3, 7, 11, 15); QR(0, 5, 10, 15); QR(1, 6, 11, 12); QR(2, 7, 8, 13); QR(3, 4, 9, 14); }
#undef QR
    for (int i = 0; i < 16; ++i) w[i] += s[i];
    memcpy(out, w, 64);
}

/* FiatShamirRng<Blake2s> (sipp/src/rng.rs:12-72): seed = H(bytes); absorb: seed = H(new || seed), stream restarts */
typedef struct { uint8_t seed[32]; uint64_t pos; } fsrng_t;
static void fsrng_from_digest(fsrng_t *g, const uint8_t digest[32]) { memcpy(g->seed, digest, 32); g->pos = 0; }
static void fsrng_from_seed(fsrng_t *g, const uint8_t *bytes, size_t n) { blake2s(bytes, n, g->seed); g->pos = 0; }
static void fsrng_absorb(fsrng_t *g, const uint8_t *bytes, size_t n) {
    blake2s_ctx c; blake2s_init(&c); blake2s_update(&c, bytes, n); blake2s_update(&c, g->seed, 32); blake2s_final(&c, g->seed); g->pos = 0;
}
static void fsrng_fill(fsrng_t *g, uint8_t *out, size_t n) {
    while (n) {
        uint8_t blk[64]; chacha20_block(g->seed, g->pos / 64, blk);
        size_t off = g->pos % 64, k = 64 - off; if (k > n) k = n;
        memcpy(out, blk + off, k); out += k; n -= k; g->pos += k;
    }
}
/* rand 0.8 `Standard` for u128: two next_u64 draws, low half first; each u64 = two LE keystream words */
static void fsrng_next_u128(fsrng_t *g, uint64_t *lo, uint64_t *hi) { uint8_t b[16]; fsrng_fill(g, b, 16); memcpy(lo, b, 8); memcpy(hi, b + 8, 8); }
#endif
