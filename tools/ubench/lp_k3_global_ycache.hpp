// Stage 2a of the pairing product, Karatsuba form: the same decomposition as k_line_products_q (fq_line_products.hpp: the Fp12 accumulator in LDS, a
// group of 3 lanes per accumulator, lane j produces the w-basis coefficients j and j + 3 of f * line, ONE Montgomery reduction per real / imaginary part)
// with a THIRD fewer limb products per Fp2 product.
//
//   out_k = sum_t x_t y_t  (three Fp2 products, u^2 = -1):   A = sum x_t0 y_t0,  B = sum x_t1 y_t1,  C = sum (x_t0 + x_t1)(y_t0 + y_t1)
//   re = A - B,   im = C - A - B                             9 products + 2 reductions (2 156 multiply-adds) where the six-product sums take 12 + 2 (2 744).
//
// What makes it fit 256 registers and cost no extra carry work:
//   * SIGNED limbs and columns (v_mad_i64_i32): a value is sum l_i 2^(28 i) with |l_i| < 2^28, so a negation is limb-wise, a difference of two values needs
//     no bias and no normalisation, and the two column sets U = -A, V = -B give re = reduce(V - U), im = reduce(U + V + C) by 64-bit adds of whole columns
//     (the Montgomery reduction is exact on signed columns: low limbs by masking the two's complement, carries by arithmetic shifts; results lie in
//     (-0.01 p, 1.01 p) and are canonicalised once, at the end of the kernel);
//   * the LINE side is prepared once per line and group, not once per use: the three lanes of a group each unpack ONE coefficient of the line and write
//     -c0, -c1, c0 + c1 and (for the coefficients that meet a wrapped-around index, w^6 = xi = 1 + u) xi's images c1 - c0, -(c0 + c1), 2 c0 -- 15 values of
//     14 limbs -- to a per-wave Y-CACHE in global memory (20 KB per wave, L2-resident; LDS is full: 19.8 KB per wave of accumulators).  The product loops
//     then fetch every line operand just in time, one term ahead, in the form they multiply with: no unpacking, no xi arithmetic, no negation in the loops,
//     and only two line operands (28 registers) resident instead of six (84) -- the room the second column set needs;
//   * the accumulator side streams from LDS four limbs at a time, as before (x_t0 and x_t1 side by side: their sum for C is one add per limb).
// Per line and lane: 4 312 multiply-adds + ~0.9 k other vector instructions against 5 492 + ~1.25 k.
// BLS12-381 only (u^2 = -1, xi = 1 + u, M-type line l0 + l1 w^2 + l2 w^3); the BLS12-377 build keeps k_line_products_q.
#pragma once
#include <type_traits>
#include "fq_line_products.hpp"

namespace ripp {

constexpr int LK_NV = 15;                                             // y-cache values per line: l0 (-c0, -c1, s), l1 and l2 (-c0, -c1, s, c1 - c0, -s, 2 c0)
constexpr int LK_ROW_BYTES = LP_GROUPS_PER_WAVE * 16;                 // one 16-byte chunk of one value for the 21 groups of a wave: 336 B, contiguous
constexpr int LK_VAL_BYTES = 4 * LK_ROW_BYTES;                        // 14 limbs + 2: 1 344 B
constexpr int LK_WAVE_BYTES = LK_NV * LK_VAL_BYTES;                   // 20 160 B per wave
inline size_t lk_ycache_bytes(size_t waves) { return waves * 2 * (size_t)LK_WAVE_BYTES; }      // two halves per wave: this line's and the next one's

#if !defined(RIPP_BLS12_377)
typedef int32_t lk_v4i __attribute__((ext_vector_type(4)));
typedef int32_t lk_v2i __attribute__((ext_vector_type(2)));
struct LkBuf { lk_v4i a, b, c; lk_v2i d; };                           // a 14-limb operand as fetched

#define LK_CHUNK(slot, part) ((slot) * 9 + (part) * 4)             // fq_line_products.hpp LQ_CHUNK (the searched bank layout; LQ_ACC_STRIDE = 59)
#define LK_TIE(c) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), \
                                   "+v"(c[14]), "+v"(c[15]), "+v"(c[16]), "+v"(c[17]), "+v"(c[18]), "+v"(c[19]), "+v"(c[20]), "+v"(c[21]), "+v"(c[22]), "+v"(c[23]), "+v"(c[24]), "+v"(c[25]), "+v"(c[26]))
#define LK_TIE14(c) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]))

// grid = (ceil(T / 21), rows), block = 64 (one wave); arguments and output as k_line_products_q, plus the y-cache: lk_ycache_bytes(gridDim.x * gridDim.y) bytes
__global__ void __launch_bounds__(64, 2) k_line_products_k(const uint4* __restrict__ lines, size_t stride, uint32_t M, uint4* __restrict__ partials, uint32_t T, uint4* ycache) {
    __shared__ uint4 lds[LP_GROUPS_PER_WAVE * LQ_ACC_STRIDE];
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace fq28;
    const uint32_t lane = threadIdx.x;
    const uint32_t g = lane / LP_GROUP, j = lane - g * LP_GROUP;               // group in wave, lane in group (lane 63: g = 21, idle)
    auto group_index = [&]() { uint32_t z = 0; asm volatile("" : "+s"(z));
        const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
        return blockIdx.x * LP_GROUPS_PER_WAVE + l / LP_GROUP; };
    const uint32_t t_ = blockIdx.x * LP_GROUPS_PER_WAVE + g;
    const bool ingrp = g < (uint32_t)LP_GROUPS_PER_WAVE;
    const bool active = ingrp && t_ < T;
    const size_t row = blockIdx.y;
    const uint32_t acc_chunk = (active ? g : 0) * LQ_ACC_STRIDE;
    const uint32_t lds_base = (uint32_t)(size_t)lds;
    uint4* acc = lds + acc_chunk;
    auto st_limbs = [&](int slot, int part, const int32_t (&v)[NL]) {
        acc[LK_CHUNK(slot, part) + 0] = uint4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
        acc[LK_CHUNK(slot, part) + 1] = uint4{(uint32_t)v[4], (uint32_t)v[5], (uint32_t)v[6], (uint32_t)v[7]};
        acc[LK_CHUNK(slot, part) + 2] = uint4{(uint32_t)v[8], (uint32_t)v[9], (uint32_t)v[10], (uint32_t)v[11]};
        acc[LK_CHUNK(slot, part) + 3] = uint4{(uint32_t)v[12], (uint32_t)v[13], 0u, 0u}; };
    if (active) {                                                               // accumulator <- 1
        int32_t one[NL], zero[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) { one[i] = (int32_t)ONE_M392.l[i]; zero[i] = 0; }
        if (j == 0) st_limbs(0, 0, one); else st_limbs(j, 0, zero);
        st_limbs(j, 1, zero); st_limbs(j + 3, 0, zero); st_limbs(j + 3, 1, zero);
    }
    // this wave's y-cache; a lane addresses chunk row r of value v at  ycw + v * LK_VAL_BYTES + r * LK_ROW_BYTES + 16 g
#if defined(LK_EXP_SHARED_YC)
    char* ycw = reinterpret_cast<char*>(ycache) + ((size_t)(blockIdx.x % LK_EXP_SHARED_YC)) * (2 * LK_WAVE_BYTES);      // (timing experiment: WRONG results)
#else
    char* ycw = reinterpret_cast<char*>(ycache) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * LK_WAVE_BYTES);
#endif
    const uint32_t g16 = (ingrp ? g : 0) * 16;
    const uint32_t yv_t1 = g16 + (j < 2 ? 3u * LK_VAL_BYTES : 0u);              // term 1 of output j: xi l1 on the lanes whose operand wraps around w^6
    const uint32_t st = (uint32_t)stride;
    const uint4* __restrict__ lrow = lines + row * 18 * stride;
    const uint32_t iters = (M + T - 1) / T;
    __syncthreads();

    // ---- memory operations by hand (the compiler would hoist every fetch and keep all operands alive, or sink a prefetch to its use).  `pin`: a column of
    // the running sums, which orders a fetch BEFORE the multiply-adds that follow it.  Vector-memory loads return in order, so "vmcnt(n)" with n = the number
    // of loads issued AFTER the wanted one is exact; stores of the y-cache that are still in flight only make a wait longer, never shorter.
    auto fetch_y = [&](LkBuf& b, uint32_t voff, const char* base, int64_t& pin) {             // 14 limbs of one y-cache value (base is wave-uniform)
        asm volatile("global_load_dwordx4 %0, %5, %6\n\tglobal_load_dwordx4 %1, %5, %6 offset:336\n\tglobal_load_dwordx4 %2, %5, %6 offset:672\n\tglobal_load_dwordx2 %3, %5, %6 offset:1008"
                     : "=&v"(b.a), "=&v"(b.b), "=&v"(b.c), "=&v"(b.d), "+v"(pin) : "v"(voff), "s"(base));
    };
    static_assert(LK_ROW_BYTES == 336, "the offsets in fetch_y");
#define LK_ARRIVED_Y(BUF_, n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(BUF_.a), "+v"(BUF_.b), "+v"(BUF_.c), "+v"(BUF_.d))
    // four limbs (chunk q < 3) or two (q = 3) of the accumulator coefficient at LDS address `a` + part * 64
#define LK_LDX(DST_, AD_, off, PIN_) asm volatile("ds_read_b128 %0, %2 offset:" #off : "=&v"(DST_), "+v"(PIN_) : "v"(AD_))
#define LK_LDX2(DST_, AD_, off, PIN_) asm volatile("ds_read_b64 %0, %2 offset:" #off : "=&v"(DST_), "+v"(PIN_) : "v"(AD_))
#define LK_ARRIVED_X(XA_, XB_, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(XA_), "+v"(XB_))
    // the six 16-byte chunks of line coefficient l_j of line `ii` (c0: raw[0..2], c1: raw[3..5]), as stage 1 stored them
    auto fetch_raw = [&](lk_v4i (&raw)[6], uint32_t ii, int64_t& pin) {
        uint32_t off[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) off[c] = ((6 * j + (uint32_t)c) * st + ii) << 4;
        asm volatile("global_load_dwordx4 %0, %7, %13\n\tglobal_load_dwordx4 %1, %8, %13\n\tglobal_load_dwordx4 %2, %9, %13\n\tglobal_load_dwordx4 %3, %10, %13\n\tglobal_load_dwordx4 %4, %11, %13\n\tglobal_load_dwordx4 %5, %12, %13"
                     : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(raw[2]), "=&v"(raw[3]), "=&v"(raw[4]), "=&v"(raw[5]), "+v"(pin)
                     : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(lrow));
    };
    // the line side, once per line: lane j turns coefficient l_j into its y-cache values (value index of l_j: 0 (l0), 3 (l1), 9 (l2)) in the cache half `ych`
    auto build_y = [&](lk_v4i (&raw)[6], char* ych) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]));
        uint32_t w0[12], w1[12];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            w0[4 * c] = (uint32_t)raw[c].x; w0[4 * c + 1] = (uint32_t)raw[c].y; w0[4 * c + 2] = (uint32_t)raw[c].z; w0[4 * c + 3] = (uint32_t)raw[c].w;
            w1[4 * c] = (uint32_t)raw[3 + c].x; w1[4 * c + 1] = (uint32_t)raw[3 + c].y; w1[4 * c + 2] = (uint32_t)raw[3 + c].z; w1[4 * c + 3] = (uint32_t)raw[3 + c].w;
        }
        const Fqn c0 = fq_unpack(w0), c1 = fq_unpack(w1);                        // stage 1 stores canonical values (< p)
        const auto s = fq_norm(fq_add(c0, c1));                                  // c0 + c1, limbs < 2^28
        uint4* dst = reinterpret_cast<uint4*>(ych + (j == 0 ? 0u : j == 1 ? 3u : 9u) * LK_VAL_BYTES + g16);
        auto put = [&](int v, const int32_t (&x)[NL]) {
            uint4* d = dst + v * (LK_VAL_BYTES / 16);
            d[0] = uint4{(uint32_t)x[0], (uint32_t)x[1], (uint32_t)x[2], (uint32_t)x[3]};
            d[LK_ROW_BYTES / 16] = uint4{(uint32_t)x[4], (uint32_t)x[5], (uint32_t)x[6], (uint32_t)x[7]};
            d[2 * LK_ROW_BYTES / 16] = uint4{(uint32_t)x[8], (uint32_t)x[9], (uint32_t)x[10], (uint32_t)x[11]};
            d[3 * LK_ROW_BYTES / 16] = uint4{(uint32_t)x[12], (uint32_t)x[13], 0u, 0u}; };
        if (ingrp) {
            int32_t v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) v[k] = -(int32_t)c0.l[k];
            put(0, v);
#pragma unroll
            for (int k = 0; k < NL; ++k) v[k] = -(int32_t)c1.l[k];
            put(1, v);
#pragma unroll
            for (int k = 0; k < NL; ++k) v[k] = (int32_t)s.l[k];
            put(2, v);
            if (j != 0) {                                                        // xi (c0 + c1 u) = (c0 - c1) + (c0 + c1) u:  -(c0 - c1), -(c0 + c1), (c0 - c1) + (c0 + c1) = 2 c0
#pragma unroll
                for (int k = 0; k < NL; ++k) v[k] = (int32_t)c1.l[k] - (int32_t)c0.l[k];
                put(3, v);
#pragma unroll
                for (int k = 0; k < NL; ++k) v[k] = -(int32_t)s.l[k];
                put(4, v);
#pragma unroll
                for (int k = 0; k < NL; ++k) {                                   // 2 c0: limb k = bits [28 k - 1, 28 k + 27) of c0's integer
                    uint32_t x;
                    if (k == 0) x = w0[0] << 1;
                    else { const int bit = W * k - 1, ww = bit >> 5, sh = bit & 31;
                           if (sh + W <= 32 || ww + 1 >= 12) x = w0[ww] >> sh; else x = __builtin_amdgcn_alignbit(w0[ww + 1], w0[ww], sh); }
                    v[k] = (int32_t)(x & MASK);
                }
                put(5, v);
            }
        }
    };

    auto limbs_of = [](const LkBuf& b, int32_t (&y)[NL]) {
        y[0] = b.a.x; y[1] = b.a.y; y[2] = b.a.z; y[3] = b.a.w; y[4] = b.b.x; y[5] = b.b.y; y[6] = b.b.z; y[7] = b.b.w;
        y[8] = b.c.x; y[9] = b.c.y; y[10] = b.c.z; y[11] = b.c.w; y[12] = b.d.x; y[13] = b.d.y; };
    // signed Montgomery reduction of 27 columns (destroys them): r = (sum col_k 2^(28 k) + m p) / 2^392, limbs 0..12 in [0, 2^28), limb 13 signed
    auto reduce_cols = [&](int64_t (&col)[2 * NL - 1], int32_t (&r)[NL]) {
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            int64_t s = col[k] + carry;
            const uint32_t m = ((uint32_t)s * INV28) & MASK;
#pragma unroll
            for (int jj = 1; jj < NL; ++jj) col[k + jj] += (int64_t)(int32_t)m * (int32_t)P28.l[jj];
            s += (int64_t)(int32_t)m * (int32_t)P28.l[0];
            carry = s >> W;
        }
#pragma unroll
        for (int k = NL; k < 2 * NL - 1; ++k) { const int64_t s = col[k] + carry; r[k - NL] = (int32_t)((uint32_t)s & MASK); carry = s >> W; }
        r[NL - 1] = (int32_t)carry;
    };
    // rows 4 q .. 4 q + 3 of two products at once: U += x0 * Y0, V += x1 * Y1
    auto mads2 = [&](int64_t (&U)[2 * NL - 1], int64_t (&V)[2 * NL - 1], int q, const lk_v4i& x0, const lk_v4i& x1, const int32_t (&Y0)[NL], const int32_t (&Y1)[NL]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * q + u;
            if (i >= NL) continue;
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) { U[i + jj] += (int64_t)x0[u] * Y0[jj]; V[i + jj] += (int64_t)x1[u] * Y1[jj]; }
        }
    };
    auto mads1 = [&](int64_t (&Wc)[2 * NL - 1], int q, const lk_v4i& xs, const int32_t (&Y)[NL]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * q + u;
            if (i >= NL) continue;
#pragma unroll
            for (int jj = 0; jj < NL; ++jj) Wc[i + jj] += (int64_t)xs[u] * Y[jj];
        }
    };
    // One output coefficient.  sl[t]: accumulator slot of term t; yo[t]: the lane's byte offset into the y-cache for term t; yb[t]: the wave-uniform address of the
    // term's first value (-y_t0; -y_t1 and y_t0 + y_t1 follow at LK_VAL_BYTES, 2 LK_VAL_BYTES).  FIRST: the first pair of line operands is already travelling
    // (p0, p1: fetched before the previous iteration's last barrier); NEXT: fetch the raw coefficient of line `nii` for the next iteration on the way.
    auto one_output = [&](auto FIRST, auto NEXT, const int (&sl)[3], const uint32_t (&yo)[3], const char* const (&yb)[3], LkBuf& p0, LkBuf& p1, lk_v4i (&raw)[6], uint32_t nii, int32_t (&re)[NL], int32_t (&im)[NL]) {
        uint32_t at[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) at[t] = lds_base + (acc_chunk + (uint32_t)sl[t] * 9u) * 16u;
        int64_t U[2 * NL - 1], V[2 * NL - 1];
#pragma unroll
        for (int c = 0; c < 2 * NL - 1; ++c) { U[c] = 0; V[c] = 0; }
        LkBuf ya[2], yc[2];
        lk_v4i x0[2], x1[2];
        if constexpr (decltype(FIRST)::value) { ya[0] = p0; yc[0] = p1; }
        else { fetch_y(ya[0], yo[0], yb[0], U[NL - 1]); fetch_y(yc[0], yo[0], yb[0] + LK_VAL_BYTES, U[NL - 1]); }
        LK_LDX(x0[0], at[0], 0, U[NL - 1]); LK_LDX(x1[0], at[0], 64, U[NL - 1]);
        // ---- U = sum x_t0 (-y_t0) = -A,  V = sum x_t1 (-y_t1) = -B
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int p = t & 1;
            int32_t Y0[NL], Y1[NL];
            if (t < 2) {
                fetch_y(ya[p ^ 1], yo[t + 1], yb[t + 1], U[NL - 1]); fetch_y(yc[p ^ 1], yo[t + 1], yb[t + 1] + LK_VAL_BYTES, U[NL - 1]);
                LK_ARRIVED_Y(ya[p], 8); LK_ARRIVED_Y(yc[p], 8);
            } else { LK_ARRIVED_Y(ya[p], 0); LK_ARRIVED_Y(yc[p], 0); }
            limbs_of(ya[p], Y0); limbs_of(yc[p], Y1);
            // chunk q of x_t0 / x_t1 is in x0[q & 1] / x1[q & 1] (t, q: 12 steps, the next step's chunks fetched one step ahead)
            LK_LDX(x0[1], at[t], 16, U[NL - 1]); LK_LDX(x1[1], at[t], 80, U[NL - 1]); LK_ARRIVED_X(x0[0], x1[0], 2);
            mads2(U, V, 0, x0[0], x1[0], Y0, Y1); LK_TIE(U); LK_TIE(V);
            LK_LDX(x0[0], at[t], 32, U[NL - 1]); LK_LDX(x1[0], at[t], 96, U[NL - 1]); LK_ARRIVED_X(x0[1], x1[1], 2);
            mads2(U, V, 1, x0[1], x1[1], Y0, Y1); LK_TIE(U); LK_TIE(V);
            { lk_v2i a2, b2;                                                     // the last chunk: limbs 12, 13
              LK_LDX2(a2, at[t], 48, U[NL - 1]); LK_LDX2(b2, at[t], 112, U[NL - 1]); LK_ARRIVED_X(x0[0], x1[0], 2);
              mads2(U, V, 2, x0[0], x1[0], Y0, Y1); LK_TIE(U); LK_TIE(V);
              if (t < 2) { LK_LDX(x0[0], at[t + 1], 0, U[NL - 1]); LK_LDX(x1[0], at[t + 1], 64, U[NL - 1]); LK_ARRIVED_X(a2, b2, 2); }
              else LK_ARRIVED_X(a2, b2, 0);
              const lk_v4i a4{a2.x, a2.y, 0, 0}, b4{b2.x, b2.y, 0, 0};
              mads2(U, V, 3, a4, b4, Y0, Y1); LK_TIE(U); LK_TIE(V); }
        }
        // ---- the sums for C, two terms ahead of their use; then re = reduce(V - U) while they travel
        LkBuf ys[3];
        fetch_y(ys[0], yo[0], yb[0] + 2 * LK_VAL_BYTES, U[NL - 1]); fetch_y(ys[1], yo[1], yb[1] + 2 * LK_VAL_BYTES, U[NL - 1]);
#pragma unroll
        for (int c = 0; c < 2 * NL - 1; ++c) { const int64_t u = U[c], v = V[c]; U[c] = u + v; V[c] = v - u; }
        LK_TIE(V);
        reduce_cols(V, re); LK_TIE14(re);
        // ---- U = -A - B + sum (x_t0 + x_t1)(y_t0 + y_t1)
        LK_LDX(x0[0], at[0], 0, U[NL - 1]); LK_LDX(x1[0], at[0], 64, U[NL - 1]);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int32_t Y[NL];
            if constexpr (decltype(NEXT)::value) {                               // the raw loads are issued behind ys[2]: 6 more loads in flight behind every wait
                if (t == 0) { fetch_y(ys[2], yo[2], yb[2] + 2 * LK_VAL_BYTES, U[NL - 1]); fetch_raw(raw, nii, U[NL - 1]); LK_ARRIVED_Y(ys[0], 14); }
                else if (t == 1) LK_ARRIVED_Y(ys[1], 10);
                else LK_ARRIVED_Y(ys[2], 6);
            } else {
                if (t == 0) { fetch_y(ys[2], yo[2], yb[2] + 2 * LK_VAL_BYTES, U[NL - 1]); LK_ARRIVED_Y(ys[0], 8); }
                else if (t == 1) LK_ARRIVED_Y(ys[1], 4);
                else LK_ARRIVED_Y(ys[2], 0);
            }
            limbs_of(ys[t], Y);
            LK_LDX(x0[1], at[t], 16, U[NL - 1]); LK_LDX(x1[1], at[t], 80, U[NL - 1]); LK_ARRIVED_X(x0[0], x1[0], 2);
            mads1(U, 0, x0[0] + x1[0], Y); LK_TIE(U);
            LK_LDX(x0[0], at[t], 32, U[NL - 1]); LK_LDX(x1[0], at[t], 96, U[NL - 1]); LK_ARRIVED_X(x0[1], x1[1], 2);
            mads1(U, 1, x0[1] + x1[1], Y); LK_TIE(U);
            { lk_v2i a2, b2;
              LK_LDX2(a2, at[t], 48, U[NL - 1]); LK_LDX2(b2, at[t], 112, U[NL - 1]); LK_ARRIVED_X(x0[0], x1[0], 2);
              mads1(U, 2, x0[0] + x1[0], Y); LK_TIE(U);
              if (t < 2) { LK_LDX(x0[0], at[t + 1], 0, U[NL - 1]); LK_LDX(x1[0], at[t + 1], 64, U[NL - 1]); LK_ARRIVED_X(a2, b2, 2); }
              else LK_ARRIVED_X(a2, b2, 0);
              const lk_v4i s4{a2.x + b2.x, a2.y + b2.y, 0, 0};
              mads1(U, 3, s4, Y); LK_TIE(U); }
        }
        reduce_cols(U, im); LK_TIE14(im);
    };

    // ---- software pipeline over the lines of this group: the y-cache has two halves; the half of line it + 1 is built between the two outputs of line it
    // (its raw coefficient fetched during the first output, its stores complete long before the next iteration reads them)
    std::integral_constant<bool, true> yes; std::integral_constant<bool, false> no;
    lk_v4i raw[6];
    LkBuf p0, p1;                                                               // the first pair of line operands of the next iteration's first output
    int64_t pin0 = 0;
    {
        const uint32_t i0 = group_index();
        fetch_raw(raw, (active && i0 < M) ? i0 : 0, pin0);
        build_y(raw, ycw);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        asm volatile("" ::: "memory");
        fetch_y(p0, g16, ycw, pin0); fetch_y(p1, g16, ycw + LK_VAL_BYTES, pin0);
    }
#pragma unroll 1
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t i = group_index() + it * T;
        const bool valid = active && i < M;
        const uint32_t inext = i + T;
        const uint32_t nii = (active && inext < M) ? inext : 0;
        const char* ych = ycw + (it & 1) * LK_WAVE_BYTES;                        // this line's half; the next line's: ycn
        char* ycn = ycw + ((it & 1) ^ 1) * LK_WAVE_BYTES;
        int32_t o1r[NL], o1i[NL], o0r[NL], o0i[NL];
        {   // k = j + 3: (f_(j+3), f_(j+1), f_j) against the plain line (l0, l1, l2)
            const int sl[3] = {(int)j + 3, (int)j + 1, (int)j};
            const uint32_t yo[3] = {g16, g16, g16};
            const char* const yb[3] = {ych, ych + 3 * LK_VAL_BYTES, ych + 9 * LK_VAL_BYTES};
            one_output(yes, yes, sl, yo, yb, p0, p1, raw, nii, o1r, o1i);
        }
        build_y(raw, ycn);                                                       // (past the last line: line 0 again, never read)
        {   // k = j: (f_j, f_(j+4 mod 6), f_(j+3)) against (l0, xi^[j<2] l1, xi l2)
            const int sl[3] = {(int)j, (int)((j + 4) % 6), (int)j + 3};
            const uint32_t yo[3] = {g16, yv_t1, g16};
            const char* const yb[3] = {ych, ych + 3 * LK_VAL_BYTES, ych + 12 * LK_VAL_BYTES};
            one_output(no, no, sl, yo, yb, p0, p1, raw, nii, o0r, o0i);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the next half's stores have left this wave
        __syncthreads();                                                         // every lane of the group has read the old coefficients; the next half is visible
        asm volatile("" ::: "memory");
        fetch_y(p0, g16, ycn, pin0); fetch_y(p1, g16, ycn + LK_VAL_BYTES, pin0);
        if (valid) { st_limbs(j, 0, o0r); st_limbs(j, 1, o0i); st_limbs(j + 3, 0, o1r); st_limbs(j + 3, 1, o1i); }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(p0.a), "+v"(p1.a));                   // (the last prefetch is not used)
    // write the group's accumulator: chunk c of the Fp12 in TOWER order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) = w-index (0, 2, 4, 1, 3, 5)
    if (active) {
        uint4* __restrict__ prow = partials + row * 36 * T;
        const uint32_t tt = group_index();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = j + 3 * u;
            const int tower = (k & 1) ? 3 + (k >> 1) : (k >> 1);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                // a value in (-0.01 p, 1.01 p) with a signed top limb: + p, carry-propagate, then at most two subtractions of p
                uint4 q[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) q[c] = acc[LK_CHUNK(k, part) + c];
                const uint32_t* wq = reinterpret_cast<const uint32_t*>(q);
                Fq<FQ_LN, 3> v; uint32_t cy = 0;
#pragma unroll
                for (int i = 0; i < NL - 1; ++i) { const uint32_t tq = wq[i] + P28.l[i] + cy; v.l[i] = tq & MASK; cy = tq >> W; }
                v.l[NL - 1] = wq[NL - 1] + P28.l[NL - 1] + cy;                   // (two's complement: the signed top limb + p's top limb is non-negative)
                auto sub_p = [&](const Fq<FQ_LN, 3>& a) { uint32_t d[NL]; uint32_t bo = 0;
#pragma unroll
                    for (int i = 0; i < NL; ++i) { const uint32_t tq = a.l[i] - P28.l[i] - bo; bo = tq >> 31; d[i] = (i < NL - 1) ? (tq & MASK) : tq; }
                    Fq<FQ_LN, 3> r;
#pragma unroll
                    for (int i = 0; i < NL; ++i) r.l[i] = bo ? a.l[i] : d[i];
                    return r; };
                const auto v1 = sub_p(v), v2 = sub_p(v1);
                Fqn cf;
#pragma unroll
                for (int i = 0; i < NL; ++i) cf.l[i] = v2.l[i];
                uint32_t w[12]; fq_pack(cf, w);
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) prow[(uint32_t)(tower * 6 + part * 3 + qq) * T + tt] = uint4{w[4 * qq], w[4 * qq + 1], w[4 * qq + 2], w[4 * qq + 3]};
            }
        }
    }
#undef LK_ARRIVED_Y
#undef LK_LDX
#undef LK_LDX2
#undef LK_ARRIVED_X
#endif
}
#undef LK_TIE
#undef LK_CHUNK
#undef LK_TIE14
#endif  // !RIPP_BLS12_377

}  // namespace ripp
