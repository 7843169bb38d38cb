// Stage 2a of the pairing product, Karatsuba form (k_line_products_k): per (product, Miller step) row, the product of the sparse line elements of all pairs,
// like k_line_products_q (fq_line_products.hpp: the Fp12 accumulator in LDS, ONE Montgomery reduction per real / imaginary part of an output coefficient)
// with a QUARTER fewer multiply-adds per line:
//
//   out_k = sum_t x_t y_t  (three Fp2 products, u^2 = -1):   A = sum x_t0 y_t0,  B = sum x_t1 y_t1,  C = sum (x_t0 + x_t1)(y_t0 + y_t1)
//   re = A - B,   im = C - A - B                             9 products + 2 reductions (2 156 multiply-adds) where the six-product sums take 12 + 2 (2 744).
//
// Lazy Karatsuba needs TWO column sets alive (A and B: 108 registers) and operands in the form they multiply with.  What makes that fit:
//   * SIGNED limbs and columns (v_mad_i64_i32): a value is sum l_i 2^(28 i) with |l_i| < 2^29, so a negation is limb-wise, a difference needs no bias and no
//     normalisation, and with U = -A, V = -B:  re = reduce(V - U), im = reduce(U + V + C) by 64-bit adds of whole columns.  The Montgomery reduction is exact on
//     signed columns (low limbs by masking the two's complement, carries by arithmetic shifts); results lie in (-0.01 p, 1.01 p) and are made canonical once,
//     when the kernel writes its partial products.
//   * SIX lanes per accumulator, one output coefficient per lane (k_line_products_q: three lanes, two outputs each): 10 accumulators per wave instead of 21
//     leave half of the wave's LDS share free, and the LINE side moves in: per line the six lanes unpack the three coefficients once and write -c0, -c1,
//     c0 + c1 and, for the coefficients that meet a wrapped-around index (w^6 = xi = 1 + u), xi's images c1 - c0, -(c0 + c1), 2 c0 -- 15 values of 14 limbs --
//     next to the accumulator.  The product loops then read every operand from LDS in the form it is used in, one term ahead: no unpacking, no xi arithmetic
//     and no negation inside the loops, and two line operands (28 registers) resident at a time instead of six (84).
//     (A first version kept three lanes per accumulator and put these values into a per-wave cache in global memory: bit-exact, but 2 040 waves x 40 KB do not
//     stay in the 4 MB L2s -- 11.0 ms per launch against 11.2 ms for k_line_products_q, 10.0 ms when every wave was pointed at one cached copy.
//     tools/ubench/lp_k3_global_ycache.hpp keeps that version.)
//   * lanes are numbered k-major (lane = 10 k + group): the ten lanes that read the same slot of ten different accumulators are neighbours, and with an
//     accumulator stride of 65 and a slot pitch of 10 sixteen-byte units (line values: stride 61) a ds_read_b128 of 16 lanes touches every bank once in 9 of the
//     12 (term, phase) patterns and twice in 3 (searched: tools/ubench/lds_layout_k6.py).
//   * the raw line of the NEXT iteration is fetched (by hand, 24 registers) when this iteration's products start: HBM latency is off the critical path.
// Per line and accumulator: 6 x ~2 800 instructions (2 156 multiply-adds) against 3 x ~6 900 (5 492).  60 of 64 lanes work.
// BLS12-381 only (u^2 = -1, xi = 1 + u, M-type line l0 + l1 w^2 + l2 w^3); the BLS12-377 build keeps k_line_products_q.
#pragma once
#include <type_traits>
#include "fq_line_products.hpp"

namespace ripp {

constexpr int LK_GROUP = 6;                                           // lanes per accumulator
constexpr int LK_GROUPS_PER_WAVE = 10;                                // 60 of 64 lanes
constexpr int LK_ACC_STRIDE = 65;                                     // accumulators: 16-byte units between groups
constexpr int LK_SLOT_PITCH = 10;                                     // f_k: real part at unit 10 k, imaginary part 4 units later
constexpr int LK_Y_STRIDE = 61;                                       // line values: 15 x 4 units per group + 1
constexpr int LK_Y_BASE = LK_GROUPS_PER_WAVE * LK_ACC_STRIDE;         // first unit of the line values
constexpr int LK_LDS_UNITS = LK_Y_BASE + LK_GROUPS_PER_WAVE * LK_Y_STRIDE;      // 1 260 units = 20 160 B per wave (8 waves: 157.5 of the CU's 160 KB)

#if !defined(RIPP_BLS12_377)
typedef int32_t lk_v4i __attribute__((ext_vector_type(4)));
typedef int32_t lk_v2i __attribute__((ext_vector_type(2)));
struct LkBuf { lk_v4i a, b, c; lk_v2i d; };                           // a 14-limb operand as fetched

#define LK_TIE(c) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), \
                                   "+v"(c[14]), "+v"(c[15]), "+v"(c[16]), "+v"(c[17]), "+v"(c[18]), "+v"(c[19]), "+v"(c[20]), "+v"(c[21]), "+v"(c[22]), "+v"(c[23]), "+v"(c[24]), "+v"(c[25]), "+v"(c[26]))
#define LK_TIE14(c) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]))

#if defined(__HIP_DEVICE_COMPILE__)
// LDS reads by hand (the compiler would hoist every fetch and keep all operands alive).  `pin`: a column of the running sums, which orders a fetch BEFORE
// the multiply-adds that follow it.  LDS operations return in order: "lgkmcnt(n)" with n = the number of reads issued after the wanted one is exact.
template <int OFF> __device__ __forceinline__ void lk_ld_val(LkBuf& b, uint32_t addr, int64_t& pin) {          // a 14-limb value: 3 x 16 + 8 bytes
    asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %5 offset:%7\n\tds_read_b128 %2, %5 offset:%8\n\tds_read_b64 %3, %5 offset:%9"
                 : "=&v"(b.a), "=&v"(b.b), "=&v"(b.c), "=&v"(b.d), "+v"(pin) : "v"(addr), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48));
}
template <int OFF> __device__ __forceinline__ void lk_ld_x(lk_v4i& x0, lk_v4i& x1, uint32_t addr, int64_t& pin) {   // limbs 4q .. 4q + 3 of the real and the imaginary part
    asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5" : "=&v"(x0), "=&v"(x1), "+v"(pin) : "v"(addr), "n"(OFF), "n"(OFF + 64));
}
template <int OFF> __device__ __forceinline__ void lk_ld_x2(lk_v2i& x0, lk_v2i& x1, uint32_t addr, int64_t& pin) {  // limbs 12, 13
    asm volatile("ds_read_b64 %0, %3 offset:%4\n\tds_read_b64 %1, %3 offset:%5" : "=&v"(x0), "=&v"(x1), "+v"(pin) : "v"(addr), "n"(OFF), "n"(OFF + 64));
}
// fetch + wait in ONE statement (hipcc puts an s_nop behind every inline-asm statement): issue the next reads, then wait until at most N reads are in flight --
// i.e. until the CURRENT operands (cx0, cx1, and a 14-limb value or two) have arrived
template <int OFF, int N> __device__ __forceinline__ void lk_ld_x_w(lk_v4i& nx0, lk_v4i& nx1, uint32_t addr, int64_t& pin, lk_v4i& cx0, lk_v4i& cx1) {
    asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %5 offset:%7\n\ts_waitcnt lgkmcnt(%8)"
                 : "=&v"(nx0), "=&v"(nx1), "+v"(pin), "+v"(cx0), "+v"(cx1) : "v"(addr), "n"(OFF), "n"(OFF + 64), "n"(N));
}
template <int OFF, int N> __device__ __forceinline__ void lk_ld_x2_w(lk_v2i& nx0, lk_v2i& nx1, uint32_t addr, int64_t& pin, lk_v4i& cx0, lk_v4i& cx1) {
    asm volatile("ds_read_b64 %0, %5 offset:%6\n\tds_read_b64 %1, %5 offset:%7\n\ts_waitcnt lgkmcnt(%8)"
                 : "=&v"(nx0), "=&v"(nx1), "+v"(pin), "+v"(cx0), "+v"(cx1) : "v"(addr), "n"(OFF), "n"(OFF + 64), "n"(N));
}
template <int OFF, int N> __device__ __forceinline__ void lk_ld_x_w2(lk_v4i& nx0, lk_v4i& nx1, uint32_t addr, int64_t& pin, lk_v2i& cx0, lk_v2i& cx1) {      // (waits for the two-limb chunk)
    asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %5 offset:%7\n\ts_waitcnt lgkmcnt(%8)"
                 : "=&v"(nx0), "=&v"(nx1), "+v"(pin), "+v"(cx0), "+v"(cx1) : "v"(addr), "n"(OFF), "n"(OFF + 64), "n"(N));
}
// (a sum that is an ordinary expression lets LLVM re-associate  col += x y  into a chain of dependent multiply-adds on a temporary plus one 64-bit add per column)
#define LK_OPAQUE(X_) asm volatile("" : "+v"(X_))
#define LK_WAIT_X(n, XA_, XB_) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(XA_), "+v"(XB_))
#define LK_WAIT_V(n, BUF_) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(BUF_.a), "+v"(BUF_.b), "+v"(BUF_.c), "+v"(BUF_.d))
__device__ __forceinline__ void lk_limbs_of(const LkBuf& b, int32_t (&y)[fq28::NL]) {
    y[0] = b.a.x; y[1] = b.a.y; y[2] = b.a.z; y[3] = b.a.w; y[4] = b.b.x; y[5] = b.b.y; y[6] = b.b.z; y[7] = b.b.w;
    y[8] = b.c.x; y[9] = b.c.y; y[10] = b.c.z; y[11] = b.c.w; y[12] = b.d.x; y[13] = b.d.y;
}
// signed Montgomery reduction of 27 columns (destroys them): r = (sum col_k 2^(28 k) + m p) / 2^392, limbs 0..12 in [0, 2^28), limb 13 signed
__device__ __forceinline__ void lk_reduce_cols(int64_t (&col)[2 * fq28::NL - 1], int32_t (&r)[fq28::NL]) {
    using namespace fq28;
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
}
// rows 4 q .. 4 q + 3 of two products at once: U += x0 * Y0, V += x1 * Y1
template <int Q> __device__ __forceinline__ void lk_mads2(int64_t (&U)[2 * fq28::NL - 1], int64_t (&V)[2 * fq28::NL - 1], const lk_v4i& x0, const lk_v4i& x1, const int32_t (&Y0)[fq28::NL], const int32_t (&Y1)[fq28::NL]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = 4 * Q + u;
        if (i >= fq28::NL) continue;
#pragma unroll
        for (int jj = 0; jj < fq28::NL; ++jj) { U[i + jj] += (int64_t)x0[u] * Y0[jj]; V[i + jj] += (int64_t)x1[u] * Y1[jj]; }
    }
}
template <int Q> __device__ __forceinline__ void lk_mads1(int64_t (&Wc)[2 * fq28::NL - 1], const lk_v4i& xs, const int32_t (&Y)[fq28::NL]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = 4 * Q + u;
        if (i >= fq28::NL) continue;
#pragma unroll
        for (int jj = 0; jj < fq28::NL; ++jj) Wc[i + jj] += (int64_t)xs[u] * Y[jj];
        LK_TIE(Wc);                                                              // (per ROW: a column that receives several rows of one step is otherwise re-associated into
    }                                                                            //  a chain of dependent multiply-adds on a temporary plus a 64-bit add)
}
#endif

// grid = (ceil(T / 10), rows), block = 64 (one wave); arguments and output layout as k_line_products_q (T accumulators per row)
__global__ void __launch_bounds__(64, 2) k_line_products_k(const uint4* __restrict__ lines, size_t stride, uint32_t M, uint4* __restrict__ partials, uint32_t T) {
    __shared__ uint4 lds[LK_LDS_UNITS];
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace fq28;
    const uint32_t lane = threadIdx.x;
    const uint32_t kq = lane / LK_GROUPS_PER_WAVE, g = lane - kq * LK_GROUPS_PER_WAVE;       // k-major: lanes 10 k .. 10 k + 9 hold output k of the ten groups; lanes 60..63 idle
    const bool inwave = kq < (uint32_t)LK_GROUP;
    const uint32_t k = inwave ? kq : 0;
    auto group_index = [&]() { uint32_t z = 0; asm volatile("" : "+s"(z));
        const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
        return blockIdx.x * LK_GROUPS_PER_WAVE + l % LK_GROUPS_PER_WAVE; };
    const bool active = inwave && blockIdx.x * LK_GROUPS_PER_WAVE + g < T;
    const size_t row = blockIdx.y;
    const uint32_t lds_base = (uint32_t)(size_t)lds;
    const uint32_t ga = active ? g : 0, gy = inwave ? g : 0;
    // this lane's operands: term 0 = f_k l0, term 1 = f_(k+4) [xi] l1 (xi if k < 2), term 2 = f_(k+3) [xi] l2 (xi if k < 3); indices mod 6
    const uint32_t ax0 = lds_base + (ga * LK_ACC_STRIDE + k * LK_SLOT_PITCH) * 16;
    const uint32_t ax1 = lds_base + (ga * LK_ACC_STRIDE + ((k + 4) % 6) * LK_SLOT_PITCH) * 16;
    const uint32_t ax2 = lds_base + (ga * LK_ACC_STRIDE + ((k + 3) % 6) * LK_SLOT_PITCH) * 16;
    // line values of a group, 4 units each: l0: 0 (-c0), 1 (-c1), 2 (c0 + c1); l1: 3, 4, 5 and xi l1: 6 (c1 - c0), 7 (-(c0 + c1)), 8 (2 c0); l2: 9, 10, 11 and xi l2: 12, 13, 14
    const uint32_t ayg = lds_base + (LK_Y_BASE + gy * LK_Y_STRIDE) * 16;
    const uint32_t ay0 = ayg;
    const uint32_t ay1 = ayg + (3 + (k < 2 ? 3 : 0)) * 64;
    const uint32_t ay2 = ayg + (9 + (k < 3 ? 3 : 0)) * 64;
    uint4* xslot = lds + ga * LK_ACC_STRIDE + k * LK_SLOT_PITCH;                // this lane's output coefficient
    auto st_limbs = [](uint4* dst, const int32_t (&v)[NL]) {
        dst[0] = uint4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
        dst[1] = uint4{(uint32_t)v[4], (uint32_t)v[5], (uint32_t)v[6], (uint32_t)v[7]};
        dst[2] = uint4{(uint32_t)v[8], (uint32_t)v[9], (uint32_t)v[10], (uint32_t)v[11]};
        dst[3] = uint4{(uint32_t)v[12], (uint32_t)v[13], 0u, 0u}; };
    if (active) {                                                               // accumulator <- 1
        int32_t one[NL], zero[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) { one[i] = (int32_t)ONE_M392.l[i]; zero[i] = 0; }
        if (k == 0) st_limbs(xslot, one); else st_limbs(xslot, zero);
        st_limbs(xslot + 4, zero);
    }
    const uint32_t st = (uint32_t)stride;
    const uint4* __restrict__ lrow = lines + row * 18 * stride;
    const uint32_t iters = (M + T - 1) / T;
    // the line side: lane (g, k) works on coefficient tc = k mod 3 of its group's line, half hc = k div 3 of the values derived from it
    const uint32_t tc = k % 3, hc = k / 3;
    // the six 16-byte chunks of coefficient l_tc of line `ii` (c0: raw[0..2], c1: raw[3..5]), as stage 1 stored them
    auto fetch_raw = [&](lk_v4i (&raw)[6], uint32_t ii) {
        uint32_t off[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) off[c] = ((6 * tc + (uint32_t)c) * st + ii) << 4;
        asm volatile("global_load_dwordx4 %0, %6, %12\n\tglobal_load_dwordx4 %1, %7, %12\n\tglobal_load_dwordx4 %2, %8, %12\n\tglobal_load_dwordx4 %3, %9, %12\n\tglobal_load_dwordx4 %4, %10, %12\n\tglobal_load_dwordx4 %5, %11, %12"
                     : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(raw[2]), "=&v"(raw[3]), "=&v"(raw[4]), "=&v"(raw[5])
                     : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(lrow));
    };
    auto build_y = [&](lk_v4i (&raw)[6]) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]));
        uint32_t w0[12], w1[12];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            w0[4 * c] = (uint32_t)raw[c].x; w0[4 * c + 1] = (uint32_t)raw[c].y; w0[4 * c + 2] = (uint32_t)raw[c].z; w0[4 * c + 3] = (uint32_t)raw[c].w;
            w1[4 * c] = (uint32_t)raw[3 + c].x; w1[4 * c + 1] = (uint32_t)raw[3 + c].y; w1[4 * c + 2] = (uint32_t)raw[3 + c].z; w1[4 * c + 3] = (uint32_t)raw[3 + c].w;
        }
        const Fqn c0 = fq_unpack(w0), c1 = fq_unpack(w1);                        // stage 1 stores canonical values (< p)
        uint4* dst = lds + LK_Y_BASE + gy * LK_Y_STRIDE + (tc == 0 ? 0 : tc == 1 ? 3 : 9) * 4;
        int32_t v[NL];
        if (!inwave) return;
        if (hc == 0) {
#pragma unroll
            for (int q = 0; q < NL; ++q) v[q] = -(int32_t)c0.l[q];
            st_limbs(dst, v);
            if (tc != 0) {                                                       // xi (c0 + c1 u) = (c0 - c1) + (c0 + c1) u:  -(c0 - c1) and the sum of the two parts, 2 c0
#pragma unroll
                for (int q = 0; q < NL; ++q) v[q] = (int32_t)c1.l[q] - (int32_t)c0.l[q];
                st_limbs(dst + 3 * 4, v);
#pragma unroll
                for (int q = 0; q < NL; ++q) {                                   // 2 c0: limb q = bits [28 q - 1, 28 q + 27) of c0's integer
                    uint32_t x;
                    if (q == 0) x = w0[0] << 1;
                    else { const int bit = W * q - 1, ww = bit >> 5, sh = bit & 31;
                           if (sh + W <= 32 || ww + 1 >= 12) x = w0[ww] >> sh; else x = __builtin_amdgcn_alignbit(w0[ww + 1], w0[ww], sh); }
                    v[q] = (int32_t)(x & MASK);
                }
                st_limbs(dst + 5 * 4, v);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NL; ++q) v[q] = -(int32_t)c1.l[q];
            st_limbs(dst + 1 * 4, v);
            const auto s = fq_norm(fq_add(c0, c1));                              // c0 + c1, limbs < 2^28
#pragma unroll
            for (int q = 0; q < NL; ++q) v[q] = (int32_t)s.l[q];
            st_limbs(dst + 2 * 4, v);
            if (tc != 0) {
#pragma unroll
                for (int q = 0; q < NL; ++q) v[q] = -(int32_t)s.l[q];
                st_limbs(dst + 4 * 4, v);
            }
        }
    };

    lk_v4i raw[6];
    { const uint32_t i0 = group_index(); fetch_raw(raw, (active && i0 < M) ? i0 : 0); }
#pragma unroll 1
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t i = group_index() + it * T;
        const bool valid = active && i < M;
        build_y(raw);
        __syncthreads();
        asm volatile("" ::: "memory");
        { const uint32_t inext = i + T; fetch_raw(raw, (active && inext < M) ? inext : 0); }      // the next line's coefficient travels while this one is multiplied
        int32_t re[NL], im[NL];
        {
            int64_t U[2 * NL - 1], V[2 * NL - 1];
#pragma unroll
            for (int c = 0; c < 2 * NL - 1; ++c) { U[c] = 0; V[c] = 0; }
            LkBuf ya[2], yc[2];
            lk_v4i x0[2], x1[2];
            lk_v2i xa, xb;
            lk_ld_val<0>(ya[0], ay0, U[NL - 1]); lk_ld_val<64>(yc[0], ay0, U[NL - 1]);
            lk_ld_x<0>(x0[0], x1[0], ax0, U[NL - 1]);
            // ---- U = sum x_t0 (-y_t0) = -A,  V = sum x_t1 (-y_t1) = -B.  In flight behind the first wait of a term: the next term's reads (NW_ of them, chunk pair included)
#define LK_AB_TERM(P_, AX_, NEXT_, NW_)                                                                                                   \
            { int32_t Y0[NL], Y1[NL];                                                                                                 \
              NEXT_                                                                                                                    \
              lk_ld_x_w<16, NW_>(x0[1], x1[1], AX_, U[NL - 1], x0[0], x1[0]); LK_WAIT_V(NW_, ya[P_]); LK_WAIT_V(NW_, yc[P_]);          \
              lk_limbs_of(ya[P_], Y0); lk_limbs_of(yc[P_], Y1);                                                                        \
              lk_mads2<0>(U, V, x0[0], x1[0], Y0, Y1); LK_TIE(U); LK_TIE(V);                                                           \
              lk_ld_x_w<32, 2>(x0[0], x1[0], AX_, U[NL - 1], x0[1], x1[1]);                                                            \
              lk_mads2<1>(U, V, x0[1], x1[1], Y0, Y1); LK_TIE(U); LK_TIE(V);                                                           \
              lk_ld_x2_w<48, 2>(xa, xb, AX_, U[NL - 1], x0[0], x1[0]);                                                                 \
              lk_mads2<2>(U, V, x0[0], x1[0], Y0, Y1); LK_TIE(U); LK_TIE(V);
#define LK_AB_END(AXN_)                                                                                                                   \
              lk_ld_x_w2<0, 2>(x0[0], x1[0], AXN_, U[NL - 1], xa, xb);                                                                 \
              { const lk_v4i a4{xa.x, xa.y, 0, 0}, b4{xb.x, xb.y, 0, 0}; lk_mads2<3>(U, V, a4, b4, Y0, Y1); } LK_TIE(U); LK_TIE(V); }
            LkBuf ys[2];
            LK_AB_TERM(0, ax0, lk_ld_val<0>(ya[1], ay1, U[NL - 1]); lk_ld_val<64>(yc[1], ay1, U[NL - 1]);, 10) LK_AB_END(ax1)
            LK_AB_TERM(1, ax1, lk_ld_val<0>(ya[0], ay2, U[NL - 1]); lk_ld_val<64>(yc[0], ay2, U[NL - 1]);, 10) LK_AB_END(ax2)
            // (the last term fetches the first sum for C and, behind its last chunk, the first chunk pair for C)
            LK_AB_TERM(0, ax2, lk_ld_val<128>(ys[0], ay0, U[NL - 1]);, 6) LK_AB_END(ax0)
#undef LK_AB_TERM
#undef LK_AB_END
            // ---- re = reduce(V - U), and U <- U + V
#pragma unroll
            for (int c = 0; c < 2 * NL - 1; ++c) { const int64_t u = U[c], v = V[c]; U[c] = u + v; V[c] = v - u; }
            LK_TIE(V); LK_TIE(U);
            lk_reduce_cols(V, re); LK_TIE14(re);
            // ---- U = -A - B + sum (x_t0 + x_t1)(y_t0 + y_t1)
#define LK_C_TERM(P_, AX_, NEXT_, NW_)                                                                                                    \
            { int32_t Y[NL];                                                                                                          \
              NEXT_                                                                                                                    \
              lk_ld_x_w<16, NW_>(x0[1], x1[1], AX_, U[NL - 1], x0[0], x1[0]); LK_WAIT_V(NW_, ys[P_]);                                  \
              lk_limbs_of(ys[P_], Y);                                                                                                  \
              lk_mads1<0>(U, x0[0] + x1[0], Y);                                                                                        \
              lk_ld_x_w<32, 2>(x0[0], x1[0], AX_, U[NL - 1], x0[1], x1[1]);                                                            \
              lk_mads1<1>(U, x0[1] + x1[1], Y);                                                                                        \
              lk_ld_x2_w<48, 2>(xa, xb, AX_, U[NL - 1], x0[0], x1[0]);                                                                 \
              lk_mads1<2>(U, x0[0] + x1[0], Y);
#define LK_C_END_NEXT(AXN_)                                                                                                               \
              lk_ld_x_w2<0, 2>(x0[0], x1[0], AXN_, U[NL - 1], xa, xb);                                                                 \
              { const lk_v4i s4{xa.x + xb.x, xa.y + xb.y, 0, 0}; lk_mads1<3>(U, s4, Y); } }
#define LK_C_END_LAST()                                                                                                                   \
              LK_WAIT_X(0, xa, xb);                                                                                                    \
              { const lk_v4i s4{xa.x + xb.x, xa.y + xb.y, 0, 0}; lk_mads1<3>(U, s4, Y); } }
            LK_C_TERM(0, ax0, lk_ld_val<128>(ys[1], ay1, U[NL - 1]);, 6) LK_C_END_NEXT(ax1)
            LK_C_TERM(1, ax1, lk_ld_val<128>(ys[0], ay2, U[NL - 1]);, 6) LK_C_END_NEXT(ax2)
            LK_C_TERM(0, ax2, , 2) LK_C_END_LAST()
#undef LK_C_TERM
#undef LK_C_END_NEXT
#undef LK_C_END_LAST
            lk_reduce_cols(U, im); LK_TIE14(im);
        }
        __syncthreads();                                                         // every lane of the group has read the old coefficients and the line values
        if (valid) { st_limbs(xslot, re); st_limbs(xslot + 4, im); }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]));      // (the last prefetch is not used)
    // write the accumulator: this lane's coefficient, chunk c of the Fp12 in TOWER order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) = w-index (0, 2, 4, 1, 3, 5)
    if (active) {
        uint4* __restrict__ prow = partials + row * 36 * T;
        const uint32_t tt = group_index();
        const uint32_t tower = (k & 1) ? 3 + (k >> 1) : (k >> 1);
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            // a value in (-0.01 p, 1.01 p) with a signed top limb: + p, carry-propagate, then at most two subtractions of p
            uint4 q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = xslot[part * 4 + c];
            const uint32_t* wq = reinterpret_cast<const uint32_t*>(q);
            Fq<FQ_LN, 3> v; uint32_t cy = 0;
#pragma unroll
            for (int i = 0; i < NL - 1; ++i) { const uint32_t tq = wq[i] + P28.l[i] + cy; v.l[i] = tq & MASK; cy = tq >> W; }
            v.l[NL - 1] = wq[NL - 1] + P28.l[NL - 1] + cy;                       // (two's complement: the signed top limb + p's top limb is non-negative)
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
            for (int qq = 0; qq < 3; ++qq) prow[(tower * 6 + (uint32_t)part * 3 + (uint32_t)qq) * T + tt] = uint4{w[4 * qq], w[4 * qq + 1], w[4 * qq + 2], w[4 * qq + 3]};
        }
    }
#endif
}
#undef LK_WAIT_X
#undef LK_OPAQUE
#undef LK_WAIT_V
#undef LK_TIE
#undef LK_TIE14
#endif  // !RIPP_BLS12_377

}  // namespace ripp
