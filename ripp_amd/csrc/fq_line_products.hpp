// Stage 2a of the pairing product on the carry-free field form (fq28.hpp): the throughput twin of line_products.hpp's k_line_products.
// Same decomposition -- the Fp12 accumulator lives in LDS, owned by a group of 3 lanes, lane j produces the w-basis coefficients j and j + 3
// of f * (l0 + l1 w^2 + l2 w^3), each real / imaginary part as ONE lazily reduced sum of six products -- but
//   * a limb product is ONE v_mad_u64_u32 into a 64-bit column (no carry instruction): 4 x (6 x 196 + 196) = 5 488 multiply-adds per line and
//     lane instead of 4 x 7 x 288 = 8 064 multiply-adds PLUS as many carry instructions;
//   * the negation of the real part and the factor xi = 1 + u are limb-wise lazy operations on the operands (no carry chains);
//   * the accumulator needs 6 LDS slots, not 9: xi is applied to the LINE coefficient (f xi . l = f . xi l), once per line.
// The line buffer is read as it is: the 12 x u32 words of a Montgomery-384 value, re-sliced into 28-bit limbs, ARE the Montgomery-392 form
// of that value times 2^-8 -- every line, hence every per-step product, is merely scaled by an element of Fp, which the final exponentiation
// removes (the same argument that lets stage 1 scale its lines, bls12_381/pairing.hpp).
// Both curves: BLS12-381 (M-type twist: line = l0 + l1 w^2 + l2 w^3, xi = 1 + u, u^2 = -1) and BLS12-377 (D-type: l0 + l1 w + l2 w^3, xi = u,
// u^2 = -5 -- the factor 5 of a real part is a transient, normalised multiple of the LINE operand; line_products.hpp has the operand tables).
#pragma once
#include <hip/hip_runtime.h>
#include "fq_curve2.hpp"
#include "line_products.hpp"

namespace ripp {

// pins the multiply-adds of one operand pass BEFORE the next pass's LDS reads are laundered (an empty asm over all 27 columns): without it the
// compiler sinks the arithmetic below every read and keeps the fetched operands -- all of them -- live
#define TIE(c) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), \
                                "+v"(c[14]), "+v"(c[15]), "+v"(c[16]), "+v"(c[17]), "+v"(c[18]), "+v"(c[19]), "+v"(c[20]), "+v"(c[21]), "+v"(c[22]), "+v"(c[23]), "+v"(c[24]), "+v"(c[25]), "+v"(c[26]))
// the same for a reduced sum: the reduction happens HERE (14 live registers), not where the value is stored (54)
#define TIE14(c) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]))
typedef uint32_t lq_v4u __attribute__((ext_vector_type(4)));
struct LqBuf { lq_v4u q[4]; };                                    // one accumulator coefficient as fetched: 14 limbs + 2
constexpr int LQ_SLOT_DW = 16;                                     // an Fq in LDS: 14 limbs + 2 (four 16-byte accesses)
constexpr int LQ_ACC_DW = 6 * 2 * LQ_SLOT_DW;                      // f_0 .. f_5 in Fp2: 768 B per accumulator
// LDS bank layout.  An accumulator is six 128-byte slots; the three lanes of a group read three different slots in the SAME instruction, and 21 groups do so
// side by side.  Two things had to be kept apart modulo the 256-byte bank row (64 banks x 4 B):
//   * the groups: with a stride of 768 B every group's slot k sat in the same banks (SQ_LDS_BANK_CONFLICT 20 % of the kernel's wave-cycles, first r03
//     PMC pass); an odd stride of 49 chunks took that to 4.7 %, the hand-streamed ds_read_b128 fetches of the second half of build round 3 back to 9.5 %
//     (profiles/r03_sq_counters_pmc.csv, profiles/r04_sq_counters_pmc.csv before this layout);
//   * the LANES OF ONE GROUP: lanes 0 and 2 read slots k and k + 2 -- 256 B apart, i.e. the same banks, a two-way conflict on every fetch.
// A ds_read_b128 is served 16 lanes (256 B) per cycle; a search over slot pitches and group strides (16 fetch patterns x phases of 16 lanes, every lane a
// distinct 16-byte bank quad wanted) gives a slot pitch of 144 B (slot k at chunk 9 k: one chunk of skew per slot) and a group stride of 59 chunks
// (944 B = 3 x 256 + 176): 4 serialised cycles over the 16 patterns where the 784-byte layout has 16 and the 768-byte one 152.  19.8 KB of LDS per wave,
// 158.6 of the CU's 160 KB at 8 waves.
// Same-box A/B in build round 3 (-DRIPP_LP_NO_PAD, the 768-byte layout): 452.0 against 454.8 ms per proof -- most conflict cycles hide behind the SIMD's other wave.
#if defined(RIPP_LP_NO_PAD)
constexpr int LQ_ACC_STRIDE = LQ_ACC_DW / 4;                       // (A/B builds: the conflicting layout)
#define LQ_CHUNK(slot, part) (((slot) * 2 + (part)) * 4)
#elif defined(RIPP_LP_PAD49)
constexpr int LQ_ACC_STRIDE = LQ_ACC_DW / 4 + 1;                   // (A/B builds: build round 3's layout, 49 chunks)
#define LQ_CHUNK(slot, part) (((slot) * 2 + (part)) * 4)
#else
constexpr int LQ_ACC_STRIDE = 59;                                  // in 16-byte chunks
#define LQ_CHUNK(slot, part) ((slot) * 9 + (part) * 4)             // first chunk of (slot, part)
#endif

// grid = (ceil(T / 21), rows), block = 64 (one wave); same arguments and output layout as k_line_products
__global__ void __launch_bounds__(64, 2) k_line_products_q(const uint4* __restrict__ lines, size_t stride, uint32_t M, uint4* __restrict__ partials, uint32_t T) {
    __shared__ uint4 lds[LP_GROUPS_PER_WAVE * LQ_ACC_STRIDE];
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace fq28;
    const uint32_t lane = threadIdx.x;
    const uint32_t g = lane / LP_GROUP, j = lane - g * LP_GROUP;               // group in wave, lane in group (lane 63: g = 21, idle)
    auto group_index = [&]() { uint32_t z = 0; asm volatile("" : "+s"(z));
        const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
        return blockIdx.x * LP_GROUPS_PER_WAVE + l / LP_GROUP; };
    const uint32_t t = blockIdx.x * LP_GROUPS_PER_WAVE + g;
    const bool active = g < (uint32_t)LP_GROUPS_PER_WAVE && t < T;
    const size_t row = blockIdx.y;
    const uint32_t acc_chunk = (active ? g : 0) * LQ_ACC_STRIDE;
    const uint32_t lds_base = (uint32_t)(size_t)lds;                            // LDS byte offset of the array (the low half of its flat address)
    uint4* acc = lds + acc_chunk;
    auto ld_fq = [&](int slot, int part) { Fqn v; uint4 q[4];                    // slot = w-index 0..5, part = 0 (real) / 1 (imaginary)
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = acc[LQ_CHUNK(slot, part) + c];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
#pragma unroll
        for (int i = 0; i < NL; ++i) v.l[i] = w[i];
        return v; };
    auto st_fq = [&](int slot, int part, const Fqn& v) { uint4 q[4]; uint32_t* w = reinterpret_cast<uint32_t*>(q);
#pragma unroll
        for (int i = 0; i < NL; ++i) w[i] = v.l[i];
        w[14] = 0; w[15] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[LQ_CHUNK(slot, part) + c] = q[c]; };
    if (active) {                                                               // accumulator <- 1
        st_fq(j, 0, j == 0 ? fq_one() : fq_zero()); st_fq(j, 1, fq_zero()); st_fq(j + 3, 0, fq_zero()); st_fq(j + 3, 1, fq_zero());
    }
    __syncthreads();
#if defined(RIPP_BLS12_377)
    // D-type twist, out_k = f_k l0 + [xi] f_(k-1) l1 + [xi] f_(k-3) l2.  k = j + 3: (f_(j+3), f_(j+2), f_j) against (l0, l1, l2);
    // k = j: (f_j, f_(j+5 mod 6), f_(j+3)) against (l0, xi^[j=0] l1, xi l2)
    const int a1 = (int)((j + 5) % 6), s1b = (int)j + 2;
    const bool xi1 = j < 1;
    constexpr int FYV = 7;                                                       // a line coefficient times xi = u: (K - 5 c1, c0), value < 7p
#else
    // operand slots of this lane's outputs.  k = j + 3: (f_(j+3), f_(j+1), f_j) against (l0, l1, l2);
    // k = j: (f_j, f_(j+4 mod 6), f_(j+3)) against (l0, xi^[j<2] l1, xi l2)
    const int a1 = (int)((j + 4) % 6), s1b = (int)j + 1;
    const bool xi1 = j < 2;
    constexpr int FYV = 3;                                                       // times xi = 1 + u: (c0 - c1, c0 + c1), value < 3p
#endif
    const uint32_t st = (uint32_t)stride;
    const uint4* __restrict__ lrow = lines + row * 18 * stride;
    const uint32_t iters = (M + T - 1) / T;
    using FY = Fq<FQ_LN, FYV>;                                                   // a line coefficient (canonical in HBM), possibly times xi (normalised)
#pragma unroll 1
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t i = group_index() + it * T;
        const bool valid = active && i < M;
        const uint32_t ii = valid ? i : 0;
        FY y[6];                                                                 // l0.c0, l0.c1, l1.c0, l1.c1, l2.c0, l2.c1
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            uint4 q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lrow) + (((uint32_t)(3 * f + c) * st + ii) << 4));
            uint32_t w[12];
#pragma unroll
            for (int c = 0; c < 3; ++c) { w[4 * c] = q[c].x; w[4 * c + 1] = q[c].y; w[4 * c + 2] = q[c].z; w[4 * c + 3] = q[c].w; }
            { const Fqn u = fq_unpack(w); Fq<FQ_LN, 1> c; for (int q = 0; q < NL; ++q) c.l[q] = u.l[q]; y[f] = fq_widen<FQ_LN, FYV>(c); }      // stage 1 stores canonical values (< p)
        }
        Fqn o1r, o1i, o0r, o0i;
        // One output coefficient part = sum of six products with ONE reduction, evaluated ROW-WISE: the limbs of the accumulator operands are
        // streamed from LDS four at a time and every limb feeds 14 multiply-adds into 14 DIFFERENT 64-bit columns -- no operand array of the
        // accumulator side in registers (84 fewer live registers than the column-wise fq_dot: the kernel no longer spills) and 14
        // independent dependency chains for the multiplier.  xsel[t]: slot / part of the t-th accumulator operand, neg: take K - x
        // (K = 3p with limbs that dominate a reduced value's: fq28::sub_bias), ysel[t]: index of the line-side operand.
        static_assert(dot_fits(6, (uint64_t)1 << 29, FQ_LN) && 6 * 4 * FYV * FQ2_BETA <= VMAX, "six products of (x or K - x) by a line coefficient fit the 64-bit columns");
        constexpr Limbs KN = sub_bias<FQ_LN, 2>();
        // One accumulator coefficient (slot, part) = four 16-byte LDS reads, issued by hand one operand pass AHEAD of their use (asm: the
        // compiler merged the repeated reads of a coefficient -- each is an operand of two sums, slots j and j + 3 of four -- into one and kept all
        // 8 coefficients, 112 registers, live across the iteration, which pushed 48 dwords of the outputs into scratch; left to schedule plain
        // re-reads it placed every one directly before its use).  Streaming costs 64 more ds_read_b128 per line; the kernel has no scratch.
        // `pin`: a column the current pass accumulates into -- orders the reads BEFORE that pass's multiply-adds.
        auto fetch = [&](LqBuf& b, int slot, int part, uint64_t& pin) {
            const uint32_t addr = lds_base + (acc_chunk + (uint32_t)LQ_CHUNK(slot, part)) * 16;
            asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48"
                         : "=&v"(b.q[0]), "=&v"(b.q[1]), "=&v"(b.q[2]), "=&v"(b.q[3]), "+v"(pin) : "v"(addr));
        };
        auto arrived = [&](LqBuf& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3])); };
        // col += (x or K - x) * yt, x = the fetched coefficient
        auto mads = [&](uint64_t (&col)[2 * NL - 1], const LqBuf& b, bool neg, const auto& yt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t xl[4] = {b.q[q].x, b.q[q].y, b.q[q].z, b.q[q].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = 4 * q + u;
                    if (i >= NL) continue;
                    const uint32_t xi = neg ? KN.l[i] - xl[u] : xl[u];
#pragma unroll
                    for (int jj = 0; jj < NL; ++jj) col[i + jj] += (uint64_t)xi * yt.l[jj];
                }
            }
        };
        auto reduce_cols = [&](uint64_t (&col)[2 * NL - 1]) {
            Fqn r; uint64_t carry = 0;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                uint64_t sacc = col[k] + carry;
                const uint32_t m = ((uint32_t)sacc * INV28) & MASK;
#pragma unroll
                for (int jj = 1; jj < NL; ++jj) col[k + jj] += (uint64_t)m * P28.l[jj];
                sacc += (uint64_t)m * P28.l[0];
                carry = sacc >> W;
            }
#pragma unroll
            for (int k = NL; k < 2 * NL - 1; ++k) { const uint64_t sacc = col[k] + carry; r.l[k - NL] = (uint32_t)sacc & MASK; carry = sacc >> W; }
            r.l[NL - 1] = (uint32_t)carry;
            return r;
        };
        // (s0, s1, s2) against (l0, l1, l2):  im = sum x.c0 y.c1 + x.c1 y.c0,  re = sum x.c0 y.c0 + (K - x.c1) y.c1
        auto two_dots = [&](int s0, int s1, int s2, Fqn& re, Fqn& im) {
            const int sl[3] = {s0, s1, s2};
            LqBuf b0, b1;
            {
                uint64_t col[2 * NL - 1];
#pragma unroll
                for (int c = 0; c < 2 * NL - 1; ++c) col[c] = 0;
                fetch(b0, s0, 0, col[NL - 1]); arrived(b0);
#pragma unroll
                for (int t = 0; t < 3; ++t) {                                    // x.c0 y.c1 + x.c1 y.c0
                    fetch(b1, sl[t], 1, col[NL - 1]); mads(col, b0, false, y[2 * t + 1]); TIE(col); arrived(b1);
                    fetch(b0, sl[(t + 1) % 3], 0, col[NL - 1]); mads(col, b1, false, y[2 * t]); TIE(col); arrived(b0);      // (t = 2: x.c0 of s0 again, for the real part)
                }
                im = reduce_cols(col); TIE14(im.l);
            }
            {
                uint64_t col[2 * NL - 1];
#pragma unroll
                for (int c = 0; c < 2 * NL - 1; ++c) col[c] = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {                                    // x.c0 y.c0 + (K - x.c1) y.c1
                    fetch(b1, sl[t], 1, col[NL - 1]); mads(col, b0, false, y[2 * t]); TIE(col); arrived(b1);
                    if (t < 2) fetch(b0, sl[t + 1], 0, col[NL - 1]);
                    // u^2 = -FQ2_BETA: the factor goes to the line operand, as a transient normalised multiple (one live value, not three)
                    if constexpr (FQ2_BETA == 1) mads(col, b1, true, y[2 * t + 1]);
                    else { const auto yb = fq_norm(fq_mul_beta(y[2 * t + 1])); mads(col, b1, true, yb); }
                    TIE(col);
                    if (t < 2) arrived(b0);
                }
                re = reduce_cols(col); TIE14(re.l);
            }
        };
        two_dots((int)j + 3, s1b, (int)j, o1r, o1i);                              // k = j + 3: plain line
        {   // xi l2 always, xi l1 on the lanes whose operand wraps around w^6 = xi; normalised
            auto raw = [&](int f) { Fq<FQ_LN, 1> c; for (int q = 0; q < NL; ++q) c.l[q] = y[f].l[q]; return c; };       // still the canonical values loaded above
#if defined(RIPP_BLS12_377)
            // xi = u: (c0 + c1 u) u = -5 c1 + c0 u
            const auto d2 = fq_norm(fq_neg(fq_mul_beta(raw(5)))); const auto s2 = raw(4);
            const auto d1 = fq_norm(fq_neg(fq_mul_beta(raw(3)))); const auto s1 = raw(2);
            static_assert(decltype(d2)::VMAXB <= FYV, "");
#else
            // xi = 1 + u: (c0 - c1, c0 + c1)
            const auto d2 = fq_norm(fq_sub(raw(4), raw(5))); const auto s2 = fq_norm(fq_add(raw(4), raw(5)));
            const auto d1 = fq_norm(fq_sub(raw(2), raw(3))); const auto s1 = fq_norm(fq_add(raw(2), raw(3)));
#endif
            static_assert(sizeof(d2) == sizeof(FY) && sizeof(s2) == sizeof(FY), "");
#pragma unroll
            for (int q = 0; q < NL; ++q) { y[4].l[q] = d2.l[q]; y[5].l[q] = s2.l[q]; y[2].l[q] = xi1 ? d1.l[q] : y[2].l[q]; y[3].l[q] = xi1 ? s1.l[q] : y[3].l[q]; }
        }
        two_dots((int)j, a1, (int)j + 3, o0r, o0i);                              // k = j
        __syncthreads();                                                         // every lane of the group has read the old coefficients
        if (valid) { st_fq(j, 0, o0r); st_fq(j, 1, o0i); st_fq(j + 3, 0, o1r); st_fq(j + 3, 1, o1i); }
        __syncthreads();
    }
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
                const Fqn c = fq_canon(ld_fq(k, part));
                uint32_t w[12]; fq_pack(c, w);
#pragma unroll
                for (int q = 0; q < 3; ++q) prow[(uint32_t)(tower * 6 + part * 3 + q) * T + tt] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            }
        }
    }
#endif
}
#undef TIE
#undef TIE14
#undef LQ_CHUNK

}  // namespace ripp
