// Stage 2a of the pairing product, spill-free form: L_row = prod_i line_{row,i}  (the accumulation side of ark-ec's multi_miller_loop,
// inner_products/src/lib.rs:104-115 / sipp/src/lib.rs:196-214).
//
// Why this kernel exists.  The one-lane-per-accumulator form (k_line_products1, kept below as the A/B reference) holds an Fp12
// accumulator (144 dwords), the sparse line (72) and the Karatsuba temporaries of mul_by_014 (>= 144) in one lane: ~400 dwords of live
// state against 256 registers at the 2 waves/SIMD the integer pipe needs -- 370 dwords spilled, ~31 GB of scratch traffic per launch
// (1 081 x the algorithmic bytes).  Here
//   * the accumulator lives in LDS (w-basis coefficients f_0..f_5 in Fp2, plus xi*f_3..xi*f_5: 864 B), owned by a GROUP OF 3 LANES;
//   * lane j of the group produces output coefficients j and j + 3 of  f * (l0 + l1 w^2 + l2 w^3):
//         out_k = f_k l0 + [xi] f_(k-2) l1 + [xi] f_(k-3) l2          (indices mod 6, xi when the index wraps; w^6 = xi = 1 + u)
//     each as ONE lazily reduced sum of products: the 6 (real part) resp. 6 (imaginary part) limb-product blocks of the three Fp2
//     products go into one 96-bit column accumulator and share ONE Montgomery reduction (6 p^2 < p R since p/R ~ 0.1);
//   * no Karatsuba sums, no additions, no temporaries: a lane's live state is its 3 + 3 Fp2 operands, two outputs and the column state.
// Limb products per line: 6 coefficients x (12 + 2) blocks of 144 = 12 096 v_mad_u64_u32 (Karatsuba-13: 11 232 + ~60 Fp additions), all in
// registers.  63 of 64 lanes work (21 groups per wave).
#pragma once
#include <hip/hip_runtime.h>
#include "bls12_381/pairing.hpp"

namespace ripp {

#if defined(__HIP_DEVICE_COMPILE__)
// Several limb products per asm statement: hipcc's hazard recogniser puts one s_nop behind EVERY inline-asm statement (it cannot see that the
// statement ends in a v_addc), i.e. one per limb product in the single-product form -- 3 850 issue slots per line and lane in this kernel.
#define RIPP_M1(a, b) "v_mad_u64_u32 %0, vcc, " a ", " b ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
__device__ __forceinline__ void madc96_x2(uint64_t& acc, uint32_t& c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    asm(RIPP_M1("%2", "%3") RIPP_M1("%4", "%5") : "+v"(acc), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
__device__ __forceinline__ void madc96_x3(uint64_t& acc, uint32_t& c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2) {
    asm(RIPP_M1("%2", "%3") RIPP_M1("%4", "%5") RIPP_M1("%6", "%7") : "+v"(acc), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2) : "vcc");
}
__device__ __forceinline__ void madc96_x6(uint64_t& acc, uint32_t& c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2,
                                          uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5) {
    asm(RIPP_M1("%2", "%3") RIPP_M1("%4", "%5") RIPP_M1("%6", "%7") RIPP_M1("%8", "%9") RIPP_M1("%10", "%11") RIPP_M1("%12", "%13")
        : "+v"(acc), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5) : "vcc");
}
#undef RIPP_M1
// the NT limb products a[t][i] * b[t][j] of one (i, j), in as few asm statements as possible
template <int NT>
__device__ __forceinline__ void madc96_terms(uint64_t& acc, uint32_t& c2, const Fp (&a)[NT], const Fp (&b)[NT], int i, int j) {
    if constexpr (NT == 6) { madc96_x3(acc, c2, a[0].l[i], b[0].l[j], a[1].l[i], b[1].l[j], a[2].l[i], b[2].l[j]); madc96_x3(acc, c2, a[3].l[i], b[3].l[j], a[4].l[i], b[4].l[j], a[5].l[i], b[5].l[j]); }
    else if constexpr (NT == 3) madc96_x3(acc, c2, a[0].l[i], b[0].l[j], a[1].l[i], b[1].l[j], a[2].l[i], b[2].l[j]);
    else if constexpr (NT == 2) madc96_x2(acc, c2, a[0].l[i], b[0].l[j], a[1].l[i], b[1].l[j]);
    else {
#pragma unroll
        for (int t = 0; t < NT; ++t) madc96(acc, c2, a[t].l[i], b[t].l[j]);
    }
}
// sum_{t < NT} a[t] * b[t] * R^-1 mod p with one Montgomery reduction.  Inputs < p (b[t] <= p allowed); NT p^2 < p R must hold (NT <= 9).
template <int NT>
__device__ __forceinline__ Fp fp_dot(const Fp (&a)[NT], const Fp (&b)[NT]) {
    constexpr int N = 12;
    uint32_t m[N];
    Fp r;
    uint64_t acc = 0; uint32_t c2 = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) madc96_terms<NT>(acc, c2, a, b, i, k - i);
#pragma unroll
        for (int i = 0; i < k; ++i) madc96_s(acc, c2, m[i], FpParams::mod(k - i));
        m[k] = (uint32_t)acc * FpParams::INV;
        madc96_s(acc, c2, m[k], FpParams::mod(0));
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) madc96_terms<NT>(acc, c2, a, b, i, k - i);
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) madc96_s(acc, c2, m[i], FpParams::mod(k - i));
        r.l[k - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
    r.l[N - 1] = (uint32_t)acc;
    reduce_once(r);
    return r;
}
#else
template <int NT> __device__ Fp fp_dot(const Fp (&a)[NT], const Fp (&b)[NT]);   // device only; the host pass merely parses the kernel below
#endif

constexpr int LP_GROUP = 3;                       // lanes per accumulator
constexpr int LP_GROUPS_PER_WAVE = 21;            // 63 of 64 lanes
constexpr int LP_SLOTS = 9;                       // f_0..f_5, xi f_3, xi f_4, xi f_5
constexpr int LP_ACC_BYTES = LP_SLOTS * 96;       // 864 B of LDS per accumulator
constexpr int LP_ACC_STRIDE = LP_SLOTS * 6 + 1;   // laid out 55 chunks (880 B) apart: 864 B repeats its bank pattern every 4 groups, 880 B every 8 (fq_line_products.hpp has the measurement)

// grid = (ceil(T / 21), rows), block = 64 (one wave).  Group t of row r multiplies lines r[t], r[t + T], ... (< M) and writes ONE dense
// partial to partials[r][36][T] (the layout k_fp12_tree consumes).  lines: [rows][18][stride] 16-byte chunks (kernels.hpp).
__global__ void __launch_bounds__(64, 2) k_line_products(const uint4* __restrict__ lines, size_t stride, uint32_t M,
                                                         uint4* __restrict__ partials, uint32_t T) {
    __shared__ uint4 lds[LP_GROUPS_PER_WAVE * LP_ACC_STRIDE];
    const uint32_t lane = threadIdx.x;
    const uint32_t g = lane / LP_GROUP, j = lane - g * LP_GROUP;               // group in wave, lane in group (lane 63: g = 21, idle)
    // t = accumulator index within the row; cheap to recompute from the lane id, so it is NOT kept live across the 8 000-instruction
    // loop body (re-derived through mbcnt behind an opaque operand: no CSE with the copy above)
    auto group_index = [&]() { uint32_t z = 0; asm volatile("" : "+s"(z));      // opaque zero: the lane id below is re-derived, not CSE'd and kept
        const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));   // one wave per block: lane id == threadIdx.x
        return blockIdx.x * LP_GROUPS_PER_WAVE + l / LP_GROUP; };
    const uint32_t t = blockIdx.x * LP_GROUPS_PER_WAVE + g;
    const bool active = g < (uint32_t)LP_GROUPS_PER_WAVE && t < T;
    const size_t row = blockIdx.y;
    uint4* acc = lds + (active ? g : 0) * LP_ACC_STRIDE;                         // 6 chunks (96 B) per Fp2 slot
    auto ld_slot = [&](int s) { Fp2 v; uint4* d = reinterpret_cast<uint4*>(&v);
#pragma unroll
        for (int q = 0; q < 6; ++q) d[q] = acc[s * 6 + q]; return v; };
    auto st_slot = [&](int s, const Fp2& v) { const uint4* d = reinterpret_cast<const uint4*>(&v);
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[s * 6 + q] = d[q]; };
    // accumulator <- 1
    if (active) {
        st_slot(j, j == 0 ? Fp2::one() : Fp2::zero()); st_slot(j + 3, Fp2::zero()); st_slot(j + 6, Fp2::zero());
    }
    __syncthreads();
    // operand slots of this lane's two outputs k = j and k = j + 3:  (f_k, [xi] f_(k-2), [xi] f_(k-3));  xi f_i sits in slot i + 3
    //   k = 0: f0, xi f4, xi f3     k = 1: f1, xi f5, xi f4     k = 2: f2, f0, xi f5     k = 3: f3, f1, f0     k = 4: f4, f2, f1     k = 5: f5, f3, f2
#if defined(RIPP_BLS12_377)
    // D-type twist: line = l0 + l1 w + l2 w^3  =>  out_k = f_k l0 + [xi] f_(k-1) l1 + [xi] f_(k-3) l2
    //   k = 0: f0, xi f5, xi f3     k = 1: f1, f0, xi f4     k = 2: f2, f1, xi f5     k = 3: f3, f2, f0     k = 4: f4, f3, f1     k = 5: f5, f4, f2
    const int s1a = j == 0 ? 8 : (int)j - 1, s2a = j == 0 ? 6 : j == 1 ? 7 : 8;          // output j
    const int s0b = j + 3, s1b = j + 2, s2b = j;                                          // output j + 3
#else
    const int s1a = j == 0 ? 7 : j == 1 ? 8 : 0, s2a = j == 0 ? 6 : j == 1 ? 7 : 8;      // output j
    const int s0b = j + 3, s1b = j + 1, s2b = j;                                          // output j + 3
#endif
    const uint32_t st = (uint32_t)stride;                                   // rows are < 2^32 chunks apart (<= 18 * 2^19 * ... per row block)
    const uint4* __restrict__ lrow = lines + row * 18 * stride;
    const uint32_t iters = (M + T - 1) / T;                                 // uniform trip count (the barriers below are reached by every lane);
#pragma unroll 1
    for (uint32_t it = 0; it < iters; ++it) {                              // groups past the end of the row recompute line 0 and discard it
        const uint32_t i = group_index() + it * T;                        // recomputed, not kept: one register instead of a spilled pair
        const bool valid = active && i < M;
        Fp2 y[3];
        {
            uint4* d = reinterpret_cast<uint4*>(&y[0]);
            const uint32_t ii = valid ? i : 0;                             // 32-bit lane offsets against a wave-uniform row base (SGPR pair)
#pragma unroll
            for (int q = 0; q < 18; ++q)                                    // 32-bit BYTE offsets (a row block is < 4 GB): SGPR base + VGPR offset addressing
                d[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lrow) + (((uint32_t)q * st + ii) << 4));
        }
        Fp2 out0, out1;                                                    // (no out[h]: a dynamically indexed array would live in scratch)
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            Fp xs[6], ys[6];
            {
                const Fp2 x0 = ld_slot(h ? s0b : (int)j), x1 = ld_slot(h ? s1b : s1a), x2 = ld_slot(h ? s2b : s2a);
                xs[0] = x0.c0; xs[1] = x0.c1; xs[2] = x1.c0; xs[3] = x1.c1; xs[4] = x2.c0; xs[5] = x2.c1;
            }
            // imaginary part: sum x.c0 y.c1 + x.c1 y.c0 (pure operand permutation: no copies once unrolled)
            ys[0] = y[0].c1; ys[1] = y[0].c0; ys[2] = y[1].c1; ys[3] = y[1].c0; ys[4] = y[2].c1; ys[5] = y[2].c0;
            const Fp im = fp_dot<6>(xs, ys);
            // real part: sum x.c0 y.c0 + (-x.c1) y.c1 -- the negation goes to x, which is reloaded for every output anyway, so that
            // y stays as loaded for the second output of the line (p - x.c1 <= p is a valid lazy operand)
#if defined(RIPP_BLS12_377)
            xs[1] = neg(mul5(xs[1])); xs[3] = neg(mul5(xs[3])); xs[5] = neg(mul5(xs[5]));          // u^2 = -5
#else
            xs[1] = neg(xs[1]); xs[3] = neg(xs[3]); xs[5] = neg(xs[5]);
#endif
            ys[0] = y[0].c0; ys[1] = y[0].c1; ys[2] = y[1].c0; ys[3] = y[1].c1; ys[4] = y[2].c0; ys[5] = y[2].c1;
            const Fp re = fp_dot<6>(xs, ys);
            if (h == 0) { out0.c0 = re; out0.c1 = im; } else { out1.c0 = re; out1.c1 = im; }
        }
        __syncthreads();                                                   // every lane of the group has read the old coefficients
        if (valid) { st_slot(j, out0); st_slot(j + 3, out1); st_slot(j + 6, mul_xi(out1)); }
        __syncthreads();
    }
    // write the group's accumulator: chunk c of the Fp12 in TOWER order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2) = w-index (0, 2, 4, 1, 3, 5)
    if (active) {
        uint4* __restrict__ prow = partials + row * 36 * T;
        const uint32_t tt = group_index();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = j + 3 * u;                                        // w-index held by this lane
            const int tower = (k & 1) ? 3 + (k >> 1) : (k >> 1);
#pragma unroll
            for (int q = 0; q < 6; ++q) prow[(uint32_t)(tower * 6 + q) * T + tt] = acc[k * 6 + q];
        }
    }
}

}  // namespace ripp
