// Per-element scalar multiplication  out[i] = k[i] * base[i]  on G1 on the carry-free field form (fq28.hpp): the throughput twin of
// scale.hpp's k_scale_g1_glv (`a.mul(r)` of sipp/src/lib.rs:61-65, :189-194; a_r = a_i r^i of groth16_aggregation.rs:119-123).
// Same algorithm -- GLV split k = k1 + k2 lambda, the multiples 1..8 of the base tabulated per element, 33 signed base-16 digit positions of
// both halves walked jointly (132 doublings + <= 66 additions, uniform control flow) -- with the group law of fq_curve.hpp: every Fp product
// is 196 + 196 multiply-adds without carry instructions, additions are limb-wise.  k_scale_g1_glv runs at the VALU issue roof (2^20 elements x
// ~2 000 products / 50 G products/s = 42 ms), so the gain is the instruction count: ~30 %.
// The table holds JACOBIAN multiples (no inversion), 3 x 12 packed words per entry; an element's 8 multiples are CONTIGUOUS (1 152 B): every lane picks
// its own digit row, so in the lane-interleaved layout of scale.hpp a 128-byte line served one or two lanes (49.5 GB of HBM traffic per 2^20-element
// launch, 174 x the algorithmic bytes); here a lookup reads one 144-byte run.
// Exceptional additions (acc = +-T: H = 0) cannot occur for the digit patterns of a proper GLV split, but they are DETECTED and such a lane
// is flagged and recomputed (k_scale_g1_fix) by plain double-and-add with the complete formulas of curve.hpp.  Both curves (the group law of G1 does not involve the tower).
#pragma once
#include "fq_curve.hpp"
#include "scale.hpp"

namespace ripp {

#if defined(__HIP_DEVICE_COMPILE__)
#define SBS() __builtin_amdgcn_sched_barrier(0)
// a table entry as the main loop reads it: reduced coordinates (the table stores 12 packed words per coordinate), Y possibly negated (lazily)
struct JacT { Fqn x; Fq<FQ_LN, 4> y; Fqn z; };
// add-2007-bl (both operands Jacobian, neither the identity) in a low-liveness order.  Returns true when H = 0 (p = +-q): result not valid (read off (2H)^2).
// Bounds out: (11, 7, 2) p; nothing is reduced (fq_curve.hpp JacQ).
__device__ __forceinline__ bool jadd_q(JacQ& p, const JacT& q) {
    const Fqn Z1Z1 = fq_sqr(p.z); SBS();
    const Fqn U2 = fq_mul(q.x, Z1Z1); SBS();
    const Fqn S2 = fq_mul(fq_mul(q.y, p.z), Z1Z1); SBS();
    const Fqn Z2Z2 = fq_sqr(q.z); SBS();
    const Fqn U1 = fq_mul(p.x, Z2Z2); SBS();
    const Fqn S1 = fq_mul(fq_mul(p.y, q.z), Z2Z2); SBS();
    const auto Zs = fq_norm(fq_sub(fq_sub(fq_sqr(fq_norm(fq_add(p.z, q.z))), Z1Z1), Z2Z2)); SBS();      // 2 Z1 Z2
    const auto H = fq_norm(fq_sub(U2, U1));                                                               // < 5p
    p.z = fq_slot<JZ>(fq_mul(Zs, H)); SBS();
    const Fqn I = fq_sqr(fq_dbl(H)); SBS();
    const bool special = fq_is_zero(I);
    const Fqn J = fq_mul(H, I); SBS();
    const auto rr = fq_norm(fq_dbl(fq_sub(S2, S1)));
    const Fqn V = fq_mul(U1, I); SBS();
    const auto X3 = fq_norm(fq_sub(fq_sub(fq_sub(fq_sqr(rr), J), V), V)); SBS();
    p.y = fq_slot<JY>(fq_mul_sub(rr, fq_norm(fq_sub(V, X3)), fq_dbl(S1), J));                            // r (V - X3) - 2 S1 J, one reduction
    p.x = fq_slot<JX>(X3);
    return special;
}
#undef SBS
// one Jacobian table entry: 3 coordinates x 12 packed words = 9 chunks (the layout of scale.hpp's table); the coordinates are reduced here (< 2p < 2^384)
__device__ __forceinline__ void st_tab_q(uint4* tab, int e, uint32_t n, uint32_t i, const JacQ& t) {
    uint32_t w[36];
    { uint32_t x[12]; fq_pack(fq_reduce(t.x), x); for (int k = 0; k < 12; ++k) w[k] = x[k]; }
    { uint32_t x[12]; fq_pack(fq_reduce(t.y), x); for (int k = 0; k < 12; ++k) w[12 + k] = x[k]; }
    { uint32_t x[12]; fq_pack(fq_reduce(t.z), x); for (int k = 0; k < 12; ++k) w[24 + k] = x[k]; }
#pragma unroll
    for (int q = 0; q < G1J_CHUNKS; ++q) tab[((size_t)i * SCALE_TAB + e) * G1J_CHUNKS + q] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}
struct JacR { Fqn x, y, z; };
__device__ __forceinline__ JacR ld_tab_q(const uint4* tab, int e, uint32_t n, uint32_t i) {
    uint32_t w[36];
#pragma unroll
    for (int q = 0; q < G1J_CHUNKS; ++q) { const uint4 v = tab[((size_t)i * SCALE_TAB + e) * G1J_CHUNKS + q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    uint32_t a[12], b[12], c[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { a[k] = w[k]; b[k] = w[12 + k]; c[k] = w[24 + k]; }
    JacR r; r.x = fq_unpack(a); r.y = fq_unpack(b); r.z = fq_unpack(c);
    return r;
}
// the complete computation of one lane (exceptional additions): plain MSB-first double-and-add with curve.hpp's formulas
__device__ __noinline__ inline G1J scale_g1_plain(const G1A& p, const Fr& k) {
    G1J acc = jac_inf<Fp>();
#pragma unroll 1
    for (int bit = 254; bit >= 0; --bit) { acc = dbl(acc); if ((k.l[bit >> 5] >> (bit & 31)) & 1u) acc = add_mixed(acc, p); }
    return acc;
}
#endif

// same arguments, table size and output as k_scale_g1_glv
__global__ void __launch_bounds__(256, 2) k_scale_g1_glv_q(const G1A* __restrict__ base, uint32_t base_stride, const Fr* __restrict__ k_mont, uint32_t n,
                                                           uint4* __restrict__ tab, G1J* __restrict__ out, uint8_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const G1A p = base[(size_t)i * base_stride];
    if (is_inf(p)) { out[i] = jac_inf<Fp>(); flag[i] = 0; return; }
    uint32_t d1[5], d2[5];
    Fr kc = from_mont(k_mont[i]);
    {
        Fr k = kc;
        const uint32_t lam[8] = RIPP_GLV_LAMBDA;
        const uint32_t lam_mu[5] = RIPP_GLV_LAMBDA_MU;
        uint32_t rem[5];
        msm_divmod<4, 5>(k.l, lam, lam_mu, rem);                      // k = q * lambda + rem
        scale_bias(rem, d1); scale_bias(k.l, d2);
    }
    bool bad = false;
    {   // multiples 1..8 of the base, Jacobian: 1 doubling + 6 mixed additions
        const AffQ pq = affq_from(p);
        JacQ t; jq_set(t, pq.x, pq.y, fq_one());
        st_tab_q(tab, 0, n, i, t);
        jdbl_q(t);
        st_tab_q(tab, 1, n, i, t);
#pragma unroll 1
        for (int e = 2; e < SCALE_TAB; ++e) { bad |= jmadd_q(t, pq.x, pq.y); st_tab_q(tab, e, n, i, t); }
    }
    const Fqn beta = fq_from_fp(fp_const(RIPP_GLV_BETA));
    JacQ acc; jq_set_identity(acc);
    bool inf = true;
#pragma unroll 1
    for (int j = 32; j >= 0; --j) {
        if (!inf) { jdbl_q(acc); jdbl_q(acc); jdbl_q(acc); jdbl_q(acc); }
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int d = scale_digit(h ? d2 : d1, j);
            if (d == 0) continue;
            const JacR e = ld_tab_q(tab, (d < 0 ? -d : d) - 1, n, i);
            JacT t; t.x = e.x; t.y = fq_widen<FQ_LN, 4>(e.y); t.z = e.z;
            if (d < 0) t.y = fq_slot<Fq<FQ_LN, 4>>(fq_neg(e.y));
            if (h) t.x = fq_mul(e.x, beta);                                        // phi in Jacobian coordinates: (beta X, Y, Z)
            if (inf) { jq_set(acc, t.x, t.y, t.z); inf = false; } else bad |= jadd_q(acc, t);
        }
    }
    flag[i] = bad;
    if (bad) return;
    if (inf) out[i] = jac_inf<Fp>();
    else out[i] = jacq_to_g1j(acc);
#endif
}
// the flagged lanes of k_scale_g1_glv_q, by plain double-and-add with the complete formulas
__global__ void __launch_bounds__(64) k_scale_g1_fix(const G1A* __restrict__ base, uint32_t base_stride, const Fr* __restrict__ k_mont, uint32_t n, G1J* __restrict__ out, const uint8_t* __restrict__ flag) {
#if defined(__HIP_DEVICE_COMPILE__)
    for_flagged(flag, n, [&](uint32_t i) { out[i] = scale_g1_plain(base[(size_t)i * base_stride], from_mont(k_mont[i])); });
#endif
}

}  // namespace ripp
