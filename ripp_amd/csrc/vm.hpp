// Lane-parallel field VM: G lanes of a wave cooperate on ONE element; every Fp value of the element lives in a
// per-element LDS workspace; a program is a list of homogeneous LAYERS (tables generated and verified offline by
// tools/vmgen.py) in which each lane performs one
//     MUL  ws[dst] = (+-ws[a0] +-ws[a1]) * (+-ws[a2] +-ws[a3])
//     LIN  ws[dst] = ((+-ws[a0] +-ws[a1] +-ws[a2] +-ws[a3]) << sh) [/ 2]
// Why: the scalar kernels (one lane per element) keep ~400 dwords of live state per lane, so they run one wave per
// SIMD and, in the small late rounds of a proof, one lone lane pays ~2.4 us per DEPENDENT Fp product.  Here the
// state sits in LDS (160 KB/CU), each lane needs ~60 VGPRs, the multiplier is inlined once (no calls), and the
// 25-54 independent products of a step run side by side: a Miller doubling step is 2 product layers instead of 25
// dependent products.  Used for the latency-bound part of the pipeline; the scalar kernels remain the throughput
// path (and the fallback if an exceptional group-law case is flagged).
#pragma once
#include <hip/hip_runtime.h>
#include "bls12_381/pairing.hpp"
#include "kernels.hpp"

namespace ripp {

struct VmOp { unsigned char dst, a0, a1, a2, a3, flags; };
}  // namespace ripp
#include "vm_programs.inc"
namespace ripp {

constexpr int VM_G = 16;                 // lanes per element
constexpr int VM_EPW = 64 / VM_G;        // elements per wave

__device__ __forceinline__ Fp vm_cneg(const Fp& x, bool f) {
    const Fp n = neg(x);
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; ++i) r.l[i] = f ? n.l[i] : x.l[i];
    return r;
}

// Run one program on this lane's element.  `ws` = LDS workspace of the element, `lg` = lane index within the group.
// A lone wave issues one VALU instruction per ~8 cycles, so every instruction of a layer is latency: the operand preparation
// (conditional negations, the second term of each operand, shifts, halving) is skipped whenever NO lane of the wave needs it -- the
// test is wave-uniform (`__any`), so there is no divergence.  Slot 0 holds zero, hence "second term absent" == slot index 0.
__device__ __forceinline__ Fp vm_operand(const Fp* ws, unsigned s0, unsigned s1, bool n0, bool n1) {
    Fp x = ws[s0];
    if (__any(n0)) x = vm_cneg(x, n0);
    if (__any(s1 != 0u)) { Fp y = ws[s1]; if (__any(n1)) y = vm_cneg(y, n1); x = add(x, y); }
    return x;
}
__device__ __forceinline__ void vm_run(Fp* ws, const unsigned char* __restrict__ kind, const VmOp* __restrict__ ops, int nlayers, int lg) {
#pragma unroll 1
    for (int l = 0; l < nlayers; ++l) {
        const VmOp op = ops[l * VM_G + lg];
        const unsigned f = op.flags;
        Fp r;
        if (kind[l] == 0) {                                    // MUL layer (uniform over the wave)
            const Fp A = vm_operand(ws, op.a0, op.a1, f & 1u, f & 2u);
            const Fp B = vm_operand(ws, op.a2, op.a3, f & 4u, f & 8u);
            r = mul(A, B);
        } else {                                                // LIN layer
            r = vm_operand(ws, op.a0, op.a1, f & 1u, f & 2u);
            if (__any((op.a2 | op.a3) != 0u)) r = add(r, vm_operand(ws, op.a2, op.a3, f & 4u, f & 8u));
            const unsigned sh = (f >> 5) & 3u;
            if (__any(sh != 0u)) {
#pragma unroll 1
                for (unsigned k = 0; k < 3; ++k) { if (!__any(k < sh)) break; const Fp d = dbl(r); const bool t = k < sh;
#pragma unroll
                    for (int i = 0; i < 12; ++i) r.l[i] = t ? d.l[i] : r.l[i]; }
            }
            if (__any((f & 16u) != 0u)) { const Fp h = half(r); const bool t = (f & 16u) != 0u;
#pragma unroll
                for (int i = 0; i < 12; ++i) r.l[i] = t ? h.l[i] : r.l[i]; }
        }
        ws[op.dst] = r;     // all lanes of the wave have issued their reads of this layer before any write (lockstep, in-order LDS)
    }
}

// ---- stage 1 of the pairing product in latency form: VM_G lanes per (P,Q) pair ---------------------------------
// grid.x covers ceil(M / (4 * VM_EPW)) blocks of 256 threads, grid.y = product; same line buffer layout as k_miller_lines.
constexpr int VM_LINES_SLOTS = (vmprog::line_double_g16_nslots > vmprog::line_add_g16_nslots) ? vmprog::line_double_g16_nslots : vmprog::line_add_g16_nslots;
__global__ void __launch_bounds__(256) k_vm_miller_lines(PairSets ps, uint32_t M, uint4* __restrict__ lines, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    Fp* const lds = reinterpret_cast<Fp*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;       // pair index handled by this group
    Fp* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_LINES_SLOTS;
    const bool active = i < M;
    const G1A* __restrict__ a = ps.a[blockIdx.y];
    const G2A* __restrict__ b = ps.b[blockIdx.y];
    namespace vp = vmprog;
    bool skip = false;
    if (active) {
        // lanes 0..7 of the group load the 8 input Fp's; everyone learns whether the pair contributes the unit line
        const G1A P = a[i]; const G2A Q = b[i];
        skip = is_inf(P) || is_inf(Q);
        if (lg == 0) { ws[0] = Fp::zero(); ws[vp::line_double_g16_in_X0] = Q.x.c0; ws[vp::line_double_g16_in_X1] = Q.x.c1; }
        if (lg == 1) { ws[vp::line_double_g16_in_Y0] = Q.y.c0; ws[vp::line_double_g16_in_Y1] = Q.y.c1; }
        if (lg == 2) { ws[vp::line_double_g16_in_Z0] = Fp::one(); ws[vp::line_double_g16_in_Z1] = Fp::zero(); }
        if (lg == 3) { ws[vp::line_double_g16_in_xP] = P.x; ws[vp::line_double_g16_in_yP] = P.y; }
    }
    const size_t row0 = (size_t)blockIdx.y * N_LINES;
    int s = 0;
#pragma unroll 1
    for (int bit = 62; bit >= 0; --bit) {
        vm_run(ws, vp::line_double_g16_kind, vp::line_double_g16_ops, vp::line_double_g16_nlayers, lg);
        if (active && lg < 6) {
            constexpr int OUT[6] = {vp::line_double_g16_out_L00, vp::line_double_g16_out_L01, vp::line_double_g16_out_L10, vp::line_double_g16_out_L11, vp::line_double_g16_out_L20, vp::line_double_g16_out_L21};
            int src = OUT[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) src = (lg == k) ? OUT[k] : src;
            Fp v = ws[src];
            if (skip) v = (lg == 0) ? Fp::one() : Fp::zero();
            const uint4* pv = reinterpret_cast<const uint4*>(&v);
#pragma unroll
            for (int c = 0; c < 3; ++c) lines[((row0 + s) * LINE_CHUNKS + 3 * lg + c) * stride + i] = pv[c];
        }
        ++s;
        if ((BLS_X_ABS >> bit) & 1ull) {
            if (active && lg == 4) { const G2A Q = b[i]; ws[vp::line_add_g16_in_qx0] = Q.x.c0; ws[vp::line_add_g16_in_qx1] = Q.x.c1; ws[vp::line_add_g16_in_qy0] = Q.y.c0; ws[vp::line_add_g16_in_qy1] = Q.y.c1; }
            vm_run(ws, vp::line_add_g16_kind, vp::line_add_g16_ops, vp::line_add_g16_nlayers, lg);
            if (active && lg < 6) {
                constexpr int OUT[6] = {vp::line_add_g16_out_L00, vp::line_add_g16_out_L01, vp::line_add_g16_out_L10, vp::line_add_g16_out_L11, vp::line_add_g16_out_L20, vp::line_add_g16_out_L21};
                int src = OUT[0];
#pragma unroll
                for (int k = 1; k < 6; ++k) src = (lg == k) ? OUT[k] : src;
                Fp v = ws[src];
                if (skip) v = (lg == 0) ? Fp::one() : Fp::zero();
                const uint4* pv = reinterpret_cast<const uint4*>(&v);
#pragma unroll
                for (int c = 0; c < 3; ++c) lines[((row0 + s) * LINE_CHUNKS + 3 * lg + c) * stride + i] = pv[c];
            }
            ++s;
        }
    }
}

// ---- dense Fp12 product tree level, one group per product: out[j] = in[j] * in[j + Tout] -----------------------
constexpr int VM_F12_SLOTS = vmprog::fp12_mul_g16_nslots;
__global__ void __launch_bounds__(128) k_vm_fp12_tree(const uint4* __restrict__ in, uint32_t Tin, uint4* __restrict__ out, uint32_t Tout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    Fp* const lds = reinterpret_cast<Fp*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t j = (blockIdx.x * (blockDim.x >> 6) + wave) * VM_EPW + grp;
    const size_t row = blockIdx.y;
    Fp* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_F12_SLOTS;
    const bool active = j < Tout, pair = active && (j + Tout < Tin);
    namespace vp = vmprog;
    if (active && lg < 12) {      // Fp k of an Fp12 = chunks 3k..3k+2; f -> slots 2.., g -> slots 14..
        Fp f; uint4* pf = reinterpret_cast<uint4*>(&f);
#pragma unroll
        for (int c = 0; c < 3; ++c) pf[c] = in[(row * FP12_CHUNKS + 3 * lg + c) * Tin + j];
        ws[vp::fp12_mul_g16_in[0] + lg] = f;
        if (pair) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pf[c] = in[(row * FP12_CHUNKS + 3 * lg + c) * Tin + j + Tout];
            ws[vp::fp12_mul_g16_in[12] + lg] = f;
        }
        if (lg == 0) ws[0] = Fp::zero();
    }
    // groups without a partner (odd tail) just copy; the program still runs wave-uniformly on harmless data
    vm_run(ws, vp::fp12_mul_g16_kind, vp::fp12_mul_g16_ops, vp::fp12_mul_g16_nlayers, lg);
    if (active && lg < 12) {
        Fp f;
        if (pair) f = ws[vp::fp12_mul_g16_out[0] + lg];
        else { uint4* pf = reinterpret_cast<uint4*>(&f);
#pragma unroll
            for (int c = 0; c < 3; ++c) pf[c] = in[(row * FP12_CHUNKS + 3 * lg + c) * Tin + j]; }
        const uint4* pf = reinterpret_cast<const uint4*>(&f);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(row * FP12_CHUNKS + 3 * lg + c) * Tout + j] = pf[c];
    }
}

// ---- scalar multiplication d * Q in latency form (homogeneous projective double-and-add on the VM) -----------------
// G2: blockIdx.y = j selects the GLS image [u^j]Q and its NAF digit string; result (Jacobian) -> parts[j][i].
// `flag` is set when an addition met lambda == 0 (T = +-Q): the caller then redoes the launch with the scalar kernel.
constexpr int VM_G2_SLOTS = (vmprog::g2_hdbl_g16_nslots > vmprog::g2_cadd_g16_nslots) ? vmprog::g2_hdbl_g16_nslots : vmprog::g2_cadd_g16_nslots;
__global__ void __launch_bounds__(256) k_vm_fold_g2_split(const G2A* __restrict__ hi, uint32_t half, GlsDigits dg, G2J* __restrict__ parts, uint32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    Fp* const lds = reinterpret_cast<Fp*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    const int j = blockIdx.y;
    Fp* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G2_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g2_cadd_g16_in_X0, SY = vp::g2_cadd_g16_in_Y0, SZ = vp::g2_cadd_g16_in_Z0, SQX = vp::g2_cadd_g16_in_qx0, SQY = vp::g2_cadd_g16_in_qy0, SQZ = vp::g2_cadd_g16_in_qz0 };
    static_assert(vp::g2_hdbl_g16_in_X0 == SX && vp::g2_hdbl_g16_in_Y0 == SY && vp::g2_hdbl_g16_in_Z0 == SZ, "accumulator slots shared by the two programs");
    G2A q = aff_inf<Fp2>(); bool qinf = true;
    if (active && lg == 0) { q = gls_image(hi[i], j); qinf = is_inf(q); ws[0] = Fp::zero(); }
    int pos = dg.len - 1;
    while (pos >= 0 && dg.d[j][pos] == 0) --pos;               // uniform: digits are shared by the whole launch
    const bool any = pos >= 0;
    if (any && active && lg == 0) {
        const Fp2 y0 = dg.d[j][pos] < 0 ? neg(q.y) : q.y;
        ws[SX] = q.x.c0; ws[SX + 1] = q.x.c1; ws[SY] = y0.c0; ws[SY + 1] = y0.c1; ws[SZ] = Fp::one(); ws[SZ + 1] = Fp::zero();
    }
    // additions use the COMPLETE projective law (g2_cadd: depth 2, no exceptional case), the addend (q.x : +-q.y : 1) is rewritten
    // before each one because the doubling program may use those slots as temporaries
    (void)flag;
#pragma unroll 1
    for (--pos; pos >= 0; --pos) {
        vm_run(ws, vp::g2_hdbl_g16_kind, vp::g2_hdbl_g16_ops, vp::g2_hdbl_g16_nlayers, lg);
        const int d = dg.d[j][pos];
        if (d != 0) {
            if (active && lg == 0) { const Fp2 y = d < 0 ? neg(q.y) : q.y; ws[SQX] = q.x.c0; ws[SQX + 1] = q.x.c1; ws[SQY] = y.c0; ws[SQY + 1] = y.c1; ws[SQZ] = Fp::one(); ws[SQZ + 1] = Fp::zero(); }
            vm_run(ws, vp::g2_cadd_g16_kind, vp::g2_cadd_g16_ops, vp::g2_cadd_g16_nlayers, lg);
        }
    }
    if (active && lg == 0) {
        // parts are left in HOMOGENEOUS projective form (X : Y : Z), the identity as (0 : 1 : 0): k_vm_combine_g2 adds them with the
        // complete addition program, so no lone-lane conversion sits on the chain
        G2J r; r.x = Fp2::zero(); r.y = Fp2::one(); r.z = Fp2::zero();
        if (any && !qinf) { r.x = {ws[SX], ws[SX + 1]}; r.y = {ws[SY], ws[SY + 1]}; r.z = {ws[SZ], ws[SZ + 1]}; }
        parts[(size_t)j * half + i] = r;
    }
}

// out[i] = parts[0][i] + parts[1][i] + parts[2][i] + parts[3][i] + lo[i]  (Jacobian out): four complete VM additions instead of the
// lone-lane Jacobian additions of k_fold_g2_combine (0.35-0.43 ms on the fold chain of every small round)
__global__ void __launch_bounds__(256) k_vm_combine_g2(const G2J* __restrict__ parts, const G2A* __restrict__ lo, uint32_t half, G2J* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    Fp* const lds = reinterpret_cast<Fp*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    Fp* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G2_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g2_cadd_g16_in_X0, SY = vp::g2_cadd_g16_in_Y0, SZ = vp::g2_cadd_g16_in_Z0, SQX = vp::g2_cadd_g16_in_qx0, SQY = vp::g2_cadd_g16_in_qy0, SQZ = vp::g2_cadd_g16_in_qz0 };
    auto put = [&](int s, const Fp2& v) { ws[s] = v.c0; ws[s + 1] = v.c1; };
    if (lg == 0) {
        ws[0] = Fp::zero();
        G2J p0; p0.x = Fp2::zero(); p0.y = Fp2::one(); p0.z = Fp2::zero();
        if (active) p0 = parts[i];
        put(SX, p0.x); put(SY, p0.y); put(SZ, p0.z);
    }
#pragma unroll 1
    for (int j = 1; j <= 4; ++j) {
        if (lg == 0) {
            G2J q; q.x = Fp2::zero(); q.y = Fp2::one(); q.z = Fp2::zero();
            if (active) {
                if (j < 4) q = parts[(size_t)j * half + i];
                else { const G2A l = lo[i]; if (!is_inf(l)) { q.x = l.x; q.y = l.y; q.z = Fp2::one(); } }
            }
            put(SQX, q.x); put(SQY, q.y); put(SQZ, q.z);
        }
        vm_run(ws, vp::g2_cadd_g16_kind, vp::g2_cadd_g16_ops, vp::g2_cadd_g16_nlayers, lg);
    }
    if (active && lg == 0) {
        const Fp2 X = {ws[SX], ws[SX + 1]}, Y = {ws[SY], ws[SY + 1]}, Z = {ws[SZ], ws[SZ + 1]};
        G2J r = jac_inf<Fp2>();
        if (!Z.is_zero()) { r.x = mul(X, Z); r.y = mul(Y, sqr(Z)); r.z = Z; }          // (X/Z, Y/Z) -> Jacobian (XZ, YZ^2, Z)
        out[i] = r;
    }
}

// G1: single NAF digit string (the 128-bit SIPP challenge); out[i] = s*hi[i] + lo[i] (Jacobian)
constexpr int VM_G1_SLOTS = (vmprog::g1_hdbl_g16_nslots > vmprog::g1_cadd_g16_nslots) ? vmprog::g1_hdbl_g16_nslots : vmprog::g1_cadd_g16_nslots;
__global__ void __launch_bounds__(256) k_vm_fold_g1(const G1A* __restrict__ hi, const G1A* __restrict__ lo, uint32_t half, NafDigits dg, G1J* __restrict__ out, uint32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    Fp* const lds = reinterpret_cast<Fp*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    Fp* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G1_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g1_cadd_g16_in_X0, SY = vp::g1_cadd_g16_in_Y0, SZ = vp::g1_cadd_g16_in_Z0, SQX = vp::g1_cadd_g16_in_qx0, SQY = vp::g1_cadd_g16_in_qy0, SQZ = vp::g1_cadd_g16_in_qz0 };
    static_assert(vp::g1_hdbl_g16_in_X0 == SX && vp::g1_hdbl_g16_in_Y0 == SY && vp::g1_hdbl_g16_in_Z0 == SZ, "accumulator slots shared by the two programs");
    G1A q = aff_inf<Fp>(); bool qinf = true;
    if (active && lg == 0) { q = hi[i]; qinf = is_inf(q); ws[0] = Fp::zero(); }
    int pos = dg.len - 1;
    while (pos >= 0 && dg.d[pos] == 0) --pos;
    const bool any = pos >= 0;
    if (any && active && lg == 0) { ws[SX] = q.x; ws[SY] = dg.d[pos] < 0 ? neg(q.y) : q.y; ws[SZ] = Fp::one(); }
    (void)flag;                                                 // complete addition (g1_cadd): nothing to report
#pragma unroll 1
    for (--pos; pos >= 0; --pos) {
        vm_run(ws, vp::g1_hdbl_g16_kind, vp::g1_hdbl_g16_ops, vp::g1_hdbl_g16_nlayers, lg);
        const int d = dg.d[pos];
        if (d != 0) {
            if (active && lg == 0) { ws[SQX] = q.x; ws[SQY] = d < 0 ? neg(q.y) : q.y; ws[SQZ] = Fp::one(); }
            vm_run(ws, vp::g1_cadd_g16_kind, vp::g1_cadd_g16_ops, vp::g1_cadd_g16_nlayers, lg);
        }
    }
    if (active && lg == 0) {
        G1J r = jac_inf<Fp>();
        if (any && !qinf) {
            const Fp X = ws[SX], Y = ws[SY], Z = ws[SZ];
            if (!Z.is_zero()) { r.x = fmul(X, Z); r.y = fmul(Y, fsqr(Z)); r.z = Z; }
        }
        out[i] = add_mixed(r, lo[i]);
    }
}

// G1 fold with a full-width scalar in latency form: the GLV halves of kernels.hpp::k_fold_g1_glv on the field VM
// (128 VM doublings, complete additions of +-P and +-phi(P)).
__global__ void __launch_bounds__(256) k_vm_fold_g1_glv(const G1A* __restrict__ hi, const G1A* __restrict__ lo, uint32_t half, GlvDigits dg, G1J* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    Fp* const lds = reinterpret_cast<Fp*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    Fp* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G1_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g1_cadd_g16_in_X0, SY = vp::g1_cadd_g16_in_Y0, SZ = vp::g1_cadd_g16_in_Z0, SQX = vp::g1_cadd_g16_in_qx0, SQY = vp::g1_cadd_g16_in_qy0, SQZ = vp::g1_cadd_g16_in_qz0 };
    G1A q = aff_inf<Fp>(); Fp bx = Fp::zero(); bool qinf = true;
    if (active && lg == 0) { q = hi[i]; qinf = is_inf(q); bx = fmul(q.x, fp_const(RIPP_GLV_BETA)); ws[0] = Fp::zero(); ws[SX] = Fp::zero(); ws[SY] = Fp::one(); ws[SZ] = Fp::zero(); }   // T = identity (0:1:0)
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        vm_run(ws, vp::g1_hdbl_g16_kind, vp::g1_hdbl_g16_ops, vp::g1_hdbl_g16_nlayers, lg);
        const int d1 = dg.d1[pos], d2 = dg.d2[pos];
        if (d1 != 0) {
            if (active && lg == 0) { ws[SQX] = q.x; ws[SQY] = d1 < 0 ? neg(q.y) : q.y; ws[SQZ] = Fp::one(); }
            vm_run(ws, vp::g1_cadd_g16_kind, vp::g1_cadd_g16_ops, vp::g1_cadd_g16_nlayers, lg);
        }
        if (d2 != 0) {
            if (active && lg == 0) { ws[SQX] = bx; ws[SQY] = d2 < 0 ? neg(q.y) : q.y; ws[SQZ] = Fp::one(); }
            vm_run(ws, vp::g1_cadd_g16_kind, vp::g1_cadd_g16_ops, vp::g1_cadd_g16_nlayers, lg);
        }
    }
    if (active && lg == 0) {
        G1J r = jac_inf<Fp>();
        if (!qinf) {
            const Fp X = ws[SX], Y = ws[SY], Z = ws[SZ];
            if (!Z.is_zero()) { r.x = fmul(X, Z); r.y = fmul(Y, fsqr(Z)); r.z = Z; }
        }
        out[i] = add_mixed(r, lo[i]);
    }
}

}  // namespace ripp
