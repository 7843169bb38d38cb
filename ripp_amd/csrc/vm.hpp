// Lane-parallel field VM: G lanes of a wave cooperate on ONE element; every Fp value of the element lives in a
// per-element LDS workspace; a program is a list of homogeneous LAYERS (tables generated and verified offline by
// tools/vmgen.py) in which each lane performs one
//     MUL  ws[dst] = (+-ws[a0] +-ws[a1]) * (+-ws[a2] +-ws[a3])
//     LIN  ws[dst] = sum_t c[t] * ws[s[t]]              (<= 16 terms, small signed integer coefficients)
// Why: the scalar kernels (one lane per element) keep ~400 dwords of live state per lane, so they run one wave per
// SIMD and, in the small late rounds of a proof, one lone lane pays ~2.4 us per DEPENDENT Fp product.  Here the
// state sits in LDS (160 KB/CU), each lane needs ~60-100 VGPRs, the multiplier is inlined once (no calls), and the
// 25-54 independent products of a step run side by side: a Miller doubling step is 2 product layers instead of 25
// dependent products.  Used for the latency-bound part of the pipeline; the scalar kernels remain the throughput
// path.
//
// Second form (build round 2): the workspace holds values in the CARRY-FREE radix of fq28.hpp (14 limbs of 28 bits, Montgomery
// R' = 2^392).  A lone wave issues one instruction per ~8 cycles whatever it is, so latency == instruction count: the product is
// ~530 instructions instead of ~1 100 (the 12 x 32-bit form pays a carry instruction and a hazard nop per limb product), and a
// linear combination with ARBITRARY small coefficients is one pass of signed multiply-adds into 64-bit columns, so the chains of
// +-1 additions / doublings / halvings between two product layers collapse into ONE layer (Miller doubling step: 9 -> 4 layers).
//
// Values cross the kernel boundary WITHOUT a Montgomery conversion: the 12 x u32 words of an engine value a R (R = 2^384) are simply
// re-sliced into limbs, i.e. read as the R'-form of a * 2^-8.  Every program that runs on such inputs is HOMOGENEOUS in them
// (projective group law in and out, bilinear Fp12 product), so the constant 2^-8 only rescales a projective representative resp.
// multiplies an Fp12 value by an element of Fp, which the final exponentiation removes.  The one inhomogeneous program (the mixed
// addition step of the Miller loop) gets its affine point properly converted, once per pair.
#pragma once
#include <hip/hip_runtime.h>
#include "bls12_381/pairing.hpp"
#include "kernels.hpp"
#include "fq28.hpp"

namespace ripp {

struct VmOp { unsigned char dst, flags; unsigned short nbias; unsigned char s[16]; signed char c[16]; };   // 36 bytes
struct alignas(16) VmSlot { uint32_t l[16]; };          // 14 limbs + padding: four 16-byte LDS accesses
}  // namespace ripp
#if defined(RIPP_BLS12_377)
#include "bls12_377/vm_programs.inc"      // the same programs for Fp2 = Fp[u]/(u^2 + 5), xi = u, the D-type twist and b = 1 (tools/vmgen.py, curve "bls12_377")
#else
#include "vm_programs.inc"
#endif
namespace ripp {

constexpr int VM_G = 16;                 // lanes per element
constexpr int VM_EPW = 64 / VM_G;        // elements per wave
constexpr int VM_SLOT_VB = 16;           // every workspace value is < 16 p with normalised limbs (vmgen.py: LIGHT_MAX)
constexpr int VM_NEG_K = 17;             // a negated MUL operand term is K17 - x, K17 = 17 p (vmgen.py: NEG_K)

#if defined(__HIP_DEVICE_COMPILE__)
using VmVal = Fq<FQ_LN, VM_SLOT_VB>;
namespace fq28 {
// 17 p with every limb >= the corresponding limb of any workspace value: n_k + 2^28 [k < 13] - [k > 0]; the top limb of a workspace
// value is bounded by its value (< 16 p)
constexpr Limbs neg_bias() { Limbs n = times_p(VM_NEG_K), r{}; for (int k = 0; k < NL; ++k) r.l[k] = n.l[k] + (k < NL - 1 ? (1u << W) : 0u) - (k > 0 ? 1u : 0u); return r; }
constexpr Limbs K17 = neg_bias();
static_assert((uint64_t)K17.l[NL - 1] >= (uint64_t)VM_SLOT_VB * (P_TOP + 1), "K17's top limb must dominate a workspace value's");
constexpr float INV_PTOP = (1.0f - 1.0f / 1048576.0f) / (float)(P_TOP + 1);     // quotient estimate of a heavy LIN: never above the true quotient
}  // namespace fq28

// Bank layout.  A slot is 64 B, half a 128-byte bank row: in a layer the 16 lanes of a group read 16 DIFFERENT slots at the same chunk index, so
// with the chunks in their natural places all even slots hit one quarter of the banks and all odd slots another -- SQ_LDS_BANK_CONFLICT was 11-31 %
// of the wave-cycles of the VM kernels (profiles/r03_sq_counters_pmc.csv), pure latency on a lone wave.  Chunk c of slot s therefore sits at
// position (c + s / 2) mod 4: eight consecutive slots cover the eight 16-byte positions of a bank row.  Every access goes through vm_ld / vm_st.
#if defined(RIPP_VM_NO_SWIZZLE)
__device__ __forceinline__ unsigned vm_rot(unsigned) { return 0u; }            // natural chunk order (A/B builds)
#else
__device__ __forceinline__ unsigned vm_rot(unsigned s) { return (s >> 1) & 3u; }
#endif
__device__ __forceinline__ VmVal vm_ld(const VmSlot* ws, unsigned s) {
    VmVal r; uint4 q[4];
    const uint4* p = reinterpret_cast<const uint4*>(ws[s].l);
    const unsigned rot = vm_rot(s);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = p[(i + rot) & 3u];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
#pragma unroll
    for (int i = 0; i < fq28::NL; ++i) r.l[i] = w[i];
    return r;
}
template <uint64_t LM, int VB>
__device__ __forceinline__ void vm_st(VmSlot* ws, unsigned s, const Fq<LM, VB>& v) {
    static_assert(LM <= FQ_LN && VB <= VM_SLOT_VB, "workspace values are normalised and < 16 p");
    uint4 q[4]; uint32_t* w = reinterpret_cast<uint32_t*>(q);
#pragma unroll
    for (int i = 0; i < fq28::NL; ++i) w[i] = v.l[i];
    w[14] = 0; w[15] = 0;
    uint4* p = reinterpret_cast<uint4*>(ws[s].l);
    const unsigned rot = vm_rot(s);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[(i + rot) & 3u] = q[i];
}
// kernel-side access to the workspace: engine values are RE-SLICED, not converted (see the header note)
__device__ __forceinline__ void vm_put(VmSlot* ws, int slot, const Fp& v) { vm_st(ws, slot, fq_unpack(v.l)); }
__device__ __forceinline__ void vm_put_converted(VmSlot* ws, int slot, const Fp& v) { vm_st(ws, slot, fq_from_fp(v)); }
__device__ __forceinline__ void vm_put_raw(VmSlot* ws, int slot, const Fqn& v) { vm_st(ws, slot, v); }
__device__ __forceinline__ void vm_zero(VmSlot* ws) { vm_st(ws, 0, fq_zero()); }
// program outputs are < 2p (vmgen.py reduces them): canonical representative, packed
__device__ __forceinline__ Fp vm_get(const VmSlot* ws, int slot) {
    const VmVal t = vm_ld(ws, slot);
    Fqn u;
#pragma unroll
    for (int i = 0; i < fq28::NL; ++i) u.l[i] = t.l[i];
    const Fqn c = fq_canon(u);
    Fp r; fq_pack(c, r.l); return r;
}

// one operand of a MUL: (+-ws[s0]) (+-ws[s1]); "second term absent" == slot index 0 (which holds zero)
__device__ __forceinline__ Fq<((uint64_t)1 << 30), 2 * VM_NEG_K> vm_operand(const VmSlot* ws, unsigned s0, unsigned s1, bool n0, bool n1) {
    using namespace fq28;
    Fq<((uint64_t)1 << 30), 2 * VM_NEG_K> r;
    const VmVal x = vm_ld(ws, s0);
    if (__any(n0)) {
#pragma unroll
        for (int i = 0; i < NL; ++i) r.l[i] = n0 ? K17.l[i] - x.l[i] : x.l[i];
    } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) r.l[i] = x.l[i];
    }
    if (__any(s1 != 0u)) {
        const VmVal y = vm_ld(ws, s1);
        if (__any(n1)) {
#pragma unroll
            for (int i = 0; i < NL; ++i) r.l[i] += n1 ? K17.l[i] - y.l[i] : y.l[i];
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) r.l[i] += y.l[i];
        }
    }
    return r;
}
// signed carry pass over 64-bit columns whose total is >= 0: limbs < 2^28, the top limb keeps the rest
__device__ __forceinline__ void vm_carry(const int64_t (&col)[fq28::NL], uint32_t (&r)[fq28::NL]) {
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < fq28::NL - 1; ++i) { const int64_t t = col[i] + carry; r[i] = (uint32_t)t & fq28::MASK; carry = t >> fq28::W; }
    r[fq28::NL - 1] = (uint32_t)(col[fq28::NL - 1] + carry);
}

// Run one program on this lane's element.  `ws` = LDS workspace of the element, `lg` = lane index within the group.
// Every instruction of a layer is latency, so the operand preparation a layer does not need (negations, second terms, the reduction of
// a LIN whose bound is small) is skipped by WAVE-UNIFORM tests -- no divergence.
__device__ __forceinline__ void vm_run(VmSlot* ws, const unsigned char* __restrict__ kind, const VmOp* __restrict__ ops, int nlayers, int lg) {
    using namespace fq28;
    // The layers rely on the wave running them in LOCKSTEP (every lane's reads of a layer are issued before any lane's writes, LDS is in
    // order).  The callers bracket vm_run with lane-0-only slot writes; without a CONVERGENT operation here the compiler may thread the
    // `lg == 0` condition through a program that is not inside a loop and run it once for lane 0 and once for the other lanes, which
    // breaks that assumption (seen in k_vm_scale_g1's first doubling).  wave_barrier is convergent and emits no instruction.
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int l = 0; l < nlayers; ++l) {
        __builtin_amdgcn_wave_barrier();
        const VmOp* __restrict__ opp = ops + (l * VM_G + lg);
        const unsigned k = kind[l];
        const unsigned dst = opp->dst;
        if (k == 0) {                                          // MUL layer (uniform over the wave)
            const unsigned f = opp->flags;
            const auto A = vm_operand(ws, opp->s[0], opp->s[1], f & 1u, f & 2u);
            const auto B = vm_operand(ws, opp->s[2], opp->s[3], f & 4u, f & 8u);
            const Fqn r = fq_mul(A, B);
            vm_st(ws, dst, r);     // all lanes of the wave have issued their reads of this layer before any write (lockstep, in-order LDS)
        } else {                                                // LIN layer: k & 31 terms, bit 6 = reduce the results below 2p
            const int nt = (int)(k & 31u);
            int64_t col[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) col[i] = 0;
#pragma unroll 1
            for (int t = 0; t < nt; ++t) {
                const int c = opp->c[t];
                const VmVal x = vm_ld(ws, opp->s[t]);
#pragma unroll
                for (int i = 0; i < NL; ++i) col[i] += (int64_t)c * (int32_t)x.l[i];
            }
            const int nb = opp->nbias;                          // + nb p: covers the negative terms, so the total is >= 0
#pragma unroll
            for (int i = 0; i < NL; ++i) col[i] += (int64_t)nb * (int32_t)P28.l[i];
            uint32_t r[NL];
            vm_carry(col, r);
            if (k & 64u) {                                      // value < ~1000 p: subtract q p, q = the estimate from the top limb (never too large, at most one too small)
                const int q = (int)((float)r[NL - 1] * INV_PTOP);
#pragma unroll
                for (int i = 0; i < NL; ++i) col[i] = (int64_t)r[i] - (int64_t)q * (int32_t)P28.l[i];
                vm_carry(col, r);
            }
            Fq<FQ_LN, VM_SLOT_VB> o;
#pragma unroll
            for (int i = 0; i < NL; ++i) o.l[i] = r[i];
            vm_st(ws, dst, o);
        }
    }
    __builtin_amdgcn_wave_barrier();
}
#else
struct VmValHost {};
__device__ void vm_run(VmSlot* ws, const unsigned char* kind, const VmOp* ops, int nlayers, int lg);
__device__ void vm_put(VmSlot* ws, int slot, const Fp& v);
__device__ void vm_put_converted(VmSlot* ws, int slot, const Fp& v);
__device__ void vm_zero(VmSlot* ws);
__device__ Fp vm_get(const VmSlot* ws, int slot);
#endif

// ---- stage 1 of the pairing product in latency form: VM_G lanes per (P,Q) pair ---------------------------------
// grid.x covers ceil(M / (4 * VM_EPW)) blocks of 256 threads, grid.y = product; same line buffer layout as k_miller_lines.
constexpr int VM_LINES_SLOTS = (vmprog::line_double_g16_nslots > vmprog::line_add_g16_nslots) ? vmprog::line_double_g16_nslots : vmprog::line_add_g16_nslots;
__global__ void __launch_bounds__(256) k_vm_miller_lines(PairSets ps, uint32_t M, uint4* __restrict__ lines, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;       // pair index handled by this group
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_LINES_SLOTS;
    const bool active = i < M;
    const G1A* __restrict__ a = ps.a[blockIdx.y];
    const G2A* __restrict__ b = ps.b[blockIdx.y];
    namespace vp = vmprog;
    bool skip = false;
    // This program set is NOT homogeneous in its inputs (mixed addition with an affine Q; the line's free coefficient against the ones scaled
    // by xP, yP), so the six input coordinates are properly converted to the workspace's Montgomery form -- one product each, on six
    // lanes side by side, once per pair.  Lanes 0..3 keep their coordinate of Q in registers for the five addition steps.
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn qc = fq_zero();
    if (active) {
        const G1A P = a[i]; const G2A Q = b[i];
        skip = is_inf(P) || is_inf(Q);
        if (lg < 4) {
            qc = fq_from_fp(lg == 0 ? Q.x.c0 : lg == 1 ? Q.x.c1 : lg == 2 ? Q.y.c0 : Q.y.c1);
            vm_put_raw(ws, lg == 0 ? vp::line_double_g16_in_X0 : lg == 1 ? vp::line_double_g16_in_X1 : lg == 2 ? vp::line_double_g16_in_Y0 : vp::line_double_g16_in_Y1, qc);
        }
        if (lg == 4) { vm_zero(ws); vm_put_raw(ws, vp::line_double_g16_in_Z0, fq_one()); vm_put_raw(ws, vp::line_double_g16_in_Z1, fq_zero()); }
        if (lg == 5) vm_put_converted(ws, vp::line_double_g16_in_xP, P.x);
        if (lg == 6) vm_put_converted(ws, vp::line_double_g16_in_yP, P.y);
    }
#endif
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
            Fp v = vm_get(ws, src);
            if (skip) v = (lg == 0) ? Fp::one() : Fp::zero();
            const uint4* pv = reinterpret_cast<const uint4*>(&v);
#pragma unroll
            for (int c = 0; c < 3; ++c) lines[((row0 + s) * LINE_CHUNKS + 3 * lg + c) * stride + i] = pv[c];
        }
        ++s;
        if ((BLS_X_ABS >> bit) & 1ull) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (active && lg < 4) vm_put_raw(ws, lg == 0 ? vp::line_add_g16_in_qx0 : lg == 1 ? vp::line_add_g16_in_qx1 : lg == 2 ? vp::line_add_g16_in_qy0 : vp::line_add_g16_in_qy1, qc);
#endif
            vm_run(ws, vp::line_add_g16_kind, vp::line_add_g16_ops, vp::line_add_g16_nlayers, lg);
            if (active && lg < 6) {
                constexpr int OUT[6] = {vp::line_add_g16_out_L00, vp::line_add_g16_out_L01, vp::line_add_g16_out_L10, vp::line_add_g16_out_L11, vp::line_add_g16_out_L20, vp::line_add_g16_out_L21};
                int src = OUT[0];
#pragma unroll
                for (int k = 1; k < 6; ++k) src = (lg == k) ? OUT[k] : src;
                Fp v = vm_get(ws, src);
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
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t j = (blockIdx.x * (blockDim.x >> 6) + wave) * VM_EPW + grp;
    const size_t row = blockIdx.y;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_F12_SLOTS;
    const bool active = j < Tout, pair = active && (j + Tout < Tin);
    namespace vp = vmprog;
    if (active && lg < 12) {      // Fp k of an Fp12 = chunks 3k..3k+2; f -> slots 2.., g -> slots 14..
        Fp f; uint4* pf = reinterpret_cast<uint4*>(&f);
#pragma unroll
        for (int c = 0; c < 3; ++c) pf[c] = in[(row * FP12_CHUNKS + 3 * lg + c) * Tin + j];
        vm_put(ws, vp::fp12_mul_g16_in[0] + lg, f);
        if (pair) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pf[c] = in[(row * FP12_CHUNKS + 3 * lg + c) * Tin + j + Tout];
            vm_put(ws, vp::fp12_mul_g16_in[12] + lg, f);
        }
        if (lg == 0) vm_zero(ws);
    }
    // groups without a partner (odd tail) just copy; the program still runs wave-uniformly on harmless data
    vm_run(ws, vp::fp12_mul_g16_kind, vp::fp12_mul_g16_ops, vp::fp12_mul_g16_nlayers, lg);
    if (active && lg < 12) {
        Fp f;
        if (pair) f = vm_get(ws, vp::fp12_mul_g16_out[0] + lg);
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
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    const int j = blockIdx.y;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G2_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g2_cadd_g16_in_X0, SY = vp::g2_cadd_g16_in_Y0, SZ = vp::g2_cadd_g16_in_Z0, SQX = vp::g2_cadd_g16_in_qx0, SQY = vp::g2_cadd_g16_in_qy0, SQZ = vp::g2_cadd_g16_in_qz0 };
    static_assert(vp::g2_hdbl_g16_in_X0 == SX && vp::g2_hdbl_g16_in_Y0 == SY && vp::g2_hdbl_g16_in_Z0 == SZ, "accumulator slots shared by the two programs");
    G2A q = aff_inf<Fp2>(); bool qinf = true;
    if (active && lg == 0) { q = gls_image(hi[i], j); qinf = is_inf(q); vm_zero(ws); }
    int pos = dg.len - 1;
    while (pos >= 0 && dg.d[j][pos] == 0) --pos;               // uniform: digits are shared by the whole launch
    const bool any = pos >= 0;
    if (any && active && lg == 0) {
        const Fp2 y0 = dg.d[j][pos] < 0 ? neg(q.y) : q.y;
        vm_put(ws, SX, q.x.c0); vm_put(ws, SX + 1, q.x.c1); vm_put(ws, SY, y0.c0); vm_put(ws, SY + 1, y0.c1); vm_put(ws, SZ, Fp::one()); vm_put(ws, SZ + 1, Fp::zero());
    }
    // additions use the COMPLETE projective law (g2_cadd: depth 2, no exceptional case), the addend (q.x : +-q.y : 1) is rewritten
    // before each one because the doubling program may use those slots as temporaries
    (void)flag;
#pragma unroll 1
    for (--pos; pos >= 0; --pos) {
        vm_run(ws, vp::g2_hdbl_g16_kind, vp::g2_hdbl_g16_ops, vp::g2_hdbl_g16_nlayers, lg);
        const int d = dg.d[j][pos];
        if (d != 0) {
            if (active && lg == 0) { const Fp2 y = d < 0 ? neg(q.y) : q.y; vm_put(ws, SQX, q.x.c0); vm_put(ws, SQX + 1, q.x.c1); vm_put(ws, SQY, y.c0); vm_put(ws, SQY + 1, y.c1); vm_put(ws, SQZ, Fp::one()); vm_put(ws, SQZ + 1, Fp::zero()); }
            vm_run(ws, vp::g2_cadd_g16_kind, vp::g2_cadd_g16_ops, vp::g2_cadd_g16_nlayers, lg);
        }
    }
    if (active && lg == 0) {
        // parts are left in HOMOGENEOUS projective form (X : Y : Z), the identity as (0 : 1 : 0): k_vm_combine_g2 adds them with the
        // complete addition program, so no lone-lane conversion sits on the chain
        G2J r; r.x = Fp2::zero(); r.y = Fp2::one(); r.z = Fp2::zero();
        if (any && !qinf) { r.x = {vm_get(ws, SX), vm_get(ws, SX + 1)}; r.y = {vm_get(ws, SY), vm_get(ws, SY + 1)}; r.z = {vm_get(ws, SZ), vm_get(ws, SZ + 1)}; }
        parts[(size_t)j * half + i] = r;
    }
}

// out[i] = parts[0][i] + parts[1][i] + parts[2][i] + parts[3][i] + lo[i]  (Jacobian out): four complete VM additions instead of the
// lone-lane Jacobian additions of k_fold_g2_combine (0.35-0.43 ms on the fold chain of every small round)
__global__ void __launch_bounds__(256) k_vm_combine_g2(const G2J* __restrict__ parts, const G2A* __restrict__ lo, uint32_t half, G2J* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G2_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g2_cadd_g16_in_X0, SY = vp::g2_cadd_g16_in_Y0, SZ = vp::g2_cadd_g16_in_Z0, SQX = vp::g2_cadd_g16_in_qx0, SQY = vp::g2_cadd_g16_in_qy0, SQZ = vp::g2_cadd_g16_in_qz0 };
    auto put = [&](int s, const Fp2& v) { vm_put(ws, s, v.c0); vm_put(ws, s + 1, v.c1); };
    if (lg == 0) {
        vm_zero(ws);
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
        const Fp2 X = {vm_get(ws, SX), vm_get(ws, SX + 1)}, Y = {vm_get(ws, SY), vm_get(ws, SY + 1)}, Z = {vm_get(ws, SZ), vm_get(ws, SZ + 1)};
        G2J r = jac_inf<Fp2>();
        if (!Z.is_zero()) { r.x = mul(X, Z); r.y = mul(Y, sqr(Z)); r.z = Z; }          // (X/Z, Y/Z) -> Jacobian (XZ, YZ^2, Z)
        out[i] = r;
    }
}

// G2 fold for MID-SIZE rounds (a few thousand outputs): ONE group per element walks the four GLS digit strings jointly -- 65 VM doublings and
// ~87 complete additions on one dependent chain (~1.4 ms), where the 4-lane scalar form (k_fold_g2_gls_split) needs ~4.7 ms for its 65
// lone-lane doublings whatever the size, and the split-by-string VM form above spends 8 groups per element.  Lane j < 4 of the group keeps
// psi^j(hi[i]) in registers and writes the addend when string j has a digit; the last addition brings lo[i] in.  out[i] Jacobian.
__global__ void __launch_bounds__(256) k_vm_fold_g2_joint(const G2A* __restrict__ hi, const G2A* __restrict__ lo, uint32_t half, GlsDigits dg, G2J* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G2_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g2_cadd_g16_in_X0, SY = vp::g2_cadd_g16_in_Y0, SZ = vp::g2_cadd_g16_in_Z0, SQX = vp::g2_cadd_g16_in_qx0, SQY = vp::g2_cadd_g16_in_qy0, SQZ = vp::g2_cadd_g16_in_qz0 };
    auto put2 = [&](int s, const Fp2& v) { vm_put(ws, s, v.c0); vm_put(ws, s + 1, v.c1); };
    G2A q = aff_inf<Fp2>(); bool qinf = true;
    if (active && lg < 4) { q = gls_image(hi[i], lg); qinf = is_inf(q); }
    if (lg == 0) { vm_zero(ws); put2(SX, Fp2::zero()); put2(SY, Fp2::one()); put2(SZ, Fp2::zero()); }      // T = identity (0 : 1 : 0)
    int pos = dg.len - 1;
    while (pos >= 0 && dg.d[0][pos] == 0 && dg.d[1][pos] == 0 && dg.d[2][pos] == 0 && dg.d[3][pos] == 0) --pos;      // uniform
    bool first = true;
#pragma unroll 1
    for (; pos >= 0; --pos) {
        if (!first) vm_run(ws, vp::g2_hdbl_g16_kind, vp::g2_hdbl_g16_ops, vp::g2_hdbl_g16_nlayers, lg);
        first = false;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int d = dg.d[j][pos];
            if (d == 0) continue;
            if (lg == j) {                                                   // the addend (x : +-y : 1), or the identity for an element at infinity
                if (qinf) { put2(SQX, Fp2::zero()); put2(SQY, Fp2::one()); put2(SQZ, Fp2::zero()); }
                else { put2(SQX, q.x); put2(SQY, d < 0 ? neg(q.y) : q.y); put2(SQZ, Fp2::one()); }
            }
            vm_run(ws, vp::g2_cadd_g16_kind, vp::g2_cadd_g16_ops, vp::g2_cadd_g16_nlayers, lg);
        }
    }
    if (lg == 0) {
        G2A l = aff_inf<Fp2>(); if (active) l = lo[i];
        if (is_inf(l)) { put2(SQX, Fp2::zero()); put2(SQY, Fp2::one()); put2(SQZ, Fp2::zero()); }
        else { put2(SQX, l.x); put2(SQY, l.y); put2(SQZ, Fp2::one()); }
    }
    vm_run(ws, vp::g2_cadd_g16_kind, vp::g2_cadd_g16_ops, vp::g2_cadd_g16_nlayers, lg);
    if (active && lg == 0) {
        const Fp2 X = {vm_get(ws, SX), vm_get(ws, SX + 1)}, Y = {vm_get(ws, SY), vm_get(ws, SY + 1)}, Z = {vm_get(ws, SZ), vm_get(ws, SZ + 1)};
        G2J r = jac_inf<Fp2>();
        if (!Z.is_zero()) { r.x = mul(X, Z); r.y = mul(Y, sqr(Z)); r.z = Z; }          // (X/Z, Y/Z) -> Jacobian (XZ, YZ^2, Z)
        out[i] = r;
    }
}

// The same joint walk over TWO bases with a full-width scalar each: out[i] = s1 * p1[i] + s2 * p2[i] (8 GLS digit strings; lanes 0..3 keep the
// images of p1[i], lanes 4..7 those of p2[i]).  This is the fold that returns an x-scaled G2 vector to the plain one (engine.hip job_fold).
struct GlsDigits2 { GlsDigits a, b; };
__global__ void __launch_bounds__(256) k_vm_fold_g2_joint2(const G2A* __restrict__ p1, const G2A* __restrict__ p2, uint32_t half, GlsDigits2 dg, G2J* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G2_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g2_cadd_g16_in_X0, SY = vp::g2_cadd_g16_in_Y0, SZ = vp::g2_cadd_g16_in_Z0, SQX = vp::g2_cadd_g16_in_qx0, SQY = vp::g2_cadd_g16_in_qy0, SQZ = vp::g2_cadd_g16_in_qz0 };
    auto put2 = [&](int s, const Fp2& v) { vm_put(ws, s, v.c0); vm_put(ws, s + 1, v.c1); };
    G2A q = aff_inf<Fp2>(); bool qinf = true;
    if (active && lg < 8) { q = gls_image(lg < 4 ? p1[i] : p2[i], lg & 3); qinf = is_inf(q); }
    if (lg == 0) { vm_zero(ws); put2(SX, Fp2::zero()); put2(SY, Fp2::one()); put2(SZ, Fp2::zero()); }      // T = identity (0 : 1 : 0)
    const int len = dg.a.len > dg.b.len ? dg.a.len : dg.b.len;
    auto digit = [&](int t, int pos) -> int { const GlsDigits& g = t < 4 ? dg.a : dg.b; return pos < g.len ? g.d[t & 3][pos] : 0; };
    bool first = true;
#pragma unroll 1
    for (int pos = len - 1; pos >= 0; --pos) {
        if (!first) vm_run(ws, vp::g2_hdbl_g16_kind, vp::g2_hdbl_g16_ops, vp::g2_hdbl_g16_nlayers, lg);
#pragma unroll 1
        for (int t = 0; t < 8; ++t) {
            const int d = digit(t, pos);
            if (d == 0) continue;
            first = false;
            if (lg == t) {
                if (qinf) { put2(SQX, Fp2::zero()); put2(SQY, Fp2::one()); put2(SQZ, Fp2::zero()); }
                else { put2(SQX, q.x); put2(SQY, d < 0 ? neg(q.y) : q.y); put2(SQZ, Fp2::one()); }
            }
            vm_run(ws, vp::g2_cadd_g16_kind, vp::g2_cadd_g16_ops, vp::g2_cadd_g16_nlayers, lg);
        }
    }
    if (active && lg == 0) {
        const Fp2 X = {vm_get(ws, SX), vm_get(ws, SX + 1)}, Y = {vm_get(ws, SY), vm_get(ws, SY + 1)}, Z = {vm_get(ws, SZ), vm_get(ws, SZ + 1)};
        G2J r = jac_inf<Fp2>();
        if (!Z.is_zero()) { r.x = mul(X, Z); r.y = mul(Y, sqr(Z)); r.z = Z; }
        out[i] = r;
    }
}

// G1: single NAF digit string (the 128-bit SIPP challenge); out[i] = s*hi[i] + lo[i] (Jacobian)
constexpr int VM_G1_SLOTS = (vmprog::g1_hdbl_g16_nslots > vmprog::g1_cadd_g16_nslots) ? vmprog::g1_hdbl_g16_nslots : vmprog::g1_cadd_g16_nslots;
__global__ void __launch_bounds__(256) k_vm_fold_g1(const G1A* __restrict__ hi, const G1A* __restrict__ lo, uint32_t half, NafDigits dg, G1J* __restrict__ out, uint32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G1_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g1_cadd_g16_in_X0, SY = vp::g1_cadd_g16_in_Y0, SZ = vp::g1_cadd_g16_in_Z0, SQX = vp::g1_cadd_g16_in_qx0, SQY = vp::g1_cadd_g16_in_qy0, SQZ = vp::g1_cadd_g16_in_qz0 };
    static_assert(vp::g1_hdbl_g16_in_X0 == SX && vp::g1_hdbl_g16_in_Y0 == SY && vp::g1_hdbl_g16_in_Z0 == SZ, "accumulator slots shared by the two programs");
    G1A q = aff_inf<Fp>(); bool qinf = true;
    if (active && lg == 0) { q = hi[i]; qinf = is_inf(q); vm_zero(ws); }
    int pos = dg.len - 1;
    while (pos >= 0 && dg.d[pos] == 0) --pos;
    const bool any = pos >= 0;
    if (any && active && lg == 0) { vm_put(ws, SX, q.x); vm_put(ws, SY, dg.d[pos] < 0 ? neg(q.y) : q.y); vm_put(ws, SZ, Fp::one()); }
    (void)flag;                                                 // complete addition (g1_cadd): nothing to report
#pragma unroll 1
    for (--pos; pos >= 0; --pos) {
        vm_run(ws, vp::g1_hdbl_g16_kind, vp::g1_hdbl_g16_ops, vp::g1_hdbl_g16_nlayers, lg);
        const int d = dg.d[pos];
        if (d != 0) {
            if (active && lg == 0) { vm_put(ws, SQX, q.x); vm_put(ws, SQY, d < 0 ? neg(q.y) : q.y); vm_put(ws, SQZ, Fp::one()); }
            vm_run(ws, vp::g1_cadd_g16_kind, vp::g1_cadd_g16_ops, vp::g1_cadd_g16_nlayers, lg);
        }
    }
    if (active && lg == 0) {
        G1J r = jac_inf<Fp>();
        if (any && !qinf) {
            const Fp X = vm_get(ws, SX), Y = vm_get(ws, SY), Z = vm_get(ws, SZ);
            if (!Z.is_zero()) { r.x = fmul(X, Z); r.y = fmul(Y, fsqr(Z)); r.z = Z; }
        }
        out[i] = add_mixed(r, lo[i]);
    }
}

// G1 fold with a full-width scalar in latency form: the GLV halves of kernels.hpp::k_fold_g1_glv on the field VM
// (128 VM doublings, complete additions of +-P and +-phi(P)).
__global__ void __launch_bounds__(256) k_vm_fold_g1_glv(const G1A* __restrict__ hi, const G1A* __restrict__ lo, uint32_t half, GlvDigits dg, G1J* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * VM_G1_SLOTS;
    const bool active = i < half;
    namespace vp = vmprog;
    enum { SX = vp::g1_cadd_g16_in_X0, SY = vp::g1_cadd_g16_in_Y0, SZ = vp::g1_cadd_g16_in_Z0, SQX = vp::g1_cadd_g16_in_qx0, SQY = vp::g1_cadd_g16_in_qy0, SQZ = vp::g1_cadd_g16_in_qz0 };
    G1A q = aff_inf<Fp>(); Fp bx = Fp::zero(); bool qinf = true;
    if (active && lg == 0) { q = hi[i]; qinf = is_inf(q); bx = fmul(q.x, fp_const(RIPP_GLV_BETA)); vm_zero(ws); vm_put(ws, SX, Fp::zero()); vm_put(ws, SY, Fp::one()); vm_put(ws, SZ, Fp::zero()); }   // T = identity (0:1:0)
#pragma unroll 1
    for (int pos = dg.len - 1; pos >= 0; --pos) {
        vm_run(ws, vp::g1_hdbl_g16_kind, vp::g1_hdbl_g16_ops, vp::g1_hdbl_g16_nlayers, lg);
        const int d1 = dg.d1[pos], d2 = dg.d2[pos];
        if (d1 != 0) {
            if (active && lg == 0) { vm_put(ws, SQX, q.x); vm_put(ws, SQY, d1 < 0 ? neg(q.y) : q.y); vm_put(ws, SQZ, Fp::one()); }
            vm_run(ws, vp::g1_cadd_g16_kind, vp::g1_cadd_g16_ops, vp::g1_cadd_g16_nlayers, lg);
        }
        if (d2 != 0) {
            if (active && lg == 0) { vm_put(ws, SQX, bx); vm_put(ws, SQY, d2 < 0 ? neg(q.y) : q.y); vm_put(ws, SQZ, Fp::one()); }
            vm_run(ws, vp::g1_cadd_g16_kind, vp::g1_cadd_g16_ops, vp::g1_cadd_g16_nlayers, lg);
        }
    }
    if (active && lg == 0) {
        G1J r = jac_inf<Fp>();
        if (!qinf) {
            const Fp X = vm_get(ws, SX), Y = vm_get(ws, SY), Z = vm_get(ws, SZ);
            if (!Z.is_zero()) { r.x = fmul(X, Z); r.y = fmul(Y, fsqr(Z)); r.z = Z; }
        }
        out[i] = add_mixed(r, lo[i]);
    }
}

}  // namespace ripp
