// Latency form of the halving-round folds with a PRECOMPUTED SECOND BASE (small rounds of SIPP::prove, sipp/src/lib.rs:87-100).
// Between a round's pairing products and its challenge the GPU idles for the host's final exponentiations (~0.8 ms); in that gap
// k_vm_pow2 computes 2^K * hi[i] (K = 64 on G1, 32 on G2) on the field VM, left in homogeneous projective form -- the complete
// addition program takes projective addends, so nothing is normalised.  Once the challenge is known the scalar is split at bit K and
// every half runs on its own group of 16 lanes (k_vm_fold_split2: blockIdx.y = digit string), so the dependent chain of a fold is
// K doublings + ~K/3 additions instead of 2K + 2K/3; k_vm_combine adds the partial sums and lo with complete additions.
// Same group elements as the one-base forms, hence bit-identical proofs.
#pragma once
#include "msm.hpp"      // VmCurve<F>
#include "scale.hpp"    // scale_bias / scale_digit

namespace ripp {

struct SplitDigits { int8_t d[8][68]; int len; };     // NAF digit strings; G1: d[0] = low half, d[1] = high half; G2: d[j] low, d[4+j] high of GLS digit j

// image t of a base: G1: t = 1 is the GLV endomorphism phi(x, y) = (beta x, y) (full-width GIPA scalars); G2: psi^t (GLS)
__device__ __forceinline__ Jac<Fp> vm_image_h(const Jac<Fp>& p, int t) { return t == 0 ? p : Jac<Fp>{fmul(p.x, fp_const(RIPP_GLV_BETA)), p.y, p.z}; }
__device__ __forceinline__ Jac<Fp2> vm_image_h(const Jac<Fp2>& q, int j) {          // psi^j on homogeneous coordinates (x = X/Z, y = Y/Z)
    if (j == 0) return q;
    const G2A c = gls_image(G2A{Fp2::one(), Fp2::one()}, j);                          // (PSIj_CX, PSIj_CY): the image of (1, 1) is the constant pair
    const bool odd = (j & 1) != 0;
    return {mul(odd ? conj(q.x) : q.x, c.x), mul(odd ? conj(q.y) : q.y, c.y), odd ? conj(q.z) : q.z};
}
__device__ __forceinline__ G1A vm_image_a(const G1A& p, int t) { return t == 0 ? p : G1A{fmul(p.x, fp_const(RIPP_GLV_BETA)), p.y}; }
__device__ __forceinline__ G2A vm_image_a(const G2A& q, int j) { return gls_image(q, j); }

template <class F> __device__ __forceinline__ Jac<F> vm_identity_h() { return {F::zero(), F::one(), F::zero()}; }

// out_h[i] = 2^k * in[i], homogeneous projective
template <class F>
__global__ void __launch_bounds__(256) k_vm_pow2(const Affine<F>* __restrict__ in, uint32_t n, int k, Jac<F>* __restrict__ out_h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    const bool active = i < n;
    if (lg == 0) {
        vm_zero(ws);
        Jac<F> t = vm_identity_h<F>();
        if (active) { const Affine<F> p = in[i]; if (!is_inf(p)) t = {p.x, p.y, F::one()}; }
        C::put(ws, C::SX, t.x); C::put(ws, C::SY, t.y); C::put(ws, C::SZ, t.z);
    }
#pragma unroll 1
    for (int s = 0; s < k; ++s) C::dbl_(ws, lg);
    if (active && lg == 0) out_h[i] = {C::get(ws, C::SX), C::get(ws, C::SY), C::get(ws, C::SZ)};
}

// batched k_vm_pow2 (grid.y = vector)
template <class F> struct Pow2Batch { const Affine<F>* in[3]; Jac<F>* out_h[3]; };
template <class F>
__global__ void __launch_bounds__(256) k_vm_pow2_b(Pow2Batch<F> pb, uint32_t n, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G, v = blockIdx.y;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = reinterpret_cast<VmSlot*>(vm_smem) + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    const bool active = i < n;
    if (lg == 0) {
        vm_zero(ws);
        Jac<F> t = vm_identity_h<F>();
        if (active) { const Affine<F> p = pb.in[v][i]; if (!is_inf(p)) t = {p.x, p.y, F::one()}; }
        C::put(ws, C::SX, t.x); C::put(ws, C::SY, t.y); C::put(ws, C::SZ, t.z);
    }
#pragma unroll 1
    for (int s = 0; s < k; ++s) C::dbl_(ws, lg);
    if (active && lg == 0) pb.out_h[v][i] = {C::get(ws, C::SX), C::get(ws, C::SY), C::get(ws, C::SZ)};
}

// parts_h[t][i] = (digit string t) * base_t(i);  base_t = image t of hi[i] for t < nimg, image t - nimg of hi2_h[i] otherwise
template <class F>
__device__ __forceinline__ void vm_fold_split2_body(VmSlot* lds, const Affine<F>* __restrict__ hi, const Jac<F>* __restrict__ hi2_h, uint32_t half, const SplitDigits& dg, int nimg,
                                                    Jac<F>* __restrict__ parts_h) {
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    const int t = blockIdx.y;
    VmSlot* const ws = lds + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    const bool active = i < half;
    Jac<F> base = vm_identity_h<F>();                        // lane 0 of the group keeps the base (projective) in registers
    if (lg == 0) {
        vm_zero(ws);
        if (active) {
            if (t < nimg) { const Affine<F> p = vm_image_a(hi[i], t); if (!is_inf(p)) base = {p.x, p.y, F::one()}; }
            else base = vm_image_h(hi2_h[i], t - nimg);
        }
        C::put(ws, C::SX, F::zero()); C::put(ws, C::SY, F::one()); C::put(ws, C::SZ, F::zero());       // T = identity
    }
    int pos = dg.len - 1;
    while (pos >= 0 && dg.d[t][pos] == 0) --pos;             // uniform over the launch row: leading zeros cost nothing
#pragma unroll 1
    for (; pos >= 0; --pos) {
        C::dbl_(ws, lg);
        const int d = dg.d[t][pos];
        if (d != 0) {
            if (lg == 0) { C::put(ws, C::QX, base.x); C::put(ws, C::QY, d < 0 ? neg(base.y) : base.y); C::put(ws, C::QZ, base.z); }
            C::add_(ws, lg);
        }
    }
    if (active && lg == 0) parts_h[(size_t)t * half + i] = {C::get(ws, C::SX), C::get(ws, C::SY), C::get(ws, C::SZ)};
}
template <class F>
__global__ void __launch_bounds__(256) k_vm_fold_split2(const Affine<F>* __restrict__ hi, const Jac<F>* __restrict__ hi2_h, uint32_t half, SplitDigits dg, int nimg,
                                                         Jac<F>* __restrict__ parts_h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    vm_fold_split2_body<F>(reinterpret_cast<VmSlot*>(vm_smem), hi, hi2_h, half, dg, nimg, parts_h);
}

// Several vectors folded by ONE launch (grid.z = vector): a GIPA round folds up to three G1 and two G2 vectors of the same length, each a
// latency-bound chain, and streams beyond the hardware queues would run them one after the other.
constexpr int FOLD_BATCH_MAX = 3;
template <class F> struct FoldSet { const Affine<F>* hi; const Jac<F>* hi2_h; const Affine<F>* lo; Jac<F>* parts_h; Affine<F>* out; };
template <class F> struct FoldBatch { FoldSet<F> s[FOLD_BATCH_MAX]; SplitDigits dg[FOLD_BATCH_MAX]; };
template <class F>
__global__ void __launch_bounds__(256) k_vm_fold_split2_b(FoldBatch<F> fb, uint32_t half, int nimg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    const int v = blockIdx.z;
    vm_fold_split2_body<F>(reinterpret_cast<VmSlot*>(vm_smem), fb.s[v].hi, fb.s[v].hi2_h, half, fb.dg[v], nimg, fb.s[v].parts_h);
}

// T (in the group's VM workspace) = sum_{t < nparts} parts_h[t][i] + lo[i]
template <class F>
__device__ __forceinline__ void vm_combine_sum(VmSlot* ws, int lg, bool active, uint32_t i, const Jac<F>* __restrict__ parts_h, int nparts, const Affine<F>* __restrict__ lo, uint32_t half) {
    using C = VmCurve<F>;
    if (lg == 0) {
        vm_zero(ws);
        const Jac<F> p0 = active ? parts_h[i] : vm_identity_h<F>();
        C::put(ws, C::SX, p0.x); C::put(ws, C::SY, p0.y); C::put(ws, C::SZ, p0.z);
    }
#pragma unroll 1
    for (int t = 1; t <= nparts; ++t) {
        if (lg == 0) {
            Jac<F> q = vm_identity_h<F>();
            if (active) {
                if (t < nparts) q = parts_h[(size_t)t * half + i];
                else { const Affine<F> l = lo[i]; if (!is_inf(l)) q = {l.x, l.y, F::one()}; }
            }
            C::put(ws, C::QX, q.x); C::put(ws, C::QY, q.y); C::put(ws, C::QZ, q.z);
        }
        C::add_(ws, lg);
    }
}

// out[i] (Jacobian) = sum_{t < nparts} parts_h[t][i] + lo[i]
template <class F>
__global__ void __launch_bounds__(256) k_vm_combine(const Jac<F>* __restrict__ parts_h, int nparts, const Affine<F>* __restrict__ lo, uint32_t half, Jac<F>* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = reinterpret_cast<VmSlot*>(vm_smem) + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    const bool active = i < half;
    vm_combine_sum<F>(ws, lg, active, i, parts_h, nparts, lo, half);
    if (active && lg == 0) {
        const F X = C::get(ws, C::SX), Y = C::get(ws, C::SY), Z = C::get(ws, C::SZ);
        Jac<F> r = jac_inf<F>();
        if (!Z.is_zero()) { r.x = fmul(X, Z); r.y = fmul(Y, fsqr(Z)); r.z = Z; }              // (X/Z, Y/Z) -> Jacobian (XZ, YZ^2, Z)
        out[i] = r;
    }
}

// batched form (grid.y = vector) that also normalises: out[i] = affine(sum), one inversion per point on the group's first lane --
// for these lengths the separate normalisation kernel would run one inversion per lane as well, after one more launch
template <class F>
__global__ void __launch_bounds__(256) k_vm_combine_aff_b(FoldBatch<F> fb, int nparts, uint32_t half) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G, v = blockIdx.y;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = reinterpret_cast<VmSlot*>(vm_smem) + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    const bool active = i < half;
    vm_combine_sum<F>(ws, lg, active, i, fb.s[v].parts_h, nparts, fb.s[v].lo, half);
    if (active && lg == 0) {
        const F Z = C::get(ws, C::SZ);
        Affine<F> r = aff_inf<F>();
        if (!Z.is_zero()) { const F zi = finv(Z); r.x = fmul(C::get(ws, C::SX), zi); r.y = fmul(C::get(ws, C::SY), zi); }
        fb.s[v].out[i] = r;
    }
}

// ---- per-element scalar multiplication in latency form: out[i] = k[i] * base[i * base_stride] (Jacobian) ---------------------------------------
// The throughput kernel (scale.hpp k_scale_g1_glv) is one lane per element: 132 doublings + 66 additions of a LONE lane, ~3.9 ms whatever n.
// Here one VM group per element: the GLV halves of its own scalar as 33 signed base-16 digits each (the same recoding), a table of the
// multiples 1..8 of the base in the group's workspace (with -Y and beta X beside them, so a lookup is three slot copies), then 128 VM
// doublings and 66 complete additions (a zero digit adds the identity: uniform control flow).  ~1.3 ms for n <= 1 K elements.
constexpr int VM_SCALE_TAB = 8;
constexpr int VM_SCALE_SLOTS = VmCurve<Fp>::SLOTS + 5 * VM_SCALE_TAB;       // T_m = (X, Y, Z), -Y, beta X  for m = 1..8
__global__ void __launch_bounds__(256) k_vm_scale_g1(const G1A* __restrict__ base, uint32_t base_stride, const Fr* __restrict__ k_mont, uint32_t n, G1J* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
#if defined(__HIP_DEVICE_COMPILE__)
    using C = VmCurve<Fp>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G;
    const uint32_t i = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = reinterpret_cast<VmSlot*>(vm_smem) + (size_t)(wave * VM_EPW + grp) * VM_SCALE_SLOTS;
    const bool active = i < n;
    constexpr int TX = C::SLOTS, TY = TX + VM_SCALE_TAB, TZ = TY + VM_SCALE_TAB, TNY = TZ + VM_SCALE_TAB, TBX = TNY + VM_SCALE_TAB;
    uint32_t d1[5] = {0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u, 8u}, d2[5] = {0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u, 8u};     // all digits zero
    G1A p = aff_inf<Fp>();
    if (active && lg == 0) {
        Fr k = from_mont(k_mont[i]);
        const uint32_t lam[8] = RIPP_GLV_LAMBDA;
        const uint32_t lam_mu[5] = RIPP_GLV_LAMBDA_MU;                  // floor(2^256 / lambda)
        uint32_t rem[5];
        msm_divmod<4, 5>(k.l, lam, lam_mu, rem);                      // k = q * lambda + rem
        scale_bias(rem, d1); scale_bias(k.l, d2);
        p = base[(size_t)i * base_stride];
    }
    auto copy_slot = [&](int dst, int src) { vm_st(ws, dst, vm_ld(ws, src)); };
    if (lg == 0) {
        vm_zero(ws);
        const bool pinf = is_inf(p);                                  // the identity stays the identity: (0 : 1 : 0)
        C::put(ws, C::SX, pinf ? Fp::zero() : p.x); C::put(ws, C::SY, pinf ? Fp::one() : p.y); C::put(ws, C::SZ, pinf ? Fp::zero() : Fp::one());
        C::put(ws, C::QX, pinf ? Fp::zero() : p.x); C::put(ws, C::QY, pinf ? Fp::one() : p.y); C::put(ws, C::QZ, pinf ? Fp::zero() : Fp::one());
        copy_slot(TX, C::SX); copy_slot(TY, C::SY); copy_slot(TZ, C::SZ);
    }
    // table: T_2 = 2 T_1, T_(m+1) = T_m + T_1 (the addend slots are rewritten before every addition: the programs use them as scratch)
    // (this doubling is the one program call of the VM kernels that is not inside a loop: see the convergence note in vm_run)
    C::dbl_(ws, lg);
    if (lg == 0) { copy_slot(TX + 1, C::SX); copy_slot(TY + 1, C::SY); copy_slot(TZ + 1, C::SZ); }
#pragma unroll 1
    for (int m = 2; m < VM_SCALE_TAB; ++m) {
        if (lg == 0) { copy_slot(C::QX, TX); copy_slot(C::QY, TY); copy_slot(C::QZ, TZ); }
        C::add_(ws, lg);
        if (lg == 0) { copy_slot(TX + m, C::SX); copy_slot(TY + m, C::SY); copy_slot(TZ + m, C::SZ); }
    }
    if (lg < VM_SCALE_TAB) {                                         // -Y_m and beta X_m, one table row per lane
        vm_put(ws, TNY + lg, neg(vm_get(ws, TY + lg)));
        vm_put(ws, TBX + lg, fmul(vm_get(ws, TX + lg), fp_const(RIPP_GLV_BETA)));
    }
    if (lg == 0) { C::put(ws, C::SX, Fp::zero()); C::put(ws, C::SY, Fp::one()); C::put(ws, C::SZ, Fp::zero()); }      // acc = identity
#pragma unroll 1
    for (int j = 32; j >= 0; --j) {
        if (j != 32) { C::dbl_(ws, lg); C::dbl_(ws, lg); C::dbl_(ws, lg); C::dbl_(ws, lg); }
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            if (lg == 0) {
                const int d = scale_digit(h ? d2 : d1, j);
                if (d == 0) { C::put(ws, C::QX, Fp::zero()); C::put(ws, C::QY, Fp::one()); C::put(ws, C::QZ, Fp::zero()); }
                else { const int m = (d < 0 ? -d : d) - 1; copy_slot(C::QX, (h ? TBX : TX) + m); copy_slot(C::QY, (d < 0 ? TNY : TY) + m); copy_slot(C::QZ, TZ + m); }
            }
            C::add_(ws, lg);
        }
    }
    if (active && lg == 0) {
        const Fp X = C::get(ws, C::SX), Y = C::get(ws, C::SY), Z = C::get(ws, C::SZ);
        G1J r = jac_inf<Fp>();
        if (!Z.is_zero()) { r.x = fmul(X, Z); r.y = fmul(Y, fsqr(Z)); r.z = Z; }
        out[i] = r;
    }
#endif
}

}  // namespace ripp
