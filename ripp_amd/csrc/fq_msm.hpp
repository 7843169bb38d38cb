// The gathered mixed additions of the Pippenger MSM (msm.hpp k_msm_slot_sum: ~2/3 of an MSM's device time) on the carry-free field form.
//   * k_msm_extend_q: once per MSM, the `split` endomorphism images of every base (G1: P, phi(P); G2: Q, psi(Q), psi^2(Q), psi^3(Q)) are written
//     to an EXTENDED base array -- a term's base is then one load (k_msm_slot_sum formed phi / psi of a re-loaded base inside every addition:
//     1 resp. 6 extra Fp products per addition, and on G2 the operands of those products pushed the kernel into scratch).  The array holds
//     the coordinates already in the carry-free form (12 words of the canonical Montgomery-392 integer), so a gathered point is only re-sliced.
//   * k_msm_slot_sum_q: the slot loop with madd-2007-bl on 14 x 28-bit limbs (G1: fq_curve.hpp's jmadd_q; G2: the same formulas over Fp2
//     products written as two lazily reduced sums of two products, fq_miller.hpp) -- fewer instructions on an issue-bound kernel, no scratch
//     for G1.  Exceptional additions (T = +-Q) are DETECTED and that slot is redone with the complete formulas (msm_slot_sum_complete), like
//     everywhere the low-liveness additions are used (here by a second, tiny launch: k_msm_slot_sum_fix[_vm]).  Both curves (Fp2 products: fq_curve2.hpp FQ2_BETA).
#pragma once
#include "fq_miller.hpp"
#include "msm.hpp"

namespace ripp {

// an Fp / Fp2 coordinate of the extended array: 12 words per Fp (canonical integer of the Montgomery-392 form)
struct QFp { uint32_t w[12]; };
template <class F> struct QAff;
template <> struct alignas(16) QAff<Fp> { QFp x, y; };
template <> struct alignas(16) QAff<Fp2> { QFp x0, x1, y0, y1; };
static_assert(sizeof(QAff<Fp>) == sizeof(G1A) && sizeof(QAff<Fp2>) == sizeof(G2A), "the extended array has the footprint of split * n affine points");

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ QFp qfp_from(const Fp& v) { QFp r; fq_pack(fq_canon(fq_from_fp_fast(v)), r.w); return r; }
__device__ __forceinline__ Fqn qfp_get(const QFp& v) { return fq_unpack(v.w); }
__device__ __forceinline__ bool qfp_zero(const QFp& v) { uint32_t z = 0; for (int k = 0; k < 12; ++k) z |= v.w[k]; return z == 0; }

#endif

// ext[j * n + i] = image j of bases[i] in the carry-free form, j < split (1: the bases themselves; 2: + phi; 4: + psi, psi^2, psi^3); the identity (0, 0) stays (0, 0)
template <class F>
__global__ void __launch_bounds__(256) k_msm_extend_q(const Affine<F>* __restrict__ bases, uint32_t n, int split, QAff<F>* __restrict__ ext) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= n || j >= split) return;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<F, Fp>::value) {
        G1A q = bases[i];
        if (j == 1) q.x = fmul(q.x, fp_const(RIPP_GLV_BETA));
        QAff<Fp> o; o.x = qfp_from(q.x); o.y = qfp_from(q.y);
        ext[(size_t)j * n + i] = o;
    } else {
        const G2A q = gls_image(bases[i], j);
        QAff<Fp2> o; o.x0 = qfp_from(q.x.c0); o.x1 = qfp_from(q.x.c1); o.y0 = qfp_from(q.y.c0); o.y1 = qfp_from(q.y.c1);
        ext[(size_t)j * n + i] = o;
    }
#endif
}

// the terms [begin, end) of window w's sorted index array that slot s sums (the slot layout of msm.hpp k_msm_slot_sum)
__device__ __forceinline__ void msm_slot_range(const MsmPlan& p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ slot_offs, int w, uint32_t s, uint32_t& begin, uint32_t& end) {
    const uint32_t* so = slot_offs + (size_t)w * p.nb;
    uint32_t lo = 0, hi = p.nb - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (so[mid] <= s) lo = mid; else hi = mid - 1; }
    uint32_t d = lo;
    while (d > 0 && (hist[(size_t)w * p.nb + d] + p.ch - 1) / p.ch + so[d] <= s) --d;
    const uint32_t part = s - so[d];
    const uint32_t cnt = hist[(size_t)w * p.nb + d];
    begin = offs[(size_t)w * p.nb + d] + part * p.ch;
    end = min(offs[(size_t)w * p.nb + d] + cnt, begin + p.ch);
}
// k_msm_slot_sum with the gathered additions on the carry-free form; `ext` from k_msm_extend_q (term t <-> ext[t]); same grid, outputs and slot layout.
// A slot with an exceptional addition is flagged and summed again by k_msm_slot_sum_fix (complete formulas), launched behind this kernel.
template <class F>
__global__ void __launch_bounds__(64, 2) k_msm_slot_sum_q(const QAff<F>* __restrict__ ext, MsmPlan p, const uint32_t* __restrict__ hist,
                                                        const uint32_t* __restrict__ offs, const uint32_t* __restrict__ slot_offs, const uint32_t* __restrict__ slots_per_window,
                                                        const uint32_t* __restrict__ sorted, Jac<F>* __restrict__ slot_sums, uint32_t max_slots, bool hom, uint8_t* __restrict__ flag) {
    __shared__ uint4 park_[7 * 64];
    const int w = blockIdx.y;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots_per_window[w]) return;
    uint32_t begin, end;
    msm_slot_range(p, hist, offs, slot_offs, w, s, begin, end);
    Jac<F> acc = jac_inf<F>();
#if defined(__HIP_DEVICE_COMPILE__)
    bool inf = true, bad = false;
    if constexpr (std::is_same<F, Fp>::value) {
        JacQ a; jq_set_identity(a);
#pragma unroll 1
        for (uint32_t k = begin; k < end; ++k) {
            const QAff<Fp> q = ext[sorted[(size_t)w * p.n + k]];
            if (qfp_zero(q.x) && qfp_zero(q.y)) continue;                                 // the identity among the bases
            const Fqn x = qfp_get(q.x), y = qfp_get(q.y);
            if (inf) { jq_set(a, x, y, fq_one()); inf = false; }
            else bad |= jmadd_q(a, x, y);
        }
        if (!inf && !bad) acc = jacq_to_g1j(a);
    } else {
        uint4* park = park_ + threadIdx.x;
        JacQ2 a; j2_set_identity(a);
#pragma unroll 1
        for (uint32_t k = begin; k < end; ++k) {
            const QAff<Fp2>* qp = ext + sorted[(size_t)w * p.n + k];
            { const QAff<Fp2> q = *qp; if (qfp_zero(q.x0) && qfp_zero(q.x1) && qfp_zero(q.y0) && qfp_zero(q.y1)) continue; }
            auto lx = [&]() { const QAff<Fp2>* o = opaque(qp); return Fq2n{qfp_get(o->x0), qfp_get(o->x1)}; };
            auto ly = [&]() { const QAff<Fp2>* o = opaque(qp); return Fq2n{qfp_get(o->y0), qfp_get(o->y1)}; };
            if (inf) { j2_set(a, lx(), ly(), Fq2n{fq_one(), fq_zero()}); inf = false; }
            else bad |= jmadd2_q(a, lx, ly, park);
        }
        if (!inf && !bad) acc = G2J{f2_to(a.x), f2_to(a.y), f2_to(a.z)};
    }
    flag[(size_t)w * max_slots + s] = bad;
    if (bad) return;
#endif
    slot_sums[(size_t)w * max_slots + s] = hom ? msm_jac_to_h(acc) : acc;
}
template <class F>
__global__ void __launch_bounds__(64) k_msm_slot_sum_fix(const Affine<F>* __restrict__ bases, MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                       const uint32_t* __restrict__ slot_offs, const uint32_t* __restrict__ slots_per_window, const uint32_t* __restrict__ sorted,
                                                       Jac<F>* __restrict__ slot_sums, uint32_t max_slots, bool hom, const uint8_t* __restrict__ flag) {
    for_flagged(flag, (uint32_t)p.nwin * max_slots, [&](uint32_t f) {
        const int w = (int)(f / max_slots); const uint32_t s = f - (uint32_t)w * max_slots;
        if (s >= slots_per_window[w]) return;                                   // (a byte the throughput kernel never wrote)
        uint32_t begin, end;
        msm_slot_range(p, hist, offs, slot_offs, w, s, begin, end);
        Jac<F> acc;
        msm_slot_sum_complete<F>(bases, sorted + (size_t)w * p.n, begin, end, p.nreal, &acc);
        slot_sums[f] = hom ? msm_jac_to_h(acc) : acc; });
}

// The same on the field VM (the form the default path launches: the stages behind the gather work on homogeneous coordinates anyway).  A flagged
// slot is a chain of <= ch complete additions; on ONE lane of a lone wave that chain costs ch x ~35 us (G1) / ~60 us (G2) -- 1.07 / 1.85 ms per MSM
// at n = 2^20 whenever a single slot is flagged (profiles/r03_msm_2p20_kernel_stats_rocprofv3.csv), and the synthetic statements flag a handful on
// every run: their bases are CONSECUTIVE multiples of the generator, so a partial sum (i1 + i2 + ..) G does meet the next base i' G now and then.
// Here one WAVE takes a flagged slot: its four 16-lane groups sum every fourth term with the VM's complete addition (~8 / ~19 us each, affine
// addend as (x : y : 1)), then group 0 adds the other three partial sums: ch / 4 + 3 additions deep.  Waves walk the flag bytes together (64 x 16
// bytes per step) and take the flagged slots of their own chunk one after the other.  nflag (optional): += flagged slots, for RIPP_TRACE.
template <class F>
__global__ void __launch_bounds__(64) k_msm_slot_sum_fix_vm(const Affine<F>* __restrict__ bases, MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                          const uint32_t* __restrict__ slot_offs, const uint32_t* __restrict__ slots_per_window, const uint32_t* __restrict__ sorted,
                                                          Jac<F>* __restrict__ slot_sums, uint32_t max_slots, const uint8_t* __restrict__ flag, uint32_t* __restrict__ nflag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, lg = lane & (VM_G - 1), grp = lane / VM_G;
    VmSlot* const ws0 = reinterpret_cast<VmSlot*>(vm_smem);
    VmSlot* const ws = ws0 + (size_t)grp * C::SLOTS;
    const uint32_t n = (uint32_t)p.nwin * max_slots;
#pragma unroll 1
    for (uint32_t c0 = blockIdx.x * 64u; (uint64_t)c0 * 16 < n; c0 += gridDim.x * 64u) {                  // wave-uniform walk: lane l looks at bytes [16 (c0 + l), 16 (c0 + l) + 16)
        uint4 f = uint4{0, 0, 0, 0};
        if ((uint64_t)(c0 + lane) * 16 < n) f = reinterpret_cast<const uint4*>(flag)[c0 + lane];
        uint64_t todo = __ballot((f.x | f.y | f.z | f.w) != 0);
#pragma unroll 1
        while (todo) {
            const int src = __ffsll((long long)todo) - 1; todo &= todo - 1;
            const uint32_t fw[4] = {(uint32_t)__shfl((int)f.x, src), (uint32_t)__shfl((int)f.y, src), (uint32_t)__shfl((int)f.z, src), (uint32_t)__shfl((int)f.w, src)};
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                const uint32_t idx = (c0 + (uint32_t)src) * 16 + (uint32_t)k;
                if (idx >= n || !((fw[k >> 2] >> (8 * (k & 3))) & 0xFFu)) continue;                             // uniform
                const int w = (int)(idx / max_slots); const uint32_t s = idx - (uint32_t)w * max_slots;
                if (s >= slots_per_window[w]) continue;                                                        // (a byte the throughput kernel never wrote)
                uint32_t begin, end;
                msm_slot_range(p, hist, offs, slot_offs, w, s, begin, end);
                const uint32_t* sw = sorted + (size_t)w * p.n;
                if (lg == 0) { vm_zero(ws); vm_put_t<F>(ws, msm_id_h<F>()); }
#pragma unroll 1
                for (uint32_t t = begin; t < end; t += VM_EPW) {                                               // group g: terms begin + g, begin + g + 4, ..
                    if (lg == 0) {
                        Jac<F> q = msm_id_h<F>();
                        if (t + grp < end) { const Affine<F> b = msm_term_base(bases, sw[t + grp], p.nreal); if (!is_inf(b)) q = Jac<F>{b.x, b.y, F::one()}; }
                        vm_put_q<F>(ws, q);
                    }
                    C::add_(ws, lg);
                }
#pragma unroll 1
                for (int g = 1; g < VM_EPW; ++g) {                                                            // group 0 += the partial sum of group g; the others add the identity (their sums are still to be read)
                    if (lg == 0) vm_put_q<F>(ws, grp == 0 ? vm_get_t<F>(ws0 + (size_t)g * C::SLOTS) : msm_id_h<F>());
                    C::add_(ws, lg);
                }
                if (lane == 0) { slot_sums[idx] = vm_get_t<F>(ws0); if (nflag) atomicAdd(nflag, 1u); }
            }
        }
    }
}

}  // namespace ripp
