// Windowed Pippenger multi-scalar multiplication  sum_i s_i * G_i  over G1 / G2 on gfx950.
// Replaces `VariableBaseMSM::msm` behind MultiexponentiationInnerProduct::inner_product
// (inner_products/src/lib.rs:128-141) and the two MSMs of SIPP::verify (sipp/src/lib.rs:174-175).
// The result is a group element, so it is algorithm independent; parity is checked on the normalised point.
//
// Pipeline (all on the engine's stream):
//   k_msm_digits      lane per term: Montgomery -> canonical scalar, c-bit digits of all windows, histogram (atomics)
//   k_msm_scan        per window: exclusive scan of the 2^c bucket counts + slot counts (LDS block scan)
//   k_msm_scatter     lane per term: counting-sort scatter of the term index into its bucket's run
//   k_msm_slot_sum    lane per SLOT (<= CH terms of one bucket): mixed additions of affine bases gathered from HBM.
//                     Slots bound the work of one lane, so skewed scalar sets (all-equal scalars put every term in
//                     one bucket per window) cannot serialise the launch.
//   k_msm_bucket_merge lane per bucket: sum of its slots
//   k_msm_segments    lane per 64-bucket segment: running-sum reduction  sum_d d*B_d  of the segment
//   k_msm_finish      one lane: segment sums -> window sums -> Horner over windows
#pragma once
#include <hip/hip_runtime.h>
#include "bls12_381/curve.hpp"

namespace ripp {

constexpr int MSM_CH = 256;          // max terms summed by one lane in k_msm_slot_sum
constexpr int MSM_SEG = 64;          // buckets per lane in k_msm_segments

struct MsmPlan { int c, nwin; uint32_t nb; uint32_t n; };

inline MsmPlan msm_plan(size_t n) {
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c = lg - 6; if (c < 4) c = 4; if (c > 13) c = 13;
    MsmPlan p; p.c = c; p.nwin = (255 + c - 1) / c; p.nb = 1u << c; p.n = (uint32_t)n; return p;
}

__global__ void __launch_bounds__(256) k_msm_digits(const Fr* __restrict__ scalars, MsmPlan p, uint16_t* __restrict__ digits, uint32_t* __restrict__ hist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const Fr k = from_mont(scalars[i]);
    for (int w = 0; w < p.nwin; ++w) {
        const int bit = w * p.c, limb = bit >> 5, sh = bit & 31;
        uint64_t v = k.l[limb];
        if (limb + 1 < 8) v |= (uint64_t)k.l[limb + 1] << 32;
        const uint32_t d = (uint32_t)(v >> sh) & (p.nb - 1);
        digits[(size_t)w * p.n + i] = (uint16_t)d;
        if (d) atomicAdd(&hist[(size_t)w * p.nb + d], 1u);
    }
}

// one block (1024 lanes) per window; nb <= 8192 -> <= 8 counters per lane
__global__ void __launch_bounds__(1024) k_msm_scan(const uint32_t* __restrict__ hist, MsmPlan p, uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor,
                                                    uint32_t* __restrict__ slot_offs, uint32_t* __restrict__ slots_per_window) {
    __shared__ uint32_t sh_a[1024], sh_b[1024];
    const int w = blockIdx.x, t = threadIdx.x;
    const uint32_t per = (p.nb + 1023) / 1024;
    uint32_t sum = 0, ssum = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t d = t * per + k; if (d < p.nb) { const uint32_t c = hist[(size_t)w * p.nb + d]; sum += c; ssum += (c + MSM_CH - 1) / MSM_CH; } }
    sh_a[t] = sum; sh_b[t] = ssum; __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        uint32_t a = 0, b = 0; if (t >= off) { a = sh_a[t - off]; b = sh_b[t - off]; }
        __syncthreads(); sh_a[t] += a; sh_b[t] += b; __syncthreads();
    }
    uint32_t run = sh_a[t] - sum, srun = sh_b[t] - ssum;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t d = t * per + k;
        if (d < p.nb) { const uint32_t c = hist[(size_t)w * p.nb + d]; offs[(size_t)w * p.nb + d] = run; cursor[(size_t)w * p.nb + d] = run; slot_offs[(size_t)w * p.nb + d] = srun; run += c; srun += (c + MSM_CH - 1) / MSM_CH; }
    }
    if (t == 1023) slots_per_window[w] = sh_b[1023];
}

__global__ void __launch_bounds__(256) k_msm_scatter(const uint16_t* __restrict__ digits, MsmPlan p, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    for (int w = 0; w < p.nwin; ++w) {
        const uint32_t d = digits[(size_t)w * p.n + i];
        if (d) { const uint32_t pos = atomicAdd(&cursor[(size_t)w * p.nb + d], 1u); sorted[(size_t)w * p.n + pos] = i; }
    }
}

// grid.y = window; lane = slot index within the window (max_slots lanes per window, surplus lanes exit)
template <class F>
__global__ void __launch_bounds__(64) k_msm_slot_sum(const Affine<F>* __restrict__ bases, MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                      const uint32_t* __restrict__ slot_offs, const uint32_t* __restrict__ slots_per_window,
                                                      const uint32_t* __restrict__ sorted, Jac<F>* __restrict__ slot_sums, uint32_t max_slots) {
    const int w = blockIdx.y;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots_per_window[w]) return;
    // bucket owning slot s: the last d with slot_offs[d] <= s among buckets that have slots (binary search; empty buckets share offsets)
    const uint32_t* so = slot_offs + (size_t)w * p.nb;
    uint32_t lo = 0, hi = p.nb - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (so[mid] <= s) lo = mid; else hi = mid - 1; }
    // lo may sit on an empty bucket that shares its offset with the owning one: walk down to the bucket that really has the slot
    uint32_t d = lo;
    while (d > 0 && (hist[(size_t)w * p.nb + d] + MSM_CH - 1) / MSM_CH + so[d] <= s) --d;
    const uint32_t part = s - so[d];
    const uint32_t cnt = hist[(size_t)w * p.nb + d];
    const uint32_t begin = offs[(size_t)w * p.nb + d] + part * MSM_CH;
    const uint32_t end = min(offs[(size_t)w * p.nb + d] + cnt, begin + MSM_CH);
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (uint32_t k = begin; k < end; ++k) acc = add_mixed(acc, bases[sorted[(size_t)w * p.n + k]]);
    slot_sums[(size_t)w * max_slots + s] = acc;
}

// lane per (window, bucket): bucket = sum of its slots
template <class F>
__global__ void __launch_bounds__(64) k_msm_bucket_merge(MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ slot_offs,
                                                          const Jac<F>* __restrict__ slot_sums, uint32_t max_slots, Jac<F>* __restrict__ buckets) {
    const int w = blockIdx.y;
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= p.nb) return;
    const uint32_t cnt = hist[(size_t)w * p.nb + d], ns = (cnt + MSM_CH - 1) / MSM_CH, s0 = slot_offs[(size_t)w * p.nb + d];
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (uint32_t k = 0; k < ns; ++k) { const Jac<F> t = slot_sums[(size_t)w * max_slots + s0 + k]; acc = (k == 0) ? t : add(acc, t); }
    buckets[(size_t)w * p.nb + d] = acc;
}

// lane per (window, segment of MSM_SEG buckets): seg = sum_{d in segment} d * B_d
template <class F>
__global__ void __launch_bounds__(64) k_msm_segments(MsmPlan p, const Jac<F>* __restrict__ buckets, Jac<F>* __restrict__ seg_out, uint32_t nseg) {
    const int w = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nseg) return;
    const uint32_t lo = j * MSM_SEG, hi = min(lo + MSM_SEG, p.nb);      // buckets [lo, hi); bucket 0 is always empty
    Jac<F> run = jac_inf<F>(), acc = jac_inf<F>();
#pragma unroll 1
    for (uint32_t d = hi; d-- > lo;) { run = add(run, buckets[(size_t)w * p.nb + d]); acc = add(acc, run); }
    // acc = sum (d - lo + 1) B_d ; add (lo - 1) * run   (for lo == 0: subtract run)
    if (lo == 0) { acc = add(acc, neg(run)); }
    else {
        const uint32_t m = lo - 1; Jac<F> t = jac_inf<F>();
#pragma unroll 1
        for (int b = 31; b >= 0; --b) { t = dbl(t); if ((m >> b) & 1u) t = add(t, run); }
        acc = add(acc, t);
    }
    seg_out[(size_t)w * nseg + j] = acc;
}

template <class F>
__global__ void __launch_bounds__(64) k_msm_finish(MsmPlan p, const Jac<F>* __restrict__ seg, uint32_t nseg, Jac<F>* __restrict__ win_sums, Jac<F>* __restrict__ out) {
    // phase 1: lane w sums the segments of window w
    const uint32_t w = threadIdx.x;
    if (w < (uint32_t)p.nwin) {
        Jac<F> acc = jac_inf<F>();
#pragma unroll 1
        for (uint32_t j = 0; j < nseg; ++j) acc = add(acc, seg[(size_t)w * nseg + j]);
        win_sums[w] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Jac<F> total = jac_inf<F>();
#pragma unroll 1
        for (int ww = p.nwin - 1; ww >= 0; --ww) {
#pragma unroll 1
            for (int k = 0; k < p.c; ++k) total = dbl(total);
            total = add(total, win_sums[ww]);
        }
        out[0] = total;
    }
}

}  // namespace ripp
