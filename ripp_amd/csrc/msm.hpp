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
//   k_msm_slot_group  leaders of 8 slots add them, hierarchically, for buckets with more than 16 slots (short top window, skew)
//   k_msm_bucket_merge / k_msm_vm_merge   per bucket: sum of its slot (group) sums
//   k_msm_vm_segments per segment of 4 buckets: sum_d d*B_d of the segment
//   k_msm_vm_reduce   per 4 segment sums (repeated until one per window remains)
//   k_msm_finish_vm   one workgroup: the Horner recurrence over the windows (nbits doublings + nwin additions, ONE dependent chain)
// Scalars are split on the device before the sort (GLV on G1, GLS on G2: see MsmPlan), so nbits is 128 / 64 instead of 255.
// Every stage after the scatter is a chain of dependent group operations per lane (~60-170 us each for a lone wave), so the chain
// lengths -- not the operation count -- set the time for n <= 2^16: slots of 4..32 terms (as many lanes as fill the chip), and
// everything from the bucket merge on runs on the lane-parallel field VM (vm.hpp: 16 lanes per point, complete projective addition,
// ~8-19 us per operation) in homogeneous coordinates.  RIPP_NO_VM=1 selects the single-lane Jacobian forms of the same stages
// (k_msm_segments, k_msm_seg_reduce, k_msm_finish), kept for A/B and covered by the switch-parametrised parity test.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "bls12_381/curve.hpp"
#include "vm.hpp"

namespace ripp {

constexpr int MSM_SEG_FAN = 4;       // segment sums added per lane in k_msm_seg_reduce / in phase 1 of the finish (short chains: every level is latency)

// n: terms the pipeline sorts and adds; nreal: bases in HBM.  n == 2 * nreal is the GLV form of a G1 MSM: scalar k = k1 + lambda k2 with
// k1, k2 < 2^128 (lambda = z^2 - 1, the eigenvalue of phi(x, y) = (beta x, y)), term i carries k1 on base i and term nreal + i carries k2
// on phi(base i), which the slot sums form with one field product while they gather.  n == 4 * nreal is the GLS form of a G2 MSM:
// k = d0 + d1 u + d2 u^2 + d3 u^3 in base u = |x| (64 bits), term j * nreal + i carries d_j on [u^j] base i = +-psi^j(base i)
// (kernels.hpp::gls_image, two Fp2 products).  Same number of gathered additions, 1/2 resp. 1/4 of the windows and Horner doublings.
struct MsmPlan { int c, nwin; uint32_t nb; uint32_t n; uint32_t ch, seg; uint32_t nreal, gmin; };   // ch: max terms per slot, seg: buckets per segment lane

struct MsmTune { int c = 0; uint32_t ch = 0, gmin = 0; };      // 0 = the plan's own choice (RIPP_MSM_C / RIPP_MSM_CH / RIPP_MSM_GMIN, read once per C-ABI call by the engine)
inline MsmPlan msm_plan(size_t nreal, int split = 1, const MsmTune& tune = MsmTune()) {
    const size_t n = (size_t)split * nreal;
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c0 = lg - 6; if (c0 < 4) c0 = 4; if (c0 > 13) c0 = 13;
    // The top window holds only  nbits - (nwin - 1) c  bits, so its few buckets collect 2^deficit times the terms of a regular
    // bucket and their slot chains set the latency: among c0 - 1 .. c0 + 1 take the width with the fullest top window.
    const int nbits = split == 1 ? 255 : split == 2 ? 128 : 64;
    int c = c0, best = 1 << 20;
    for (int cc = c0 - 1; cc <= c0 + 1; ++cc) {
        if (cc < 4 || cc > 13) continue;
        const int nw = (nbits + cc - 1) / cc, deficit = nw * cc - nbits;
        const int score = 2 * deficit + (cc == c0 ? 0 : 1);
        if (score < best) { best = score; c = cc; }
    }
    if (tune.c) c = tune.c;
    MsmPlan p; p.c = c; p.nwin = (nbits + c - 1) / c; p.nb = 1u << c; p.n = (uint32_t)n; p.nreal = (uint32_t)nreal;
    // slot length: the shortest chain that still gives every SIMD two waves (131072 lanes; a lone wave issues at half rate), between
    // 4 and 32 terms; 64 once the launch is throughput bound (measured crossover, tools/msm_sweep.py)
    const size_t adds = n * (size_t)p.nwin;
    p.ch = 4; while (p.ch < 32 && (size_t)p.ch * 131072 < adds) p.ch *= 2;
    if (n > ((size_t)1 << 20)) p.ch = 64;
    if (tune.ch) p.ch = tune.ch;
    p.gmin = tune.gmin ? tune.gmin : 16;   // buckets of <= gmin slots go straight to the merge
    p.seg = 4;                                   // nb >= 16; chain of 2 * seg additions + the (lo - 1) multiple per lane
    return p;
}

__device__ __forceinline__ void msm_emit_digits(const uint32_t* k, int nlimb, uint32_t i, const MsmPlan& p, uint16_t* __restrict__ digits, uint32_t* __restrict__ hist) {
    for (int w = 0; w < p.nwin; ++w) {
        const int bit = w * p.c, limb = bit >> 5, sh = bit & 31;
        uint64_t v = k[limb];
        if (limb + 1 < nlimb) v |= (uint64_t)k[limb + 1] << 32;
        const uint32_t d = (uint32_t)(v >> sh) & (p.nb - 1);
        digits[(size_t)w * p.n + i] = (uint16_t)d;
        if (d && hist) atomicAdd(&hist[(size_t)w * p.nb + d], 1u);      // (hist == nullptr: the large-sort form counts through LDS, k_msm_hist_lds)
    }
}

// k[0..8) (< 2^255) = q * d + rem by Barrett division: mu = floor(2^256 / d) (MM limbs), q^ = floor(k mu / 2^256) >= q - 2, then at most
// two corrections.  d has ML limbs, rem ML + 1; q overwrites k.
template <int ML, int MM> __device__ __forceinline__ void msm_divmod(uint32_t* k, const uint32_t* d, const uint32_t* mu, uint32_t* rem) {
    uint32_t prod[8 + MM];
    for (int t = 0; t < 8 + MM; ++t) prod[t] = 0;
    for (int i = 0; i < 8; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < MM; ++j) { const uint64_t t = (uint64_t)k[i] * mu[j] + prod[i + j] + carry; prod[i + j] = (uint32_t)t; carry = t >> 32; }
        prod[i + MM] = (uint32_t)carry;
    }
    uint32_t q[8];
    for (int t = 0; t < 8; ++t) q[t] = t < MM ? prod[8 + t] : 0u;
    uint32_t qd[ML + 1];                                                             // q^ d mod 2^(32 (ML + 1))
    for (int t = 0; t <= ML; ++t) qd[t] = 0;
    for (int i = 0; i <= ML; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < ML && i + j <= ML; ++j) { const uint64_t t = (uint64_t)q[i] * d[j] + qd[i + j] + carry; qd[i + j] = (uint32_t)t; carry = t >> 32; }
        if (i + ML <= ML) qd[i + ML] += (uint32_t)carry;
    }
    uint64_t br = 0;
    for (int t = 0; t <= ML; ++t) { const uint64_t x = (uint64_t)k[t] - qd[t] - br; rem[t] = (uint32_t)x; br = (x >> 32) & 1u; }
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        uint32_t df[ML + 1]; br = 0;
        for (int t = 0; t <= ML; ++t) { const uint64_t x = (uint64_t)rem[t] - (t < ML ? d[t] : 0u) - br; df[t] = (uint32_t)x; br = (x >> 32) & 1u; }
        if (br) break;
        for (int t = 0; t <= ML; ++t) rem[t] = df[t];
        uint32_t c = 1; for (int t = 0; t < 8; ++t) { const uint32_t v = q[t] + c; c = (v < c) ? 1u : 0u; q[t] = v; }
    }
    for (int t = 0; t < 8; ++t) k[t] = q[t];
}

__global__ void __launch_bounds__(256) k_msm_digits(const Fr* __restrict__ scalars, MsmPlan p, uint16_t* __restrict__ digits, uint32_t* __restrict__ hist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nreal) return;
    Fr k = from_mont(scalars[i]);
    if (p.n == p.nreal) { msm_emit_digits(k.l, 8, i, p, digits, hist); return; }
    if (p.n == 2 * p.nreal) {                                                        // GLV (G1): k = q * lambda + rem, both < 2^128
        const uint32_t lam[8] = RIPP_GLV_LAMBDA;
        const uint32_t lam_mu[5] = RIPP_GLV_LAMBDA_MU;                  // floor(2^256 / lambda)
        uint32_t rem[5];
        msm_divmod<4, 5>(k.l, lam, lam_mu, rem);
        msm_emit_digits(rem, 5, i, p, digits, hist);
        msm_emit_digits(k.l, 8, p.nreal + i, p, digits, hist);
        return;
    }
    const uint32_t u[2] = RIPP_X_ABS_LIMBS;                                          // GLS (G2): base-|x| digits
    const uint32_t u_mu[7] = RIPP_X_ABS_MU;                                          // floor(2^256 / |x|)
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
        uint32_t rem[3];
        msm_divmod<2, 7>(k.l, u, u_mu, rem);
        msm_emit_digits(rem, 3, j * p.nreal + i, p, digits, hist);
    }
    msm_emit_digits(k.l, 8, 3 * p.nreal + i, p, digits, hist);                       // k < r < u^4: the last quotient is the top digit
}

// one block (1024 lanes) per window; nb <= 8192 -> <= 8 counters per lane
__global__ void __launch_bounds__(1024) k_msm_scan(const uint32_t* __restrict__ hist, MsmPlan p, uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor,
                                                    uint32_t* __restrict__ slot_offs, uint32_t* __restrict__ slots_per_window) {
    __shared__ uint32_t sh_a[1024], sh_b[1024];
    const int w = blockIdx.x, t = threadIdx.x;
    const uint32_t per = (p.nb + 1023) / 1024;
    uint32_t sum = 0, ssum = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t d = t * per + k; if (d < p.nb) { const uint32_t c = hist[(size_t)w * p.nb + d]; sum += c; ssum += (c + p.ch - 1) / p.ch; } }
    sh_a[t] = sum; sh_b[t] = ssum; __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        uint32_t a = 0, b = 0; if (t >= off) { a = sh_a[t - off]; b = sh_b[t - off]; }
        __syncthreads(); sh_a[t] += a; sh_b[t] += b; __syncthreads();
    }
    uint32_t run = sh_a[t] - sum, srun = sh_b[t] - ssum;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t d = t * per + k;
        if (d < p.nb) { const uint32_t c = hist[(size_t)w * p.nb + d]; offs[(size_t)w * p.nb + d] = run; cursor[(size_t)w * p.nb + d] = run; slot_offs[(size_t)w * p.nb + d] = srun; run += c; srun += (c + p.ch - 1) / p.ch; }
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

// The default sort (RIPP_MSM_LDS_SORT_MIN selects the lane-per-term form above below a size, for A/B): histogram and scatter through LDS, one block per (tile of terms, window).  The lane-per-term forms above issue one
// device-scope atomic and one lone 4-byte store per term and window -- 21 M of each at n = 2^20, and the counters show every one of them leaving
// the chip as its own 32 / 64-byte HBM write (0.71 + 1.25 GB per MSM against 42 + 84 MB of digits and indices: profiles/r04_msm_2p20_hbm_traffic_pmc.csv).
// Here a block counts its tile's digits in LDS (nb <= 8192 counters), adds each non-empty count to the window's histogram / reserves a run of that
// length in the bucket with ONE global atomic, and ranks the tile's terms inside their runs with LDS atomics: ~5 consecutive indices per bucket and
// tile, written by one workgroup within microseconds, leave the L2 as combined lines.  The order inside a bucket is as arbitrary as before (sums).
constexpr uint32_t MSM_SORT_BLOCK = 1024;
inline uint32_t msm_sort_tile(const MsmPlan& p) {           // ~480 blocks per launch: two 1024-lane blocks per CU, no second wave of blocks
    const uint32_t tiles = p.nwin >= 480 ? 1u : 480u / (uint32_t)p.nwin;
    const uint32_t t = (p.n + tiles - 1) / tiles;
    return (t + MSM_SORT_BLOCK - 1) / MSM_SORT_BLOCK * MSM_SORT_BLOCK;
}
__global__ void __launch_bounds__(1024) k_msm_hist_lds(const uint16_t* __restrict__ digits, MsmPlan p, uint32_t tile, uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt[8192];
    const uint32_t w = blockIdx.y, lo = blockIdx.x * tile, hi = min(lo + tile, p.n);
    for (uint32_t d = threadIdx.x; d < p.nb; d += blockDim.x) cnt[d] = 0;
    __syncthreads();
    const uint16_t* dg = digits + (size_t)w * p.n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { const uint32_t d = dg[i]; if (d) atomicAdd(&cnt[d], 1u); }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < p.nb; d += blockDim.x) { const uint32_t c = cnt[d]; if (c) atomicAdd(&hist[(size_t)w * p.nb + d], c); }
}
__global__ void __launch_bounds__(1024) k_msm_scatter_lds(const uint16_t* __restrict__ digits, MsmPlan p, uint32_t tile, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cnt[8192];                              // the tile's count per bucket, then the next free position of the tile's run in it
    const uint32_t w = blockIdx.y, lo = blockIdx.x * tile, hi = min(lo + tile, p.n);
    for (uint32_t d = threadIdx.x; d < p.nb; d += blockDim.x) cnt[d] = 0;
    __syncthreads();
    const uint16_t* dg = digits + (size_t)w * p.n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { const uint32_t d = dg[i]; if (d) atomicAdd(&cnt[d], 1u); }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < p.nb; d += blockDim.x) { const uint32_t c = cnt[d]; if (c) cnt[d] = atomicAdd(&cursor[(size_t)w * p.nb + d], c); }
    __syncthreads();
    uint32_t* out = sorted + (size_t)w * p.n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { const uint32_t d = dg[i]; if (d) out[atomicAdd(&cnt[d], 1u)] = i; }
}

// ---- homogeneous projective coordinates (x = X/Z, y = Y/Z; identity (0 : 1 : 0)) ---------------------------------------------------
// Everything after the gathered mixed additions is kept in this form so that the latency-bound stages can use the field VM's complete
// addition (vm.hpp, 16 lanes per point, ~8x shorter latency than one lane); add_h is the one-lane twin of that program
// (Renes-Costello-Batina 2015, Alg. 7, a = 0: 12 products, no exceptional case), used where a stage has lanes to spare instead.
#if defined(RIPP_BLS12_377)
__device__ __forceinline__ Fp msm_mul_b3(const Fp& t) { return add(dbl(t), t); }                                                      // 3b = 3
// 3b' = 3 / u: (c0 + c1 u) / u = c1 + (c0 / u^2) u = (c1, c0 * (-1/5)) -- RIPP_FP_TWIST_B1 is -1/5
__device__ __forceinline__ Fp2 msm_mul_b3(const Fp2& t) { const Fp2 x = {t.c1, fmul(t.c0, fp_const(RIPP_FP_TWIST_B1))}; return add(dbl(x), x); }
#else
__device__ __forceinline__ Fp msm_mul_b3(const Fp& t) { const Fp t4 = dbl(dbl(t)); return add(dbl(t4), t4); }                     // 3b = 12
__device__ __forceinline__ Fp2 msm_mul_b3(const Fp2& t) { const Fp2 x = mul_xi(t), x4 = dbl(dbl(x)); return add(dbl(x4), x4); }     // 3b' = 12 (1 + u)
#endif
template <class F> __device__ __forceinline__ Jac<F> msm_id_h() { return {F::zero(), F::one(), F::zero()}; }
template <class F> __device__ __forceinline__ Jac<F> msm_jac_to_h(const Jac<F>& a) {                                                // (X/Z^2, Y/Z^3) -> (X Z : Y : Z^3)
    if (a.z.is_zero()) return msm_id_h<F>();
    return {fmul(a.x, a.z), a.y, fmul(fsqr(a.z), a.z)};
}
template <class F> __device__ __noinline__ Jac<F> add_h(const Jac<F>& p, const Jac<F>& q) {
    F t0 = fmul(p.x, q.x), t1 = fmul(p.y, q.y), t2 = fmul(p.z, q.z);
    const F t3 = sub(sub(fmul(add(p.x, p.y), add(q.x, q.y)), t0), t1);
    const F t4 = sub(sub(fmul(add(p.y, p.z), add(q.y, q.z)), t1), t2);
    F y3 = sub(sub(fmul(add(p.x, p.z), add(q.x, q.z)), t0), t2);
    t0 = add(dbl(t0), t0);
    t2 = msm_mul_b3(t2);
    const F z3 = add(t1, t2); t1 = sub(t1, t2);
    y3 = msm_mul_b3(y3);
    Jac<F> r;
    r.x = sub(fmul(t3, t1), fmul(t4, y3));
    r.y = add(fmul(t1, z3), fmul(y3, t0));
    r.z = add(fmul(z3, t4), fmul(t0, t3));
    return r;
}

// term -> base: terms >= nreal are the phi images of the GLV form (G1 only)
__device__ __forceinline__ G1A msm_term_base(const G1A* __restrict__ bases, uint32_t t, uint32_t nreal) {
    if (t < nreal) return bases[t];
    G1A q = bases[t - nreal]; q.x = fmul(q.x, fp_const(RIPP_GLV_BETA)); return q;      // (0,0) stays the identity
}
__device__ __forceinline__ G2A msm_term_base(const G2A* __restrict__ bases, uint32_t t, uint32_t nreal) {
    if (t < nreal) return bases[t];
    const uint32_t j = t / nreal;
    return gls_image(bases[t - j * nreal], (int)j);
}

// the textbook loop of one slot, out of line: the complete-formula fallback of the low-liveness G2 form must not shape its register allocation
template <class F>
__device__ __noinline__ void msm_slot_sum_complete(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ sorted_w, uint32_t begin, uint32_t end, uint32_t nreal, Jac<F>* out) {
    Jac<F> acc = jac_inf<F>();
#pragma unroll 1
    for (uint32_t k = begin; k < end; ++k) acc = add_mixed(acc, msm_term_base(bases, sorted_w[k], nreal));
    *out = acc;
}

// grid.y = window; lane = slot index within the window (max_slots lanes per window, surplus lanes exit)
template <class F>
__global__ void __launch_bounds__(64, 2) k_msm_slot_sum(const Affine<F>* __restrict__ bases, MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                      const uint32_t* __restrict__ slot_offs, const uint32_t* __restrict__ slots_per_window,
                                                      const uint32_t* __restrict__ sorted, Jac<F>* __restrict__ slot_sums, uint32_t max_slots, bool hom) {
    const int w = blockIdx.y;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots_per_window[w]) return;
    // bucket owning slot s: the last d with slot_offs[d] <= s among buckets that have slots (binary search; empty buckets share offsets)
    const uint32_t* so = slot_offs + (size_t)w * p.nb;
    uint32_t lo = 0, hi = p.nb - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (so[mid] <= s) lo = mid; else hi = mid - 1; }
    // lo may sit on an empty bucket that shares its offset with the owning one: walk down to the bucket that really has the slot
    uint32_t d = lo;
    while (d > 0 && (hist[(size_t)w * p.nb + d] + p.ch - 1) / p.ch + so[d] <= s) --d;
    const uint32_t part = s - so[d];
    const uint32_t cnt = hist[(size_t)w * p.nb + d];
    const uint32_t begin = offs[(size_t)w * p.nb + d] + part * p.ch;
    const uint32_t end = min(offs[(size_t)w * p.nb + d] + cnt, begin + p.ch);
    Jac<F> acc = jac_inf<F>();
    if constexpr (std::is_same<F, Fp2>::value) {
        // G2: the textbook mixed addition keeps ~9 Fp2 temporaries beside the accumulator and spilled 78 dwords per lane (7.8 GB of scratch writes per
        // launch at n = 2^20).  The low-liveness form of the fold kernels (kernels.hpp jmadd_lo: coordinates of the addend fetched where they are
        // used -- here: the psi image formed from a re-loaded base --, Y1 parked in LDS) needs no scratch; it REPORTS the exceptional cases and such a
        // lane redoes its slot with the complete formulas.
        __shared__ uint4 park_[6 * 64];
        uint4* park = park_ + threadIdx.x;
        Fp2 X = Fp2::one(), Y = Fp2::one(), Z = Fp2::zero();
        bool inf = true, bad = false;
#pragma unroll 1
        for (uint32_t k = begin; k < end; ++k) {
            const uint32_t t = sorted[(size_t)w * p.n + k];
            const int jj = t < p.nreal ? 0 : (int)(t / p.nreal);
            const G2A* bp = bases + (t - (uint32_t)jj * p.nreal);
            auto lx = [&]() { return gls_image_x(opaque(bp)->x, jj); };
            auto ly = [&]() { return gls_image_y(opaque(bp)->y, jj); };
            if (inf) { X = lx(); Y = ly(); Z = Fp2::one(); inf = false; bad |= X.is_zero() && Y.is_zero(); }
            else bad |= jmadd_lo(X, Y, Z, lx, ly, false, park);
        }
        if (bad) msm_slot_sum_complete<F>(bases, sorted + (size_t)w * p.n, begin, end, p.nreal, &acc);
        else if (!inf) acc = Jac<F>{X, Y, Z};
    } else {
#pragma unroll 1
        for (uint32_t k = begin; k < end; ++k) acc = add_mixed(acc, msm_term_base(bases, sorted[(size_t)w * p.n + k], p.nreal));
    }
    slot_sums[(size_t)w * max_slots + s] = hom ? msm_jac_to_h(acc) : acc;
}

// Buckets that collect many slots (the short TOP window puts n / 2^(255 mod c) terms into each of its few buckets; skewed scalar sets
// do the same anywhere) would serialise the merge.  The slot sums of such a bucket are first added in groups of MSM_SLOT_GROUP, in
// place and hierarchically: pass j (stride 8^j) lets the leader of every 8 * 8^j slots add the 8 partial sums below it, for buckets
// with more than 8 * 8^j slots.  After P passes the per-bucket merge walks at most max(8, ns / 8^P) partial sums.
constexpr uint32_t MSM_SLOT_GROUP = 8;
constexpr int MSM_GROUP_PASSES = 3;
template <class F>
__global__ void __launch_bounds__(64) k_msm_slot_group(MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ slot_offs,
                                                        const uint32_t* __restrict__ slots_per_window, Jac<F>* __restrict__ slot_sums, uint32_t max_slots, uint32_t stride, bool hom) {
    const int w = blockIdx.y;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;                     // slot index in the window; leaders are relative to their BUCKET's first slot
    if (s >= slots_per_window[w]) return;
    const uint32_t* so = slot_offs + (size_t)w * p.nb;
    uint32_t lo = 0, hi = p.nb - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (so[mid] <= s) lo = mid; else hi = mid - 1; }
    uint32_t d = lo;
    while (d > 0 && (hist[(size_t)w * p.nb + d] + p.ch - 1) / p.ch + so[d] <= s) --d;
    const uint32_t ns = (hist[(size_t)w * p.nb + d] + p.ch - 1) / p.ch, k0 = s - so[d];
    if (ns <= p.gmin * stride || (k0 % (MSM_SLOT_GROUP * stride)) != 0) return;
    Jac<F>* base = slot_sums + (size_t)w * max_slots + so[d];
    Jac<F> acc = base[k0];
#pragma unroll 1
    for (uint32_t k = k0 + stride; k < k0 + MSM_SLOT_GROUP * stride && k < ns; k += stride) acc = hom ? add_h(acc, base[k]) : add(acc, base[k]);
    base[k0] = acc;
}

// lane per (window, bucket): bucket = sum of its slots (of its group sums when the bucket went through `passes` grouping passes)
template <class F>
__global__ void __launch_bounds__(64) k_msm_bucket_merge(MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ slot_offs,
                                                          const Jac<F>* __restrict__ slot_sums, uint32_t max_slots, Jac<F>* __restrict__ buckets, uint32_t passes, bool hom) {
    const int w = blockIdx.y;
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= p.nb) return;
    const uint32_t cnt = hist[(size_t)w * p.nb + d], ns = (cnt + p.ch - 1) / p.ch, s0 = slot_offs[(size_t)w * p.nb + d];
    uint32_t step = 1;
    for (uint32_t j = 0; j < passes && ns > p.gmin * step; ++j) step *= MSM_SLOT_GROUP;
    Jac<F> acc = hom ? msm_id_h<F>() : jac_inf<F>();
#pragma unroll 1
    for (uint32_t k = 0; k < ns; k += step) { const Jac<F> t = slot_sums[(size_t)w * max_slots + s0 + k]; acc = (k == 0) ? t : hom ? add_h(acc, t) : add(acc, t); }
    buckets[(size_t)w * p.nb + d] = acc;
}

// lane per (window, segment of p.seg buckets): seg = sum_{d in segment} d * B_d
template <class F>
__global__ void __launch_bounds__(64) k_msm_segments(MsmPlan p, const Jac<F>* __restrict__ buckets, Jac<F>* __restrict__ seg_out, uint32_t nseg) {
    const int w = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nseg) return;
    const uint32_t lo = j * p.seg, hi = min(lo + p.seg, p.nb);      // buckets [lo, hi); bucket 0 is always empty
    Jac<F> run = jac_inf<F>(), acc = jac_inf<F>();
#pragma unroll 1
    for (uint32_t d = hi; d-- > lo;) { run = add(run, buckets[(size_t)w * p.nb + d]); acc = add(acc, run); }
    // acc = sum (d - lo + 1) B_d ; add (lo - 1) * run   (for lo == 0: subtract run)
    if (lo == 0) { acc = add(acc, neg(run)); }
    else if (lo > 1) {
        const uint32_t m = lo - 1; Jac<F> t = run;
#pragma unroll 1
        for (int b = 30 - __clz(m); b >= 0; --b) { t = dbl(t); if ((m >> b) & 1u) t = add(t, run); }
        acc = add(acc, t);
    }
    seg_out[(size_t)w * nseg + j] = acc;
}

// lane per (window, group of MSM_SEG_FAN segment sums)
template <class F>
__global__ void __launch_bounds__(64) k_msm_seg_reduce(const Jac<F>* __restrict__ in, uint32_t nin, Jac<F>* __restrict__ out, uint32_t nout) {
    const int w = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nout) return;
    const uint32_t lo = j * MSM_SEG_FAN, hi = min(lo + MSM_SEG_FAN, nin);
    Jac<F> acc = in[(size_t)w * nin + lo];
#pragma unroll 1
    for (uint32_t k = lo + 1; k < hi; ++k) acc = add(acc, in[(size_t)w * nin + k]);
    out[(size_t)w * nout + j] = acc;
}

// ---- Horner over the windows on the field VM ------------------------------------------------------------------------
template <class F> struct VmCurve;
template <> struct VmCurve<Fp> {
    static constexpr int NF = 1;
    static constexpr int SLOTS = (vmprog::g1_hdbl_g16_nslots > vmprog::g1_cadd_g16_nslots) ? vmprog::g1_hdbl_g16_nslots : vmprog::g1_cadd_g16_nslots;
    static constexpr int SX = vmprog::g1_cadd_g16_in_X0, SY = vmprog::g1_cadd_g16_in_Y0, SZ = vmprog::g1_cadd_g16_in_Z0;
    static constexpr int QX = vmprog::g1_cadd_g16_in_qx0, QY = vmprog::g1_cadd_g16_in_qy0, QZ = vmprog::g1_cadd_g16_in_qz0;
    static_assert(vmprog::g1_hdbl_g16_in_X0 == SX && vmprog::g1_hdbl_g16_in_Y0 == SY && vmprog::g1_hdbl_g16_in_Z0 == SZ, "accumulator slots");
    static_assert(vmprog::g1_cadd_g16_out_X0 == SX && vmprog::g1_cadd_g16_out_Y0 == SY && vmprog::g1_cadd_g16_out_Z0 == SZ, "accumulator slots");
    __device__ static void dbl_(VmSlot* ws, int lg) { vm_run(ws, vmprog::g1_hdbl_g16_kind, vmprog::g1_hdbl_g16_ops, vmprog::g1_hdbl_g16_nlayers, lg); }
    __device__ static void add_(VmSlot* ws, int lg) { vm_run(ws, vmprog::g1_cadd_g16_kind, vmprog::g1_cadd_g16_ops, vmprog::g1_cadd_g16_nlayers, lg); }
    __device__ static void put(VmSlot* ws, int slot, const Fp& v) { vm_put(ws, slot, v); }
    __device__ static Fp get(const VmSlot* ws, int slot) { return vm_get(ws, slot); }
};
template <> struct VmCurve<Fp2> {
    static constexpr int NF = 2;
    static constexpr int SLOTS = (vmprog::g2_hdbl_g16_nslots > vmprog::g2_cadd_g16_nslots) ? vmprog::g2_hdbl_g16_nslots : vmprog::g2_cadd_g16_nslots;
    static constexpr int SX = vmprog::g2_cadd_g16_in_X0, SY = vmprog::g2_cadd_g16_in_Y0, SZ = vmprog::g2_cadd_g16_in_Z0;
    static constexpr int QX = vmprog::g2_cadd_g16_in_qx0, QY = vmprog::g2_cadd_g16_in_qy0, QZ = vmprog::g2_cadd_g16_in_qz0;
    static_assert(vmprog::g2_hdbl_g16_in_X0 == SX && vmprog::g2_hdbl_g16_in_Y0 == SY && vmprog::g2_hdbl_g16_in_Z0 == SZ, "accumulator slots");
    static_assert(vmprog::g2_cadd_g16_out_X0 == SX && vmprog::g2_cadd_g16_out_Y0 == SY && vmprog::g2_cadd_g16_out_Z0 == SZ, "accumulator slots");
    static_assert(vmprog::g2_cadd_g16_in_X1 == SX + 1 && vmprog::g2_cadd_g16_in_qz1 == QZ + 1 && vmprog::g2_hdbl_g16_in_Z1 == SZ + 1, "c0/c1 adjacent");
    __device__ static void dbl_(VmSlot* ws, int lg) { vm_run(ws, vmprog::g2_hdbl_g16_kind, vmprog::g2_hdbl_g16_ops, vmprog::g2_hdbl_g16_nlayers, lg); }
    __device__ static void add_(VmSlot* ws, int lg) { vm_run(ws, vmprog::g2_cadd_g16_kind, vmprog::g2_cadd_g16_ops, vmprog::g2_cadd_g16_nlayers, lg); }
    __device__ static void put(VmSlot* ws, int slot, const Fp2& v) { vm_put(ws, slot, v.c0); vm_put(ws, slot + 1, v.c1); }
    __device__ static Fp2 get(const VmSlot* ws, int slot) { return {vm_get(ws, slot), vm_get(ws, slot + 1)}; }
};

// one block of 64 lanes: seg[w] is the homogeneous sum of window w (k_msm_vm_reduce ran down to one per window); lanes 0..15 run
//     T <- W_top;  for w = top-1 .. 0:  T <- 2^c T (c VM doublings);  T <- T + W_w (complete VM addition)
// and lane 0 converts T to Jacobian.  The complete addition law has no exceptional case, so no fallback is needed.
template <class F>
__global__ void __launch_bounds__(64) k_msm_finish_vm(MsmPlan p, const Jac<F>* __restrict__ seg, Jac<F>* __restrict__ win_h, Jac<F>* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    VmSlot* const lds = reinterpret_cast<VmSlot*>(vm_smem);
    const int lane = threadIdx.x, lg = lane & (VM_G - 1), grp = lane / VM_G;
    VmSlot* const ws = lds + (size_t)grp * C::SLOTS;
    if ((uint32_t)lane < (uint32_t)p.nwin) win_h[lane] = seg[lane];
    __syncthreads();
    const bool lead = (lane == 0);
    if (lg == 0) vm_zero(ws);
    if (lead) { const Jac<F> t = win_h[p.nwin - 1]; C::put(ws, C::SX, t.x); C::put(ws, C::SY, t.y); C::put(ws, C::SZ, t.z); }
    else if (lg == 0) { C::put(ws, C::SX, F::zero()); C::put(ws, C::SY, F::one()); C::put(ws, C::SZ, F::zero()); }   // idle groups: a valid point
#pragma unroll 1
    for (int w = p.nwin - 2; w >= 0; --w) {
#pragma unroll 1
        for (int k = 0; k < p.c; ++k) C::dbl_(ws, lg);
        if (lg == 0) { const Jac<F> t = win_h[w]; C::put(ws, C::QX, t.x); C::put(ws, C::QY, t.y); C::put(ws, C::QZ, t.z); }
        C::add_(ws, lg);
    }
    if (lead) {
        const F X = C::get(ws, C::SX), Y = C::get(ws, C::SY), Z = C::get(ws, C::SZ);
        Jac<F> r = jac_inf<F>();
        if (!Z.is_zero()) { r.x = fmul(X, Z); r.y = fmul(Y, fsqr(Z)); r.z = Z; }
        out[0] = r;
    }
}

// ---- the latency-bound middle of the pipeline on the field VM ---------------------------------------------------------------------
// One point per group of 16 lanes (4 per wave, 16 per block), homogeneous coordinates in and out.  A complete addition costs ~8 us
// (G1) / ~19 us (G2) here against ~60 / ~170 us for a lone wave's single lane, and these stages have few points: buckets, segments,
// segment sums.  Groups whose chain is shorter keep adding the identity (the addition law is complete), so a wave stays converged.
template <class F> __device__ __forceinline__ void vm_put_t(VmSlot* ws, const Jac<F>& t) { using C = VmCurve<F>; C::put(ws, C::SX, t.x); C::put(ws, C::SY, t.y); C::put(ws, C::SZ, t.z); }
template <class F> __device__ __forceinline__ void vm_put_q(VmSlot* ws, const Jac<F>& t) { using C = VmCurve<F>; C::put(ws, C::QX, t.x); C::put(ws, C::QY, t.y); C::put(ws, C::QZ, t.z); }
template <class F> __device__ __forceinline__ Jac<F> vm_get_t(const VmSlot* ws) { using C = VmCurve<F>; return {C::get(ws, C::SX), C::get(ws, C::SY), C::get(ws, C::SZ)}; }

// group per (window, bucket): bucket = sum of its slot sums (of its group sums after `passes` grouping passes)
template <class F>
__global__ void __launch_bounds__(256) k_msm_vm_merge(MsmPlan p, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ slot_offs,
                                                       const Jac<F>* __restrict__ slot_sums, uint32_t max_slots, Jac<F>* __restrict__ buckets, uint32_t passes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G, w = blockIdx.y;
    const uint32_t d = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = reinterpret_cast<VmSlot*>(vm_smem) + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    uint32_t ns = 0, s0 = 0, step = 1;
    if (d < p.nb) {
        ns = (hist[(size_t)w * p.nb + d] + p.ch - 1) / p.ch; s0 = slot_offs[(size_t)w * p.nb + d];
        for (uint32_t j = 0; j < passes && ns > p.gmin * step; ++j) step *= MSM_SLOT_GROUP;
    }
    const Jac<F>* src = slot_sums + (size_t)w * max_slots + s0;
    if (lg == 0) { vm_zero(ws); vm_put_t<F>(ws, ns ? src[0] : msm_id_h<F>()); }
    uint32_t k = step;
#pragma unroll 1
    while (__any(k < ns)) {
        if (lg == 0) vm_put_q<F>(ws, k < ns ? src[k] : msm_id_h<F>());
        C::add_(ws, lg);
        k += step;
    }
    if (d < p.nb && lg == 0) buckets[(size_t)w * p.nb + d] = vm_get_t<F>(ws);
}

// group per (window, segment of 4 buckets lo .. lo + 3, lo = 4 j):  sum_r (lo + r) B_r  =  4 j S + (B1 + 2 B2 + 3 B3),  S = B0 + B1 + B2 + B3.
//   U = B2 + B3;  local = 2 U + B1 + B3;  S = U + B1 + B0;  T = j S by double-and-add over the bits of j (every group walks the same
//   number of bits and adds S or the identity);  out = 4 T + local.
template <class F>
__global__ void __launch_bounds__(256) k_msm_vm_segments(MsmPlan p, const Jac<F>* __restrict__ buckets, Jac<F>* __restrict__ seg_out, uint32_t nseg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G, w = blockIdx.y;
    const uint32_t j = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = reinterpret_cast<VmSlot*>(vm_smem) + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    const bool active = j < nseg, lead = lg == 0;
    const Jac<F>* b = buckets + (size_t)w * p.nb + (size_t)j * 4;
    Jac<F> B0 = msm_id_h<F>(), B1 = B0, B3 = B0, S = B0, local = B0;
    if (lead) {
        vm_zero(ws);
        if (active) { B0 = b[0]; B1 = b[1]; B3 = b[3]; }
        vm_put_t<F>(ws, active ? b[2] : msm_id_h<F>()); vm_put_q<F>(ws, B3);
    }
    C::add_(ws, lg);                                                  // U
    if (lead) S = vm_get_t<F>(ws);                                    // S holds U for the moment
    C::dbl_(ws, lg);
    if (lead) vm_put_q<F>(ws, B1);                                    // (the doubling program's scratch overlays the addend slots: write Q after it)
    C::add_(ws, lg);                                                  // 2U + B1
    if (lead) vm_put_q<F>(ws, B3);
    C::add_(ws, lg);                                                  // local
    if (lead) { local = vm_get_t<F>(ws); vm_put_t<F>(ws, S); vm_put_q<F>(ws, B1); }
    C::add_(ws, lg);
    if (lead) vm_put_q<F>(ws, B0);
    C::add_(ws, lg);                                                  // S
    if (lead) { S = vm_get_t<F>(ws); vm_put_t<F>(ws, msm_id_h<F>()); }
    int top = 31 - __clz((int)(nseg > 1 ? nseg - 1 : 1));
#pragma unroll 1
    for (int bit = top; bit >= 0; --bit) {
        C::dbl_(ws, lg);
        const bool set = active && ((j >> bit) & 1u);
        if (__any(set)) {
            if (lead) vm_put_q<F>(ws, set ? S : msm_id_h<F>());
            C::add_(ws, lg);
        }
    }
    C::dbl_(ws, lg); C::dbl_(ws, lg);
    if (lead) vm_put_q<F>(ws, local);
    C::add_(ws, lg);
    if (active && lead) seg_out[(size_t)w * nseg + j] = vm_get_t<F>(ws);
}

// group per (window, MSM_SEG_FAN consecutive sums)
template <class F>
__global__ void __launch_bounds__(256) k_msm_vm_reduce(const Jac<F>* __restrict__ in, uint32_t nin, Jac<F>* __restrict__ out, uint32_t nout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem[];
    using C = VmCurve<F>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane & (VM_G - 1), grp = lane / VM_G, w = blockIdx.y;
    const uint32_t j = (blockIdx.x * 4 + wave) * VM_EPW + grp;
    VmSlot* const ws = reinterpret_cast<VmSlot*>(vm_smem) + (size_t)(wave * VM_EPW + grp) * C::SLOTS;
    const uint32_t lo = j * MSM_SEG_FAN, hi = j < nout ? min(lo + MSM_SEG_FAN, nin) : lo;
    const Jac<F>* src = in + (size_t)w * nin;
    if (lg == 0) { vm_zero(ws); vm_put_t<F>(ws, lo < hi ? src[lo] : msm_id_h<F>()); }
#pragma unroll 1
    for (uint32_t k = 1; k < (uint32_t)MSM_SEG_FAN; ++k) {
        if (!__any(lo + k < hi)) break;
        if (lg == 0) vm_put_q<F>(ws, lo + k < hi ? src[lo + k] : msm_id_h<F>());
        C::add_(ws, lg);
    }
    if (j < nout && lg == 0) out[(size_t)w * nout + j] = vm_get_t<F>(ws);
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
