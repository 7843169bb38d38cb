// Fq28 (14 x 28-bit carry-free limbs) against the 12 x 32-bit multiplier: agreement with the host field + throughput (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -I../../ripp_amd/csrc -o build/fqbench fqbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "line_products.hpp"
#include "fq28.hpp"

using namespace ripp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)


__global__ void k_check(const Fp* a, const Fp* b, Fp* prod, Fp* sum, Fp* dif, Fp* dot, Fp* rt, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const Fqn x = fq_from_fp(a[i]), y = fq_from_fp(b[i]);
    prod[i] = fq_to_fp(fq_mul(x, y));
    rt[i] = fq_to_fp(x);
    const auto s = fq_add(x, y);                 // lazy: <29, 4>
    sum[i] = fq_to_fp(fq_mul(s, fq_one()));
    const auto d = fq_sub(x, y);
    dif[i] = fq_to_fp(fq_mul(d, fq_one()));
    // (x+y)(x-y) + x*y + y*y  == x^2 + xy  as a lazy 3-product sum with lazy operands
    const decltype(fq_add(x, y)) A[3] = {s, fq_add(x, fq_zero()), fq_add(y, fq_zero())};
    const auto dn = fq_norm(d);
    const decltype(dn) B[3] = {dn, fq_widen<FQ_LN, 5>(y), fq_widen<FQ_LN, 5>(y)};
    dot[i] = fq_to_fp(fq_dot<3>(A, B));
#endif
}

__global__ void __launch_bounds__(256) k_fq_mul_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn x = fq_unpack(in[tid & 1023].l), y = fq_unpack(in[(tid + 1) & 1023].l);
    for (int i = 0; i < iters; ++i) x = fq_mul(x, y);
    Fp r; fq_pack(x, r.l); out[tid] = r;
#endif
}
__global__ void __launch_bounds__(256, 2) k_fq_dot6_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn x[6], y[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { x[c] = fq_unpack(in[(tid + 7 * c) & 1023].l); y[c] = fq_unpack(in[(tid + 11 * c + 1) & 1023].l); }
    for (int i = 0; i < iters; ++i) { const Fqn r = fq_dot<6>(x, y); x[i % 6 == 0 ? 0 : 1] = r; }
    Fqn acc = x[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) { const auto w = fq_norm(fq_add(acc, x[c])); for (int q = 0; q < 14; ++q) acc.l[q] = w.l[q]; acc.l[13] &= 0xfffff; }
    Fp r; fq_pack(acc, r.l); out[tid] = r;
#endif
}
// group-law shaped mix: 2 lazy adds + 1 lazy sub per product
__global__ void __launch_bounds__(256) k_fq_mix_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn x = fq_unpack(in[tid & 1023].l), y = fq_unpack(in[(tid + 1) & 1023].l), z = fq_unpack(in[(tid + 2) & 1023].l);
    for (int i = 0; i < iters; ++i) { const Fqn t = fq_mul(fq_add(x, y), fq_sub(z, x)); x = y; y = z; z = t; }
    Fp r; fq_pack(z, r.l); out[tid] = r;
#endif
}
__global__ void __launch_bounds__(256) k_fp_mix_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023], z = in[(tid + 2) & 1023];
    for (int i = 0; i < iters; ++i) { const Fp t = mul(add(x, y), sub(z, x)); x = y; y = z; z = t; }
    out[tid] = z;
}
__global__ void __launch_bounds__(256) k_fp_mul_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023];
    for (int i = 0; i < iters; ++i) x = mul(x, y);
    out[tid] = x;
}
__global__ void __launch_bounds__(256, 2) k_fp_dot6_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x[6], y[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { x[c] = in[(tid + 7 * c) & 1023]; y[c] = in[(tid + 11 * c + 1) & 1023]; }
    for (int i = 0; i < iters; ++i) { const Fp r = fp_dot<6>(x, y); x[i % 6 == 0 ? 0 : 1] = r; }
    Fp acc = x[0];
#pragma unroll
    for (int c = 1; c < 6; ++c) acc = add(acc, x[c]);
    out[tid] = acc;
}


// ---- experiment: the 12 x 32-bit product with SEVERAL limb products per asm block (hipcc puts one s_nop after every asm statement)
#if defined(__HIP_DEVICE_COMPILE__)
#define M1(a,b) "v_mad_u64_u32 %0, vcc, " a ", " b ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
__device__ __forceinline__ void madc96_2(uint64_t& acc, uint32_t& c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    asm(M1("%2","%3") M1("%4","%5") : "+v"(acc), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
__device__ __forceinline__ void madc96_4(uint64_t& acc, uint32_t& c2, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
    asm(M1("%2","%3") M1("%4","%5") M1("%6","%7") M1("%8","%9") : "+v"(acc), "+v"(c2) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3) : "vcc");
}
// column = list of (x, y) pairs gathered first, then issued in blocks of 4 / 2 / 1
template <int N> struct Col { uint32_t x[N > 0 ? N : 1], y[N > 0 ? N : 1]; };
template <int N> __device__ __forceinline__ void col_issue(uint64_t& acc, uint32_t& c2, const uint32_t* x, const uint32_t* y) {
    int i = 0;
#pragma unroll
    for (; i + 4 <= N; i += 4) madc96_4(acc, c2, x[i], y[i], x[i + 1], y[i + 1], x[i + 2], y[i + 2], x[i + 3], y[i + 3]);
#pragma unroll
    for (; i + 2 <= N; i += 2) madc96_2(acc, c2, x[i], y[i], x[i + 1], y[i + 1]);
#pragma unroll
    for (; i < N; ++i) madc96(acc, c2, x[i], y[i]);
}
template <int K> struct ColStep {
    __device__ __forceinline__ static void lo(uint64_t& acc, uint32_t& c2, const Fp& a, const Fp& b, uint32_t* m) {   // column K < 12
        uint32_t x[2 * K + 1], y[2 * K + 1];
#pragma unroll
        for (int i = 0; i <= K; ++i) { x[i] = a.l[i]; y[i] = b.l[K - i]; }
#pragma unroll
        for (int i = 0; i < K; ++i) { x[K + 1 + i] = m[i]; y[K + 1 + i] = FpParams::mod(K - i); }
        col_issue<2 * K + 1>(acc, c2, x, y);
        m[K] = (uint32_t)acc * FpParams::INV;
        madc96_s(acc, c2, m[K], FpParams::mod(0));
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
    __device__ __forceinline__ static void hi(uint64_t& acc, uint32_t& c2, const Fp& a, const Fp& b, const uint32_t* m, Fp& r) {   // column K >= 12
        constexpr int N = 12, CNT = 2 * N - 1 - K;
        uint32_t x[2 * CNT], y[2 * CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) { x[i] = a.l[K - N + 1 + i]; y[i] = b.l[N - 1 - i]; x[CNT + i] = m[K - N + 1 + i]; y[CNT + i] = FpParams::mod(N - 1 - i); }
        col_issue<2 * CNT>(acc, c2, x, y);
        r.l[K - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32); c2 = 0;
    }
};
template <int K> __device__ __forceinline__ void cols_lo(uint64_t& acc, uint32_t& c2, const Fp& a, const Fp& b, uint32_t* m) { if constexpr (K < 12) { ColStep<K>::lo(acc, c2, a, b, m); cols_lo<K + 1>(acc, c2, a, b, m); } }
template <int K> __device__ __forceinline__ void cols_hi(uint64_t& acc, uint32_t& c2, const Fp& a, const Fp& b, const uint32_t* m, Fp& r) { if constexpr (K < 23) { ColStep<K>::hi(acc, c2, a, b, m, r); cols_hi<K + 1>(acc, c2, a, b, m, r); } }
__device__ __forceinline__ Fp mul_blk(const Fp& a, const Fp& b) {
    uint32_t m[12]; Fp r; uint64_t acc = 0; uint32_t c2 = 0;
    cols_lo<0>(acc, c2, a, b, m);
    cols_hi<12>(acc, c2, a, b, m, r);
    r.l[11] = (uint32_t)acc;
    reduce_once(r);
    return r;
}
#endif
__global__ void __launch_bounds__(256) k_fp_mulblk_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023];
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < iters; ++i) x = mul_blk(x, y);
#endif
    out[tid] = x;
}
__global__ void k_check_blk(const Fp* a, const Fp* b, Fp* prod, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
    prod[i] = mul_blk(a[i], b[i]);
#endif
}


// occupancy-limited chains (dynamic LDS caps the waves per SIMD): what the register-heavy kernels see
__global__ void __launch_bounds__(256) k_fq_mul_chain_occ(const Fp* in, Fp* out, int iters) {
    extern __shared__ uint32_t occ_smem[];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn x = fq_unpack(in[tid & 1023].l), y = fq_unpack(in[(tid + 1) & 1023].l);
    for (int i = 0; i < iters; ++i) x = fq_mul(x, y);
    if (iters < 0) occ_smem[threadIdx.x] = x.l[0];
    Fp r; fq_pack(x, r.l); out[tid] = r;
#endif
}
__global__ void __launch_bounds__(64) k_fq_dot2_chain_occ(const Fp* in, Fp* out, int iters) {
    extern __shared__ uint32_t occ_smem[];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn x[2], y[2];
    x[0] = fq_unpack(in[tid & 1023].l); x[1] = fq_unpack(in[(tid + 3) & 1023].l); y[0] = fq_unpack(in[(tid + 1) & 1023].l); y[1] = fq_unpack(in[(tid + 2) & 1023].l);
    for (int i = 0; i < iters; ++i) { const Fqn r = fq_dot<2>(x, y); x[i & 1] = r; }
    if (iters < 0) occ_smem[threadIdx.x] = x[0].l[0];
    Fp r; fq_pack(x[0], r.l); out[tid] = r;
#endif
}
__global__ void __launch_bounds__(64) k_fp_mul_chain_occ(const Fp* in, Fp* out, int iters) {
    extern __shared__ uint32_t occ_smem[];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023];
    for (int i = 0; i < iters; ++i) x = mul(x, y);
    if (iters < 0) occ_smem[threadIdx.x] = x.l[0];
    out[tid] = x;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t splitmix() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    const int n = 1024;
    std::vector<Fp> ha(n), hb(n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < 12; j += 2) { uint64_t v = splitmix(); ha[i].l[j] = (uint32_t)v; ha[i].l[j + 1] = (uint32_t)(v >> 32); v = splitmix(); hb[i].l[j] = (uint32_t)v; hb[i].l[j + 1] = (uint32_t)(v >> 32); }
        ha[i].l[11] &= 0x0fffffffu; hb[i].l[11] &= 0x0fffffffu;   // < 2^380 < p
    }
    // edge values: 0, 1, p-1, all-ones-ish
    ha[0] = Fp::zero(); hb[0] = Fp::one(); ha[1] = neg(Fp::one()); hb[1] = neg(Fp::one()); ha[2] = neg(Fp::one()); hb[2] = Fp::zero(); ha[3] = Fp::one(); hb[3] = neg(Fp::one());
    Fp *da, *db, *dp, *ds, *dd, *dt, *dr, *dout;
    CK(hipMalloc(&da, n * sizeof(Fp))); CK(hipMalloc(&db, n * sizeof(Fp))); CK(hipMalloc(&dp, n * sizeof(Fp)));
    CK(hipMalloc(&ds, n * sizeof(Fp))); CK(hipMalloc(&dd, n * sizeof(Fp))); CK(hipMalloc(&dt, n * sizeof(Fp))); CK(hipMalloc(&dr, n * sizeof(Fp)));
    CK(hipMemcpy(da, ha.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, da, db, dp, ds, dd, dt, dr, n);
    CK(hipDeviceSynchronize());
    std::vector<Fp> hp(n), hs(n), hd(n), ht(n), hr(n);
    CK(hipMemcpy(hp.data(), dp, n * sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), ds, n * sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), dd, n * sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht.data(), dt, n * sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), dr, n * sizeof(Fp), hipMemcpyDeviceToHost));
    int bad[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        if (hp[i] != mul(ha[i], hb[i])) ++bad[0];
        if (hs[i] != add(ha[i], hb[i])) ++bad[1];
        if (hd[i] != sub(ha[i], hb[i])) ++bad[2];
        const Fp e = add(add(mul(add(ha[i], hb[i]), sub(ha[i], hb[i])), mul(ha[i], hb[i])), mul(hb[i], hb[i]));
        if (ht[i] != e) ++bad[3];
        if (hr[i] != ha[i]) ++bad[4];
    }
    printf("fq28 vs host field: mul %d, add %d, sub %d, lazy dot3 %d, round trip %d mismatches of %d -> %s\n", bad[0], bad[1], bad[2], bad[3], bad[4], n, (bad[0] | bad[1] | bad[2] | bad[3] | bad[4]) ? "FAIL" : "ok");

    const int blocks = prop.multiProcessorCount * 8, iters = 4096;
    CK(hipMalloc(&dout, (size_t)blocks * 256 * sizeof(Fp)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, double ops_per_iter) -> int {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, da, dout, 16);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, da, dout, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double ops = (double)blocks * 256 * iters * ops_per_iter;
        printf("%-44s %8.3f ms   %7.2f G products/s\n", name, best, ops / (best * 1e-3) * 1e-9);
        return 0;
    };
    hipLaunchKernelGGL(k_check_blk, dim3(n / 256), dim3(256), 0, 0, da, db, dp, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hp.data(), dp, n * sizeof(Fp), hipMemcpyDeviceToHost));
    { int badb = 0; for (int i = 0; i < n; ++i) if (hp[i] != mul(ha[i], hb[i])) ++badb; printf("blocked-asm 12x32 mul vs host: %d mismatches\n", badb); }
    run("fp  (12x32) mul chain", k_fp_mul_chain, 1);
    run("fp  (12x32) mul chain, 4 limb products / asm block", k_fp_mulblk_chain, 1);
    run("fq28 (14x28) mul chain", k_fq_mul_chain, 1);
    run("fp  dot<6> (6 products, 1 reduction)", k_fp_dot6_chain, 6);
    run("fq28 dot<6> (6 products, 1 reduction)", k_fq_dot6_chain, 6);
    run("fp  mix: (x+y)(z-x) per product", k_fp_mix_chain, 1);
    run("fq28 mix: (x+y)(z-x) per product, lazy", k_fq_mix_chain, 1);
    auto run_occ = [&](const char* name, auto kern, int waves_per_simd, double ops_per_iter, int bs = 64) -> int {
        const size_t lds = (size_t)(160 * 1024 / (4 * waves_per_simd) * (bs / 64)) - 512;          // per block
        if (lds > 64 * 1024) { printf("%s: block of %d needs %zu B of LDS at %d waves/SIMD: skipped\n", name, bs, lds, waves_per_simd); return 0; }
        const int nb = prop.multiProcessorCount * 4 * waves_per_simd * (waves_per_simd >= 8 ? 1 : 2) / (bs / 64);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(bs), lds, 0, da, dout, 16);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(bs), lds, 0, da, dout, 2048);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double ops = (double)nb * bs * 2048 * ops_per_iter;
        printf("%-40s %d waves/SIMD %8.3f ms   %7.2f G products/s\n", name, waves_per_simd, best, ops / (best * 1e-3) * 1e-9);
        return 0;
    };
    auto run_grid = [&](const char* name, auto kern, int waves_per_simd, int bs) -> int {      // occupancy set by the GRID size (no LDS), one resident batch
        const int nb = prop.multiProcessorCount * 4 * waves_per_simd * 64 / bs;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(bs), 0, 0, da, dout, 16);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(bs), 0, 0, da, dout, 4096);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-34s grid-limited %d waves/SIMD, blocks of %3d: %8.3f ms   %7.2f G products/s\n", name, waves_per_simd, bs, best, (double)nb * bs * 4096 / (best * 1e-3) * 1e-9);
        return 0;
    };
    for (int w = 1; w <= 8; w *= 2) for (int bs = 64; bs <= 256; bs *= 2) run_grid("fq28 mul chain", k_fq_mul_chain_occ, w, bs);
    for (int w = 1; w <= 8; w *= 2) { run_occ("fp  mul chain", k_fp_mul_chain_occ, w, 1); run_occ("fq28 mul chain, blocks of 64", k_fq_mul_chain_occ, w, 1); run_occ("fq28 mul chain, blocks of 128", k_fq_mul_chain_occ, w, 1, 128); run_occ("fq28 mul chain, blocks of 256", k_fq_mul_chain_occ, w, 1, 256); run_occ("fq28 dot<2> chain", k_fq_dot2_chain_occ, w, 2); }
    return 0;
}
