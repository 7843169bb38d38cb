// The carry-free group law in isolation (fq_curve.hpp / fq_curve2.hpp): chains of Fp products, G1 doublings / mixed additions and G2 doublings / mixed additions at the
// occupancy the throughput kernels run with (2 waves per SIMD, forced with dynamic LDS).  Built twice to compare forms of the field product:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../ripp_amd/csrc [-DFQ_CHAIN_TIES=0] [-DFQ_DEDICATED_SQR=0] -o build/fqgroup fqgroup.hip
// The checksums of two builds must agree (same values, canonical output).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "fq_curve2.hpp"

using namespace ripp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

extern __shared__ uint4 dyn_lds[];

__global__ void __launch_bounds__(256, 2) k_mul_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn x = fq_from_fp(in[tid & 1023]), y = fq_from_fp(in[(tid + 1) & 1023]);
#pragma unroll 1
    for (int i = 0; i < iters; ++i) { x = fq_mul(x, y); y = fq_sqr(x); }
    out[tid] = fq_to_fp(fq_reduce(fq_add(x, y)));
#endif
}
// the same products with a loop body of NB products: what the code size alone costs (a product is ~4.5 KB of instructions, the instruction cache 64 KB per two CUs)
template <int NB>
__global__ void __launch_bounds__(256, 2) k_mul_body(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    Fqn x = fq_from_fp(in[tid & 1023]), y = fq_from_fp(in[(tid + 1) & 1023]);
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int b = 0; b < NB / 2; ++b) { x = fq_mul(x, y); fq_pin(x); y = fq_mul(y, x); fq_pin(y); }
    }
    out[tid] = fq_to_fp(fq_reduce(fq_add(x, y)));
#endif
}
// one product + one piece of the group law's glue per step: 0 nothing, 1 fq_norm of a lazy sum, 2 fq_reduce of a lazy difference, 3 the zero test, 4 two lazy additions + a lazy subtraction
template <int WHAT>
__global__ void __launch_bounds__(256, 2) k_glue(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    FqC x = fq_coord(fq_from_fp(in[tid & 1023])); const Fqn y = fq_from_fp(in[(tid + 1) & 1023]);
    bool z = false;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        const Fqn t = fq_mul(x, y);
        if (WHAT == 0) x = fq_coord(t);
        else if (WHAT == 1) x = fq_coord(fq_norm(fq_add(t, y)));
        else if (WHAT == 2) x = fq_coord(fq_reduce(fq_sub(t, y)));
        else if (WHAT == 3) { z |= fq_is_zero(t); x = fq_coord(t); }
        else x = fq_coord(fq_mul(fq_sub(fq_add(t, y), fq_add(y, y)), y));
    }
    out[tid] = fq_to_fp(fq_reduce(x));
    if (z && iters < 0) out[0] = Fp{};
#endif
}
template <int WHAT>
__global__ void __launch_bounds__(256, 2) k_g1_part(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    JacQ p; jq_set(p, fq_from_fp(in[tid & 1023]), fq_from_fp(in[(tid + 1) & 1023]), fq_one());
    const Fqn qx = fq_from_fp(in[(tid + 2) & 1023]), qy = fq_from_fp(in[(tid + 3) & 1023]);
    bool bad = false;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) { if (WHAT == 0) jdbl_q(p); else bad |= jmadd_q(p, qx, qy); }
    out[tid] = fq_to_fp(fq_reduce(fq_add(fq_add(p.x, p.y), p.z)));
    if (bad && iters < 0) out[0] = Fp{};
#endif
}
__global__ void __launch_bounds__(256, 2) k_g1_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    JacQ p; jq_set(p, fq_from_fp(in[tid & 1023]), fq_from_fp(in[(tid + 1) & 1023]), fq_one());
    const Fqn qx = fq_from_fp(in[(tid + 2) & 1023]), qy = fq_from_fp(in[(tid + 3) & 1023]);
    bool bad = false;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) { jdbl_q(p); bad |= jmadd_q(p, qx, qy); }      // (not curve points: the formulas are polynomial maps, the values are what is compared)
    out[tid] = fq_to_fp(fq_reduce(fq_add(fq_add(p.x, p.y), p.z)));
    if (bad && iters < 0) out[0] = Fp{};
#endif
}
__global__ void __launch_bounds__(64, 2) k_g2_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* park = dyn_lds + threadIdx.x;
    auto ld = [&](int o) { return Fq2n{fq_from_fp(in[(tid + o) & 1023]), fq_from_fp(in[(tid + o + 1) & 1023])}; };
    JacQ2 p; j2_set(p, ld(0), ld(2), Fq2n{fq_one(), fq_zero()});
    bool bad = false;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) { jdbl2_q(p); bad |= jmadd2_q(p, [&]() { return ld(4); }, [&]() { return ld(6); }, park); }
    out[tid] = fq_to_fp(fq_reduce(fq_add(fq_add(fq_add(p.x.c0, p.x.c1), fq_add(p.y.c0, p.y.c1)), fq_add(p.z.c0, p.z.c1))));
    if (bad && iters < 0) out[0] = Fp{};
#endif
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t splitmix() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n = 1024;
    std::vector<Fp> h(n);
    for (int i = 0; i < n; ++i) { for (int j = 0; j < 12; j += 2) { const uint64_t v = splitmix(); h[i].l[j] = (uint32_t)v; h[i].l[j + 1] = (uint32_t)(v >> 32); } h[i].l[11] &= 0x0fffffffu; }
    const int threads = prop.multiProcessorCount * 4 * 2 * 64 * 4;             // four rounds of 2 waves per SIMD
    Fp *din, *dout; CK(hipMalloc(&din, n * sizeof(Fp))); CK(hipMalloc(&dout, (size_t)threads * sizeof(Fp)));
    CK(hipMemcpy(din, h.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Fp> ho(threads);
    auto sum = [&]() { CK(hipMemcpy(ho.data(), dout, (size_t)threads * sizeof(Fp), hipMemcpyDeviceToHost)); uint64_t s = 0; for (int i = 0; i < threads; ++i) for (int j = 0; j < 12; ++j) s = s * 0x100000001B3ull ^ ho[i].l[j]; printf("   checksum %016llx\n", (unsigned long long)s); return 0; };
    // dynamic LDS: 80 KB per 256-thread block, 20 KB per 64-thread block -> 8 waves per CU
    for (int which = 3; which < 9; ++which) {
        float best = 1e30f;
        const char* nm[] = {"", "", "", "products, loop body of 2", "products, loop body of 8", "products, loop body of 18", "products, loop body of 36", "G1 doublings alone", "G1 mixed additions alone"};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            const dim3 g(threads / 256), b(256);
            if (which == 3) hipLaunchKernelGGL(k_mul_body<2>, g, b, 80 * 1024, 0, din, dout, iters * 36);
            else if (which == 4) hipLaunchKernelGGL(k_mul_body<8>, g, b, 80 * 1024, 0, din, dout, iters * 9);
            else if (which == 5) hipLaunchKernelGGL(k_mul_body<18>, g, b, 80 * 1024, 0, din, dout, iters * 4);
            else if (which == 6) hipLaunchKernelGGL(k_mul_body<36>, g, b, 80 * 1024, 0, din, dout, iters * 2);
            else if (which == 7) hipLaunchKernelGGL(k_g1_part<0>, g, b, 80 * 1024, 0, din, dout, iters);
            else hipLaunchKernelGGL(k_g1_part<1>, g, b, 80 * 1024, 0, din, dout, iters);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = ms < best ? ms : best;
        }
        const double ops = which <= 6 ? (double)threads * iters * 72 : (double)threads * iters;
        printf("%-44s %8.3f ms   %8.2f G %s/s\n", nm[which], best, ops / best / 1e6, which <= 6 ? "products" : "group operations");
    }
    for (int which = 0; which < 5; ++which) {
        float best = 1e30f;
        const char* nm[] = {"product", "product + fq_norm(lazy sum)", "product + fq_reduce(lazy difference)", "product + zero test", "2 products + 2 lazy adds + 1 lazy sub"};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            const dim3 g(threads / 256), b(256);
            if (which == 0) hipLaunchKernelGGL(k_glue<0>, g, b, 80 * 1024, 0, din, dout, iters * 16);
            else if (which == 1) hipLaunchKernelGGL(k_glue<1>, g, b, 80 * 1024, 0, din, dout, iters * 16);
            else if (which == 2) hipLaunchKernelGGL(k_glue<2>, g, b, 80 * 1024, 0, din, dout, iters * 16);
            else if (which == 3) hipLaunchKernelGGL(k_glue<3>, g, b, 80 * 1024, 0, din, dout, iters * 16);
            else hipLaunchKernelGGL(k_glue<4>, g, b, 80 * 1024, 0, din, dout, iters * 16);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = ms < best ? ms : best;
        }
        printf("%-44s %8.3f ms   %8.2f ns per step, wave and SIMD\n", nm[which], best, best * 1e6 / ((double)iters * 16 * 4));
    }
    for (int which = 0; which < 3; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k_mul_chain, dim3(threads / 256), dim3(256), 80 * 1024, 0, din, dout, iters * 8);
            else if (which == 1) hipLaunchKernelGGL(k_g1_chain, dim3(threads / 256), dim3(256), 80 * 1024, 0, din, dout, iters);
            else hipLaunchKernelGGL(k_g2_chain, dim3(threads / 64), dim3(64), 20 * 1024, 0, din, dout, iters / 2);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = ms < best ? ms : best;
        }
        const double ops = which == 0 ? (double)threads * iters * 8 * 2 : which == 1 ? (double)threads * iters : (double)threads * (iters / 2);
        printf("%-44s %8.3f ms   %8.2f G %s/s\n", which == 0 ? "Fp mul + sqr chain" : which == 1 ? "G1 doubling + mixed addition chain" : "G2 doubling + mixed addition chain", best, ops / best / 1e6,
               which == 0 ? "products" : "dbl+madd");
        if (sum()) return 1;
    }
    return 0;
}
