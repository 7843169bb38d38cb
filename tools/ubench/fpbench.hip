// Fp multiplication throughput + host/device agreement check (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -I../../ripp_amd/csrc -o fpbench fpbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "line_products.hpp"

using namespace ripp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int CHAINS>
__global__ void __launch_bounds__(256) k_mul_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x[CHAINS];
    Fp y = in[(tid + 1) & 1023];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = in[(tid + 7 * c) & 1023];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = mul(x[c], y);
    }
    Fp acc = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; ++c) acc = add(acc, x[c]);
    out[tid] = acc;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k_mulcios_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x[CHAINS];
    Fp y = in[(tid + 1) & 1023];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = in[(tid + 7 * c) & 1023];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = mul_cios(x[c], y);
    }
    Fp acc = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; ++c) acc = add(acc, x[c]);
    out[tid] = acc;
}
__device__ __noinline__ Fp mul_noinline(const Fp& a, const Fp& b) { return mul(a, b); }
__global__ void __launch_bounds__(256) k_mulni_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023];
    for (int i = 0; i < iters; ++i) x = mul_noinline(x, y);
    out[tid] = x;
}

// the lazily reduced 6-product sum of line_products.hpp: 6 limb-product blocks + ONE Montgomery reduction per result
__global__ void __launch_bounds__(256, 2) k_dot6_chain(const Fp* in, Fp* out, int iters) {
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
// the group law's Fp2 product is two of these calls (48 dwords of arguments: 16 of them travel through the stack)
__global__ void __launch_bounds__(256, 2) k_mul2call_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023], z = in[(tid + 2) & 1023], w = in[(tid + 3) & 1023];
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < iters; ++i) { x = fmul2_add(x, y, z, w); z = fmul2_add(z, w, x, y); }
#endif
    out[tid] = add(x, z);
}
__global__ void __launch_bounds__(256, 2) k_mul2inl_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023], z = in[(tid + 2) & 1023], w = in[(tid + 3) & 1023];
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < iters; ++i) { x = mul2_add(x, y, z, w); z = mul2_add(z, w, x, y); }
#endif
    out[tid] = add(x, z);
}
__global__ void __launch_bounds__(256, 2) k_mulcall_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023];
    for (int i = 0; i < iters; ++i) { x = fmul(x, y); y = fmul(y, x); }
    out[tid] = add(x, y);
}
__global__ void __launch_bounds__(256) k_add_chain(const Fp* in, Fp* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x = in[tid & 1023], y = in[(tid + 1) & 1023];
    for (int i = 0; i < iters; ++i) { x = add(x, y); y = sub(y, x); }
    out[tid] = add(x, y);
}

__global__ void k_check(const Fp* a, const Fp* b, Fp* prod, Fp* sum, Fp* dif, Fp* inv_out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    prod[i] = mul(a[i], b[i]);
    sum[i] = add(a[i], b[i]);
    dif[i] = sub(a[i], b[i]);
    if (i < 64) inv_out[i] = mul(inv(a[i]), a[i]);
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
    Fp *da, *db, *dp, *ds, *dd, *di, *dout;
    CK(hipMalloc(&da, n * sizeof(Fp))); CK(hipMalloc(&db, n * sizeof(Fp))); CK(hipMalloc(&dp, n * sizeof(Fp)));
    CK(hipMalloc(&ds, n * sizeof(Fp))); CK(hipMalloc(&dd, n * sizeof(Fp))); CK(hipMalloc(&di, 64 * sizeof(Fp)));
    CK(hipMemcpy(da, ha.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * sizeof(Fp), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, da, db, dp, ds, dd, di, n);
    CK(hipDeviceSynchronize());
    std::vector<Fp> hp(n), hs(n), hd(n), hi(64);
    CK(hipMemcpy(hp.data(), dp, n * sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), ds, n * sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), dd, n * sizeof(Fp), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hi.data(), di, 64 * sizeof(Fp), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        if (hp[i] != mul(ha[i], hb[i])) ++bad;
        if (hs[i] != add(ha[i], hb[i])) ++bad;
        if (hd[i] != sub(ha[i], hb[i])) ++bad;
    }
    for (int i = 0; i < 64; ++i) if (hi[i] != Fp::one()) ++bad;
    printf("host/device agreement: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    printf("prod[0] limbs:"); for (int j = 0; j < 12; ++j) printf(" %08x", hp[0].l[j]); printf("\n");
    printf("a[0] limbs:"); for (int j = 0; j < 12; ++j) printf(" %08x", ha[0].l[j]); printf("\n");
    printf("b[0] limbs:"); for (int j = 0; j < 12; ++j) printf(" %08x", hb[0].l[j]); printf("\n");

    const int blocks = prop.multiProcessorCount * 8, iters = 4096;
    CK(hipMalloc(&dout, (size_t)blocks * 256 * sizeof(Fp)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int chains, double ops_per_iter) -> int {
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
        printf("%-22s %8.3f ms   %.2f G field-ops/s   (%.1f SIMD-cycles per wave-op at 2.4 GHz)\n", name, best, ops / (best * 1e-3) * 1e-9,
               (prop.multiProcessorCount * 4 * 2.4e9) / (ops / 64 / (best * 1e-3)));
        return 0;
    };
    run("fp_mul x1 chain", k_mul_chain<1>, 1, 1);
    run("fp_mul CIOS(compiler) x1", k_mulcios_chain<1>, 1, 1);
    run("fp_mul noinline call x1", k_mulni_chain, 1, 1);
    run("fp_mul x2 chains", k_mul_chain<2>, 2, 2);
    run("fp_mul x4 chains", k_mul_chain<4>, 4, 4);
    run("fp_dot<6> (6 products, 1 reduction)", k_dot6_chain, 1, 6);
    run("fmul (fp_mul_call, 24 reg args) x2", k_mulcall_chain, 1, 2);
    run("fmul2_add call (48 args, 16 via stack) x2", k_mul2call_chain, 1, 4);
    run("mul2_add inlined x2", k_mul2inl_chain, 1, 4);
    run("fp_add+fp_sub", k_add_chain, 1, 2);
    return 0;
}
