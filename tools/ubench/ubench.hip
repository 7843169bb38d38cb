// Instruction-throughput microbenchmark for gfx950 integer / fp64 VALU ops that
// decide the limb representation of the BLS12-381 field arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run on the GPU box.
// Reports wave-instructions per SIMD-cycle relative to v_add_u32 (known: 2 cyc / wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 32768;   // loop trips
constexpr int UNROLL = 16;    // asm statements per trip (8 chains x 2)

// Each kernel: 8 independent chains, UNROLL instrs per trip.
#define KERNEL_BEGIN(name) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, \
             a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7; \
    uint32_t m = seed | 1u, k = seed * 2654435761u + 12345u; (void)m; (void)k;

#define KERNEL_END \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; }

#define REP8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)

// ---- 32-bit ops ----
#define S_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(m));
KERNEL_BEGIN(k_add_u32) for (int i = 0; i < ITERS; ++i) { REP8(S_ADD) REP8(S_ADD) } KERNEL_END

#define S_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(m));
KERNEL_BEGIN(k_mul_lo_u32) for (int i = 0; i < ITERS; ++i) { REP8(S_MULLO) REP8(S_MULLO) } KERNEL_END

#define S_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(k));
KERNEL_BEGIN(k_mul_hi_u32) for (int i = 0; i < ITERS; ++i) { REP8(S_MULHI) REP8(S_MULHI) } KERNEL_END

#define S_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(m));
KERNEL_BEGIN(k_mul_u32_u24) for (int i = 0; i < ITERS; ++i) { REP8(S_MUL24) REP8(S_MUL24) } KERNEL_END

#define S_MULHI24(x) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(k));
KERNEL_BEGIN(k_mul_hi_u32_u24) for (int i = 0; i < ITERS; ++i) { REP8(S_MULHI24) REP8(S_MULHI24) } KERNEL_END

#define S_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k));
KERNEL_BEGIN(k_mad_u32_u24) for (int i = 0; i < ITERS; ++i) { REP8(S_MAD24) REP8(S_MAD24) } KERNEL_END

#define S_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k));
KERNEL_BEGIN(k_add3_u32) for (int i = 0; i < ITERS; ++i) { REP8(S_ADD3) REP8(S_ADD3) } KERNEL_END

#define S_ADDCO(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(x) : "v"(m), "v"(k) : "vcc");
KERNEL_BEGIN(k_addco_addc_pair) for (int i = 0; i < ITERS; ++i) { REP8(S_ADDCO) } KERNEL_END

#define S_ALIGNBIT(x) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(m));
KERNEL_BEGIN(k_alignbit) for (int i = 0; i < ITERS; ++i) { REP8(S_ALIGNBIT) REP8(S_ALIGNBIT) } KERNEL_END

#define S_MADU16(x) asm volatile("v_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k));
KERNEL_BEGIN(k_mad_u16) for (int i = 0; i < ITERS; ++i) { REP8(S_MADU16) REP8(S_MADU16) } KERNEL_END

#define S_PKMADU16(x) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k));
KERNEL_BEGIN(k_pk_mad_u16) for (int i = 0; i < ITERS; ++i) { REP8(S_PKMADU16) REP8(S_PKMADU16) } KERNEL_END

#define S_DOT4(x) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(m), "v"(k));
KERNEL_BEGIN(k_dot4_u32_u8) for (int i = 0; i < ITERS; ++i) { REP8(S_DOT4) REP8(S_DOT4) } KERNEL_END

#define S_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k));
KERNEL_BEGIN(k_fma_f32) for (int i = 0; i < ITERS; ++i) { REP8(S_FMA32) REP8(S_FMA32) } KERNEL_END

// ---- 64-bit ops (register pairs) ----
#define KERNEL64_BEGIN(name) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
    uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, \
             a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7; \
    uint32_t m = seed | 1u, k = seed * 2654435761u + 12345u; (void)m; (void)k; \
    uint64_t mm = ((uint64_t)k << 32) | m; (void)mm;
#define KERNEL64_END \
    uint64_t r_ = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r_ ^ (uint32_t)(r_ >> 32); }

// D.u64 = S0.u32 * S1.u32 + S2.u64 (carry-out to vcc)
#define S_MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(m), "v"(k) : "vcc");
KERNEL64_BEGIN(k_mad_u64_u32) for (int i = 0; i < ITERS; ++i) { REP8(S_MAD64) REP8(S_MAD64) } KERNEL64_END

// dependent-on-multiplicand chain: product operand comes from previous result (latency probe, 1 chain)
__global__ void __launch_bounds__(64) k_mad_u64_u32_lat(uint32_t* out, uint32_t seed) {
    uint64_t a0 = threadIdx.x + seed; uint32_t m = seed | 1u, k = seed * 2654435761u + 12345u;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(m), "v"(k) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32);
}

#define S_LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x) : "v"(mm));
KERNEL64_BEGIN(k_lshl_add_u64) for (int i = 0; i < ITERS; ++i) { REP8(S_LSHLADD64) REP8(S_LSHLADD64) } KERNEL64_END

#define S_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(mm));
KERNEL64_BEGIN(k_fma_f64) for (int i = 0; i < ITERS; ++i) { REP8(S_FMA64) REP8(S_FMA64) } KERNEL64_END

#define S_MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(mm));
KERNEL64_BEGIN(k_mul_f64) for (int i = 0; i < ITERS; ++i) { REP8(S_MUL64) REP8(S_MUL64) } KERNEL64_END

#define S_ADD64F(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(mm));
KERNEL64_BEGIN(k_add_f64) for (int i = 0; i < ITERS; ++i) { REP8(S_ADD64F) REP8(S_ADD64F) } KERNEL64_END


// MAD with carry-out captured by a following addc (candidate column-accumulate primitive)
__global__ void __launch_bounds__(256) k_madc_pair(uint32_t* out, uint32_t seed) {
    uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint32_t m = seed | 1u, k = seed * 2654435761u + 12345u;
    for (int i = 0; i < ITERS; ++i) {
#define S_MADC(x, h) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(x), "+v"(h) : "v"(m), "v"(k) : "vcc");
        S_MADC(a0, h0) S_MADC(a1, h1) S_MADC(a2, h2) S_MADC(a3, h3)
        S_MADC(a0, h0) S_MADC(a1, h1) S_MADC(a2, h2) S_MADC(a3, h3)
    }
    uint64_t r_ = a0 ^ a1 ^ a2 ^ a3; out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r_ ^ (uint32_t)(r_ >> 32) ^ h0 ^ h1 ^ h2 ^ h3;
}
// same but single accumulator chain (what a column of a product-scanning multiplier looks like)
__global__ void __launch_bounds__(256) k_madc_chain1(uint32_t* out, uint32_t seed) {
    uint64_t a0 = threadIdx.x + seed; uint32_t h0 = 0;
    uint32_t m = seed | 1u, k = seed * 2654435761u + 12345u;
    for (int i = 0; i < ITERS; ++i) {
        S_MADC(a0, h0) S_MADC(a0, h0) S_MADC(a0, h0) S_MADC(a0, h0)
        S_MADC(a0, h0) S_MADC(a0, h0) S_MADC(a0, h0) S_MADC(a0, h0)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32) ^ h0;
}
// MAD with one SGPR multiplicand
#define S_MAD64S(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(m), "s"(seed) : "vcc");
KERNEL64_BEGIN(k_mad_u64_u32_sgpr) for (int i = 0; i < ITERS; ++i) { REP8(S_MAD64S) REP8(S_MAD64S) } KERNEL64_END
#define S_ADDCO1(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(m) : "vcc");
KERNEL_BEGIN(k_add_co_only) for (int i = 0; i < ITERS; ++i) { REP8(S_ADDCO1) REP8(S_ADDCO1) } KERNEL_END
#define S_ADDC1(x) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
KERNEL_BEGIN(k_addc_only) for (int i = 0; i < ITERS; ++i) { REP8(S_ADDC1) REP8(S_ADDC1) } KERNEL_END
#define S_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(m));
KERNEL_BEGIN(k_mov) for (int i = 0; i < ITERS; ++i) { REP8(S_MOV) REP8(S_MOV) } KERNEL_END
#define S_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
KERNEL_BEGIN(k_cndmask) for (int i = 0; i < ITERS; ++i) { REP8(S_CNDMASK) REP8(S_CNDMASK) } KERNEL_END
#define S_MADI64(x) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(m), "v"(k) : "vcc");
KERNEL64_BEGIN(k_mad_i64_i32) for (int i = 0; i < ITERS; ++i) { REP8(S_MADI64) REP8(S_MADI64) } KERNEL64_END
#define S_LSHR64(x) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(x));
KERNEL64_BEGIN(k_lshrrev_b64) for (int i = 0; i < ITERS; ++i) { REP8(S_LSHR64) REP8(S_LSHR64) } KERNEL64_END

typedef void (*kfn)(uint32_t*, uint32_t);
struct Entry { const char* name; kfn fn; int instr_per_trip; int block; };

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  arch=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.gcnArchName);
    const int blocks = prop.multiProcessorCount * 8;   // 8 blocks x 4 waves = 8 waves/SIMD
    uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(uint32_t)));
    std::vector<Entry> es = {
        {"v_add_u32", k_add_u32, 16, 256}, {"v_add3_u32", k_add3_u32, 16, 256},
        {"v_add_co+v_addc_co (pair)", k_addco_addc_pair, 16, 256},
        {"v_alignbit_b32", k_alignbit, 16, 256},
        {"v_mul_lo_u32", k_mul_lo_u32, 16, 256}, {"v_mul_hi_u32", k_mul_hi_u32, 16, 256},
        {"v_mul_u32_u24", k_mul_u32_u24, 16, 256}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 16, 256},
        {"v_mad_u32_u24", k_mad_u32_u24, 16, 256}, {"v_mad_u16", k_mad_u16, 16, 256},
        {"v_pk_mad_u16", k_pk_mad_u16, 16, 256}, {"v_dot4_u32_u8", k_dot4_u32_u8, 16, 256},
        {"v_fma_f32", k_fma_f32, 16, 256},
        {"v_mad_u64_u32", k_mad_u64_u32, 16, 256}, {"v_lshl_add_u64", k_lshl_add_u64, 16, 256},
        {"v_mad_u64_u32 (sgpr src)", k_mad_u64_u32_sgpr, 16, 256}, {"v_mad_i64_i32", k_mad_i64_i32, 16, 256},
        {"mad_u64+addc (x4 chains, 2 instr)", k_madc_pair, 16, 256}, {"mad_u64+addc (1 chain, 2 instr)", k_madc_chain1, 16, 256},
        {"v_add_co_u32", k_add_co_only, 16, 256}, {"v_addc_co_u32", k_addc_only, 16, 256}, {"v_mov_b32", k_mov, 16, 256}, {"v_cndmask_b32", k_cndmask, 16, 256},
        {"v_lshrrev_b64", k_lshrrev_b64, 16, 256},
        {"v_fma_f64", k_fma_f64, 16, 256}, {"v_mul_f64", k_mul_f64, 16, 256}, {"v_add_f64", k_add_f64, 16, 256},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double base = 0;
    for (auto& e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(e.block), 0, 0, out, 12345u);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(e.block), 0, 0, out, 12345u + r);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double wave_instr = (double)blocks * (e.block / 64) * (double)ITERS * e.instr_per_trip;
        double per_simd_per_s = wave_instr / (prop.multiProcessorCount * 4.0) / (best * 1e-3);
        if (base == 0) base = per_simd_per_s;   // v_add_u32 == 2 cycles/wave-instr
        printf("%-34s %8.3f ms  %8.2f G wave-instr/s/SIMD   cyc/wave-instr (rel. add=2): %6.2f   lane-ops/s chip: %.2f T\n",
               e.name, best, per_simd_per_s * 1e-9, 2.0 * base / per_simd_per_s,
               wave_instr * 64 / (best * 1e-3) * 1e-12);
    }
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD sweep for the MAD (latency hiding)
        int b2 = prop.multiProcessorCount * wps;
        hipLaunchKernelGGL(k_mad_u64_u32, dim3(b2), dim3(256), 0, 0, out, 1u); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mad_u64_u32, dim3(b2), dim3(256), 0, 0, out, 2u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("v_mad_u64_u32 @%d waves/SIMD (8 indep chains): %.2f ns per wave-instr per SIMD\n", wps, ms * 1e6 / ((double)ITERS * 16 * wps));
        hipLaunchKernelGGL(k_madc_chain1, dim3(b2), dim3(256), 0, 0, out, 2u); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_madc_chain1, dim3(b2), dim3(256), 0, 0, out, 2u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mad+addc single chain @%d waves/SIMD: %.2f ns per wave-instr per SIMD\n", wps, ms * 1e6 / ((double)ITERS * 16 * wps));
    }
    // latency probe
    {
        hipLaunchKernelGGL(k_mad_u64_u32_lat, dim3(prop.multiProcessorCount * 4), dim3(64), 0, 0, out, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mad_u64_u32_lat, dim3(prop.multiProcessorCount * 4), dim3(64), 0, 0, out, 2u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double n = (double)ITERS * UNROLL;
        printf("v_mad_u64_u32 dependent chain (1 wave/SIMD): %.2f ns per instr (x clock GHz = cycles)\n", ms * 1e6 / n);
    }
    return 0;
}
