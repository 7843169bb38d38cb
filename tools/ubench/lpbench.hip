// Stage 2a of the pairing product in isolation: k_line_products_q (fq_line_products.hpp) against its variants on a synthetic line buffer.
// Every variant must write the SAME bytes (canonical per-group partial products); the bench compares them and times each at the engine's launch shape.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../ripp_amd/csrc -o build/lpbench lpbench.hip
// Run:   build/lpbench [log2 pairs = 17] [products = 2] [reps = 5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fq_line_products.hpp"
#include "fq_line_products_k.hpp"

using namespace ripp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// pseudo-random canonical field elements (< 2^380 < p) for every 48-byte coefficient of the line buffer: chunk c of coefficient f of line i of row r
__global__ void k_fill_lines(uint4* lines, size_t stride, uint32_t M, size_t rows) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t rc = blockIdx.y;                     // row * 18 + chunk
    if (i >= M) return;
    uint64_t z = (rc * 0x9E3779B97F4A7C15ull) ^ (i * 0xBF58476D1CE4E5B9ull) ^ 0x1234567ull;
    auto next = [&]() { z += 0x9E3779B97F4A7C15ull; uint64_t t = z; t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ull; t = (t ^ (t >> 27)) * 0x94D049BB133111EBull; return t ^ (t >> 31); };
    const uint64_t a = next(), b = next();
    uint4 v{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if (rc % 3 == 2) v.w &= 0x0fffffffu;              // the top word of the 12
    lines[rc * stride + i] = v;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 17;
    const int nprod = argc > 2 ? atoi(argv[2]) : 2;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const size_t n_simd = (size_t)prop.multiProcessorCount * 4;
    const uint32_t M = 1u << lg;
    const size_t rows = (size_t)nprod * 68, stride = (M + 63) & ~(size_t)63;
    auto t_for = [&](uint32_t per_wave) { uint32_t t = (uint32_t)std::max<size_t>(per_wave, (n_simd * 2 / rows) * per_wave); return t > M ? M : t; };
    const uint32_t Tq = t_for(LP_GROUPS_PER_WAVE), Tk = t_for(LK_GROUPS_PER_WAVE);
    printf("device: %s CUs=%d   rows=%zu M=%u   T: %u (3 lanes per accumulator, %zu waves) / %u (6 lanes, %zu waves)\n", prop.name, prop.multiProcessorCount, rows, M,
           Tq, rows * ((Tq + 20) / 21), Tk, rows * ((Tk + 9) / 10));
    uint4 *lines, *pa, *pb;
    CK(hipMalloc(&lines, rows * 18 * stride * sizeof(uint4)));
    const size_t pbytes = rows * 36 * (size_t)Tq * sizeof(uint4), kbytes = rows * 36 * (size_t)Tk * sizeof(uint4);
    CK(hipMalloc(&pa, pbytes)); CK(hipMalloc(&pb, pbytes));
    hipLaunchKernelGGL(k_fill_lines, dim3((M + 255) / 256, (unsigned)(rows * 18)), dim3(256), 0, 0, lines, stride, M, rows);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // which: 0 = k_line_products_q at its own T, 1 = k_line_products_k, 2 = k_line_products_q at the T of 1 (the same partition of the lines: the same bytes)
    auto run = [&](int which, uint4* out) {
        if (which == 0) hipLaunchKernelGGL(k_line_products_q, dim3((Tq + LP_GROUPS_PER_WAVE - 1) / LP_GROUPS_PER_WAVE, (unsigned)rows), dim3(64), 0, 0, lines, stride, M, out, Tq);
        else if (which == 1) hipLaunchKernelGGL(k_line_products_k, dim3((Tk + LK_GROUPS_PER_WAVE - 1) / LK_GROUPS_PER_WAVE, (unsigned)rows), dim3(64), 0, 0, lines, stride, M, out, Tk);
        else hipLaunchKernelGGL(k_line_products_q, dim3((Tk + LP_GROUPS_PER_WAVE - 1) / LP_GROUPS_PER_WAVE, (unsigned)rows), dim3(64), 0, 0, lines, stride, M, out, Tk);
    };
    const char* names[2] = {"k_line_products_q (six-product sums)", "k_line_products_k (Karatsuba sums)"};
    for (int which = 0; which < 2; ++which) {
        uint4* out = which ? pb : pa;
        CK(hipMemset(out, 0xff, pbytes));
        run(which, out); CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0)); run(which, out); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-40s best %8.3f ms  mean %8.3f ms   %.2f M pairs/s\n", names[which], best, sum / reps, (double)M * nprod / best / 1e3);
    }
    CK(hipMemset(pa, 0xff, pbytes)); run(2, pa); CK(hipDeviceSynchronize());
    std::vector<uint8_t> ha(kbytes), hb(kbytes);
    CK(hipMemcpy(ha.data(), pa, kbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), pb, kbytes, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < kbytes; ++i) bad += ha[i] != hb[i];
    printf("outputs at T = %u: %zu of %zu bytes differ -> %s\n", Tk, bad, kbytes, bad ? "MISMATCH" : "identical");
    return bad ? 2 : 0;
}
