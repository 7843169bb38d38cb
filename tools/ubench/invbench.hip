// Fp inversion on the device: binary GCD with 30-step inner loops (fp_inv_bingcd) against the bit-serial Kaliski form, checked against the host Fermat inverse.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../ripp_amd/csrc -o build/invbench invbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bls12_381/curve.hpp"
using namespace ripp;
__global__ void k_inv(const Fp* a, Fp* o1, Fp* o2, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
  o1[i] = fp_inv_bingcd(a[i]); o2[i] = fp_inv_kaliski(a[i]);
#endif
}
__global__ void k_time(const Fp* a, Fp* o, int n, int which) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
  Fp x = a[i]; for (int r = 0; r < 8; ++r) x = which ? fp_inv_bingcd(x) : fp_inv_kaliski(x); o[i] = x;
#endif
}
static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t sm() { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main() {
  const int n = 4096; std::vector<Fp> h(n);
  for (int i = 0; i < n; ++i) { for (int j = 0; j < 12; j += 2) { uint64_t v = sm(); h[i].l[j] = (uint32_t)v; h[i].l[j + 1] = (uint32_t)(v >> 32); } h[i].l[11] &= 0x0fffffffu; if (i % 7 == 0) for (int j = 1 + i % 11; j < 12; ++j) h[i].l[j] = 0; }
  h[0] = Fp::zero(); h[1] = Fp::one(); h[2] = neg(Fp::one()); for (int j = 0; j < 12; ++j) h[3].l[j] = 0; h[3].l[0] = 1; h[4] = h[3]; h[4].l[0] = 2; h[5] = neg(h[3]);
  Fp *d, *o1, *o2; hipMalloc(&d, n * sizeof(Fp)); hipMalloc(&o1, n * sizeof(Fp)); hipMalloc(&o2, n * sizeof(Fp));
  hipMemcpy(d, h.data(), n * sizeof(Fp), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_inv, dim3(n / 64), dim3(64), 0, 0, d, o1, o2, n); hipDeviceSynchronize();
  std::vector<Fp> r1(n), r2(n); hipMemcpy(r1.data(), o1, n * sizeof(Fp), hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, n * sizeof(Fp), hipMemcpyDeviceToHost);
  int bad = 0, badk = 0; for (int i = 0; i < n; ++i) { const Fp e = h[i].is_zero() ? Fp::zero() : inv(h[i]); if (r1[i] != e) { if (bad < 5) printf("mismatch at %d\n", i); ++bad; } if (r2[i] != e) ++badk; }
  printf("bingcd mismatches %d, kaliski mismatches %d of %d\n", bad, badk, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) for (int m : {64, 1024 * 64}) { hipLaunchKernelGGL(k_time, dim3(m / 64), dim3(64), 0, 0, d, o1, m > n ? n : m, which); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_time, dim3((m > n ? n : m) / 64), dim3(64), 0, 0, d, o1, m > n ? n : m, which); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %d lanes, 8 dependent inversions: %.3f ms (%.1f us each)\n", which ? "bingcd " : "kaliski", m > n ? n : m, ms, ms * 1000 / 8); }
  return 0; }
