// Round-6 experiment: the cache-policy bits of global stores (sc0 / sc1 / nt) on the C2 stream set (4 float64 in, 10 float64 out,
// 16 sub-tiles per wave) and on a pure ten-stream write.  The generated kernels use __builtin_nontemporal_store (= "nt").
//   hipcc --offload-arch=gfx950 -O3 -o store_policy_probe store_policy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
struct Ptrs { const unsigned long long* in[4]; unsigned long long* out[10]; };
template <int POL> __device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
  if (POL == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (POL == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if (POL == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if (POL == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (POL == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  if (POL == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
  if (POL == 6) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
  if (POL == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int NR, int POL>
__global__ void __launch_bounds__(256) K(const Ptrs P, size_t n) {
  constexpr int U = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = n / (256 * U);
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const size_t base = t * (256 * U) + (size_t)wave * (64 * U) + lane;
    unsigned long long x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      x[u] = t;
#pragma unroll
      for (int r = 0; r < NR; r++) x[u] ^= __builtin_nontemporal_load(P.in[r] + base + 64 * u);
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int w = 0; w < 10; w++) st<POL>(P.out[w] + base + 64 * u, x[u] + w);
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int NR, int POL> float Run(const Ptrs& P, size_t n, hipEvent_t a, hipEvent_t b) {
  std::vector<float> ms;
  for (int it = 0; it < 9; it++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((K<NR, POL>), dim3(256 * 8), dim3(256), 0, 0, P, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float m; CK(hipEventElapsedTime(&m, a, b)); if (it >= 2) ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}
int main() {
  const size_t n = 1u << 27;   // 1 GiB per stream
  Ptrs P;
  for (int i = 0; i < 4; i++) { void* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, i + 1, n * 8)); P.in[i] = (const unsigned long long*)p; }
  for (int i = 0; i < 10; i++) { void* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); P.out[i] = (unsigned long long*)p; }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  for (int rep = 0; rep < 3; rep++) {
#define R(POL) { float m4 = Run<4, POL>(P, n, a, b), m0 = Run<0, POL>(P, n, a, b); \
    printf("rep %d stores %-11s: 4 in / 10 out %.4f ms %.3f of 8 TB/s | 10 out only %.4f ms %.3f\n", rep, names[POL], m4, 14.0 * 8 * n / (m4 * 1e-3) / 8e12, m0, 10.0 * 8 * n / (m0 * 1e-3) / 8e12); }
    R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7)
  }
  return 0;
}
