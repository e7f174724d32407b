// Round-6 experiment: does the ORDER in which a wave issues its stores matter for a 4-in / 10-out float64 stream set
// (the C2 shape)?  Sub-tile-major (what the projection kernels do: for each of the 16 sub-tiles, one 512-byte store to
// each of the ten outputs) against stream-major (for each output, its sixteen 512-byte stores back to back: an 8 KiB
// burst per stream per wave).  No arithmetic; same loads.  hipcc --offload-arch=gfx950 -O3 -o store_order_probe store_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
struct Ptrs { const unsigned long long* in[4]; unsigned long long* out[10]; };
template <int U, int ORDER, int NW>
__global__ void __launch_bounds__(256) K(const Ptrs P, size_t n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = n / (256 * U);
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const size_t base = t * (256 * U) + (size_t)wave * (64 * U) + lane;
    unsigned long long x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      x[u] = t;
#pragma unroll
      for (int r = 0; r < 4; r++) x[u] ^= __builtin_nontemporal_load(P.in[r] + base + 64 * u);
    }
    if (ORDER == 0) {
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int w = 0; w < NW; w++) __builtin_nontemporal_store(x[u] + w, P.out[w] + base + 64 * u);
    } else if (ORDER == 1) {
#pragma unroll
      for (int w = 0; w < NW; w++)
#pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(x[u] + w, P.out[w] + base + 64 * u);
    } else {  // groups of four sub-tiles per stream
#pragma unroll
      for (int g = 0; g < U; g += 4)
#pragma unroll
        for (int w = 0; w < NW; w++)
#pragma unroll
          for (int u = g; u < g + 4; u++) __builtin_nontemporal_store(x[u] + w, P.out[w] + base + 64 * u);
    }
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
  const size_t n = 1u << 24;
  Ptrs P;
  for (int i = 0; i < 4; i++) { void* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, i + 1, n * 8)); P.in[i] = (const unsigned long long*)p; }
  for (int i = 0; i < 10; i++) { void* p; CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); P.out[i] = (unsigned long long*)p; }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](int order, int grid) {
    std::vector<float> ms;
    for (int it = 0; it < 12; it++) {
      CK(hipEventRecord(a));
      if (order == 0) hipLaunchKernelGGL((K<16, 0, 10>), dim3(grid), dim3(256), 0, 0, P, n);
      if (order == 1) hipLaunchKernelGGL((K<16, 1, 10>), dim3(grid), dim3(256), 0, 0, P, n);
      if (order == 2) hipLaunchKernelGGL((K<16, 2, 10>), dim3(grid), dim3(256), 0, 0, P, n);
      if (order == 3) hipLaunchKernelGGL((K<4, 0, 10>), dim3(grid), dim3(256), 0, 0, P, n);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float m; CK(hipEventElapsedTime(&m, a, b)); if (it >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
  };
  const char* names[] = {"U=16 sub-tile-major", "U=16 stream-major  ", "U=16 groups of four", "U=4  sub-tile-major"};
  for (int rep = 0; rep < 4; rep++)
    for (int grid_per_cu : {4, 8})
      for (int o = 0; o < 4; o++) {
        const float m = run(o, 256 * grid_per_cu);
        printf("rep %d grid x%d %s: %.4f ms  %.3f of 8 TB/s\n", rep, grid_per_cu, names[o], m, 14.0 * 8 * n / (m * 1e-3) / 8e12);
      }
  return 0;
}
