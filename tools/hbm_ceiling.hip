// What a plain streaming kernel reaches on this box: read-only, write-only and copy rates at the
// sizes of the BASELINE workloads.  Context for the roofline fractions in DESIGN.md §4 (the
// microarch guide quotes 6.29 TB/s for a float4 copy).  hipcc --offload-arch=gfx950 -O3 -o hbm_ceiling hbm_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <int U>
__global__ void __launch_bounds__(256) k_read(const u64x2* __restrict__ p, size_t n, u64* out) {
  u64 acc = 0;
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + 256 * (U - 1) < n; i += stride) {
    u64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p + i + 256 * u);
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y;
  }
  if (acc == 0x1234567) out[0] = acc;
}
template <int U>
__global__ void __launch_bounds__(256) k_read8(const u64* __restrict__ p, size_t n, u64* out) {  // 8 B per lane
  u64 acc = 0;
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + 256 * (U - 1) < n; i += stride) {
    u64 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p + i + 256 * u);
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  if (acc == 0x1234567) out[0] = acc;
}
// two columns, 8 B per lane each, one wave = 64 consecutive rows per sub-tile (the filter's shape)
template <int U>
__global__ void __launch_bounds__(256) k_read8x2(const u64* __restrict__ p, const u64* __restrict__ q, size_t n, u64* out) {
  u64 acc = 0;
  size_t i = (size_t)blockIdx.x * 256 * U + (threadIdx.x >> 6) * 64 * U + (threadIdx.x & 63);
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + 64 * (U - 1) < n; i += stride) {
    u64 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = __builtin_nontemporal_load(p + i + 64 * u); w[u] = __builtin_nontemporal_load(q + i + 64 * u); }
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u] ^ w[u];
  }
  if (acc == 0x1234567) out[0] = acc;
}
// the filter's predicate kernel, rebuilt step by step on top of the bare access shape:
//   STEP 1: compare + ballot + match-word store + per-tile count     STEP 2: + validity-word loads
template <int U, int STEP>
__global__ void __launch_bounds__(256) k_pred(const long long* __restrict__ p, const long long* __restrict__ q, size_t n,
                                              const u64* __restrict__ ones, u64* __restrict__ mask, unsigned* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const size_t nwt = n / (64 * U);
  for (size_t wt = (size_t)blockIdx.x * 4 + wave; wt < nwt; wt += (size_t)gridDim.x * 4) {
    const size_t rbase = wt * 64 * U;
    long long a[U], b[U];
    u64 vw0 = ~0ull, vw1 = ~0ull;
    if (STEP >= 2) {  // one vector load per column: lane u <-> word u (clamped to the one all-ones word)
      const u64 lo0 = ones[0], hi0 = ones[0], lo1 = ones[0], hi1 = ones[0];
      vw0 = (lo0 >> 0) | ((hi0 << 1) << 63);
      vw1 = (lo1 >> 0) | ((hi1 << 1) << 63);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      a[u] = __builtin_nontemporal_load(p + rbase + 64 * u + lane);
      b[u] = __builtin_nontemporal_load(q + rbase + 64 * u + lane);
    }
    u64 acc = 0;
    unsigned cnt = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      bool ok = a[u] > 499 && b[u] < 250;
      if (STEP >= 2) {
        const u64 v0 = ((u64)(unsigned)__builtin_amdgcn_readlane((unsigned)(vw0 >> 32), u) << 32) | (unsigned)__builtin_amdgcn_readlane((unsigned)vw0, u);
        const u64 v1 = ((u64)(unsigned)__builtin_amdgcn_readlane((unsigned)(vw1 >> 32), u) << 32) | (unsigned)__builtin_amdgcn_readlane((unsigned)vw1, u);
        ok = ok && ((v0 >> lane) & 1) && ((v1 >> lane) & 1);
      }
      const u64 fm = __ballot(ok);
      cnt += __popcll(fm);
      acc = lane == u ? fm : acc;
    }
    if (lane < U) mask[wt * U + lane] = acc;
    if (lane == 0) counts[wt] = cnt;
  }
}
template <int U>
__global__ void __launch_bounds__(256) k_write(u64x2* __restrict__ p, size_t n, u64 val) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  u64x2 v; v.x = val; v.y = val + 1;
  for (; i + 256 * (U - 1) < n; i += stride) {
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(v, p + i + 256 * u);
  }
}
template <int U>
__global__ void __launch_bounds__(256) k_copy(const u64x2* __restrict__ a, u64x2* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + 256 * (U - 1) < n; i += stride) {
    u64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(a + i + 256 * u);
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], b + i + 256 * u);
  }
}
template <typename F>
double timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 7; it++) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2 && ms < best) best = ms;
  }
  return best;
}
int main() {
  const size_t bytes = (size_t)16 << 30, n = bytes / 16;
  u64x2 *a, *b; u64* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  u64 *ones, *mask; unsigned* counts;
  CK(hipMalloc(&ones, 256)); CK(hipMemset(ones, 0xff, 256));
  CK(hipMalloc(&mask, bytes / 64 + 1024)); CK(hipMalloc(&counts, bytes / 8 / 1024 * 4 + 1024));
  for (int g : {2048}) {
    const size_t rows = (size_t)1000000000;
    double s1 = timeit([&] { hipLaunchKernelGGL((k_pred<16, 1>), dim3(g), dim3(256), 0, 0, (const long long*)a, (const long long*)b, rows, ones, mask, counts); });
    double s2 = timeit([&] { hipLaunchKernelGGL((k_pred<16, 2>), dim3(g), dim3(256), 0, 0, (const long long*)a, (const long long*)b, rows, ones, mask, counts); });
    printf("predicate kernel rebuilt by hand, 10^9 rows x 2 int64 columns (16 GB): compare+ballot+stores %.3f ms = %.2f TB/s | + validity-word loads %.3f ms = %.2f TB/s\n",
           s1, 16e9 / s1 / 1e9 * 1e-3 * 1e3, s2, 16e9 / s2 / 1e9 * 1e-3 * 1e3);
  }
  for (int g : {2048, 4096, 8192}) {
    double r = timeit([&] { hipLaunchKernelGGL((k_read<8>), dim3(g), dim3(256), 0, 0, a, n, out); });
    double w = timeit([&] { hipLaunchKernelGGL((k_write<8>), dim3(g), dim3(256), 0, 0, b, n, 7ull); });
    double c = timeit([&] { hipLaunchKernelGGL((k_copy<8>), dim3(g), dim3(256), 0, 0, a, b, n); });
    double r8 = timeit([&] { hipLaunchKernelGGL((k_read8<16>), dim3(g), dim3(256), 0, 0, (const u64*)a, n * 2, out); });
    double r82 = timeit([&] { hipLaunchKernelGGL((k_read8x2<16>), dim3(g), dim3(256), 0, 0, (const u64*)a, (const u64*)b, n * 2, out); });
    printf("grid %5d: 8 B/lane x16 read %.3f ms = %.2f TB/s | two columns 8 B/lane x16, wave-contiguous %.3f ms = %.2f TB/s\n", g, r8, bytes / r8 / 1e9, r82, 2.0 * bytes / r82 / 1e9);
    printf("grid %5d x 256, 8 x 16 B per lane in flight, 16 GiB: read %.3f ms = %.2f TB/s | write %.3f ms = %.2f TB/s | copy %.3f ms = %.2f TB/s (read+write bytes)\n",
           g, r, bytes / r / 1e9, w, bytes / w / 1e9, c, 2.0 * bytes / c / 1e9);
  }
  return 0;
}
