// Ahead-of-time kernels that do not depend on the expression: the SelectionVector
// construction behind Filter::Evaluate.  Replaces the reference's
// SelectionVector::PopulateFromBitMap (§3.3 hot loop #3: a serial
// ctz / clear-lowest-bit walk over the result bitmap) with a wave-level stream compaction:
//
//   predicate kernel (generated)  : 64-bit match word per 64 rows via __ballot, one
//                                   selected-row count per wave tile
//   gdv_scan_* (here)             : exclusive prefix sum of the per-tile counts
//   gdv_emit_indices (here)       : lane i of a set bit writes row id at
//                                   tile_offset + popcount(word & lanemask_lt(i))
//
// Indices come out ascending by construction, exactly as the reference produces them.
#include <hip/hip_runtime.h>

#include "gdv_kernels.h"

#include <cstdlib>

namespace gdv {

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 16;
constexpr int kScanChunk = kScanThreads * kScanPerThread;  // counts per workgroup

__device__ __forceinline__ uint64_t WaveInclusiveScan(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Exclusive scan of one value per thread across a 256-thread workgroup; returns the
// exclusive prefix and leaves the workgroup total in *total.
__device__ __forceinline__ uint64_t BlockExclusiveScan(uint64_t v, uint64_t* total) {
  __shared__ uint64_t wave_sums[kScanThreads / 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint64_t incl = WaveInclusiveScan(v, lane);
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint64_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; w++) {
    uint64_t s = wave_sums[w];
    if (w < wave) base += s;
    sum += s;
  }
  __syncthreads();
  *total = sum;
  return base + incl - v;
}

struct ClosingOffsets { int32_t* p[kMaxScanSegments]; };

__global__ void __launch_bounds__(kScanThreads)
ScanReduce(const uint32_t* __restrict__ counts, int64_t m, uint64_t* __restrict__ sums,
           int64_t stride) {
  counts += (int64_t)blockIdx.y * stride;  // segment (one per var-len output; 0 for filters)
  sums += (int64_t)blockIdx.y * gridDim.x;
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanPerThread;
  uint64_t local = 0;
  if (base + kScanPerThread <= m) {
    const uint4* p = reinterpret_cast<const uint4*>(counts + base);
#pragma unroll
    for (int i = 0; i < kScanPerThread / 4; i++) {
      uint4 q = p[i];
      local += (uint64_t)q.x + q.y + q.z + q.w;
    }
  } else {
    for (int i = 0; i < kScanPerThread; i++)
      if (base + i < m) local += counts[base + i];
  }
  uint64_t total;
  (void)BlockExclusiveScan(local, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// Single workgroup: exclusive scan of the chunk sums in place; grand total -> *total.
__global__ void __launch_bounds__(kScanThreads)
ScanSpine(uint64_t* __restrict__ sums, int64_t nb, uint64_t* __restrict__ total_out,
          ClosingOffsets closing, const uint64_t* __restrict__ carry) {
  sums += (int64_t)blockIdx.x * nb;  // one workgroup per segment
  total_out += blockIdx.x;
  const int64_t per = (nb + kScanThreads - 1) / kScanThreads;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = lo + per < nb ? lo + per : nb;
  uint64_t local = 0;
  for (int64_t i = lo; i < hi; i++) local += sums[i];
  // (carry: what earlier chunks of a pipelined filter selected — the offsets come out global)
  const uint64_t carried = carry != nullptr ? carry[blockIdx.x] : 0ull;
  uint64_t total;
  uint64_t prefix = BlockExclusiveScan(local, &total) + carried;
  total += carried;
  for (int64_t i = lo; i < hi; i++) {
    uint64_t c = sums[i];
    sums[i] = prefix;
    prefix += c;
  }
  if (threadIdx.x == 0) {
    *total_out = total;
    // var-len outputs: the closing entry of the Arrow offsets buffer is the byte total
    if (closing.p[blockIdx.x] != nullptr) *closing.p[blockIdx.x] = static_cast<int32_t>(total);
  }
}

__global__ void __launch_bounds__(kScanThreads)
ScanApply(const uint32_t* __restrict__ counts, int64_t m, const uint64_t* __restrict__ sums,
          uint64_t* __restrict__ offsets, int64_t stride) {
  counts += (int64_t)blockIdx.y * stride;
  offsets += (int64_t)blockIdx.y * stride;
  sums += (int64_t)blockIdx.y * gridDim.x;
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanPerThread;
  uint32_t c[kScanPerThread];
  uint64_t local = 0;
  if (base + kScanPerThread <= m) {
    const uint4* p = reinterpret_cast<const uint4*>(counts + base);
#pragma unroll
    for (int i = 0; i < kScanPerThread / 4; i++) {
      uint4 q = p[i];
      c[4 * i] = q.x; c[4 * i + 1] = q.y; c[4 * i + 2] = q.z; c[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < kScanPerThread; i++) c[i] = (base + i < m) ? counts[base + i] : 0u;
  }
#pragma unroll
  for (int i = 0; i < kScanPerThread; i++) local += c[i];
  uint64_t total;
  uint64_t prefix = BlockExclusiveScan(local, &total) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanPerThread; i++) {
    if (base + i < m) offsets[base + i] = prefix;
    prefix += c[i];
  }
}

// The whole scan in one launch when a segment fits one workgroup's chunk (m <= 4096 counts,
// i.e. batches up to ~10^6 rows): small batches are launch-bound, two launches fewer matter.
__global__ void __launch_bounds__(kScanThreads)
ScanSmall(const uint32_t* __restrict__ counts, int64_t m, int64_t stride,
          uint64_t* __restrict__ offsets, uint64_t* __restrict__ total_out, ClosingOffsets closing,
          const uint64_t* __restrict__ carry) {
  counts += (int64_t)blockIdx.x * stride;
  offsets += (int64_t)blockIdx.x * stride;
  const int64_t base = (int64_t)threadIdx.x * kScanPerThread;
  uint32_t c[kScanPerThread];
  uint64_t local = 0;
#pragma unroll
  for (int i = 0; i < kScanPerThread; i++) {
    c[i] = (base + i < m) ? counts[base + i] : 0u;
    local += c[i];
  }
  const uint64_t carried = carry != nullptr ? carry[blockIdx.x] : 0ull;
  uint64_t total;
  uint64_t prefix = BlockExclusiveScan(local, &total) + carried;
  total += carried;
#pragma unroll
  for (int i = 0; i < kScanPerThread; i++) {
    if (base + i < m) offsets[base + i] = prefix;
    prefix += c[i];
  }
  if (threadIdx.x == 0) {
    total_out[blockIdx.x] = total;
    if (closing.p[blockIdx.x] != nullptr) *closing.p[blockIdx.x] = static_cast<int32_t>(total);
  }
}

// Index emission, LDS-staged.  One wavefront owns 64 consecutive match words (4096 rows):
// lane i takes word i, a wave-level exclusive scan of the popcounts gives every lane its
// slot range, the lane walks its word's set bits (ctz / clear-lowest) into the wave's
// private LDS window, and the wave then streams the window to HBM with fully coalesced
// stores.  (Writing straight from the bit walk would emit ~8 indices = 32 B per 64-lane
// store instruction at C3's selectivity: measured 0.82 ms for 10^9 rows vs the ~0.15 ms
// the 0.66 GB of traffic costs.)  `offsets` holds the selected-row count before every group
// of `subtiles` words (subtiles divides 64), so the wave's base is offsets[first group].
constexpr int kEmitWords = 64;  // (launcher default; the kernel is a template on it: GDV_EMIT_WORDS=32 for experiments)

template <typename IndexT, int kWords>
__global__ void __launch_bounds__(256)
EmitIndices(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ offsets,
            int64_t nwords, int subtiles, int64_t row_base, IndexT* __restrict__ out) {
  // positions inside the wave's 4096-row tile fit 12 bits: staging them as uint16 keeps the
  // window at 8 KiB per wave (32 KiB per workgroup, 5 workgroups per CU) whatever IndexT is
  __shared__ uint16_t stage[4][kWords * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint16_t* buf = stage[wave];
  const int64_t ntiles = (nwords + kWords - 1) / kWords;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  // the next tile's mask word and base are loaded while the current tile is walked
  uint64_t m_next = 0, base_next = 0;
  if (t < ntiles) {
    const int64_t w = t * kWords + lane;
    m_next = (lane < kWords && w < nwords) ? mask[w] : 0ull;
    base_next = offsets[(t * kWords) / subtiles];
  }
  for (; t < ntiles; t += stride) {
    uint64_t m = m_next;
    const uint64_t base = base_next;
    if (t + stride < ntiles) {
      const int64_t w = (t + stride) * kWords + lane;
      m_next = (lane < kWords && w < nwords) ? mask[w] : 0ull;
      base_next = offsets[((t + stride) * kWords) / subtiles];
    }
    const uint32_t c = (uint32_t)__popcll(m);
    const uint32_t incl = (uint32_t)WaveInclusiveScan(c, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t slot = incl - c;
    const int64_t tile_row0 = row_base + t * (int64_t)(kWords * 64);
    while (m) {
      buf[slot++] = static_cast<uint16_t>((lane << 6) + __builtin_ctzll(m));
      m &= m - 1;
    }
    __builtin_amdgcn_wave_barrier();
    // four indices per lane per step: one 8-byte LDS read, one 4*sizeof(IndexT) store
    IndexT* dst = out + base;
    for (uint32_t j = (uint32_t)lane * 4; j < total; j += 256) {
      if (j + 4 <= total) {
        uint16_t q[4];
        __builtin_memcpy(q, buf + j, 8);
        IndexT v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = static_cast<IndexT>(tile_row0 + q[k]);
        __builtin_memcpy(dst + j, v, sizeof(v));
      } else {
        for (uint32_t k = j; k < total; k++) dst[k] = static_cast<IndexT>(tile_row0 + buf[k]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Serial-semantics reference check helper is NOT provided here on purpose: the CPU
// restatement lives in oracle/ only.

// ---- what plain streaming kernels reach on THIS box, right now (round 3: the same binary measured
// 4.84 .. 6.20 ms on C2 across boxes of the pool; a bench line that also carries the box's own
// read / write / copy ceilings can be compared across boxes).  16 B per lane, 8 in flight,
// non-temporal, grid-stride: the shape of tools/hbm_ceiling.hip.
typedef unsigned long long CeilU64x2 __attribute__((ext_vector_type(2)));
constexpr int kCeilU = 8;
__global__ void __launch_bounds__(256) CeilRead(const CeilU64x2* __restrict__ p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  size_t i = (size_t)blockIdx.x * 256 * kCeilU + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * kCeilU;
  for (; i + 256 * (kCeilU - 1) < n; i += stride) {
    CeilU64x2 v[kCeilU];
#pragma unroll
    for (int u = 0; u < kCeilU; u++) v[u] = __builtin_nontemporal_load(p + i + 256 * u);
#pragma unroll
    for (int u = 0; u < kCeilU; u++) acc += v[u].x ^ v[u].y;
  }
  if (acc == 0x1234567) out[0] = acc;
}
__global__ void __launch_bounds__(256) CeilWrite(CeilU64x2* __restrict__ p, size_t n, unsigned long long val) {
  size_t i = (size_t)blockIdx.x * 256 * kCeilU + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * kCeilU;
  CeilU64x2 v;
  v.x = val;
  v.y = val + 1;
  for (; i + 256 * (kCeilU - 1) < n; i += stride) {
#pragma unroll
    for (int u = 0; u < kCeilU; u++) __builtin_nontemporal_store(v, p + i + 256 * u);
  }
}
__global__ void __launch_bounds__(256) CeilCopy(const CeilU64x2* __restrict__ a, CeilU64x2* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * kCeilU + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * kCeilU;
  for (; i + 256 * (kCeilU - 1) < n; i += stride) {
    CeilU64x2 v[kCeilU];
#pragma unroll
    for (int u = 0; u < kCeilU; u++) v[u] = __builtin_nontemporal_load(a + i + 256 * u);
#pragma unroll
    for (int u = 0; u < kCeilU; u++) __builtin_nontemporal_store(v[u], b + i + 256 * u);
  }
}

// Round 4: the ceiling for a workload's OWN traffic shape.  The three kernels above drive one (or two)
// 16-byte streams; the round-3 driver run had the product's C2 kernel — four input streams, ten output
// streams — moving MORE bytes per second than the mix those ceilings predicted (frac 1.057: an
// instrument that under-drives HBM says nothing about slow box vs slow kernel).  CeilStream is the
// projection kernel's skeleton with the arithmetic taken out: NR input streams and NW output streams
// of 8-byte elements, one row per lane, GDV_U = 4 sub-tiles per wave, 4 waves per workgroup, every load
// of the tile in flight before the first store, grid-stride.  MeasureStreamCeiling sweeps the grid
// (2 .. 32 workgroups per CU) x {plain, non-temporal} and reports the best: by construction at least
// what any kernel of that shape reaches.
struct CeilPointers {
  const unsigned long long* in[10];
  unsigned long long* out[10];
};
// (U sub-tiles per wave: 4 = the projection kernels' shape of rounds 1-5, 16 = round 6's for elements of up to 8 bytes)
template <int NR, int NW, bool NT, int U = 4>
__global__ void __launch_bounds__(256) CeilStream(const CeilPointers P, size_t n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = n / (256 * U);
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const size_t base = t * (256 * U) + (size_t)wave * (64 * U) + lane;
    unsigned long long v[NR > 0 ? NR : 1][U];
    unsigned long long acc = 0;
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int r = 0; r < NR; r++)
        v[r][u] = NT ? __builtin_nontemporal_load(P.in[r] + base + 64 * u) : P.in[r][base + 64 * u];
#pragma unroll
    for (int u = 0; u < U; u++) {
      unsigned long long x = t;
#pragma unroll
      for (int r = 0; r < NR; r++) x ^= v[r][u];
      acc |= x;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        if (NT) __builtin_nontemporal_store(x + w, P.out[w] + base + 64 * u);
        else P.out[w][base + 64 * u] = x + w;
      }
    }
    if (NW == 0 && acc == 0x12345678abcdefull) P.out[0][0] = acc;  // (keeps the loads of a read-only shape alive)
  }
}

}  // namespace

namespace {
template <int NR, int NW>
hipError_t StreamCeilingFor(const CeilPointers& P, size_t n, int num_cus, double* best_gbs, int* best_grid, int* best_nt) {
  hipEvent_t e0, e1;
  hipError_t err = hipEventCreate(&e0);
  if (err != hipSuccess) return err;
  err = hipEventCreate(&e1);
  if (err != hipSuccess) { (void)hipEventDestroy(e0); return err; }
  const double moved = (double)(NR + NW) * 8.0 * (double)(n / 1024 * 1024);
  // (per_cu > 100: the 16-sub-tile shape, workgroups per CU = per_cu - 100 — one more axis of the sweep, kept in the one integer
  // the callers report)
  auto time_once = [&](int per_cu, int nt) -> float {
    const bool wide = per_cu > 100;
    const int grid = num_cus * (wide ? per_cu - 100 : per_cu);
    if (err == hipSuccess) err = hipEventRecord(e0, nullptr);
    if (wide && nt) hipLaunchKernelGGL((CeilStream<NR, NW, true, 16>), dim3(grid), dim3(256), 0, nullptr, P, n);
    else if (wide) hipLaunchKernelGGL((CeilStream<NR, NW, false, 16>), dim3(grid), dim3(256), 0, nullptr, P, n);
    else if (nt) hipLaunchKernelGGL((CeilStream<NR, NW, true>), dim3(grid), dim3(256), 0, nullptr, P, n);
    else hipLaunchKernelGGL((CeilStream<NR, NW, false>), dim3(grid), dim3(256), 0, nullptr, P, n);
    if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 1e30f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    return ms;
  };
  // coarse pass: every (grid, nt) once after one warm-up launch; then the two best configurations
  // are timed six more times each and the minimum taken — a single short launch on a GPU whose clocks
  // are still moving is not a measurement (round 4: 5.3 .. 6.2 TB/s for the same configuration within
  // one second on one box)
  struct Cfg { int per_cu, nt; float ms; };
  Cfg cfgs[20];
  int nc = 0;
  (void)time_once(8, 0);
  for (int nt = 0; nt < 2; nt++)
    for (int per_cu = 2; per_cu <= 32; per_cu *= 2) cfgs[nc++] = Cfg{per_cu, nt, time_once(per_cu, nt)};
  if (NR * 16 * 2 <= 160)  // (the wide shape keeps NR x 16 eight-byte values in registers)
    for (int nt = 0; nt < 2; nt++)
      for (int per_cu = 2; per_cu <= 32; per_cu *= 2) cfgs[nc++] = Cfg{100 + per_cu, nt, time_once(100 + per_cu, nt)};
  for (int i = 0; i < nc; i++)
    for (int j = i + 1; j < nc; j++)
      if (cfgs[j].ms < cfgs[i].ms) { const Cfg t = cfgs[i]; cfgs[i] = cfgs[j]; cfgs[j] = t; }
  float best = 1e30f;
  for (int c = 0; c < 2 && err == hipSuccess; c++) {
    for (int it = 0; it < 6; it++) {
      const float ms = time_once(cfgs[c].per_cu, cfgs[c].nt);
      if (ms < best) { best = ms; *best_grid = cfgs[c].per_cu; *best_nt = cfgs[c].nt; }
    }
  }
  *best_gbs = moved / (best * 1e-3) / 1e9;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return err != hipSuccess ? err : hipGetLastError();
}
}  // namespace

// streams[0 .. nr) are read, streams[nr .. nr + nw) written; every stream holds `elems` 8-byte elements.
// Shapes: (nr, nw) in {(1,0) (2,0) (4,0) (0,1) (0,4) (0,10) (2,1) (3,1) (4,10) (7,5) (2,3)}.
hipError_t MeasureStreamCeiling(void* const* streams, int nr, int nw, size_t elems, int num_cus, double* gbs,
                                int* workgroups_per_cu, int* nontemporal) {
  CeilPointers P;
  for (int i = 0; i < 10; i++) { P.in[i] = nullptr; P.out[i] = nullptr; }
  for (int i = 0; i < nr; i++) P.in[i] = static_cast<const unsigned long long*>(streams[i]);
  for (int i = 0; i < nw; i++) P.out[i] = static_cast<unsigned long long*>(streams[nr + i]);
  if (nw == 0) P.out[0] = const_cast<unsigned long long*>(P.in[0]);  // never written (see the kernel)
#define GDV_CEIL_SHAPE(R, W) if (nr == R && nw == W) return StreamCeilingFor<R, W>(P, elems, num_cus, gbs, workgroups_per_cu, nontemporal)
  GDV_CEIL_SHAPE(1, 0); GDV_CEIL_SHAPE(2, 0); GDV_CEIL_SHAPE(4, 0);
  GDV_CEIL_SHAPE(0, 1); GDV_CEIL_SHAPE(0, 4); GDV_CEIL_SHAPE(0, 10);
  GDV_CEIL_SHAPE(2, 1); GDV_CEIL_SHAPE(3, 1); GDV_CEIL_SHAPE(4, 10); GDV_CEIL_SHAPE(7, 5); GDV_CEIL_SHAPE(2, 3);
#undef GDV_CEIL_SHAPE
  return hipErrorInvalidValue;
}

// ---- placement probe of the device pool (round 6).  Where the driver puts a set of buffers that are streamed together
// decides what a projection over them runs at (profiles/r05_box_states.txt: 4.9 .. 7.0 ms for the same C2 kernel), and a
// non-temporal WRITE sweep over the set — every buffer at the same offset at the same time, as the projection kernel
// writes its output columns — predicts it (profiles/r06_placement_probe.txt: rank correlation 0.98 with the kernel's time).
struct ProbePointers { unsigned long long* p[32]; };
__global__ void __launch_bounds__(256) ProbeWriteSet(const ProbePointers P, int count, size_t elems) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = elems / (256 * U);
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const size_t base = t * (256 * U) + (size_t)wave * (64 * U) + lane;
#pragma unroll
    for (int u = 0; u < U; u++)
      for (int w = 0; w < count; w++) __builtin_nontemporal_store((unsigned long long)(t + w), P.p[w] + base + 64 * u);
  }
}
hipError_t MeasureWriteSet(void* const* bufs, int count, size_t bytes_each, int num_cus, double* gbs) {
  if (count < 1 || count > 32) return hipErrorInvalidValue;
  ProbePointers P;
  for (int i = 0; i < 32; i++) P.p[i] = i < count ? static_cast<unsigned long long*>(bufs[i]) : nullptr;
  const size_t elems = bytes_each / 8;
  hipEvent_t e0, e1;
  hipError_t err = hipEventCreate(&e0);
  if (err != hipSuccess) return err;
  err = hipEventCreate(&e1);
  if (err != hipSuccess) { (void)hipEventDestroy(e0); return err; }
  const int grid = num_cus * 8;
  float best = 1e30f;
  for (int rep = 0; rep < 4 && err == hipSuccess; rep++) {  // the first launch warms up; the minimum of the others counts
    err = hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(ProbeWriteSet, dim3(grid), dim3(256), 0, nullptr, P, count, elems);
    if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 1e30f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (err == hipSuccess) *gbs = (double)count * (double)(elems / 1024 * 1024) * 8.0 / (best * 1e-3) / 1e9;
  return err;
}

int64_t ScanChunks(int64_t m) { return (m + kScanChunk - 1) / kScanChunk; }

namespace {
hipError_t ScanImpl(const uint32_t* counts, int64_t m, int64_t stride, int nseg, uint64_t* chunk_sums,
                    uint64_t* offsets, uint64_t* totals, int32_t* const* closing, const uint64_t* carry,
                    hipStream_t stream) {
  if (nseg <= 0) return hipSuccess;
  if (nseg > kMaxScanSegments) return hipErrorInvalidValue;
  ClosingOffsets c;
  for (int i = 0; i < kMaxScanSegments; i++) c.p[i] = (closing != nullptr && i < nseg) ? closing[i] : nullptr;
  if (m <= 0) {
    hipError_t e = carry != nullptr
                       ? hipMemcpyAsync(totals, carry, sizeof(uint64_t) * nseg, hipMemcpyDeviceToDevice, stream)
                       : hipMemsetAsync(totals, 0, sizeof(uint64_t) * nseg, stream);
    for (int i = 0; e == hipSuccess && i < nseg; i++)
      if (c.p[i] != nullptr) e = hipMemsetAsync(c.p[i], 0, sizeof(int32_t), stream);
    return e;
  }
  const int64_t nb = ScanChunks(m);
  if (nb == 1) {
    hipLaunchKernelGGL(ScanSmall, dim3((unsigned)nseg), dim3(kScanThreads), 0, stream, counts, m, stride,
                       offsets, totals, c, carry);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(ScanReduce, dim3((unsigned)nb, (unsigned)nseg), dim3(kScanThreads), 0, stream,
                     counts, m, chunk_sums, stride);
  hipLaunchKernelGGL(ScanSpine, dim3((unsigned)nseg), dim3(kScanThreads), 0, stream, chunk_sums, nb,
                     totals, c, carry);
  hipLaunchKernelGGL(ScanApply, dim3((unsigned)nb, (unsigned)nseg), dim3(kScanThreads), 0, stream,
                     counts, m, chunk_sums, offsets, stride);
  return hipGetLastError();
}

}  // namespace

hipError_t LaunchSegmentedOffsetsScan(const uint32_t* counts, int64_t m, int64_t stride, int nseg,
                                      uint64_t* chunk_sums, uint64_t* offsets, uint64_t* totals,
                                      int32_t* const* closing, hipStream_t stream) {
  return ScanImpl(counts, m, stride, nseg, chunk_sums, offsets, totals, closing, nullptr, stream);
}

hipError_t LaunchOffsetsScan(const uint32_t* counts, int64_t m, uint64_t* chunk_sums,
                             uint64_t* offsets, uint64_t* total, hipStream_t stream, const uint64_t* carry) {
  return ScanImpl(counts, m, m, 1, chunk_sums, offsets, total, nullptr, carry, stream);
}

hipError_t MeasureHbmCeilings(void* a, void* b, size_t bytes, int grid, double* read_gbs, double* write_gbs,
                              double* copy_gbs) {
  const size_t n = bytes / 16;
  hipEvent_t e0, e1;
  hipError_t err = hipEventCreate(&e0);
  if (err != hipSuccess) return err;
  err = hipEventCreate(&e1);
  if (err != hipSuccess) { (void)hipEventDestroy(e0); return err; }
  auto best_of = [&](auto&& launch, double moved, double* out) {
    float best = 1e30f;
    for (int it = 0; it < 5 && err == hipSuccess; it++) {
      err = hipEventRecord(e0, nullptr);
      launch();
      if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
      if (err == hipSuccess) err = hipEventSynchronize(e1);
      float ms = 0;
      if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
      if (it >= 1 && ms < best) best = ms;
    }
    *out = moved / (best * 1e-3) / 1e9;
  };
  best_of([&] { hipLaunchKernelGGL(CeilRead, dim3(grid), dim3(256), 0, nullptr, static_cast<const CeilU64x2*>(a), n,
                                   static_cast<unsigned long long*>(b)); }, (double)bytes, read_gbs);
  best_of([&] { hipLaunchKernelGGL(CeilWrite, dim3(grid), dim3(256), 0, nullptr, static_cast<CeilU64x2*>(b), n, 7ull); },
          (double)bytes, write_gbs);
  best_of([&] { hipLaunchKernelGGL(CeilCopy, dim3(grid), dim3(256), 0, nullptr, static_cast<const CeilU64x2*>(a),
                                   static_cast<CeilU64x2*>(b), n); }, 2.0 * bytes, copy_gbs);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return err != hipSuccess ? err : hipGetLastError();
}

hipError_t LaunchEmitIndices(const uint64_t* mask, const uint64_t* offsets, int64_t nwords,
                             int subtiles, int64_t row_base, int index_bytes, void* out,
                             int num_cus, hipStream_t stream) {
  if (nwords <= 0) return hipSuccess;
  // experiments (read once): words per wave tile (64, or 32: half the LDS window, twice the workgroups per CU) and the
  // grid cap in workgroups per CU
  static const int words = [] { const char* e = std::getenv("GDV_EMIT_WORDS"); return e != nullptr && atoi(e) == 32 ? 32 : kEmitWords; }();
  static const int per_cu = [] { const char* e = std::getenv("GDV_EMIT_WGS_PER_CU"); return e != nullptr && atoi(e) > 0 ? atoi(e) : 0; }();
  if (subtiles <= 0 || words % subtiles != 0) return hipErrorInvalidValue;
  const int64_t ntiles = (nwords + words - 1) / words;
  int64_t grid = (ntiles + 3) / 4;
  // LDS would allow 5 workgroups per CU at 64 words (8 at 32); FOUR measured best at 10^9 rows, 1/8 selected, one box:
  // 3 / 4 / 5 per CU: 0.218 / 0.189 / 0.232 ms; 32-word tiles 0.216-0.264 (profiles/r05_emit_indices_experiments.txt)
  const int64_t cap = (int64_t)num_cus * (per_cu > 0 ? per_cu : 4);
  if (grid > cap) grid = cap;
#define GDV_EMIT(T, W) hipLaunchKernelGGL((EmitIndices<T, W>), dim3((unsigned)grid), dim3(256), 0, stream, mask, offsets, nwords, subtiles, row_base, static_cast<T*>(out))
  if (words == 32) {
    switch (index_bytes) { case 2: GDV_EMIT(uint16_t, 32); break; case 4: GDV_EMIT(uint32_t, 32); break; default: GDV_EMIT(uint64_t, 32); break; }
  } else {
    switch (index_bytes) { case 2: GDV_EMIT(uint16_t, 64); break; case 4: GDV_EMIT(uint32_t, 64); break; default: GDV_EMIT(uint64_t, 64); break; }
  }
#undef GDV_EMIT
  return hipGetLastError();
}

// ---- the gate between the two stages of an asynchronous two-stage plan (gdv_kernels.h)
__global__ void StageGate(const uint64_t* __restrict__ stage_result, int num_outputs, StageCaps caps,
                          const int64_t* __restrict__ rows_in, int64_t rows, int64_t* __restrict__ rows_out,
                          uint64_t* __restrict__ status_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t bad = stage_result[0] & ~uint64_t{64};   // (64: the exact string kernels' "saw UTF-8" note)
  for (int e = 0; e < num_outputs; e++) {
    const int64_t total = static_cast<int64_t>(stage_result[1 + e]);
    if (total > caps.cap[e]) {
      bad |= kStageOverflow;
    } else if (caps.data[e] != nullptr) {
      char* end = static_cast<char*>(caps.data[e]) + total;
      for (int i = 0; i < 16; i++) end[i] = 0;
    }
  }
  int64_t n = rows_in != nullptr ? *rows_in : rows;
  if (n < 0) n = 0;
  if (n > rows) n = rows;
  *rows_out = bad != 0 ? 0 : n;
  *status_out = bad;
}
__global__ void OrStatus(uint64_t* __restrict__ result, const uint64_t* __restrict__ status) {
  if (threadIdx.x == 0 && blockIdx.x == 0) result[0] |= status[0];
}
hipError_t LaunchStageGate(const uint64_t* stage_result, int num_outputs, const StageCaps& caps, const int64_t* rows_in,
                           int64_t rows, int64_t* rows_out, uint64_t* status_out, hipStream_t stream) {
  if (num_outputs < 0 || num_outputs > kMaxStageOutputs) return hipErrorInvalidValue;
  hipLaunchKernelGGL(StageGate, dim3(1), dim3(64), 0, stream, stage_result, num_outputs, caps, rows_in, rows, rows_out, status_out);
  return hipGetLastError();
}
__global__ void PublishStatus(uint64_t* __restrict__ result, const uint32_t* __restrict__ err, uint32_t clear) {
  if (threadIdx.x == 0 && blockIdx.x == 0) result[0] = err[0] & ~clear;
}
hipError_t LaunchPublishStatus(uint64_t* result, const uint32_t* err, uint32_t clear, hipStream_t stream) {
  hipLaunchKernelGGL(PublishStatus, dim3(1), dim3(64), 0, stream, result, err, clear);
  return hipGetLastError();
}
// `share` (optional) receives the selected rows per 1024 of THIS batch — count and row number of one launch, so that a reader
// never pairs the count of one batch with the rows of another (-1: the launch did not complete)
__global__ void PublishCount(int64_t* __restrict__ dst, int64_t* __restrict__ share, const int64_t* __restrict__ count,
                             const uint32_t* __restrict__ err, uint32_t fatal, int64_t rows) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t v = (err[0] & fatal) != 0 ? int64_t{-1} : count[0];
    if (dst != nullptr) dst[0] = v;
    if (share != nullptr) share[0] = v < 0 || rows <= 0 ? int64_t{-1} : v * 1024 / rows;
  }
}
hipError_t LaunchPublishCount(int64_t* dst, const int64_t* count, const uint32_t* err, uint32_t fatal, hipStream_t stream,
                              int64_t* share, int64_t rows) {
  hipLaunchKernelGGL(PublishCount, dim3(1), dim3(64), 0, stream, dst, share, count, err, fatal, rows);
  return hipGetLastError();
}
hipError_t LaunchOrStatus(uint64_t* result, const uint64_t* status, hipStream_t stream) {
  hipLaunchKernelGGL(OrStatus, dim3(1), dim3(64), 0, stream, result, status);
  return hipGetLastError();
}

}  // namespace gdv
