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

__global__ void __launch_bounds__(kScanThreads)
ScanReduce(const uint32_t* __restrict__ counts, int64_t m, uint64_t* __restrict__ sums) {
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
ScanSpine(uint64_t* __restrict__ sums, int64_t nb, uint64_t* __restrict__ total_out) {
  const int64_t per = (nb + kScanThreads - 1) / kScanThreads;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = lo + per < nb ? lo + per : nb;
  uint64_t local = 0;
  for (int64_t i = lo; i < hi; i++) local += sums[i];
  uint64_t total;
  uint64_t prefix = BlockExclusiveScan(local, &total);
  for (int64_t i = lo; i < hi; i++) {
    uint64_t c = sums[i];
    sums[i] = prefix;
    prefix += c;
  }
  if (threadIdx.x == 0) *total_out = total;
}

__global__ void __launch_bounds__(kScanThreads)
ScanApply(const uint32_t* __restrict__ counts, int64_t m, const uint64_t* __restrict__ sums,
          uint64_t* __restrict__ offsets) {
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

// In-place inclusive scan of int32 lengths -> Arrow var-len offsets (data[i] becomes
// sum(data[0..i])); `sums` holds the exclusive prefix of every chunk (ScanSpine output).
__global__ void __launch_bounds__(kScanThreads)
ScanApplyInclusiveI32(int32_t* __restrict__ data, int64_t m, const uint64_t* __restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * kScanPerThread;
  uint32_t c[kScanPerThread];
  uint64_t local = 0;
#pragma unroll
  for (int i = 0; i < kScanPerThread; i++) {
    c[i] = (base + i < m) ? static_cast<uint32_t>(data[base + i]) : 0u;
    local += c[i];
  }
  uint64_t total;
  uint64_t prefix = BlockExclusiveScan(local, &total) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanPerThread; i++) {
    prefix += c[i];
    if (base + i < m) data[base + i] = static_cast<int32_t>(prefix);
  }
}

// One wavefront per wave tile (`subtiles` consecutive 64-row match words).
template <typename IndexT>
__global__ void __launch_bounds__(256)
EmitIndices(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ offsets,
            int64_t nwords, int subtiles, int64_t row_base, IndexT* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t ntiles = (nwords + subtiles - 1) / subtiles;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int64_t wt = (int64_t)blockIdx.x * 4 + wave; wt < ntiles; wt += (int64_t)gridDim.x * 4) {
    uint64_t off = offsets[wt];
    for (int u = 0; u < subtiles; u++) {
      const int64_t w = wt * subtiles + u;
      if (w >= nwords) break;
      const uint64_t m = mask[w];
      if ((m >> lane) & 1ull)
        out[off + __popcll(m & lt)] = static_cast<IndexT>(row_base + w * 64 + lane);
      off += __popcll(m);
    }
  }
}

// Serial-semantics reference check helper is NOT provided here on purpose: the CPU
// restatement lives in oracle/ only.

}  // namespace

int64_t ScanChunks(int64_t m) { return (m + kScanChunk - 1) / kScanChunk; }

hipError_t LaunchOffsetsScan(const uint32_t* counts, int64_t m, uint64_t* chunk_sums,
                             uint64_t* offsets, uint64_t* total, hipStream_t stream) {
  if (m <= 0) return hipMemsetAsync(total, 0, sizeof(uint64_t), stream);
  const int64_t nb = ScanChunks(m);
  hipLaunchKernelGGL(ScanReduce, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, counts, m,
                     chunk_sums);
  hipLaunchKernelGGL(ScanSpine, dim3(1), dim3(kScanThreads), 0, stream, chunk_sums, nb, total);
  hipLaunchKernelGGL(ScanApply, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, counts, m,
                     chunk_sums, offsets);
  return hipGetLastError();
}

hipError_t LaunchInclusiveScanI32(int32_t* data, int64_t m, uint64_t* chunk_sums, uint64_t* total,
                                  hipStream_t stream) {
  if (m <= 0) return hipMemsetAsync(total, 0, sizeof(uint64_t), stream);
  const int64_t nb = ScanChunks(m);
  hipLaunchKernelGGL(ScanReduce, dim3((unsigned)nb), dim3(kScanThreads), 0, stream,
                     reinterpret_cast<const uint32_t*>(data), m, chunk_sums);
  hipLaunchKernelGGL(ScanSpine, dim3(1), dim3(kScanThreads), 0, stream, chunk_sums, nb, total);
  hipLaunchKernelGGL(ScanApplyInclusiveI32, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, data,
                     m, chunk_sums);
  return hipGetLastError();
}

hipError_t LaunchEmitIndices(const uint64_t* mask, const uint64_t* offsets, int64_t nwords,
                             int subtiles, int64_t row_base, int index_bytes, void* out,
                             int num_cus, hipStream_t stream) {
  if (nwords <= 0) return hipSuccess;
  const int64_t ntiles = (nwords + subtiles - 1) / subtiles;
  int64_t grid = (ntiles + 3) / 4;
  const int64_t cap = (int64_t)num_cus * 8;
  if (grid > cap) grid = cap;
  switch (index_bytes) {
    case 2:
      hipLaunchKernelGGL(EmitIndices<uint16_t>, dim3((unsigned)grid), dim3(256), 0, stream, mask,
                         offsets, nwords, subtiles, row_base, static_cast<uint16_t*>(out));
      break;
    case 4:
      hipLaunchKernelGGL(EmitIndices<uint32_t>, dim3((unsigned)grid), dim3(256), 0, stream, mask,
                         offsets, nwords, subtiles, row_base, static_cast<uint32_t*>(out));
      break;
    default:
      hipLaunchKernelGGL(EmitIndices<uint64_t>, dim3((unsigned)grid), dim3(256), 0, stream, mask,
                         offsets, nwords, subtiles, row_base, static_cast<uint64_t*>(out));
      break;
  }
  return hipGetLastError();
}

}  // namespace gdv
