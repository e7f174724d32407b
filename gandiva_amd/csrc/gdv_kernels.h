// Launchers of the ahead-of-time kernels in gdv_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gdv {

// Number of scan workgroups (= entries of `chunk_sums`) for m counts.
int64_t ScanChunks(int64_t m);

// offsets[i] = sum(counts[0..i)) for i < m; *total = sum of all counts.
hipError_t LaunchOffsetsScan(const uint32_t* counts, int64_t m, uint64_t* chunk_sums,
                             uint64_t* offsets, uint64_t* total, hipStream_t stream);

// data[i] = sum(data[0..i]) in place (int32 lengths -> var-len offsets); *total = grand sum.
hipError_t LaunchInclusiveScanI32(int32_t* data, int64_t m, uint64_t* chunk_sums, uint64_t* total,
                                  hipStream_t stream);

// Writes row_base + (position of every set bit of mask[0..nwords)) in ascending order to
// out[]; offsets[] holds, per group of `subtiles` words, the number of set bits before it.
hipError_t LaunchEmitIndices(const uint64_t* mask, const uint64_t* offsets, int64_t nwords,
                             int subtiles, int64_t row_base, int index_bytes, void* out,
                             int num_cus, hipStream_t stream);

}  // namespace gdv
