// Launchers of the ahead-of-time kernels in gdv_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gdv {

// Number of scan workgroups (= entries of `chunk_sums`) for m counts.
// placement probe of the device pool (round 6): GB/s of a non-temporal write sweep over `count` buffers of bytes_each bytes,
// all written at the same offsets at the same time (minimum of three timed launches after one warm-up; default stream)
hipError_t MeasureWriteSet(void* const* bufs, int count, size_t bytes_each, int num_cus, double* gbs);

int64_t ScanChunks(int64_t m);

// offsets[i] = sum(counts[0..i)) for i < m; *total = sum of all counts.
// carry (device pointer, may be null): added to every offset and to *total — a pipelined filter
// scans chunk k with carry = the running total after chunk k - 1, so the offsets come out global.
hipError_t LaunchOffsetsScan(const uint32_t* counts, int64_t m, uint64_t* chunk_sums,
                             uint64_t* offsets, uint64_t* total, hipStream_t stream,
                             const uint64_t* carry = nullptr);

// The same for `nseg` (<= 16) independent segments in one set of launches: segment s reads
// counts[s*stride .. s*stride+m), writes offsets[s*stride ..), totals[s], and — when
// closing[s] is not null — *closing[s] = (int32) totals[s] (the closing entry of an Arrow
// var-len offsets buffer).  chunk_sums needs nseg * ScanChunks(m) entries.
constexpr int kMaxScanSegments = 16;
hipError_t LaunchSegmentedOffsetsScan(const uint32_t* counts, int64_t m, int64_t stride, int nseg,
                                      uint64_t* chunk_sums, uint64_t* offsets, uint64_t* totals,
                                      int32_t* const* closing, hipStream_t stream);

// Writes row_base + (position of every set bit of mask[0..nwords)) in ascending order to
// out[]; offsets[] holds, per group of `subtiles` words, the number of set bits before it.
// Asynchronous two-stage plans (round 4): the gate between the stages.  stage_result = the first stage's
// { status, bytes of output 0, ... } block (gdv_projector_evaluate_async); caps = the capacities of the
// temporaries.  *rows_out = the rows the second stage may process: rows_in (device word, may be null: then
// `rows`) when the first stage completed and every temporary fits, 0 otherwise; *status_out = the first
// stage's error bits (its SAWUTF8 note dropped) | kStageOverflow when a temporary was too small.
constexpr uint64_t kStageOverflow = 128;
constexpr int kMaxStageOutputs = 8;
// (data[e]: the temporary's byte buffer, cap[e] + 16 bytes long — the gate zeroes the 16 bytes behind what the
// first stage wrote: the second stage's byte sweep reads whole 16-byte pieces and must not take pool garbage
// for bytes >= 0x80)
struct StageCaps { int64_t cap[kMaxStageOutputs]; void* data[kMaxStageOutputs]; };
hipError_t LaunchStageGate(const uint64_t* stage_result, int num_outputs, const StageCaps& caps, const int64_t* rows_in,
                           int64_t rows, int64_t* rows_out, uint64_t* status_out, hipStream_t stream);
// result[0] = *err & ~clear: an asynchronous evaluation publishes its launch's error word WITHOUT the bits that are
// notes to the host, not errors (the exact string kernels' "saw UTF-8", bit 64) — `0 = complete` is the contract
hipError_t LaunchPublishStatus(uint64_t* result, const uint32_t* err, uint32_t clear, hipStream_t stream);
// *dst = (*err & fatal) ? -1 : *count — the selected-row count of a fused filter-project launch as an asynchronous
// caller sees it: negative when the launch did not complete (its look-back gave up)
// (share: may be null — the operator's own pinned word; receives the rows selected per 1024 rows of THIS launch (`rows` of
// them), through which an asynchronous caller's selectivity reaches the host one call late: enough to choose the kernel
// shape of the next batch.  One word, written by one launch: count and row number cannot come from different batches.)
hipError_t LaunchPublishCount(int64_t* dst, const int64_t* count, const uint32_t* err, uint32_t fatal, hipStream_t stream,
                              int64_t* share = nullptr, int64_t rows = 0);
// result[0] |= *status
hipError_t LaunchOrStatus(uint64_t* result, const uint64_t* status, hipStream_t stream);

hipError_t LaunchEmitIndices(const uint64_t* mask, const uint64_t* offsets, int64_t nwords,
                             int subtiles, int64_t row_base, int index_bytes, void* out,
                             int num_cus, hipStream_t stream);

// Read-only / write-only / copy rate (GB/s, best of 4) of plain streaming kernels over two device
// buffers of `bytes` bytes each (bytes a multiple of 16; use >= 1 GiB: the Infinity Cache holds 256 MiB).
hipError_t MeasureStreamCeiling(void* const* streams, int nr, int nw, size_t elems, int num_cus, double* gbs,
                                int* workgroups_per_cu, int* nontemporal);
hipError_t MeasureHbmCeilings(void* a, void* b, size_t bytes, int grid, double* read_gbs, double* write_gbs,
                              double* copy_gbs);

}  // namespace gdv
