// Device pool (round 6, verdict item 2): library-owned HBM for the buffers a caller streams together — above all the output
// columns of a projection.  Two things a plain allocation does not give:
//   * placement.  Where the driver puts a set of large buffers decides what a kernel over them runs at, and stays with the
//     buffers (C2 on one box: 4.88 .. 6.30 ms per Evaluate over ten placements, profiles/r06_placement_probe.txt).  What is
//     slow is a set of NEIGHBOURS in allocation order; members spread over a wide span of allocations are fast.  ReserveSet
//     allocates four times the set as single buffers, forms candidates from every fourth one, times a non-temporal write
//     sweep over each candidate, keeps the fastest and gives everything else back to the driver;
//   * retention.  Buffers freed to the pool stay in it: a placement found once serves every later batch of that shape.
#pragma once
#include <map>
#include <mutex>
#include <vector>

#include "gdv_runtime.h"

namespace gdv {

class DevicePool {
 public:
  DevicePool();   // on the calling thread's device context
  ~DevicePool();  // everything goes back to the driver
  // `count` buffers of `bytes` each; up to min(candidates, 4) strided candidates out of 4 x count single allocations
  // (less when free memory does not hold them).  rates (may be null, capacity `candidates`): GB/s of every candidate's
  // sweep, in the order tried; *tried = how many; *kept = the index of the one that was kept.
  Status ReserveSet(int count, int64_t bytes, int candidates, void** ptrs, double* rates, int* tried, int* kept);
  Status Alloc(int64_t bytes, void** ptr);  // a retained buffer of exactly this size, else a fresh allocation
  Status Free(void* ptr);                   // back to the pool, retained
  Status Trim();                            // retained-but-unused buffers go back to the driver
  int64_t bytes_held(int64_t* in_use) const;
  int device() const { return device_; }

 private:
  Status Raw(int64_t bytes, void** ptr);
  int device_ = 0;
  Runtime* rt_ = nullptr;
  mutable std::mutex mu_;
  std::multimap<int64_t, void*> free_;
  std::map<void*, int64_t> live_;
};

}  // namespace gdv
