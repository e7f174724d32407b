// gandiva/host_memory.h — NOT part of the reference's API (its buffers are the CPU's own): the one
// thing a C++ caller of this implementation may want to add.  Evaluate() on CPU-resident Arrow
// buffers stages them through a page-locked block; buffers inside a range registered here are read
// and written by the GPU in place (include/gandiva_amd.h: gdv_host_register).  Typical use: an
// arrow::MemoryPool that carves its allocations out of a few large chunks registers each chunk
// when it maps it and unregisters it before it unmaps it — Projector::Evaluate(batch, pool, &out)
// then allocates its outputs there and nothing is copied.
#pragma once
#include <cstdint>
#include <memory>
#include <string>

#include "arrow/memory_pool.h"
#include "gandiva/arrow.h"

namespace gandiva {

// Page-lock [ptr, ptr + bytes) and map it into every GPU.  The caller owns the memory and
// unregisters it BEFORE freeing it.
Status RegisterHostMemory(void* ptr, int64_t bytes);
Status UnregisterHostMemory(void* ptr);
// Bytes Evaluate() calls on CPU-resident buffers have copied through staging blocks so far.
int64_t HostStagedBytes();

// An arrow::MemoryPool whose memory the GPUs address directly: page-locked chunks from the library
// (gdv_host_alloc), carved into power-of-two blocks (64 bytes up; free lists per size).  Arrays built
// in it and outputs allocated from it —
//     HostMemoryPool pool;  arrow::Int64Builder b(&pool); ...  projector->Evaluate(*batch, &pool, &out);
// — are evaluated in place: nothing is staged, nothing is copied.  Requests above half a chunk get a
// page-locked block of their own (hipHostMalloc: ~0.1 ms per MiB — size the chunk for the batches).
// Thread-safe.  Destroy it after every buffer it handed out is gone.
class HostMemoryPool : public arrow::MemoryPool {
 public:
  explicit HostMemoryPool(int64_t chunk_bytes = int64_t{64} << 20);
  ~HostMemoryPool() override;
  using arrow::MemoryPool::Allocate;
  using arrow::MemoryPool::Free;
  using arrow::MemoryPool::Reallocate;
  Status Allocate(int64_t size, int64_t alignment, uint8_t** out) override;
  Status Reallocate(int64_t old_size, int64_t new_size, int64_t alignment, uint8_t** ptr) override;
  void Free(uint8_t* buffer, int64_t size, int64_t alignment) override;
  int64_t bytes_allocated() const override;
  int64_t max_memory() const override;
  int64_t total_bytes_allocated() const override;
  int64_t num_allocations() const override;
  std::string backend_name() const override { return "gandiva_amd-host"; }

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace gandiva
