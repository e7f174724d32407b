// gandiva/host_memory.h — NOT part of the reference's API (its buffers are the CPU's own): the one
// thing a C++ caller of this implementation may want to add.  Evaluate() on CPU-resident Arrow
// buffers stages them through a page-locked block; buffers inside a range registered here are read
// and written by the GPU in place (include/gandiva_amd.h: gdv_host_register).  Typical use: an
// arrow::MemoryPool that carves its allocations out of a few large chunks registers each chunk
// when it maps it and unregisters it before it unmaps it — Projector::Evaluate(batch, pool, &out)
// then allocates its outputs there and nothing is copied.
#pragma once
#include <cstdint>

#include "gandiva/arrow.h"

namespace gandiva {

// Page-lock [ptr, ptr + bytes) and map it into every GPU.  The caller owns the memory and
// unregisters it BEFORE freeing it.
Status RegisterHostMemory(void* ptr, int64_t bytes);
Status UnregisterHostMemory(void* ptr);
// Bytes Evaluate() calls on CPU-resident buffers have copied through staging blocks so far.
int64_t HostStagedBytes();

}  // namespace gandiva
