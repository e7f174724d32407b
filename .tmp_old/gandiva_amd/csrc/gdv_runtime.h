// HIP runtime plumbing: kernel compilation (hipRTC) with an in-memory + on-disk code-object
// cache, kernel launch with a by-value argument block, and a caching device allocator for
// scratch and staged buffers.  Replaces the reference's Engine and the
// compiled-module cache (row 12).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gdv_types.h"

namespace gdv {

#define GDV_HIP_RETURN_NOT_OK(expr)                                                        \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return ::gdv::Status::ExecutionError(std::string(#expr) + " failed: " +              \
                                           hipGetErrorString(_e));                         \
  } while (0)

struct CompiledKernel {
  hipModule_t module = nullptr;
  hipFunction_t function = nullptr;
  std::string name;
};

class Runtime {
 public:
  static Runtime& Get();

  // True when a HIP device is usable in this process.
  bool has_device();
  Status EnsureDevice();
  int num_cus();
  const std::string& arch();  // "gfx950" when no device is present (cross-compile)

  // Compiles `source` (which #includes "gdv_device_lib.hpp") for arch() and returns the
  // code object; cached on disk by kernel name (= hash of the source) + library hash.
  Status CompileToCodeObject(const std::string& source, const std::string& kernel_name,
                             std::vector<char>* code, bool* from_cache = nullptr,
                             bool ignore_cached = false);
  // CompileToCodeObject + hipModuleLoadData; cached per process.
  Status GetKernel(const std::string& source, const std::string& kernel_name,
                   const CompiledKernel** out);

  // 256-byte aligned device memory from a size-bucketed free list.
  Status Alloc(size_t bytes, void** ptr);
  void Free(void* ptr);
  void TrimPool();

  // Page-locked host blocks of kPinnedBlock bytes (small-batch host path: one H2D and one
  // D2H per Evaluate instead of one per buffer).  Returned blocks are kept for reuse.
  static constexpr size_t kPinnedBlock = 16u << 20;
  Status AcquirePinned(char** p);
  void ReleasePinned(char* p);

  // Device-resident bitmap word with all 64 bits set: what a column WITHOUT a validity
  // (or with an elided all-valid) buffer is bound to, so kernels never branch on "has nulls".
  Status AllOnesWord(const uint64_t** ptr);

  Status Launch(const CompiledKernel& k, int64_t grid, int block, const void* args,
                size_t arg_bytes, hipStream_t stream);

  std::string cache_dir();

 private:
  Runtime() = default;
  std::mutex mu_;
  bool probed_ = false;
  bool has_device_ = false;
  int num_cus_ = 256;
  int device_ = 0;  // the device that was current at first use
  std::string arch_ = "gfx950";
  std::map<std::string, std::unique_ptr<CompiledKernel>> kernels_;
  std::multimap<size_t, void*> free_blocks_;
  std::map<void*, size_t> live_blocks_;
  std::vector<char*> pinned_free_;
  size_t cached_bytes_ = 0;
  uint64_t* all_ones_ = nullptr;
  void Probe();
};

// RAII scratch buffer from the runtime pool.
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  ~DeviceBuffer() { reset(); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  Status Allocate(size_t bytes) {
    reset();
    n_ = bytes;
    return Runtime::Get().Alloc(bytes ? bytes : 1, &p_);
  }
  void reset() {
    if (p_) Runtime::Get().Free(p_);
    p_ = nullptr;
    n_ = 0;
  }
  void* get() const { return p_; }
  template <typename T>
  T* as() const { return static_cast<T*>(p_); }
  size_t size() const { return n_; }

 private:
  void* p_ = nullptr;
  size_t n_ = 0;
};

extern const char gdv_device_lib_src[];  // generated: gdv_device_lib_embed.cc

}  // namespace gdv
