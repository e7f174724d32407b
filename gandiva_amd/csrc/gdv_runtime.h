// HIP runtime plumbing: kernel compilation (hipRTC) with an in-memory + on-disk code-object
// cache, kernel launch with a by-value argument block, a caching device allocator for scratch and
// staged buffers — one context per device, chosen by the calling thread.  Replaces the reference's Engine and the
// compiled-module cache (row 12).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <shared_mutex>
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
  hipFunction_t function_many = nullptr;  // <name>_many(const gdv_args* table), when the plan has one
  hipFunction_t function_small = nullptr; // <name>_small(const gdv_args* table): fused small-batch filter
  hipFunction_t function_small1 = nullptr;  // <name>_small1(const gdv_args A): the same for one batch, by value
  std::string name;
};

// One Runtime per device CONTEXT (round 3; rounds 1-2: a process singleton bound to the device
// current at first use).  A context owns what lives on one GPU: loaded code objects, the buffer
// pool, the all-ones bitmap word, side streams and events.  Which context a call uses is decided
// by the CALLING THREAD: the device it selected with SelectDevice() (gdv_set_device in the C ABI),
// else its current HIP device — so one host thread per device drives N GPUs from one process
// (SURVEY.md §8e), and torch / HIP callers that hipSetDevice() themselves need nothing else.
// Virtual devices: ids beyond the physical count map onto physical devices round-robin
// (id % physical) with a context of their own — N contexts on one GPU, which is how the in-process
// sharding tests run on the single-GPU test box.
class Runtime {
 public:
  static constexpr int kMaxDevices = 64;
  static Runtime& Get();                 // the calling thread's context
  static Runtime& ForDevice(int id);     // created on first use
  static int PhysicalDeviceCount();      // 0 without a HIP device
  static int DeviceCount();              // max(physical, SetVirtualDevices / GDV_VIRTUAL_DEVICES)
  static void SetVirtualDevices(int n);
  static Status SelectDevice(int id);    // thread-local; also makes the physical device current
  static int SelectedDevice();           // the id Get() resolves to on this thread

  int id() const { return id_; }
  int physical() const { return physical_; }

  // True when a HIP device is usable in this process.
  bool has_device();
  Status EnsureDevice();      // the calling thread's HIP device := this context's
  int num_cus();
  const std::string& arch();  // "gfx950" when no device is present (cross-compile)

  // Compiles `source` (which #includes "gdv_device_lib.hpp") for arch() and returns the
  // code object; cached in the process (every context loads the same object) and on disk by
  // kernel name (= hash of the source + reached library items) + hash of the whole library.
  Status CompileToCodeObject(const std::string& source, const std::string& kernel_name,
                             std::vector<char>* code, bool* from_cache = nullptr,
                             bool ignore_cached = false);
  // Tier 0 (round 6).  CodeObjectState: 1 = the code object of `kernel_name` is at hand (in memory, or on disk: a load
  // away), 0 = not yet, -1 = a background compilation of it failed (the blocking path will report why).
  // CompileInBackground: queue the compilation on the process's one compiler thread and return at once.
  int CodeObjectState(const std::string& kernel_name, bool memory_only = false);
  bool CompileInBackground(const std::string& source, const std::string& kernel_name);  // false after ShutdownBackgroundCompiler
  // stops the background compiler: queued compilations are dropped, the one in flight is waited for.  Plans whose
  // compilation was dropped compile on their next blocking use.  Safe to call more than once.
  static void ShutdownBackgroundCompiler();
  // CompileToCodeObject + hipModuleLoadData on this context's device; cached per context.
  Status GetKernel(const std::string& source, const std::string& kernel_name,
                   const CompiledKernel** out);

  // 256-byte aligned device memory from a size-bucketed free list.
  Status Alloc(size_t bytes, void** ptr);
  void Free(void* ptr);
  // Free once everything enqueued on `stream` so far has run: asynchronous evaluations return
  // before their scratch is idle.  An event guards the block; Alloc recycles completed ones.
  void FreeAfter(void* ptr, hipStream_t stream);
  void TrimPool();

  // Page-locked host blocks of kPinnedBlock bytes (small-batch host path: one H2D and one
  // D2H per Evaluate instead of one per buffer).  Returned blocks are kept for reuse.
  static constexpr size_t kPinnedBlock = 16u << 20;
  Status AcquirePinned(char** p);
  void ReleasePinned(char* p);
  // ... and small ones (argument tables of multi-batch launches: asynchronous calls hold theirs
  // until the stream has passed, so several are in flight at a time)
  static constexpr size_t kPinnedSmall = 64u << 10;
  Status AcquirePinnedSmall(char** p);
  void ReleasePinnedSmall(char* p);

  // Device-resident bitmap word with all 64 bits set: what a column WITHOUT a validity
  // (or with an elided all-valid) buffer is bound to, so kernels never branch on "has nulls".
  Status AllOnesWord(const uint64_t** ptr);

  Status Launch(const CompiledKernel& k, int64_t grid, int block, const void* args,
                size_t arg_bytes, hipStream_t stream, hipFunction_t entry = nullptr);
  // the multi-batch entry point: grid (grid_x, batches), argument blocks in a device table
  Status LaunchMany(const CompiledKernel& k, int64_t grid_x, int64_t batches, int block, const void* table_device,
                    hipStream_t stream, bool small = false);
  // run `fn` once everything enqueued on `stream` so far has completed (polled from Alloc /
  // AcquirePinned: no callback thread) — how asynchronous calls give pinned blocks back
  void Defer(hipStream_t stream, std::function<void()> fn);

  // Side streams / events for pipelines inside one Evaluate (filter: index emission of chunk k
  // behind the predicate kernel of chunk k + 1).  Pooled: creation costs tens of microseconds.
  Status AcquireStream(hipStream_t* s);
  void ReleaseStream(hipStream_t s);
  Status AcquireEvent(hipEvent_t* e);
  void ReleaseEvent(hipEvent_t e);

  std::string cache_dir();

 private:
  explicit Runtime(int id) : id_(id) {}
  std::mutex mu_;
  int id_ = 0;
  int physical_ = 0;
  bool probed_ = false;
  bool has_device_ = false;
  int num_cus_ = 256;
  std::string arch_ = "gfx950";
  std::map<std::string, std::unique_ptr<CompiledKernel>> kernels_;
  std::multimap<size_t, void*> free_blocks_;
  std::map<void*, size_t> live_blocks_;
  std::vector<char*> pinned_free_, pinned_small_free_;
  std::vector<hipStream_t> streams_free_;
  std::vector<hipEvent_t> events_free_;
  std::vector<std::pair<hipEvent_t, void*>> deferred_;  // FreeAfter: blocks waiting for their event
  std::vector<std::pair<hipEvent_t, std::function<void()>>> deferred_fns_;
  void Reap(bool wait);
  size_t cached_bytes_ = 0;
  uint64_t* all_ones_ = nullptr;
  void Probe();
};

// RAII scratch buffer from a context's pool (freed to the context it came from, whatever device
// the freeing thread has selected).
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  ~DeviceBuffer() { reset(); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_), owner_(o.owner_) { o.p_ = nullptr; o.n_ = 0; }
  Status Allocate(size_t bytes) {
    reset();
    n_ = bytes;
    owner_ = &Runtime::Get();
    return owner_->Alloc(bytes ? bytes : 1, &p_);
  }
  void reset() {
    if (p_) owner_->Free(p_);
    p_ = nullptr;
    n_ = 0;
  }
  // give the block back once the work enqueued on `stream` so far has run
  void release_after(hipStream_t stream) {
    if (p_) owner_->FreeAfter(p_, stream);
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
  Runtime* owner_ = nullptr;
};

extern const char gdv_device_lib_src[];  // generated: gdv_device_lib_embed.cc

// Host memory the GPUs can address directly (round 4; process-wide, every device): ranges the caller
// page-locked through gdv_host_register (hipHostRegister: its own arenas — an Arrow MemoryPool's,
// JNI direct buffers) or obtained from gdv_host_alloc (hipHostMalloc).  The host-buffer path of
// Evaluate binds fixed-width columns, bitmaps and outputs that lie inside such a range straight into
// the kernel's argument block — the kernel reads and writes them over the fabric, no staging memcpy,
// no H2D / D2H copy — and stages everything else as before.  Nothing is registered behind the
// caller's back: a registration outliving a free() would leave the device a window onto pages that
// now belong to someone else.
class HostRegistry {
 public:
  static HostRegistry& Get();
  Status Register(void* p, size_t bytes);
  Status Unregister(void* p);
  Status Alloc(size_t bytes, void** p);
  Status Free(void* p);
  // Device-visible address of [p, p + bytes) — widened to 8-byte boundaries, the granularity the
  // kernels read bitmaps at — when all of it lies inside ONE registered range; nullptr otherwise.
  void* View(const void* p, size_t bytes) const;
  bool empty() const { return count_.load(std::memory_order_relaxed) == 0; }
  // bytes GDV_MEM_HOST evaluations have copied through staging blocks so far (process-wide): what a
  // caller watches to see whether its registrations take effect
  static std::atomic<int64_t>& StagedBytes() {
    static std::atomic<int64_t> n{0};
    return n;
  }

 private:
  struct Range {
    size_t size;
    char* dev;
    bool owned;  // gdv_host_alloc
  };
  Status Insert(void* p, size_t bytes, bool owned);
  mutable std::shared_mutex mu_;
  std::map<uintptr_t, Range> ranges_;
  std::atomic<int> count_{0};
};

}  // namespace gdv
