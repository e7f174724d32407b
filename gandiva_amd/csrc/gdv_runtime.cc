#include "gdv_runtime.h"

#include <dlfcn.h>
#include <sys/syscall.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <set>
#include <thread>

namespace gdv {

namespace {
std::mutex g_contexts_mu;
Runtime* g_contexts[Runtime::kMaxDevices] = {};
int g_virtual_devices = 0;
thread_local int tl_selected_device = -1;

int PhysicalCountUncached() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  return count;
}
}  // namespace

int Runtime::PhysicalDeviceCount() {
  static const int count = PhysicalCountUncached();
  return count;
}

int Runtime::DeviceCount() {
  int v = 0;
  {
    std::lock_guard<std::mutex> g(g_contexts_mu);
    v = g_virtual_devices;
  }
  if (v == 0) {  // GDV_VIRTUAL_DEVICES: read once per process
    static const int from_env = [] { const char* e = std::getenv("GDV_VIRTUAL_DEVICES"); return e ? atoi(e) : 0; }();
    v = from_env;
  }
  const int phys = PhysicalDeviceCount();
  if (phys == 0) return 0;
  return std::min<int>(kMaxDevices, std::max(phys, v));
}

void Runtime::SetVirtualDevices(int n) {
  std::lock_guard<std::mutex> g(g_contexts_mu);
  g_virtual_devices = std::max(0, std::min<int>(n, kMaxDevices));
}

Runtime& Runtime::ForDevice(int id) {
  if (id < 0 || id >= kMaxDevices) id = 0;
  std::lock_guard<std::mutex> g(g_contexts_mu);
  // never destroyed: cached Projectors / Filters (process-wide LRUs) own pooled device blocks and
  // hand them back during static destruction, in an order the contexts must outlive
  if (g_contexts[id] == nullptr) g_contexts[id] = new Runtime(id);
  return *g_contexts[id];
}

int Runtime::SelectedDevice() {
  if (tl_selected_device >= 0) return tl_selected_device;
  int dev = 0;
  if (PhysicalDeviceCount() > 0 && hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = 0;
  }
  return dev;
}

Runtime& Runtime::Get() { return ForDevice(SelectedDevice()); }

Status Runtime::SelectDevice(int id) {
  const int n = DeviceCount();
  if (n == 0) return Status::ExecutionError("no HIP device available");
  if (id < 0 || id >= n)
    return Status::Invalid("device " + std::to_string(id) + " out of range (" + std::to_string(n) + " devices)");
  tl_selected_device = id;
  return ForDevice(id).EnsureDevice();
}

void Runtime::Probe() {
  if (probed_) return;
  probed_ = true;
  const int count = PhysicalDeviceCount();
  if (count <= 0) {
    has_device_ = false;
    if (const char* a = std::getenv("GDV_ARCH")) arch_ = a;
    return;
  }
  has_device_ = true;
  physical_ = id_ % count;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, physical_) == hipSuccess) {
    num_cus_ = prop.multiProcessorCount;
    std::string a = prop.gcnArchName;  // e.g. "gfx950:sramecc+:xnack-"
    size_t colon = a.find(':');
    arch_ = colon == std::string::npos ? a : a.substr(0, colon);
  }
}

bool Runtime::has_device() {
  std::lock_guard<std::mutex> g(mu_);
  Probe();
  return has_device_;
}

Status Runtime::EnsureDevice() {
  if (!has_device())
    return Status::ExecutionError(
        "no HIP device available: gandiva_amd evaluates on the GPU only (there is no CPU "
        "fallback)");
  // Code objects, the buffer pool and the all-ones word of this context live on physical_:
  // make it the calling thread's HIP device (a thread that never chose one starts on device 0).
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != physical_) {
    hipError_t e = hipSetDevice(physical_);
    if (e != hipSuccess)
      return Status::ExecutionError("could not make HIP device " + std::to_string(physical_) +
                                    " current on the calling thread: " + hipGetErrorString(e));
  }
  return Status::OK();
}

int Runtime::num_cus() {
  std::lock_guard<std::mutex> g(mu_);
  Probe();
  return num_cus_;
}

const std::string& Runtime::arch() {
  std::lock_guard<std::mutex> g(mu_);
  Probe();
  return arch_;
}

static bool DirWritable(const std::string& d) {
  mkdir(d.c_str(), 0755);
  return access(d.c_str(), W_OK) == 0;
}

// A cache directory outside the installation is trusted only if it is ours and nobody else
// can write to it: code objects found there are loaded and run against the caller's HBM.
static bool PrivateDir(const std::string& d) {
  mkdir(d.c_str(), 0700);
  struct stat st;
  if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  if (st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) return false;
  return access(d.c_str(), W_OK) == 0;
}

std::string Runtime::cache_dir() {
  if (const char* e = std::getenv("GANDIVA_AMD_CACHE_DIR")) {
    std::string d = e;
    if (DirWritable(d)) return d;
  }
  // in-tree cache next to the shared library: travels with the repo snapshot
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&gdv_device_lib_src), &info) && info.dli_fname) {
    std::string so = info.dli_fname;
    size_t slash = so.rfind('/');
    std::string d = (slash == std::string::npos ? std::string(".") : so.substr(0, slash)) +
                    "/_kcache";
    if (DirWritable(d)) return d;
  }
  // per-user cache: $XDG_CACHE_HOME or ~/.cache, created 0700 and verified; as a last
  // resort a per-uid directory under /tmp with the same checks.  "" = no disk cache.
  std::string base;
  if (const char* x = std::getenv("XDG_CACHE_HOME")) base = x;
  else if (const char* h = std::getenv("HOME")) base = std::string(h) + "/.cache";
  if (!base.empty()) {
    mkdir(base.c_str(), 0700);
    std::string d = base + "/gandiva_amd_kcache";
    if (PrivateDir(d)) return d;
  }
  std::string d = "/tmp/gandiva_amd_kcache_" + std::to_string(static_cast<long>(geteuid()));
  if (PrivateDir(d)) return d;
  return "";
}

static uint64_t Fnv(const char* s, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) {
    h ^= static_cast<unsigned char>(s[i]);
    h *= 1099511628211ull;
  }
  return h;
}

namespace {
// (never destroyed: the background compiler thread may still be inside CompileToCodeObject while the process's statics are
// torn down — it is joined by ~BackgroundCompiler, which can run AFTER these would have been destroyed)
std::mutex& CodeCacheMutex() { static std::mutex* m = new std::mutex; return *m; }
std::map<std::string, std::vector<char>>& CodeCache() { static auto* c = new std::map<std::string, std::vector<char>>; return *c; }
std::set<std::string>& FailedCompilations() { static auto* s = new std::set<std::string>; return *s; }
std::string LibraryHashTag() {  // the on-disk cache is keyed by the hash of the whole device library as well
  static const uint64_t lib_hash = Fnv(gdv_device_lib_src, strlen(gdv_device_lib_src));
  char tag[40];
  snprintf(tag, sizeof(tag), "%016llx", static_cast<unsigned long long>(lib_hash));
  return tag;
}

// The process's one background compiler thread (tier 0): compilations are serialised anyway (CompileToCodeObject), a
// queue keeps Make from waiting for them.  Joined at exit: a detached thread inside hipRTC while the statics of the
// process are torn down is a crash.
class BackgroundCompiler {
 public:
  // hipRTC loads its compiler (libamd_comgr) with dlopen on first use, i.e. from the worker thread, AFTER this object
  // exists: comgr's static destructors would then run BEFORE this object's at exit — while the worker may still be
  // compiling inside it (SIGSEGV at exit of any process that ends within a few hundred milliseconds of a Make; found by
  // tests/test_tier0.py and the C++ acceptance binary).  Loading comgr here, in the constructor, registers its
  // teardown first: this object's destructor — which stops the queue and JOINS the worker — runs before it.
  BackgroundCompiler() {
    for (const char* name : {"libamd_comgr.so.3", "libamd_comgr.so"}) (void)dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  static BackgroundCompiler& Get() { static BackgroundCompiler b; return b; }
  // stop accepting work, forget what is queued, wait for the compilation in flight (gdv_shutdown; also the destructor)
  void Shutdown() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; jobs_.clear(); }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
  }
  bool Push(std::string source, std::string name) {  // false: shut down — the caller compiles on its own thread
    std::lock_guard<std::mutex> g(mu_);
    if (stop_) return false;
    for (auto& j : jobs_) if (j.second == name) return true;
    jobs_.emplace_back(std::move(source), std::move(name));
    if (!started_) { started_ = true; worker_ = std::thread([this] { Run(); }); }
    cv_.notify_one();
    return true;
  }
  ~BackgroundCompiler() { Shutdown(); }

 private:
  void Run() {
    for (;;) {
      std::pair<std::string, std::string> job;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return stop_ || !jobs_.empty(); });
        if (stop_) return;
        job = std::move(jobs_.front());
        jobs_.pop_front();
      }
      std::vector<char> code;
      Status st = Runtime::ForDevice(0).CompileToCodeObject(job.first, job.second, &code);
      if (!st.ok()) {
        std::lock_guard<std::mutex> g(CodeCacheMutex());
        FailedCompilations().insert(job.second);
      }
      // (the compiler's function-local statics that THIS compilation constructed are torn down, at exit, after a handler
      // registered now — see ExitHook)
      (void)std::atexit(&BackgroundCompiler::ExitHook);
    }
  }
 public:
  static void ExitHook() { Get().Shutdown(); }
 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::pair<std::string, std::string>> jobs_;
  std::thread worker_;
  bool started_ = false, stop_ = false;
};
}  // namespace

int Runtime::CodeObjectState(const std::string& kernel_name, bool memory_only) {
  const std::string a = arch();
  {
    std::lock_guard<std::mutex> g(CodeCacheMutex());
    if (CodeCache().count(kernel_name + "." + a)) return 1;
    if (FailedCompilations().count(kernel_name)) return -1;
  }
  if (memory_only || std::getenv("GDV_NO_DISK_CACHE") != nullptr) return 0;
  const std::string dir = cache_dir();
  if (dir.empty()) return 0;
  const std::string path = dir + "/" + kernel_name + "." + LibraryHashTag() + "." + a + ".hsaco";
  return access(path.c_str(), R_OK) == 0 ? 1 : 0;
}

// Process exit with a compilation in flight.  exit() runs the exit handlers newest first, and the compiler (LLVM inside
// libamd_comgr) registers the destructors of its function-local statics as it first executes them, i.e. DURING compilations:
// they are newer than anything this library registered before, so at exit they are torn down first — under a worker that is
// still compiling (round 6, found with a cold code-object cache: the C++ acceptance binary, which finishes on tier 0 within a
// second of its first Make, crashed or hung in its exit).  Three layers:
//  * exit() destroys the CALLING thread's thread_local objects before it runs any handler: a guard object on every thread that
//    queues a compilation stops the queue and joins the worker from there when that thread is the process's main thread;
//  * a handler registered after every finished compilation (BackgroundCompiler::ExitHook) orders the join ahead of the teardown
//    of everything earlier compilations constructed — for processes whose main thread never called Make;
//  * gdv_shutdown() for embedders that can call it (the Python package does, from atexit).
namespace {
struct MainThreadExitGuard {
  MainThreadExitGuard() : tid(static_cast<long>(syscall(SYS_gettid))) {}
  ~MainThreadExitGuard() {
    if (tid == static_cast<long>(getpid())) Runtime::ShutdownBackgroundCompiler();
  }
  long tid;
};
}  // namespace

bool Runtime::CompileInBackground(const std::string& source, const std::string& kernel_name) {
  thread_local MainThreadExitGuard guard;
  (void)guard.tid;
  return BackgroundCompiler::Get().Push(source, kernel_name);
}
void Runtime::ShutdownBackgroundCompiler() { BackgroundCompiler::Get().Shutdown(); }

Status Runtime::CompileToCodeObject(const std::string& source, const std::string& kernel_name,
                                    std::vector<char>* code, bool* from_cache,
                                    bool ignore_cached) {
  const std::string a = arch();
  const std::string tag = LibraryHashTag();
  if (std::getenv("GDV_DUMP_SOURCE")) {
    const std::string d = cache_dir();
    std::ofstream f((d.empty() ? std::string("/tmp") : d) + "/" + kernel_name + ".hip");
    f << source;
  }
  // every context loads the same code object: compiled (or read from disk) once per process
  std::mutex& code_mu = CodeCacheMutex();
  std::map<std::string, std::vector<char>>& code_cache = CodeCache();
  const std::string mem_key = kernel_name + "." + a;
  if (!ignore_cached) {
    std::lock_guard<std::mutex> g(code_mu);
    auto it = code_cache.find(mem_key);
    if (it != code_cache.end()) {
      *code = it->second;
      if (from_cache) *from_cache = true;
      return Status::OK();
    }
  }
  auto remember = [&]() {
    std::lock_guard<std::mutex> g(code_mu);
    if (code_cache.size() > 2000) code_cache.clear();
    code_cache[mem_key] = *code;
  };
  const std::string dir = cache_dir();
  const std::string path = dir + "/" + kernel_name + "." + tag + "." + a + ".hsaco";
  const bool use_disk = !dir.empty() && std::getenv("GDV_NO_DISK_CACHE") == nullptr;
  if (use_disk && ignore_cached) unlink(path.c_str());  // stale or corrupt: recompiled below
  if (use_disk && !ignore_cached) {
    std::ifstream f(path, std::ios::binary);
    if (f) {
      code->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      if (!code->empty()) {
        if (from_cache) *from_cache = true;
        remember();
        return Status::OK();
      }
    }
  }
  if (from_cache) *from_cache = false;

  // one compilation at a time: they are rare (cached in memory and on disk) and comgr's
  // temporary-file handling has no need to be exercised concurrently
  static std::mutex& compile_mu = *new std::mutex;  // (leaked on purpose: see CodeCacheMutex)
  std::lock_guard<std::mutex> compile_guard(compile_mu);
  if (!ignore_cached) {  // (another thread — the background compiler of tier 0 — may have produced it while this one waited)
    std::lock_guard<std::mutex> g(code_mu);
    auto it = code_cache.find(mem_key);
    if (it != code_cache.end()) {
      *code = it->second;
      if (from_cache) *from_cache = true;
      return Status::OK();
    }
  }
  hiprtcProgram prog;
  const char* hdr_src[] = {gdv_device_lib_src};
  const char* hdr_name[] = {"gdv_device_lib.hpp"};
  if (hiprtcCreateProgram(&prog, source.c_str(), (kernel_name + ".hip").c_str(), 1, hdr_src,
                          hdr_name) != HIPRTC_SUCCESS)
    return Status::CodeGenError("hiprtcCreateProgram failed");
  std::string arch_opt = "--offload-arch=" + a;
  // -ffp-contract=off is part of the semantics (bit-exact vs separate mul/add)
  std::vector<const char*> opts = {arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off",
                                   "-fhip-fp32-correctly-rounded-divide-sqrt"};
  // GDV_RTC_OPT: extra compiler options for experiments (e.g. -DGDV_COLD= inlines the cold
  // paths); not part of the cache key, so combine it with GDV_NO_DISK_CACHE=1
  std::vector<std::string> extra_opts;
  if (const char* extra = std::getenv("GDV_RTC_OPT")) {  // blank-separated list
    std::string cur;
    for (const char* c = extra;; c++) {
      if (*c == ' ' || *c == '\0') {
        if (!cur.empty()) extra_opts.push_back(cur);
        cur.clear();
        if (*c == '\0') break;
      } else {
        cur.push_back(*c);
      }
    }
    for (auto& o : extra_opts) opts.push_back(o.c_str());
  }
  hiprtcResult r = hiprtcCompileProgram(prog, static_cast<int>(opts.size()), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    return Status::CodeGenError("kernel compilation failed:\n" + log);
  }
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  code->resize(n);
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  if (use_disk) {
    std::string tmp = path + ".tmp." + std::to_string(getpid());
    std::ofstream f(tmp, std::ios::binary);
    if (f) {
      f.write(code->data(), static_cast<std::streamsize>(code->size()));
      f.close();
      rename(tmp.c_str(), path.c_str());
    }
  }
  remember();
  return Status::OK();
}

Status Runtime::GetKernel(const std::string& source, const std::string& kernel_name,
                          const CompiledKernel** out) {
  GDV_RETURN_NOT_OK(EnsureDevice());
  {
    std::lock_guard<std::mutex> g(mu_);
    auto it = kernels_.find(kernel_name);
    if (it != kernels_.end()) {
      *out = it->second.get();
      return Status::OK();
    }
  }
  std::vector<char> code;
  bool from_cache = false;
  GDV_RETURN_NOT_OK(CompileToCodeObject(source, kernel_name, &code, &from_cache));
  auto k = std::make_unique<CompiledKernel>();
  k->name = kernel_name;
  hipError_t le = hipModuleLoadData(&k->module, code.data());
  if (le == hipSuccess) le = hipModuleGetFunction(&k->function, k->module, kernel_name.c_str());
  if (le != hipSuccess && from_cache) {
    // a cached code object that does not load (truncated, built by another toolchain): drop
    // the file and compile afresh instead of failing every Make from now on
    (void)hipGetLastError();
    if (k->module != nullptr) (void)hipModuleUnload(k->module);
    k->module = nullptr;
    GDV_RETURN_NOT_OK(CompileToCodeObject(source, kernel_name, &code, nullptr, true));
    le = hipModuleLoadData(&k->module, code.data());
    if (le == hipSuccess) le = hipModuleGetFunction(&k->function, k->module, kernel_name.c_str());
  }
  if (le != hipSuccess)
    return Status::ExecutionError(std::string("loading the compiled kernel failed: ") + hipGetErrorString(le));
  if (source.find(kernel_name + "_many(") != std::string::npos &&
      hipModuleGetFunction(&k->function_many, k->module, (kernel_name + "_many").c_str()) != hipSuccess) {
    (void)hipGetLastError();
    k->function_many = nullptr;
  }
  if (source.find(kernel_name + "_small(") != std::string::npos &&
      hipModuleGetFunction(&k->function_small, k->module, (kernel_name + "_small").c_str()) != hipSuccess) {
    (void)hipGetLastError();
    k->function_small = nullptr;
  }
  if (k->function_small != nullptr &&
      hipModuleGetFunction(&k->function_small1, k->module, (kernel_name + "_small1").c_str()) != hipSuccess) {
    (void)hipGetLastError();
    k->function_small1 = nullptr;
  }
  std::lock_guard<std::mutex> g(mu_);
  auto& slot = kernels_[kernel_name];
  if (!slot) slot = std::move(k);
  *out = slot.get();
  return Status::OK();
}

static size_t RoundSize(size_t b) {
  if (b < 256) return 256;
  if (b < (1u << 20)) {
    size_t p = 256;
    while (p < b) p <<= 1;
    return p;
  }
  const size_t g = 2u << 20;
  return (b + g - 1) / g * g;
}

void Runtime::FreeAfter(void* ptr, hipStream_t stream) {
  if (!ptr) return;
  hipEvent_t e = nullptr;
  if (!AcquireEvent(&e).ok() || hipEventRecord(e, stream) != hipSuccess) {
    // no event to wait on: fall back to waiting for the stream itself
    (void)hipGetLastError();
    ReleaseEvent(e);
    (void)hipStreamSynchronize(stream);
    Free(ptr);
    return;
  }
  std::lock_guard<std::mutex> g(mu_);
  deferred_.emplace_back(e, ptr);
}

void Runtime::Reap(bool wait) {
  std::vector<std::pair<hipEvent_t, void*>> done;
  {
    std::lock_guard<std::mutex> g(mu_);
    if (deferred_.empty() && deferred_fns_.empty()) return;
    size_t keep = 0;
    for (size_t i = 0; i < deferred_.size(); i++) {
      hipError_t q = wait ? hipEventSynchronize(deferred_[i].first) : hipEventQuery(deferred_[i].first);
      if (q == hipSuccess) {
        done.push_back(deferred_[i]);
      } else {
        if (q != hipErrorNotReady) (void)hipGetLastError();
        deferred_[keep++] = deferred_[i];
      }
    }
    deferred_.resize(keep);
  }
  for (auto& d : done) {
    ReleaseEvent(d.first);
    Free(d.second);
  }
  std::vector<std::pair<hipEvent_t, std::function<void()>>> fns;
  {
    std::lock_guard<std::mutex> g(mu_);
    size_t keep = 0;
    for (size_t i = 0; i < deferred_fns_.size(); i++) {
      hipError_t q = wait ? hipEventSynchronize(deferred_fns_[i].first) : hipEventQuery(deferred_fns_[i].first);
      if (q == hipSuccess) {
        fns.push_back(std::move(deferred_fns_[i]));
      } else {
        if (q != hipErrorNotReady) (void)hipGetLastError();
        if (keep != i) deferred_fns_[keep] = std::move(deferred_fns_[i]);
        keep++;
      }
    }
    deferred_fns_.resize(keep);
  }
  for (auto& f : fns) {
    ReleaseEvent(f.first);
    f.second();
  }
}

Status Runtime::Alloc(size_t bytes, void** ptr) {
  GDV_RETURN_NOT_OK(EnsureDevice());
  Reap(false);
  size_t sz = RoundSize(bytes);
  {
    std::lock_guard<std::mutex> g(mu_);
    auto it = free_blocks_.find(sz);
    if (it != free_blocks_.end()) {
      *ptr = it->second;
      free_blocks_.erase(it);
      cached_bytes_ -= sz;
      live_blocks_[*ptr] = sz;
      return Status::OK();
    }
  }
  // Bounded run-ahead (round 6).  An asynchronous caller enqueues faster than the GPU completes: every call would find the
  // blocks of the calls before it still in flight and take NEW ones from the driver — unbounded memory, and a hipMalloc that
  // has to map fresh memory stalls the kernels that are running (C5's asynchronous Evaluate as the fourth workload of one
  // process: a 3-5 ms gap every 13-14 calls, profiles/r06_bench_step_outliers.txt).  With kMaxDeferredPerSize blocks of this
  // size already waiting for their streams, wait for the oldest of them instead and take that one.
  constexpr int kMaxDeferredPerSize = 8;
  {
    hipEvent_t oldest = nullptr;
    {
      std::lock_guard<std::mutex> g(mu_);
      int pending = 0;
      for (auto& d : deferred_) {
        auto it = live_blocks_.find(d.second);
        if (it != live_blocks_.end() && it->second == sz && pending++ == 0) oldest = d.first;
      }
      if (pending < kMaxDeferredPerSize) oldest = nullptr;
    }
    if (oldest != nullptr) {
      if (hipEventSynchronize(oldest) != hipSuccess) (void)hipGetLastError();
      Reap(false);
      std::lock_guard<std::mutex> g(mu_);
      auto it = free_blocks_.find(sz);
      if (it != free_blocks_.end()) {
        *ptr = it->second;
        free_blocks_.erase(it);
        cached_bytes_ -= sz;
        live_blocks_[*ptr] = sz;
        return Status::OK();
      }
    }
  }
  hipError_t e = hipMalloc(ptr, sz);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    Reap(true);
    TrimPool();
    e = hipMalloc(ptr, sz);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return Status::OutOfMemory("hipMalloc of " + std::to_string(sz) + " bytes failed");
  }
  std::lock_guard<std::mutex> g(mu_);
  live_blocks_[*ptr] = sz;
  return Status::OK();
}

Status Runtime::AcquirePinned(char** p) {
  GDV_RETURN_NOT_OK(EnsureDevice());
  Reap(false);
  {
    std::lock_guard<std::mutex> g(mu_);
    if (!pinned_free_.empty()) {
      *p = pinned_free_.back();
      pinned_free_.pop_back();
      return Status::OK();
    }
  }
  void* q = nullptr;
  if (hipHostMalloc(&q, kPinnedBlock, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return Status::OutOfMemory("hipHostMalloc of the pinned staging block failed");
  }
  *p = static_cast<char*>(q);
  return Status::OK();
}

void Runtime::ReleasePinned(char* p) {
  if (p == nullptr) return;
  std::lock_guard<std::mutex> g(mu_);
  if (pinned_free_.size() < 8) {
    pinned_free_.push_back(p);
  } else {
    (void)hipHostFree(p);
  }
}

void Runtime::Free(void* ptr) {
  if (!ptr) return;
  std::lock_guard<std::mutex> g(mu_);
  auto it = live_blocks_.find(ptr);
  if (it == live_blocks_.end()) return;
  size_t sz = it->second;
  live_blocks_.erase(it);
  // keep at most kPoolCapBytes of idle blocks (staging buffers of the host path can be tens
  // of GB); beyond that, blocks go straight back to the driver
  static const size_t cap = [] {
    const char* e = std::getenv("GDV_POOL_CAP_MB");
    return (e ? static_cast<size_t>(atoll(e)) : static_cast<size_t>(16384)) << 20;
  }();
  if (cached_bytes_ + sz > cap) {
    (void)hipFree(ptr);
    return;
  }
  free_blocks_.emplace(sz, ptr);
  cached_bytes_ += sz;
}

void Runtime::TrimPool() {
  std::multimap<size_t, void*> blocks;
  {
    std::lock_guard<std::mutex> g(mu_);
    blocks.swap(free_blocks_);
    cached_bytes_ = 0;
  }
  for (auto& b : blocks) (void)hipFree(b.second);
}

Status Runtime::AllOnesWord(const uint64_t** ptr) {
  GDV_RETURN_NOT_OK(EnsureDevice());
  std::lock_guard<std::mutex> g(mu_);
  if (all_ones_ == nullptr) {
    void* p = nullptr;
    GDV_HIP_RETURN_NOT_OK(hipMalloc(&p, 256));
    GDV_HIP_RETURN_NOT_OK(hipMemset(p, 0xff, 256));
    all_ones_ = static_cast<uint64_t*>(p);
  }
  *ptr = all_ones_;
  return Status::OK();
}

Status Runtime::LaunchMany(const CompiledKernel& k, int64_t grid_x, int64_t batches, int block,
                           const void* table_device, hipStream_t stream, bool small) {
  hipFunction_t fn = small ? k.function_small : k.function_many;
  if (fn == nullptr) return Status::ExecutionError("kernel has no multi-batch entry point");
  void* params[] = {&table_device};
  GDV_HIP_RETURN_NOT_OK(hipModuleLaunchKernel(fn, static_cast<unsigned>(grid_x),
                                              static_cast<unsigned>(batches), 1, static_cast<unsigned>(block), 1, 1,
                                              0, stream, params, nullptr));
  return Status::OK();
}

void Runtime::Defer(hipStream_t stream, std::function<void()> fn) {
  hipEvent_t e = nullptr;
  if (!AcquireEvent(&e).ok() || hipEventRecord(e, stream) != hipSuccess) {
    (void)hipGetLastError();
    ReleaseEvent(e);
    (void)hipStreamSynchronize(stream);
    fn();
    return;
  }
  std::lock_guard<std::mutex> g(mu_);
  deferred_fns_.emplace_back(e, std::move(fn));
}

Status Runtime::AcquireStream(hipStream_t* out) {
  GDV_RETURN_NOT_OK(EnsureDevice());
  {
    std::lock_guard<std::mutex> g(mu_);
    if (!streams_free_.empty()) {
      *out = streams_free_.back();
      streams_free_.pop_back();
      return Status::OK();
    }
  }
  GDV_HIP_RETURN_NOT_OK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return Status::OK();
}

void Runtime::ReleaseStream(hipStream_t s) {
  if (s == nullptr) return;
  std::lock_guard<std::mutex> g(mu_);
  if (streams_free_.size() < 16) streams_free_.push_back(s);
  else (void)hipStreamDestroy(s);
}

Status Runtime::AcquireEvent(hipEvent_t* out) {
  GDV_RETURN_NOT_OK(EnsureDevice());
  {
    std::lock_guard<std::mutex> g(mu_);
    if (!events_free_.empty()) {
      *out = events_free_.back();
      events_free_.pop_back();
      return Status::OK();
    }
  }
  GDV_HIP_RETURN_NOT_OK(hipEventCreateWithFlags(out, hipEventDisableTiming));
  return Status::OK();
}

void Runtime::ReleaseEvent(hipEvent_t e) {
  if (e == nullptr) return;
  std::lock_guard<std::mutex> g(mu_);
  if (events_free_.size() < 64) events_free_.push_back(e);
  else (void)hipEventDestroy(e);
}

Status Runtime::AcquirePinnedSmall(char** p) {
  GDV_RETURN_NOT_OK(EnsureDevice());
  Reap(false);
  {
    std::lock_guard<std::mutex> g(mu_);
    if (!pinned_small_free_.empty()) {
      *p = pinned_small_free_.back();
      pinned_small_free_.pop_back();
      return Status::OK();
    }
  }
  void* q = nullptr;
  if (hipHostMalloc(&q, kPinnedSmall, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return Status::OutOfMemory("hipHostMalloc of a pinned argument-table block failed");
  }
  *p = static_cast<char*>(q);
  return Status::OK();
}

void Runtime::ReleasePinnedSmall(char* p) {
  if (p == nullptr) return;
  std::lock_guard<std::mutex> g(mu_);
  if (pinned_small_free_.size() < 256) pinned_small_free_.push_back(p);
  else (void)hipHostFree(p);
}

Status Runtime::Launch(const CompiledKernel& k, int64_t grid, int block, const void* args,
                       size_t arg_bytes, hipStream_t stream, hipFunction_t entry) {
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args),
                    HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_bytes, HIP_LAUNCH_PARAM_END};
  GDV_HIP_RETURN_NOT_OK(hipModuleLaunchKernel(entry != nullptr ? entry : k.function, static_cast<unsigned>(grid), 1, 1,
                                              static_cast<unsigned>(block), 1, 1, 0, stream,
                                              nullptr, config));
  return Status::OK();
}

// ------------------------------------------------------------------ registered host memory

HostRegistry& HostRegistry::Get() {
  static HostRegistry r;
  return r;
}

Status HostRegistry::Insert(void* p, size_t bytes, bool owned) {
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess || dev == nullptr) {
    (void)hipGetLastError();
    return Status::ExecutionError("hipHostGetDevicePointer failed for a page-locked range");
  }
  std::unique_lock<std::shared_mutex> g(mu_);
  ranges_[reinterpret_cast<uintptr_t>(p)] = Range{bytes, static_cast<char*>(dev), owned};
  count_.store(static_cast<int>(ranges_.size()), std::memory_order_relaxed);
  return Status::OK();
}

Status HostRegistry::Register(void* p, size_t bytes) {
  if (p == nullptr || bytes == 0) return Status::Invalid("gdv_host_register: empty range");
  GDV_RETURN_NOT_OK(Runtime::Get().EnsureDevice());
  {
    std::shared_lock<std::shared_mutex> g(mu_);
    if (ranges_.count(reinterpret_cast<uintptr_t>(p))) return Status::Invalid("gdv_host_register: range already registered");
  }
  if (hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {
    (void)hipGetLastError();
    return Status::ExecutionError("hipHostRegister failed (" + std::to_string(bytes) + " bytes)");
  }
  Status s = Insert(p, bytes, false);
  if (!s.ok()) (void)hipHostUnregister(p);
  return s;
}

Status HostRegistry::Unregister(void* p) {
  {
    std::unique_lock<std::shared_mutex> g(mu_);
    auto it = ranges_.find(reinterpret_cast<uintptr_t>(p));
    if (it == ranges_.end() || it->second.owned) return Status::Invalid("gdv_host_unregister: not a registered range");
    ranges_.erase(it);
    count_.store(static_cast<int>(ranges_.size()), std::memory_order_relaxed);
  }
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return Status::ExecutionError("hipHostUnregister failed");
  }
  return Status::OK();
}

Status HostRegistry::Alloc(size_t bytes, void** p) {
  if (p == nullptr) return Status::Invalid("gdv_host_alloc: null output pointer");
  GDV_RETURN_NOT_OK(Runtime::Get().EnsureDevice());
  void* q = nullptr;
  if (hipHostMalloc(&q, std::max<size_t>(bytes, 8), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    return Status::OutOfMemory("hipHostMalloc failed (" + std::to_string(bytes) + " bytes)");
  }
  Status s = Insert(q, std::max<size_t>(bytes, 8), true);
  if (!s.ok()) {
    (void)hipHostFree(q);
    return s;
  }
  *p = q;
  return Status::OK();
}

Status HostRegistry::Free(void* p) {
  if (p == nullptr) return Status::OK();
  {
    std::unique_lock<std::shared_mutex> g(mu_);
    auto it = ranges_.find(reinterpret_cast<uintptr_t>(p));
    if (it == ranges_.end() || !it->second.owned) return Status::Invalid("gdv_host_free: not a gdv_host_alloc block");
    ranges_.erase(it);
    count_.store(static_cast<int>(ranges_.size()), std::memory_order_relaxed);
  }
  (void)hipHostFree(p);
  return Status::OK();
}

void* HostRegistry::View(const void* p, size_t bytes) const {
  if (p == nullptr || empty()) return nullptr;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uintptr_t lo = a & ~uintptr_t{7}, hi = (a + bytes + 7) & ~uintptr_t{7};
  std::shared_lock<std::shared_mutex> g(mu_);
  auto it = ranges_.upper_bound(a);
  if (it == ranges_.begin()) return nullptr;
  --it;
  const uintptr_t base = it->first, end = (base + it->second.size + 7) & ~uintptr_t{7};
  if (lo < (base & ~uintptr_t{7}) || hi > end) return nullptr;
  return it->second.dev + (a - base);
}

}  // namespace gdv
