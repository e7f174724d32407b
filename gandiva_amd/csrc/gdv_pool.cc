#include "gdv_pool.h"

#include <algorithm>
#include <string>

#include "gdv_kernels.h"

namespace gdv {

DevicePool::DevicePool() : device_(Runtime::SelectedDevice()), rt_(&Runtime::Get()) {}

DevicePool::~DevicePool() {
  if (rt_->EnsureDevice().ok()) {
    for (auto& f : free_) (void)hipFree(f.second);
    for (auto& l : live_) (void)hipFree(l.first);
  }
}

Status DevicePool::Raw(int64_t bytes, void** ptr) {
  GDV_RETURN_NOT_OK(rt_->EnsureDevice());
  hipError_t e = hipMalloc(ptr, static_cast<size_t>(std::max<int64_t>(bytes, 256)));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return Status::OutOfMemory("device pool: hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
  }
  return Status::OK();
}

Status DevicePool::Alloc(int64_t bytes, void** ptr) {
  if (bytes < 0 || ptr == nullptr) return Status::Invalid("device pool: bad argument");
  {
    std::lock_guard<std::mutex> g(mu_);
    auto it = free_.find(bytes);
    if (it != free_.end()) {
      *ptr = it->second;
      free_.erase(it);
      live_[*ptr] = bytes;
      return Status::OK();
    }
  }
  GDV_RETURN_NOT_OK(Raw(bytes, ptr));
  std::lock_guard<std::mutex> g(mu_);
  live_[*ptr] = bytes;
  return Status::OK();
}

Status DevicePool::Free(void* ptr) {
  if (ptr == nullptr) return Status::OK();
  std::lock_guard<std::mutex> g(mu_);
  auto it = live_.find(ptr);
  if (it == live_.end()) return Status::Invalid("device pool: pointer does not belong to this pool");
  free_.emplace(it->second, ptr);
  live_.erase(it);
  return Status::OK();
}

Status DevicePool::Trim() {
  GDV_RETURN_NOT_OK(rt_->EnsureDevice());
  std::lock_guard<std::mutex> g(mu_);
  for (auto& f : free_) (void)hipFree(f.second);
  free_.clear();
  return Status::OK();
}

int64_t DevicePool::bytes_held(int64_t* in_use) const {
  std::lock_guard<std::mutex> g(mu_);
  int64_t used = 0, idle = 0;
  for (auto& l : live_) used += l.second;
  for (auto& f : free_) idle += f.first;
  if (in_use) *in_use = used;
  return used + idle;
}

Status DevicePool::ReserveSet(int count, int64_t bytes, int candidates, void** ptrs, double* rates, int* tried, int* kept) {
  if (count < 1 || count > 32 || bytes < 0 || ptrs == nullptr) return Status::Invalid("device pool: 1..32 buffers per set");
  GDV_RETURN_NOT_OK(rt_->EnsureDevice());
  if (tried) *tried = 0;
  if (kept) *kept = 0;
  // a set that is already retained (an earlier ReserveSet's, handed back with Free) is the known-good one: no search
  {
    std::lock_guard<std::mutex> g(mu_);
    if (static_cast<int>(free_.count(bytes)) >= count) {
      for (int i = 0; i < count; i++) {
        auto it = free_.find(bytes);
        ptrs[i] = it->second;
        free_.erase(it);
        live_[ptrs[i]] = bytes;
      }
      return Status::OK();
    }
  }
  candidates = std::max(1, candidates);
  // small sets are not worth a search (the sweep needs at least a few MiB per buffer to say anything)
  if (bytes < (int64_t{64} << 20)) candidates = 1;
  using Set = std::vector<void*>;
  auto release = [](Set* s) {
    for (void* p : *s) (void)hipFree(p);
    s->clear();
  };
  auto allocate = [&](Set* s) -> Status {
    for (int i = 0; i < count; i++) {
      void* p = nullptr;
      Status st = Raw(bytes, &p);
      if (!st.ok()) {
        release(s);
        return st;
      }
      s->push_back(p);
    }
    return Status::OK();
  };
  const int64_t need = static_cast<int64_t>(count) * bytes;
  Set best, loser;
  double best_rate = -1;
  int n = 0;
  for (int c = 0; c < candidates; c++) {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (static_cast<int64_t>(free_b) < need + (int64_t{2} << 30)) {
      if (loser.empty()) break;  // not even one more placement fits next to the best one
      release(&loser);           // (its pages may come straight back: still a candidate)
      (void)hipMemGetInfo(&free_b, &total_b);
      if (static_cast<int64_t>(free_b) < need + (int64_t{2} << 30)) break;
    }
    Set cand;
    Status st = allocate(&cand);
    if (!st.ok()) {
      if (best.empty()) { release(&loser); return st; }
      break;
    }
    release(&loser);  // the previous loser stayed until this candidate existed: the driver could not hand its pages back
    double rate = 0;
    if (candidates > 1) {
      hipError_t e = MeasureWriteSet(cand.data(), count, static_cast<size_t>(bytes), rt_->num_cus(), &rate);
      if (e != hipSuccess) {
        release(&cand);
        release(&best);
        return Status::ExecutionError(std::string("device pool: placement probe failed: ") + hipGetErrorString(e));
      }
    }
    if (rates) rates[n] = rate;
    if (rate > best_rate) {
      loser.swap(best);
      best.swap(cand);
      best_rate = rate;
      if (kept) *kept = n;
    } else {
      loser.swap(cand);
    }
    n++;
  }
  release(&loser);
  if (best.empty()) return Status::OutOfMemory("device pool: no room for " + std::to_string(need) + " bytes");
  if (tried) *tried = n;
  std::lock_guard<std::mutex> g(mu_);
  for (int i = 0; i < count; i++) {
    ptrs[i] = best[i];
    live_[best[i]] = bytes;
  }
  return Status::OK();
}

}  // namespace gdv
