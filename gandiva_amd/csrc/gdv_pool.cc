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
  // What is slow is a set whose members are NEIGHBOURS in the driver's allocation order (one region of the HBM address
  // map: ten such columns run C2 at 5.1 .. 6.4 ms depending on the region, whatever the spacing between them —
  // profiles/r06_placement_stagger.txt); members SPREAD over a wide span of allocations run at 4.81 .. 5.36, strided picks
  // at 4.91 .. 4.93 (profiles/r06_placement_subsets.txt), and how fast a buffer is alone says nothing about the set
  // (profiles/r06_placement_map.txt).  So: allocate `spread` times the set as single buffers, take every spread-th one
  // (candidate k = buffers k, k + spread, k + 2 spread ...), probe the candidates with the write sweep, keep the best,
  // give everything else back.
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const int64_t room = (static_cast<int64_t>(free_b) - (int64_t{4} << 30)) / std::max<int64_t>(bytes, 1);
  int spread = candidates > 1 ? 4 : 1;
  while (spread > 1 && static_cast<int64_t>(count) * spread > room) spread--;
  std::vector<void*> all;
  auto release_all = [&](const std::vector<void*>& keep) {
    for (void* p : all)
      if (std::find(keep.begin(), keep.end(), p) == keep.end()) (void)hipFree(p);
    all.clear();
  };
  for (int i = 0; i < count * spread; i++) {
    void* p = nullptr;
    Status st = Raw(bytes, &p);
    if (!st.ok()) {
      if (static_cast<int>(all.size()) >= count) break;  // (less spread than hoped for)
      release_all({});
      return st;
    }
    all.push_back(p);
  }
  spread = static_cast<int>(all.size()) / count;
  const int ncand = std::min(candidates, spread);
  std::vector<void*> best;
  double best_rate = -1;
  for (int k = 0; k < ncand; k++) {
    std::vector<void*> cand;
    for (int i = 0; i < count; i++) cand.push_back(all[static_cast<size_t>(k + i * spread)]);
    double rate = 0;
    if (ncand > 1) {
      hipError_t e = MeasureWriteSet(cand.data(), count, static_cast<size_t>(bytes), rt_->num_cus(), &rate);
      if (e != hipSuccess) {
        release_all({});
        return Status::ExecutionError(std::string("device pool: placement probe failed: ") + hipGetErrorString(e));
      }
    }
    if (rates) rates[k] = rate;
    if (rate > best_rate) {
      best_rate = rate;
      best = cand;
      if (kept) *kept = k;
    }
  }
  if (tried) *tried = ncand;
  release_all(best);
  std::lock_guard<std::mutex> g(mu_);
  for (int i = 0; i < count; i++) {
    ptrs[i] = best[i];
    live_[best[i]] = bytes;
  }
  return Status::OK();
}

}  // namespace gdv
