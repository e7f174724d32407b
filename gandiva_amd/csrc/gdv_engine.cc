#include "gdv_engine.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <list>
#include <unordered_map>

#include "gdv_kernels.h"

namespace gdv {

namespace {

// ------------------------------------------------------------------ LRU cache of built modules
// (the reference keeps a process-wide, mutex-guarded LRU of compiled modules keyed on
// schema + expressions + configuration: SURVEY.md §2 row 12)
template <typename T>
class LruCache {
 public:
  explicit LruCache(size_t cap) : cap_(cap) {}
  std::shared_ptr<T> Get(const std::string& key) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = map_.find(key);
    if (it == map_.end()) return nullptr;
    order_.splice(order_.begin(), order_, it->second.second);
    return it->second.first;
  }
  void Put(const std::string& key, std::shared_ptr<T> v) {
    std::lock_guard<std::mutex> g(mu_);
    if (map_.count(key)) return;
    order_.push_front(key);
    map_[key] = {std::move(v), order_.begin()};
    if (map_.size() > cap_) {
      map_.erase(order_.back());
      order_.pop_back();
    }
  }

 private:
  size_t cap_;
  std::mutex mu_;
  std::list<std::string> order_;
  std::unordered_map<std::string,
                     std::pair<std::shared_ptr<T>, std::list<std::string>::iterator>>
      map_;
};

std::string SchemaKey(const Schema& s) {
  std::string k;
  for (auto& f : s) k += std::to_string(f.name.size()) + ":" + f.name + ":" + f.type.ToString() + ";";
  return k;
}

// ------------------------------------------------------------------ argument block

struct HostBitmap {
  const uint64_t* p = nullptr;
  int32_t shift = 0;
  int32_t pad = 0;
  int64_t nwords = 0;
};
static_assert(sizeof(HostBitmap) == 24, "must match struct gdv_bitmap in gdv_device_lib.hpp");

class ArgBlock {
 public:
  explicit ArgBlock(const ArgLayout& l) : layout_(l), buf_(l.total(), 0) {}
  void Set64(int off, uint64_t v) { std::memcpy(&buf_[off], &v, 8); }
  uint64_t Get64(int off) const { uint64_t v; std::memcpy(&v, &buf_[off], 8); return v; }
  void SetPtr(int off, const void* p) { Set64(off, reinterpret_cast<uint64_t>(p)); }
  void SetInData(int k, const void* p) { SetPtr(layout_.in_base() + k * ArgLayout::kInStride, p); }
  void SetInValid(int k, const HostBitmap& b) {
    std::memcpy(&buf_[layout_.in_base() + k * ArgLayout::kInStride + 8], &b, 24);
  }
  void SetInBits(int k, const HostBitmap& b) {
    std::memcpy(&buf_[layout_.in_base() + k * ArgLayout::kInStride + 32], &b, 24);
  }
  void SetInOffsets(int k, const void* p) {
    SetPtr(layout_.in_base() + k * ArgLayout::kInStride + 56, p);
  }
  void SetOutData(int e, void* p) { SetPtr(layout_.out_base() + e * ArgLayout::kOutStride, p); }
  void SetOutValid(int e, void* p) {
    SetPtr(layout_.out_base() + e * ArgLayout::kOutStride + 8, p);
  }
  void SetOutOffsets(int e, void* p) {
    SetPtr(layout_.out_base() + e * ArgLayout::kOutStride + 16, p);
  }
  void SetLit(int i, uint64_t v) { Set64(layout_.lit_base() + i * 8, v); }
  // input slot k := slot `src_k` of another block (same column, already bound / staged there)
  void CopyInSlot(int k, const ArgBlock& src, int src_k) {
    std::memcpy(&buf_[layout_.in_base() + k * ArgLayout::kInStride],
                &src.buf_[src.layout_.in_base() + src_k * ArgLayout::kInStride], ArgLayout::kInStride);
  }
  void SetOutCap(int e, int64_t bytes) {
    Set64(layout_.out_base() + e * ArgLayout::kOutStride + 24, static_cast<uint64_t>(bytes));
  }
  // input slot k moved forward by `rows` rows (a multiple of 64).  width > 0: fixed-width values;
  // 0: bool values (a bitmap); -1: var-len (offsets move, the byte buffer stays)
  void AdvanceInSlot(int k, int64_t rows, int width) {
    const int base = layout_.in_base() + k * ArgLayout::kInStride;
    auto bump_ptr = [&](int off, int64_t bytes) {
      uint64_t p;
      std::memcpy(&p, &buf_[off], 8);
      if (p != 0) p += static_cast<uint64_t>(bytes);
      std::memcpy(&buf_[off], &p, 8);
    };
    auto bump_bitmap = [&](int off) {
      HostBitmap b;
      std::memcpy(&b, &buf_[off], 24);
      if (b.p != nullptr && b.nwords > 1) {  // (nwords == 1: the all-ones word, index clamped)
        b.p += rows / 64;
        b.nwords = std::max<int64_t>(b.nwords - rows / 64, 1);
      }
      std::memcpy(&buf_[off], &b, 24);
    };
    if (width > 0) bump_ptr(base, rows * width);
    bump_bitmap(base + 8);
    if (width == 0) bump_bitmap(base + 32);
    if (width < 0) bump_ptr(base + 56, rows * 4);
  }
  const void* data() const { return buf_.data(); }
  size_t size() const { return buf_.size(); }

 private:
  ArgLayout layout_;
  std::vector<char> buf_;
};

// Folds buffer misalignment and the Arrow array offset into (8-byte aligned word pointer,
// shift < 64, readable words).  The buffer must be readable up to the next 8-byte boundary
// (Arrow pads buffers to 64 bytes: pyarrow/include/arrow/type_fwd.h:759).
HostBitmap FoldBitmap(const void* ptr, int64_t size, int64_t bit_offset) {
  HostBitmap b;
  if (ptr == nullptr) return b;
  uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  uintptr_t aligned = a & ~uintptr_t(7);
  int64_t lead = static_cast<int64_t>(a & 7) * 8 + bit_offset;
  uintptr_t p = aligned + static_cast<uintptr_t>(lead >> 6) * 8;
  b.p = reinterpret_cast<const uint64_t*>(p);
  b.shift = static_cast<int32_t>(lead & 63);
  int64_t bytes = static_cast<int64_t>(a + size - p);
  b.nwords = bytes > 0 ? (bytes + 7) / 8 : 0;
  return b;
}

int64_t BytesForBits(int64_t bits) { return (bits + 7) / 8; }

// Error paths must not hand staging blocks back to the pool while copies or kernels that
// use them are still queued: declared AFTER the Staging object, this drains the stream first.
struct StreamDrain {
  hipStream_t stream;
  bool armed;
  ~StreamDrain() {
    if (armed) (void)hipStreamSynchronize(stream);
  }
};

// Debug switches of the evaluation path, read from the environment ONCE per process (first use): no
// getenv is reachable from Evaluate (round-3 verdict: a getenv per call on a path that takes 0.6-7 us
// per batch, and not safe against a concurrent setenv).  Code-generation switches are read at Make
// (CodegenOptions::FromEnv).
struct EngineKnobs {
  bool trace = false;              // GDV_TRACE: one line per Evaluate on stderr
  bool no_optflat = false;         // GDV_NO_OPTFLAT: var-len plans go straight to the general kernel
  bool no_evaluate_many = false;   // GDV_NO_EVALUATE_MANY: multi-batch calls run batch by batch
  bool no_small_filter = false;    // GDV_NO_SMALL_FILTER: default of Filter "small_filter" tuning (read at Make)
  int filter_chunks = 1;           // GDV_FILTER_CHUNKS: default of Filter "chunks" tuning (read at Make)
  int grid_mult = 0;               // GDV_GRID_MULT: workgroups per CU of the grid-stride launch (0: default)
  bool fp_window_only = false;     // GDV_FP_WINDOW_ONLY: fused filter-project never moves to its direct kernel (tests, sweeps)
  bool fp_force_stall = false;     // GDV_FP_FORCE_STALL: treat every fused launch as stalled (exercises the chain re-run)
  bool no_tier0 = false;           // GDV_NO_TIER0: Make waits for the specialised kernel as before round 6
  bool force_tier0 = false;        // GDV_FORCE_TIER0: every plan that has a tier-0 program runs on it, always (tests)
  static const EngineKnobs& Get() {
    static const EngineKnobs k = [] {
      EngineKnobs x;
      x.trace = std::getenv("GDV_TRACE") != nullptr;
      x.no_optflat = std::getenv("GDV_NO_OPTFLAT") != nullptr;
      x.no_evaluate_many = std::getenv("GDV_NO_EVALUATE_MANY") != nullptr;
      x.no_small_filter = std::getenv("GDV_NO_SMALL_FILTER") != nullptr;
      x.fp_window_only = std::getenv("GDV_FP_WINDOW_ONLY") != nullptr;
      x.fp_force_stall = std::getenv("GDV_FP_FORCE_STALL") != nullptr;
      x.no_tier0 = std::getenv("GDV_NO_TIER0") != nullptr;
      x.force_tier0 = std::getenv("GDV_FORCE_TIER0") != nullptr;
      if (const char* s = std::getenv("GDV_GRID_MULT")) x.grid_mult = std::max(1, atoi(s));
      if (const char* s = std::getenv("GDV_FILTER_CHUNKS")) x.filter_chunks = std::max(1, std::min(64, atoi(s)));
      return x;
    }();
    return k;
  }
};

// GDV_TRACE=1: one line per Evaluate on stderr (kind, kernel, rows, device time between two
// HIP events on the launch stream, rows/s).  The reference has no tracing of its own
// (SURVEY.md §5); this is the hook its micro-benchmarks' std::chrono timers stood in for.
// Tracing synchronises the stream, so it also serialises asynchronous evaluations.
class EvalTrace {
 public:
  EvalTrace(const char* kind, const std::string& kernel, int64_t rows, hipStream_t stream)
      : kind_(kind), kernel_(kernel), rows_(rows), stream_(stream) {
    on_ = EngineKnobs::Get().trace;
    if (on_ && hipEventCreate(&t0_) == hipSuccess && hipEventCreate(&t1_) == hipSuccess) {
      (void)hipEventRecord(t0_, stream_);
    } else {
      on_ = false;
    }
  }
  ~EvalTrace() {
    if (!on_) return;
    (void)hipEventRecord(t1_, stream_);
    (void)hipEventSynchronize(t1_);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, t0_, t1_);
    fprintf(stderr, "[gdv] %s %s rows=%lld device_ms=%.4f Mrows/s=%.1f\n", kind_, kernel_.c_str(),
            static_cast<long long>(rows_), ms, ms > 0 ? rows_ / (ms * 1e3) : 0.0);
    (void)hipEventDestroy(t0_);
    (void)hipEventDestroy(t1_);
  }

 private:
  const char* kind_;
  std::string kernel_;
  int64_t rows_;
  hipStream_t stream_;
  bool on_ = false;
  hipEvent_t t0_ = nullptr, t1_ = nullptr;
};

// Host-buffer path.  Large batches: one device buffer and one copy per Arrow buffer (the
// copies are PCIe-bound anyway).  Small batches (<= kPackRows rows, while they fit the
// block): every staged input and every fixed-size output shares ONE device block mirrored by
// ONE pinned host block — one H2D before the launches, one D2H after them — because a
// pageable hipMemcpyAsync costs 10-25 us however small it is and a ten-expression projection
// would issue ~30 of them (C2 at 1024 rows: 397 -> 74 us per Evaluate).
struct Staging {
  static constexpr int64_t kPackRows = 131072;
  std::deque<DeviceBuffer> buffers;  // deque: references stay valid across Add()
  DeviceBuffer& Add() {
    buffers.emplace_back();
    return buffers.back();
  }
  ~Staging() {
    if (pin_ != nullptr) Runtime::Get().ReleasePinned(pin_);
  }

  Status EnablePacked() {
    GDV_RETURN_NOT_OK(Runtime::Get().AcquirePinned(&pin_));
    GDV_RETURN_NOT_OK(block_.Allocate(Runtime::kPinnedBlock));
    packed_ = true;
    return Status::OK();
  }

  // device copy of n host bytes, readable (zero-filled) up to `alloc` bytes
  Status In(const void* src, size_t n, size_t alloc, hipStream_t stream, void** dev) {
    if (alloc < n) alloc = n;
    HostRegistry::StagedBytes().fetch_add(static_cast<int64_t>(n), std::memory_order_relaxed);
    size_t off = 0;
    if (packed_ && !flushed_ && Reserve(alloc, &off)) {
      if (n > 0) std::memcpy(pin_ + off, src, n);
      if (alloc > n) std::memset(pin_ + off + n, 0, alloc - n);
      *dev = block_.as<char>() + off;
      in_end_ = used_;
      return Status::OK();
    }
    DeviceBuffer& d = Add();
    GDV_RETURN_NOT_OK(d.Allocate(std::max<size_t>(alloc, 8)));
    if (alloc > n) GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(d.get(), 0, alloc, stream));
    if (n > 0) GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(d.get(), src, n, hipMemcpyHostToDevice, stream));
    *dev = d.get();
    return Status::OK();
  }
  // all In() regions -> device with one copy; call once, before the first launch
  Status FlushIn(hipStream_t stream) {
    flushed_ = true;
    if (packed_ && in_end_ > 0)
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(block_.get(), pin_, in_end_, hipMemcpyHostToDevice, stream));
    return Status::OK();
  }
  // device region of `alloc` bytes whose first `copy` bytes FetchOut/Deliver bring to `user`
  Status Out(size_t alloc, size_t copy, void* user, void** dev) {
    size_t off = 0;
    OutCopy oc{user, nullptr, 0, copy, false};
    HostRegistry::StagedBytes().fetch_add(static_cast<int64_t>(copy), std::memory_order_relaxed);
    if (packed_ && Reserve(alloc, &off)) {
      oc.off = off;
      oc.packed = true;
      *dev = block_.as<char>() + off;
    } else {
      DeviceBuffer& d = Add();
      GDV_RETURN_NOT_OK(d.Allocate(std::max<size_t>(alloc, 8)));
      oc.dev = d.get();
      *dev = d.get();
    }
    outs_.push_back(oc);
    return Status::OK();
  }
  Status FetchOut(hipStream_t stream) {
    size_t lo = used_, hi = 0;
    for (auto& o : outs_) {
      if (o.n == 0) continue;
      if (o.packed) {
        lo = std::min(lo, o.off);
        hi = std::max(hi, o.off + o.n);
      } else {
        GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(o.user, o.dev, o.n, hipMemcpyDeviceToHost, stream));
      }
    }
    if (hi > lo)
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(pin_ + lo, block_.as<char>() + lo, hi - lo,
                                           hipMemcpyDeviceToHost, stream));
    return Status::OK();
  }
  void Deliver() {  // after the stream is drained
    for (auto& o : outs_)
      if (o.packed && o.n > 0) std::memcpy(o.user, pin_ + o.off, o.n);
  }

 private:
  struct OutCopy {
    void* user;
    void* dev;
    size_t off, n;
    bool packed;
  };
  bool Reserve(size_t bytes, size_t* off) {
    const size_t at = (used_ + 255) & ~size_t{255};
    if (at + bytes > Runtime::kPinnedBlock) return false;
    *off = at;
    used_ = at + bytes;
    return true;
  }
  bool packed_ = false, flushed_ = false;
  DeviceBuffer block_;
  char* pin_ = nullptr;
  size_t used_ = 0, in_end_ = 0;
  std::vector<OutCopy> outs_;
};

// host bitmap bytes covering bits [off, off+rows) -> zero-padded device words
Status StageBitmap(const void* host, int64_t off, int64_t rows, hipStream_t stream,
                   Staging* st, HostBitmap* out) {
  const int64_t first = off / 8;
  const int64_t len = BytesForBits(off + rows) - first;
  const int64_t words = (len + 7) / 8 + 1;
  void* dev = nullptr;
  GDV_RETURN_NOT_OK(st->In(static_cast<const char*>(host) + first, len, words * 8, stream, &dev));
  out->p = static_cast<const uint64_t*>(dev);
  out->shift = static_cast<int32_t>(off % 8);
  out->nwords = words;
  return Status::OK();
}

Status BindInputs(const KernelPlan& plan, const Schema& schema, const ColumnBuffers* cols,
                  int num_cols, int64_t batch_rows, MemKind mem, hipStream_t stream,
                  ArgBlock* args, Staging* st, int64_t compact_rows = -1) {
  if (num_cols != static_cast<int>(schema.size()))
    return Status::Invalid("number of columns in batch (" + std::to_string(num_cols) +
                           ") does not match the schema (" + std::to_string(schema.size()) + ")");
  for (size_t k = 0; k < plan.input_fields.size(); k++) {
    const int idx = plan.input_fields[k];
    // (temporaries of a selection-mode first stage hold one row per slot, not per batch row)
    const int64_t num_rows = (idx >= plan.compact_from && compact_rows >= 0) ? compact_rows : batch_rows;
    const ColumnBuffers& c = cols[idx];
    const DataType& t = schema[idx].type;
    const std::string& name = schema[idx].name;
    if (plan.input_needs_values[k]) {
      if (c.data == nullptr && num_rows > 0 && !t.is_varlen())
        return Status::Invalid("column '" + name + "' has no data buffer");
      if (t.is_varlen()) {
        // int32 offsets (rows + 1 of them, from the array offset) + the whole byte buffer
        const int64_t need = (c.offset + num_rows + 1) * 4;
        if (c.offsets == nullptr || c.offsets_size < need)
          return Status::Invalid("column '" + name + "': offsets buffer too small");
        const char* osrc = static_cast<const char*>(c.offsets) + c.offset * 4;
        if (mem == MemKind::kHost) {
          // Round 5: offsets and bytes that lie in REGISTERED host memory (gdv_host_register / gdv_host_alloc) are
          // read in place over the fabric, like fixed-width columns since round 4.  The bytes need their 16-byte
          // granule behind the last byte inside the registered range as well (the sweep reads whole pieces: Arrow's
          // zeroed 64-byte padding covers it; a buffer that ends flush with its range is staged as before).
          void* dof = (reinterpret_cast<uintptr_t>(osrc) & 3) == 0 ? HostRegistry::Get().View(osrc, (num_rows + 1) * 4) : nullptr;
          void* dd = c.data_size >= 8 ? HostRegistry::Get().View(c.data, c.data_size + 16) : nullptr;
          if (dd != nullptr) {
            // (round 6: the sweep reads the last piece up to its 16-byte boundary; the staged copy zeroes what lies behind
            // the last byte, a caller's own registered buffer need not — a stale byte >= 0x80 there would send an ASCII
            // batch to the exact string kernels for nothing.  Host memory: look, and stage when the tail is not clean.)
            const unsigned char* end = static_cast<const unsigned char*>(c.data) + c.data_size;
            const unsigned char* stop = reinterpret_cast<const unsigned char*>((reinterpret_cast<uintptr_t>(end) + 15) & ~uintptr_t{15});
            for (const unsigned char* q = end; q < stop; q++)
              if (*q & 0x80) { dd = nullptr; break; }
          }
          if (dof == nullptr) GDV_RETURN_NOT_OK(st->In(osrc, (num_rows + 1) * 4, (num_rows + 1) * 4, stream, &dof));
          // (16 zero bytes behind the last byte: the byte sweep reads whole 16-byte pieces, and whatever
          // the block held before must not look like a byte >= 0x80 — it would send an ASCII batch to
          // the exact variant of the string kernels for nothing)
          if (dd == nullptr) GDV_RETURN_NOT_OK(st->In(c.data, c.data_size, c.data_size + 16, stream, &dd));
          args->SetInOffsets(static_cast<int>(k), dof);
          args->SetInData(static_cast<int>(k), dd);
        } else if (c.data_size < 8) {
          // the kernels' 8-byte loads need 8 readable bytes ending at the limit: a tiny
          // buffer is copied into a zero-padded one
          DeviceBuffer& dd = st->Add();
          GDV_RETURN_NOT_OK(dd.Allocate(8));
          GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(dd.get(), 0, 8, stream));
          if (c.data_size > 0)
            GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(dd.get(), c.data, c.data_size,
                                                 hipMemcpyDeviceToDevice, stream));
          args->SetInOffsets(static_cast<int>(k), osrc);
          args->SetInData(static_cast<int>(k), dd.get());
        } else {
          args->SetInOffsets(static_cast<int>(k), osrc);
          args->SetInData(static_cast<int>(k), c.data);
        }
        // readable extent of the byte buffer (the kernels' 8-byte loads stop at this limit)
        HostBitmap extent;
        extent.nwords = std::max<int64_t>(c.data_size, 8);
        args->SetInBits(static_cast<int>(k), extent);
      } else if (t.id == kBool) {
        if (c.data_size < BytesForBits(c.offset + num_rows))
          return Status::Invalid("column '" + name + "': data buffer too small");
        HostBitmap b;
        if (mem == MemKind::kHost) {
          // (a buffer inside a registered host range is read in place: gdv_host_register)
          if (const void* v = HostRegistry::Get().View(c.data, c.data_size)) b = FoldBitmap(v, c.data_size, c.offset);
          else GDV_RETURN_NOT_OK(StageBitmap(c.data, c.offset, num_rows, stream, st, &b));
        } else {
          b = FoldBitmap(c.data, c.data_size, c.offset);
        }
        args->SetInBits(static_cast<int>(k), b);
      } else {
        const int w = t.byte_width();
        if (c.data_size < (c.offset + num_rows) * w)
          return Status::Invalid("column '" + name + "': data buffer too small (" +
                                 std::to_string(c.data_size) + " bytes for " +
                                 std::to_string(c.offset + num_rows) + " rows)");
        const char* src = static_cast<const char*>(c.data) + c.offset * w;
        if (mem == MemKind::kHost) {
          void* d = HostRegistry::Get().View(src, num_rows * w);
          if (d == nullptr) GDV_RETURN_NOT_OK(st->In(src, num_rows * w, num_rows * w, stream, &d));
          args->SetInData(static_cast<int>(k), d);
        } else {
          args->SetInData(static_cast<int>(k), src);
        }
      }
    }
    if (plan.input_needs_validity[k]) {
      HostBitmap b;
      if (c.validity == nullptr) {
        // no validity buffer = no nulls: bind the one-word all-ones bitmap (index clamped)
        GDV_RETURN_NOT_OK(Runtime::Get().AllOnesWord(&b.p));
        b.shift = 0;
        b.nwords = 1;
      } else {
        if (c.validity_size < BytesForBits(c.offset + num_rows))
          return Status::Invalid("column '" + name + "': validity buffer too small");
        if (mem == MemKind::kHost) {
          if (const void* v = HostRegistry::Get().View(c.validity, c.validity_size)) b = FoldBitmap(v, c.validity_size, c.offset);
          else GDV_RETURN_NOT_OK(StageBitmap(c.validity, c.offset, num_rows, stream, st, &b));
        } else {
          b = FoldBitmap(c.validity, c.validity_size, c.offset);
        }
      }
      args->SetInValid(static_cast<int>(k), b);
    }
  }
  return Status::OK();
}

int64_t GridFor(const KernelPlan& plan, int64_t rows) {
  if (plan.wave_tiles) {
    // wave shape: one WAVE per tile, no scanner workgroup
    const int64_t nwt = (rows + plan.rows_per_tile() - 1) / plan.rows_per_tile();
    return std::max<int64_t>(1, (nwt + plan.opts.waves - 1) / plan.opts.waves);
  }
  if (plan.string_skeleton) {
    // one workgroup per tile (+ the scanner workgroup when there are var-len outputs)
    const int64_t ntiles = (rows + plan.rows_per_tile() - 1) / plan.rows_per_tile();
    return std::max<int64_t>(1, ntiles) + (plan.num_varlen_outputs > 0 ? 1 : 0);
  }
  const int64_t nwords = (rows + 63) / 64;
  const int64_t per_tile = static_cast<int64_t>(plan.opts.subtiles) * plan.opts.waves;
  int64_t ntiles = (nwords + per_tile - 1) / per_tile;
  int blocks_per_cu = std::max(1, 32 / plan.opts.waves);
  // String work per tile varies with the data (lengths, divergent per-row loops): four times
  // as many, smaller shares of the grid-stride loop even out the tail (C5: 2.26 -> 2.01 ms;
  // fixed-width plans measured best at the base value)
  if (plan.has_varlen_input || plan.has_varlen_output) blocks_per_cu *= 4;
  // (read-dominated 16-sub-tile plans, PlanProjectorShape / PlanFilter: fewer resident workgroups stream better — once every
  // workgroup has a long loop to run; a batch of a few tiles per CU is over sooner with all of them in flight)
  else if (plan.grid_blocks_per_cu > 0 && ntiles >= 64LL * Runtime::Get().num_cus()) blocks_per_cu = plan.grid_blocks_per_cu;
  if (EngineKnobs::Get().grid_mult > 0) blocks_per_cu = EngineKnobs::Get().grid_mult;
  int64_t cap = static_cast<int64_t>(Runtime::Get().num_cus()) * blocks_per_cu;
  return std::max<int64_t>(1, std::min(ntiles, cap));
}

constexpr uint32_t kErrStall = 8u;  // GDV_ERR_STALL (gdv_device_lib.hpp): a look-back / scanner hand-off gave up

std::string ErrorMessage(uint32_t bits) {
  std::string m;
  if (bits & 1u) m += "divide by zero error";
  if (bits & 2u) m += (m.empty() ? "" : "; ") + std::string("overflow");
  if (bits & 4u) m += (m.empty() ? "" : "; ") + std::string("invalid argument");
  if (bits & 8u) m += (m.empty() ? "" : "; ") + std::string("device scan stalled");
  if (bits & 48u) m += (m.empty() ? "" : "; ") + std::string("internal: optimistic var-len kernel was not re-run");
  if (bits & 128u) m += (m.empty() ? "" : "; ") + std::string("asynchronous two-stage evaluation: a temporary was too small");
  return m.empty() ? "execution error" : m;
}

// String literals, LIKE patterns and IN tables of a plan live in one small device block that the
// kernel reaches through gdv_args::aux0 (uploaded once, at Make).
Status UploadConstBlock(const KernelPlan& plan, DeviceBuffer* out) {
  if (plan.const_block.empty()) return Status::OK();
  GDV_RETURN_NOT_OK(out->Allocate(plan.const_block.size() + 16));
  GDV_HIP_RETURN_NOT_OK(hipMemcpy(out->get(), plan.const_block.data(), plan.const_block.size(), hipMemcpyHostToDevice));
  return Status::OK();
}

void BindLiterals(const KernelPlan& plan, const DeviceBuffer& consts, ArgBlock* args) {
  for (size_t i = 0; i < plan.literals.size(); i++) args->SetLit(static_cast<int>(i), plan.literals[i]);
  args->SetPtr(ArgLayout::kOffAux0, consts.get());
}

}  // namespace

Status PlanDeviceStates::Get(const KernelPlan& plan, const PlanDeviceState** out, bool need_kernel) const {
  Runtime& rt = Runtime::Get();
  const int id = rt.id();
  PlanDeviceState* have = slots_[id].load(std::memory_order_acquire);
  if (have != nullptr && (!need_kernel || have->kernel.load(std::memory_order_acquire) != nullptr)) {
    *out = have;
    return Status::OK();
  }
  std::lock_guard<std::mutex> g(mu_);
  have = slots_[id].load(std::memory_order_acquire);
  if (have == nullptr) {
    std::unique_ptr<PlanDeviceState> st(new PlanDeviceState);
    GDV_RETURN_NOT_OK(UploadConstBlock(plan, &st->consts));
    if (plan.prepass) GDV_RETURN_NOT_OK(UploadConstBlock(*plan.prepass, &st->consts_pre));
    have = st.release();
    slots_[id].store(have, std::memory_order_release);
  }
  if (need_kernel && have->kernel.load(std::memory_order_acquire) == nullptr) {
    if (plan.prepass) GDV_RETURN_NOT_OK(rt.GetKernel(plan.prepass->source, plan.prepass->kernel_name, &have->kernel_pre));
    const CompiledKernel* k = nullptr;
    GDV_RETURN_NOT_OK(rt.GetKernel(plan.source, plan.kernel_name, &k));
    have->kernel.store(k, std::memory_order_release);
  }
  *out = have;
  return Status::OK();
}

namespace {

LruCache<Projector>& ProjectorCache() {
  static LruCache<Projector> c(500);
  return c;
}
LruCache<Filter>& FilterCache() {
  static LruCache<Filter> c(500);
  return c;
}

}  // namespace

// ------------------------------------------------------------------ two-stage plans

Status StageColumns::Run(const Projector& pre, int64_t batch_rows, const ColumnBuffers* in, int num_cols,
                         MemKind mem, hipStream_t stream, const SelectionView* sel,
                         std::vector<std::atomic<int64_t>>* hints) {
  const int np = pre.num_outputs();
  // under a selection vector the first stage evaluates the SELECTED rows only (it may raise only
  // where the caller's projector may) and its temporaries hold one row per slot
  const int64_t num_rows = sel != nullptr ? sel->num_slots : batch_rows;
  cols.assign(in, in + num_cols);
  auto alloc = [&](int64_t bytes, void** p) -> Status {
    bytes = std::max<int64_t>(bytes, 8);
    if (mem == MemKind::kHost) {
      host.emplace_back(new std::vector<uint8_t>(static_cast<size_t>(bytes)));
      *p = host.back()->data();
      return Status::OK();
    }
    dev.emplace_back(new DeviceBuffer());
    GDV_RETURN_NOT_OK(dev.back()->Allocate(static_cast<size_t>(bytes)));
    *p = dev.back()->get();
    return Status::OK();
  };
  // first guess for the byte buffers: as many bytes as the var-len inputs hold plus 32 per row
  // (device memory; the host path starts from nothing, it sizes its buffers by a length pass
  // anyway).  The evaluation reports what it needs, so a short buffer costs one retry.
  int64_t guess = 0;
  if (mem == MemKind::kDevice) {
    guess = 32 * num_rows;
    for (int k = 0; k < num_cols; k++)
      if (in[k].offsets != nullptr) guess += in[k].data_size;
    guess = std::min<int64_t>(guess, (int64_t{1} << 31) - 64);
  }
  std::vector<OutputBuffers> po(np);
  std::vector<int64_t> cap(np, guess);
  // (round-2 advisor: the blanket guess grabbed gigabytes of HBM scratch per call however small the
  // temporaries were) — what the previous batch of this plan produced, per row, + 25 % is a far
  // better first guess; a short buffer still costs one retry
  if (hints != nullptr && mem == MemKind::kDevice)
    for (int e = 0; e < np && e < static_cast<int>(hints->size()); e++) {
      const int64_t per_row_x16 = (*hints)[e].load(std::memory_order_relaxed);
      if (per_row_x16 > 0)
        cap[e] = std::min<int64_t>(guess, (per_row_x16 * num_rows / 16) * 5 / 4 + 4096);
    }
  const int64_t vbytes = mem == MemKind::kHost ? BytesForBits(num_rows) : Projector::ValidityBytes(num_rows);
  for (int e = 0; e < np; e++) {
    if (!pre.output_type(e).is_varlen()) return Status::Invalid("two-stage plan: first stage must produce utf8 / binary");
    GDV_RETURN_NOT_OK(alloc(vbytes, &po[e].validity));
    po[e].validity_size = std::max<int64_t>(vbytes, 8);
    GDV_RETURN_NOT_OK(alloc((num_rows + 1) * 4, &po[e].offsets));
    po[e].offsets_size = (num_rows + 1) * 4;
    GDV_RETURN_NOT_OK(alloc(cap[e] + 16, &po[e].data));  // (+16: zeroed behind the bytes produced, below)
    po[e].data_size = cap[e];
  }
  Status s = pre.Evaluate(batch_rows, in, num_cols, sel, po.data(), np, mem, stream, 0);
  if (!s.ok()) {
    bool grew = false;
    for (int e = 0; e < np; e++) {
      if (po[e].data_size > cap[e]) {
        cap[e] = po[e].data_size;
        GDV_RETURN_NOT_OK(alloc(cap[e] + 16, &po[e].data));
        grew = true;
      }
      po[e].data_size = cap[e];
    }
    if (!grew) return s;
    GDV_RETURN_NOT_OK(pre.Evaluate(batch_rows, in, num_cols, sel, po.data(), np, mem, stream, 0));
  }
  if (hints != nullptr)
    for (int e = 0; e < np && e < static_cast<int>(hints->size()); e++)
      (*hints)[e].store(std::max<int64_t>(1, po[e].data_size * 16 / std::max<int64_t>(num_rows, 1) + 1),
                        std::memory_order_relaxed);
  for (int e = 0; e < np; e++) {
    ColumnBuffers c;
    c.validity = po[e].validity;
    c.validity_size = po[e].validity_size;
    c.offsets = po[e].offsets;
    c.offsets_size = po[e].offsets_size;
    c.data = po[e].data;
    // The second stage's byte sweep reads whole 16-byte pieces: the 16 bytes behind the text are zeroed and
    // readable, so that pool garbage is never taken for bytes >= 0x80 (round 4: every synchronous two-stage
    // batch went optimistic kernel -> NOTASCII -> exact variant because of it).
    if (mem == MemKind::kDevice) {
      GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(static_cast<char*>(po[e].data) + po[e].data_size, 0, 16, stream));
      c.data_size = po[e].data_size + 16;
    } else {
      c.data_size = po[e].data_size;
    }
    cols.push_back(c);
  }
  return Status::OK();
}

// ------------------------------------------------------------------ Projector

int64_t Projector::VarlenBytesHint(int i, int64_t rows) const {
  if (i < 0 || static_cast<size_t>(i) >= out_bytes_x16_.size() || rows <= 0) return 0;
  const int64_t x16 = out_bytes_x16_[i].load(std::memory_order_relaxed);
  if (x16 <= 0) return 0;
  // (an eighth of head room: batches of one column rarely differ by more)
  const __int128 bytes = static_cast<__int128>(x16) * rows / 16 * 9 / 8 + 256;
  return static_cast<int64_t>(std::min<__int128>(bytes, (int64_t{1} << 31) - 64));
}

Status Projector::Make(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                       SelectionMode mode, const Configuration& config,
                       std::shared_ptr<Projector>* out) {
  if (out == nullptr) return Status::Invalid("Projector::Make: null output pointer");
  if (exprs.empty()) return Status::Invalid("Expressions cannot be empty");
  CodegenOptions opts = CodegenOptions::FromEnv();
  std::string key = "P|" + SchemaKey(schema) + "|";
  for (auto& e : exprs) {
    if (!e) return Status::Invalid("Expression cannot be null");
    key += e->CacheKey() + ";";
  }
  key += "|m" + std::to_string(static_cast<int>(mode)) + "|" + opts.Key() +
         (config.optimize ? "|O" : "|o");
  if (auto hit = ProjectorCache().Get(key)) {
    *out = hit;
    return Status::OK();
  }
  auto p = std::make_shared<Projector>();
  p->schema_ = schema;
  p->plan_schema_ = schema;
  const std::vector<ExpressionPtr>* planned = &exprs;
  StagedExpressions staged;
  // (round 3: in every selection mode — the first stage is built in the SAME mode, so it evaluates,
  // and can raise, only on the selected rows, and writes one temporary row per slot)
  StageMaterialisedValues(schema, exprs, &staged);
  if (!staged.pre.empty()) {
    for (auto& e : exprs) GDV_RETURN_NOT_OK(ValidateExpression(schema, *e));  // errors name the caller's trees
    GDV_RETURN_NOT_OK(Projector::Make(schema, staged.pre, mode, config, &p->pre_));
    p->plan_schema_ = staged.schema;
    planned = &staged.main;
    p->stage_hints_ = std::vector<std::atomic<int64_t>>(staged.pre.size());
  }
  if (p->pre_) opts.rows_word = true;  // (second stage: GDV_ROWS reads the gate's word of an asynchronous evaluation)
  GDV_RETURN_NOT_OK(PlanProjector(p->plan_schema_, *planned, mode, opts, &p->plan_,
                                  mode == SelectionMode::kNone ? 0x7fffffff : static_cast<int>(schema.size())));
  p->out_bytes_x16_ = std::vector<std::atomic<int64_t>>(exprs.size());
  const PlanDeviceState* st = nullptr;
  // Tier 0 (round 6): a plan the ahead-of-time interpreter takes, whose specialised kernel is not at hand yet, does not
  // wait for hipRTC (0.25-0.9 s): the compilation is queued, Make returns, and Evaluate runs the plan's post-fix program
  // until the code object is there.  GDV_NO_TIER0=1: as before.  GDV_FORCE_TIER0=1 (tests): tier 0 always.
  GDV_RETURN_NOT_OK(Runtime::Get().EnsureDevice());
  if (!EngineKnobs::Get().no_tier0 && p->pre_ == nullptr && mode == SelectionMode::kNone) {
    std::unique_ptr<tier0::Args> prog(new tier0::Args);
    if (BuildTier0Program(schema, exprs, /*filter=*/false, p->plan_, prog.get(), nullptr)) {
      const int state = EngineKnobs::Get().force_tier0 ? 0 : Runtime::Get().CodeObjectState(p->plan_.kernel_name);
      if (state == 0 || EngineKnobs::Get().force_tier0) {
        if (EngineKnobs::Get().force_tier0 || Runtime::Get().CompileInBackground(p->plan_.source, p->plan_.kernel_name)) {
          p->tier0_ = std::move(prog);
          p->tier0_pending_.store(true);
        }  // (else: the background compiler was shut down — Make waits for the compilation below, as before round 6)
      }
    }
  }
  if (!p->tier0_) GDV_RETURN_NOT_OK(p->states_.Get(p->plan_, &st));  // compiles + loads on the calling thread's device
  ProjectorCache().Put(key, p);
  *out = p;
  return Status::OK();
}

// Tier 0 is on while the plan has a program and its specialised code object has not arrived (or always, under
// GDV_FORCE_TIER0).  A background compilation that failed turns it off: the blocking path then reports the error.
static bool Tier0Active(const tier0::Args* prog, std::atomic<bool>* pending, const std::string& kernel_name) {
  if (prog == nullptr) return false;
  if (EngineKnobs::Get().force_tier0) return true;
  if (!pending->load(std::memory_order_relaxed)) return false;
  if (Runtime::Get().CodeObjectState(kernel_name, /*memory_only=*/true) != 0) {
    pending->store(false, std::memory_order_relaxed);
    return false;
  }
  return true;
}
bool Projector::UseTier0() const { return Tier0Active(tier0_.get(), &tier0_pending_, plan_.kernel_name); }
bool Filter::UseTier0() const { return Tier0Active(tier0_.get(), &tier0_pending_, plan_.kernel_name); }

Status Projector::Evaluate(int64_t num_rows, const ColumnBuffers* cols, int num_cols,
                           const SelectionView* sel, OutputBuffers* outs, int num_outs,
                           MemKind mem, hipStream_t stream, uint32_t flags, const void* rows_word, void* err_word) const {
  if (num_rows <= 0) return Status::Invalid("RecordBatch must be non-empty.");
  const bool two_stage = pre_ != nullptr && !(flags & kEvalStaged);  // (kEvalStaged: the caller ran the first stage)
  if (outs == nullptr) return Status::Invalid("Output array vector cannot be null");
  if (num_outs != num_outputs())
    return Status::Invalid("number of output buffers (" + std::to_string(num_outs) +
                           ") does not match the number of expressions (" +
                           std::to_string(num_outputs()) + ")");
  const bool has_sel = sel != nullptr && sel->mode != SelectionMode::kNone;
  if (has_sel != (plan_.mode != SelectionMode::kNone) || (has_sel && sel->mode != plan_.mode))
    return Status::Invalid("selection vector type does not match the mode the projector was built for");
  const int64_t out_rows = has_sel ? sel->num_slots : num_rows;
  if (has_sel) {
    // what the selection vector's index type can address bounds its slot count
    const int64_t cap = sel->mode == SelectionMode::kUInt16 ? 65536
                        : sel->mode == SelectionMode::kUInt32 ? (int64_t{1} << 32)
                                                              : INT64_MAX;
    if (sel->num_slots < 0 || sel->num_slots > cap)
      return Status::Invalid("selection vector: invalid slot count " + std::to_string(sel->num_slots));
  }
  if (has_sel && sel->num_slots_device != nullptr &&
      (mem != MemKind::kDevice || plan_.num_varlen_outputs > 0 || two_stage))
    return Status::Invalid("a device-resident slot count needs device buffers and fixed-width outputs "
                           "(read the count back and pass it as num_slots instead)");
  Runtime& rt = Runtime::Get();
  GDV_RETURN_NOT_OK(rt.EnsureDevice());
  const PlanDeviceState* dev = nullptr;
  // tier 0: while the specialised kernel is still compiling this evaluation interprets the plan's program instead
  const bool tier0 = UseTier0() && !has_sel && rows_word == nullptr && err_word == nullptr;
  GDV_RETURN_NOT_OK(states_.Get(plan_, &dev, /*need_kernel=*/!tier0));

  ArgBlock args(plan_.layout);
  Staging st;
  DeviceBuffer err;
  // var-len outputs: grand totals / per-tile granules of the in-kernel offsets scan
  DeviceBuffer tile_counts, tile_starts, wave_head, wave_counts, wave_bases, wave_chunks;
  StageColumns stage;  // two-stage plans: the first stage's temporary columns (outlive the drain below)
  // declared last: drains first (the byte pass of a var-len plan reads pooled scratch)
  StreamDrain drain{stream, mem == MemKind::kHost || plan_.has_varlen_output || two_stage};
  if (two_stage) {
    if (num_cols != static_cast<int>(schema_.size()))
      return Status::Invalid("number of columns in batch (" + std::to_string(num_cols) +
                             ") does not match the schema (" + std::to_string(schema_.size()) + ")");
    GDV_RETURN_NOT_OK(stage.Run(*pre_, num_rows, cols, num_cols, mem, stream, has_sel ? sel : nullptr, &stage_hints_));
    cols = stage.cols.data();
    num_cols = static_cast<int>(stage.cols.size());
  }
  if (mem == MemKind::kHost && num_rows <= Staging::kPackRows) GDV_RETURN_NOT_OK(st.EnablePacked());
  GDV_RETURN_NOT_OK(BindInputs(plan_, plan_schema_, cols, num_cols, num_rows, mem, stream, &args, &st,
                               has_sel ? out_rows : -1));
  BindLiterals(plan_, dev->consts, &args);
  // pooled staging blocks (e.g. the zero-padded copy of a tiny var-len buffer) go back to the
  // pool when this call returns: an asynchronous evaluation must not outlive them
  drain.armed = drain.armed || !st.buffers.empty();
  args.Set64(ArgLayout::kOffN, static_cast<uint64_t>(out_rows));

  if (has_sel) {
    const int w = plan_.mode == SelectionMode::kUInt16 ? 2 : plan_.mode == SelectionMode::kUInt32 ? 4 : 8;
    if (out_rows > 0 && sel->indices == nullptr) return Status::Invalid("selection vector has no buffer");
    if (mem == MemKind::kHost) {
      void* d = nullptr;
      GDV_RETURN_NOT_OK(st.In(sel->indices, out_rows * w, std::max<int64_t>(out_rows, 1) * w, stream, &d));
      args.SetPtr(ArgLayout::kOffSel, d);
    } else {
      args.SetPtr(ArgLayout::kOffSel, sel->indices);
      args.SetPtr(ArgLayout::kOffAux2, sel->num_slots_device);  // null: the count is kOffN
    }
  }
  if (rows_word != nullptr) args.SetPtr(ArgLayout::kOffAux2, rows_word);

  // outputs
  std::vector<void*> dev_data(num_outs, nullptr), dev_valid(num_outs), dev_offs(num_outs, nullptr);
  for (int e = 0; e < num_outs; e++) {
    const DataType& t = plan_.output_types[e];
    const int64_t need_valid_dev = ValidityBytes(out_rows);
    const int64_t need_data_dev = t.is_varlen() ? 0 : DataBytes(t, out_rows);
    const int64_t need_offs = t.is_varlen() ? (out_rows + 1) * 4 : 0;
    if (t.is_varlen() && (outs[e].offsets == nullptr || outs[e].offsets_size < need_offs))
      return Status::Invalid("output buffer " + std::to_string(e) + ": offsets buffer too small (" +
                             std::to_string(need_offs) + " bytes needed)");
    if (mem == MemKind::kHost) {
      const int64_t need_data_host = t.id == kBool ? BytesForBits(out_rows) : need_data_dev;
      if (outs[e].validity_size < BytesForBits(out_rows) || outs[e].data_size < need_data_host ||
          (out_rows > 0 && (outs[e].validity == nullptr || (outs[e].data == nullptr && !t.is_varlen()))))
        return Status::Invalid("output buffer " + std::to_string(e) + " too small");
      const int64_t vbytes = out_rows > 0 ? BytesForBits(out_rows) : 0;
      // Buffers inside a registered host range (gdv_host_register / gdv_host_alloc) that hold whole
      // 8-byte words are written in place by the kernel; the others come back through the staging block.
      const bool fixed = !t.is_varlen();
      dev_valid[e] = fixed && outs[e].validity_size >= need_valid_dev && (reinterpret_cast<uintptr_t>(outs[e].validity) & 7) == 0
                         ? HostRegistry::Get().View(outs[e].validity, need_valid_dev) : nullptr;
      if (dev_valid[e] == nullptr)
        GDV_RETURN_NOT_OK(st.Out(std::max<int64_t>(need_valid_dev, 8), vbytes, outs[e].validity, &dev_valid[e]));
      if (t.is_varlen()) {
        GDV_RETURN_NOT_OK(st.Out(need_offs, out_rows > 0 ? need_offs : 0, outs[e].offsets, &dev_offs[e]));
      } else {
        const int64_t dbytes = out_rows == 0 ? 0 : (t.id == kBool ? vbytes : need_data_dev);
        dev_data[e] = fixed && outs[e].data_size >= need_data_dev && (reinterpret_cast<uintptr_t>(outs[e].data) & 15) == 0
                          ? HostRegistry::Get().View(outs[e].data, need_data_dev) : nullptr;
        if (dev_data[e] == nullptr)
          GDV_RETURN_NOT_OK(st.Out(std::max<int64_t>(need_data_dev, 8), dbytes, outs[e].data, &dev_data[e]));
      }
    } else {
      if (outs[e].validity_size < need_valid_dev || outs[e].data_size < need_data_dev)
        return Status::Invalid("output buffer " + std::to_string(e) +
                               " too small (device buffers need 8-byte word granularity: " +
                               std::to_string(need_valid_dev) + " validity bytes, " +
                               std::to_string(need_data_dev) + " data bytes)");
      dev_valid[e] = outs[e].validity;
      dev_data[e] = outs[e].data;
      dev_offs[e] = outs[e].offsets;
    }
    args.SetOutData(e, dev_data[e]);
    args.SetOutValid(e, dev_valid[e]);
    args.SetOutOffsets(e, dev_offs[e]);
    // offsets[0] = 0 is written by the byte pass with every other offset; an empty selection
    // launches nothing (the closing offset comes from the scan launcher)
    if (t.is_varlen() && out_rows == 0) GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(dev_offs[e], 0, 4, stream));
  }

  const int nv = plan_.num_varlen_outputs;
  const bool has_err = plan_.can_raise && nv == 0;  // var-len plans keep the error word in their scan-state block
  const bool own_err = has_err && err_word == nullptr;
  if (own_err) {
    GDV_RETURN_NOT_OK(err.Allocate(8));
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(err.get(), 0, 8, stream));
    args.SetPtr(ArgLayout::kOffErr, err.get());
  } else if (has_err) {
    args.SetPtr(ArgLayout::kOffErr, err_word);  // the caller's word: raised into, never read here
  }

  GDV_RETURN_NOT_OK(st.FlushIn(stream));
  EvalTrace trace(tier0 ? "project (tier 0: interpreted)" : "project", plan_.kernel_name, out_rows, stream);
  std::vector<uint64_t> totals(num_outs, 0);
  uint32_t err_bits = 0;
  if (nv == 0) {
    if (out_rows > 0 && tier0) {
      tier0::Args t0 = *tier0_;
      std::memcpy(t0.block, args.data(), args.size());
      GDV_HIP_RETURN_NOT_OK(LaunchTier0(t0, out_rows, rt.num_cus(), stream));
      CountTier0Launch();
    } else if (out_rows > 0) {
      GDV_RETURN_NOT_OK(rt.Launch(*dev->kernel.load(), GridFor(plan_, out_rows), plan_.opts.waves * 64, args.data(),
                                  args.size(), stream));
    }
  } else if (out_rows > 0) {
    // Scanner shape — single launch: workgroup 0 scans the tile totals (granules: tile_starts;
    // grand totals: tile_counts), workers post one granule and poll one.  Wave shape (plans whose
    // output lengths follow from the offsets, gdv_planner.cc): pre-pass -> offsets scan -> main
    // kernel of independent wave tiles; a batch that breaks its ASCII / flat assumption is re-run
    // on the scanner-shaped general kernel.  Device buffers: the caller's capacities are honoured
    // inside the kernels (tiles that do not fit skip their bytes) and the totals say what was
    // needed.  Host buffers: a first launch with capacity 0 sizes the device byte buffers, a
    // second one fills them (the path is PCIe-bound anyway).
    const int ng = (nv + 1) / 2;
    // (tile of the scanner-shaped kernel: a wave plan's fallback has its own)
    const int sc_u = plan_.general_subtiles > 0 ? plan_.general_subtiles : plan_.opts.subtiles;
    const int sc_w = plan_.general_waves > 0 ? plan_.general_waves : plan_.opts.waves;
    const int64_t rows_wg = 64 * static_cast<int64_t>(sc_u) * sc_w;
    const int64_t ntiles = (out_rows + rows_wg - 1) / rows_wg;
    // head of the state block, one memset and one read-back per launch:
    // [error word | grand totals (2 * ng) | wave shape: totals of the scanned segments]
    int nseg = 0;
    for (int sgm : plan_.wave_segments) nseg = std::max(nseg, sgm + 1);
    const size_t totals_bytes = static_cast<size_t>(2 * ng) * 8;
    const size_t head_bytes = 8 + totals_bytes + static_cast<size_t>(nseg) * 8;
    std::vector<uint64_t> back(head_bytes / 8, 0);
    std::vector<int> vl;
    for (int e = 0; e < num_outs; e++)
      if (plan_.output_types[e].is_varlen()) vl.push_back(e);
    std::vector<uint64_t> seg(2 * ng, 0);
    const CompiledKernel* active = dev->kernel.load();
    char* state = nullptr;
    size_t state_bytes = 0;
    auto run = [&](int64_t grid) -> Status {  // scanner shape
      if (state == nullptr) {
        state_bytes = 8 + totals_bytes + static_cast<size_t>(2 * ng * ntiles) * 8;
        GDV_RETURN_NOT_OK(tile_starts.Allocate(state_bytes));
        state = tile_starts.as<char>();
      }
      args.SetPtr(ArgLayout::kOffErr, state);
      args.SetPtr(ArgLayout::kOffCounts, state + 8);
      args.SetPtr(ArgLayout::kOffMask, state + 8 + totals_bytes);
      GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(state, 0, state_bytes, stream));
      GDV_RETURN_NOT_OK(rt.Launch(*active, grid, sc_w * 64, args.data(), args.size(), stream));
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(back.data(), state, 8 + totals_bytes, hipMemcpyDeviceToHost, stream));
      GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
      err_bits = static_cast<uint32_t>(back[0]);
      for (int i = 0; i < 2 * ng; i++) seg[i] = back[1 + i];
      return Status::OK();
    };
    // wave shape
    const int64_t rows_wt = 64 * static_cast<int64_t>(plan_.opts.subtiles);
    const int64_t nwt = (out_rows + rows_wt - 1) / rows_wt;
    const int64_t seg_stride = (nwt + 3) & ~int64_t{3};  // the scan kernels read the totals 16 bytes at a time
    std::unique_ptr<ArgBlock> pargs;
    const bool has_exact = plan_.wave_tiles && plan_.exact != nullptr;
    auto exact_kernels = [&]() -> Status {  // compiled the first time a batch needs them
      PlanDeviceState* d = const_cast<PlanDeviceState*>(dev);
      if (d->kernel_exact.load() == nullptr) {
        const CompiledKernel* k = nullptr;
        GDV_RETURN_NOT_OK(rt.GetKernel(plan_.exact->source, plan_.exact->kernel_name, &k));
        if (plan_.exact->prepass) {
          const CompiledKernel* kp = nullptr;
          GDV_RETURN_NOT_OK(rt.GetKernel(plan_.exact->prepass->source, plan_.exact->prepass->kernel_name, &kp));
          d->kernel_pre_exact.store(kp);
        }
        d->kernel_exact.store(k);
      }
      return Status::OK();
    };
    auto run_wave = [&](bool exact) -> Status {
      const CompiledKernel* k_main = dev->kernel.load();
      const CompiledKernel* k_pre = dev->kernel_pre;
      if (exact) {
        GDV_RETURN_NOT_OK(exact_kernels());
        k_main = dev->kernel_exact.load();
        k_pre = dev->kernel_pre_exact.load();
      }
      if (wave_head.get() == nullptr) {
        GDV_RETURN_NOT_OK(wave_head.Allocate(head_bytes));
        if (nseg > 0) {
          GDV_RETURN_NOT_OK(wave_counts.Allocate(static_cast<size_t>(nseg * seg_stride) * 4 + 64));
          GDV_RETURN_NOT_OK(wave_bases.Allocate(static_cast<size_t>(nseg * seg_stride) * 8));
          GDV_RETURN_NOT_OK(wave_chunks.Allocate(static_cast<size_t>(nseg * ScanChunks(nwt)) * 8));
          // the pre-pass reads columns the main kernel has bound (and, on the host path, staged) already
          const KernelPlan& pp = *plan_.prepass;
          pargs.reset(new ArgBlock(pp.layout));
          for (size_t kp = 0; kp < pp.input_fields.size(); kp++) {
            int k = -1;
            for (size_t j = 0; j < plan_.input_fields.size(); j++)
              if (plan_.input_fields[j] == pp.input_fields[kp]) k = static_cast<int>(j);
            if (k < 0 || (pp.input_needs_values[kp] && !plan_.input_needs_values[k]) ||
                (pp.input_needs_validity[kp] && !plan_.input_needs_validity[k]))
              return Status::ExecutionError("internal: pre-pass input not bound by the main kernel");
            pargs->CopyInSlot(static_cast<int>(kp), args, k);
          }
          BindLiterals(pp, dev->consts_pre, pargs.get());
          pargs->Set64(ArgLayout::kOffN, static_cast<uint64_t>(out_rows));
          pargs->SetPtr(ArgLayout::kOffErr, wave_head.get());
          pargs->SetPtr(ArgLayout::kOffCounts, wave_counts.get());
          pargs->Set64(ArgLayout::kOffAux1, static_cast<uint64_t>(seg_stride));
          // selection mode (round 5): the pre-pass walks the same slots — the (staged) selection vector, the rows word
          pargs->Set64(ArgLayout::kOffSel, args.Get64(ArgLayout::kOffSel));
          pargs->Set64(ArgLayout::kOffAux2, args.Get64(ArgLayout::kOffAux2));
        }
      }
      char* const head = wave_head.as<char>();
      args.SetPtr(ArgLayout::kOffErr, head);
      args.SetPtr(ArgLayout::kOffCounts, head + 8);
      args.SetPtr(ArgLayout::kOffMask, wave_bases.get());
      args.Set64(ArgLayout::kOffAux1, static_cast<uint64_t>(seg_stride));
      GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(head, 0, head_bytes, stream));
      const int64_t grid = GridFor(plan_, out_rows);
      if (nseg > 0) {
        GDV_RETURN_NOT_OK(rt.Launch(*k_pre, std::min<int64_t>(grid, static_cast<int64_t>(rt.num_cus()) * 16),
                                    plan_.opts.waves * 64, pargs->data(), pargs->size(), stream));
        int32_t* closing[kMaxScanSegments] = {};
        for (int v = 0; v < nv; v++)
          if (plan_.wave_segments[v] >= 0)
            closing[plan_.wave_segments[v]] = static_cast<int32_t*>(dev_offs[vl[v]]) + out_rows;
        GDV_HIP_RETURN_NOT_OK(LaunchSegmentedOffsetsScan(wave_counts.as<uint32_t>(), nwt, seg_stride, nseg,
                                                         wave_chunks.as<uint64_t>(), wave_bases.as<uint64_t>(),
                                                         reinterpret_cast<uint64_t*>(head + 8 + totals_bytes), closing, stream));
      }
      GDV_RETURN_NOT_OK(rt.Launch(*k_main, grid, plan_.opts.waves * 64, args.data(), args.size(), stream));
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(back.data(), head, head_bytes, hipMemcpyDeviceToHost, stream));
      GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
      err_bits = static_cast<uint32_t>(back[0]);
      for (int v = 0; v < nv; v++)
        seg[v] = plan_.wave_segments[v] >= 0 ? back[1 + 2 * ng + plan_.wave_segments[v]] : back[1 + v];
      return Status::OK();
    };
    // Outputs that are an input column's (mapped) bytes — a column passed through, upper(col),
    // lower(col) — are first evaluated OPTIMISTICALLY: bytes copied by the byte sweep as they
    // are read, offsets = input offsets rebased, no scan.  That holds unless a NULL row carries
    // bytes (Arrow allows it, producers rarely do it); the kernel then raises NOTFLAT and the
    // batch is re-run with those outputs on the general path.  The wave shape adds the ASCII
    // assumption of its pre-pass (NOTASCII).
    // Round 4 — which kernels a batch runs on is decided PER BATCH (it used to be sticky for good:
    // one byte >= 0x80 sent every later batch of the Projector to the scanner kernel, 0.27 of the
    // roofline):  NOTASCII -> the wave shape's EXACT variant (flags from the sweep: same structure,
    // the pre-pass reads the bytes once more), which also tells whether the batch really held such
    // bytes — if not, the next batch starts on the optimistic kernels again;  NOTFLAT -> the
    // scanner-shaped general kernel, and the optimistic kernels get another try every 16th batch.
    const bool has_optimistic = plan_.wave_tiles || plan_.has_flat_output;
    constexpr uint32_t kNotFlat = 16u, kNotAscii = 32u, kSawUtf8 = 64u;
    auto general_kernel = [&]() -> Status {
      if (dev->kernel_general.load() == nullptr) {
        const CompiledKernel* k = nullptr;
        GDV_RETURN_NOT_OK(rt.GetKernel(plan_.source_general, plan_.kernel_name_general, &k));
        const_cast<PlanDeviceState*>(dev)->kernel_general.store(k);
      }
      return Status::OK();
    };
    const int64_t scanner_grid = std::max<int64_t>(1, ntiles) + 1;  // one workgroup per tile + the scanner
    auto launch = [&]() -> Status {
      if (!has_optimistic) {
        active = dev->kernel.load();
        GDV_RETURN_NOT_OK(run(scanner_grid));
      } else {
        int path = EngineKnobs::Get().no_optflat ? 2 : path_hint_.load(std::memory_order_relaxed);
        if (path == 1 && !has_exact) path = 2;
        if (path == 2 && !EngineKnobs::Get().no_optflat &&
            (general_batches_.fetch_add(1, std::memory_order_relaxed) & 15u) == 15u)
          path = 0;
        if (path == 0) {
          if (plan_.wave_tiles) {
            GDV_RETURN_NOT_OK(run_wave(false));
          } else {
            active = dev->kernel.load();
            GDV_RETURN_NOT_OK(run(scanner_grid));
          }
          if ((err_bits & kNotAscii) && !(err_bits & kNotFlat) && has_exact) path = 1;
          else if (err_bits & (kNotAscii | kNotFlat)) path = 2;
          else path_hint_.store(0, std::memory_order_relaxed);
        }
        if (EngineKnobs::Get().trace)
          fprintf(stderr, "[gdv] var-len path after the optimistic attempt: %d (error bits 0x%x)\n", path, err_bits);
        if (path == 1) {
          GDV_RETURN_NOT_OK(run_wave(true));
          if (EngineKnobs::Get().trace) fprintf(stderr, "[gdv] exact wave variant ran (error bits 0x%x)\n", err_bits);
          if (err_bits & kNotFlat) path = 2;
          else path_hint_.store((err_bits & kSawUtf8) ? 1 : 0, std::memory_order_relaxed);
        }
        if (path == 2) {
          GDV_RETURN_NOT_OK(general_kernel());
          active = dev->kernel_general.load();
          GDV_RETURN_NOT_OK(run(scanner_grid));
          path_hint_.store(2, std::memory_order_relaxed);
        }
      }
      err_bits &= ~(kNotFlat | kNotAscii | kSawUtf8);
      if (err_bits & 8u) {
        // The scan made no progress for a very long time: some workgroup of the grid was not
        // scheduled while later ones waited for it.  Never observed (workgroups start in index
        // order); the serial-safe configuration — scanner + ONE worker workgroup walking all
        // tiles in order — cannot wait on anything unscheduled.
        GDV_RETURN_NOT_OK(run(2));
        if (err_bits & 8u) return Status::ExecutionError("var-len projection: device scan stalled");
      }
      return Status::OK();
    };
    for (int v = 0; v < nv; v++) args.SetOutCap(vl[v], mem == MemKind::kHost ? 0 : outs[vl[v]].data_size);
    GDV_RETURN_NOT_OK(launch());
    Status capacity = Status::OK();
    for (int v = 0; v < nv; v++) {
      const int e = vl[v];
      totals[e] = seg[v];
      // (totals saturate at 2^31 - 1, so a total of exactly that many bytes cannot be told from an
      // overflow: rejected too — one byte short of what int32 offsets could address)
      if (totals[e] >= 0x7fffffffull)
        return Status::Invalid("var-len output " + std::to_string(e) + " exceeds 2 GiB");
      const int64_t have = outs[e].data_size;
      outs[e].data_size = static_cast<int64_t>(totals[e]);  // bytes needed / produced
      if (static_cast<size_t>(e) < out_bytes_x16_.size() && out_rows > 0) {
        const int64_t seen = static_cast<int64_t>(totals[e]) * 16 / out_rows + 1;
        // a DECAYING maximum: a batch that produces more raises the hint at once, one that produces
        // less lets it sink by an eighth towards what it produced — one outlier batch no longer makes
        // every later call allocate for its ratio for good (round-3 advisor)
        int64_t cur = out_bytes_x16_[e].load(std::memory_order_relaxed);
        for (;;) {
          const int64_t next = seen >= cur ? seen : std::max(seen, cur - (cur >> 3) - 1);
          if (next == cur || out_bytes_x16_[e].compare_exchange_weak(cur, next, std::memory_order_relaxed)) break;
        }
      }
      if (have < static_cast<int64_t>(totals[e]) || (totals[e] > 0 && outs[e].data == nullptr))
        capacity = Status::Invalid("output buffer " + std::to_string(e) + ": data capacity " +
                                   std::to_string(have) + " < " + std::to_string(totals[e]) +
                                   " bytes needed (data_size updated; retry with a larger buffer)");
    }
    if (err_bits != 0) return Status::ExecutionError(ErrorMessage(err_bits));
    GDV_RETURN_NOT_OK(capacity);
    if (mem == MemKind::kHost) {
      bool any = false;
      for (int v = 0; v < nv; v++) {
        const int e = vl[v];
        DeviceBuffer& dd = st.Add();
        GDV_RETURN_NOT_OK(dd.Allocate(std::max<uint64_t>(totals[e], 8)));
        dev_data[e] = dd.get();
        args.SetOutData(e, dev_data[e]);
        args.SetOutCap(e, static_cast<int64_t>(totals[e]));
        any |= totals[e] > 0;
      }
      if (any) GDV_RETURN_NOT_OK(launch());
    }
  } else {
    for (int e = 0; e < num_outs; e++)
      if (plan_.output_types[e].is_varlen()) outs[e].data_size = 0;
  }

  if (own_err)
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&err_bits, err.get(), 4, hipMemcpyDeviceToHost, stream));
  if (mem == MemKind::kHost) {
    GDV_RETURN_NOT_OK(st.FetchOut(stream));  // validity, fixed-width values, offsets
    for (int e = 0; e < num_outs; e++)       // var-len bytes: sized after the length pass
      if (plan_.output_types[e].is_varlen() && out_rows > 0 && totals[e] > 0)
        GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(outs[e].data, dev_data[e], totals[e],
                                             hipMemcpyDeviceToHost, stream));
  }
  const bool must_sync = mem == MemKind::kHost || (plan_.can_raise && err_word == nullptr) || !(flags & kEvalAsync) || two_stage;
  if (must_sync) GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
  if (err_bits != 0) return Status::ExecutionError(ErrorMessage(err_bits));
  if (mem == MemKind::kHost) st.Deliver();
  return Status::OK();
}

Status Projector::EvaluateMany(const BatchView* batches, int nb, hipStream_t stream, uint32_t flags) const {
  if (nb <= 0) return Status::OK();
  if (batches == nullptr) return Status::Invalid("null batch list");
  Runtime& rt = Runtime::Get();
  GDV_RETURN_NOT_OK(rt.EnsureDevice());
  const PlanDeviceState* dev = nullptr;
  GDV_RETURN_NOT_OK(states_.Get(plan_, &dev));
  const size_t stride = static_cast<size_t>(plan_.layout.total());
  const bool one_launch = plan_.has_many_entry && dev->kernel.load()->function_many != nullptr && pre_ == nullptr &&
                          plan_.num_varlen_outputs == 0 && !plan_.string_skeleton &&
                          stride * static_cast<size_t>(nb) <= Runtime::kPinnedBlock && nb <= 65535 &&
                          !EngineKnobs::Get().no_evaluate_many;
  if (!one_launch) {
    // batch by batch, all enqueued on `stream`; one wait at the end unless the caller asked for none
    for (int b = 0; b < nb; b++)
      GDV_RETURN_NOT_OK(Evaluate(batches[b].num_rows, batches[b].cols, batches[b].num_cols, nullptr, batches[b].outs,
                                 batches[b].num_outs, MemKind::kDevice, stream, kEvalAsync));
    if (!(flags & kEvalAsync)) GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
    return Status::OK();
  }
  const bool small_pin = stride * static_cast<size_t>(nb) <= Runtime::kPinnedSmall;
  char* pin = nullptr;
  GDV_RETURN_NOT_OK(small_pin ? rt.AcquirePinnedSmall(&pin) : rt.AcquirePinned(&pin));
  struct PinGuard {
    Runtime& rt; char*& pin; bool small;
    ~PinGuard() { if (pin != nullptr) { if (small) rt.ReleasePinnedSmall(pin); else rt.ReleasePinned(pin); } }
  } pin_guard{rt, pin, small_pin};
  DeviceBuffer table, err;
  StreamDrain drain{stream, false};  // declared after the pooled blocks: an error return waits for what was enqueued
  GDV_RETURN_NOT_OK(table.Allocate(stride * nb));
  if (plan_.can_raise) GDV_RETURN_NOT_OK(err.Allocate(8));
  Staging st;  // (device buffers bind in place: nothing is staged)
  int64_t grid = 1;
  // every batch is validated and its argument block written (host memory only) BEFORE anything is
  // enqueued: an Invalid return frees `table` / `err` with nothing pending on them
  for (int b = 0; b < nb; b++) {
    const BatchView& v = batches[b];
    if (v.num_rows <= 0) return Status::Invalid("RecordBatch must be non-empty.");
    if (v.outs == nullptr || v.num_outs != num_outputs())
      return Status::Invalid("batch " + std::to_string(b) + ": number of output buffers does not match the number of expressions");
    ArgBlock args(plan_.layout);
    GDV_RETURN_NOT_OK(BindInputs(plan_, plan_schema_, v.cols, v.num_cols, v.num_rows, MemKind::kDevice, stream, &args, &st));
    if (!st.buffers.empty()) return Status::Invalid("internal: staged input in a multi-batch evaluation");
    BindLiterals(plan_, dev->consts, &args);
    args.Set64(ArgLayout::kOffN, static_cast<uint64_t>(v.num_rows));
    if (plan_.can_raise) args.SetPtr(ArgLayout::kOffErr, err.get());
    for (int e = 0; e < v.num_outs; e++) {
      const DataType& t = plan_.output_types[e];
      if (v.outs[e].validity == nullptr || v.outs[e].data == nullptr || v.outs[e].validity_size < ValidityBytes(v.num_rows) ||
          v.outs[e].data_size < DataBytes(t, v.num_rows))
        return Status::Invalid("batch " + std::to_string(b) + ", output buffer " + std::to_string(e) + " too small");
      args.SetOutData(e, v.outs[e].data);
      args.SetOutValid(e, v.outs[e].validity);
    }
    std::memcpy(pin + stride * b, args.data(), stride);
    grid = std::max(grid, GridFor(plan_, v.num_rows));
  }
  EvalTrace trace("project-many", plan_.kernel_name, nb, stream);
  drain.armed = true;
  if (plan_.can_raise) GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(err.get(), 0, 8, stream));
  GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(table.get(), pin, stride * nb, hipMemcpyHostToDevice, stream));
  GDV_RETURN_NOT_OK(rt.LaunchMany(*dev->kernel.load(), grid, nb, plan_.opts.waves * 64, table.get(), stream));
  uint32_t err_bits = 0;
  if (plan_.can_raise)
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&err_bits, err.get(), 4, hipMemcpyDeviceToHost, stream));
  if ((flags & kEvalAsync) && !plan_.can_raise) {
    // the table and the pinned block go back once the stream has passed this point
    table.release_after(stream);
    char* p = pin;
    pin = nullptr;
    Runtime* owner = &rt;
    const bool small = small_pin;
    rt.Defer(stream, [owner, p, small] { if (small) owner->ReleasePinnedSmall(p); else owner->ReleasePinned(p); });
    drain.armed = false;
    return Status::OK();
  }
  GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
  drain.armed = false;
  if (err_bits != 0) return Status::ExecutionError(ErrorMessage(err_bits));
  return Status::OK();
}

// ------------------------------------------------------------------ var-len plans, asynchronously

Status Projector::EvaluateAsync(int64_t num_rows, const ColumnBuffers* cols, int num_cols, const SelectionView* sel,
                                OutputBuffers* outs, int num_outs, hipStream_t stream, void* result) const {
  if (pre_ != nullptr) return EvaluateAsyncTwoStage(num_rows, cols, num_cols, sel, outs, num_outs, stream, result);
  return EvaluateAsyncStage(num_rows, cols, num_cols, sel, outs, num_outs, stream, result, nullptr);
}

// The stages of a staged plan on the stream, a gate kernel between each two (gdv_kernels.h: StageGate).
Status Projector::EvaluateAsyncTwoStage(int64_t num_rows, const ColumnBuffers* cols, int num_cols, const SelectionView* sel,
                                        OutputBuffers* outs, int num_outs, hipStream_t stream, void* result) const {
  if (num_rows <= 0) return Status::Invalid("RecordBatch must be non-empty.");
  if (outs == nullptr || result == nullptr) return Status::Invalid("Output array vector and result block cannot be null");
  if (num_outs != num_outputs()) return Status::Invalid("number of output buffers does not match the number of expressions");
  if (num_cols != static_cast<int>(schema_.size()))
    return Status::Invalid("number of columns in batch (" + std::to_string(num_cols) +
                           ") does not match the schema (" + std::to_string(schema_.size()) + ")");
  const int np = pre_->num_outputs();
  if (np > kMaxStageOutputs) return Status::Invalid("too many temporaries for an asynchronous two-stage evaluation");
  const bool has_sel = sel != nullptr && sel->mode != SelectionMode::kNone;
  const int64_t stage_rows = has_sel ? sel->num_slots : num_rows;   // (with a device-resident count: the capacity)
  if (stage_rows <= 0) return EvaluateAsyncStage(num_rows, cols, num_cols, sel, outs, num_outs, stream, result, nullptr);
  Runtime& rt = Runtime::Get();
  GDV_RETURN_NOT_OK(rt.EnsureDevice());
  // temporaries: validity | offsets | bytes per first-stage output, sized like StageColumns::Run sizes them
  int64_t guess = 32 * stage_rows;
  for (int k = 0; k < num_cols; k++)
    if (cols[k].offsets != nullptr) guess += cols[k].data_size;
  guess = std::min<int64_t>(guess, (int64_t{1} << 31) - 64);
  StageCaps caps{};
  std::vector<DeviceBuffer> blocks(3 * static_cast<size_t>(np) + 1);
  std::vector<OutputBuffers> po(np);
  std::vector<ColumnBuffers> all(cols, cols + num_cols);
  StreamDrain drain{stream, false};  // declared after the blocks: an error return after the first enqueue waits
  const int64_t vbytes = ValidityBytes(stage_rows);
  for (int e = 0; e < np; e++) {
    if (!pre_->output_type(e).is_varlen()) return Status::Invalid("two-stage plan: first stage must produce utf8 / binary");
    int64_t cap = guess;
    const int64_t per_row_x16 = e < static_cast<int>(stage_hints_.size()) ? stage_hints_[e].load(std::memory_order_relaxed) : 0;
    if (per_row_x16 > 0) cap = std::min<int64_t>(guess, (per_row_x16 * stage_rows / 16) * 5 / 4 + 4096);
    cap = std::max<int64_t>(cap, 8);
    GDV_RETURN_NOT_OK(blocks[3 * e].Allocate(static_cast<size_t>(std::max<int64_t>(vbytes, 8))));
    GDV_RETURN_NOT_OK(blocks[3 * e + 1].Allocate(static_cast<size_t>((stage_rows + 1) * 4)));
    GDV_RETURN_NOT_OK(blocks[3 * e + 2].Allocate(static_cast<size_t>(cap) + 16));  // (+16: zeroed by the gate)
    po[e].validity = blocks[3 * e].get();
    po[e].validity_size = std::max<int64_t>(vbytes, 8);
    po[e].offsets = blocks[3 * e + 1].get();
    po[e].offsets_size = (stage_rows + 1) * 4;
    po[e].data = blocks[3 * e + 2].get();
    po[e].data_size = cap;
    caps.cap[e] = cap;
    caps.data[e] = po[e].data;
    ColumnBuffers c;
    c.validity = po[e].validity;
    c.validity_size = po[e].validity_size;
    c.offsets = po[e].offsets;
    c.offsets_size = po[e].offsets_size;
    c.data = po[e].data;
    c.data_size = cap + 16;
    all.push_back(c);
  }
  // gate block: first stage's result (1 + np words) | rows word | status word
  DeviceBuffer& gate = blocks[3 * static_cast<size_t>(np)];
  GDV_RETURN_NOT_OK(gate.Allocate(256));
  uint64_t* const stage_result = gate.as<uint64_t>();
  int64_t* const rows_word = reinterpret_cast<int64_t*>(gate.as<char>() + 128);
  uint64_t* const status_word = reinterpret_cast<uint64_t*>(gate.as<char>() + 136);
  drain.armed = true;
  // (a first stage that is itself staged — upper(reverse(replace(..))) — goes through this function again: its result block,
  // status and byte totals, is what the gate reads either way; round 5: three and more stages were synchronous only)
  if (pre_->pre_ != nullptr)
    GDV_RETURN_NOT_OK(pre_->EvaluateAsyncTwoStage(num_rows, cols, num_cols, sel, po.data(), np, stream, stage_result));
  else
    GDV_RETURN_NOT_OK(pre_->EvaluateAsyncStage(num_rows, cols, num_cols, sel, po.data(), np, stream, stage_result, nullptr));
  GDV_HIP_RETURN_NOT_OK(LaunchStageGate(stage_result, np, caps, has_sel ? static_cast<const int64_t*>(sel->num_slots_device) : nullptr,
                                        stage_rows, rows_word, status_word, stream));
  if (plan_.num_varlen_outputs > 0) {
    GDV_RETURN_NOT_OK(EvaluateAsyncStage(num_rows, all.data(), static_cast<int>(all.size()), sel, outs, num_outs, stream, result,
                                         rows_word));
  } else {  // fixed-width outputs only: the ordinary asynchronous launch over the staged columns, rows from the gate
    // (a second stage that can raise — divide, castINT of the staged text ... — raises into result[0] itself: round 4
    // sent such plans to the synchronous call)
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(result, 0, 8 * (1 + static_cast<size_t>(num_outs)), stream));
    GDV_RETURN_NOT_OK(Evaluate(num_rows, all.data(), static_cast<int>(all.size()), sel, outs, num_outs, MemKind::kDevice, stream,
                               kEvalAsync | kEvalStaged, rows_word, plan_.can_raise ? result : nullptr));
  }
  GDV_HIP_RETURN_NOT_OK(LaunchOrStatus(static_cast<uint64_t*>(result), status_word, stream));
  for (auto& b : blocks) b.release_after(stream);
  drain.armed = false;
  return Status::OK();
}

Status Projector::EvaluateAsyncStage(int64_t num_rows, const ColumnBuffers* cols, int num_cols, const SelectionView* sel,
                                     OutputBuffers* outs, int num_outs, hipStream_t stream, void* result,
                                     const void* rows_word) const {
  if (num_rows <= 0) return Status::Invalid("RecordBatch must be non-empty.");
  if (outs == nullptr || result == nullptr) return Status::Invalid("Output array vector and result block cannot be null");
  if (num_outs != num_outputs()) return Status::Invalid("number of output buffers does not match the number of expressions");
  const bool has_sel = sel != nullptr && sel->mode != SelectionMode::kNone;
  if (has_sel != (plan_.mode != SelectionMode::kNone) || (has_sel && sel->mode != plan_.mode))
    return Status::Invalid("selection vector type does not match the mode the projector was built for");
  const int64_t out_rows = has_sel ? sel->num_slots : num_rows;   // (with a device-resident count: the capacity)
  if (has_sel && (sel->num_slots < 0 || (out_rows > 0 && sel->indices == nullptr)))
    return Status::Invalid("selection vector: invalid slot count or no buffer");
  const int nv = plan_.num_varlen_outputs;
  uint64_t* const res = static_cast<uint64_t*>(result);
  if (nv == 0) {  // fixed-width plans: the ordinary asynchronous launch; no byte totals; a plan that can raise raises into result[0]
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(result, 0, 8 * (1 + static_cast<size_t>(num_outs)), stream));
    return Evaluate(num_rows, cols, num_cols, sel, outs, num_outs, MemKind::kDevice, stream, kEvalAsync, nullptr,
                    plan_.can_raise ? result : nullptr);
  }
  if (out_rows == 0) {
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(result, 0, 8 * (1 + static_cast<size_t>(num_outs)), stream));
    for (int e = 0; e < num_outs; e++)
      if (plan_.output_types[e].is_varlen() && outs[e].offsets != nullptr)
        GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(outs[e].offsets, 0, 4, stream));
    return Status::OK();
  }
  Runtime& rt = Runtime::Get();
  GDV_RETURN_NOT_OK(rt.EnsureDevice());
  const PlanDeviceState* dev = nullptr;
  GDV_RETURN_NOT_OK(states_.Get(plan_, &dev));

  ArgBlock args(plan_.layout);
  Staging st;
  DeviceBuffer state, wave_head, wave_counts, wave_bases, wave_chunks;
  StreamDrain drain{stream, false};  // declared last: an error return after the first enqueue waits before the blocks go back
  GDV_RETURN_NOT_OK(BindInputs(plan_, plan_schema_, cols, num_cols, num_rows, MemKind::kDevice, stream, &args, &st,
                               has_sel ? out_rows : -1));
  BindLiterals(plan_, dev->consts, &args);
  drain.armed = !st.buffers.empty();   // (a tiny var-len buffer was copied into a padded pool block)
  args.Set64(ArgLayout::kOffN, static_cast<uint64_t>(out_rows));
  if (has_sel) {
    args.SetPtr(ArgLayout::kOffSel, sel->indices);
    args.SetPtr(ArgLayout::kOffAux2, sel->num_slots_device);  // null: the count is kOffN
  }
  if (rows_word != nullptr) args.SetPtr(ArgLayout::kOffAux2, rows_word);  // second stage: the gate's word (it folds the slot count in)
  std::vector<int> vl;
  for (int e = 0; e < num_outs; e++) {
    const DataType& t = plan_.output_types[e];
    const int64_t need_valid = ValidityBytes(out_rows), need_data = t.is_varlen() ? 0 : DataBytes(t, out_rows);
    if (outs[e].validity == nullptr || outs[e].validity_size < need_valid || outs[e].data_size < need_data ||
        (outs[e].data == nullptr && (need_data > 0 || outs[e].data_size > 0)))
      return Status::Invalid("output buffer " + std::to_string(e) + " too small");
    if (t.is_varlen()) {
      if (outs[e].offsets == nullptr || outs[e].offsets_size < (out_rows + 1) * 4)
        return Status::Invalid("output buffer " + std::to_string(e) + ": offsets buffer too small");
      vl.push_back(e);
    }
    args.SetOutData(e, outs[e].data);
    args.SetOutValid(e, outs[e].validity);
    args.SetOutOffsets(e, outs[e].offsets);
    if (t.is_varlen()) args.SetOutCap(e, outs[e].data_size);
  }
  GDV_RETURN_NOT_OK(st.FlushIn(stream));
  GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(result, 0, 8 * (1 + static_cast<size_t>(num_outs)), stream));
  drain.armed = true;

  const int ng = (nv + 1) / 2;
  const size_t totals_bytes = static_cast<size_t>(2 * ng) * 8;
  const bool has_optimistic = plan_.wave_tiles || plan_.has_flat_output;
  int path = !has_optimistic ? 2 : (EngineKnobs::Get().no_optflat ? 2 : path_hint_.load(std::memory_order_relaxed));
  if (path == 1 && !(plan_.wave_tiles && plan_.exact != nullptr)) path = 2;
  EvalTrace trace("project-async", plan_.kernel_name, out_rows, stream);
  if (path != 2 && plan_.wave_tiles) {
    // ---- wave shape: pre-pass -> offsets scan -> main kernel (the optimistic pair or its exact variant)
    const CompiledKernel* k_main = dev->kernel.load();
    const CompiledKernel* k_pre = dev->kernel_pre;
    const KernelPlan* pp = plan_.prepass.get();
    if (path == 1) {
      PlanDeviceState* d = const_cast<PlanDeviceState*>(dev);
      if (d->kernel_exact.load() == nullptr) {
        const CompiledKernel* k = nullptr;
        GDV_RETURN_NOT_OK(rt.GetKernel(plan_.exact->source, plan_.exact->kernel_name, &k));
        if (plan_.exact->prepass) {
          const CompiledKernel* kp = nullptr;
          GDV_RETURN_NOT_OK(rt.GetKernel(plan_.exact->prepass->source, plan_.exact->prepass->kernel_name, &kp));
          d->kernel_pre_exact.store(kp);
        }
        d->kernel_exact.store(k);
      }
      k_main = dev->kernel_exact.load();
      k_pre = dev->kernel_pre_exact.load();
    }
    int nseg = 0;
    for (int sgm : plan_.wave_segments) nseg = std::max(nseg, sgm + 1);
    const size_t head_bytes = 8 + totals_bytes + static_cast<size_t>(nseg) * 8;
    const int64_t rows_wt = 64 * static_cast<int64_t>(plan_.opts.subtiles);
    const int64_t nwt = (out_rows + rows_wt - 1) / rows_wt;
    const int64_t seg_stride = (nwt + 3) & ~int64_t{3};
    GDV_RETURN_NOT_OK(wave_head.Allocate(head_bytes));
    char* const head = wave_head.as<char>();
    args.SetPtr(ArgLayout::kOffErr, head);
    args.SetPtr(ArgLayout::kOffCounts, head + 8);
    args.Set64(ArgLayout::kOffAux1, static_cast<uint64_t>(seg_stride));
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(head, 0, head_bytes, stream));
    const int64_t grid = GridFor(plan_, out_rows);
    if (nseg > 0) {
      if (pp == nullptr || k_pre == nullptr) return Status::ExecutionError("internal: wave plan without a pre-pass");
      GDV_RETURN_NOT_OK(wave_counts.Allocate(static_cast<size_t>(nseg * seg_stride) * 4 + 64));
      GDV_RETURN_NOT_OK(wave_bases.Allocate(static_cast<size_t>(nseg * seg_stride) * 8));
      GDV_RETURN_NOT_OK(wave_chunks.Allocate(static_cast<size_t>(nseg * ScanChunks(nwt)) * 8));
      // a second stage whose gate is closed walks 0 rows: its pre-pass writes no count, the scan must still see zeros
      // (the same for a selection whose slot count sits in device memory: wave tiles past it write no count)
      if (rows_word != nullptr || (has_sel && sel->num_slots_device != nullptr))
        GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(wave_counts.get(), 0, static_cast<size_t>(nseg * seg_stride) * 4 + 64, stream));
      args.SetPtr(ArgLayout::kOffMask, wave_bases.get());
      ArgBlock pargs(pp->layout);
      for (size_t kp = 0; kp < pp->input_fields.size(); kp++) {
        int k = -1;
        for (size_t j = 0; j < plan_.input_fields.size(); j++)
          if (plan_.input_fields[j] == pp->input_fields[kp]) k = static_cast<int>(j);
        if (k < 0) return Status::ExecutionError("internal: pre-pass input not bound by the main kernel");
        pargs.CopyInSlot(static_cast<int>(kp), args, k);
      }
      BindLiterals(*pp, dev->consts_pre, &pargs);
      pargs.Set64(ArgLayout::kOffN, static_cast<uint64_t>(out_rows));
      pargs.SetPtr(ArgLayout::kOffErr, head);
      pargs.SetPtr(ArgLayout::kOffCounts, wave_counts.get());
      pargs.Set64(ArgLayout::kOffAux1, static_cast<uint64_t>(seg_stride));
      pargs.Set64(ArgLayout::kOffSel, args.Get64(ArgLayout::kOffSel));    // (the pre-pass walks the same rows / slots as the main kernel)
      pargs.Set64(ArgLayout::kOffAux2, args.Get64(ArgLayout::kOffAux2));
      GDV_RETURN_NOT_OK(rt.Launch(*k_pre, std::min<int64_t>(grid, static_cast<int64_t>(rt.num_cus()) * 16),
                                  plan_.opts.waves * 64, pargs.data(), pargs.size(), stream));
      int32_t* closing[kMaxScanSegments] = {};
      for (int v = 0; v < nv; v++)
        if (plan_.wave_segments[v] >= 0) closing[plan_.wave_segments[v]] = static_cast<int32_t*>(outs[vl[v]].offsets) + out_rows;
      GDV_HIP_RETURN_NOT_OK(LaunchSegmentedOffsetsScan(wave_counts.as<uint32_t>(), nwt, seg_stride, nseg,
                                                       wave_chunks.as<uint64_t>(), wave_bases.as<uint64_t>(),
                                                       reinterpret_cast<uint64_t*>(head + 8 + totals_bytes), closing, stream));
    }
    GDV_RETURN_NOT_OK(rt.Launch(*k_main, grid, plan_.opts.waves * 64, args.data(), args.size(), stream));
    // the error word, minus the exact kernels' note that the batch did hold bytes >= 0x80 (64: not an error —
    // round 4 published it, and every asynchronous call on non-ASCII text looked failed to its caller)
    GDV_HIP_RETURN_NOT_OK(LaunchPublishStatus(res, reinterpret_cast<const uint32_t*>(head), 64u, stream));
    for (int v = 0; v < nv; v++) {
      const char* src = plan_.wave_segments[v] >= 0 ? head + 8 + totals_bytes + 8 * plan_.wave_segments[v] : head + 8 + 8 * v;
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(res + 1 + vl[v], src, 8, hipMemcpyDefault, stream));
    }
  } else {
    // ---- scanner shape: one launch (selection-mode plans; plans without a wave shape; path 2: the general kernel)
    const CompiledKernel* active = dev->kernel.load();
    if (has_optimistic && path == 2) {
      if (dev->kernel_general.load() == nullptr) {
        const CompiledKernel* k = nullptr;
        GDV_RETURN_NOT_OK(rt.GetKernel(plan_.source_general, plan_.kernel_name_general, &k));
        const_cast<PlanDeviceState*>(dev)->kernel_general.store(k);
      }
      active = dev->kernel_general.load();
    }
    const bool general = has_optimistic && path == 2 && plan_.wave_tiles;
    const int sc_u = general && plan_.general_subtiles > 0 ? plan_.general_subtiles : plan_.opts.subtiles;
    const int sc_w = general && plan_.general_waves > 0 ? plan_.general_waves : plan_.opts.waves;
    const int64_t rows_wg = 64 * static_cast<int64_t>(sc_u) * sc_w;
    const int64_t ntiles = (out_rows + rows_wg - 1) / rows_wg;
    const size_t state_bytes = 8 + totals_bytes + static_cast<size_t>(2 * ng * ntiles) * 8;
    GDV_RETURN_NOT_OK(state.Allocate(state_bytes));
    char* const sp = state.as<char>();
    args.SetPtr(ArgLayout::kOffErr, sp);
    args.SetPtr(ArgLayout::kOffCounts, sp + 8);
    args.SetPtr(ArgLayout::kOffMask, sp + 8 + totals_bytes);
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(sp, 0, state_bytes, stream));
    GDV_RETURN_NOT_OK(rt.Launch(*active, std::max<int64_t>(1, ntiles) + 1, sc_w * 64, args.data(), args.size(), stream));
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(res, sp, 4, hipMemcpyDefault, stream));
    for (int v = 0; v < nv; v++)
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(res + 1 + vl[v], sp + 8 + 8 * v, 8, hipMemcpyDefault, stream));
  }
  // scratch goes back to the pool when the stream has passed this point
  state.release_after(stream);
  wave_head.release_after(stream);
  wave_counts.release_after(stream);
  wave_bases.release_after(stream);
  wave_chunks.release_after(stream);
  for (auto& b : st.buffers) b.release_after(stream);
  drain.armed = false;
  return Status::OK();
}

// ------------------------------------------------------------------ Filter

Status Filter::Make(const Schema& schema, const ExpressionPtr& condition,
                    const Configuration& config, std::shared_ptr<Filter>* out) {
  if (out == nullptr) return Status::Invalid("Filter::Make: null output pointer");
  if (!condition) return Status::Invalid("Condition cannot be null");
  CodegenOptions opts = CodegenOptions::FromEnv();
  std::string key = "F|" + SchemaKey(schema) + "|" + condition->CacheKey() + "|" + opts.Key() +
                    (config.optimize ? "|O" : "|o");
  if (auto hit = FilterCache().Get(key)) {
    *out = hit;
    return Status::OK();
  }
  auto f = std::make_shared<Filter>();
  f->schema_ = schema;
  f->plan_schema_ = schema;
  f->chunks_.store(EngineKnobs::Get().filter_chunks);
  f->small_filter_.store(!EngineKnobs::Get().no_small_filter);
  ExpressionPtr planned = condition;
  StagedExpressions staged;
  StageMaterialisedValues(schema, {condition}, &staged);
  if (!staged.pre.empty()) {
    GDV_RETURN_NOT_OK(ValidateExpression(schema, *condition));
    GDV_RETURN_NOT_OK(Projector::Make(schema, staged.pre, SelectionMode::kNone, config, &f->pre_));
    f->plan_schema_ = staged.schema;
    planned = staged.main[0];
    f->stage_hints_ = std::vector<std::atomic<int64_t>>(staged.pre.size());
  }
  GDV_RETURN_NOT_OK(PlanFilter(f->plan_schema_, planned, opts, &f->plan_));
  const PlanDeviceState* st = nullptr;
  GDV_RETURN_NOT_OK(Runtime::Get().EnsureDevice());
  if (!EngineKnobs::Get().no_tier0 && f->pre_ == nullptr) {  // tier 0, as Projector::Make
    std::unique_ptr<tier0::Args> prog(new tier0::Args);
    if (BuildTier0Program(schema, {condition}, /*filter=*/true, f->plan_, prog.get(), nullptr)) {
      const int state = EngineKnobs::Get().force_tier0 ? 0 : Runtime::Get().CodeObjectState(f->plan_.kernel_name);
      if (state == 0 || EngineKnobs::Get().force_tier0) {
        if (EngineKnobs::Get().force_tier0 || Runtime::Get().CompileInBackground(f->plan_.source, f->plan_.kernel_name)) {
          f->tier0_ = std::move(prog);
          f->tier0_pending_.store(true);
        }
      }
    }
  }
  if (!f->tier0_) GDV_RETURN_NOT_OK(f->states_.Get(f->plan_, &st));  // compiles + loads on the calling thread's device
  FilterCache().Put(key, f);
  *out = f;
  return Status::OK();
}

namespace {
struct ScratchPart {  // a piece of a scratch block, spelled like a DeviceBuffer
  char* p;
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};
}  // namespace

Status Filter::SetTuning(const std::string& key, int64_t value) {
  if (key == "chunks") {
    if (value < 1 || value > 64) return Status::Invalid("filter tuning 'chunks': 1..64");
    chunks_.store(static_cast<int>(value));
  } else if (key == "small_filter") {
    small_filter_.store(value != 0);
  } else {
    return Status::Invalid("unknown filter tuning key '" + key + "'");
  }
  return Status::OK();
}

int64_t Filter::SmallBatchRows() const {
  if (!plan_.has_small_entry || plan_.string_skeleton || pre_ != nullptr) return 0;
  // one workgroup: at most 1024 wave tiles (LDS offsets), and no more rows than a workgroup gets
  // through in about the time the three-launch pipeline needs to start (~20 us)
  return std::min<int64_t>(64 * static_cast<int64_t>(plan_.opts.subtiles) * 1024, int64_t{1} << 17);
}

Status Filter::EvaluateMany(const BatchView* batches, int nb, SelectionMode mode, int64_t* counts_host,
                            void* counts_device, hipStream_t stream, uint32_t flags) const {
  if (nb <= 0) return Status::OK();
  if (batches == nullptr) return Status::Invalid("null batch list");
  if (mode == SelectionMode::kNone) return Status::Invalid("Selection vector type cannot be NONE");
  if (counts_host == nullptr && counts_device == nullptr) return Status::Invalid("Selection vector cannot be null");
  const int w = mode == SelectionMode::kUInt16 ? 2 : mode == SelectionMode::kUInt32 ? 4 : 8;
  Runtime& rt = Runtime::Get();
  GDV_RETURN_NOT_OK(rt.EnsureDevice());
  const PlanDeviceState* dev = nullptr;
  GDV_RETURN_NOT_OK(states_.Get(plan_, &dev));
  const int64_t cap_rows = SmallBatchRows();
  const size_t stride = static_cast<size_t>(plan_.layout.total());
  bool fused = cap_rows > 0 && dev->kernel.load()->function_small != nullptr && nb <= 65535 &&
               stride * static_cast<size_t>(nb) <= Runtime::kPinnedBlock / 2 &&
               small_filter_.load(std::memory_order_relaxed);
  for (int b = 0; fused && b < nb; b++) fused = batches[b].num_rows <= cap_rows;
  if (!fused) {
    for (int b = 0; b < nb; b++) {
      int64_t count = 0;
      GDV_RETURN_NOT_OK(Evaluate(batches[b].num_rows, batches[b].cols, batches[b].num_cols, mode, batches[b].out_indices,
                                 batches[b].max_slots, &count, MemKind::kDevice, stream, flags | kEvalNoSmall,
                                 counts_device != nullptr ? static_cast<char*>(counts_device) + 8 * b : nullptr));
      if (counts_host != nullptr) counts_host[b] = count;
    }
    return Status::OK();
  }
  const int64_t tile_rows = 64 * static_cast<int64_t>(plan_.opts.subtiles);
  // scratch: [argument table | error word | counts (int64 per batch) | per batch: match words, wave-tile counts]
  size_t scratch = stride * nb;
  scratch = (scratch + 255) & ~size_t{255};
  const size_t err_off = scratch;
  scratch += 256;
  const size_t cnt_off = scratch;
  scratch += (static_cast<size_t>(nb) * 8 + 255) & ~size_t{255};
  std::vector<size_t> mask_off(nb), tiles_off(nb);
  for (int b = 0; b < nb; b++) {
    const BatchView& v = batches[b];
    if (v.num_rows <= 0) return Status::Invalid("RecordBatch must be non-empty.");
    if (v.out_indices == nullptr) return Status::Invalid("Selection vector cannot be null");
    if (v.max_slots < v.num_rows)
      return Status::Invalid("Selection vector too small: max slots " + std::to_string(v.max_slots) + " < rows " +
                             std::to_string(v.num_rows));
    if (w == 2 && v.num_rows > 65536)
      return Status::Invalid("uint16 selection vector cannot address " + std::to_string(v.num_rows) + " rows");
    const int64_t nwords = (v.num_rows + 63) / 64, m = (v.num_rows + tile_rows - 1) / tile_rows;
    mask_off[b] = scratch;
    scratch += (static_cast<size_t>(nwords) * 8 + 255) & ~size_t{255};
    tiles_off[b] = scratch;
    scratch += (static_cast<size_t>(m) * 4 + 64 + 255) & ~size_t{255};
  }
  DeviceBuffer block;
  GDV_RETURN_NOT_OK(block.Allocate(scratch));
  char* const base = block.as<char>();
  // one batch: its argument block goes by value; several: a table, staged through a pinned block
  const bool by_value = nb == 1 && dev->kernel.load()->function_small1 != nullptr;
  const bool small_pin = stride * static_cast<size_t>(nb) <= Runtime::kPinnedSmall;
  char* pin = nullptr;
  std::vector<char> one(by_value ? stride : 0);
  if (by_value) pin = one.data();
  else GDV_RETURN_NOT_OK(small_pin ? rt.AcquirePinnedSmall(&pin) : rt.AcquirePinned(&pin));
  struct PinGuard {
    Runtime& rt; char*& pin; bool small, owned;
    ~PinGuard() { if (pin != nullptr && owned) { if (small) rt.ReleasePinnedSmall(pin); else rt.ReleasePinned(pin); } }
  } pin_guard{rt, pin, small_pin, !by_value};
  StreamDrain drain{stream, false};  // armed once something is enqueued: error returns wait before the blocks go back
  Staging st;
  for (int b = 0; b < nb; b++) {
    const BatchView& v = batches[b];
    ArgBlock args(plan_.layout);
    GDV_RETURN_NOT_OK(BindInputs(plan_, plan_schema_, v.cols, v.num_cols, v.num_rows, MemKind::kDevice, stream, &args, &st));
    if (!st.buffers.empty()) return Status::Invalid("internal: staged input in a multi-batch evaluation");
    BindLiterals(plan_, dev->consts, &args);
    args.Set64(ArgLayout::kOffN, static_cast<uint64_t>(v.num_rows));
    args.SetPtr(ArgLayout::kOffErr, base + err_off);
    args.SetPtr(ArgLayout::kOffMask, base + mask_off[b]);
    args.SetPtr(ArgLayout::kOffCounts, base + tiles_off[b]);
    args.SetPtr(ArgLayout::kOffAux1, v.out_indices);
    args.Set64(ArgLayout::kOffSel, static_cast<uint64_t>(w));
    args.SetPtr(ArgLayout::kOffAux2, base + cnt_off + 8 * b);
    std::memcpy(pin + stride * b, args.data(), stride);
  }
  EvalTrace trace("filter-small", plan_.kernel_name, nb, stream);
  drain.armed = true;
  if (plan_.can_raise) GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(base + err_off, 0, 8, stream));
  if (by_value) {
    GDV_RETURN_NOT_OK(rt.Launch(*dev->kernel.load(), 1, plan_.opts.waves * 64, pin, stride, stream, dev->kernel.load()->function_small1));
  } else {
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(base, pin, stride * nb, hipMemcpyHostToDevice, stream));
    GDV_RETURN_NOT_OK(rt.LaunchMany(*dev->kernel.load(), 1, nb, plan_.opts.waves * 64, base, stream, /*small=*/true));
  }
  if (counts_device != nullptr)
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(counts_device, base + cnt_off, 8 * static_cast<size_t>(nb), hipMemcpyDefault, stream));
  const bool async = (flags & kEvalAsync) != 0 && !plan_.can_raise && counts_device != nullptr;
  if (async) {
    if (counts_host != nullptr)
      for (int b = 0; b < nb; b++) counts_host[b] = -1;
    block.release_after(stream);
    if (!by_value) {
      char* p = pin;
      pin = nullptr;
      Runtime* owner = &rt;
      const bool small = small_pin;
      rt.Defer(stream, [owner, p, small] { if (small) owner->ReleasePinnedSmall(p); else owner->ReleasePinned(p); });
    }
    drain.armed = false;
    return Status::OK();
  }
  std::vector<int64_t> counts(nb, 0);
  uint32_t err_bits = 0;
  GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(counts.data(), base + cnt_off, 8 * static_cast<size_t>(nb), hipMemcpyDeviceToHost, stream));
  if (plan_.can_raise)
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&err_bits, base + err_off, 4, hipMemcpyDeviceToHost, stream));
  GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
  drain.armed = false;
  if (err_bits != 0) return Status::ExecutionError(ErrorMessage(err_bits));
  if (counts_host != nullptr)
    for (int b = 0; b < nb; b++) counts_host[b] = counts[b];
  return Status::OK();
}

// Input slots of an argument block advanced by `lo` rows (lo a multiple of 64): what a chunk of a
// pipelined filter binds.  Value pointers move by lo * width, bitmap word pointers by lo / 64 words
// (their bit shift is unchanged), var-len offsets by lo entries (the byte buffer stays whole).
static void AdvanceInputs(const KernelPlan& plan, const Schema& schema, const ArgBlock& base, int64_t lo,
                          ArgBlock* out) {
  *out = base;
  for (size_t k = 0; k < plan.input_fields.size(); k++) {
    const DataType& t = schema[plan.input_fields[k]].type;
    out->AdvanceInSlot(static_cast<int>(k), lo, t.is_varlen() ? -1 : (t.id == kBool ? 0 : t.byte_width()));
  }
}

Status Filter::Evaluate(int64_t num_rows, const ColumnBuffers* cols, int num_cols,
                        SelectionMode mode, void* out_indices, int64_t max_slots,
                        int64_t* num_selected, MemKind mem, hipStream_t stream, uint32_t flags,
                        void* count_out, int64_t row_base) const {
  if (num_rows <= 0) return Status::Invalid("RecordBatch must be non-empty.");
  if (row_base < 0) return Status::Invalid("negative row base");
  if (row_base != 0) flags |= kEvalNoSmall;  // (the one-workgroup kernel emits local positions)
  const bool tier0 = UseTier0();  // the predicate is interpreted; scan and index emission are the ahead-of-time kernels anyway
  if (tier0) flags |= kEvalNoSmall;
  if (out_indices == nullptr || (num_selected == nullptr && count_out == nullptr))
    return Status::Invalid("Selection vector cannot be null");
  if (mode == SelectionMode::kNone) return Status::Invalid("Selection vector type cannot be NONE");
  if (max_slots < num_rows)
    return Status::Invalid("Selection vector too small: max slots " + std::to_string(max_slots) +
                           " < rows " + std::to_string(num_rows));
  const int w = mode == SelectionMode::kUInt16 ? 2 : mode == SelectionMode::kUInt32 ? 4 : 8;
  if (w == 2 && row_base + num_rows > 65536)
    return Status::Invalid("uint16 selection vector cannot address " + std::to_string(row_base + num_rows) + " rows");
  if (w == 4 && row_base + num_rows > (int64_t(1) << 32))
    return Status::Invalid("uint32 selection vector cannot address " + std::to_string(row_base + num_rows) + " rows");
  // small HBM-resident batches: predicate + scan + emission by one workgroup in one launch
  // (one workgroup is the right tool up to a few thousand rows; beyond that the three-launch path,
  // which spreads the predicate over the chip, is faster for a single batch —
  // profiles/r03_small_batches.txt)
  if (mem == MemKind::kDevice && !(flags & kEvalNoSmall) && num_rows <= std::min<int64_t>(SmallBatchRows(), 8192) &&
      small_filter_.load(std::memory_order_relaxed)) {
    BatchView v;
    v.num_rows = num_rows; v.cols = cols; v.num_cols = num_cols; v.out_indices = out_indices; v.max_slots = max_slots;
    int64_t count = -1;
    GDV_RETURN_NOT_OK(EvaluateMany(&v, 1, mode, &count, count_out, stream, flags));
    if (num_selected != nullptr) *num_selected = count;
    return Status::OK();
  }
  Runtime& rt = Runtime::Get();
  GDV_RETURN_NOT_OK(rt.EnsureDevice());
  const PlanDeviceState* dev = nullptr;
  GDV_RETURN_NOT_OK(states_.Get(plan_, &dev, /*need_kernel=*/!tier0));
  // Asynchronous evaluation (device buffers; plans that cannot raise, no first stage): everything is
  // enqueued on `stream`, nothing waits, the selected-row count lands in *count_out (8 bytes of
  // device or pinned memory) in stream order — a selection-mode Projector can take it from there
  // (SelectionView::num_slots_device) without a host round trip.
  bool async = (flags & kEvalAsync) != 0 && mem == MemKind::kDevice && !plan_.can_raise && pre_ == nullptr &&
               count_out != nullptr;

  ArgBlock args(plan_.layout);
  Staging st;
  DeviceBuffer scratch, err, staged_out;
  StageColumns stage;  // two-stage plans: the first stage's temporary columns
  StreamDrain drain{stream, !async};  // declared last: drains before any pooled block is freed
  if (pre_) {
    if (num_cols != static_cast<int>(schema_.size()))
      return Status::Invalid("number of columns in batch (" + std::to_string(num_cols) +
                             ") does not match the schema (" + std::to_string(schema_.size()) + ")");
    GDV_RETURN_NOT_OK(stage.Run(*pre_, num_rows, cols, num_cols, mem, stream, nullptr, &stage_hints_));
    cols = stage.cols.data();
    num_cols = static_cast<int>(stage.cols.size());
  }
  if (mem == MemKind::kHost && num_rows <= Staging::kPackRows) GDV_RETURN_NOT_OK(st.EnablePacked());
  GDV_RETURN_NOT_OK(BindInputs(plan_, plan_schema_, cols, num_cols, num_rows, mem, stream, &args, &st));
  BindLiterals(plan_, dev->consts, &args);
  GDV_RETURN_NOT_OK(st.FlushIn(stream));
  if (async && !st.buffers.empty()) {  // a pooled staging block is in use (tiny var-len buffer): wait after all
    async = false;
    drain.armed = true;
  }

  // Chunked pipeline (GDV_FILTER_CHUNKS=n, fixed-width plans over HBM-resident batches; OFF by
  // default): the batch is cut into chunks; the predicate kernel of chunk k + 1 runs on `stream`
  // while the offsets scan and the index emission of chunk k run on a side stream.  The scan of
  // chunk k carries the running total of the chunks before it (device memory), so indices land at
  // their global places.  The round-2 verdict asked for it to hide the emission (0.23 ms at 10^9
  // rows) behind the predicate kernels; MEASURED (profiles/r03_c3_pipeline.txt, C3, one box): 1
  // chunk 2.855 ms, 4 chunks 2.925, 8 chunks 2.956, 16 chunks 2.988 — the emission competes with
  // the predicate kernel for the same HBM bandwidth and every extra launch adds a tail, so the
  // pipeline loses what the overlap wins.  Kept for re-measurement, and because the carried scan
  // is what lets the count stay on the device for the asynchronous API.
  const int64_t tile_rows = 64 * static_cast<int64_t>(plan_.opts.subtiles);   // one count per wave tile
  int chunks = chunks_.load(std::memory_order_relaxed);
  if (plan_.string_skeleton || mem != MemKind::kDevice) chunks = 1;
  // chunk boundaries: whole index-emission tiles (64 match words) and whole workgroup tiles
  const int64_t gran = 4096 * static_cast<int64_t>(std::max(1, plan_.opts.subtiles * plan_.opts.waves / 64 + 1));
  int64_t chunk_rows = (num_rows + chunks - 1) / chunks;
  chunk_rows = (chunk_rows + gran - 1) / gran * gran;
  chunks = static_cast<int>((num_rows + chunk_rows - 1) / chunk_rows);

  const int64_t nwords = (num_rows + 63) / 64;
  const int64_t m = (num_rows + tile_rows - 1) / tile_rows;  // wave tiles
  // one scratch block (one pool round trip, one deferred release): match words | wave-tile counts |
  // offsets | scan chunk sums | running totals
  auto up = [](size_t v) { return (v + 255) & ~size_t{255}; };
  const size_t mask_b = up(static_cast<size_t>(nwords) * 8), counts_b = up(static_cast<size_t>(m) * 4 + 64),
               offsets_b = up(static_cast<size_t>(m) * 8),
               sums_b = up(static_cast<size_t>(ScanChunks((chunk_rows + tile_rows - 1) / tile_rows) + 1) * 8 * chunks),
               totals_b = up(8 * static_cast<size_t>(chunks + 1));
  GDV_RETURN_NOT_OK(scratch.Allocate(mask_b + counts_b + offsets_b + sums_b + totals_b));
  const ScratchPart mask{scratch.as<char>()}, counts{mask.p + mask_b}, offsets{counts.p + counts_b},
      chunk_sums{offsets.p + offsets_b}, totals{chunk_sums.p + sums_b};
  if (plan_.can_raise) {
    GDV_RETURN_NOT_OK(err.Allocate(8));
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(err.get(), 0, 8, stream));
    args.SetPtr(ArgLayout::kOffErr, err.get());
  }
  void* dev_out = out_indices;
  if (mem == MemKind::kHost) {
    GDV_RETURN_NOT_OK(staged_out.Allocate(num_rows * w));
    dev_out = staged_out.get();
  }

  EvalTrace trace(tier0 ? "filter (tier 0: interpreted predicate)" : "filter", plan_.kernel_name, num_rows, stream);
  // From here on kernels that write `scratch` are in flight: an error return must not hand the block
  // back to the pool (another thread could be given it) before the streams have passed them.  The
  // drain is armed for every exit; the one successful asynchronous exit disarms it again and releases
  // the scratch behind an event instead.
  drain.armed = true;
  bool enqueued_all = false;
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> events;
  struct SideGuard {  // hands the side stream and the events back whatever path leaves the function
    Runtime& rt; hipStream_t& side; std::vector<hipEvent_t>& events; const bool& done;
    ~SideGuard() {
      if (!done && side != nullptr) (void)hipStreamSynchronize(side);  // error path: work of the side stream may still use the scratch
      for (auto e : events) rt.ReleaseEvent(e);
      rt.ReleaseStream(side);
    }
  } side_guard{rt, side, events, enqueued_all};
  if (chunks > 1) GDV_RETURN_NOT_OK(rt.AcquireStream(&side));
  const int64_t sums_per_chunk = ScanChunks((chunk_rows + tile_rows - 1) / tile_rows) + 1;
  for (int c = 0; c < chunks; c++) {
    const int64_t lo = c * chunk_rows, n = std::min(chunk_rows, num_rows - lo);
    const int64_t words = (n + 63) / 64, tiles = (n + tile_rows - 1) / tile_rows;
    ArgBlock cargs(plan_.layout);
    AdvanceInputs(plan_, plan_schema_, args, lo, &cargs);
    cargs.Set64(ArgLayout::kOffN, static_cast<uint64_t>(n));
    cargs.SetPtr(ArgLayout::kOffMask, mask.as<uint64_t>() + lo / 64);
    cargs.SetPtr(ArgLayout::kOffCounts, counts.as<uint32_t>() + lo / tile_rows);
    if (tier0) {
      tier0::Args t0 = *tier0_;
      std::memcpy(t0.block, cargs.data(), cargs.size());
      GDV_HIP_RETURN_NOT_OK(LaunchTier0(t0, n, rt.num_cus(), stream));
      CountTier0Launch();
    } else {
      GDV_RETURN_NOT_OK(rt.Launch(*dev->kernel.load(), GridFor(plan_, n), plan_.opts.waves * 64, cargs.data(), cargs.size(), stream));
    }
    hipStream_t s2 = stream;
    if (chunks > 1) {
      hipEvent_t e = nullptr;
      GDV_RETURN_NOT_OK(rt.AcquireEvent(&e));
      events.push_back(e);
      GDV_HIP_RETURN_NOT_OK(hipEventRecord(e, stream));
      GDV_HIP_RETURN_NOT_OK(hipStreamWaitEvent(side, e, 0));
      s2 = side;
    }
    GDV_HIP_RETURN_NOT_OK(LaunchOffsetsScan(counts.as<uint32_t>() + lo / tile_rows, tiles,
                                            chunk_sums.as<uint64_t>() + c * sums_per_chunk,
                                            offsets.as<uint64_t>() + lo / tile_rows, totals.as<uint64_t>() + c + 1, s2,
                                            c == 0 ? nullptr : totals.as<uint64_t>() + c));
    GDV_HIP_RETURN_NOT_OK(LaunchEmitIndices(mask.as<uint64_t>() + lo / 64, offsets.as<uint64_t>() + lo / tile_rows,
                                            words, plan_.opts.subtiles, row_base + lo, w, dev_out, rt.num_cus(), s2));
  }
  if (chunks > 1) {  // `stream` continues only after the side stream's last emission
    hipEvent_t e = nullptr;
    GDV_RETURN_NOT_OK(rt.AcquireEvent(&e));
    events.push_back(e);
    GDV_HIP_RETURN_NOT_OK(hipEventRecord(e, side));
    GDV_HIP_RETURN_NOT_OK(hipStreamWaitEvent(stream, e, 0));
  }
  const uint64_t* total_dev = totals.as<uint64_t>() + chunks;
  enqueued_all = true;  // (`stream` now waits for the side stream's last kernel)
  if (async) {
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(count_out, total_dev, 8, hipMemcpyDefault, stream));
    if (num_selected != nullptr) *num_selected = -1;
    // scratch goes back to the pool when the stream has passed this point
    scratch.release_after(stream);
    drain.armed = false;
    return Status::OK();
  }
  uint64_t count = 0;
  uint32_t err_bits = 0;
  GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&count, total_dev, 8, hipMemcpyDeviceToHost, stream));
  if (count_out != nullptr) GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(count_out, total_dev, 8, hipMemcpyDefault, stream));
  if (plan_.can_raise)
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&err_bits, err.get(), 4, hipMemcpyDeviceToHost, stream));
  GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
  if (err_bits != 0) return Status::ExecutionError(ErrorMessage(err_bits));
  if (mem == MemKind::kHost && count > 0) {
    GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(out_indices, dev_out, count * w, hipMemcpyDeviceToHost, stream));
    GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
  }
  if (num_selected != nullptr) *num_selected = static_cast<int64_t>(count);
  return Status::OK();
}

// ------------------------------------------------------------------ fused filter -> project

Status FilterProject::Make(const Schema& schema, const ExpressionPtr& condition, const std::vector<ExpressionPtr>& exprs,
                           SelectionMode index_mode, const Configuration& config, std::shared_ptr<FilterProject>* out) {
  (void)config;
  if (out == nullptr) return Status::Invalid("FilterProject::Make: null output pointer");
  if (!condition) return Status::Invalid("Condition cannot be null");
  if (exprs.empty()) return Status::Invalid("Expressions cannot be empty");
  // materialised values (concat / castVARCHAR ...) need a first stage: the chain handles them
  StagedExpressions staged;
  std::vector<ExpressionPtr> all = exprs;
  all.push_back(condition);
  StageMaterialisedValues(schema, all, &staged);
  if (!staged.pre.empty()) return Status::CodeGenError("fused filter-project: two-stage plans take the filter + projector chain");
  auto fp = std::make_shared<FilterProject>();
  fp->schema_ = schema;
  fp->condition_ = condition;
  fp->exprs_ = exprs;
  GDV_RETURN_NOT_OK(PlanFilterProject(schema, condition, exprs, index_mode, CodegenOptions::FromEnv(), &fp->plan_));
  fp->raises_ = fp->plan_.exprs_raise;
  const PlanDeviceState* st = nullptr;
  GDV_RETURN_NOT_OK(fp->states_.Get(fp->plan_, &st));
  *out = fp;
  return Status::OK();
}

Status FilterProject::SetTuning(const std::string& key, int64_t value) {
  if (key == "kernel" && value >= -1 && value <= 1) pinned_kernel_.store(static_cast<int>(value));
  else return Status::Invalid("FilterProject tuning: unknown key or value out of range: " + key);
  return Status::OK();
}

FilterProject::~FilterProject() {
  if (int64_t* p = pinned_count_.load()) (void)hipHostFree(p);
}

int FilterProject::which_kernel() const {
  if (plan_.fp_window_rows <= 0 || plan_.exact == nullptr) return -1;
  if (pinned_kernel_.load(std::memory_order_relaxed) >= 0) return pinned_kernel_.load(std::memory_order_relaxed);
  // the window holds fp_window_rows of a wave tile's 64 x subtiles rows; beyond ~85 % of that on average, wave
  // tiles start to overflow into the re-read path and the direct kernel is the better one
  const int limit = plan_.fp_window_rows * 1024 / (64 * plan_.opts.subtiles * std::max(1, plan_.fp_rounds)) * 85 / 100;
  return selected_per_1024_.load(std::memory_order_relaxed) > limit ? 1 : 0;
}

Status FilterProject::Evaluate(int64_t num_rows, const ColumnBuffers* cols, int num_cols, OutputBuffers* outs,
                               int num_outs, void* out_indices, int64_t max_slots, int64_t* num_selected, MemKind mem,
                               hipStream_t stream, uint32_t flags, void* count_out) const {
  bool stalled = false;
  GDV_RETURN_NOT_OK(EvaluateFused(num_rows, cols, num_cols, outs, num_outs, out_indices, max_slots, num_selected, mem, stream,
                                  flags, count_out, &stalled));
  if (!stalled) return Status::OK();
  // The look-back waited 5 s for an earlier workgroup tile (a device time-sliced away, or workgroups not dispatched
  // in index order): the launch is over, its outputs are not complete.  Round 4 returned ExecutionError here; the
  // reference's own chain gives the same results without any cross-workgroup wait.
  return EvaluateChain(num_rows, cols, num_cols, outs, num_outs, out_indices, max_slots, num_selected, mem, stream, count_out);
}

Status FilterProject::EvaluateChain(int64_t num_rows, const ColumnBuffers* cols, int num_cols, OutputBuffers* outs, int num_outs,
                                    void* out_indices, int64_t max_slots, int64_t* num_selected, MemKind mem,
                                    hipStream_t stream, void* count_out) const {
  // index width of the chain: the plan's own, or the narrowest that addresses the batch when it emits none
  const SelectionMode mode = plan_.mode != SelectionMode::kNone ? plan_.mode
                             : (num_rows <= (int64_t{1} << 32) ? SelectionMode::kUInt32 : SelectionMode::kUInt64);
  const int w = mode == SelectionMode::kUInt16 ? 2 : mode == SelectionMode::kUInt32 ? 4 : 8;
  // (the two operators are held through locals: a concurrent call that needs the other index width replaces
  // chain_projector_ under the lock, and must not free the one this call is still evaluating)
  std::shared_ptr<Filter> chain_filter;
  std::shared_ptr<Projector> chain_projector;
  {
    std::lock_guard<std::mutex> lock(chain_mu_);
    if (chain_filter_ == nullptr) GDV_RETURN_NOT_OK(Filter::Make(schema_, condition_, Configuration{}, &chain_filter_));
    if (chain_projector_ == nullptr || chain_projector_->plan().mode != mode)
      GDV_RETURN_NOT_OK(Projector::Make(schema_, exprs_, mode, Configuration{}, &chain_projector_));
    chain_filter = chain_filter_;
    chain_projector = chain_projector_;
  }
  std::vector<char> host_idx;
  DeviceBuffer dev_idx;
  void* idx = out_indices;
  if (plan_.mode == SelectionMode::kNone) {
    if (mem == MemKind::kHost) {
      host_idx.resize(static_cast<size_t>(num_rows) * w);
      idx = host_idx.data();
    } else {
      GDV_RETURN_NOT_OK(dev_idx.Allocate(static_cast<size_t>(num_rows) * w));
      idx = dev_idx.get();
    }
    max_slots = num_rows;
  }
  int64_t count = 0;
  GDV_RETURN_NOT_OK(chain_filter->Evaluate(num_rows, cols, num_cols, mode, idx, max_slots, &count, mem, stream, 0, count_out));
  if (count > 0) {
    SelectionView sel;
    sel.mode = mode;
    sel.indices = idx;
    sel.num_slots = count;
    // the projector sizes its checks for `count` rows; the caller's buffers hold num_rows
    GDV_RETURN_NOT_OK(chain_projector->Evaluate(num_rows, cols, num_cols, &sel, outs, num_outs, mem, stream, 0));
  }
  if (num_selected != nullptr) *num_selected = count;
  return Status::OK();
}

Status FilterProject::EvaluateFused(int64_t num_rows, const ColumnBuffers* cols, int num_cols, OutputBuffers* outs,
                                    int num_outs, void* out_indices, int64_t max_slots, int64_t* num_selected, MemKind mem,
                                    hipStream_t stream, uint32_t flags, void* count_out, bool* stalled) const {
  *stalled = false;
  if (num_rows < 0) return Status::Invalid("negative row count");
  if (num_outs != num_outputs() || (num_outs > 0 && outs == nullptr))
    return Status::Invalid("number of output buffers does not match the number of expressions");
  const SelectionMode mode = plan_.mode;
  const int w = mode == SelectionMode::kUInt16 ? 2 : mode == SelectionMode::kUInt32 ? 4 : mode == SelectionMode::kUInt64 ? 8 : 0;
  if (w != 0) {
    if (out_indices == nullptr && num_rows > 0) return Status::Invalid("Selection vector cannot be null");
    if (max_slots < num_rows)
      return Status::Invalid("Selection vector too small: max slots " + std::to_string(max_slots) + " < rows " +
                             std::to_string(num_rows));
    if (w == 2 && num_rows > 65536) return Status::Invalid("uint16 selection vector cannot address " + std::to_string(num_rows) + " rows");
    if (w == 4 && num_rows > (int64_t(1) << 32)) return Status::Invalid("uint32 selection vector cannot address " + std::to_string(num_rows) + " rows");
  }
  Runtime& rt = Runtime::Get();
  GDV_RETURN_NOT_OK(rt.EnsureDevice());
  const PlanDeviceState* dev = nullptr;
  GDV_RETURN_NOT_OK(states_.Get(plan_, &dev));
  bool async = (flags & kEvalAsync) != 0 && mem == MemKind::kDevice && !raises_ && count_out != nullptr;

  ArgBlock args(plan_.layout);
  Staging st;
  DeviceBuffer scratch;                      // look-back granules | count | error word | tile ticket
  std::vector<DeviceBuffer> staged(mem == MemKind::kHost ? 2 * num_outs + 1 : 0);  // host path: results are produced in HBM first
  StreamDrain drain{stream, !async};         // declared last: drains before any pooled block is freed
  if (num_rows == 0) {
    if (num_selected != nullptr) *num_selected = 0;
    if (count_out != nullptr) GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(count_out, 0, 8, stream));
    return Status::OK();
  }
  GDV_RETURN_NOT_OK(BindInputs(plan_, schema_, cols, num_cols, num_rows, mem, stream, &args, &st));
  BindLiterals(plan_, dev->consts, &args);
  if (!st.buffers.empty()) { async = false; drain.armed = true; }
  args.Set64(ArgLayout::kOffN, static_cast<uint64_t>(num_rows));

  if (int64_t* seen = pinned_count_.load(std::memory_order_relaxed)) {  // what an earlier asynchronous call selected
    const int64_t share = *reinterpret_cast<volatile int64_t*>(seen);  // rows selected per 1024, written by one launch
    if (share >= 0 && share <= 1024) selected_per_1024_.store(static_cast<int>(share), std::memory_order_relaxed);
  }
  // which shape: the windowed kernel unless recent batches selected more rows than its LDS window holds (the
  // direct kernel takes the same argument block: PlanFilterProject checks that its literals and constants are a
  // prefix of the windowed plan's)
  const CompiledKernel* kernel = dev->kernel.load();
  const KernelPlan* running = &plan_;
  if (which_kernel() == 1 && !EngineKnobs::Get().fp_window_only) {
    PlanDeviceState* d = const_cast<PlanDeviceState*>(dev);
    if (d->kernel_exact.load() == nullptr) {
      const CompiledKernel* k = nullptr;
      GDV_RETURN_NOT_OK(rt.GetKernel(plan_.exact->source, plan_.exact->kernel_name, &k));
      d->kernel_exact.store(k);
    }
    kernel = dev->kernel_exact.load();
    running = plan_.exact.get();
  }
  // one workgroup tile: waves x rounds x sub-tiles x 64 rows (the windowed kernel walks GDV_FP_K rounds per look-back)
  const int64_t rows_per_wg = 64 * static_cast<int64_t>(plan_.opts.subtiles) * plan_.opts.waves * std::max(1, running->fp_rounds);
  const int64_t grid = (num_rows + rows_per_wg - 1) / rows_per_wg;
  if (grid > 0x7fffffff) return Status::Invalid("batch too large for the fused filter-project launch");
  auto up = [](size_t v) { return (v + 255) & ~size_t{255}; };
  const size_t state_b = up(static_cast<size_t>(grid) * 8);
  GDV_RETURN_NOT_OK(scratch.Allocate(state_b + 256));
  char* const base = scratch.as<char>();
  // granules, count (+0), error word (+64) and the tile ticket (+128, round 6) start at zero (one memset)
  GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(base, 0, state_b + 256, stream));
  drain.armed = true;  // from here on an error return must wait for what was enqueued (re-disarmed on the async exit)
  args.SetPtr(ArgLayout::kOffMask, base);
  args.SetPtr(ArgLayout::kOffCounts, base + state_b);
  args.SetPtr(ArgLayout::kOffErr, base + state_b + 64);

  // outputs: the validity (and bool value) bitmaps are OR-ed into at tile boundaries -> pre-zeroed
  std::vector<void*> dev_data(num_outs), dev_valid(num_outs);
  for (int e = 0; e < num_outs; e++) {
    const DataType& t = plan_.output_types[e];
    const int64_t vbytes = Projector::ValidityBytes(num_rows), dbytes = Projector::DataBytes(t, num_rows);
    if (mem == MemKind::kHost) {
      const int64_t host_v = BytesForBits(num_rows), host_d = t.id == kBool ? BytesForBits(num_rows) : dbytes;
      if (outs[e].validity == nullptr || outs[e].data == nullptr || outs[e].validity_size < host_v || outs[e].data_size < host_d)
        return Status::Invalid("output buffer " + std::to_string(e) + " too small");
      GDV_RETURN_NOT_OK(staged[2 * e].Allocate(std::max<int64_t>(vbytes, 8)));
      GDV_RETURN_NOT_OK(staged[2 * e + 1].Allocate(std::max<int64_t>(dbytes, 8)));
      dev_valid[e] = staged[2 * e].get();
      dev_data[e] = staged[2 * e + 1].get();
    } else {
      if (outs[e].validity == nullptr || outs[e].data == nullptr || outs[e].validity_size < vbytes || outs[e].data_size < dbytes)
        return Status::Invalid("output buffer " + std::to_string(e) + " too small (device buffers need 8-byte word granularity: " +
                               std::to_string(vbytes) + " validity bytes, " + std::to_string(dbytes) + " data bytes)");
      dev_valid[e] = outs[e].validity;
      dev_data[e] = outs[e].data;
    }
    GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(dev_valid[e], 0, vbytes, stream));
    if (t.id == kBool) GDV_HIP_RETURN_NOT_OK(hipMemsetAsync(dev_data[e], 0, dbytes, stream));
    args.SetOutData(e, dev_data[e]);
    args.SetOutValid(e, dev_valid[e]);
  }
  void* dev_idx = out_indices;
  if (w != 0 && mem == MemKind::kHost) {
    GDV_RETURN_NOT_OK(staged[2 * num_outs].Allocate(num_rows * w));
    dev_idx = staged[2 * num_outs].get();
  }
  args.SetPtr(ArgLayout::kOffSel, dev_idx);
  GDV_RETURN_NOT_OK(st.FlushIn(stream));

  EvalTrace trace("filter-project", running->kernel_name, num_rows, stream);
  GDV_RETURN_NOT_OK(rt.Launch(*kernel, grid, plan_.opts.waves * 64, args.data(), args.size(), stream));
  const char* count_dev = base + state_b;
  // the count leaves through a one-thread kernel: -1 when the look-back gave up (GDV_ERR_STALL in the error word) —
  // round 4 copied the word as it was and an asynchronous caller never learnt that the outputs were not complete
  int64_t* telemetry = nullptr;
  if (async && plan_.exact != nullptr) {  // (two shapes to choose between: let the next call learn this one's count)
    telemetry = pinned_count_.load(std::memory_order_relaxed);
    if (telemetry == nullptr) {
      int64_t* fresh = nullptr;
      if (hipHostMalloc(reinterpret_cast<void**>(&fresh), 64, hipHostMallocDefault) == hipSuccess && fresh != nullptr) {
        fresh[0] = -1;
        int64_t* expected = nullptr;
        if (pinned_count_.compare_exchange_strong(expected, fresh)) telemetry = fresh;
        else { (void)hipHostFree(fresh); telemetry = expected; }
      } else {
        (void)hipGetLastError();
      }
    }
  }
  if (count_out != nullptr || telemetry != nullptr)
    GDV_HIP_RETURN_NOT_OK(LaunchPublishCount(static_cast<int64_t*>(count_out), reinterpret_cast<const int64_t*>(count_dev),
                                             reinterpret_cast<const uint32_t*>(base + state_b + 64), kErrStall, stream, telemetry, num_rows));
  if (async) {
    if (num_selected != nullptr) *num_selected = -1;
    scratch.release_after(stream);
    drain.armed = false;
    return Status::OK();
  }
  int64_t count = 0;
  uint32_t err_bits = 0;
  GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&count, count_dev, 8, hipMemcpyDeviceToHost, stream));
  GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&err_bits, base + state_b + 64, 4, hipMemcpyDeviceToHost, stream));
  GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
  if ((err_bits & kErrStall) != 0 || EngineKnobs::Get().fp_force_stall) {
    drain.armed = false;
    *stalled = true;
    return Status::OK();
  }
  if (err_bits != 0) return Status::ExecutionError(ErrorMessage(err_bits));
  selected_per_1024_.store(static_cast<int>(count * 1024 / num_rows), std::memory_order_relaxed);
  if (mem == MemKind::kHost && count > 0) {
    for (int e = 0; e < num_outs; e++) {
      const DataType& t = plan_.output_types[e];
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(outs[e].validity, dev_valid[e], BytesForBits(count), hipMemcpyDeviceToHost, stream));
      GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(outs[e].data, dev_data[e], t.id == kBool ? BytesForBits(count) : count * t.byte_width(),
                                           hipMemcpyDeviceToHost, stream));
    }
    if (w != 0) GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(out_indices, dev_idx, count * w, hipMemcpyDeviceToHost, stream));
    GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
  }
  drain.armed = false;
  if (num_selected != nullptr) *num_selected = count;
  return Status::OK();
}

// ------------------------------------------------------------------ tier 0: the program of a plan as text (no device)

Status Tier0Describe(const Schema& schema, const std::vector<ExpressionPtr>& exprs, bool is_condition, std::string* text) {
  KernelPlan plan;
  StagedExpressions staged;
  StageMaterialisedValues(schema, exprs, &staged);
  if (!staged.pre.empty()) return Status::NotImplemented("no tier 0: the plan materialises values in a first stage");
  if (is_condition) {
    if (exprs.size() != 1) return Status::Invalid("one condition expected");
    GDV_RETURN_NOT_OK(PlanFilter(schema, exprs[0], CodegenOptions::FromEnv(), &plan));
  } else {
    GDV_RETURN_NOT_OK(PlanProjector(schema, exprs, SelectionMode::kNone, CodegenOptions::FromEnv(), &plan));
  }
  std::unique_ptr<tier0::Args> prog(new tier0::Args);
  std::string why;
  if (!BuildTier0Program(schema, exprs, is_condition, plan, prog.get(), &why)) return Status::NotImplemented("no tier 0: " + why);
  *text = DescribeTier0Program(*prog);
  return Status::OK();
}

// ------------------------------------------------------------------ precompile (no device)

Status PrecompileProjector(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                           SelectionMode mode) {
  KernelPlan plan;
  StagedExpressions staged;
  StageMaterialisedValues(schema, exprs, &staged);
  if (!staged.pre.empty()) {
    for (auto& e : exprs) GDV_RETURN_NOT_OK(ValidateExpression(schema, *e));
    GDV_RETURN_NOT_OK(PrecompileProjector(schema, staged.pre, mode));
    CodegenOptions second = CodegenOptions::FromEnv();
    second.rows_word = true;
    GDV_RETURN_NOT_OK(PlanProjector(staged.schema, staged.main, mode, second, &plan,
                                    mode == SelectionMode::kNone ? 0x7fffffff : static_cast<int>(schema.size())));
  } else {
    GDV_RETURN_NOT_OK(PlanProjector(schema, exprs, mode, CodegenOptions::FromEnv(), &plan));
  }
  std::vector<char> code;
  GDV_RETURN_NOT_OK(Runtime::Get().CompileToCodeObject(plan.source, plan.kernel_name, &code));
  // the variant without the optimistic flat path is otherwise compiled only when a batch needs it
  // (offline tool path, not reachable from Evaluate)
  if (!plan.source_general.empty() && std::getenv("GDV_PRECOMPILE_SKIP_GENERAL") == nullptr)
    GDV_RETURN_NOT_OK(Runtime::Get().CompileToCodeObject(plan.source_general, plan.kernel_name_general, &code));
  if (plan.prepass)
    GDV_RETURN_NOT_OK(Runtime::Get().CompileToCodeObject(plan.prepass->source, plan.prepass->kernel_name, &code));
  if (plan.exact && std::getenv("GDV_PRECOMPILE_SKIP_GENERAL") == nullptr) {  // (as the general variant: on demand at run time)
    GDV_RETURN_NOT_OK(Runtime::Get().CompileToCodeObject(plan.exact->source, plan.exact->kernel_name, &code));
    if (plan.exact->prepass)
      GDV_RETURN_NOT_OK(Runtime::Get().CompileToCodeObject(plan.exact->prepass->source, plan.exact->prepass->kernel_name, &code));
  }
  return Status::OK();
}

Status PrecompileFilter(const Schema& schema, const ExpressionPtr& condition) {
  KernelPlan plan;
  StagedExpressions staged;
  StageMaterialisedValues(schema, {condition}, &staged);
  if (!staged.pre.empty()) {
    GDV_RETURN_NOT_OK(ValidateExpression(schema, *condition));
    GDV_RETURN_NOT_OK(PrecompileProjector(schema, staged.pre, SelectionMode::kNone));
    GDV_RETURN_NOT_OK(PlanFilter(staged.schema, staged.main[0], CodegenOptions::FromEnv(), &plan));
  } else {
    GDV_RETURN_NOT_OK(PlanFilter(schema, condition, CodegenOptions::FromEnv(), &plan));
  }
  std::vector<char> code;
  return Runtime::Get().CompileToCodeObject(plan.source, plan.kernel_name, &code);
}

Status PrecompileFilterProject(const Schema& schema, const ExpressionPtr& condition,
                               const std::vector<ExpressionPtr>& exprs, SelectionMode index_mode) {
  KernelPlan plan;
  GDV_RETURN_NOT_OK(PlanFilterProject(schema, condition, exprs, index_mode, CodegenOptions::FromEnv(), &plan));
  std::vector<char> code;
  GDV_RETURN_NOT_OK(Runtime::Get().CompileToCodeObject(plan.source, plan.kernel_name, &code));
  if (plan.exact)  // round 5: the direct kernel behind the windowed one
    GDV_RETURN_NOT_OK(Runtime::Get().CompileToCodeObject(plan.exact->source, plan.exact->kernel_name, &code));
  return Status::OK();
}

}  // namespace gdv
