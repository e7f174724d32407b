// Projector / Filter: the objects behind gandiva::Projector::{Make,Evaluate} and
// gandiva::Filter::{Make,Evaluate} (libgandiva.pxd:214-256), expressed over raw Arrow
// buffers so that the same code serves the C-ABI (include/gandiva_amd.h), the gandiva::
// C++ layer and the Python mirror.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gdv_node.h"
#include "gdv_planner.h"
#include "gdv_runtime.h"
#include "gdv_tier0.h"

namespace gdv {

enum class MemKind : int32_t {
  kHost = 0,    // buffers are host memory: staged to HBM, results copied back (correctness path)
  kDevice = 1,  // buffers are HBM-resident on the current device: zero-copy (the fast path)
};

// One Arrow array as raw buffers (pyarrow/include/arrow/array/data.h:85-95).
struct ColumnBuffers {
  const void* validity = nullptr;  // may be null: no nulls
  int64_t validity_size = 0;
  const void* data = nullptr;      // fixed-width values | bool bits | var-len bytes
  int64_t data_size = 0;
  const void* offsets = nullptr;   // var-len only (int32 offsets)
  int64_t offsets_size = 0;
  int64_t offset = 0;              // Arrow array offset, in rows
};

struct OutputBuffers {
  void* validity = nullptr;  // >= 8 * ceil(rows / 64) bytes
  int64_t validity_size = 0;
  void* data = nullptr;      // >= rows * width bytes (bool: 8 * ceil(rows / 64)); var-len: bytes
  int64_t data_size = 0;     // var-len: capacity in; bytes needed out (also on failure)
  void* offsets = nullptr;   // var-len outputs only: (rows + 1) int32 offsets
  int64_t offsets_size = 0;
};

struct SelectionView {
  SelectionMode mode = SelectionMode::kNone;
  const void* indices = nullptr;
  int64_t num_slots = 0;
  // Device-resident slot count (round 3): when not null, the kernel reads the number of slots from
  // this int64 in device (or pinned) memory — where an asynchronous Filter::Evaluate left it — and
  // `num_slots` is only the capacity the outputs and the launch are sized for.  Device buffers,
  // fixed-width outputs only.
  const void* num_slots_device = nullptr;
};

struct Configuration {
  bool optimize = true;
  bool dump_ir = false;
};

enum EvalFlags : uint32_t {
  kEvalAsync = 1u,    // device buffers only: return after enqueueing on `stream`
  kEvalNoSmall = 2u,  // internal: a batch-by-batch fallback must not re-enter the fused small-batch path
  kEvalStaged = 4u,   // internal: the columns already hold the first stage's temporaries (asynchronous two-stage plans)
};

// What a built plan owns on ONE device context (round 3: a Projector / Filter can be evaluated by
// threads that have selected different devices; each context loads the code objects and holds
// the constant block on its own GPU, lazily, the first time the plan runs there).
struct PlanDeviceState {
  // (atomic since round 6: a tier-0 plan's state exists before its kernel is compiled; the kernel is loaded into it later)
  std::atomic<const CompiledKernel*> kernel{nullptr};
  std::atomic<const CompiledKernel*> kernel_general{nullptr};  // fallback variant (compiled on demand)
  const CompiledKernel* kernel_pre = nullptr;  // wave-shaped plans: the pre-pass kernel
  // the exact variant of a wave-shaped plan (main + pre-pass), compiled when a batch first needs it
  std::atomic<const CompiledKernel*> kernel_exact{nullptr}, kernel_pre_exact{nullptr};
  DeviceBuffer consts, consts_pre;  // string literals / patterns / IN tables (gdv_args::aux0)
};
class PlanDeviceStates {
 public:
  PlanDeviceStates() { for (auto& s : slots_) s.store(nullptr); }
  ~PlanDeviceStates() { for (auto& s : slots_) delete s.load(); }
  // the state of `plan` on the calling thread's context, created on first use
  // need_kernel = false (tier 0): the state may come back without its specialised kernel (kernel == nullptr: still
  // compiling); a later call with need_kernel = true compiles / loads it
  Status Get(const KernelPlan& plan, const PlanDeviceState** out, bool need_kernel = true) const;

 private:
  mutable std::mutex mu_;
  mutable std::atomic<PlanDeviceState*> slots_[Runtime::kMaxDevices];
};

class Projector {
 public:
  static Status Make(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                     SelectionMode mode, const Configuration& config,
                     std::shared_ptr<Projector>* out);

  // `cols` has one entry per schema field (unused fields may be empty).  With a selection
  // view, outputs have sel->num_slots rows, else num_rows rows.
  // (rows_word, internal: device word second-stage kernels take their row count from — the gate of an
  // asynchronous two-stage evaluation)
  // (err_word, internal: a pre-zeroed device word the kernel raises its error bits into INSTEAD of a word of this
  // call's own — the caller reads it, so a plan that can raise may still be enqueued without a wait: the second
  // stage of an asynchronous two-stage evaluation reports through result[0])
  Status Evaluate(int64_t num_rows, const ColumnBuffers* cols, int num_cols,
                  const SelectionView* sel, OutputBuffers* outs, int num_outs, MemKind mem,
                  hipStream_t stream, uint32_t flags, const void* rows_word = nullptr, void* err_word = nullptr) const;

  // Many (small) HBM-resident batches in ONE launch (round 3): the reference is fed 4K-64K-row
  // batches, where a launch + argument marshalling per batch is all overhead.  The argument blocks
  // of all batches travel as one table (one H2D copy), blockIdx.y picks the batch.  Row-mode plans
  // with fixed-width outputs; anything else is evaluated batch by batch on the same stream.
  struct BatchView {
    int64_t num_rows = 0;
    const ColumnBuffers* cols = nullptr;
    int num_cols = 0;
    OutputBuffers* outs = nullptr;
    int num_outs = 0;
  };
  Status EvaluateMany(const BatchView* batches, int num_batches, hipStream_t stream, uint32_t flags) const;

  // Var-len plans WITHOUT a host synchronisation (round 4; device buffers, single-stage plans).
  // Everything is enqueued on `stream`: the kernels of the path this Projector is currently on (the
  // optimistic wave pair, its exact variant or the scanner-shaped kernel; selection-mode plans: the
  // scanner shape, which may take its slot count from sel->num_slots_device), then `result`
  // ((1 + num_outs) x uint64 in device or pinned memory) receives, in stream order,
  //   [0]      the device error word: 0 = the outputs are complete; any bit = discard them and evaluate
  //            the batch with the synchronous call (a raised error, or an optimistic assumption —
  //            ASCII, flat — that did not hold; the synchronous call then also moves the Projector to
  //            the kernels that take such batches)
  //   [1 + e]  the bytes output e produced (fixed-width outputs: 0).  More than the capacity
  //            outs[e].data_size means the buffer was too small: nothing was written past it.
  // Scratch goes back to the pool behind the stream.  outs[e].data_size is left as it was.
  // Two-stage plans (var-len outputs; round 4): both stages are enqueued; the temporaries are sized from
  // what recent batches produced (the first guess before any).  A device-side gate between the stages
  // gives the second stage 0 rows when the first did not complete (status bits, or a temporary too
  // small: bit 128 in result[0]) — it then touches nothing, and the caller re-runs synchronously.
  // (round 5: a second stage — or a single-stage fixed-width plan — that can raise raises into result[0] itself)
  Status EvaluateAsync(int64_t num_rows, const ColumnBuffers* cols, int num_cols, const SelectionView* sel,
                       OutputBuffers* outs, int num_outs, hipStream_t stream, void* result) const;

 private:
  // rows_word (may be null): device word the kernels take their row count from (GDV_ROWS; plans of a
  // second stage and selection-mode plans read it)
  Status EvaluateAsyncStage(int64_t num_rows, const ColumnBuffers* cols, int num_cols, const SelectionView* sel,
                            OutputBuffers* outs, int num_outs, hipStream_t stream, void* result, const void* rows_word) const;
  Status EvaluateAsyncTwoStage(int64_t num_rows, const ColumnBuffers* cols, int num_cols, const SelectionView* sel,
                               OutputBuffers* outs, int num_outs, hipStream_t stream, void* result) const;

 public:

  const Schema& schema() const { return schema_; }
  const KernelPlan& plan() const { return plan_; }
  const std::shared_ptr<Projector>& first_stage() const { return pre_; }
  int num_outputs() const { return static_cast<int>(plan_.output_types.size()); }
  const DataType& output_type(int i) const { return plan_.output_types[i]; }
  std::string DumpIR() const { return pre_ ? pre_->DumpIR() + plan_.ir : plan_.ir; }

  static int64_t ValidityBytes(int64_t rows) { return ((rows + 63) / 64) * 8; }
  static int64_t DataBytes(const DataType& t, int64_t rows) {
    return t.id == kBool ? ValidityBytes(rows) : rows * t.byte_width();
  }

 private:
  Schema schema_;
  KernelPlan plan_;
  PlanDeviceStates states_;  // code objects + constant block per device context
  // Which kernels the NEXT var-len batch starts on (round 4: per batch, no longer sticky for good):
  // 0 = optimistic (ASCII + flat assumed), 1 = the wave shape's exact variant (the last batch held
  // bytes >= 0x80), 2 = the scanner-shaped general kernel (a NULL row carried bytes under a flat
  // output; the optimistic kernels are tried again every 16th batch).
  mutable std::atomic<int> path_hint_{0};
  mutable std::atomic<uint32_t> general_batches_{0};
  // Tier 0 (round 6): the post-fix program of this plan for the ahead-of-time interpreter kernel (null: the plan has
  // none), and whether the specialised kernel is still being compiled in the background
  std::unique_ptr<tier0::Args> tier0_;
  mutable std::atomic<bool> tier0_pending_{false};
  bool UseTier0() const;

 public:
  int path_hint() const { return path_hint_.load(std::memory_order_relaxed); }

 private:
  // two-stage plans (StageMaterialisedValues): pre_ materialises the hoisted sub-trees as
  // temporary columns, plan_ is built over plan_schema_ = schema_ + those columns
  std::shared_ptr<Projector> pre_;
  Schema plan_schema_;
  mutable std::vector<std::atomic<int64_t>> stage_hints_;  // sizes of the first stage's temporaries, learnt from the last batch
  // var-len outputs: bytes per row (x 16) recent batches produced — a decaying maximum (rises at once,
  // sinks by an eighth per batch towards what that batch produced); 0 = no batch yet
  mutable std::vector<std::atomic<int64_t>> out_bytes_x16_;

 public:
  // A capacity HINT for var-len output i over `rows` rows, from what recent batches produced per row
  // (a decaying maximum + an eighth of head room; 0 before the first batch).  Callers that size their byte buffer by it avoid the
  // "too small -> bytes needed -> retry" round trip on every batch after the first.
  int64_t VarlenBytesHint(int i, int64_t rows) const;
};

// Temporary columns of a two-stage plan: the first-stage Projector's outputs, kept in the
// memory kind of the call, appended to the caller's columns for the second stage.
struct StageColumns {
  std::vector<std::unique_ptr<DeviceBuffer>> dev;
  std::vector<std::unique_ptr<std::vector<uint8_t>>> host;
  std::vector<ColumnBuffers> cols;  // caller's columns + the temporaries
  // hints (may be null): per first-stage output, bytes per row x 16 the previous batch produced —
  // read to size the temporaries, updated with what this batch produced
  Status Run(const Projector& pre, int64_t num_rows, const ColumnBuffers* in, int num_cols, MemKind mem,
             hipStream_t stream, const SelectionView* sel = nullptr,
             std::vector<std::atomic<int64_t>>* hints = nullptr);
};

class Filter {
 public:
  static Status Make(const Schema& schema, const ExpressionPtr& condition,
                     const Configuration& config, std::shared_ptr<Filter>* out);

  // Fills out_indices (capacity `max_slots`, element type by `mode`) with the ascending
  // positions of rows where the condition is true and valid; *num_selected = count.
  // flags & kEvalAsync (device buffers, plans that cannot raise): enqueue on `stream` and return;
  // the count then arrives in *count_out (8 bytes of device or pinned memory, int64) in stream
  // order and *num_selected is set to -1.  count_out may also be given to a synchronous call.
  // row_base (round 6, sharded evaluation): added to every emitted position — shard s of a logical batch emits
  // lo_s + local position, so the shards' vectors concatenate into a globally ascending one without a pass of their own
  // (the index type must address row_base + num_rows).
  Status Evaluate(int64_t num_rows, const ColumnBuffers* cols, int num_cols, SelectionMode mode,
                  void* out_indices, int64_t max_slots, int64_t* num_selected, MemKind mem,
                  hipStream_t stream, uint32_t flags = 0, void* count_out = nullptr, int64_t row_base = 0) const;

  // Many small HBM-resident batches in ONE launch (round 3): every batch is filtered by one
  // workgroup that runs predicate, offsets scan and index emission back to back
  // (gdv_small_filter_finish).  counts_host (may be null with kEvalAsync) / counts_device (may be
  // null; int64[num_batches] in device or pinned memory) receive the selected-row counts.
  // Batches too big for one workgroup, string plans and two-stage plans go batch by batch.
  struct BatchView {
    int64_t num_rows = 0;
    const ColumnBuffers* cols = nullptr;
    int num_cols = 0;
    void* out_indices = nullptr;
    int64_t max_slots = 0;
  };
  Status EvaluateMany(const BatchView* batches, int num_batches, SelectionMode mode, int64_t* counts_host,
                      void* counts_device, hipStream_t stream, uint32_t flags) const;
  // rows one workgroup takes (0: the plan has no small-batch entry point)
  int64_t SmallBatchRows() const;

  const Schema& schema() const { return schema_; }
  const KernelPlan& plan() const { return plan_; }
  const std::shared_ptr<Projector>& first_stage() const { return pre_; }
  std::string DumpIR() const { return pre_ ? pre_->DumpIR() + plan_.ir : plan_.ir; }

  // Per-object tuning, for tests and measurements (gdv_filter_set_tuning); never read from the
  // environment during Evaluate.  "chunks": cut big HBM-resident batches into n pipelined chunks
  // (1 = off, the default: measured slower); "small_filter": 0 keeps small batches on the
  // three-launch path.  Defaults come from GDV_FILTER_CHUNKS / GDV_NO_SMALL_FILTER once, at Make.
  Status SetTuning(const std::string& key, int64_t value);

 private:
  std::atomic<int> chunks_{1};
  std::atomic<bool> small_filter_{true};
  std::unique_ptr<tier0::Args> tier0_;  // as Projector's
  mutable std::atomic<bool> tier0_pending_{false};
  bool UseTier0() const;
  Schema schema_;
  KernelPlan plan_;
  PlanDeviceStates states_;  // code objects + constant block per device context
  std::shared_ptr<Projector> pre_;  // two-stage plans, as in Projector
  Schema plan_schema_;
  mutable std::vector<std::atomic<int64_t>> stage_hints_;
};

// Filter -> SelectionVector -> Projector(selection) in ONE kernel (round 4): the condition, the
// output base of every workgroup tile (decoupled look-back) and the projections of the selected
// rows, stored compacted — the batch is read once (gdv_planner.cc, PlanFilterProject).  Results are
// bit-identical to Filter::Evaluate followed by a selection-mode Projector::Evaluate.  Fixed-width /
// bool outputs over fixed-width columns; Make returns CodeGenError for anything else and callers
// chain the two operators (gandiva::FilterProject and gandiva_amd.make_filter_project do).
class FilterProject {
 public:
  // index_mode: element type of the selection vector the evaluation ALSO emits; kNone = none.
  static Status Make(const Schema& schema, const ExpressionPtr& condition, const std::vector<ExpressionPtr>& exprs,
                     SelectionMode index_mode, const Configuration& config, std::shared_ptr<FilterProject>* out);
  ~FilterProject();

  // outs[e]: buffers for up to num_rows rows (the count is only known afterwards).  out_indices
  // (max_slots >= num_rows elements of the index mode) may be null when index_mode is kNone.
  // *num_selected = rows produced.  flags & kEvalAsync (device buffers): everything is enqueued, the
  // count lands in *count_out (8 bytes of device or pinned memory) in stream order, *num_selected = -1
  // and device-side errors (divide by zero ...) are NOT reported — plans that can raise wait anyway.
  // Round 5: a launch whose look-back gave up leaves -1 in *count_out (the outputs are not complete: evaluate
  // the batch with the synchronous call, which re-runs such a launch on the filter + projector chain).
  // which_kernel(): 0 = the windowed kernel runs next, 1 = the direct one, -1 = the plan has one shape only.
  Status Evaluate(int64_t num_rows, const ColumnBuffers* cols, int num_cols, OutputBuffers* outs, int num_outs,
                  void* out_indices, int64_t max_slots, int64_t* num_selected, MemKind mem, hipStream_t stream,
                  uint32_t flags = 0, void* count_out = nullptr) const;

  const Schema& schema() const { return schema_; }
  const KernelPlan& plan() const { return plan_; }
  int num_outputs() const { return static_cast<int>(plan_.output_types.size()); }
  const DataType& output_type(int i) const { return plan_.output_types[i]; }
  SelectionMode index_mode() const { return plan_.mode; }
  std::string DumpIR() const { return plan_.ir; }
  int which_kernel() const;
  // "kernel" (-1 follow the selectivity, 0 windowed, 1 direct): pin the shape (tests, measurements)
  Status SetTuning(const std::string& key, int64_t value);

 private:
  // the kernel launch of one evaluation; *stalled = the look-back gave up (the outputs are not complete)
  Status EvaluateFused(int64_t num_rows, const ColumnBuffers* cols, int num_cols, OutputBuffers* outs, int num_outs,
                       void* out_indices, int64_t max_slots, int64_t* num_selected, MemKind mem, hipStream_t stream,
                       uint32_t flags, void* count_out, bool* stalled) const;
  // Filter::Evaluate + selection-mode Projector::Evaluate over the same buffers: what a stalled fused launch is
  // re-run on (round 5; both operators are built on first need)
  Status EvaluateChain(int64_t num_rows, const ColumnBuffers* cols, int num_cols, OutputBuffers* outs, int num_outs,
                       void* out_indices, int64_t max_slots, int64_t* num_selected, MemKind mem, hipStream_t stream,
                       void* count_out) const;
  Schema schema_;
  ExpressionPtr condition_;
  std::vector<ExpressionPtr> exprs_;
  KernelPlan plan_;
  bool raises_ = false;  // some expression can raise: asynchronous calls wait for the error word
  PlanDeviceStates states_;
  // Round 5: plan_ is the WINDOWED kernel (selected rows staged in LDS, GDV_FP_CAP per wave tile) where that
  // shape exists, plan_.exact the direct round-4 kernel.  Synchronous evaluations record the share of rows they
  // selected (x 1024); once it is beyond what the window holds, the next batches run on the direct kernel.
  mutable std::atomic<int> selected_per_1024_{-1};
  // asynchronous evaluations never see their count on the host: the count also lands in this pinned word, and the NEXT
  // call reads what the previous one left there (one batch late is early enough to pick the kernel shape)
  mutable std::atomic<int64_t*> pinned_count_{nullptr};
  mutable std::atomic<int> resident_per_cu_{0};
  std::atomic<int> pinned_kernel_{-1};  // pipelined shape: workgroups of the kernel one CU holds at once (queried once)
  mutable std::mutex chain_mu_;
  mutable std::shared_ptr<Filter> chain_filter_;
  mutable std::shared_ptr<Projector> chain_projector_;
};

// Builds the plan and compiles it to a gfx950 code object without touching a device
// (used by the build check and to pre-populate the on-disk kernel cache).
Status PrecompileProjector(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                           SelectionMode mode);
Status PrecompileFilter(const Schema& schema, const ExpressionPtr& condition);
// tier 0 (round 6): the post-fix program the interpreter kernel would run for these expressions / this condition
Status Tier0Describe(const Schema& schema, const std::vector<ExpressionPtr>& exprs, bool is_condition, std::string* text);
Status PrecompileFilterProject(const Schema& schema, const ExpressionPtr& condition,
                               const std::vector<ExpressionPtr>& exprs, SelectionMode index_mode);

}  // namespace gdv
