// Planner: expression trees -> one fused HIP kernel per Projector / Filter.
//
// Replaces the reference's ExprValidator + ExprDecomposer + Annotator + LLVMGenerator
// (SURVEY.md §2 rows 3, 4, 6).  The reference's central idea is kept: every expression is
// split into a VALUE computation, evaluated for every row, and a VALIDITY computation that
// for null-if-null functions is just the intersection of the input validity bitmaps.  What
// changes is the execution shape: instead of one scalar row loop per expression, ALL
// expressions of a Projector are fused into one kernel that reads every referenced column
// once, and validity is merged per 64-row word in scalar registers.
#pragma once
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "gdv_node.h"
#include "gdv_registry.h"

namespace gdv {

enum class SelectionMode : int32_t { kNone = 0, kUInt16 = 1, kUInt32 = 2, kUInt64 = 3 };

enum class KernelKind { kProject, kFilter, kFilterProject };

// Code-generation knobs (part of the cache key).  Defaults come from measurements on
// MI355X; environment variables GDV_U / GDV_NT / GDV_WAVES override
// them for sweeps.
struct CodegenOptions {
  int subtiles = 4;        // 64-row sub-tiles each wavefront handles per tile (loads in flight)
  int waves = 4;           // wavefronts per workgroup
  bool nontemporal = true; // non-temporal stores for output value buffers
  bool nt_loads = true;    // non-temporal loads of input value buffers: every value is read
                           // exactly once (+3 % on C2, +7 % on C1, neutral on C3; GDV_NTLOAD=0)
  // Wave-shaped string kernels: the byte sweep of a sub-tile also drops the bytes into an LDS mirror
  // and the staged copy of the rows reads them from there instead of going back to L2, which the
  // lines have left by then (profiles/r03_c5_traffic.txt).  GDV_NO_LDS_MIRROR=1 switches it off.
  bool lds_mirror = true;
  // Read from the environment ONCE, at Make (FromEnv); nothing below is looked up during Evaluate.
  bool subtiles_forced = false, waves_forced = false;  // GDV_U / GDV_WAVES were given: the planner keeps them
  bool no_inline_string_args = false;  // GDV_NO_INLINE_STRING_ARGS
  bool no_wave_shape = false;          // GDV_NO_WAVE_SHAPE: var-len plans take the scanner shape
  bool wave_bytefree_only = false;     // GDV_WAVE_BYTEFREE_ONLY
  bool ablation = false;               // GDV_ABLATION=1: emit the GDV_ABL experiment branches into the kernels
  bool prepass_rolled = false;         // GDV_PREPASS_ROLLED=1: keep the row loop of optimistic offsets-only pre-passes rolled
  // Second stage of a two-stage plan (set by the engine, not from the environment): the kernel takes its row
  // count from the device word aux2 points at when there is one — an asynchronous evaluation's gate writes 0
  // there when the first stage did not complete, and the second stage then touches nothing.
  bool rows_word = false;
  int sweep_group = 1;                 // GDV_SWEEP_GROUP: sub-tiles whose spans a wave-shaped main kernel sweeps in ONE go (1 = rounds 3-5: one at a time; 4: three full 1024-byte steps per four sub-tiles of 12-byte rows instead of four three-quarter-full ones)
  bool cast_x86_indefinite = false;    // GDV_CAST_X86_INDEFINITE=1 at Make: float -> integer casts of NaN / out-of-range values give the x86 "indefinite integer" (0x80..0) instead of saturating
  // Fused filter-project, windowed shape (round 5): bytes of LDS window per wave tile (every windowed output + the
  // row index, GDV_FP_CAP rows of them); 0 = the direct round-4 shape only.  GDV_FP_WINDOW=<bytes>.
  int fp_window_bytes = 9984;
  int fp_rounds = 3;                   // GDV_FP_K: rounds of GDV_U sub-tiles per wave tile of the windowed shape (one look-back per K x 8192 rows)
  int fp_experiment = 0;               // GDV_FP_EXPERIMENT=<n>: timing experiments on the fused kernel (wrong results; tools only)
  // exact pre-pass of the wave kernels (round 5): all GDV_U sub-tile spans of a wave tile are swept before the row loop, their
  // first pieces requested back to back (GDV_U loads in flight per lane; one bitmap per sub-tile in LDS).  GDV_PREPASS_AHEAD=0:
  // one span per row-loop iteration, the next one's first piece prefetched (round 4).
  bool prepass_ahead = true;
  bool no_sel_wave = false;            // GDV_NO_SEL_WAVE=1: selection-mode var-len plans take the scanner shape (rounds 2-4)
  bool runtime_needles = false;        // GDV_RUNTIME_NEEDLES=1: wave kernels load their '%needle%' bytes instead of carrying them as immediates
  static CodegenOptions FromEnv();
  std::string Key() const;
};

// Byte layout of the single by-value kernel argument (struct gdv_args in the generated
// source).  Host (gdv_engine.cc) and device agree on this through these offsets only.
struct ArgLayout {
  static constexpr int kHeaderBytes = 64;  // n, err, sel, mask, counts, aux0..2
  static constexpr int kOffN = 0, kOffErr = 8, kOffSel = 16, kOffMask = 24, kOffCounts = 32,
                       kOffAux0 = 40, kOffAux1 = 48, kOffAux2 = 56;
  int n_in = 0, n_out = 0, n_lit = 0;
  // per input slot: data ptr (8) | validity gdv_bitmap (24) | value-bits gdv_bitmap (24) | offsets ptr (8)
  static constexpr int kInStride = 64;
  // per output slot: data ptr (8) | validity ptr (8) | offsets ptr (8) | capacity (8)
  static constexpr int kOutStride = 32;  // ... | byte capacity of a var-len data buffer (8)
  int in_base() const { return kHeaderBytes; }
  int out_base() const { return kHeaderBytes + std::max(n_in, 1) * kInStride; }
  int lit_base() const { return out_base() + std::max(n_out, 1) * kOutStride; }
  int total() const { return lit_base() + std::max(n_lit, 1) * 8; }
};

struct KernelPlan {
  KernelKind kind = KernelKind::kProject;
  SelectionMode mode = SelectionMode::kNone;
  CodegenOptions opts;
  std::string kernel_name;
  std::string source;              // complete HIP translation unit (minus the library header)
  std::string ir;                  // human-readable plan dump (DumpIR)
  // plans with flat var-len outputs: the variant without the optimistic flat path (compiled on demand)
  std::string kernel_name_general, source_general;
  std::vector<int> input_fields;   // input slot -> index into the schema
  std::vector<bool> input_needs_values;
  std::vector<bool> input_needs_validity;
  std::vector<DataType> output_types;  // one per expression (filter: none)
  ArgLayout layout;
  std::vector<uint64_t> literals;  // fixed-width literal values -> gdv_args::lit (kernel arguments)
  std::string const_block;         // string literals, LIKE patterns, IN tables -> device memory (aux0)
  bool can_raise = false;          // kernel may set error bits
  bool exprs_raise = false;        // fused filter-project: some EXPRESSION can raise (can_raise is always set there: the look-back's stall bit)
  // Some output is utf8/binary: single launch, workgroup 0 scans the tile totals (granules in
  // `mask`, grand totals in `counts`); workers are workgroups 1.. (gdv_planner.cc, string plans)
  bool has_varlen_output = false;
  bool has_varlen_input = false;   // some expression reads utf8/binary bytes
  bool string_skeleton = false;    // tile = workgroup (waves x subtiles x 64 rows), no grid-stride
  int num_varlen_outputs = 0;
  bool has_flat_output = false;    // some var-len output is an input column's (mapped) bytes
  // Wave shape (round 3): tile = ONE wave (64 * subtiles rows), no scanner workgroup.  Var-len
  // outputs that are not flat take their wave-tile bases from `prepass` (a kernel of its own: byte
  // totals per wave tile from the offsets alone) + the offsets scan; wave_segments[v] = the
  // segment of var-len output v, -1 for flat outputs.  source_general / kernel_name_general hold
  // the scanner-shaped fallback (a batch that breaks the ASCII / flat assumption is re-run on it).
  bool has_small_entry = false;  // filters: <kernel_name>_small(table): predicate + scan + emission, one workgroup per batch
  bool has_many_entry = false;  // the code object also holds <kernel_name>_many(const gdv_args* table): one launch, many batches
  int compact_from = 0x7fffffff;  // selection mode: schema fields from here on are compact temporaries
  bool wave_tiles = false;
  // Round 4: the EXACT variant of a wave-shaped plan whose row bodies consult the ASCII flag (main
  // kernel; its own pre-pass hangs off exact->prepass).  Same argument blocks as the optimistic pair.
  // A batch that raises NOTASCII is re-run on it — and the next batches start there until one of
  // them turns out to be pure ASCII again.  Null when nothing consults the flag.
  std::shared_ptr<KernelPlan> exact;
  std::shared_ptr<KernelPlan> prepass;
  std::vector<int> wave_segments;
  int fp_rounds = 1;  // fused filter-project: rounds of `subtiles` sub-tiles per wave tile (windowed shape: GDV_FP_K; direct: 1)
  int fp_window_rows = 0;  // fused filter-project, windowed shape: GDV_FP_CAP (0: the direct shape); `exact` = the direct shape
  int grid_blocks_per_cu = 0;  // workgroups per CU of the grid-stride launch the planner asks for (0: the engine's default)
  int general_subtiles = 0, general_waves = 0;  // tile of the scanner-shaped fallback (0: opts')
  int rows_per_tile() const { return 64 * opts.subtiles * (wave_tiles ? 1 : opts.waves); }
};
constexpr int kMaxWaveSegments = 8;

// Validates every expression against the schema and the function registry, then emits the
// fused kernel.  Errors: ExpressionValidationError for type / signature problems,
// CodeGenError for constructs the HIP backend does not cover yet.
// compact_from (selection mode, two-stage plans): schema fields from this index on are the
// temporaries a selection-mode first stage produced — one row per SLOT, read at the slot's own
// position; the caller's columns are gathered through the selection vector.
Status PlanProjector(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                     SelectionMode mode, const CodegenOptions& opts, KernelPlan* out,
                     int compact_from = 0x7fffffff);
Status PlanFilter(const Schema& schema, const ExpressionPtr& condition,
                  const CodegenOptions& opts, KernelPlan* out);

// Fused filter -> project (round 4): ONE kernel evaluates the condition, finds every workgroup tile's
// output base by a decoupled look-back and stores the projections of the selected rows compacted
// (+ the selection vector when index_mode != kNone).  plan->mode = index_mode.  CodeGenError for
// plans the fused shape does not take (var-len columns or outputs): callers chain Filter + Projector.
Status PlanFilterProject(const Schema& schema, const ExpressionPtr& condition, const std::vector<ExpressionPtr>& exprs,
                         SelectionMode index_mode, const CodegenOptions& opts, KernelPlan* out);

Status ValidateExpression(const Schema& schema, const Expression& expr);

// Values that only exist once their bytes are written — concat / || results, lpad / rpad,
// reverse, castVARCHAR(number) — can be a kernel's OUTPUT (or a concat argument) but not the
// argument of another function inside the same kernel.  Where an expression consumes one, the
// sub-tree is hoisted: it becomes an expression of a first-stage Projector that materialises it
// as a temporary utf8 column ("__gdv_stage<k>", appended to the schema), and the consumer reads
// that column.  `pre` stays empty when nothing needs a first stage.
struct StagedExpressions {
  std::vector<ExpressionPtr> pre;   // first stage, over the caller's schema
  Schema schema;                    // caller's schema + one field per first-stage expression
  std::vector<ExpressionPtr> main;  // the caller's expressions, rewritten over `schema`
};
void StageMaterialisedValues(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                             StagedExpressions* out);

}  // namespace gdv
