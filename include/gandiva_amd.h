/*
 * gandiva_amd — C ABI of the MI355X-native Projector / Filter evaluator.
 *
 * This is the drop-in boundary for the one hot path this library replaces:
 * gandiva::Projector::Evaluate / gandiva::Filter::Evaluate over Arrow record batches
 * (SURVEY.md §8).  Everything is plain C: opaque handles, pointers, sizes.  No Arrow C++,
 * torch or HIP types appear in a signature (a stream is passed as `void*`).
 *
 * What each group replaces in the reference (the mounted reference holds no source —
 * SURVEY.md §0 — so citations are to the reference lineage's binding spec that IS present:
 * PA = pyarrow/includes/libgandiva.pxd, and to the JNI boundary recalled in SURVEY.md §2
 * row 18):
 *
 *   gdv_node_* / gdv_expression_*   gandiva::TreeExprBuilder::Make*           PA:110-212
 *                                   (over JNI these trees arrive as protobuf bytes; a C
 *                                   caller builds them with these constructors instead)
 *   gdv_projector_make              gandiva::Projector::Make                  PA:230-240
 *                                   JNI buildProjector(schema, exprs, selMode, configId)
 *   gdv_projector_evaluate          gandiva::Projector::Evaluate              PA:218-226
 *                                   JNI evaluateProjector(moduleId, numRows, bufAddrs[],
 *                                   bufSizes[], selVecType, selVecRows, selVecAddr, …,
 *                                   outAddrs[], outSizes[]) — same raw-address convention
 *   gdv_projector_dump_ir           gandiva::Projector::DumpIR                PA:228
 *   gdv_filter_make                 gandiva::Filter::Make                     PA:252-256
 *   gdv_filter_evaluate             gandiva::Filter::Evaluate +
 *                                   SelectionVector::MakeInt16/32/64          PA:246-248, 58-71
 *                                   JNI evaluateFilter(moduleId, numRows, bufAddrs[],
 *                                   bufSizes[], selVecType, outAddr, outSize) -> count
 *   gdv_registry_*                  gandiva::GetRegisteredFunctionSignatures  PA:274-277
 *   gdv_config_t                    gandiva::Configuration                    PA:279-298
 *   status codes                    arrow::StatusCode 40/41/42  pyarrow/include/arrow/status.h:97-100
 *
 * Memory domains.  Buffers follow the Arrow columnar layout (validity bitmap LSB-first,
 * values, int32 offsets for var-len).  `GDV_MEM_DEVICE` buffers are resident in the HBM of
 * the current HIP device and are used in place (zero-copy; this is the measured path);
 * `GDV_MEM_HOST` buffers are staged through HBM by the library (correctness path).
 * There is no CPU evaluation path: without a HIP device every evaluate call fails with
 * GDV_EXECUTION_ERROR.
 * Device buffers are read in aligned words: a validity / bool bitmap up to the next 8-byte boundary, the
 * bytes of a utf8 / binary column up to the next 16-byte boundary (never across it, so never into another
 * page).  Arrow's builders zero the padding of their buffers; a byte buffer that is followed by other
 * (non-zero) bytes inside that last 16-byte block is still evaluated correctly, but a byte >= 0x80 there
 * makes the string kernels take their exact (UTF-8 aware, slower) variant for the batch.
 *
 * Devices (round 3).  Every call runs on the CALLING THREAD's device: the one it chose with
 * gdv_set_device(), else its current HIP device (hipSetDevice / torch.cuda.set_device).  The
 * library keeps one context per device — loaded code objects, buffer pool, streams — so one
 * process drives all GPUs of a node with one host thread per device; a handle
 * (Projector / Filter) may be evaluated from any of them, it is loaded onto a device the first
 * time it runs there.  Device buffers handed to evaluate must live on (or be peer-accessible
 * from) the calling thread's device.
 *
 * Threading: all functions may be called concurrently; evaluate is re-entrant on one
 * handle (per-call state only), as the reference's `nogil` bindings require (PA:27-279).
 *
 * Errors: functions returning `int` return a gdv_status_code; the message of the last
 * failure on the calling thread is available from gdv_last_error().  Constructors return
 * NULL on failure.
 */
#ifndef GANDIVA_AMD_H_
#define GANDIVA_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Numerically identical to arrow::StatusCode. */
typedef enum {
  GDV_OK = 0,
  GDV_OUT_OF_MEMORY = 1,
  GDV_INVALID = 4,
  GDV_NOT_IMPLEMENTED = 10,
  GDV_CODE_GEN_ERROR = 40,
  GDV_EXPRESSION_VALIDATION_ERROR = 41,
  GDV_EXECUTION_ERROR = 42
} gdv_status_code;

/* Numerically identical to arrow::Type::type (pyarrow/include/arrow/type_fwd.h:330-402). */
typedef enum {
  GDV_TYPE_BOOL = 1,
  GDV_TYPE_UINT8 = 2,
  GDV_TYPE_INT8 = 3,
  GDV_TYPE_UINT16 = 4,
  GDV_TYPE_INT16 = 5,
  GDV_TYPE_UINT32 = 6,
  GDV_TYPE_INT32 = 7,
  GDV_TYPE_UINT64 = 8,
  GDV_TYPE_INT64 = 9,
  GDV_TYPE_FLOAT = 11,
  GDV_TYPE_DOUBLE = 12,
  GDV_TYPE_STRING = 13,
  GDV_TYPE_BINARY = 14,
  GDV_TYPE_DATE32 = 16,
  GDV_TYPE_DATE64 = 17,
  GDV_TYPE_TIMESTAMP = 18,
  GDV_TYPE_TIME32 = 19,
  GDV_TYPE_TIME64 = 20,
  GDV_TYPE_DECIMAL128 = 23
} gdv_type_id;

/* precision: decimal precision, or arrow::TimeUnit (0 s, 1 ms, 2 us, 3 ns) for
 * time32/time64/timestamp; scale: decimal scale. */
typedef struct {
  int32_t id;
  int32_t precision;
  int32_t scale;
} gdv_type_t;

/* gandiva::SelectionVector::Mode (PA:48-56). */
typedef enum {
  GDV_SEL_NONE = 0,
  GDV_SEL_UINT16 = 1,
  GDV_SEL_UINT32 = 2,
  GDV_SEL_UINT64 = 3
} gdv_selection_mode;

typedef enum { GDV_MEM_HOST = 0, GDV_MEM_DEVICE = 1 } gdv_mem_kind;

/* gdv_projector_evaluate flags */
#define GDV_EVAL_ASYNC 1u /* device buffers: enqueue on `stream` and return without waiting
                           * (plans that can raise, or that produce utf8/binary, wait anyway:
                           * the error word / byte totals are read back) */

typedef struct gdv_schema gdv_schema_t;
typedef struct gdv_node gdv_node_t;
typedef struct gdv_expression gdv_expression_t; /* also used for conditions */
typedef struct gdv_projector gdv_projector_t;
typedef struct gdv_filter gdv_filter_t;
typedef struct gdv_filter_project gdv_filter_project_t;

/* gandiva::Configuration (PA:279-298). */
typedef struct {
  int32_t optimize;
  int32_t dump_ir;
} gdv_config_t;

/* One input column = one Arrow array as raw buffers. */
typedef struct {
  const void* validity; /* NULL: no nulls */
  int64_t validity_size;
  const void* data; /* fixed-width values | bool bits | var-len bytes */
  int64_t data_size;
  const void* offsets; /* var-len only: int32 offsets */
  int64_t offsets_size;
  int64_t offset; /* Arrow array offset in rows (pyarrow/include/arrow/array/data.h:88-90) */
} gdv_column_t;

/* One output column, allocated by the caller (as the JNI caller allocates outAddrs[]).
 * Required sizes for `rows` output rows: gdv_projector_output_sizes(). */
typedef struct {
  void* validity;
  int64_t validity_size;
  void* data;        /* fixed-width values | bool bits | var-len bytes */
  int64_t data_size; /* var-len: capacity in; bytes produced (or needed, on GDV_INVALID) out */
  void* offsets;     /* var-len outputs only: (rows + 1) int32 offsets; NULL otherwise */
  int64_t offsets_size;
} gdv_out_column_t;

typedef struct {
  int32_t mode;        /* gdv_selection_mode */
  const void* indices; /* uint16/uint32/uint64 row positions */
  int64_t num_slots;
} gdv_selection_t;

/* ---- errors, version ------------------------------------------------------------- */
const char* gdv_last_error(void);
const char* gdv_version(void);
void gdv_free_string(char* s);

/* ---- schema ---------------------------------------------------------------------- */
gdv_schema_t* gdv_schema_new(void);
int gdv_schema_add_field(gdv_schema_t* schema, const char* name, gdv_type_t type, int nullable);
int gdv_schema_num_fields(const gdv_schema_t* schema);
void gdv_schema_free(gdv_schema_t* schema);

/* ---- expression trees: TreeExprBuilder (PA:110-212) ------------------------------ */
/* MakeField */
gdv_node_t* gdv_node_field(const char* name, gdv_type_t type);
/* MakeLiteral for fixed-width types: `value` points at the little-endian image of the
 * value (1/2/4/8/16 bytes by type; bool: one byte); MakeNull when is_null != 0. */
gdv_node_t* gdv_node_literal(gdv_type_t type, const void* value, int is_null);
/* MakeStringLiteral / MakeBinaryLiteral */
gdv_node_t* gdv_node_literal_bytes(gdv_type_t type, const char* data, int64_t len, int is_null);
/* MakeFunction(name, children, return_type) */
gdv_node_t* gdv_node_function(const char* name, gdv_node_t* const* children, int num_children,
                              gdv_type_t return_type);
/* MakeIf(condition, then, else, return_type) */
gdv_node_t* gdv_node_if(gdv_node_t* condition, gdv_node_t* then_node, gdv_node_t* else_node,
                        gdv_type_t return_type);
/* MakeAnd / MakeOr */
gdv_node_t* gdv_node_and(gdv_node_t* const* children, int num_children);
gdv_node_t* gdv_node_or(gdv_node_t* const* children, int num_children);
/* MakeInExpression{Int32,Int64,Date32,Date64,Time32,Time64,TimeStamp}: `values` holds
 * num_values elements of the type's width. */
gdv_node_t* gdv_node_in(gdv_node_t* node, gdv_type_t value_type, const void* values,
                        int num_values);
/* MakeInExpression{String,Binary} */
gdv_node_t* gdv_node_in_bytes(gdv_node_t* node, gdv_type_t value_type, const char* const* values,
                              const int64_t* lengths, int num_values);
/* Node::ToString / Node::return_type (PA:29-31) */
char* gdv_node_to_string(const gdv_node_t* node);
gdv_type_t gdv_node_return_type(const gdv_node_t* node);
/* Handles are reference-counted views of immutable shared trees: freeing a child handle
 * after it was used to build a parent is fine. */
void gdv_node_free(gdv_node_t* node);

/* MakeExpression(root, result_field) / MakeCondition(root) */
gdv_expression_t* gdv_expression_new(gdv_node_t* root, const char* result_name,
                                     gdv_type_t result_type);
gdv_expression_t* gdv_condition_new(gdv_node_t* root);
char* gdv_expression_to_string(const gdv_expression_t* expr);
gdv_type_t gdv_expression_result_type(const gdv_expression_t* expr);
void gdv_expression_free(gdv_expression_t* expr);

/* ---- Projector -------------------------------------------------------------------- */
int gdv_projector_make(const gdv_schema_t* schema, gdv_expression_t* const* exprs, int num_exprs,
                       int selection_mode, const gdv_config_t* config /* NULL = default */,
                       gdv_projector_t** out);
int gdv_projector_num_outputs(const gdv_projector_t* p);
/* Diagnostics: the kernels the NEXT var-len batch of this projector starts on — 0 the optimistic pair
 * (ASCII + flat outputs assumed), 1 the exact variant of the wave shape (the last batch held bytes
 * >= 0x80), 2 the scanner-shaped general kernel (a NULL row carried bytes under a flat output).
 * Decided per batch: an all-ASCII batch brings the projector back to 0. */
int gdv_projector_path_hint(const gdv_projector_t* p);
gdv_type_t gdv_projector_output_type(const gdv_projector_t* p, int i);
/* Bytes the caller must provide for output i with `rows` output rows.  Device buffers
 * are written in whole 64-bit bitmap words: validity (and bool data) = 8 * ceil(rows/64). */
int gdv_projector_output_sizes(const gdv_projector_t* p, int i, int64_t rows, int mem_kind,
                               int64_t* validity_bytes, int64_t* data_bytes);
/* var-len (utf8/binary) outputs: offsets need (rows + 1) * 4 bytes; *data_bytes above is
 * reported as 0 before this projector has evaluated a batch and afterwards as a capacity HINT
 * (bytes per row recent batches produced, with an eighth of head room: a maximum that rises at
 * once and decays by an eighth per batch towards what that batch produced, so one outlier batch
 * does not size every later buffer — sizing the buffer by it saves the retry below on every batch
 * after the first).  It is not a bound:
 * the byte total is only known once the rows have been evaluated: call evaluate
 * with any capacity; the kernel never writes past it, and when it is too small the call fails with
 * GDV_INVALID and data_size is updated to the bytes needed (the reference's JNI path grows its buffer through an expander callback
 * for the same reason). */
/* cols: one entry per schema field, in schema order.  sel: NULL, or the selection vector
 * (mode must equal the mode given to make).  Output row count = sel ? sel->num_slots
 * : num_rows.  stream: hipStream_t as void* (NULL = default stream). */
int gdv_projector_evaluate(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols,
                           int num_cols, const gdv_selection_t* sel, gdv_out_column_t* outs,
                           int num_outs, int mem_kind, void* stream, uint32_t flags);
/* Many HBM-resident batches in ONE call (round 3).  The reference is fed 4K-64K-row batches by its
 * query engine; at that size a kernel launch and the marshalling of one argument block per batch
 * cost more than the evaluation.  Row-mode plans with fixed-width outputs run ALL batches in one
 * launch (the argument blocks travel as one table, the grid's second dimension picks the batch);
 * other plans are evaluated batch by batch on `stream`.  Device buffers only.  With
 * GDV_EVAL_ASYNC nothing waits (plans that can raise wait anyway). */
typedef struct {
  int64_t num_rows;
  const gdv_column_t* cols; /* one per schema field */
  int num_cols;
  gdv_out_column_t* outs;   /* one per expression */
  int num_outs;
} gdv_batch_t;
int gdv_projector_evaluate_many(const gdv_projector_t* p, const gdv_batch_t* batches, int num_batches, void* stream,
                                uint32_t flags);
/* The same under a selection vector whose slot COUNT is still on the device: the kernel reads the
 * number of slots from *num_slots_device (int64 in device or pinned memory — where
 * gdv_filter_evaluate_async left it); sel->num_slots is only the capacity the outputs and the launch
 * are sized for.  Filter -> project then needs no host round trip between the two calls.  Device
 * buffers, fixed-width outputs; rows of the outputs beyond the real count are not written. */
int gdv_projector_evaluate_selected(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols,
                                    int num_cols, const gdv_selection_t* sel, const void* num_slots_device,
                                    gdv_out_column_t* outs, int num_outs, void* stream, uint32_t flags);
/* Var-len (utf8 / binary) plans WITHOUT a host synchronisation (round 4).  gdv_projector_evaluate reads a
 * var-len plan's byte totals — and whether its optimistic kernels' assumptions held — back before it
 * returns.  This entry point enqueues the kernels of the path the projector is currently on
 * (gdv_projector_path_hint) and returns; `result` — (1 + num_outs) x uint64 in device or pinned host memory —
 * receives, in stream order:
 *   result[0]      the device error word.  0: the outputs are complete.  Anything else (a raised error,
 *                  or an assumption — ASCII, flat — that did not hold for this batch): discard them and
 *                  evaluate the batch with gdv_projector_evaluate, which also moves the projector to the
 *                  kernels that take such batches.
 *   result[1 + e]  the bytes output e produced (0 for fixed-width outputs).  A value above the capacity
 *                  outs[e].data_size means the buffer was too small: nothing was written past it.
 * outs[e].data_size is not updated.  sel may be NULL (row mode) or a selection vector of the projector's
 * mode; with num_slots_device != NULL the slot count is read from that int64 on the device
 * (gdv_filter_evaluate_async left it there) and sel->num_slots is the capacity: filter -> upper(s) without
 * a host round trip.  Device buffers.
 * Two-stage plans (a function over a value that is materialised first: upper(concat(a, b)) ...): both stages
 * are enqueued, the temporaries sized from what earlier batches produced; a one-thread gate kernel between the
 * stages hands the second stage its row count — 0 when the first stage did not complete or a temporary was too
 * small (bit 128 of result[0]), so that it touches nothing.  A second stage with fixed-width outputs only is
 * launched the same way; if it can raise (a division, a text -> integer cast over the staged value ...) it raises
 * into result[0] itself (round 5 — as does a single-stage fixed-width plan that can raise).  Plans with more than two
 * stages: GDV_INVALID (evaluate them synchronously). */
int gdv_projector_evaluate_async(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                 const gdv_selection_t* sel, const void* num_slots_device, gdv_out_column_t* outs,
                                 int num_outs, void* stream, void* result);
char* gdv_projector_dump_ir(const gdv_projector_t* p);
void gdv_projector_free(gdv_projector_t* p);

/* ---- Filter ----------------------------------------------------------------------- */
int gdv_filter_make(const gdv_schema_t* schema, gdv_expression_t* condition,
                    const gdv_config_t* config, gdv_filter_t** out);
/* out_indices: caller-allocated selection vector of `max_slots` (>= num_rows) elements of
 * the width `selection_mode` implies; *num_selected receives the slot count.  Indices are
 * ascending.  Rows whose predicate is null are not selected. */
int gdv_filter_evaluate(const gdv_filter_t* f, int64_t num_rows, const gdv_column_t* cols,
                        int num_cols, int selection_mode, void* out_indices, int64_t max_slots,
                        int64_t* num_selected, int mem_kind, void* stream);
/* Asynchronous variant for HBM-resident batches: everything is enqueued on `stream` and the call
 * returns without waiting; the selected-row count (int64) lands in *num_selected_device — 8 bytes of
 * device or pinned host memory — in stream order.  Plans that can raise (divide, mod ...) and plans
 * with a materialising first stage wait like gdv_filter_evaluate does (the count is written either way). */
int gdv_filter_evaluate_async(const gdv_filter_t* f, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                              int selection_mode, void* out_indices, int64_t max_slots, void* num_selected_device,
                              void* stream);
/* Many small HBM-resident batches in ONE launch: each batch is filtered by one workgroup that runs
 * the predicate, the offsets scan and the index emission back to back (batches of up to 2^17 rows;
 * bigger ones, string plans and two-stage plans are evaluated one by one on `stream`).  A single
 * small batch handed to gdv_filter_evaluate takes the same kernel.  num_selected (host array, may
 * be NULL with GDV_EVAL_ASYNC) / num_selected_device (int64[num_batches] in device or pinned
 * memory, may be NULL) receive the counts; with GDV_EVAL_ASYNC and a device array nothing waits. */
typedef struct {
  int64_t num_rows;
  const gdv_column_t* cols; /* one per schema field */
  int num_cols;
  void* out_indices;        /* max_slots elements of the selection mode's width */
  int64_t max_slots;        /* >= num_rows */
} gdv_filter_batch_t;
int gdv_filter_evaluate_many(const gdv_filter_t* f, const gdv_filter_batch_t* batches, int num_batches,
                             int selection_mode, int64_t* num_selected, void* num_selected_device, void* stream,
                             uint32_t flags);
char* gdv_filter_dump_ir(const gdv_filter_t* f);
void gdv_filter_free(gdv_filter_t* f);
/* Per-object tuning for tests and measurements (nothing on the Evaluate path reads the environment):
 *   "chunks"        1..64  cut big HBM-resident batches into n pipelined chunks (default 1 = off)
 *   "small_filter"  0 / 1  small batches filtered by one workgroup in one launch (default 1)
 * Defaults are taken once, at Make, from GDV_FILTER_CHUNKS / GDV_NO_SMALL_FILTER.  Filters are
 * cached per (schema, condition): the setting applies to every holder of the same plan. */
int gdv_filter_set_tuning(gdv_filter_t* f, const char* key, int64_t value);

/* ---- Filter -> Projector in one pass (round 4) ---------------------------------------
 * Replaces the caller-side chain of the reference (pyarrow/tests/test_gandiva.py:329-373:
 * Filter::Evaluate -> SelectionVector -> Projector::Evaluate(batch, selection_vector)) by ONE kernel
 * that reads the batch once: the condition, the output position of every selected row (decoupled
 * look-back across workgroup tiles) and the projections of the selected rows, stored compacted.
 * Results are bit-identical to the chain.  index_mode = GDV_SEL_UINT16/32/64: the selection vector is
 * ALSO written to out_indices; GDV_SEL_NONE: only the projected columns are produced.
 * make fails with GDV_CODE_GEN_ERROR for plans the fused shape does not take (var-len columns or
 * outputs, materialised values): chain gdv_filter_* and gdv_projector_* as before.
 * evaluate: outs[e] must hold num_rows rows (the count is known only afterwards; sizes as
 * gdv_projector_output_sizes reports for num_rows rows); *num_selected receives the row count.
 * GDV_EVAL_ASYNC (device buffers, num_selected_device given, plans that cannot raise): everything
 * is enqueued on `stream`, the count lands in *num_selected_device (int64, device or pinned memory)
 * in stream order and *num_selected is set to -1. */
int gdv_filter_project_make(const gdv_schema_t* schema, gdv_expression_t* condition, gdv_expression_t* const* exprs,
                            int num_exprs, int index_mode, const gdv_config_t* config /* NULL = default */,
                            gdv_filter_project_t** out);
int gdv_filter_project_num_outputs(const gdv_filter_project_t* fp);
gdv_type_t gdv_filter_project_output_type(const gdv_filter_project_t* fp, int i);
int gdv_filter_project_evaluate(const gdv_filter_project_t* fp, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                gdv_out_column_t* outs, int num_outs, void* out_indices /* NULL with GDV_SEL_NONE */,
                                int64_t max_slots, int64_t* num_selected, void* num_selected_device /* may be NULL */,
                                int mem_kind, void* stream, uint32_t flags);
char* gdv_filter_project_dump_ir(const gdv_filter_project_t* fp);
/* Round 5.  The fused plan has two kernels: the WINDOWED one (selected rows are staged, at their rank, in a
 * wave-private LDS window and leave with full-width stores once the look-back has returned the output base; a wave
 * tile is three rounds of loads, so the look-back is paid once per 24576 rows) and
 * the DIRECT one of round 4 (values held in registers across the look-back, stored from there).  Synchronous
 * evaluations record the share of rows they selected; beyond what the window holds the next batches run on the
 * direct kernel, and come back when the share drops.  Returns 0 = windowed next, 1 = direct next, -1 = the plan has
 * one shape only (rows too wide for the window).
 * A launch whose look-back gives up (GDV_ERR_STALL: 5 s without progress) is re-run inside evaluate on the
 * filter + selection-mode projector chain (round 4: ExecutionError); under GDV_EVAL_ASYNC *num_selected_device
 * then receives -1 — evaluate the batch with the synchronous call. */
int gdv_filter_project_kernel_shape(const gdv_filter_project_t* fp);
/* "kernel": -1 follow the selectivity (default), 0 / 1 pin the windowed / the direct kernel (tests, measurements). */
int gdv_filter_project_set_tuning(gdv_filter_project_t* fp, const char* key, int64_t value);
void gdv_filter_project_free(gdv_filter_project_t* fp);

/* ---- One call, all the GPUs of a node (round 6) -------------------------
 * The reference's Projector::Evaluate(batch, ...) / Filter::Evaluate(batch, ...) are ONE call (PA:218-226, 246-248).
 * These entry points keep it one call on an N-GPU node: the logical batch of num_rows rows is cut into num_shards
 * row ranges by gdv_shard_bounds (1024-row bounds: no validity word or cache line straddles two shards), shard s is
 * evaluated on device shards[s].device by a host thread of its own — own device context, own stream; shard 0 on the
 * calling thread — and nothing is exchanged between the shards: the path has no collective.  The call returns when
 * every shard has finished; the first failing shard's status is returned (its device named in gdv_last_error()).
 *
 * Device-resident shards (gdv_*_evaluate_sharded): shards[s].cols / outs describe buffers in the HBM of
 * shards[s].device that hold ONLY the shard's rows [lo_s, hi_s) — row 0 of these buffers is row lo_s of the batch —
 * exactly what one-process-per-GPU ranks hold.  Projector outputs stay sharded (the logical result is their
 * concatenation in shard order; var-len outputs: per-shard offsets, rebased by the consumer that joins them).
 * Filter: shards[s].out_indices receives the shard's ascending positions, num_selected its count; with
 * GDV_SHARD_GLOBAL_INDICES the positions are lo_s + local (written so by the index-emission kernel, no extra pass),
 * and the shards' vectors concatenate into the globally ascending selection vector; *total_selected = the sum.
 * gdv_filter_gather_sharded lays them end to end on one device (hipMemcpyPeerAsync; the only inter-device traffic,
 * optional).
 *
 * Host-resident batches (gdv_*_evaluate_host_sharded): the caller passes ONE batch in host memory (what
 * gandiva::Projector::Evaluate receives from Arrow C++ callers); the library slices it (array offset + lo_s, output
 * pointers advanced to the shard's rows), stages every shard through its own device and writes the results into the
 * caller's ONE set of output buffers: N PCIe links instead of one.  Plans with var-len OUTPUTS are not sliced
 * (their byte positions depend on the shards before them): they run on devices[0] alone.  The filter's vector is
 * global and ascending; shards write into disjoint parts of out_indices and are closed up on the host. */
typedef struct {
  int32_t device;           /* gdv_set_device numbering (virtual devices included) */
  const gdv_column_t* cols; /* num_cols entries: the shard's rows, resident on `device` */
  gdv_out_column_t* outs;   /* projector: num_outs entries on `device`, sized for the shard's rows */
  void* out_indices;        /* filter: max_slots (>= the shard's rows) elements on `device` */
  int64_t max_slots;
  int64_t num_selected;     /* filter, out: rows this shard selected */
} gdv_shard_t;
#define GDV_SHARD_GLOBAL_INDICES 2u
int gdv_projector_evaluate_sharded(const gdv_projector_t* p, int64_t num_rows, int num_cols, int num_outs,
                                   gdv_shard_t* shards, int num_shards, uint32_t flags);
int gdv_filter_evaluate_sharded(const gdv_filter_t* f, int64_t num_rows, int num_cols, int selection_mode,
                                gdv_shard_t* shards, int num_shards, uint32_t flags, int64_t* total_selected);
/* dst_indices: dst_slots (>= the total) elements of the selection mode's width on device dst_device */
int gdv_filter_gather_sharded(const gdv_shard_t* shards, int num_shards, int selection_mode, int dst_device,
                              void* dst_indices, int64_t dst_slots);
int gdv_projector_evaluate_host_sharded(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                        gdv_out_column_t* outs, int num_outs, const int32_t* devices, int num_devices);
int gdv_filter_evaluate_host_sharded(const gdv_filter_t* f, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                     int selection_mode, void* out_indices, int64_t max_slots, int64_t* num_selected,
                                     const int32_t* devices, int num_devices);

/* ---- function registry ------------------------------------------------------------ */
int gdv_registry_size(void);
/* name: borrowed pointer valid for the process lifetime; params: up to max_params entries
 * are written, *num_params receives the real count. */
int gdv_registry_get(int index, const char** name, gdv_type_t* return_type, gdv_type_t* params,
                     int max_params, int* num_params);

/* ---- device helpers (for hosts without their own HIP binding, e.g. a JNI caller) --- */
/* Devices this library can be pointed at: the physical HIP devices, or more when virtual devices
 * were asked for (gdv_set_virtual_devices / GDV_VIRTUAL_DEVICES): device d then runs on physical
 * device d % gdv_physical_device_count() with a context of its own — N contexts on one GPU, for
 * testing the N-device code path on a single-GPU box. */
int gdv_device_count(void);
int gdv_physical_device_count(void);
int gdv_set_virtual_devices(int n);
/* Select the device of the CALLING THREAD (thread-local; also makes it the thread's current HIP
 * device).  gdv_get_device: the device calls from this thread run on. */
int gdv_set_device(int device);
int gdv_get_device(void);
/* Row range [*lo, *hi) of shard `shard` of `num_shards` over `num_rows` rows: near-equal shards on
 * 1024-row boundaries (no validity word or cache line straddles two shards), no exchange step —
 * Projector outputs concatenate in shard order, Filter shards yield local indices + *lo. */
int gdv_shard_bounds(int64_t num_rows, int num_shards, int shard, int64_t* lo, int64_t* hi);
int gdv_device_num_cus(void);
const char* gdv_device_arch(void);
int gdv_device_alloc(int64_t bytes, void** ptr);
int gdv_device_free(void* ptr);
int gdv_memcpy_h2d(void* dst_device, const void* src_host, int64_t bytes);
int gdv_memcpy_d2h(void* dst_host, const void* src_device, int64_t bytes);
int gdv_device_synchronize(void);
/* Round 4 — host memory the GPUs address directly.  GDV_MEM_HOST evaluations stage the caller's buffers
 * through a page-locked block (two memcpys + two DMA copies per call: 133 us for 16 K rows of the C2
 * shape).  Fixed-width columns, bitmaps and outputs that lie INSIDE a range registered here are instead
 * bound straight into the kernel — it reads and writes them over the fabric, nothing is copied — and
 * everything else is staged as before (per buffer; output bitmaps / values also need whole 8-byte
 * words of room, i.e. Arrow's own 64-byte padding).
 *   gdv_host_register(p, bytes): page-lock and map a range the CALLER owns (hipHostRegister) — the arena
 *     of an Arrow MemoryPool, JNI direct buffers.  Costs ~0.1 ms per MiB, once.  The caller unregisters
 *     before it frees the memory; the library never registers anything on its own (a registration that
 *     outlives a free() would leave the device a window onto somebody else's pages).
 *   gdv_host_alloc / gdv_host_free: page-locked memory from the library (hipHostMalloc), registered.
 * Process-wide, visible to every device.  The reference has no counterpart: its buffers are the CPU's. */
int gdv_host_register(void* ptr, int64_t bytes);
int gdv_host_unregister(void* ptr);
int gdv_host_alloc(int64_t bytes, void** ptr);
int gdv_host_free(void* ptr);
/* Bytes GDV_MEM_HOST evaluations have moved through staging blocks so far (process-wide, cumulative):
 * unchanged across a call = every buffer of that call was addressed in place. */
int64_t gdv_host_staged_bytes(void);
/* What plain streaming kernels reach on the calling thread's device right now: read-only, write-only
 * and copy rates in GB/s over two scratch buffers of `bytes` bytes (>= 1 GiB: the Infinity Cache holds
 * 256 MiB).  Boxes of one pool differ by 10-25 % on the same binary; a benchmark line that carries
 * these can be compared across boxes (bench.py: roofline.box). */
int gdv_device_hbm_ceilings(int64_t bytes, double* read_gbs, double* write_gbs, double* copy_gbs);
/* Round 4: the ceiling for a given TRAFFIC SHAPE — num_read input streams and num_write output streams of
 * 8-byte elements, bytes_per_stream each, moved by the projection kernel's own skeleton without its
 * arithmetic; the best rate over a sweep of grid sizes (2..32 workgroups per CU) x {plain, non-temporal}
 * accesses x {4, 16} sub-tiles per wave (round 6; *workgroups_per_cu is reported + 100 for the 16-sub-tile shape),
 * and where it was found.  Shapes: (1,0) (2,0) (4,0) (0,1) (0,4) (0,10) (2,1) (3,1) (4,10) (7,5)
 * (2,3).  A product kernel of that shape should not beat it: achieved / ceiling <= 1 on every box. */
int gdv_device_stream_ceiling(int64_t bytes_per_stream, int num_read, int num_write, double* gbs, int* workgroups_per_cu,
                              int* nontemporal);
/* The same sweep over the CALLER's buffers: streams[0 .. num_read) are read, the next num_write are
 * OVERWRITTEN, `elems` 8-byte elements each.  bench.py runs it on the very buffers the timed loop used
 * (after verifying them): same addresses, same sizes, same traffic shape — what is left between that
 * rate and the product kernel's is the kernel, not the box or where its memory happens to sit. */
int gdv_device_stream_ceiling_on(void* const* streams, int num_read, int num_write, int64_t elems, double* gbs,
                                 int* workgroups_per_cu, int* nontemporal);

/* ---- JNI-shaped flat entry points --------------------------------- */
/* What the reference's JNI layer receives from Java (JniWrapper.evaluateProjector /
 * evaluateFilter: long[] bufAddrs, long[] bufSizes, long[] outAddrs, long[] outSizes) and
 * would forward unchanged: every field's buffers flattened in schema order — validity,
 * then offsets (utf8/binary only), then data — as raw addresses and byte sizes; outputs the
 * same way, one group per expression.  Array offsets are 0 (Java's vectors have none).
 * sel_mode/sel_addr/sel_slots describe an optional selection vector (GDV_SEL_NONE: none).
 * A var-len output whose data capacity is too small fails with GDV_INVALID and
 * out_sizes[data slot] is updated to the bytes needed (the JNI expander callback's role). */
int gdv_projector_evaluate_flat(const gdv_projector_t* p, int64_t num_rows, const int64_t* buf_addrs,
                                const int64_t* buf_sizes, int num_bufs, int sel_mode,
                                int64_t sel_addr, int64_t sel_slots, const int64_t* out_addrs,
                                int64_t* out_sizes, int num_out_bufs, int mem_kind);
int gdv_filter_evaluate_flat(const gdv_filter_t* f, int64_t num_rows, const int64_t* buf_addrs,
                             const int64_t* buf_sizes, int num_bufs, int sel_mode, int64_t out_addr,
                             int64_t out_size_bytes, int64_t* num_selected, int mem_kind);

/* ---- build from protobuf bytes: the other half of the JNI boundary ---- */
/* What the reference's JNI buildProjector / buildFilter receive from Java: the Schema, and the
 * ExpressionList / Condition, serialised with protobuf (java: GandivaTypes from proto/Types.proto).
 * The message layout assumed is restated in gandiva_amd/csrc/gdv_proto.cc (from memory: the
 * reference mount holds no source — diff it against the real Types.proto); the wire format is
 * decoded by hand (no protoc / libprotobuf in this image).  selection_mode: gdv_selection_mode —
 * the proto's SelectionVectorType values SV_NONE / SV_INT16 / SV_INT32 are numerically the same.
 * The handles are the ordinary ones: evaluate with gdv_*_evaluate_flat, as the JNI layer would. */
int gdv_projector_make_from_proto(const void* schema_bytes, int64_t schema_len, const void* exprs_bytes,
                                  int64_t exprs_len, int selection_mode, const gdv_config_t* config,
                                  gdv_projector_t** out);
int gdv_filter_make_from_proto(const void* schema_bytes, int64_t schema_len, const void* condition_bytes,
                               int64_t condition_len, const gdv_config_t* config, gdv_filter_t** out);
/* the fused filter -> project operator (gdv_filter_project_*) from the same three messages */
int gdv_filter_project_make_from_proto(const void* schema_bytes, int64_t schema_len, const void* condition_bytes,
                                       int64_t condition_len, const void* exprs_bytes, int64_t exprs_len, int index_mode,
                                       const gdv_config_t* config, gdv_filter_project_t** out);
/* The decoded schema and trees rendered as text (NULL + gdv_last_error on malformed bytes). */
char* gdv_proto_describe(const void* schema_bytes, int64_t schema_len, const void* exprs_bytes, int64_t exprs_len,
                         int is_condition);

/* ---- Arrow C Device Data Interface (the step BEFORE the path: other ROCm producers) --- */
/* The ABI-stable structs of the Arrow C data / C device data interfaces
 * (pyarrow/include/arrow/c/abi.h).  Declared here under the spec's own include guards so
 * this header can be used with or without Arrow's. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif
#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11
struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event; /* hipEvent_t* for ROCM arrays, or NULL */
  int64_t reserved[3];
};
#endif
/* Evaluate over a record batch handed over as an ArrowDeviceArray: a struct array with one
 * child per schema field (what arrow::ExportDeviceRecordBatch / pyarrow's
 * RecordBatch._export_to_c_device produce).  device_type ARROW_DEVICE_ROCM is used in
 * place (zero-copy; outputs must be HBM buffers); ARROW_DEVICE_CPU / ROCM_HOST take the
 * staged host path.  A non-NULL sync_event is waited for on `stream` before the kernel.
 * The batch is borrowed: it is NOT released by these calls. */
int gdv_projector_evaluate_device_array(const gdv_projector_t* p,
                                        const struct ArrowDeviceArray* batch,
                                        const gdv_selection_t* sel, gdv_out_column_t* outs,
                                        int num_outs, void* stream, uint32_t flags);
int gdv_filter_evaluate_device_array(const gdv_filter_t* f, const struct ArrowDeviceArray* batch,
                                     int selection_mode, void* out_indices, int64_t max_slots,
                                     int64_t* num_selected, void* stream);

/* The step AFTER the path: evaluate and hand the results on as an ArrowDeviceArray (a struct
 * array, one child per expression; what arrow::ImportDeviceRecordBatch /
 * pyarrow.RecordBatch._import_from_c_device consume).  The library allocates the result
 * buffers: in HBM when `batch` is ARROW_DEVICE_ROCM (out->device_type ARROW_DEVICE_ROCM,
 * out->sync_event = hipEvent_t* recorded on `stream` after the last kernel), in 64-byte
 * aligned host memory (ARROW_DEVICE_CPU) otherwise.  `out_schema` (may be NULL) receives the
 * matching ArrowSchema (field names = the expressions' result fields).  Both are owned by
 * the consumer and freed through their release callbacks; children may be moved out and
 * released independently (buffers are reference-counted). */
int gdv_projector_evaluate_export(const gdv_projector_t* p, const struct ArrowDeviceArray* batch,
                                  const gdv_selection_t* sel, void* stream,
                                  struct ArrowDeviceArray* out, struct ArrowSchema* out_schema);

/* ---- device pool (round 6): placement-aware, retaining HBM for buffers that are streamed together ----------------
 * Where the driver puts a set of large buffers decides what a kernel over them runs at, and the placement stays with the
 * buffers: the same C2 kernel takes 4.88 .. 6.30 ms per Evaluate over ten fresh allocations of its ten output columns on
 * one box (profiles/r06_placement_probe.txt; rounds 4-5 saw 4.9 .. 7.0).  A caller that allocates its outputs once —
 * arrow::MemoryPool (pyarrow/include/arrow/memory_pool.h:120-124, the pool argument of Projector::Evaluate), a JNI caller's
 * allocator — gets whichever it gets.  The pool owns that choice:
 *   gdv_device_pool_reserve_set   `count` buffers of `bytes` each (the output columns of a projection over a batch of
 *       that size).  What is slow is a set whose members are NEIGHBOURS in the driver's allocation order (one region of
 *       the HBM address map; 5.1 .. 6.4 ms for C2 depending on the region, whatever the spacing inside it), members
 *       spread over a wide span of allocations are fast (4.81 .. 5.36, strided picks 4.91 .. 4.93 — profiles/
 *       r06_placement_{map,stagger,subsets}.txt).  So: four times the set is allocated as single buffers, candidate k =
 *       buffers k, k + 4, k + 8 ..., up to min(candidates, 4) candidates are timed with a non-temporal write sweep over
 *       the whole set (all buffers at the same offset at the same time, as the projection kernel writes them), the
 *       fastest is kept, everything else goes back to the driver.  Tens of milliseconds per GiB reserved, once.
 *       rates (may be NULL; capacity `candidates`): every candidate's sweep in GB/s; *tried, *kept (may be NULL).
 *       Sets of buffers below 64 MiB, and candidates = 1, are plain allocations.
 *   gdv_device_pool_free          the buffer goes back TO THE POOL and stays there: a later alloc / reserve_set of the
 *       same size gets it back — the placement found once serves every later batch of that shape.
 *   gdv_device_pool_alloc         one buffer: a retained one of exactly this size, else a fresh allocation.
 *   gdv_device_pool_trim / _destroy   retained buffers / everything go back to the driver.
 * A pool belongs to the device the calling thread had selected when it was created.  Thread-safe. */
typedef struct gdv_device_pool gdv_device_pool_t;
int gdv_device_pool_create(gdv_device_pool_t** out);
void gdv_device_pool_destroy(gdv_device_pool_t* pool);
int gdv_device_pool_reserve_set(gdv_device_pool_t* pool, int count, int64_t bytes, int candidates, void** ptrs,
                                double* rates, int* tried, int* kept);
int gdv_device_pool_alloc(gdv_device_pool_t* pool, int64_t bytes, void** ptr);
int gdv_device_pool_free(gdv_device_pool_t* pool, void* ptr);
int gdv_device_pool_trim(gdv_device_pool_t* pool);
int64_t gdv_device_pool_bytes(const gdv_device_pool_t* pool, int64_t* in_use);

/* ---- tier 0 (round 6) ---------------------------------------------------------------
 * Make of an unseen tree used to wait for hipRTC (0.25-0.9 s; the reference's LLVM JIT takes tens of milliseconds).
 * Plans inside the fixed-width core — add / subtract / multiply, the six comparisons, not / isnull / isnotnull, the numeric
 * casts, if / else, AND / OR, literals, over bool / integer / float / date / time columns, row mode — now come with a
 * post-fix PROGRAM for an ahead-of-time interpreter kernel: gdv_projector_make / gdv_filter_make queue the compilation on a
 * background thread and return (milliseconds); evaluations interpret the program until the specialised code object
 * has arrived and then switch to it.  Same argument block, same device functions, same flags: results are
 * bit-identical, the interpreted evaluation is slower.  Other plans wait for their compilation as before.
 * Environment, read once per process: GDV_NO_TIER0=1 (Make waits, as before round 6), GDV_FORCE_TIER0=1 (plans that
 * have a program always run on it: how the parity suite is held to tier 0).
 * gdv_tier0_program: the program of a projector's expressions (is_condition = 0) or a filter's condition (1) as text,
 * one instruction per line — or NULL, with the reason why the plan has no tier 0 in gdv_last_error().  No device needed.
 * gdv_tier0_launches: evaluations that ran on tier 0 so far (process-wide, cumulative). */
char* gdv_tier0_program(const gdv_schema_t* schema, gdv_expression_t* const* exprs, int num_exprs, int is_condition);
int64_t gdv_tier0_launches(void);
/* Stops the background compiler: queued compilations are dropped, the one in flight is waited for (<= ~1 s).  The library
 * does this itself when the process exits through its MAIN thread and that thread has called a Make (a thread_local guard,
 * which exit() destroys before it runs any exit handler, joins the compiler thread: the compiler's own statics — registered
 * while it compiles, hence torn down first — are then no longer in use); a process whose main thread never calls Make, a
 * JNI_OnUnload, a dlclose, or an embedder that tears the process down in an order of its own calls this before.  Later Makes
 * wait for their compilation as before round 6. */
void gdv_shutdown(void);

/* ---- build support ----------------------------------------------------------------- */
/* Plan + compile to a gfx950 code object without a device; fills the on-disk kernel cache. */
int gdv_precompile_projector(const gdv_schema_t* schema, gdv_expression_t* const* exprs,
                             int num_exprs, int selection_mode);
int gdv_precompile_filter(const gdv_schema_t* schema, gdv_expression_t* condition);
int gdv_precompile_filter_project(const gdv_schema_t* schema, gdv_expression_t* condition, gdv_expression_t* const* exprs,
                                  int num_exprs, int index_mode);
/* Pattern compilers of the planner, callable on their own (no device): a caller can check a regexp_like / regexp_matches
 * pattern, or a to_date pattern, before building an expression around it; tests drive the device functions' host build with
 * the tables.  gdv_compile_regex: `table` receives GDV_REGEX_TABLE_BYTES bytes (flags, nullable, predicates[8], first[8], last[8],
 * follow[64][8], match[256] as 64-bit words: gandiva_amd/csrc/gdv_regex.h).  gdv_compile_date_format: `ops` (capacity `cap`) receives one byte per
 * strptime directive ('L' c for a literal byte), *n their number.  0 or a status code; gdv_last_error() has the reason. */
#define GDV_REGEX_TABLE_BYTES 6352
int gdv_compile_regex(const char* pattern, int64_t pattern_len, uint8_t* table);
int gdv_compile_date_format(const char* pattern, int64_t pattern_len, uint8_t* ops, int64_t cap, int64_t* n);
/* Kernel identity (diagnostics, tests).  A fused kernel is named after a hash of its generated text
 * and of the device-library functions that text reaches — not of the whole library, so an edit of a
 * function a kernel never calls leaves its name (and every profile taken on it) alone.
 * gdv_kernel_library_tag: that library hash for `kernel_text` (what gdv_projector_dump_ir returns);
 * gdv_kernel_library_items: the names of the hashed items, one per line; library_source NULL = the
 * library embedded in this build, which gdv_device_library_source returns.  Strings are freed with
 * gdv_free_string (the library source is a static string: do not free it). */
char* gdv_kernel_library_tag(const char* library_source, const char* kernel_text);
char* gdv_kernel_library_items(const char* library_source, const char* kernel_text);
const char* gdv_device_library_source(void);

#ifdef __cplusplus
}
#endif
#endif /* GANDIVA_AMD_H_ */
