// extern "C" boundary (include/gandiva_amd.h) over the C++ core.
#include "../../include/gandiva_amd.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gdv_engine.h"
#include "gdv_kernels.h"
#include "gdv_libtag.h"
#include "gdv_pool.h"
#include "gdv_proto.h"
#include "gdv_regex.h"

using namespace gdv;

struct gdv_schema { Schema fields; };
struct gdv_node { NodePtr node; };
struct gdv_expression { ExpressionPtr expr; };
struct gdv_projector {
  std::shared_ptr<Projector> p;
  std::vector<std::string> output_names;  // result field names, for the C data export
};
struct gdv_filter { std::shared_ptr<Filter> f; };
struct gdv_filter_project { std::shared_ptr<FilterProject> fp; };
struct gdv_device_pool { DevicePool pool; };

namespace {

thread_local std::string g_last_error;

int Fail(const Status& s) {
  g_last_error = s.ToString();
  return static_cast<int>(s.code);
}
int Check(const Status& s) {
  if (s.ok()) return GDV_OK;
  return Fail(s);
}
template <typename T>
T* FailPtr(const std::string& msg) {
  g_last_error = "Invalid: " + msg;
  return nullptr;
}

bool ToType(gdv_type_t t, DataType* out) {
  switch (t.id) {
    case kBool: case kUInt8: case kInt8: case kUInt16: case kInt16: case kUInt32: case kInt32:
    case kUInt64: case kInt64: case kFloat: case kDouble: case kString: case kBinary:
    case kDate32: case kDate64: case kTimestamp: case kTime32: case kTime64: case kDecimal128:
      *out = DataType(static_cast<TypeId>(t.id), t.precision, t.scale);
      return true;
    default:
      return false;
  }
}
gdv_type_t FromType(const DataType& t) { return gdv_type_t{t.id, t.precision, t.scale}; }

char* DupString(const std::string& s) {
  char* p = static_cast<char*>(malloc(s.size() + 1));
  if (p) std::memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

bool CollectChildren(gdv_node_t* const* children, int n, NodeVector* out) {
  if (n < 0 || (n > 0 && children == nullptr)) return false;
  for (int i = 0; i < n; i++) {
    if (children[i] == nullptr || !children[i]->node) return false;
    out->push_back(children[i]->node);
  }
  return true;
}

std::vector<ColumnBuffers> ToColumns(const gdv_column_t* cols, int n) {
  std::vector<ColumnBuffers> v(n > 0 ? n : 0);
  for (int i = 0; i < n; i++) {
    v[i].validity = cols[i].validity;
    v[i].validity_size = cols[i].validity_size;
    v[i].data = cols[i].data;
    v[i].data_size = cols[i].data_size;
    v[i].offsets = cols[i].offsets;
    v[i].offsets_size = cols[i].offsets_size;
    v[i].offset = cols[i].offset;
  }
  return v;
}

// No C++ exception may cross the C boundary (std::bad_alloc from a vector, a std::string
// length_error …): entry points that allocate run through this guard.
template <typename F>
int Guarded(F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return Fail(Status::OutOfMemory("host allocation failed"));
  } catch (const std::exception& e) {
    return Fail(Status::ExecutionError(std::string("internal error: ") + e.what()));
  } catch (...) {
    return Fail(Status::ExecutionError("internal error: unknown exception"));
  }
}

template <typename F>
auto GuardedPtr(F&& body) -> decltype(body()) {
  try {
    return body();
  } catch (const std::exception& e) {
    g_last_error = std::string("ExecutionError: internal error: ") + e.what();
  } catch (...) {
    g_last_error = "ExecutionError: internal error: unknown exception";
  }
  return nullptr;
}

bool ToSelectionMode(int m, SelectionMode* out) {
  if (m < 0 || m > 3) return false;
  *out = static_cast<SelectionMode>(m);
  return true;
}

}  // namespace

extern "C" {

const char* gdv_last_error(void) { return g_last_error.c_str(); }
const char* gdv_version(void) { return "gandiva_amd 0.1.0 (gfx950)"; }
void gdv_free_string(char* s) { free(s); }

// ---------------------------------------------------------------- schema
gdv_schema_t* gdv_schema_new(void) { return new gdv_schema(); }
int gdv_schema_add_field(gdv_schema_t* schema, const char* name, gdv_type_t type, int nullable) {
  return Guarded([&]() -> int {
  DataType t;
  if (!schema || !name) return Fail(Status::Invalid("null schema or field name"));
  if (!ToType(type, &t)) return Fail(Status::Invalid("unsupported type id " + std::to_string(type.id)));
  schema->fields.push_back(Field{name, t, nullable != 0});
  return GDV_OK;
  });
}
int gdv_schema_num_fields(const gdv_schema_t* schema) {
  return schema ? static_cast<int>(schema->fields.size()) : 0;
}
void gdv_schema_free(gdv_schema_t* schema) { delete schema; }

// ---------------------------------------------------------------- nodes
gdv_node_t* gdv_node_field(const char* name, gdv_type_t type) {
  return GuardedPtr([&]() -> gdv_node_t* {
  DataType t;
  if (!name) return FailPtr<gdv_node_t>("field name is null");
  if (!ToType(type, &t)) return FailPtr<gdv_node_t>("unsupported type id");
  return new gdv_node{std::make_shared<FieldNode>(Field{name, t, true})};
  });
}

gdv_node_t* gdv_node_literal(gdv_type_t type, const void* value, int is_null) {
  return GuardedPtr([&]() -> gdv_node_t* {
  DataType t;
  if (!ToType(type, &t)) return FailPtr<gdv_node_t>("unsupported type id");
  if (t.is_varlen()) return FailPtr<gdv_node_t>("use gdv_node_literal_bytes for var-len types");
  Literal lit;
  lit.is_null = is_null != 0;
  if (!lit.is_null) {
    if (!value) return FailPtr<gdv_node_t>("literal value is null");
    int w = t.id == kBool ? 1 : t.byte_width();
    unsigned char raw[16] = {0};
    std::memcpy(raw, value, w);
    std::memcpy(&lit.lo, raw, 8);
    std::memcpy(&lit.hi, raw + 8, 8);
    if (t.id == kBool) lit.lo = raw[0] ? 1 : 0;
    // sign-extend narrow signed integers so the payload is the value's int64 image
    if (t.id == kInt8) lit.lo = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int8_t>(raw[0])));
    if (t.id == kInt16) { int16_t v; std::memcpy(&v, raw, 2); lit.lo = static_cast<uint64_t>(static_cast<int64_t>(v)); }
    if (t.id == kInt32 || t.id == kDate32 || t.id == kTime32) {
      int32_t v; std::memcpy(&v, raw, 4); lit.lo = static_cast<uint64_t>(static_cast<int64_t>(v));
    }
  }
  return new gdv_node{std::make_shared<LiteralNode>(t, lit)};
  });
}

gdv_node_t* gdv_node_literal_bytes(gdv_type_t type, const char* data, int64_t len, int is_null) {
  return GuardedPtr([&]() -> gdv_node_t* {
  DataType t;
  if (!ToType(type, &t) || !t.is_varlen()) return FailPtr<gdv_node_t>("type must be string or binary");
  Literal lit;
  lit.is_null = is_null != 0;
  if (!lit.is_null) {
    if (len < 0 || (len > 0 && !data)) return FailPtr<gdv_node_t>("bad literal bytes");
    lit.bytes.assign(data ? data : "", static_cast<size_t>(len));
  }
  return new gdv_node{std::make_shared<LiteralNode>(t, lit)};
  });
}

gdv_node_t* gdv_node_function(const char* name, gdv_node_t* const* children, int num_children,
                              gdv_type_t return_type) {
  return GuardedPtr([&]() -> gdv_node_t* {
  DataType t;
  NodeVector kids;
  if (!name) return FailPtr<gdv_node_t>("function name is null");
  if (!ToType(return_type, &t)) return FailPtr<gdv_node_t>("unsupported return type id");
  if (!CollectChildren(children, num_children, &kids)) return FailPtr<gdv_node_t>("null child node");
  return new gdv_node{MakeFunctionNode(name, std::move(kids), t)};
  });
}

gdv_node_t* gdv_node_if(gdv_node_t* c, gdv_node_t* t, gdv_node_t* e, gdv_type_t return_type) {
  return GuardedPtr([&]() -> gdv_node_t* {
  DataType rt;
  if (!c || !t || !e || !c->node || !t->node || !e->node) return FailPtr<gdv_node_t>("null child node");
  if (!ToType(return_type, &rt)) return FailPtr<gdv_node_t>("unsupported return type id");
  return new gdv_node{std::make_shared<IfNode>(c->node, t->node, e->node, rt)};
  });
}

gdv_node_t* gdv_node_and(gdv_node_t* const* children, int n) {
  return GuardedPtr([&]() -> gdv_node_t* {
  NodeVector kids;
  if (!CollectChildren(children, n, &kids)) return FailPtr<gdv_node_t>("null child node");
  return new gdv_node{std::make_shared<BooleanNode>(BooleanNode::kAnd, std::move(kids))};
  });
}

gdv_node_t* gdv_node_or(gdv_node_t* const* children, int n) {
  return GuardedPtr([&]() -> gdv_node_t* {
  NodeVector kids;
  if (!CollectChildren(children, n, &kids)) return FailPtr<gdv_node_t>("null child node");
  return new gdv_node{std::make_shared<BooleanNode>(BooleanNode::kOr, std::move(kids))};
  });
}

gdv_node_t* gdv_node_in(gdv_node_t* node, gdv_type_t value_type, const void* values, int n) {
  return GuardedPtr([&]() -> gdv_node_t* {
  DataType t;
  if (!node || !node->node) return FailPtr<gdv_node_t>("null child node");
  if (!ToType(value_type, &t) || t.is_varlen()) return FailPtr<gdv_node_t>("bad IN value type");
  if (n < 0 || (n > 0 && !values)) return FailPtr<gdv_node_t>("bad IN values");
  const int w = t.byte_width();
  if (w == 0) return FailPtr<gdv_node_t>("IN over this type is not supported");
  std::vector<Literal> lits(n);
  const char* p = static_cast<const char*>(values);
  for (int i = 0; i < n; i++) {
    unsigned char raw[16] = {0};
    std::memcpy(raw, p + static_cast<size_t>(i) * w, w);
    std::memcpy(&lits[i].lo, raw, 8);
    std::memcpy(&lits[i].hi, raw + 8, 8);
  }
  return new gdv_node{std::make_shared<InNode>(node->node, t, std::move(lits))};
  });
}

gdv_node_t* gdv_node_in_bytes(gdv_node_t* node, gdv_type_t value_type, const char* const* values,
                              const int64_t* lengths, int n) {
  return GuardedPtr([&]() -> gdv_node_t* {
  DataType t;
  if (!node || !node->node) return FailPtr<gdv_node_t>("null child node");
  if (!ToType(value_type, &t) || !t.is_varlen()) return FailPtr<gdv_node_t>("bad IN value type");
  if (n < 0 || (n > 0 && (!values || !lengths))) return FailPtr<gdv_node_t>("bad IN values");
  std::vector<Literal> lits(n);
  for (int i = 0; i < n; i++) lits[i].bytes.assign(values[i] ? values[i] : "", static_cast<size_t>(lengths[i]));
  return new gdv_node{std::make_shared<InNode>(node->node, t, std::move(lits))};
  });
}

char* gdv_node_to_string(const gdv_node_t* node) {
  return GuardedPtr([&]() -> char* {
  return node && node->node ? DupString(node->node->ToString()) : nullptr;
  });
}
gdv_type_t gdv_node_return_type(const gdv_node_t* node) {
  return node && node->node ? FromType(node->node->return_type()) : gdv_type_t{0, 0, 0};
}
void gdv_node_free(gdv_node_t* node) { delete node; }

gdv_expression_t* gdv_expression_new(gdv_node_t* root, const char* result_name, gdv_type_t rt) {
  return GuardedPtr([&]() -> gdv_expression_t* {
  DataType t;
  if (!root || !root->node) return FailPtr<gdv_expression_t>("root node is null");
  if (!result_name) return FailPtr<gdv_expression_t>("result field is null");
  if (!ToType(rt, &t)) return FailPtr<gdv_expression_t>("unsupported result type id");
  return new gdv_expression{std::make_shared<Expression>(root->node, Field{result_name, t, true})};
  });
}
gdv_expression_t* gdv_condition_new(gdv_node_t* root) {
  return GuardedPtr([&]() -> gdv_expression_t* {
  if (!root || !root->node) return FailPtr<gdv_expression_t>("root node is null");
  return new gdv_expression{std::make_shared<Expression>(root->node, Field{"cond", boolean(), true})};
  });
}
char* gdv_expression_to_string(const gdv_expression_t* e) {
  return GuardedPtr([&]() -> char* {
  return e && e->expr ? DupString(e->expr->ToString()) : nullptr;
  });
}
gdv_type_t gdv_expression_result_type(const gdv_expression_t* e) {
  return e && e->expr ? FromType(e->expr->result().type) : gdv_type_t{0, 0, 0};
}
void gdv_expression_free(gdv_expression_t* e) { delete e; }

// ---------------------------------------------------------------- projector
static bool CollectExprs(gdv_expression_t* const* exprs, int n, std::vector<ExpressionPtr>* out) {
  if (n < 0 || (n > 0 && !exprs)) return false;
  for (int i = 0; i < n; i++) {
    if (!exprs[i] || !exprs[i]->expr) return false;
    out->push_back(exprs[i]->expr);
  }
  return true;
}

int gdv_projector_make(const gdv_schema_t* schema, gdv_expression_t* const* exprs, int num_exprs,
                       int selection_mode, const gdv_config_t* config, gdv_projector_t** out) {
  return Guarded([&]() -> int {
  if (!schema || !out) return Fail(Status::Invalid("null schema or output pointer"));
  std::vector<ExpressionPtr> ex;
  if (!CollectExprs(exprs, num_exprs, &ex)) return Fail(Status::Invalid("null expression"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  Configuration cfg;
  if (config) { cfg.optimize = config->optimize != 0; cfg.dump_ir = config->dump_ir != 0; }
  std::shared_ptr<Projector> p;
  Status s = Projector::Make(schema->fields, ex, mode, cfg, &p);
  if (!s.ok()) return Fail(s);
  std::vector<std::string> names;
  for (auto& e : ex) names.push_back(e->result().name);
  *out = new gdv_projector{p, std::move(names)};
  return GDV_OK;
  });
}
int gdv_projector_num_outputs(const gdv_projector_t* p) { return p ? p->p->num_outputs() : 0; }
int gdv_projector_path_hint(const gdv_projector_t* p) { return p ? p->p->path_hint() : -1; }
gdv_type_t gdv_projector_output_type(const gdv_projector_t* p, int i) {
  if (!p || i < 0 || i >= p->p->num_outputs()) return gdv_type_t{0, 0, 0};
  return FromType(p->p->output_type(i));
}
int gdv_projector_output_sizes(const gdv_projector_t* p, int i, int64_t rows, int mem_kind,
                               int64_t* validity_bytes, int64_t* data_bytes) {
  if (!p || i < 0 || i >= p->p->num_outputs() || rows < 0) return Fail(Status::Invalid("bad argument"));
  const DataType& t = p->p->output_type(i);
  const bool dev = mem_kind == GDV_MEM_DEVICE;
  if (validity_bytes) *validity_bytes = dev ? Projector::ValidityBytes(rows) : (rows + 7) / 8;
  if (data_bytes && t.is_varlen()) {
    *data_bytes = p->p->VarlenBytesHint(i, rows);  // 0 until a batch has been evaluated
    return GDV_OK;
  }
  if (data_bytes)
    *data_bytes = t.id == kBool ? (dev ? Projector::ValidityBytes(rows) : (rows + 7) / 8)
                                : Projector::DataBytes(t, rows);
  return GDV_OK;
}
namespace {
int ProjectorEvaluate(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols,
                      int num_cols, const gdv_selection_t* sel, const void* num_slots_device,
                      gdv_out_column_t* outs, int num_outs, int mem_kind, void* stream, uint32_t flags) {
  return Guarded([&]() -> int {
  if (!p) return Fail(Status::Invalid("null projector"));
  if (num_cols > 0 && !cols) return Fail(Status::Invalid("null column array"));
  if (!outs) return Fail(Status::Invalid("Output array vector cannot be null"));
  std::vector<ColumnBuffers> c = ToColumns(cols, num_cols);
  std::vector<OutputBuffers> o(num_outs > 0 ? num_outs : 0);
  for (int i = 0; i < num_outs; i++) {
    o[i].validity = outs[i].validity;
    o[i].validity_size = outs[i].validity_size;
    o[i].data = outs[i].data;
    o[i].data_size = outs[i].data_size;
    o[i].offsets = outs[i].offsets;
    o[i].offsets_size = outs[i].offsets_size;
  }
  SelectionView sv;
  if (sel) {
    if (!ToSelectionMode(sel->mode, &sv.mode)) return Fail(Status::Invalid("bad selection mode"));
    sv.indices = sel->indices;
    sv.num_slots = sel->num_slots;
    sv.num_slots_device = num_slots_device;
  }
  Status st = p->p->Evaluate(num_rows, c.data(), num_cols, sel ? &sv : nullptr, o.data(), num_outs,
                             mem_kind == GDV_MEM_DEVICE ? MemKind::kDevice : MemKind::kHost,
                             static_cast<hipStream_t>(stream), flags);
  for (int i = 0; i < num_outs; i++) outs[i].data_size = o[i].data_size;  // var-len: bytes produced / needed
  return Check(st);
  });
}
}  // namespace

int gdv_projector_evaluate(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols,
                           int num_cols, const gdv_selection_t* sel, gdv_out_column_t* outs,
                           int num_outs, int mem_kind, void* stream, uint32_t flags) {
  return ProjectorEvaluate(p, num_rows, cols, num_cols, sel, nullptr, outs, num_outs, mem_kind, stream, flags);
}
int gdv_projector_evaluate_selected(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols,
                                    int num_cols, const gdv_selection_t* sel, const void* num_slots_device,
                                    gdv_out_column_t* outs, int num_outs, void* stream, uint32_t flags) {
  if (!sel || !num_slots_device) return Fail(Status::Invalid("selection vector and device slot count are required"));
  return ProjectorEvaluate(p, num_rows, cols, num_cols, sel, num_slots_device, outs, num_outs, GDV_MEM_DEVICE, stream,
                           flags);
}
int gdv_projector_evaluate_async(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                 const gdv_selection_t* sel, const void* num_slots_device, gdv_out_column_t* outs,
                                 int num_outs, void* stream, void* result) {
  return Guarded([&]() -> int {
  if (!p) return Fail(Status::Invalid("null projector"));
  if (num_cols > 0 && !cols) return Fail(Status::Invalid("null column array"));
  if (!outs || !result) return Fail(Status::Invalid("Output array vector and result block cannot be null"));
  std::vector<ColumnBuffers> c = ToColumns(cols, num_cols);
  std::vector<OutputBuffers> o(num_outs > 0 ? num_outs : 0);
  for (int i = 0; i < num_outs; i++) {
    o[i].validity = outs[i].validity; o[i].validity_size = outs[i].validity_size;
    o[i].data = outs[i].data; o[i].data_size = outs[i].data_size;
    o[i].offsets = outs[i].offsets; o[i].offsets_size = outs[i].offsets_size;
  }
  SelectionView sv;
  if (sel) {
    if (!ToSelectionMode(sel->mode, &sv.mode)) return Fail(Status::Invalid("bad selection mode"));
    sv.indices = sel->indices;
    sv.num_slots = sel->num_slots;
    sv.num_slots_device = num_slots_device;
  }
  return Check(p->p->EvaluateAsync(num_rows, c.data(), num_cols, sel ? &sv : nullptr, o.data(), num_outs,
                                   static_cast<hipStream_t>(stream), result));
  });
}
int gdv_projector_evaluate_many(const gdv_projector_t* p, const gdv_batch_t* batches, int num_batches, void* stream,
                                uint32_t flags) {
  return Guarded([&]() -> int {
  if (!p) return Fail(Status::Invalid("null projector"));
  if (num_batches < 0 || (num_batches > 0 && !batches)) return Fail(Status::Invalid("null batch list"));
  std::vector<std::vector<ColumnBuffers>> cols(num_batches);
  std::vector<std::vector<OutputBuffers>> outs(num_batches);
  std::vector<Projector::BatchView> views(num_batches);
  for (int b = 0; b < num_batches; b++) {
    const gdv_batch_t& g = batches[b];
    if ((g.num_cols > 0 && !g.cols) || (g.num_outs > 0 && !g.outs)) return Fail(Status::Invalid("null column array"));
    cols[b] = ToColumns(g.cols, g.num_cols);
    outs[b].resize(g.num_outs > 0 ? g.num_outs : 0);
    for (int i = 0; i < g.num_outs; i++) {
      outs[b][i].validity = g.outs[i].validity;
      outs[b][i].validity_size = g.outs[i].validity_size;
      outs[b][i].data = g.outs[i].data;
      outs[b][i].data_size = g.outs[i].data_size;
      outs[b][i].offsets = g.outs[i].offsets;
      outs[b][i].offsets_size = g.outs[i].offsets_size;
    }
    views[b].num_rows = g.num_rows;
    views[b].cols = cols[b].data();
    views[b].num_cols = g.num_cols;
    views[b].outs = outs[b].data();
    views[b].num_outs = g.num_outs;
  }
  Status st = p->p->EvaluateMany(views.data(), num_batches, static_cast<hipStream_t>(stream), flags);
  for (int b = 0; b < num_batches; b++)
    for (int i = 0; i < batches[b].num_outs; i++) batches[b].outs[i].data_size = outs[b][i].data_size;
  return Check(st);
  });
}
char* gdv_projector_dump_ir(const gdv_projector_t* p) { return p ? DupString(p->p->DumpIR()) : nullptr; }
void gdv_projector_free(gdv_projector_t* p) { delete p; }

// ---------------------------------------------------------------- filter
int gdv_filter_make(const gdv_schema_t* schema, gdv_expression_t* condition,
                    const gdv_config_t* config, gdv_filter_t** out) {
  return Guarded([&]() -> int {
  if (!schema || !out) return Fail(Status::Invalid("null schema or output pointer"));
  if (!condition || !condition->expr) return Fail(Status::Invalid("Condition cannot be null"));
  Configuration cfg;
  if (config) { cfg.optimize = config->optimize != 0; cfg.dump_ir = config->dump_ir != 0; }
  std::shared_ptr<Filter> f;
  Status s = Filter::Make(schema->fields, condition->expr, cfg, &f);
  if (!s.ok()) return Fail(s);
  *out = new gdv_filter{f};
  return GDV_OK;
  });
}
int gdv_filter_evaluate(const gdv_filter_t* f, int64_t num_rows, const gdv_column_t* cols,
                        int num_cols, int selection_mode, void* out_indices, int64_t max_slots,
                        int64_t* num_selected, int mem_kind, void* stream) {
  return Guarded([&]() -> int {
  if (!f) return Fail(Status::Invalid("null filter"));
  if (num_cols > 0 && !cols) return Fail(Status::Invalid("null column array"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  std::vector<ColumnBuffers> c = ToColumns(cols, num_cols);
  return Check(f->f->Evaluate(num_rows, c.data(), num_cols, mode, out_indices, max_slots,
                              num_selected, mem_kind == GDV_MEM_DEVICE ? MemKind::kDevice : MemKind::kHost,
                              static_cast<hipStream_t>(stream)));
  });
}
int gdv_filter_evaluate_async(const gdv_filter_t* f, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                              int selection_mode, void* out_indices, int64_t max_slots, void* num_selected_device,
                              void* stream) {
  return Guarded([&]() -> int {
  if (!f) return Fail(Status::Invalid("null filter"));
  if (num_cols > 0 && !cols) return Fail(Status::Invalid("null column array"));
  if (!num_selected_device) return Fail(Status::Invalid("null count pointer"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  std::vector<ColumnBuffers> c = ToColumns(cols, num_cols);
  int64_t unused = 0;
  return Check(f->f->Evaluate(num_rows, c.data(), num_cols, mode, out_indices, max_slots, &unused, MemKind::kDevice,
                              static_cast<hipStream_t>(stream), kEvalAsync, num_selected_device));
  });
}
int gdv_filter_evaluate_many(const gdv_filter_t* f, const gdv_filter_batch_t* batches, int num_batches,
                             int selection_mode, int64_t* num_selected, void* num_selected_device, void* stream,
                             uint32_t flags) {
  return Guarded([&]() -> int {
  if (!f) return Fail(Status::Invalid("null filter"));
  if (num_batches < 0 || (num_batches > 0 && !batches)) return Fail(Status::Invalid("null batch list"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  std::vector<std::vector<ColumnBuffers>> cols(num_batches);
  std::vector<Filter::BatchView> views(num_batches);
  for (int b = 0; b < num_batches; b++) {
    if (batches[b].num_cols > 0 && !batches[b].cols) return Fail(Status::Invalid("null column array"));
    cols[b] = ToColumns(batches[b].cols, batches[b].num_cols);
    views[b].num_rows = batches[b].num_rows;
    views[b].cols = cols[b].data();
    views[b].num_cols = batches[b].num_cols;
    views[b].out_indices = batches[b].out_indices;
    views[b].max_slots = batches[b].max_slots;
  }
  return Check(f->f->EvaluateMany(views.data(), num_batches, mode, num_selected, num_selected_device,
                                  static_cast<hipStream_t>(stream), flags));
  });
}

// ---------------------------------------------------------------- build from protobuf bytes (JNI)
int gdv_projector_make_from_proto(const void* schema_bytes, int64_t schema_len, const void* exprs_bytes,
                                  int64_t exprs_len, int selection_mode, const gdv_config_t* config,
                                  gdv_projector_t** out) {
  return Guarded([&]() -> int {
  if (!out || schema_len < 0 || exprs_len < 0) return Fail(Status::Invalid("bad argument"));
  Schema schema;
  std::vector<ExpressionPtr> ex;
  Status s = DecodeSchema(static_cast<const uint8_t*>(schema_bytes), static_cast<size_t>(schema_len), &schema);
  if (s.ok()) s = DecodeExpressionList(static_cast<const uint8_t*>(exprs_bytes), static_cast<size_t>(exprs_len), &ex);
  if (!s.ok()) return Fail(s);
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  Configuration cfg;
  if (config) { cfg.optimize = config->optimize != 0; cfg.dump_ir = config->dump_ir != 0; }
  std::shared_ptr<Projector> p;
  s = Projector::Make(schema, ex, mode, cfg, &p);
  if (!s.ok()) return Fail(s);
  std::vector<std::string> names;
  for (auto& e : ex) names.push_back(e->result().name);
  *out = new gdv_projector{p, std::move(names)};
  return GDV_OK;
  });
}
int gdv_filter_make_from_proto(const void* schema_bytes, int64_t schema_len, const void* condition_bytes,
                               int64_t condition_len, const gdv_config_t* config, gdv_filter_t** out) {
  return Guarded([&]() -> int {
  if (!out || schema_len < 0 || condition_len < 0) return Fail(Status::Invalid("bad argument"));
  Schema schema;
  ExpressionPtr cond;
  Status s = DecodeSchema(static_cast<const uint8_t*>(schema_bytes), static_cast<size_t>(schema_len), &schema);
  if (s.ok()) s = DecodeCondition(static_cast<const uint8_t*>(condition_bytes), static_cast<size_t>(condition_len), &cond);
  if (!s.ok()) return Fail(s);
  Configuration cfg;
  if (config) { cfg.optimize = config->optimize != 0; cfg.dump_ir = config->dump_ir != 0; }
  std::shared_ptr<Filter> f;
  s = Filter::Make(schema, cond, cfg, &f);
  if (!s.ok()) return Fail(s);
  *out = new gdv_filter{f};
  return GDV_OK;
  });
}
int gdv_filter_project_make_from_proto(const void* schema_bytes, int64_t schema_len, const void* condition_bytes,
                                       int64_t condition_len, const void* exprs_bytes, int64_t exprs_len, int index_mode,
                                       const gdv_config_t* config, gdv_filter_project_t** out) {
  return Guarded([&]() -> int {
  if (!out || schema_len < 0 || condition_len < 0 || exprs_len < 0) return Fail(Status::Invalid("bad argument"));
  SelectionMode mode;
  if (!ToSelectionMode(index_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  Schema schema;
  ExpressionPtr cond;
  std::vector<ExpressionPtr> exprs;
  Status s = DecodeSchema(static_cast<const uint8_t*>(schema_bytes), static_cast<size_t>(schema_len), &schema);
  if (s.ok()) s = DecodeCondition(static_cast<const uint8_t*>(condition_bytes), static_cast<size_t>(condition_len), &cond);
  if (s.ok()) s = DecodeExpressionList(static_cast<const uint8_t*>(exprs_bytes), static_cast<size_t>(exprs_len), &exprs);
  if (!s.ok()) return Fail(s);
  Configuration cfg;
  if (config) { cfg.optimize = config->optimize != 0; cfg.dump_ir = config->dump_ir != 0; }
  std::shared_ptr<FilterProject> fp;
  s = FilterProject::Make(schema, cond, exprs, mode, cfg, &fp);
  if (!s.ok()) return Fail(s);
  *out = new gdv_filter_project{fp};
  return GDV_OK;
  });
}
// the decoded trees, rendered (what a test — or a maintainer diffing against the Java side — reads)
char* gdv_proto_describe(const void* schema_bytes, int64_t schema_len, const void* exprs_bytes, int64_t exprs_len,
                         int is_condition) {
  return GuardedPtr([&]() -> char* {
  if (schema_len < 0 || exprs_len < 0 || (schema_len > 0 && schema_bytes == nullptr) ||
      (exprs_len > 0 && exprs_bytes == nullptr)) {
    Fail(Status::Invalid("gdv_proto_describe: negative length or null message"));
    return nullptr;
  }
  Schema schema;
  Status s = DecodeSchema(static_cast<const uint8_t*>(schema_bytes), static_cast<size_t>(schema_len), &schema);
  std::string text;
  if (s.ok()) {
    for (auto& f : schema) text += "field " + f.name + ": " + f.type.ToString() + (f.nullable ? "" : " not null") + "\n";
    if (is_condition) {
      ExpressionPtr cond;
      s = DecodeCondition(static_cast<const uint8_t*>(exprs_bytes), static_cast<size_t>(exprs_len), &cond);
      if (s.ok()) text += "condition " + cond->ToString() + "\n";
    } else {
      std::vector<ExpressionPtr> ex;
      s = DecodeExpressionList(static_cast<const uint8_t*>(exprs_bytes), static_cast<size_t>(exprs_len), &ex);
      if (s.ok())
        for (auto& e : ex) text += "expr " + e->result().name + ": " + e->result().type.ToString() + " = " + e->ToString() + "\n";
    }
  }
  if (!s.ok()) { Fail(s); return nullptr; }
  return DupString(text);
  });
}
char* gdv_filter_dump_ir(const gdv_filter_t* f) { return f ? DupString(f->f->DumpIR()) : nullptr; }
void gdv_filter_free(gdv_filter_t* f) { delete f; }
int gdv_filter_set_tuning(gdv_filter_t* f, const char* key, int64_t value) {
  return Guarded([&]() -> int {
    if (f == nullptr || key == nullptr) return Fail(Status::Invalid("gdv_filter_set_tuning: null argument"));
    Status st = f->f->SetTuning(key, value);
    return st.ok() ? GDV_OK : Fail(st);
  });
}

// ---------------------------------------------------------------- fused filter -> project
int gdv_filter_project_make(const gdv_schema_t* schema, gdv_expression_t* condition, gdv_expression_t* const* exprs,
                            int num_exprs, int index_mode, const gdv_config_t* config, gdv_filter_project_t** out) {
  return Guarded([&]() -> int {
  if (!schema || !out) return Fail(Status::Invalid("null schema or output pointer"));
  if (!condition || !condition->expr) return Fail(Status::Invalid("Condition cannot be null"));
  if (num_exprs <= 0 || !exprs) return Fail(Status::Invalid("Expressions cannot be empty"));
  SelectionMode mode;
  if (!ToSelectionMode(index_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  std::vector<ExpressionPtr> ex;
  for (int i = 0; i < num_exprs; i++) {
    if (!exprs[i] || !exprs[i]->expr) return Fail(Status::Invalid("Expression cannot be null"));
    ex.push_back(exprs[i]->expr);
  }
  Configuration cfg;
  if (config) { cfg.optimize = config->optimize != 0; cfg.dump_ir = config->dump_ir != 0; }
  std::shared_ptr<FilterProject> fp;
  Status s = FilterProject::Make(schema->fields, condition->expr, ex, mode, cfg, &fp);
  if (!s.ok()) return Fail(s);
  *out = new gdv_filter_project{fp};
  return GDV_OK;
  });
}
int gdv_filter_project_num_outputs(const gdv_filter_project_t* fp) { return fp ? fp->fp->num_outputs() : 0; }
gdv_type_t gdv_filter_project_output_type(const gdv_filter_project_t* fp, int i) {
  if (!fp || i < 0 || i >= fp->fp->num_outputs()) return gdv_type_t{0, 0, 0};
  return FromType(fp->fp->output_type(i));
}
int gdv_filter_project_evaluate(const gdv_filter_project_t* fp, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                gdv_out_column_t* outs, int num_outs, void* out_indices, int64_t max_slots,
                                int64_t* num_selected, void* num_selected_device, int mem_kind, void* stream,
                                uint32_t flags) {
  return Guarded([&]() -> int {
  if (!fp) return Fail(Status::Invalid("null filter-project"));
  if ((num_cols > 0 && !cols) || (num_outs > 0 && !outs)) return Fail(Status::Invalid("null column array"));
  std::vector<ColumnBuffers> c = ToColumns(cols, num_cols);
  std::vector<OutputBuffers> o(num_outs > 0 ? num_outs : 0);
  for (int i = 0; i < num_outs; i++) {
    o[i].validity = outs[i].validity; o[i].validity_size = outs[i].validity_size;
    o[i].data = outs[i].data; o[i].data_size = outs[i].data_size;
  }
  return Check(fp->fp->Evaluate(num_rows, c.data(), num_cols, o.data(), num_outs, out_indices, max_slots, num_selected,
                                mem_kind == GDV_MEM_DEVICE ? MemKind::kDevice : MemKind::kHost,
                                static_cast<hipStream_t>(stream), flags, num_selected_device));
  });
}
char* gdv_filter_project_dump_ir(const gdv_filter_project_t* fp) { return fp ? DupString(fp->fp->DumpIR()) : nullptr; }
int gdv_filter_project_kernel_shape(const gdv_filter_project_t* fp) { return fp ? fp->fp->which_kernel() : -1; }
int gdv_filter_project_set_tuning(gdv_filter_project_t* fp, const char* key, int64_t value) {
  return Guarded([&]() -> int {
    if (fp == nullptr || key == nullptr) return Fail(Status::Invalid("gdv_filter_project_set_tuning: null argument"));
    Status st = fp->fp->SetTuning(key, value);
    return st.ok() ? GDV_OK : Fail(st);
  });
}
void gdv_filter_project_free(gdv_filter_project_t* fp) { delete fp; }

// ---------------------------------------------------------------- JNI-shaped flat entry points
namespace {
// validity, [offsets,] data per field, in schema order
Status UnflattenInputs(const Schema& schema, const int64_t* addrs, const int64_t* sizes, int num_bufs,
                       std::vector<ColumnBuffers>* cols) {
  int want = 0;
  for (auto& f : schema) want += f.type.is_varlen() ? 3 : 2;
  if (num_bufs != want || (want > 0 && (addrs == nullptr || sizes == nullptr)))
    return Status::Invalid("expected " + std::to_string(want) + " input buffers (validity, [offsets,] data per field), got " +
                           std::to_string(num_bufs));
  cols->assign(schema.size(), ColumnBuffers());
  int b = 0;
  for (size_t i = 0; i < schema.size(); i++) {
    ColumnBuffers& c = (*cols)[i];
    c.validity = reinterpret_cast<const void*>(addrs[b]);
    c.validity_size = c.validity ? sizes[b] : 0;
    b++;
    if (schema[i].type.is_varlen()) {
      c.offsets = reinterpret_cast<const void*>(addrs[b]);
      c.offsets_size = sizes[b];
      b++;
    }
    c.data = reinterpret_cast<const void*>(addrs[b]);
    c.data_size = sizes[b];
    b++;
  }
  return Status::OK();
}
}  // namespace

int gdv_projector_evaluate_flat(const gdv_projector_t* p, int64_t num_rows, const int64_t* buf_addrs,
                                const int64_t* buf_sizes, int num_bufs, int sel_mode,
                                int64_t sel_addr, int64_t sel_slots, const int64_t* out_addrs,
                                int64_t* out_sizes, int num_out_bufs, int mem_kind) {
  return Guarded([&]() -> int {
  if (!p) return Fail(Status::Invalid("null projector"));
  std::vector<ColumnBuffers> cols;
  Status st = UnflattenInputs(p->p->schema(), buf_addrs, buf_sizes, num_bufs, &cols);
  if (!st.ok()) return Fail(st);
  const int n_out = p->p->num_outputs();
  int want = 0;
  for (int e = 0; e < n_out; e++) want += p->p->output_type(e).is_varlen() ? 3 : 2;
  if (num_out_bufs != want || out_addrs == nullptr || out_sizes == nullptr)
    return Fail(Status::Invalid("expected " + std::to_string(want) + " output buffers, got " +
                                std::to_string(num_out_bufs)));
  std::vector<OutputBuffers> o(n_out);
  std::vector<int> data_slot(n_out);
  int b = 0;
  for (int e = 0; e < n_out; e++) {
    o[e].validity = reinterpret_cast<void*>(out_addrs[b]);
    o[e].validity_size = out_sizes[b];
    b++;
    if (p->p->output_type(e).is_varlen()) {
      o[e].offsets = reinterpret_cast<void*>(out_addrs[b]);
      o[e].offsets_size = out_sizes[b];
      b++;
    }
    o[e].data = reinterpret_cast<void*>(out_addrs[b]);
    o[e].data_size = out_sizes[b];
    data_slot[e] = b++;
  }
  SelectionView sv;
  if (!ToSelectionMode(sel_mode, &sv.mode)) return Fail(Status::Invalid("bad selection mode"));
  sv.indices = reinterpret_cast<const void*>(sel_addr);
  sv.num_slots = sel_slots;
  const bool has_sel = sv.mode != SelectionMode::kNone;
  st = p->p->Evaluate(num_rows, cols.data(), static_cast<int>(cols.size()), has_sel ? &sv : nullptr,
                      o.data(), n_out, mem_kind == GDV_MEM_DEVICE ? MemKind::kDevice : MemKind::kHost,
                      nullptr, 0);
  for (int e = 0; e < n_out; e++)
    if (p->p->output_type(e).is_varlen()) out_sizes[data_slot[e]] = o[e].data_size;
  return Check(st);
  });
}

int gdv_filter_evaluate_flat(const gdv_filter_t* f, int64_t num_rows, const int64_t* buf_addrs,
                             const int64_t* buf_sizes, int num_bufs, int sel_mode, int64_t out_addr,
                             int64_t out_size_bytes, int64_t* num_selected, int mem_kind) {
  return Guarded([&]() -> int {
  if (!f) return Fail(Status::Invalid("null filter"));
  SelectionMode mode;
  if (!ToSelectionMode(sel_mode, &mode) || mode == SelectionMode::kNone)
    return Fail(Status::Invalid("bad selection mode"));
  std::vector<ColumnBuffers> cols;
  Status st = UnflattenInputs(f->f->schema(), buf_addrs, buf_sizes, num_bufs, &cols);
  if (!st.ok()) return Fail(st);
  const int w = mode == SelectionMode::kUInt16 ? 2 : mode == SelectionMode::kUInt32 ? 4 : 8;
  return Check(f->f->Evaluate(num_rows, cols.data(), static_cast<int>(cols.size()), mode,
                              reinterpret_cast<void*>(out_addr), out_size_bytes / w, num_selected,
                              mem_kind == GDV_MEM_DEVICE ? MemKind::kDevice : MemKind::kHost, nullptr));
  });
}

// ---------------------------------------------------------------- registry
int gdv_registry_size(void) { return static_cast<int>(FunctionRegistry::Get().all().size()); }
int gdv_registry_get(int index, const char** name, gdv_type_t* return_type, gdv_type_t* params,
                     int max_params, int* num_params) {
  return Guarded([&]() -> int {
  auto& all = FunctionRegistry::Get().all();
  if (index < 0 || index >= static_cast<int>(all.size())) return Fail(Status::Invalid("index out of range"));
  const FunctionDef& d = all[index];
  if (name) *name = d.name.c_str();
  if (return_type) *return_type = FromType(d.ret);
  if (num_params) *num_params = static_cast<int>(d.params.size());
  for (int i = 0; params && i < max_params && i < static_cast<int>(d.params.size()); i++)
    params[i] = FromType(d.params[i]);
  return GDV_OK;
  });
}

// ---------------------------------------------------------------- device helpers
int gdv_device_count(void) { return Runtime::DeviceCount(); }
int gdv_physical_device_count(void) { return Runtime::PhysicalDeviceCount(); }
int gdv_set_virtual_devices(int n) {
  if (n < 0 || n > Runtime::kMaxDevices) return Fail(Status::Invalid("bad virtual device count"));
  Runtime::SetVirtualDevices(n);
  return GDV_OK;
}
int gdv_set_device(int device) { return Check(Runtime::SelectDevice(device)); }
int gdv_get_device(void) { return Runtime::SelectedDevice(); }
int gdv_shard_bounds(int64_t num_rows, int num_shards, int shard, int64_t* lo, int64_t* hi) {
  if (num_rows < 0 || num_shards < 1 || shard < 0 || shard >= num_shards || !lo || !hi)
    return Fail(Status::Invalid("bad shard arguments"));
  // near-equal shards on 1024-row boundaries (one workgroup tile = one 128-byte line of every
  // validity bitmap); the last shard takes the ragged tail — gandiva_amd/shard.py: shard_bounds
  const int64_t align = 1024;
  const int64_t tiles = (num_rows + align - 1) / align;
  const int64_t per = tiles / num_shards, extra = tiles % num_shards;
  const int64_t lo_tile = shard * per + std::min<int64_t>(shard, extra);
  const int64_t hi_tile = lo_tile + per + (shard < extra ? 1 : 0);
  *lo = std::min(lo_tile * align, num_rows);
  *hi = std::min(hi_tile * align, num_rows);
  return GDV_OK;
}

// ---------------------------------------------------------------------------------------------------
// One call, all devices (round 6): one host thread per shard, each on its own device context and stream.
}  // extern "C"
namespace {
void ShardBounds(int64_t num_rows, int num_shards, int shard, int64_t* lo, int64_t* hi) {
  (void)gdv_shard_bounds(num_rows, num_shards, shard, lo, hi);
}
// body(shard, stream) runs on device devices[shard]; returns the first failing shard's status
template <typename Fn>
int RunShards(int n, const int32_t* devices, Fn&& body) {
  const int before = Runtime::SelectedDevice();
  std::vector<Status> st(static_cast<size_t>(n));
  auto work = [&](int s) {
    try {
      Status sel = Runtime::SelectDevice(devices[s]);
      if (!sel.ok()) { st[s] = sel; return; }
      Runtime& rt = Runtime::Get();
      Status dev = rt.EnsureDevice();
      if (!dev.ok()) { st[s] = dev; return; }
      hipStream_t stream = nullptr;
      Status a = rt.AcquireStream(&stream);
      if (!a.ok()) { st[s] = a; return; }
      st[s] = body(s, stream);
      (void)hipStreamSynchronize(stream);
      rt.ReleaseStream(stream);
    } catch (const std::bad_alloc&) {
      st[s] = Status::OutOfMemory("host allocation failed");
    } catch (const std::exception& e) {
      st[s] = Status::ExecutionError(std::string("internal error: ") + e.what());
    }
  };
  std::vector<std::thread> threads;
  threads.reserve(n > 1 ? n - 1 : 0);
  int started = 1;  // (shard 0 runs on the calling thread)
  try {
    for (int s = 1; s < n; s++, started++) threads.emplace_back(work, s);
  } catch (const std::exception&) {
    // the process is out of threads: the shards that got none run here, one after the other (a joinable std::thread
    // must never be destroyed — the ones that did start are joined below whatever happens)
  }
  work(0);
  for (int s = started; s < n; s++) work(s);
  for (auto& t : threads) t.join();
  if (before >= 0) (void)Runtime::SelectDevice(before);
  for (int s = 0; s < n; s++)
    if (!st[s].ok())
      return Fail(Status(st[s].code, "shard " + std::to_string(s) + " (device " + std::to_string(devices[s]) + "): " + st[s].msg));
  return GDV_OK;
}
std::vector<OutputBuffers> ToOutputs(const gdv_out_column_t* outs, int n) {
  std::vector<OutputBuffers> o(n > 0 ? n : 0);
  for (int i = 0; i < n; i++) {
    o[i].validity = outs[i].validity; o[i].validity_size = outs[i].validity_size;
    o[i].data = outs[i].data; o[i].data_size = outs[i].data_size;
    o[i].offsets = outs[i].offsets; o[i].offsets_size = outs[i].offsets_size;
  }
  return o;
}
int IndexWidth(SelectionMode m) { return m == SelectionMode::kUInt16 ? 2 : m == SelectionMode::kUInt32 ? 4 : 8; }
}  // namespace
extern "C" {

int gdv_projector_evaluate_sharded(const gdv_projector_t* p, int64_t num_rows, int num_cols, int num_outs,
                                   gdv_shard_t* shards, int num_shards, uint32_t flags) {
  return Guarded([&]() -> int {
  (void)flags;
  if (!p) return Fail(Status::Invalid("null projector"));
  if (num_shards < 1 || !shards || num_rows < 0) return Fail(Status::Invalid("bad shard list"));
  if (p->p->plan().mode != SelectionMode::kNone) return Fail(Status::Invalid("sharded evaluation takes row-mode projectors"));
  std::vector<int32_t> devices(num_shards);
  for (int s = 0; s < num_shards; s++) {
    devices[s] = shards[s].device;
    if ((num_cols > 0 && !shards[s].cols) || !shards[s].outs) return Fail(Status::Invalid("shard without columns / outputs"));
  }
  return RunShards(num_shards, devices.data(), [&](int s, hipStream_t stream) -> Status {
    int64_t lo = 0, hi = 0;
    ShardBounds(num_rows, num_shards, s, &lo, &hi);
    if (hi == lo) return Status::OK();
    std::vector<ColumnBuffers> c = ToColumns(shards[s].cols, num_cols);
    std::vector<OutputBuffers> o = ToOutputs(shards[s].outs, num_outs);
    Status st = p->p->Evaluate(hi - lo, c.data(), num_cols, nullptr, o.data(), num_outs, MemKind::kDevice, stream, 0);
    for (int i = 0; i < num_outs; i++) shards[s].outs[i].data_size = o[i].data_size;  // var-len: bytes produced / needed
    return st;
  });
  });
}

int gdv_filter_evaluate_sharded(const gdv_filter_t* f, int64_t num_rows, int num_cols, int selection_mode,
                                gdv_shard_t* shards, int num_shards, uint32_t flags, int64_t* total_selected) {
  return Guarded([&]() -> int {
  if (!f) return Fail(Status::Invalid("null filter"));
  if (num_shards < 1 || !shards || num_rows < 0) return Fail(Status::Invalid("bad shard list"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode) || mode == SelectionMode::kNone) return Fail(Status::Invalid("bad selection mode"));
  std::vector<int32_t> devices(num_shards);
  for (int s = 0; s < num_shards; s++) {
    devices[s] = shards[s].device;
    shards[s].num_selected = 0;
    if ((num_cols > 0 && !shards[s].cols) || !shards[s].out_indices) return Fail(Status::Invalid("shard without columns / indices"));
  }
  const bool global = (flags & GDV_SHARD_GLOBAL_INDICES) != 0;
  int rc = RunShards(num_shards, devices.data(), [&](int s, hipStream_t stream) -> Status {
    int64_t lo = 0, hi = 0;
    ShardBounds(num_rows, num_shards, s, &lo, &hi);
    if (hi == lo) return Status::OK();
    std::vector<ColumnBuffers> c = ToColumns(shards[s].cols, num_cols);
    return f->f->Evaluate(hi - lo, c.data(), num_cols, mode, shards[s].out_indices, shards[s].max_slots,
                          &shards[s].num_selected, MemKind::kDevice, stream, 0, nullptr, global ? lo : 0);
  });
  if (rc != GDV_OK) return rc;
  if (total_selected) {
    *total_selected = 0;
    for (int s = 0; s < num_shards; s++) *total_selected += shards[s].num_selected;
  }
  return GDV_OK;
  });
}

int gdv_filter_gather_sharded(const gdv_shard_t* shards, int num_shards, int selection_mode, int dst_device,
                              void* dst_indices, int64_t dst_slots) {
  return Guarded([&]() -> int {
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode) || mode == SelectionMode::kNone) return Fail(Status::Invalid("bad selection mode"));
  if (num_shards < 1 || !shards || !dst_indices) return Fail(Status::Invalid("bad shard list"));
  const int w = IndexWidth(mode);
  int64_t total = 0;
  for (int s = 0; s < num_shards; s++) total += shards[s].num_selected;
  if (total > dst_slots) return Fail(Status::Invalid("gathered selection vector needs " + std::to_string(total) + " slots"));
  const int before = Runtime::SelectedDevice();
  Status st = Runtime::SelectDevice(dst_device);
  if (!st.ok()) return Fail(st);
  Runtime& dst = Runtime::Get();
  st = dst.EnsureDevice();
  hipStream_t stream = nullptr;
  if (st.ok()) st = dst.AcquireStream(&stream);
  if (st.ok()) {
    int64_t at = 0;
    for (int s = 0; s < num_shards && st.ok(); s++) {
      const int64_t n = shards[s].num_selected;
      if (n > 0) {
        // (virtual devices share a physical one: the copy is then an ordinary device-to-device one)
        const int src_phys = Runtime::ForDevice(shards[s].device).physical();
        hipError_t e = src_phys == dst.physical()
                           ? hipMemcpyAsync(static_cast<char*>(dst_indices) + at * w, shards[s].out_indices, n * w, hipMemcpyDeviceToDevice, stream)
                           : hipMemcpyPeerAsync(static_cast<char*>(dst_indices) + at * w, dst.physical(), shards[s].out_indices, src_phys, n * w, stream);
        if (e != hipSuccess) st = Status::ExecutionError(std::string("gather: ") + hipGetErrorString(e));
      }
      at += n;
    }
    if (hipStreamSynchronize(stream) != hipSuccess && st.ok()) st = Status::ExecutionError("gather: stream synchronisation failed");
    dst.ReleaseStream(stream);
  }
  if (before >= 0) (void)Runtime::SelectDevice(before);
  return Check(st);
  });
}

int gdv_projector_evaluate_host_sharded(const gdv_projector_t* p, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                        gdv_out_column_t* outs, int num_outs, const int32_t* devices, int num_devices) {
  return Guarded([&]() -> int {
  if (!p) return Fail(Status::Invalid("null projector"));
  if (num_cols > 0 && !cols) return Fail(Status::Invalid("null column array"));
  if (!outs || !devices || num_devices < 1) return Fail(Status::Invalid("outputs and a device list are required"));
  if (p->p->plan().mode != SelectionMode::kNone) return Fail(Status::Invalid("sharded evaluation takes row-mode projectors"));
  if (num_outs != p->p->num_outputs()) return Fail(Status::Invalid("number of outputs does not match the projector"));
  bool varlen_out = false;
  for (int i = 0; i < num_outs; i++) varlen_out |= p->p->output_type(i).is_varlen();
  // var-len outputs: byte positions depend on the shards before -> one device; tiny batches: not worth the threads
  const int n = (varlen_out || num_rows < 2048) ? 1 : num_devices;
  if (n == 1) {
    const int before = Runtime::SelectedDevice();
    Status sel = Runtime::SelectDevice(devices[0]);
    if (!sel.ok()) return Fail(sel);
    int rc = ProjectorEvaluate(p, num_rows, cols, num_cols, nullptr, nullptr, outs, num_outs, GDV_MEM_HOST, nullptr, 0);
    if (before >= 0) (void)Runtime::SelectDevice(before);
    return rc;
  }
  for (int i = 0; i < num_outs; i++) {
    const DataType& t = p->p->output_type(i);
    const int64_t vneed = (num_rows + 7) / 8, dneed = t.id == kBool ? (num_rows + 7) / 8 : Projector::DataBytes(t, num_rows);
    if (!outs[i].validity || !outs[i].data || outs[i].validity_size < vneed || outs[i].data_size < dneed)
      return Fail(Status::Invalid("output buffer " + std::to_string(i) + " too small"));
  }
  return RunShards(n, devices, [&](int s, hipStream_t stream) -> Status {
    int64_t lo = 0, hi = 0;
    ShardBounds(num_rows, n, s, &lo, &hi);
    if (hi == lo) return Status::OK();
    // a slice of the caller's batch: array offset + lo (var-len columns keep their whole byte buffer)
    std::vector<ColumnBuffers> c = ToColumns(cols, num_cols);
    for (auto& col : c) col.offset += lo;
    std::vector<OutputBuffers> o(num_outs);
    for (int i = 0; i < num_outs; i++) {
      const DataType& t = p->p->output_type(i);
      // lo is a multiple of 1024: whole bytes of every bitmap
      o[i].validity = static_cast<char*>(outs[i].validity) + lo / 8;
      o[i].validity_size = (hi - lo + 7) / 8;
      if (t.id == kBool) {
        o[i].data = static_cast<char*>(outs[i].data) + lo / 8;
        o[i].data_size = (hi - lo + 7) / 8;
      } else {
        o[i].data = static_cast<char*>(outs[i].data) + lo * t.byte_width();
        o[i].data_size = (hi - lo) * t.byte_width();
      }
    }
    return p->p->Evaluate(hi - lo, c.data(), num_cols, nullptr, o.data(), num_outs, MemKind::kHost, stream, 0);
  });
  });
}

int gdv_filter_evaluate_host_sharded(const gdv_filter_t* f, int64_t num_rows, const gdv_column_t* cols, int num_cols,
                                     int selection_mode, void* out_indices, int64_t max_slots, int64_t* num_selected,
                                     const int32_t* devices, int num_devices) {
  return Guarded([&]() -> int {
  if (!f) return Fail(Status::Invalid("null filter"));
  if (num_cols > 0 && !cols) return Fail(Status::Invalid("null column array"));
  if (!out_indices || !num_selected || !devices || num_devices < 1) return Fail(Status::Invalid("Selection vector cannot be null"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode) || mode == SelectionMode::kNone) return Fail(Status::Invalid("bad selection mode"));
  if (max_slots < num_rows)
    return Fail(Status::Invalid("Selection vector too small: max slots " + std::to_string(max_slots) + " < rows " + std::to_string(num_rows)));
  const int n = num_rows < 2048 ? 1 : num_devices;
  const int w = IndexWidth(mode);
  std::vector<int64_t> counts(n, 0);
  int rc = RunShards(n, devices, [&](int s, hipStream_t stream) -> Status {
    int64_t lo = 0, hi = 0;
    ShardBounds(num_rows, n, s, &lo, &hi);
    if (hi == lo) return Status::OK();
    std::vector<ColumnBuffers> c = ToColumns(cols, num_cols);
    for (auto& col : c) col.offset += lo;
    // shard s can select at most hi - lo rows: its part of the vector is [lo, hi), closed up below
    return f->f->Evaluate(hi - lo, c.data(), num_cols, mode, static_cast<char*>(out_indices) + lo * w, hi - lo, &counts[s],
                          MemKind::kHost, stream, 0, nullptr, lo);
  });
  if (rc != GDV_OK) return rc;
  int64_t at = 0;
  for (int s = 0; s < n; s++) {
    int64_t lo = 0, hi = 0;
    ShardBounds(num_rows, n, s, &lo, &hi);
    if (counts[s] > 0 && at != lo)
      std::memmove(static_cast<char*>(out_indices) + at * w, static_cast<char*>(out_indices) + lo * w, static_cast<size_t>(counts[s]) * w);
    at += counts[s];
  }
  *num_selected = at;
  return GDV_OK;
  });
}
int gdv_device_num_cus(void) { return Runtime::Get().num_cus(); }
const char* gdv_device_arch(void) { return Runtime::Get().arch().c_str(); }
int gdv_device_alloc(int64_t bytes, void** ptr) {
  if (!ptr || bytes < 0) return Fail(Status::Invalid("bad argument"));
  return Check(Runtime::Get().Alloc(static_cast<size_t>(bytes ? bytes : 1), ptr));
}
int gdv_device_free(void* ptr) { Runtime::Get().Free(ptr); return GDV_OK; }
int gdv_device_pool_create(gdv_device_pool_t** out) {
  return Guarded([&]() -> int {
    if (!out) return Fail(Status::Invalid("null output pointer"));
    Status st = Runtime::Get().EnsureDevice();
    if (!st.ok()) return Fail(st);
    *out = new gdv_device_pool();
    return GDV_OK;
  });
}
void gdv_device_pool_destroy(gdv_device_pool_t* pool) { delete pool; }
int gdv_device_pool_reserve_set(gdv_device_pool_t* pool, int count, int64_t bytes, int candidates, void** ptrs, double* rates,
                                int* tried, int* kept) {
  return Guarded([&]() -> int {
    if (!pool) return Fail(Status::Invalid("null pool"));
    return Check(pool->pool.ReserveSet(count, bytes, candidates, ptrs, rates, tried, kept));
  });
}
int gdv_device_pool_alloc(gdv_device_pool_t* pool, int64_t bytes, void** ptr) {
  return Guarded([&]() -> int {
    if (!pool) return Fail(Status::Invalid("null pool"));
    return Check(pool->pool.Alloc(bytes, ptr));
  });
}
int gdv_device_pool_free(gdv_device_pool_t* pool, void* ptr) {
  return Guarded([&]() -> int {
    if (!pool) return Fail(Status::Invalid("null pool"));
    return Check(pool->pool.Free(ptr));
  });
}
int gdv_device_pool_trim(gdv_device_pool_t* pool) {
  return Guarded([&]() -> int {
    if (!pool) return Fail(Status::Invalid("null pool"));
    return Check(pool->pool.Trim());
  });
}
int64_t gdv_device_pool_bytes(const gdv_device_pool_t* pool, int64_t* in_use) { return pool ? pool->pool.bytes_held(in_use) : 0; }
int gdv_memcpy_h2d(void* dst, const void* src, int64_t bytes) {
  hipError_t e = hipMemcpy(dst, src, static_cast<size_t>(bytes), hipMemcpyHostToDevice);
  return e == hipSuccess ? GDV_OK : Fail(Status::ExecutionError(hipGetErrorString(e)));
}
int gdv_memcpy_d2h(void* dst, const void* src, int64_t bytes) {
  hipError_t e = hipMemcpy(dst, src, static_cast<size_t>(bytes), hipMemcpyDeviceToHost);
  return e == hipSuccess ? GDV_OK : Fail(Status::ExecutionError(hipGetErrorString(e)));
}
int gdv_host_register(void* ptr, int64_t bytes) {
  return Guarded([&]() -> int {
  if (bytes < 0) return Fail(Status::Invalid("bad argument"));
  return Check(HostRegistry::Get().Register(ptr, static_cast<size_t>(bytes)));
  });
}
int gdv_host_unregister(void* ptr) {
  return Guarded([&]() -> int { return Check(HostRegistry::Get().Unregister(ptr)); });
}
int gdv_host_alloc(int64_t bytes, void** ptr) {
  return Guarded([&]() -> int {
  if (bytes < 0) return Fail(Status::Invalid("bad argument"));
  return Check(HostRegistry::Get().Alloc(static_cast<size_t>(bytes), ptr));
  });
}
int gdv_host_free(void* ptr) {
  return Guarded([&]() -> int { return Check(HostRegistry::Get().Free(ptr)); });
}
int64_t gdv_host_staged_bytes(void) { return HostRegistry::StagedBytes().load(std::memory_order_relaxed); }
int gdv_device_hbm_ceilings(int64_t bytes, double* read_gbs, double* write_gbs, double* copy_gbs) {
  return Guarded([&]() -> int {
  if (bytes < (1 << 20) || !read_gbs || !write_gbs || !copy_gbs) return Fail(Status::Invalid("bad argument"));
  bytes &= ~int64_t{4095};
  Runtime& rt = Runtime::Get();
  Status st = rt.EnsureDevice();
  if (!st.ok()) return Fail(st);
  DeviceBuffer a, b;
  st = a.Allocate(static_cast<size_t>(bytes));
  if (st.ok()) st = b.Allocate(static_cast<size_t>(bytes));
  if (!st.ok()) return Fail(st);
  hipError_t e = hipMemset(a.get(), 1, static_cast<size_t>(bytes));
  if (e == hipSuccess) e = hipMemset(b.get(), 2, static_cast<size_t>(bytes));
  if (e == hipSuccess)
    e = MeasureHbmCeilings(a.get(), b.get(), static_cast<size_t>(bytes), rt.num_cus() * 16, read_gbs, write_gbs, copy_gbs);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e == hipSuccess ? GDV_OK : Fail(Status::ExecutionError(hipGetErrorString(e)));
  });
}
int gdv_device_stream_ceiling(int64_t bytes_per_stream, int num_read, int num_write, double* gbs, int* workgroups_per_cu,
                              int* nontemporal) {
  return Guarded([&]() -> int {
  if (bytes_per_stream < (1 << 20) || !gbs || num_read < 0 || num_write < 0 || num_read + num_write < 1 ||
      num_read > 10 || num_write > 10)
    return Fail(Status::Invalid("bad argument"));
  bytes_per_stream &= ~int64_t{8191};
  Runtime& rt = Runtime::Get();
  Status st = rt.EnsureDevice();
  if (!st.ok()) return Fail(st);
  std::vector<DeviceBuffer> bufs(num_read + num_write);
  std::vector<void*> ptrs;
  for (auto& b : bufs) {
    st = b.Allocate(static_cast<size_t>(bytes_per_stream));
    if (!st.ok()) return Fail(st);
    hipError_t e = hipMemset(b.get(), 1, static_cast<size_t>(bytes_per_stream));
    if (e != hipSuccess) return Fail(Status::ExecutionError(hipGetErrorString(e)));
    ptrs.push_back(b.get());
  }
  int wg = 0, nt = 0;
  hipError_t e = MeasureStreamCeiling(ptrs.data(), num_read, num_write, static_cast<size_t>(bytes_per_stream / 8), rt.num_cus(),
                                      gbs, &wg, &nt);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipErrorInvalidValue) return Fail(Status::Invalid("no ceiling kernel for this (reads, writes) shape"));
  if (workgroups_per_cu) *workgroups_per_cu = wg;
  if (nontemporal) *nontemporal = nt;
  return e == hipSuccess ? GDV_OK : Fail(Status::ExecutionError(hipGetErrorString(e)));
  });
}
int gdv_device_stream_ceiling_on(void* const* streams, int num_read, int num_write, int64_t elems, double* gbs,
                                 int* workgroups_per_cu, int* nontemporal) {
  return Guarded([&]() -> int {
  if (!streams || !gbs || elems < 1024 || num_read < 0 || num_write < 0 || num_read + num_write < 1 || num_read > 10 ||
      num_write > 10)
    return Fail(Status::Invalid("bad argument"));
  for (int i = 0; i < num_read + num_write; i++)
    if (streams[i] == nullptr) return Fail(Status::Invalid("null stream"));
  Runtime& rt = Runtime::Get();
  Status st = rt.EnsureDevice();
  if (!st.ok()) return Fail(st);
  int wg = 0, nt = 0;
  hipError_t e = MeasureStreamCeiling(streams, num_read, num_write, static_cast<size_t>(elems), rt.num_cus(), gbs, &wg, &nt);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipErrorInvalidValue) return Fail(Status::Invalid("no ceiling kernel for this (reads, writes) shape"));
  if (workgroups_per_cu) *workgroups_per_cu = wg;
  if (nontemporal) *nontemporal = nt;
  return e == hipSuccess ? GDV_OK : Fail(Status::ExecutionError(hipGetErrorString(e)));
  });
}
int gdv_device_synchronize(void) {
  hipError_t e = hipDeviceSynchronize();
  return e == hipSuccess ? GDV_OK : Fail(Status::ExecutionError(hipGetErrorString(e)));
}

// ---------------------------------------------------------------- C device data interface
namespace {

// struct array (one child per field) -> gdv_column_t[]; sizes are derived from
// offset + length and the field type because the C data interface carries no buffer sizes
Status ImportBatch(const Schema& schema, const ArrowDeviceArray* batch, hipStream_t stream,
                   std::vector<ColumnBuffers>* cols, MemKind* mem, int64_t* num_rows) {
  if (batch == nullptr) return Status::Invalid("null ArrowDeviceArray");
  const ArrowArray& a = batch->array;
  if (a.release == nullptr) return Status::Invalid("ArrowDeviceArray was already released");
  if (a.n_children != static_cast<int64_t>(schema.size()))
    return Status::Invalid("ArrowDeviceArray has " + std::to_string(a.n_children) +
                           " children, the schema has " + std::to_string(schema.size()) + " fields");
  if (a.offset != 0) return Status::Invalid("struct-level offset is not supported");
  switch (batch->device_type) {
    case ARROW_DEVICE_ROCM: *mem = MemKind::kDevice; break;
    case ARROW_DEVICE_CPU: case ARROW_DEVICE_ROCM_HOST: *mem = MemKind::kHost; break;
    default: return Status::Invalid("unsupported ArrowDeviceType " + std::to_string(batch->device_type));
  }
  if (batch->sync_event != nullptr && *mem == MemKind::kDevice)
    GDV_HIP_RETURN_NOT_OK(hipStreamWaitEvent(stream, *static_cast<hipEvent_t*>(batch->sync_event), 0));
  *num_rows = a.length;
  cols->assign(schema.size(), ColumnBuffers());
  for (size_t i = 0; i < schema.size(); i++) {
    const ArrowArray* c = a.children[i];
    if (c == nullptr) return Status::Invalid("null child array");
    if (c->length != a.length) return Status::Invalid("child length differs from the batch length");
    const DataType& t = schema[i].type;
    ColumnBuffers& col = (*cols)[i];
    const int64_t rows = c->offset + c->length;
    col.offset = c->offset;
    const int64_t want = t.is_varlen() ? 3 : 2;
    if (c->n_buffers < want) continue;  // e.g. a null-type child: fails later only if referenced
    col.validity = c->buffers[0];
    col.validity_size = col.validity ? (rows + 7) / 8 : 0;
    if (t.is_varlen()) {
      col.offsets = c->buffers[1];
      col.offsets_size = (rows + 1) * 4;
      col.data = c->buffers[2];
      int32_t last = 0;  // byte extent = the last offset
      if (col.offsets != nullptr && rows >= 0) {
        const char* src = static_cast<const char*>(col.offsets) + rows * 4;
        if (*mem == MemKind::kDevice) {
          GDV_HIP_RETURN_NOT_OK(hipMemcpyAsync(&last, src, 4, hipMemcpyDeviceToHost, stream));
          GDV_HIP_RETURN_NOT_OK(hipStreamSynchronize(stream));
        } else {
          std::memcpy(&last, src, 4);
        }
      }
      col.data_size = last;
    } else {
      col.data = c->buffers[1];
      col.data_size = t.id == kBool ? (rows + 7) / 8 : rows * t.byte_width();
    }
  }
  return Status::OK();
}

}  // namespace

int gdv_projector_evaluate_device_array(const gdv_projector_t* p, const ArrowDeviceArray* batch,
                                        const gdv_selection_t* sel, gdv_out_column_t* outs,
                                        int num_outs, void* stream, uint32_t flags) {
  return Guarded([&]() -> int {
  if (!p) return Fail(Status::Invalid("null projector"));
  if (!outs) return Fail(Status::Invalid("Output array vector cannot be null"));
  std::vector<ColumnBuffers> cols;
  MemKind mem;
  int64_t rows = 0;
  Status st = ImportBatch(p->p->schema(), batch, static_cast<hipStream_t>(stream), &cols, &mem, &rows);
  if (!st.ok()) return Fail(st);
  std::vector<OutputBuffers> o(num_outs > 0 ? num_outs : 0);
  for (int i = 0; i < num_outs; i++) {
    o[i].validity = outs[i].validity;
    o[i].validity_size = outs[i].validity_size;
    o[i].data = outs[i].data;
    o[i].data_size = outs[i].data_size;
    o[i].offsets = outs[i].offsets;
    o[i].offsets_size = outs[i].offsets_size;
  }
  SelectionView sv;
  if (sel) {
    if (!ToSelectionMode(sel->mode, &sv.mode)) return Fail(Status::Invalid("bad selection mode"));
    sv.indices = sel->indices;
    sv.num_slots = sel->num_slots;
  }
  st = p->p->Evaluate(rows, cols.data(), static_cast<int>(cols.size()), sel ? &sv : nullptr, o.data(),
                      num_outs, mem, static_cast<hipStream_t>(stream), flags);
  for (int i = 0; i < num_outs; i++) outs[i].data_size = o[i].data_size;
  return Check(st);
  });
}

int gdv_filter_evaluate_device_array(const gdv_filter_t* f, const ArrowDeviceArray* batch,
                                     int selection_mode, void* out_indices, int64_t max_slots,
                                     int64_t* num_selected, void* stream) {
  return Guarded([&]() -> int {
  if (!f) return Fail(Status::Invalid("null filter"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  std::vector<ColumnBuffers> cols;
  MemKind mem;
  int64_t rows = 0;
  Status st = ImportBatch(f->f->schema(), batch, static_cast<hipStream_t>(stream), &cols, &mem, &rows);
  if (!st.ok()) return Fail(st);
  return Check(f->f->Evaluate(rows, cols.data(), static_cast<int>(cols.size()), mode, out_indices,
                              max_slots, num_selected, mem, static_cast<hipStream_t>(stream)));
  });
}

// ---------------------------------------------------------------- C device data export
namespace {

// Buffers of one exported batch: owned jointly by the parent array and every child (a
// consumer may move children out and release them on their own).
struct ExportBlock {
  MemKind mem = MemKind::kHost;
  std::vector<void*> bufs;
  hipEvent_t event = nullptr;
  ~ExportBlock() {
    for (void* b : bufs) {
      if (mem == MemKind::kDevice) Runtime::Get().Free(b); else std::free(b);
    }
    if (event != nullptr) (void)hipEventDestroy(event);
  }
  Status Allocate(int64_t bytes, void** out) {
    const size_t padded = static_cast<size_t>((std::max<int64_t>(bytes, 1) + 63) / 64 * 64);
    if (mem == MemKind::kDevice) {
      GDV_RETURN_NOT_OK(Runtime::Get().Alloc(padded, out));
    } else {
      *out = std::aligned_alloc(64, padded);
      if (*out == nullptr) return Status::OutOfMemory("host allocation of " + std::to_string(padded) + " bytes failed");
    }
    bufs.push_back(*out);
    return Status::OK();
  }
  void Drop(void* b) {  // give one buffer back early (var-len data regrown)
    for (auto it = bufs.begin(); it != bufs.end(); ++it)
      if (*it == b) { bufs.erase(it); break; }
    if (mem == MemKind::kDevice) Runtime::Get().Free(b); else std::free(b);
  }
};

struct ExportNode {  // private_data of an exported ArrowArray
  std::shared_ptr<ExportBlock> block;
  const void* buffers[3] = {nullptr, nullptr, nullptr};
  std::vector<ArrowArray*> children;
};

void ReleaseExportedArray(ArrowArray* a) {
  if (a == nullptr || a->release == nullptr) return;
  auto* node = static_cast<ExportNode*>(a->private_data);
  for (ArrowArray* c : node->children) {
    if (c->release != nullptr) c->release(c);
    delete c;
  }
  delete node;
  a->release = nullptr;
}

struct SchemaNode {  // private_data of an exported ArrowSchema
  std::string format, name;
  std::vector<ArrowSchema*> children;
};

void ReleaseExportedSchema(ArrowSchema* s) {
  if (s == nullptr || s->release == nullptr) return;
  auto* node = static_cast<SchemaNode*>(s->private_data);
  for (ArrowSchema* c : node->children) {
    if (c->release != nullptr) c->release(c);
    delete c;
  }
  delete node;
  s->release = nullptr;
}

// Arrow C data interface format string (pyarrow/include/arrow/c/abi.h; format spec §"Data
// type description")
std::string FormatOf(const DataType& t) {
  static const char* const units = "smun";
  switch (t.id) {
    case kBool: return "b";
    case kInt8: return "c";
    case kUInt8: return "C";
    case kInt16: return "s";
    case kUInt16: return "S";
    case kInt32: return "i";
    case kUInt32: return "I";
    case kInt64: return "l";
    case kUInt64: return "L";
    case kFloat: return "f";
    case kDouble: return "g";
    case kString: return "u";
    case kBinary: return "z";
    case kDate32: return "tdD";
    case kDate64: return "tdm";
    case kTimestamp: return std::string("ts") + units[t.precision & 3] + ":";
    case kTime32: return std::string("tt") + units[t.precision & 3];
    case kTime64: return std::string("tt") + units[t.precision & 3];
    case kDecimal128: return "d:" + std::to_string(t.precision) + "," + std::to_string(t.scale);
    default: return "n";
  }
}

void FillSchema(ArrowSchema* s, const std::string& format, const std::string& name, int64_t flags) {
  auto* node = new SchemaNode{format, name, {}};
  std::memset(s, 0, sizeof(*s));
  s->format = node->format.c_str();
  s->name = node->name.c_str();
  s->flags = flags;
  s->private_data = node;
  s->release = ReleaseExportedSchema;
}

}  // namespace

int gdv_projector_evaluate_export(const gdv_projector_t* p, const ArrowDeviceArray* batch,
                                  const gdv_selection_t* sel, void* stream_ptr,
                                  ArrowDeviceArray* out, ArrowSchema* out_schema) {
  return Guarded([&]() -> int {
  if (!p) return Fail(Status::Invalid("null projector"));
  if (!out) return Fail(Status::Invalid("null output ArrowDeviceArray"));
  hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
  std::vector<ColumnBuffers> cols;
  MemKind mem;
  int64_t rows = 0;
  Status st = ImportBatch(p->p->schema(), batch, stream, &cols, &mem, &rows);
  if (!st.ok()) return Fail(st);
  SelectionView sv;
  if (sel) {
    if (!ToSelectionMode(sel->mode, &sv.mode)) return Fail(Status::Invalid("bad selection mode"));
    sv.indices = sel->indices;
    sv.num_slots = sel->num_slots;
  }
  const int64_t out_rows = sel ? sel->num_slots : rows;
  const int n_out = p->p->num_outputs();
  const bool dev = mem == MemKind::kDevice;
  auto block = std::make_shared<ExportBlock>();
  block->mem = mem;
  std::vector<OutputBuffers> o(n_out);
  int64_t varlen_guess = 64;
  for (auto& c : cols) if (c.offsets != nullptr) varlen_guess += c.data_size;
  for (int e = 0; e < n_out; e++) {
    const DataType& t = p->p->output_type(e);
    o[e].validity_size = dev ? Projector::ValidityBytes(out_rows) : (out_rows + 7) / 8;
    if (t.is_varlen()) {
      o[e].offsets_size = (out_rows + 1) * 4;
      const int64_t hint = p->p->VarlenBytesHint(e, out_rows);  // what earlier batches produced per row
      o[e].data_size = hint > 0 ? hint : varlen_guess;
      st = block->Allocate(o[e].offsets_size, &o[e].offsets);
      if (!st.ok()) return Fail(st);
    } else {
      o[e].data_size = t.id == kBool ? o[e].validity_size : Projector::DataBytes(t, out_rows);
    }
    st = block->Allocate(o[e].validity_size, &o[e].validity);
    if (st.ok()) st = block->Allocate(o[e].data_size, &o[e].data);
    if (!st.ok()) return Fail(st);
  }
  for (int attempt = 0; attempt < 2; attempt++) {
    std::vector<int64_t> caps(n_out);
    for (int e = 0; e < n_out; e++) caps[e] = o[e].data_size;
    st = p->p->Evaluate(rows, cols.data(), static_cast<int>(cols.size()), sel ? &sv : nullptr, o.data(),
                        n_out, mem, stream, 0);
    if (st.ok() || attempt == 1) break;
    bool grown = false;  // a var-len output needed more bytes than guessed: regrow once
    for (int e = 0; e < n_out; e++) {
      if (!p->p->output_type(e).is_varlen()) continue;
      if (o[e].data_size > caps[e]) {
        block->Drop(o[e].data);
        Status a = block->Allocate(o[e].data_size, &o[e].data);
        if (!a.ok()) return Fail(a);
        grown = true;
      } else {
        o[e].data_size = caps[e];
      }
    }
    if (!grown) break;
  }
  if (!st.ok()) return Fail(st);
  if (dev) {
    hipError_t he = hipEventCreateWithFlags(&block->event, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventRecord(block->event, stream);
    if (he != hipSuccess) return Fail(Status::ExecutionError(hipGetErrorString(he)));
  }
  // ---- assemble the struct array
  auto* parent = new ExportNode();
  parent->block = block;
  for (int e = 0; e < n_out; e++) {
    const DataType& t = p->p->output_type(e);
    auto* node = new ExportNode();
    node->block = block;
    auto* child = new ArrowArray();
    std::memset(child, 0, sizeof(*child));
    child->length = out_rows;
    child->null_count = -1;  // not computed
    node->buffers[0] = o[e].validity;
    if (t.is_varlen()) {
      node->buffers[1] = o[e].offsets;
      node->buffers[2] = o[e].data;
      child->n_buffers = 3;
    } else {
      node->buffers[1] = o[e].data;
      child->n_buffers = 2;
    }
    child->buffers = node->buffers;
    child->private_data = node;
    child->release = ReleaseExportedArray;
    parent->children.push_back(child);
  }
  std::memset(out, 0, sizeof(*out));
  out->array.length = out_rows;
  out->array.null_count = 0;
  out->array.n_buffers = 1;
  out->array.buffers = parent->buffers;  // {NULL}: a struct array without a validity bitmap
  out->array.n_children = n_out;
  out->array.children = parent->children.data();
  out->array.private_data = parent;
  out->array.release = ReleaseExportedArray;
  int device_id = 0;
  if (dev) (void)hipGetDevice(&device_id);
  out->device_id = dev ? device_id : -1;
  out->device_type = dev ? ARROW_DEVICE_ROCM : ARROW_DEVICE_CPU;
  out->sync_event = dev ? static_cast<void*>(&block->event) : nullptr;
  if (out_schema != nullptr) {
    FillSchema(out_schema, "+s", "", 0);
    auto* sn = static_cast<SchemaNode*>(out_schema->private_data);
    for (int e = 0; e < n_out; e++) {
      auto* cs = new ArrowSchema();
      FillSchema(cs, FormatOf(p->p->output_type(e)), p->output_names[e], /*ARROW_FLAG_NULLABLE*/ 2);
      sn->children.push_back(cs);
    }
    out_schema->n_children = n_out;
    out_schema->children = sn->children.data();
  }
  return GDV_OK;
  });
}

// ---------------------------------------------------------------- build support
char* gdv_tier0_program(const gdv_schema_t* schema, gdv_expression_t* const* exprs, int num_exprs, int is_condition) {
  return GuardedPtr([&]() -> char* {
    if (!schema || !exprs || num_exprs < 1) return FailPtr<char>("schema and expressions are required");
    std::vector<ExpressionPtr> v;
    for (int i = 0; i < num_exprs; i++) {
      if (!exprs[i]) return FailPtr<char>("null expression");
      v.push_back(exprs[i]->expr);
    }
    std::string text;
    Status st = Tier0Describe(schema->fields, v, is_condition != 0, &text);
    if (!st.ok()) {
      Fail(st);
      return nullptr;
    }
    return DupString(text);
  });
}
int64_t gdv_tier0_launches(void) { return Tier0Launches(); }
void gdv_shutdown(void) { Runtime::ShutdownBackgroundCompiler(); }

int gdv_precompile_projector(const gdv_schema_t* schema, gdv_expression_t* const* exprs,
                             int num_exprs, int selection_mode) {
  return Guarded([&]() -> int {
  if (!schema) return Fail(Status::Invalid("null schema"));
  std::vector<ExpressionPtr> ex;
  if (!CollectExprs(exprs, num_exprs, &ex)) return Fail(Status::Invalid("null expression"));
  SelectionMode mode;
  if (!ToSelectionMode(selection_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  return Check(PrecompileProjector(schema->fields, ex, mode));
  });
}
int gdv_precompile_filter_project(const gdv_schema_t* schema, gdv_expression_t* condition, gdv_expression_t* const* exprs,
                                  int num_exprs, int index_mode) {
  return Guarded([&]() -> int {
  if (!schema || !condition || !condition->expr || !exprs || num_exprs <= 0) return Fail(Status::Invalid("null argument"));
  SelectionMode mode;
  if (!ToSelectionMode(index_mode, &mode)) return Fail(Status::Invalid("bad selection mode"));
  std::vector<ExpressionPtr> ex;
  for (int i = 0; i < num_exprs; i++) {
    if (!exprs[i] || !exprs[i]->expr) return Fail(Status::Invalid("Expression cannot be null"));
    ex.push_back(exprs[i]->expr);
  }
  return Check(PrecompileFilterProject(schema->fields, condition->expr, ex, mode));
  });
}
int gdv_precompile_filter(const gdv_schema_t* schema, gdv_expression_t* condition) {
  return Guarded([&]() -> int {
  if (!schema || !condition || !condition->expr) return Fail(Status::Invalid("null argument"));
  return Check(PrecompileFilter(schema->fields, condition->expr));
  });
}

int gdv_compile_regex(const char* pattern, int64_t pattern_len, uint8_t* table) {
  return Guarded([&]() -> int {
  if (!pattern || pattern_len < 0 || !table) return Fail(Status::Invalid("null argument"));
  std::string bytes;
  Status st = CompileRegex(std::string(pattern, static_cast<size_t>(pattern_len)), &bytes);
  if (!st.ok()) return Fail(st);
  std::memcpy(table, bytes.data(), bytes.size());
  return 0;
  });
}

int gdv_compile_date_format(const char* pattern, int64_t pattern_len, uint8_t* ops, int64_t cap, int64_t* n) {
  return Guarded([&]() -> int {
  if (!pattern || pattern_len < 0 || !ops || !n) return Fail(Status::Invalid("null argument"));
  std::string bytes;
  Status st = CompileDateFormat(std::string(pattern, static_cast<size_t>(pattern_len)), &bytes);
  if (!st.ok()) return Fail(st);
  if (static_cast<int64_t>(bytes.size()) > cap) return Fail(Status::Invalid("ops buffer too small"));
  std::memcpy(ops, bytes.data(), bytes.size());
  *n = static_cast<int64_t>(bytes.size());
  return 0;
  });
}

// The part of a kernel's name that comes from the device function library: the hash of the library
// items `kernel_text` reaches (gdv_libtag.h).  library_source NULL = the embedded library.
char* gdv_kernel_library_tag(const char* library_source, const char* kernel_text) {
  if (kernel_text == nullptr) return nullptr;
  std::string tag;
  if (library_source == nullptr) {
    tag = LibraryIndex::Embedded().TagFor(kernel_text);
  } else {
    tag = LibraryIndex(library_source).TagFor(kernel_text);
  }
  return DupString(tag);
}
char* gdv_kernel_library_items(const char* library_source, const char* kernel_text) {
  if (kernel_text == nullptr) return nullptr;
  std::vector<std::string> names = library_source == nullptr
                                       ? LibraryIndex::Embedded().ReachedFrom(kernel_text)
                                       : LibraryIndex(library_source).ReachedFrom(kernel_text);
  std::string out;
  for (auto& n : names) out += n + "\n";
  return DupString(out);
}
const char* gdv_device_library_source(void) { return gdv_device_lib_src; }

}  // extern "C"
