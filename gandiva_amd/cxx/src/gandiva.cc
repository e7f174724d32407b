// libgandiva.so — the reference's public C++ API (namespace gandiva, signatures pinned by
// pyarrow/includes/libgandiva.pxd:27-298) implemented as a thin marshalling layer over the
// C ABI of libgandiva_amd.so (include/gandiva_amd.h).  Nothing is evaluated here: trees are
// forwarded to gdv_node_*, Evaluate hands raw Arrow buffer addresses to
// gdv_projector_evaluate / gdv_filter_evaluate, which launch the HIP kernels.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "arrow/api.h"
#include "arrow/device.h"
#include "gandiva/condition.h"
#include "gandiva/configuration.h"
#include "gandiva/expression_registry.h"
#include "gandiva/filter.h"
#include "gandiva/filter_project.h"
#include "gandiva/sharded.h"
#include "gandiva/function_signature.h"
#include "gandiva/host_memory.h"
#include "gandiva/node.h"
#include "gandiva/projector.h"
#include "gandiva/selection_vector.h"
#include "gandiva/tree_expr_builder.h"
#include "gandiva_amd.h"

namespace gandiva {

namespace {

Status LastError(int rc) {
  std::string msg = gdv_last_error();
  size_t colon = msg.find(": ");  // drop the code name: arrow::Status prints its own
  if (colon != std::string::npos) msg = msg.substr(colon + 2);
  return Status(static_cast<arrow::StatusCode>(rc), msg);
}
#define GDV_CXX_RETURN_NOT_OK(rc)            \
  do {                                       \
    int _rc = (rc);                          \
    if (_rc != GDV_OK) return LastError(_rc); \
  } while (0)

bool ToGdvType(const arrow::DataType& t, gdv_type_t* out) {
  out->id = static_cast<int32_t>(t.id());
  out->precision = 0;
  out->scale = 0;
  switch (t.id()) {
    case arrow::Type::BOOL: case arrow::Type::UINT8: case arrow::Type::INT8:
    case arrow::Type::UINT16: case arrow::Type::INT16: case arrow::Type::UINT32:
    case arrow::Type::INT32: case arrow::Type::UINT64: case arrow::Type::INT64:
    case arrow::Type::FLOAT: case arrow::Type::DOUBLE: case arrow::Type::STRING:
    case arrow::Type::BINARY: case arrow::Type::DATE32: case arrow::Type::DATE64:
      return true;
    case arrow::Type::TIMESTAMP:
      out->precision = static_cast<int32_t>(static_cast<const arrow::TimestampType&>(t).unit());
      return true;
    case arrow::Type::TIME32:
      out->precision = static_cast<int32_t>(static_cast<const arrow::Time32Type&>(t).unit());
      return true;
    case arrow::Type::TIME64:
      out->precision = static_cast<int32_t>(static_cast<const arrow::Time64Type&>(t).unit());
      return true;
    case arrow::Type::DECIMAL128: {
      auto& d = static_cast<const arrow::Decimal128Type&>(t);
      out->precision = d.precision();
      out->scale = d.scale();
      return true;
    }
    default:
      return false;
  }
}

DataTypePtr FromGdvType(gdv_type_t g) {
  switch (g.id) {
    case GDV_TYPE_BOOL: return arrow::boolean();
    case GDV_TYPE_UINT8: return arrow::uint8();
    case GDV_TYPE_INT8: return arrow::int8();
    case GDV_TYPE_UINT16: return arrow::uint16();
    case GDV_TYPE_INT16: return arrow::int16();
    case GDV_TYPE_UINT32: return arrow::uint32();
    case GDV_TYPE_INT32: return arrow::int32();
    case GDV_TYPE_UINT64: return arrow::uint64();
    case GDV_TYPE_INT64: return arrow::int64();
    case GDV_TYPE_FLOAT: return arrow::float32();
    case GDV_TYPE_DOUBLE: return arrow::float64();
    case GDV_TYPE_STRING: return arrow::utf8();
    case GDV_TYPE_BINARY: return arrow::binary();
    case GDV_TYPE_DATE32: return arrow::date32();
    case GDV_TYPE_DATE64: return arrow::date64();
    case GDV_TYPE_TIMESTAMP: return arrow::timestamp(static_cast<arrow::TimeUnit::type>(g.precision));
    case GDV_TYPE_TIME32: return arrow::time32(static_cast<arrow::TimeUnit::type>(g.precision));
    case GDV_TYPE_TIME64: return arrow::time64(static_cast<arrow::TimeUnit::type>(g.precision));
    case GDV_TYPE_DECIMAL128: return arrow::decimal128(g.precision, g.scale);
    default: return arrow::null();
  }
}

// An unsupported type still yields a node: the error surfaces at Make() as a validation
// error, which is where the reference reports unsupported signatures.
gdv_type_t GdvTypeOrNull(const DataTypePtr& t) {
  gdv_type_t g{0, 0, 0};
  if (t) ToGdvType(*t, &g);
  return g;
}

bool IsVarlen(const arrow::DataType& t) {
  return t.id() == arrow::Type::STRING || t.id() == arrow::Type::BINARY;
}

template <typename T>
NodePtr FixedLiteral(TreeExprBuilder*, DataTypePtr type, T value);

const void* BufAddr(const std::shared_ptr<arrow::Buffer>& b) {
  return b ? reinterpret_cast<const void*>(b->address()) : nullptr;
}
int64_t BufSize(const std::shared_ptr<arrow::Buffer>& b) { return b ? b->size() : 0; }

// Arrow array -> raw buffers; *device becomes true if any buffer is not CPU-accessible
Status ToColumn(const arrow::ArrayData& d, gdv_column_t* col, bool* device,
                std::shared_ptr<arrow::MemoryManager>* mm) {
  std::memset(col, 0, sizeof(*col));
  const bool varlen = IsVarlen(*d.type);
  const size_t need = varlen ? 3 : 2;
  if (d.buffers.size() < need) {
    // NullArray-like or unsupported layouts are only a problem if the column is referenced
    return Status::OK();
  }
  col->validity = BufAddr(d.buffers[0]);
  col->validity_size = BufSize(d.buffers[0]);
  if (varlen) {
    col->offsets = BufAddr(d.buffers[1]);
    col->offsets_size = BufSize(d.buffers[1]);
    col->data = BufAddr(d.buffers[2]);
    col->data_size = BufSize(d.buffers[2]);
  } else {
    col->data = BufAddr(d.buffers[1]);
    col->data_size = BufSize(d.buffers[1]);
  }
  col->offset = d.offset;
  for (auto& b : d.buffers) {
    if (b && !b->is_cpu()) {
      *device = true;
      if (mm && !*mm) *mm = b->memory_manager();
    }
  }
  return Status::OK();
}

Status MarshalBatch(const arrow::RecordBatch& batch, const SchemaPtr& schema,
                    std::vector<gdv_column_t>* cols, bool* device,
                    std::shared_ptr<arrow::MemoryManager>* mm) {
  if (!batch.schema()->Equals(*schema, /*check_metadata=*/false))
    return Status::Invalid("Schema in RecordBatch must match schema in Make()");
  if (batch.num_rows() == 0) return Status::Invalid("RecordBatch must be non-empty.");
  cols->resize(batch.num_columns());
  for (int i = 0; i < batch.num_columns(); i++)
    ARROW_RETURN_NOT_OK(ToColumn(*batch.column_data(i), &(*cols)[i], device, mm));
  return Status::OK();
}

// bytes the kernels may write behind `address()`: Arrow pads its own allocations to 64 bytes (capacity),
// which is what lets a validity bitmap in registered host memory be written in whole 8-byte words
int64_t Room(const arrow::Buffer& b) { return std::max(b.size(), b.capacity()); }

arrow::Result<std::shared_ptr<arrow::Buffer>> AllocOut(int64_t bytes, bool device,
                                                       arrow::MemoryPool* pool,
                                                       const std::shared_ptr<arrow::MemoryManager>& mm) {
  if (bytes < 8) bytes = 8;
  if (device) {
    if (!mm) return Status::Invalid("device-resident batch without a MemoryManager");
    ARROW_ASSIGN_OR_RAISE(auto b, mm->AllocateBuffer(bytes));
    return std::shared_ptr<arrow::Buffer>(std::move(b));
  }
  ARROW_ASSIGN_OR_RAISE(auto b, arrow::AllocateBuffer(bytes, pool ? pool : arrow::default_memory_pool()));
  std::memset(b->mutable_data(), 0, static_cast<size_t>(b->size()));
  return std::shared_ptr<arrow::Buffer>(std::move(b));
}

}  // namespace

// ------------------------------------------------------------------ Node / Expression

Status RegisterHostMemory(void* ptr, int64_t bytes) {
  GDV_CXX_RETURN_NOT_OK(gdv_host_register(ptr, bytes));
  return Status::OK();
}
Status UnregisterHostMemory(void* ptr) {
  GDV_CXX_RETURN_NOT_OK(gdv_host_unregister(ptr));
  return Status::OK();
}
int64_t HostStagedBytes() { return gdv_host_staged_bytes(); }

// ---- HostMemoryPool: power-of-two blocks carved from page-locked chunks
struct HostMemoryPool::Impl {
  struct Chunk { uint8_t* base; int64_t size, used; };
  std::mutex mu;
  int64_t chunk_bytes;
  std::vector<Chunk> chunks;
  std::map<int64_t, std::vector<uint8_t*>> free_blocks;  // block size -> blocks
  std::map<uint8_t*, int64_t> large;                      // blocks with a page-locked allocation of their own
  int64_t allocated = 0, peak = 0, total = 0, count = 0;
  static int64_t BlockSize(int64_t size, int64_t alignment) {
    int64_t b = 64;
    while (b < size || b < alignment) b <<= 1;
    return b;
  }
};
namespace {
alignas(64) uint8_t g_zero_size_area[64];
}
HostMemoryPool::HostMemoryPool(int64_t chunk_bytes) : impl_(new Impl) {
  impl_->chunk_bytes = std::max<int64_t>(chunk_bytes, int64_t{1} << 20);
}
HostMemoryPool::~HostMemoryPool() {
  for (auto& c : impl_->chunks) gdv_host_free(c.base);
  for (auto& l : impl_->large) gdv_host_free(l.first);
}
Status HostMemoryPool::Allocate(int64_t size, int64_t alignment, uint8_t** out) {
  if (size < 0) return Status::Invalid("negative malloc size");
  if (size == 0) { *out = g_zero_size_area; return Status::OK(); }
  const int64_t block = Impl::BlockSize(size, alignment);
  std::lock_guard<std::mutex> g(impl_->mu);
  uint8_t* p = nullptr;
  auto fl = impl_->free_blocks.find(block);
  if (fl != impl_->free_blocks.end() && !fl->second.empty()) {
    p = fl->second.back();
    fl->second.pop_back();
  } else if (block > impl_->chunk_bytes / 2) {
    void* q = nullptr;
    GDV_CXX_RETURN_NOT_OK(gdv_host_alloc(block, &q));
    p = static_cast<uint8_t*>(q);
    impl_->large[p] = block;
  } else {
    for (auto& c : impl_->chunks) {
      const int64_t at = (c.used + block - 1) / block * block;  // (chunks are page aligned, blocks powers of two)
      if (at + block <= c.size) { p = c.base + at; c.used = at + block; break; }
    }
    if (p == nullptr) {
      void* q = nullptr;
      GDV_CXX_RETURN_NOT_OK(gdv_host_alloc(impl_->chunk_bytes, &q));
      impl_->chunks.push_back(Impl::Chunk{static_cast<uint8_t*>(q), impl_->chunk_bytes, block});
      p = static_cast<uint8_t*>(q);
    }
  }
  impl_->allocated += size;
  impl_->peak = std::max(impl_->peak, impl_->allocated);
  impl_->total += size;
  impl_->count++;
  *out = p;
  return Status::OK();
}
void HostMemoryPool::Free(uint8_t* buffer, int64_t size, int64_t alignment) {
  if (buffer == nullptr || size == 0 || buffer == g_zero_size_area) return;
  const int64_t block = Impl::BlockSize(size, alignment);
  std::lock_guard<std::mutex> g(impl_->mu);
  impl_->allocated -= size;
  impl_->free_blocks[block].push_back(buffer);   // (blocks with an allocation of their own are recycled too)
}
Status HostMemoryPool::Reallocate(int64_t old_size, int64_t new_size, int64_t alignment, uint8_t** ptr) {
  if (new_size < 0) return Status::Invalid("negative realloc size");
  if (old_size > 0 && new_size > 0 && Impl::BlockSize(old_size, alignment) == Impl::BlockSize(new_size, alignment)) {
    std::lock_guard<std::mutex> g(impl_->mu);
    impl_->allocated += new_size - old_size;
    impl_->peak = std::max(impl_->peak, impl_->allocated);
    return Status::OK();  // same block
  }
  uint8_t* fresh = nullptr;
  ARROW_RETURN_NOT_OK(Allocate(new_size, alignment, &fresh));
  if (old_size > 0 && new_size > 0) std::memcpy(fresh, *ptr, static_cast<size_t>(std::min(old_size, new_size)));
  Free(*ptr, old_size, alignment);
  *ptr = fresh;
  return Status::OK();
}
int64_t HostMemoryPool::bytes_allocated() const { std::lock_guard<std::mutex> g(impl_->mu); return impl_->allocated; }
int64_t HostMemoryPool::max_memory() const { std::lock_guard<std::mutex> g(impl_->mu); return impl_->peak; }
int64_t HostMemoryPool::total_bytes_allocated() const { std::lock_guard<std::mutex> g(impl_->mu); return impl_->total; }
int64_t HostMemoryPool::num_allocations() const { std::lock_guard<std::mutex> g(impl_->mu); return impl_->count; }

Node::~Node() { gdv_node_free(handle_); }
std::string Node::ToString() const {
  char* s = gdv_node_to_string(handle_);
  std::string r = s ? s : "";
  gdv_free_string(s);
  return r;
}
Expression::~Expression() { gdv_expression_free(handle_); }

// ------------------------------------------------------------------ TreeExprBuilder

#define GDV_LITERAL(CTYPE, ARROW_TYPE)                                         \
  NodePtr TreeExprBuilder::MakeLiteral(CTYPE value) {                          \
    auto t = ARROW_TYPE;                                                       \
    gdv_node* h = gdv_node_literal(GdvTypeOrNull(t), &value, 0);               \
    return h ? NodePtr(new Node(h, t)) : nullptr;                              \
  }
NodePtr TreeExprBuilder::MakeLiteral(bool value) {
  uint8_t v = value ? 1 : 0;
  gdv_node* h = gdv_node_literal(GdvTypeOrNull(arrow::boolean()), &v, 0);
  return h ? NodePtr(new Node(h, arrow::boolean())) : nullptr;
}
GDV_LITERAL(uint8_t, arrow::uint8())
GDV_LITERAL(uint16_t, arrow::uint16())
GDV_LITERAL(uint32_t, arrow::uint32())
GDV_LITERAL(uint64_t, arrow::uint64())
GDV_LITERAL(int8_t, arrow::int8())
GDV_LITERAL(int16_t, arrow::int16())
GDV_LITERAL(int32_t, arrow::int32())
GDV_LITERAL(int64_t, arrow::int64())
GDV_LITERAL(float, arrow::float32())
GDV_LITERAL(double, arrow::float64())

NodePtr TreeExprBuilder::MakeStringLiteral(const std::string& value) {
  gdv_node* h = gdv_node_literal_bytes(GdvTypeOrNull(arrow::utf8()), value.data(),
                                       static_cast<int64_t>(value.size()), 0);
  return h ? NodePtr(new Node(h, arrow::utf8())) : nullptr;
}
NodePtr TreeExprBuilder::MakeBinaryLiteral(const std::string& value) {
  gdv_node* h = gdv_node_literal_bytes(GdvTypeOrNull(arrow::binary()), value.data(),
                                       static_cast<int64_t>(value.size()), 0);
  return h ? NodePtr(new Node(h, arrow::binary())) : nullptr;
}
NodePtr TreeExprBuilder::MakeDecimalLiteral(int64_t high, uint64_t low, int32_t precision,
                                            int32_t scale) {
  auto t = arrow::decimal128(precision, scale);
  uint64_t words[2] = {low, static_cast<uint64_t>(high)};
  gdv_node* h = gdv_node_literal(GdvTypeOrNull(t), words, 0);
  return h ? NodePtr(new Node(h, t)) : nullptr;
}
NodePtr TreeExprBuilder::MakeLiteral(const DecimalScalar128& value) {
  return MakeDecimalLiteral(value.value().high_bits(), value.value().low_bits(), value.precision(), value.scale());
}
NodePtr TreeExprBuilder::MakeNull(DataTypePtr data_type) {
  if (!data_type) return nullptr;
  gdv_type_t g = GdvTypeOrNull(data_type);
  gdv_node* h = IsVarlen(*data_type) ? gdv_node_literal_bytes(g, nullptr, 0, 1) : gdv_node_literal(g, nullptr, 1);
  return h ? NodePtr(new Node(h, data_type)) : nullptr;
}
NodePtr TreeExprBuilder::MakeField(FieldPtr field) {
  if (!field) return nullptr;
  gdv_node* h = gdv_node_field(field->name().c_str(), GdvTypeOrNull(field->type()));
  return h ? NodePtr(new Node(h, field->type())) : nullptr;
}
static bool Handles(const NodeVector& v, std::vector<gdv_node*>* out) {
  for (auto& n : v) {
    if (!n) return false;
    out->push_back(n->handle());
  }
  return true;
}
NodePtr TreeExprBuilder::MakeFunction(const std::string& name, const NodeVector& params,
                                      DataTypePtr return_type) {
  std::vector<gdv_node*> hs;
  if (!return_type || !Handles(params, &hs)) return nullptr;
  gdv_node* h = gdv_node_function(name.c_str(), hs.data(), static_cast<int>(hs.size()),
                                  GdvTypeOrNull(return_type));
  return h ? NodePtr(new Node(h, return_type)) : nullptr;
}
NodePtr TreeExprBuilder::MakeIf(NodePtr condition, NodePtr then_node, NodePtr else_node,
                                DataTypePtr result_type) {
  if (!condition || !then_node || !else_node || !result_type) return nullptr;
  gdv_node* h = gdv_node_if(condition->handle(), then_node->handle(), else_node->handle(),
                            GdvTypeOrNull(result_type));
  return h ? NodePtr(new Node(h, result_type)) : nullptr;
}
NodePtr TreeExprBuilder::MakeAnd(const NodeVector& children) {
  std::vector<gdv_node*> hs;
  if (!Handles(children, &hs)) return nullptr;
  gdv_node* h = gdv_node_and(hs.data(), static_cast<int>(hs.size()));
  return h ? NodePtr(new Node(h, arrow::boolean())) : nullptr;
}
NodePtr TreeExprBuilder::MakeOr(const NodeVector& children) {
  std::vector<gdv_node*> hs;
  if (!Handles(children, &hs)) return nullptr;
  gdv_node* h = gdv_node_or(hs.data(), static_cast<int>(hs.size()));
  return h ? NodePtr(new Node(h, arrow::boolean())) : nullptr;
}
ExpressionPtr TreeExprBuilder::MakeExpression(NodePtr root_node, FieldPtr result_field) {
  if (!root_node || !result_field) return nullptr;
  gdv_expression* h = gdv_expression_new(root_node->handle(), result_field->name().c_str(),
                                         GdvTypeOrNull(result_field->type()));
  return h ? ExpressionPtr(new Expression(h, root_node, result_field)) : nullptr;
}
ExpressionPtr TreeExprBuilder::MakeExpression(const std::string& function,
                                              const FieldVector& in_fields, FieldPtr out_field) {
  if (!out_field) return nullptr;
  NodeVector args;
  for (auto& f : in_fields) args.push_back(MakeField(f));
  return MakeExpression(MakeFunction(function, args, out_field->type()), out_field);
}
ConditionPtr TreeExprBuilder::MakeCondition(NodePtr root_node) {
  if (!root_node) return nullptr;
  gdv_expression* h = gdv_condition_new(root_node->handle());
  return h ? ConditionPtr(new Condition(h, root_node)) : nullptr;
}
ConditionPtr TreeExprBuilder::MakeCondition(const std::string& function,
                                            const FieldVector& in_fields) {
  NodeVector args;
  for (auto& f : in_fields) args.push_back(MakeField(f));
  return MakeCondition(MakeFunction(function, args, arrow::boolean()));
}

template <typename T>
static NodePtr MakeInFixed(NodePtr node, const std::unordered_set<T>& constants, DataTypePtr type,
                           NodePtr (*wrap)(gdv_node*, DataTypePtr)) {
  if (!node) return nullptr;
  std::vector<T> vals(constants.begin(), constants.end());
  gdv_node* h = gdv_node_in(node->handle(), GdvTypeOrNull(type), vals.data(), static_cast<int>(vals.size()));
  return wrap(h, arrow::boolean());
}
#define GDV_IN_FIXED(NAME, CTYPE, ARROW_TYPE)                                                        \
  NodePtr TreeExprBuilder::NAME(NodePtr node, const std::unordered_set<CTYPE>& constants) {          \
    if (!node) return nullptr;                                                                       \
    std::vector<CTYPE> vals(constants.begin(), constants.end());                                     \
    gdv_node* h = gdv_node_in(node->handle(), GdvTypeOrNull(ARROW_TYPE), vals.data(),                \
                              static_cast<int>(vals.size()));                                        \
    return h ? NodePtr(new Node(h, arrow::boolean())) : nullptr;                                     \
  }
GDV_IN_FIXED(MakeInExpressionInt32, int32_t, arrow::int32())
GDV_IN_FIXED(MakeInExpressionInt64, int64_t, arrow::int64())
GDV_IN_FIXED(MakeInExpressionDate32, int32_t, arrow::date32())
GDV_IN_FIXED(MakeInExpressionDate64, int64_t, arrow::date64())
GDV_IN_FIXED(MakeInExpressionTime32, int32_t, arrow::time32(arrow::TimeUnit::MILLI))
GDV_IN_FIXED(MakeInExpressionTime64, int64_t, arrow::time64(arrow::TimeUnit::MICRO))
GDV_IN_FIXED(MakeInExpressionTimeStamp, int64_t, arrow::timestamp(arrow::TimeUnit::MILLI))
GDV_IN_FIXED(MakeInExpressionFloat, float, arrow::float32())
GDV_IN_FIXED(MakeInExpressionDouble, double, arrow::float64())
NodePtr TreeExprBuilder::MakeInExpressionDecimal(NodePtr node, std::unordered_set<DecimalScalar128>& constants,
                                                 int32_t precision, int32_t scale) {
  if (!node) return nullptr;
  std::vector<uint64_t> words;  // (low, high) per value: the 16-byte little-endian image
  for (auto& c : constants) {
    words.push_back(c.value().low_bits());
    words.push_back(static_cast<uint64_t>(c.value().high_bits()));
  }
  gdv_node* h = gdv_node_in(node->handle(), GdvTypeOrNull(arrow::decimal128(precision, scale)), words.data(),
                            static_cast<int>(constants.size()));
  return h ? NodePtr(new Node(h, arrow::boolean())) : nullptr;
}

#define GDV_IN_BYTES(NAME, ARROW_TYPE)                                                               \
  NodePtr TreeExprBuilder::NAME(NodePtr node, const std::unordered_set<std::string>& constants) {    \
    if (!node) return nullptr;                                                                       \
    std::vector<const char*> ptrs;                                                                   \
    std::vector<int64_t> lens;                                                                       \
    for (auto& s : constants) { ptrs.push_back(s.data()); lens.push_back(static_cast<int64_t>(s.size())); } \
    gdv_node* h = gdv_node_in_bytes(node->handle(), GdvTypeOrNull(ARROW_TYPE), ptrs.data(),          \
                                    lens.data(), static_cast<int>(ptrs.size()));                     \
    return h ? NodePtr(new Node(h, arrow::boolean())) : nullptr;                                     \
  }
GDV_IN_BYTES(MakeInExpressionString, arrow::utf8())
GDV_IN_BYTES(MakeInExpressionBinary, arrow::binary())

// ------------------------------------------------------------------ SelectionVector

static Status MakeSel(SelectionVector::Mode mode, int bytes, int64_t max_slots, arrow::MemoryPool* pool,
                      std::shared_ptr<SelectionVector>* out) {
  if (!out) return Status::Invalid("selection vector output cannot be null");
  if (max_slots < 0) return Status::Invalid("max_slots cannot be negative");
  ARROW_ASSIGN_OR_RAISE(auto buf, arrow::AllocateBuffer(std::max<int64_t>(max_slots, 1) * bytes,
                                                        pool ? pool : arrow::default_memory_pool()));
  return SelectionVector::Make(mode, max_slots, std::shared_ptr<arrow::Buffer>(std::move(buf)), out);
}
Status SelectionVector::Make(Mode mode, int64_t max_slots, std::shared_ptr<arrow::Buffer> buffer,
                             std::shared_ptr<SelectionVector>* out) {
  if (!out || !buffer) return Status::Invalid("selection vector buffer cannot be null");
  const int w = mode == MODE_UINT16 ? 2 : mode == MODE_UINT32 ? 4 : 8;
  if (mode == MODE_NONE) return Status::Invalid("selection vector mode cannot be NONE");
  if (buffer->size() < max_slots * w) return Status::Invalid("buffer too small for max_slots");
  if (mode == MODE_UINT16 && max_slots > 65536)
    return Status::Invalid("max_slots cannot exceed 65536 for a 16-bit selection vector");
  *out = std::shared_ptr<SelectionVector>(new SelectionVector(mode, max_slots, std::move(buffer)));
  return Status::OK();
}
Status SelectionVector::MakeInt16(int64_t n, arrow::MemoryPool* pool, std::shared_ptr<SelectionVector>* out) {
  return MakeSel(MODE_UINT16, 2, n, pool, out);
}
Status SelectionVector::MakeInt32(int64_t n, arrow::MemoryPool* pool, std::shared_ptr<SelectionVector>* out) {
  return MakeSel(MODE_UINT32, 4, n, pool, out);
}
Status SelectionVector::MakeInt64(int64_t n, arrow::MemoryPool* pool, std::shared_ptr<SelectionVector>* out) {
  return MakeSel(MODE_UINT64, 8, n, pool, out);
}
uint64_t SelectionVector::GetIndex(int64_t i) const {
  const uint8_t* p = buffer_->data();
  switch (mode_) {
    case MODE_UINT16: return reinterpret_cast<const uint16_t*>(p)[i];
    case MODE_UINT32: return reinterpret_cast<const uint32_t*>(p)[i];
    default: return reinterpret_cast<const uint64_t*>(p)[i];
  }
}
void SelectionVector::SetIndex(int64_t i, uint64_t v) {
  uint8_t* p = buffer_->mutable_data();
  switch (mode_) {
    case MODE_UINT16: reinterpret_cast<uint16_t*>(p)[i] = static_cast<uint16_t>(v); break;
    case MODE_UINT32: reinterpret_cast<uint32_t*>(p)[i] = static_cast<uint32_t>(v); break;
    default: reinterpret_cast<uint64_t*>(p)[i] = v; break;
  }
}
ArrayPtr SelectionVector::ToArray() const {
  DataTypePtr t = mode_ == MODE_UINT16 ? arrow::uint16() : mode_ == MODE_UINT32 ? arrow::uint32() : arrow::uint64();
  auto data = arrow::ArrayData::Make(t, num_slots_, {nullptr, buffer_}, /*null_count=*/0);
  return arrow::MakeArray(data);
}

// ------------------------------------------------------------------ Projector

static gdv_schema_t* MakeSchema(const SchemaPtr& schema, Status* st) {
  gdv_schema_t* h = gdv_schema_new();
  for (auto& f : schema->fields()) {
    gdv_type_t g;
    if (!ToGdvType(*f->type(), &g)) {
      // unsupported column types are fine as long as no expression references them
      g = gdv_type_t{GDV_TYPE_BINARY, 0, 0};
    }
    int rc = gdv_schema_add_field(h, f->name().c_str(), g, f->nullable());
    if (rc != GDV_OK) {
      *st = LastError(rc);
      gdv_schema_free(h);
      return nullptr;
    }
  }
  return h;
}

Projector::~Projector() { gdv_projector_free(handle_); }

Status Projector::Make(SchemaPtr schema, const ExpressionVector& exprs,
                       std::shared_ptr<Projector>* projector) {
  return Make(schema, exprs, SelectionVector::MODE_NONE, ConfigurationBuilder::DefaultConfiguration(), projector);
}
Status Projector::Make(SchemaPtr schema, const ExpressionVector& exprs,
                       std::shared_ptr<Configuration> configuration,
                       std::shared_ptr<Projector>* projector) {
  return Make(schema, exprs, SelectionVector::MODE_NONE, configuration, projector);
}
Status Projector::Make(SchemaPtr schema, const ExpressionVector& exprs,
                       SelectionVector::Mode mode, std::shared_ptr<Configuration> configuration,
                       std::shared_ptr<Projector>* projector) {
  if (!schema) return Status::Invalid("Schema cannot be null");
  if (exprs.empty()) return Status::Invalid("Expressions cannot be empty");
  if (!configuration) return Status::Invalid("Configuration cannot be null");
  if (!projector) return Status::Invalid("Projector output cannot be null");
  std::vector<gdv_expression*> hs;
  FieldVector outs;
  for (auto& e : exprs) {
    if (!e) return Status::Invalid("Expression cannot be null");
    hs.push_back(e->handle());
    outs.push_back(e->result());
  }
  Status st;
  gdv_schema_t* sh = MakeSchema(schema, &st);
  if (!sh) return st;
  gdv_config_t cfg{configuration->optimize(), configuration->dump_ir()};
  gdv_projector* h = nullptr;
  int rc = gdv_projector_make(sh, hs.data(), static_cast<int>(hs.size()), static_cast<int>(mode), &cfg, &h);
  gdv_schema_free(sh);
  GDV_CXX_RETURN_NOT_OK(rc);
  *projector = std::shared_ptr<Projector>(new Projector(h, schema, outs, mode));
  return Status::OK();
}

Status Projector::Evaluate(const arrow::RecordBatch& batch, arrow::MemoryPool* pool,
                           ArrayVector* output) const {
  return Evaluate(batch, nullptr, pool, output);
}

Status Projector::Evaluate(const arrow::RecordBatch& batch, const SelectionVector* selection,
                           arrow::MemoryPool* pool, ArrayVector* output) const {
  if (!output) return Status::Invalid("Output array vector cannot be null");
  std::vector<gdv_column_t> cols;
  bool device = false;
  std::shared_ptr<arrow::MemoryManager> mm;
  ARROW_RETURN_NOT_OK(MarshalBatch(batch, schema_, &cols, &device, &mm));
  const int64_t out_rows = selection ? selection->GetNumSlots() : batch.num_rows();
  gdv_selection_t sel;
  if (selection) {
    sel.mode = static_cast<int32_t>(selection->GetMode());
    sel.indices = reinterpret_cast<const void*>(selection->GetBuffer().address());
    sel.num_slots = selection->GetNumSlots();
    if (!selection->GetBuffer().is_cpu()) device = true;
  }
  const int n_out = static_cast<int>(output_fields_.size());
  const int mem = device ? GDV_MEM_DEVICE : GDV_MEM_HOST;
  std::vector<gdv_out_column_t> outs(n_out);
  std::vector<std::shared_ptr<arrow::Buffer>> vbuf(n_out), dbuf(n_out), obuf(n_out);
  int64_t varlen_guess = 64;
  for (auto& c : cols) if (c.offsets) varlen_guess += c.data_size;
  for (int e = 0; e < n_out; e++) {
    const bool varlen = IsVarlen(*output_fields_[e]->type());
    int64_t vbytes = 0, dbytes = 0;
    GDV_CXX_RETURN_NOT_OK(gdv_projector_output_sizes(handle_, e, out_rows, mem, &vbytes, &dbytes));
    if (varlen && dbytes == 0) dbytes = varlen_guess;  // (non-zero: what earlier batches produced per row)
    ARROW_ASSIGN_OR_RAISE(vbuf[e], AllocOut(vbytes, device, pool, mm));
    ARROW_ASSIGN_OR_RAISE(dbuf[e], AllocOut(dbytes, device, pool, mm));
    std::memset(&outs[e], 0, sizeof(outs[e]));
    outs[e].validity = reinterpret_cast<void*>(vbuf[e]->address());
    outs[e].validity_size = Room(*vbuf[e]);
    outs[e].data = reinterpret_cast<void*>(dbuf[e]->address());
    outs[e].data_size = varlen ? dbuf[e]->size() : Room(*dbuf[e]);
    if (varlen) {
      ARROW_ASSIGN_OR_RAISE(obuf[e], AllocOut((out_rows + 1) * 4, device, pool, mm));
      outs[e].offsets = reinterpret_cast<void*>(obuf[e]->address());
      outs[e].offsets_size = obuf[e]->size();
    }
  }
  for (int attempt = 0;; attempt++) {
    std::vector<int64_t> caps(n_out);
    for (int e = 0; e < n_out; e++) caps[e] = outs[e].data_size;
    int rc = gdv_projector_evaluate(handle_, batch.num_rows(), cols.data(), static_cast<int>(cols.size()),
                                    selection ? &sel : nullptr, outs.data(), n_out, mem, nullptr, 0);
    bool grown = false;
    if (rc == GDV_INVALID && attempt == 0) {
      // a var-len byte buffer was too small: data_size now carries the bytes needed
      for (int e = 0; e < n_out; e++) {
        if (!obuf[e]) continue;
        if (outs[e].data_size > caps[e]) {
          ARROW_ASSIGN_OR_RAISE(dbuf[e], AllocOut(outs[e].data_size, device, pool, mm));
          outs[e].data = reinterpret_cast<void*>(dbuf[e]->address());
          outs[e].data_size = dbuf[e]->size();
          grown = true;
        } else {
          outs[e].data_size = caps[e];
        }
      }
    }
    if (grown) continue;
    GDV_CXX_RETURN_NOT_OK(rc);
    break;
  }
  for (int e = 0; e < n_out; e++) {
    std::vector<std::shared_ptr<arrow::Buffer>> bufs;
    if (obuf[e]) {
      bufs = {vbuf[e], obuf[e], arrow::SliceBuffer(dbuf[e], 0, outs[e].data_size)};
    } else {
      bufs = {vbuf[e], dbuf[e]};
    }
    output->push_back(arrow::MakeArray(arrow::ArrayData::Make(output_fields_[e]->type(), out_rows, std::move(bufs))));
  }
  return Status::OK();
}

Status Projector::Evaluate(const arrow::RecordBatch& batch, const ArrayDataVector& output) const {
  return Evaluate(batch, nullptr, output);
}

Status Projector::Evaluate(const arrow::RecordBatch& batch, const SelectionVector* selection,
                           const ArrayDataVector& output) const {
  const int n_out = static_cast<int>(output_fields_.size());
  if (static_cast<int>(output.size()) != n_out)
    return Status::Invalid("number of buffers for output_data_vecs is ", output.size(), ", expected ", n_out);
  std::vector<gdv_column_t> cols;
  bool device = false;
  std::shared_ptr<arrow::MemoryManager> mm;
  ARROW_RETURN_NOT_OK(MarshalBatch(batch, schema_, &cols, &device, &mm));
  const int64_t out_rows = selection ? selection->GetNumSlots() : batch.num_rows();
  gdv_selection_t sel;
  if (selection) {
    sel.mode = static_cast<int32_t>(selection->GetMode());
    sel.indices = reinterpret_cast<const void*>(selection->GetBuffer().address());
    sel.num_slots = selection->GetNumSlots();
    if (!selection->GetBuffer().is_cpu()) device = true;
  }
  const int mem = device ? GDV_MEM_DEVICE : GDV_MEM_HOST;
  std::vector<gdv_out_column_t> outs(n_out);
  for (int e = 0; e < n_out; e++) {
    const ArrayDataPtr& d = output[e];
    if (!d) return Status::Invalid("output array data ", e, " cannot be null");
    if (!d->type || !d->type->Equals(*output_fields_[e]->type()))
      return Status::Invalid("output array data ", e, " has type ", d->type ? d->type->ToString() : "null",
                             ", the expression returns ", output_fields_[e]->type()->ToString());
    if (d->length < out_rows)
      return Status::Invalid("output array data ", e, " holds ", d->length, " rows, ", out_rows, " needed");
    if (d->offset != 0) return Status::Invalid("output array data ", e, " must have offset 0");
    const bool varlen = IsVarlen(*d->type);
    const size_t need = varlen ? 3 : 2;
    if (d->buffers.size() < need) return Status::Invalid("output array data ", e, " needs ", need, " buffers");
    for (size_t b = 0; b < need; b++) {
      if (!d->buffers[b]) return Status::Invalid("output array data ", e, ": buffer ", b, " cannot be null");
      if (!d->buffers[b]->is_mutable()) return Status::Invalid("output array data ", e, ": buffer ", b, " is not mutable");
      if (d->buffers[b]->is_cpu() == device)
        return Status::Invalid("output array data ", e, ": buffer ", b, " does not live in the batch's memory domain");
    }
    int64_t vbytes = 0, dbytes = 0;
    GDV_CXX_RETURN_NOT_OK(gdv_projector_output_sizes(handle_, e, out_rows, mem, &vbytes, &dbytes));
    std::memset(&outs[e], 0, sizeof(outs[e]));
    outs[e].validity = reinterpret_cast<void*>(d->buffers[0]->address());
    outs[e].validity_size = Room(*d->buffers[0]);
    if (outs[e].validity_size < vbytes)
      return Status::Invalid("output array data ", e, ": validity buffer of ", outs[e].validity_size, " bytes, ", vbytes, " needed");
    if (varlen) {
      outs[e].offsets = reinterpret_cast<void*>(d->buffers[1]->address());
      outs[e].offsets_size = d->buffers[1]->size();
      outs[e].data = reinterpret_cast<void*>(d->buffers[2]->address());
      outs[e].data_size = d->buffers[2]->size();
    } else {
      outs[e].data = reinterpret_cast<void*>(d->buffers[1]->address());
      outs[e].data_size = d->buffers[1]->size();
      if (outs[e].data_size < dbytes)
        return Status::Invalid("output array data ", e, ": data buffer of ", outs[e].data_size, " bytes, ", dbytes, " needed");
    }
  }
  std::vector<int64_t> caps(n_out);
  for (int e = 0; e < n_out; e++) caps[e] = outs[e].data_size;
  int rc = gdv_projector_evaluate(handle_, batch.num_rows(), cols.data(), static_cast<int>(cols.size()),
                                  selection ? &sel : nullptr, outs.data(), n_out, mem, nullptr, 0);
  if (rc == GDV_INVALID) {
    for (int e = 0; e < n_out; e++)
      if (outs[e].offsets != nullptr && outs[e].data_size > caps[e])
        return Status::Invalid("output array data ", e, ": var-len data buffer of ", caps[e], " bytes, ", outs[e].data_size,
                               " needed");
  }
  GDV_CXX_RETURN_NOT_OK(rc);
  for (int e = 0; e < n_out; e++) {
    output[e]->length = out_rows;
    output[e]->null_count = arrow::kUnknownNullCount;
  }
  return Status::OK();
}

std::string Projector::DumpIR() {
  char* s = gdv_projector_dump_ir(handle_);
  std::string r = s ? s : "";
  gdv_free_string(s);
  return r;
}

// ------------------------------------------------------------------ Filter

Filter::~Filter() { gdv_filter_free(handle_); }

Status Filter::Make(SchemaPtr schema, ConditionPtr condition, std::shared_ptr<Filter>* filter) {
  return Make(schema, condition, ConfigurationBuilder::DefaultConfiguration(), filter);
}
Status Filter::Make(SchemaPtr schema, ConditionPtr condition,
                    std::shared_ptr<Configuration> configuration, std::shared_ptr<Filter>* filter) {
  if (!schema) return Status::Invalid("Schema cannot be null");
  if (!condition) return Status::Invalid("Condition cannot be null");
  if (!configuration) return Status::Invalid("Configuration cannot be null");
  if (!filter) return Status::Invalid("Filter output cannot be null");
  Status st;
  gdv_schema_t* sh = MakeSchema(schema, &st);
  if (!sh) return st;
  gdv_config_t cfg{configuration->optimize(), configuration->dump_ir()};
  gdv_filter* h = nullptr;
  int rc = gdv_filter_make(sh, condition->handle(), &cfg, &h);
  gdv_schema_free(sh);
  GDV_CXX_RETURN_NOT_OK(rc);
  *filter = std::shared_ptr<Filter>(new Filter(h, schema));
  return Status::OK();
}

Status Filter::Evaluate(const arrow::RecordBatch& batch, std::shared_ptr<SelectionVector> out) {
  if (!out) return Status::Invalid("Selection vector cannot be null");
  std::vector<gdv_column_t> cols;
  bool device = false;
  std::shared_ptr<arrow::MemoryManager> mm;
  ARROW_RETURN_NOT_OK(MarshalBatch(batch, schema_, &cols, &device, &mm));
  if (out->GetMaxSlots() < batch.num_rows())
    return Status::Invalid("Selection vector too small: max slots ", out->GetMaxSlots(),
                           " < number of rows ", batch.num_rows());
  const bool out_device = !out->GetBuffer().is_cpu();
  if (out_device != device)
    return Status::Invalid("batch and selection vector must live in the same memory domain");
  int64_t count = 0;
  GDV_CXX_RETURN_NOT_OK(gdv_filter_evaluate(
      handle_, batch.num_rows(), cols.data(), static_cast<int>(cols.size()), static_cast<int>(out->GetMode()),
      reinterpret_cast<void*>(out->GetBuffer().address()), out->GetMaxSlots(), &count,
      device ? GDV_MEM_DEVICE : GDV_MEM_HOST, nullptr));
  out->SetNumSlots(count);
  return Status::OK();
}

std::string Filter::DumpIR() {
  char* s = gdv_filter_dump_ir(handle_);
  std::string r = s ? s : "";
  gdv_free_string(s);
  return r;
}

// ------------------------------------------------------------------ one call, all GPUs (round 6)

void ShardBounds(int64_t num_rows, int num_shards, int shard, int64_t* lo, int64_t* hi) {
  if (gdv_shard_bounds(num_rows, num_shards, shard, lo, hi) != GDV_OK) *lo = *hi = 0;
}
int DeviceCount() { return gdv_device_count(); }

static std::vector<int> AllDevicesIfEmpty(std::vector<int> devices) {
  if (devices.empty())
    for (int d = 0; d < std::max(gdv_device_count(), 1); d++) devices.push_back(d);
  return devices;
}

Status ShardedProjector::Make(SchemaPtr schema, const ExpressionVector& exprs, std::vector<int> devices,
                              std::shared_ptr<Configuration> configuration, std::shared_ptr<ShardedProjector>* out) {
  if (!out) return Status::Invalid("ShardedProjector output cannot be null");
  std::shared_ptr<ShardedProjector> sp(new ShardedProjector());
  ARROW_RETURN_NOT_OK(Projector::Make(schema, exprs, SelectionVector::MODE_NONE, configuration, &sp->projector_));
  sp->devices_ = AllDevicesIfEmpty(std::move(devices));
  *out = sp;
  return Status::OK();
}

Status ShardedProjector::Evaluate(const arrow::RecordBatch& batch, arrow::MemoryPool* pool, ArrayVector* output) const {
  if (!output) return Status::Invalid("Output array vector cannot be null");
  const Projector& p = *projector_;
  std::vector<gdv_column_t> cols;
  bool device = false;
  std::shared_ptr<arrow::MemoryManager> mm;
  ARROW_RETURN_NOT_OK(MarshalBatch(batch, p.schema_, &cols, &device, &mm));
  if (device) return Status::Invalid("a device-resident batch is already on one GPU: pass one batch per device (the shards overload)");
  for (auto& f : p.output_fields_)
    if (IsVarlen(*f->type())) return p.Evaluate(batch, pool, output);  // byte positions depend on the rows before: one device
  const int n_out = static_cast<int>(p.output_fields_.size());
  const int64_t rows = batch.num_rows();
  std::vector<gdv_out_column_t> outs(n_out);
  std::vector<std::shared_ptr<arrow::Buffer>> vbuf(n_out), dbuf(n_out);
  for (int e = 0; e < n_out; e++) {
    int64_t vbytes = 0, dbytes = 0;
    GDV_CXX_RETURN_NOT_OK(gdv_projector_output_sizes(p.handle_, e, rows, GDV_MEM_HOST, &vbytes, &dbytes));
    ARROW_ASSIGN_OR_RAISE(vbuf[e], AllocOut(vbytes, false, pool, mm));
    ARROW_ASSIGN_OR_RAISE(dbuf[e], AllocOut(dbytes, false, pool, mm));
    std::memset(&outs[e], 0, sizeof(outs[e]));
    outs[e].validity = reinterpret_cast<void*>(vbuf[e]->address());
    outs[e].validity_size = Room(*vbuf[e]);
    outs[e].data = reinterpret_cast<void*>(dbuf[e]->address());
    outs[e].data_size = Room(*dbuf[e]);
  }
  std::vector<int32_t> devs(devices_.begin(), devices_.end());
  GDV_CXX_RETURN_NOT_OK(gdv_projector_evaluate_host_sharded(p.handle_, rows, cols.data(), static_cast<int>(cols.size()), outs.data(),
                                                            n_out, devs.data(), static_cast<int>(devs.size())));
  for (int e = 0; e < n_out; e++)
    output->push_back(arrow::MakeArray(arrow::ArrayData::Make(p.output_fields_[e]->type(), rows, {vbuf[e], dbuf[e]})));
  return Status::OK();
}

Status ShardedProjector::Evaluate(const std::vector<std::shared_ptr<arrow::RecordBatch>>& shards,
                                  std::vector<ArrayVector>* outputs) const {
  if (!outputs) return Status::Invalid("Output vector cannot be null");
  const Projector& p = *projector_;
  const int n = static_cast<int>(shards.size());
  if (n < 1 || n > static_cast<int>(devices_.size()))
    return Status::Invalid("expected between 1 and ", devices_.size(), " shards, got ", n);
  const int n_out = static_cast<int>(p.output_fields_.size());
  int64_t rows = 0;
  for (auto& b : shards) {
    if (!b) return Status::Invalid("null shard");
    rows += b->num_rows();
  }
  std::vector<std::vector<gdv_column_t>> cols(n);
  std::vector<std::vector<gdv_out_column_t>> outs(n, std::vector<gdv_out_column_t>(n_out));
  std::vector<std::vector<std::shared_ptr<arrow::Buffer>>> vbuf(n), dbuf(n), obuf(n);
  std::vector<gdv_shard_t> sh(n);
  for (int s = 0; s < n; s++) {
    int64_t lo = 0, hi = 0;
    ShardBounds(rows, n, s, &lo, &hi);
    if (shards[s]->num_rows() != hi - lo)
      return Status::Invalid("shard ", s, " holds ", shards[s]->num_rows(), " rows; gdv_shard_bounds gives ", hi - lo);
    bool device = false;
    std::shared_ptr<arrow::MemoryManager> mm;
    ARROW_RETURN_NOT_OK(MarshalBatch(*shards[s], p.schema_, &cols[s], &device, &mm));
    if (!device) return Status::Invalid("shard ", s, " is host-resident: pass ONE host batch to the other overload");
    vbuf[s].resize(n_out); dbuf[s].resize(n_out); obuf[s].resize(n_out);
    int64_t varlen_guess = 64;
    for (auto& c : cols[s]) if (c.offsets) varlen_guess += c.data_size;
    for (int e = 0; e < n_out; e++) {
      const bool varlen = IsVarlen(*p.output_fields_[e]->type());
      int64_t vbytes = 0, dbytes = 0;
      GDV_CXX_RETURN_NOT_OK(gdv_projector_output_sizes(p.handle_, e, hi - lo, GDV_MEM_DEVICE, &vbytes, &dbytes));
      if (varlen) dbytes = std::max(dbytes, varlen_guess);
      ARROW_ASSIGN_OR_RAISE(vbuf[s][e], AllocOut(vbytes, true, nullptr, mm));
      ARROW_ASSIGN_OR_RAISE(dbuf[s][e], AllocOut(dbytes, true, nullptr, mm));
      std::memset(&outs[s][e], 0, sizeof(gdv_out_column_t));
      outs[s][e].validity = reinterpret_cast<void*>(vbuf[s][e]->address());
      outs[s][e].validity_size = Room(*vbuf[s][e]);
      outs[s][e].data = reinterpret_cast<void*>(dbuf[s][e]->address());
      outs[s][e].data_size = varlen ? dbuf[s][e]->size() : Room(*dbuf[s][e]);
      if (varlen) {
        ARROW_ASSIGN_OR_RAISE(obuf[s][e], AllocOut((hi - lo + 1) * 4, true, nullptr, mm));
        outs[s][e].offsets = reinterpret_cast<void*>(obuf[s][e]->address());
        outs[s][e].offsets_size = obuf[s][e]->size();
      }
    }
    std::memset(&sh[s], 0, sizeof(gdv_shard_t));
    sh[s].device = devices_[s];
    sh[s].cols = cols[s].data();
    sh[s].outs = outs[s].data();
  }
  GDV_CXX_RETURN_NOT_OK(gdv_projector_evaluate_sharded(p.handle_, rows, static_cast<int>(cols[0].size()), n_out, sh.data(), n, 0));
  outputs->assign(n, {});
  for (int s = 0; s < n; s++)
    for (int e = 0; e < n_out; e++) {
      std::vector<std::shared_ptr<arrow::Buffer>> bufs;
      if (obuf[s][e]) bufs = {vbuf[s][e], obuf[s][e], arrow::SliceBuffer(dbuf[s][e], 0, outs[s][e].data_size)};
      else bufs = {vbuf[s][e], dbuf[s][e]};
      (*outputs)[s].push_back(arrow::MakeArray(arrow::ArrayData::Make(p.output_fields_[e]->type(), shards[s]->num_rows(), std::move(bufs))));
    }
  return Status::OK();
}

Status ShardedFilter::Make(SchemaPtr schema, ConditionPtr condition, std::vector<int> devices,
                           std::shared_ptr<Configuration> configuration, std::shared_ptr<ShardedFilter>* out) {
  if (!out) return Status::Invalid("ShardedFilter output cannot be null");
  std::shared_ptr<ShardedFilter> sf(new ShardedFilter());
  ARROW_RETURN_NOT_OK(Filter::Make(schema, condition, configuration, &sf->filter_));
  sf->devices_ = AllDevicesIfEmpty(std::move(devices));
  *out = sf;
  return Status::OK();
}

Status ShardedFilter::Evaluate(const arrow::RecordBatch& batch, std::shared_ptr<SelectionVector> out) const {
  if (!out) return Status::Invalid("Selection vector cannot be null");
  std::vector<gdv_column_t> cols;
  bool device = false;
  std::shared_ptr<arrow::MemoryManager> mm;
  ARROW_RETURN_NOT_OK(MarshalBatch(batch, filter_->schema_, &cols, &device, &mm));
  if (device || !out->GetBuffer().is_cpu()) return Status::Invalid("the one-batch overload takes host-resident batches and vectors");
  if (out->GetMaxSlots() < batch.num_rows())
    return Status::Invalid("Selection vector too small: max slots ", out->GetMaxSlots(), " < number of rows ", batch.num_rows());
  int64_t count = 0;
  std::vector<int32_t> devs(devices_.begin(), devices_.end());
  GDV_CXX_RETURN_NOT_OK(gdv_filter_evaluate_host_sharded(filter_->handle_, batch.num_rows(), cols.data(), static_cast<int>(cols.size()),
                                                         static_cast<int>(out->GetMode()), reinterpret_cast<void*>(out->GetBuffer().address()),
                                                         out->GetMaxSlots(), &count, devs.data(), static_cast<int>(devs.size())));
  out->SetNumSlots(count);
  return Status::OK();
}

Status ShardedFilter::Evaluate(const std::vector<std::shared_ptr<arrow::RecordBatch>>& shards,
                               const std::vector<std::shared_ptr<SelectionVector>>& outs, int64_t* total) const {
  const int n = static_cast<int>(shards.size());
  if (n < 1 || n > static_cast<int>(devices_.size()) || outs.size() != shards.size())
    return Status::Invalid("expected between 1 and ", devices_.size(), " shards and one selection vector per shard");
  int64_t rows = 0;
  for (auto& b : shards) {
    if (!b) return Status::Invalid("null shard");
    rows += b->num_rows();
  }
  std::vector<std::vector<gdv_column_t>> cols(n);
  std::vector<gdv_shard_t> sh(n);
  for (int s = 0; s < n; s++) {
    int64_t lo = 0, hi = 0;
    ShardBounds(rows, n, s, &lo, &hi);
    if (shards[s]->num_rows() != hi - lo)
      return Status::Invalid("shard ", s, " holds ", shards[s]->num_rows(), " rows; gdv_shard_bounds gives ", hi - lo);
    bool device = false;
    std::shared_ptr<arrow::MemoryManager> mm;
    ARROW_RETURN_NOT_OK(MarshalBatch(*shards[s], filter_->schema_, &cols[s], &device, &mm));
    if (!outs[s] || outs[s]->GetMode() != outs[0]->GetMode()) return Status::Invalid("selection vectors of one mode are required");
    if (!device || outs[s]->GetBuffer().is_cpu()) return Status::Invalid("shard ", s, ": batch and selection vector must be device-resident");
    std::memset(&sh[s], 0, sizeof(gdv_shard_t));
    sh[s].device = devices_[s];
    sh[s].cols = cols[s].data();
    sh[s].out_indices = reinterpret_cast<void*>(outs[s]->GetBuffer().address());
    sh[s].max_slots = outs[s]->GetMaxSlots();
  }
  int64_t sum = 0;
  GDV_CXX_RETURN_NOT_OK(gdv_filter_evaluate_sharded(filter_->handle_, rows, static_cast<int>(cols[0].size()), static_cast<int>(outs[0]->GetMode()),
                                                    sh.data(), n, GDV_SHARD_GLOBAL_INDICES, &sum));
  for (int s = 0; s < n; s++) outs[s]->SetNumSlots(sh[s].num_selected);
  if (total) *total = sum;
  return Status::OK();
}

// ------------------------------------------------------------------ FilterProject (fused, round 4)

FilterProject::~FilterProject() { gdv_filter_project_free(handle_); }

Status FilterProject::Make(SchemaPtr schema, ConditionPtr condition, const ExpressionVector& exprs,
                           SelectionVector::Mode mode, std::shared_ptr<Configuration> configuration,
                           std::shared_ptr<FilterProject>* out) {
  if (!schema) return Status::Invalid("Schema cannot be null");
  if (!condition) return Status::Invalid("Condition cannot be null");
  if (exprs.empty()) return Status::Invalid("Expressions cannot be empty");
  if (!configuration) return Status::Invalid("Configuration cannot be null");
  if (!out) return Status::Invalid("FilterProject output cannot be null");
  std::shared_ptr<FilterProject> fp(new FilterProject());
  fp->schema_ = schema;
  fp->mode_ = mode;
  std::vector<gdv_expression*> hs;
  for (auto& e : exprs) {
    if (!e) return Status::Invalid("Expression cannot be null");
    hs.push_back(e->handle());
    fp->output_fields_.push_back(e->result());
  }
  Status st;
  gdv_schema_t* sh = MakeSchema(schema, &st);
  if (!sh) return st;
  gdv_config_t cfg{configuration->optimize(), configuration->dump_ir()};
  int rc = gdv_filter_project_make(sh, condition->handle(), hs.data(), static_cast<int>(hs.size()), static_cast<int>(mode),
                                   &cfg, &fp->handle_);
  gdv_schema_free(sh);
  if (rc == GDV_CODE_GEN_ERROR) {
    // not a fused shape (var-len columns / outputs): the reference's own chain behind the same interface
    fp->handle_ = nullptr;
    ARROW_RETURN_NOT_OK(Filter::Make(schema, condition, configuration, &fp->filter_));
    ARROW_RETURN_NOT_OK(Projector::Make(schema, exprs, mode == SelectionVector::MODE_NONE ? SelectionVector::MODE_UINT32 : mode,
                                        configuration, &fp->projector_));
  } else {
    GDV_CXX_RETURN_NOT_OK(rc);
  }
  *out = fp;
  return Status::OK();
}

Status FilterProject::Evaluate(const arrow::RecordBatch& batch, arrow::MemoryPool* pool, ArrayVector* output,
                               std::shared_ptr<SelectionVector> out_selection) const {
  if (!output) return Status::Invalid("Output array vector cannot be null");
  if (mode_ != SelectionVector::MODE_NONE) {
    if (!out_selection) return Status::Invalid("Selection vector cannot be null");
    if (out_selection->GetMode() != mode_) return Status::Invalid("selection vector of another mode than the one given to Make");
    if (out_selection->GetMaxSlots() < batch.num_rows())
      return Status::Invalid("Selection vector too small: max slots ", out_selection->GetMaxSlots(), " < number of rows ",
                             batch.num_rows());
  } else if (out_selection) {
    // MODE_NONE promises no selection vector: the fused kernel writes no index, and round 4 still set the
    // vector's slot count (uninitialised indices) while the chain below filled it — refused, both ways
    return Status::Invalid("FilterProject was made with MODE_NONE: it fills no selection vector");
  }
  if (handle_ == nullptr) {  // the chain
    std::shared_ptr<SelectionVector> sv = out_selection;
    if (!sv) {
      // the temporary lives where the batch does (a CPU vector under HBM-resident columns is refused by Filter)
      std::vector<gdv_column_t> probe;
      bool on_device = false;
      std::shared_ptr<arrow::MemoryManager> bmm;
      ARROW_RETURN_NOT_OK(MarshalBatch(batch, schema_, &probe, &on_device, &bmm));
      if (on_device) {
        ARROW_ASSIGN_OR_RAISE(auto ibuf, AllocOut(batch.num_rows() * 4, true, pool, bmm));
        ARROW_RETURN_NOT_OK(SelectionVector::Make(SelectionVector::MODE_UINT32, batch.num_rows(), ibuf, &sv));
      } else {
        ARROW_RETURN_NOT_OK(SelectionVector::MakeInt32(batch.num_rows(), pool, &sv));
      }
    }
    ARROW_RETURN_NOT_OK(filter_->Evaluate(batch, sv));
    return projector_->Evaluate(batch, sv.get(), pool, output);
  }
  std::vector<gdv_column_t> cols;
  bool device = false;
  std::shared_ptr<arrow::MemoryManager> mm;
  ARROW_RETURN_NOT_OK(MarshalBatch(batch, schema_, &cols, &device, &mm));
  if (out_selection && (!out_selection->GetBuffer().is_cpu()) != device)
    return Status::Invalid("batch and selection vector must live in the same memory domain");
  const int n_out = static_cast<int>(output_fields_.size());
  const int mem = device ? GDV_MEM_DEVICE : GDV_MEM_HOST;
  const int64_t rows = batch.num_rows();
  std::vector<gdv_out_column_t> outs(n_out);
  std::vector<std::shared_ptr<arrow::Buffer>> vbuf(n_out), dbuf(n_out);
  for (int e = 0; e < n_out; e++) {
    const auto& t = *output_fields_[e]->type();
    const int64_t words = (rows + 63) / 64 * 8;
    const int64_t dbytes = t.id() == arrow::Type::BOOL ? words : rows * (t.bit_width() / 8);
    ARROW_ASSIGN_OR_RAISE(vbuf[e], AllocOut(words, device, pool, mm));
    ARROW_ASSIGN_OR_RAISE(dbuf[e], AllocOut(dbytes, device, pool, mm));
    std::memset(&outs[e], 0, sizeof(outs[e]));
    outs[e].validity = reinterpret_cast<void*>(vbuf[e]->address());
    outs[e].validity_size = Room(*vbuf[e]);
    outs[e].data = reinterpret_cast<void*>(dbuf[e]->address());
    outs[e].data_size = Room(*dbuf[e]);
  }
  int64_t count = 0;
  GDV_CXX_RETURN_NOT_OK(gdv_filter_project_evaluate(
      handle_, rows, cols.data(), static_cast<int>(cols.size()), outs.data(), n_out,
      out_selection ? reinterpret_cast<void*>(out_selection->GetBuffer().address()) : nullptr,
      out_selection ? out_selection->GetMaxSlots() : 0, &count, nullptr, mem, nullptr, 0));
  if (out_selection) out_selection->SetNumSlots(count);
  for (int e = 0; e < n_out; e++)
    output->push_back(arrow::MakeArray(arrow::ArrayData::Make(output_fields_[e]->type(), count, {vbuf[e], dbuf[e]})));
  return Status::OK();
}

std::string FilterProject::DumpIR() {
  if (handle_ == nullptr) return filter_->DumpIR() + projector_->DumpIR();
  char* s = gdv_filter_project_dump_ir(handle_);
  std::string r = s ? s : "";
  gdv_free_string(s);
  return r;
}

// ------------------------------------------------------------------ registry

std::string FunctionSignature::ToString() const {
  std::string s = ret_type_->ToString() + " " + base_name_ + "(";
  for (size_t i = 0; i < param_types_.size(); i++) s += (i ? ", " : "") + param_types_[i]->ToString();
  return s + ")";
}

std::vector<std::shared_ptr<FunctionSignature>> GetRegisteredFunctionSignatures() {
  std::vector<std::shared_ptr<FunctionSignature>> out;
  const int n = gdv_registry_size();
  for (int i = 0; i < n; i++) {
    const char* name = nullptr;
    gdv_type_t ret, params[8];
    int np = 0;
    if (gdv_registry_get(i, &name, &ret, params, 8, &np) != GDV_OK) continue;
    DataTypeVector ps;
    for (int k = 0; k < np && k < 8; k++) ps.push_back(FromGdvType(params[k]));
    out.push_back(std::make_shared<FunctionSignature>(name, ps, FromGdvType(ret)));
  }
  return out;
}

ExpressionRegistry::ExpressionRegistry() : signatures_(GetRegisteredFunctionSignatures()) {}
ExpressionRegistry::~ExpressionRegistry() {}
DataTypeVector ExpressionRegistry::supported_types() {
  // what include/gandiva_amd.h's gdv_type_t can carry (time units: the ones the registry's
  // signatures are written for; decimal128(38, 0) stands for every precision / scale, as in the lineage)
  return {arrow::boolean(), arrow::uint8(), arrow::uint16(), arrow::uint32(), arrow::uint64(), arrow::int8(),
          arrow::int16(), arrow::int32(), arrow::int64(), arrow::float32(), arrow::float64(), arrow::utf8(),
          arrow::binary(), arrow::date32(), arrow::date64(), arrow::timestamp(arrow::TimeUnit::MILLI),
          arrow::time32(arrow::TimeUnit::MILLI), arrow::time64(arrow::TimeUnit::MICRO), arrow::decimal128(38, 0)};
}
const ExpressionRegistry::FunctionSignatureIterator ExpressionRegistry::function_signature_begin() {
  return FunctionSignatureIterator(&signatures_, 0);
}
const ExpressionRegistry::FunctionSignatureIterator ExpressionRegistry::function_signature_end() const {
  return FunctionSignatureIterator(&signatures_, signatures_.size());
}

}  // namespace gandiva
