// gandiva_amd — MI355X-native expression evaluator for Arrow record batches.
//
// Data-type descriptors.  Type ids are numerically identical to arrow::Type::type
// (pyarrow/include/arrow/type_fwd.h:330-402) so that the gandiva:: C++ layer and the C-ABI
// can pass Arrow ids straight through.  Only the types the Projector/Filter hot path
// evaluates are representable.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace gdv {

enum TypeId : int32_t {
  kNA = 0,
  kBool = 1,
  kUInt8 = 2,
  kInt8 = 3,
  kUInt16 = 4,
  kInt16 = 5,
  kUInt32 = 6,
  kInt32 = 7,
  kUInt64 = 8,
  kInt64 = 9,
  kFloat = 11,
  kDouble = 12,
  kString = 13,
  kBinary = 14,
  kDate32 = 16,
  kDate64 = 17,
  kTimestamp = 18,
  kTime32 = 19,
  kTime64 = 20,
  kDecimal128 = 23,
};

// arrow::TimeUnit::type
enum TimeUnit : int32_t { kSecond = 0, kMilli = 1, kMicro = 2, kNano = 3 };

struct DataType {
  TypeId id = kNA;
  int32_t precision = 0;  // decimal128 precision; TimeUnit for time32/time64/timestamp
  int32_t scale = 0;      // decimal128 scale

  DataType() = default;
  DataType(TypeId i, int32_t p = 0, int32_t s = 0) : id(i), precision(p), scale(s) {}

  bool operator==(const DataType& o) const {
    if (id != o.id) return false;
    if (id == kDecimal128) return precision == o.precision && scale == o.scale;
    if (id == kTimestamp || id == kTime32 || id == kTime64) return precision == o.precision;
    return true;
  }
  bool operator!=(const DataType& o) const { return !(*this == o); }

  bool is_varlen() const { return id == kString || id == kBinary; }
  bool is_decimal() const { return id == kDecimal128; }
  bool is_floating() const { return id == kFloat || id == kDouble; }
  bool is_integer() const { return id >= kUInt8 && id <= kInt64; }
  bool is_signed_integer() const {
    return id == kInt8 || id == kInt16 || id == kInt32 || id == kInt64;
  }
  // Width in bytes of one slot of the values buffer; 0 for bool (bit-packed) and
  // var-len (offsets buffer is int32, data buffer is bytes).
  int byte_width() const;
  // Same spelling as arrow::DataType::ToString() for these types.
  std::string ToString() const;
  // Suffix used in the names of the device function library ("int32", "float64", "utf8"…),
  // the naming scheme of the reference's precompiled functions.
  std::string Suffix() const;
  // C++ type a value of this type has inside a kernel.
  std::string CType() const;
};

inline DataType boolean() { return DataType(kBool); }
inline DataType int8() { return DataType(kInt8); }
inline DataType int16() { return DataType(kInt16); }
inline DataType int32() { return DataType(kInt32); }
inline DataType int64() { return DataType(kInt64); }
inline DataType uint8() { return DataType(kUInt8); }
inline DataType uint16() { return DataType(kUInt16); }
inline DataType uint32() { return DataType(kUInt32); }
inline DataType uint64() { return DataType(kUInt64); }
inline DataType float32() { return DataType(kFloat); }
inline DataType float64() { return DataType(kDouble); }
inline DataType utf8() { return DataType(kString); }
inline DataType binary() { return DataType(kBinary); }
inline DataType date32() { return DataType(kDate32); }
inline DataType date64() { return DataType(kDate64); }
inline DataType timestamp(int32_t unit = kMilli) { return DataType(kTimestamp, unit); }
inline DataType time32(int32_t unit = kMilli) { return DataType(kTime32, unit); }
inline DataType time64(int32_t unit = kMicro) { return DataType(kTime64, unit); }
inline DataType decimal128(int32_t p, int32_t s) { return DataType(kDecimal128, p, s); }

struct Field {
  std::string name;
  DataType type;
  bool nullable = true;
};

using Schema = std::vector<Field>;

// Status codes mirror arrow::StatusCode (pyarrow/include/arrow/status.h:97-100): the three
// Gandiva-specific codes keep their numeric values so the gandiva:: layer can rebuild an
// arrow::Status without a translation table.
enum StatusCode : int32_t {
  kOK = 0,
  kOutOfMemory = 1,
  kInvalid = 4,
  kNotImplemented = 10,
  kCodeGenError = 40,
  kExpressionValidationError = 41,
  kExecutionError = 42,
};

struct Status {
  StatusCode code = kOK;
  std::string msg;
  Status() = default;
  Status(StatusCode c, std::string m) : code(c), msg(std::move(m)) {}
  bool ok() const { return code == kOK; }
  static Status OK() { return Status(); }
  static Status Invalid(std::string m) { return Status(kInvalid, std::move(m)); }
  static Status CodeGenError(std::string m) { return Status(kCodeGenError, std::move(m)); }
  static Status ValidationError(std::string m) {
    return Status(kExpressionValidationError, std::move(m));
  }
  static Status ExecutionError(std::string m) { return Status(kExecutionError, std::move(m)); }
  static Status NotImplemented(std::string m) { return Status(kNotImplemented, std::move(m)); }
  static Status OutOfMemory(std::string m) { return Status(kOutOfMemory, std::move(m)); }
  std::string ToString() const;
};

#define GDV_RETURN_NOT_OK(expr)          \
  do {                                   \
    ::gdv::Status _s = (expr);           \
    if (!_s.ok()) return _s;             \
  } while (0)

}  // namespace gdv
