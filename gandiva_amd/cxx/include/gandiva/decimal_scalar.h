// gandiva/decimal_scalar.h — gandiva::DecimalScalar128: a decimal128 value with its precision and scale,
// the argument of TreeExprBuilder::MakeLiteral(const DecimalScalar128&) and of MakeInExpressionDecimal.
// [M] restated from the reference lineage's decimal_scalar.h / basic_decimal_scalar.h as recalled — the
// reference mount holds no source and libgandiva.pxd does not bind this class; member names follow the
// lineage (value / precision / scale / ToString, equality, std::hash).
#pragma once
#include <functional>
#include <string>

#include "arrow/util/decimal.h"
#include "gandiva/arrow.h"

namespace gandiva {

class DecimalScalar128 {
 public:
  DecimalScalar128() : value_(0), precision_(0), scale_(0) {}
  DecimalScalar128(const arrow::Decimal128& value, int32_t precision, int32_t scale)
      : value_(value), precision_(precision), scale_(scale) {}
  DecimalScalar128(int64_t high_bits, uint64_t low_bits, int32_t precision, int32_t scale)
      : value_(high_bits, low_bits), precision_(precision), scale_(scale) {}
  // the UNSCALED digits ("-12345" with scale 2 is -123.45), as the lineage's string constructor takes them
  DecimalScalar128(const std::string& value, int32_t precision, int32_t scale)
      : value_(arrow::Decimal128::FromString(value).ValueOr(arrow::Decimal128(0))), precision_(precision), scale_(scale) {}

  const arrow::Decimal128& value() const { return value_; }
  int32_t precision() const { return precision_; }
  int32_t scale() const { return scale_; }
  std::string ToString() const {
    return value_.ToIntegerString() + "," + std::to_string(precision_) + "," + std::to_string(scale_);
  }
  friend bool operator==(const DecimalScalar128& a, const DecimalScalar128& b) {
    return a.value_ == b.value_ && a.precision_ == b.precision_ && a.scale_ == b.scale_;
  }
  friend bool operator!=(const DecimalScalar128& a, const DecimalScalar128& b) { return !(a == b); }

 private:
  arrow::Decimal128 value_;
  int32_t precision_, scale_;
};

}  // namespace gandiva

namespace std {
template <>
struct hash<gandiva::DecimalScalar128> {
  size_t operator()(const gandiva::DecimalScalar128& s) const noexcept {
    size_t h = std::hash<uint64_t>()(s.value().low_bits());
    h ^= std::hash<int64_t>()(s.value().high_bits()) + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h ^= std::hash<int32_t>()(s.precision() * 64 + s.scale()) + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h;
  }
};
}  // namespace std
