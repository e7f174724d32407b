// Function registry: (name, parameter types) -> return type, null policy and the symbol
// in the device function library (gdv_device_lib.hpp).  Host-side mirror of the
// reference's FunctionRegistry / NativeFunction; enumerable through
// GetRegisteredFunctionSignatures (libgandiva.pxd:258-277).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "gdv_types.h"

namespace gdv {

// How a function treats null arguments — the reference's ResultNullableType:
//  kNullIfNull   result is null iff any argument is null; the value function never sees
//                validity (validity = AND of the arguments' validity words).
//  kNullNever    result is always valid; the value function receives each argument's
//                validity as an extra bool (isnull, hash, is_distinct_from …).
//  kNullInternal the value function decides per row and returns validity through an
//                out-parameter.
enum class NullPolicy { kNullIfNull, kNullNever, kNullInternal };

enum FunctionFlags : uint32_t {
  kNeedsContext = 1u,    // may raise an execution error (divide by zero …)
  kDecimalResult = 2u,   // return precision/scale follow the decimal result-type rules
  kPatternArg = 4u,      // last argument must be a literal compiled at Make time (like)
  kVarlenResult = 8u,    // returns utf8/binary: two-pass (length, then copy) evaluation
  kDecimalArgs = 16u,    // device function takes (precision, scale) after every decimal
                         // argument and the result's (precision, scale) last
  kDateFormatArg = 32u,  // to_date: the pattern (and suppress_errors) must be literals; compiled at Make time
};

struct FunctionDef {
  std::string name;
  std::vector<DataType> params;
  DataType ret;
  NullPolicy policy = NullPolicy::kNullIfNull;
  uint32_t flags = 0;
  std::string symbol;  // device function name

  std::string SignatureString() const;
};

class FunctionRegistry {
 public:
  static const FunctionRegistry& Get();
  // Exact match on parameter type ids (decimal precision/scale are wildcards).
  const FunctionDef* Lookup(const std::string& name, const std::vector<DataType>& params) const;
  const std::vector<FunctionDef>& all() const { return defs_; }

 private:
  FunctionRegistry();
  void Add(FunctionDef def);
  std::vector<FunctionDef> defs_;
  std::multimap<std::string, size_t> by_name_;
};

// Result type of a decimal operation (precision <= 38; scale reduced, but not below 6,
// when the natural precision overflows) — the reference's DecimalTypeUtil rules.
enum class DecimalOp { kAdd, kSubtract, kMultiply, kDivide, kMod };
DataType DecimalResultType(DecimalOp op, const DataType& a, const DataType& b);

}  // namespace gdv
