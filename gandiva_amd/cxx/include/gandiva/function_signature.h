// gandiva/function_signature.h (pyarrow/includes/libgandiva.pxd:258-272).
#pragma once
#include "gandiva/arrow.h"

namespace gandiva {
class FunctionSignature {
 public:
  FunctionSignature(std::string base_name, DataTypeVector param_types, DataTypePtr ret_type)
      : base_name_(std::move(base_name)), param_types_(std::move(param_types)), ret_type_(std::move(ret_type)) {}
  DataTypePtr ret_type() const { return ret_type_; }
  const std::string& base_name() const { return base_name_; }
  DataTypeVector param_types() const { return param_types_; }
  std::string ToString() const;

 private:
  std::string base_name_;
  DataTypeVector param_types_;
  DataTypePtr ret_type_;
};
}  // namespace gandiva
