// gandiva/expression_registry.h (pyarrow/includes/libgandiva.pxd:274-277 binds GetRegisteredFunctionSignatures).
// class ExpressionRegistry — supported_types() and the function-signature iterators — is [M]: restated
// from the lineage's expression_registry.h as recalled (the .pxd does not bind it; no source in the
// reference mount).  The iterator walks a snapshot of the registry the device library was built with.
#pragma once
#include <memory>
#include <vector>

#include "gandiva/function_signature.h"

namespace gandiva {

class ExpressionRegistry {
 public:
  ExpressionRegistry();
  ~ExpressionRegistry();
  // the data types expressions may use (columns, literals, results)
  static DataTypeVector supported_types();

  class FunctionSignatureIterator {
   public:
    FunctionSignatureIterator(const std::vector<std::shared_ptr<FunctionSignature>>* all, size_t pos)
        : all_(all), pos_(pos) {}
    bool operator!=(const FunctionSignatureIterator& other) const { return pos_ != other.pos_ || all_ != other.all_; }
    bool operator==(const FunctionSignatureIterator& other) const { return !(*this != other); }
    FunctionSignature operator*() const { return *(*all_)[pos_]; }
    FunctionSignatureIterator& operator++() { ++pos_; return *this; }
    FunctionSignatureIterator operator++(int) { FunctionSignatureIterator old = *this; ++pos_; return old; }

   private:
    const std::vector<std::shared_ptr<FunctionSignature>>* all_;
    size_t pos_;
  };
  const FunctionSignatureIterator function_signature_begin();
  const FunctionSignatureIterator function_signature_end() const;

 private:
  std::vector<std::shared_ptr<FunctionSignature>> signatures_;
};

std::vector<std::shared_ptr<FunctionSignature>> GetRegisteredFunctionSignatures();
}  // namespace gandiva
