// gandiva/expression_registry.h (pyarrow/includes/libgandiva.pxd:274-277).
#pragma once
#include "gandiva/function_signature.h"

namespace gandiva {
std::vector<std::shared_ptr<FunctionSignature>> GetRegisteredFunctionSignatures();
}  // namespace gandiva
