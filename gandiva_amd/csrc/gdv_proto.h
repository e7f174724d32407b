// Schema / ExpressionList / Condition from the protobuf bytes the reference's Java side hands to
// JNI buildProjector / buildFilter (gdv_proto.cc holds the message layout it assumes).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "gdv_node.h"

namespace gdv {

Status DecodeSchema(const uint8_t* data, size_t size, Schema* out);
Status DecodeExpressionList(const uint8_t* data, size_t size, std::vector<ExpressionPtr>* out);
Status DecodeCondition(const uint8_t* data, size_t size, ExpressionPtr* out);

}  // namespace gdv
