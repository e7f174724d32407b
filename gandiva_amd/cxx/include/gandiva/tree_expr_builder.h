// gandiva/tree_expr_builder.h (pyarrow/includes/libgandiva.pxd:110-212).
#pragma once
#include <unordered_set>

#include "gandiva/condition.h"
#include "gandiva/decimal_scalar.h"
#include "gandiva/node.h"

namespace gandiva {

class TreeExprBuilder {
 public:
  static NodePtr MakeLiteral(bool value);
  static NodePtr MakeLiteral(uint8_t value);
  static NodePtr MakeLiteral(uint16_t value);
  static NodePtr MakeLiteral(uint32_t value);
  static NodePtr MakeLiteral(uint64_t value);
  static NodePtr MakeLiteral(int8_t value);
  static NodePtr MakeLiteral(int16_t value);
  static NodePtr MakeLiteral(int32_t value);
  static NodePtr MakeLiteral(int64_t value);
  static NodePtr MakeLiteral(float value);
  static NodePtr MakeLiteral(double value);
  static NodePtr MakeStringLiteral(const std::string& value);
  static NodePtr MakeBinaryLiteral(const std::string& value);
  // unscaled 128-bit value as (high, low) words
  static NodePtr MakeDecimalLiteral(int64_t high, uint64_t low, int32_t precision, int32_t scale);
  // [M] the lineage's decimal literal (tree_expr_builder.h as recalled; not bound by the .pxd)
  static NodePtr MakeLiteral(const DecimalScalar128& value);
  static NodePtr MakeNull(DataTypePtr data_type);

  static NodePtr MakeField(FieldPtr field);
  static NodePtr MakeFunction(const std::string& name, const NodeVector& params,
                              DataTypePtr return_type);
  static NodePtr MakeIf(NodePtr condition, NodePtr then_node, NodePtr else_node,
                        DataTypePtr result_type);
  static NodePtr MakeAnd(const NodeVector& children);
  static NodePtr MakeOr(const NodeVector& children);

  static ExpressionPtr MakeExpression(NodePtr root_node, FieldPtr result_field);
  static ExpressionPtr MakeExpression(const std::string& function, const FieldVector& in_fields,
                                      FieldPtr out_field);
  static ConditionPtr MakeCondition(NodePtr root_node);
  static ConditionPtr MakeCondition(const std::string& function, const FieldVector& in_fields);

  static NodePtr MakeInExpressionInt32(NodePtr node, const std::unordered_set<int32_t>& constants);
  static NodePtr MakeInExpressionInt64(NodePtr node, const std::unordered_set<int64_t>& constants);
  static NodePtr MakeInExpressionString(NodePtr node, const std::unordered_set<std::string>& constants);
  static NodePtr MakeInExpressionBinary(NodePtr node, const std::unordered_set<std::string>& constants);
  static NodePtr MakeInExpressionDate32(NodePtr node, const std::unordered_set<int32_t>& constants);
  static NodePtr MakeInExpressionDate64(NodePtr node, const std::unordered_set<int64_t>& constants);
  static NodePtr MakeInExpressionTime32(NodePtr node, const std::unordered_set<int32_t>& constants);
  static NodePtr MakeInExpressionTime64(NodePtr node, const std::unordered_set<int64_t>& constants);
  static NodePtr MakeInExpressionTimeStamp(NodePtr node, const std::unordered_set<int64_t>& constants);
  // [M] the lineage's remaining IN builders (as recalled; not bound by the .pxd).  Floating-point
  // values compare by VALUE, as a hash set of floats does: -0.0 and +0.0 are one value, NaN equals nothing.
  static NodePtr MakeInExpressionFloat(NodePtr node, const std::unordered_set<float>& constants);
  static NodePtr MakeInExpressionDouble(NodePtr node, const std::unordered_set<double>& constants);
  static NodePtr MakeInExpressionDecimal(NodePtr node, std::unordered_set<DecimalScalar128>& constants,
                                         int32_t precision, int32_t scale);
};

}  // namespace gandiva
