// gandiva/condition.h — gandiva::Condition: an Expression whose result is ("cond", bool)
// (pyarrow/includes/libgandiva.pxd:98-103).
#pragma once
#include "gandiva/node.h"

namespace gandiva {
class Condition : public Expression {
 private:
  friend class TreeExprBuilder;
  Condition(gdv_expression* h, NodePtr root)
      : Expression(h, std::move(root), arrow::field("cond", arrow::boolean())) {}
};
using ConditionPtr = std::shared_ptr<Condition>;
}  // namespace gandiva
