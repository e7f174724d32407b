// gandiva/node.h — gandiva::Node, gandiva::Expression (pyarrow/includes/libgandiva.pxd:27-41).
// Nodes are immutable handles onto trees owned by libgandiva_amd (gdv_node_t).
#pragma once
#include "gandiva/arrow.h"

struct gdv_node;
struct gdv_expression;

namespace gandiva {

class Node {
 public:
  ~Node();
  Node(const Node&) = delete;
  Node& operator=(const Node&) = delete;
  const DataTypePtr& return_type() const { return return_type_; }
  std::string ToString() const;
  gdv_node* handle() const { return handle_; }

 private:
  friend class TreeExprBuilder;
  Node(gdv_node* h, DataTypePtr t) : handle_(h), return_type_(std::move(t)) {}
  gdv_node* handle_;
  DataTypePtr return_type_;
};
using NodePtr = std::shared_ptr<Node>;
using NodeVector = std::vector<NodePtr>;

class Expression {
 public:
  virtual ~Expression();
  Expression(const Expression&) = delete;
  Expression& operator=(const Expression&) = delete;
  const NodePtr& root() const { return root_; }
  const FieldPtr& result() const { return result_; }
  std::string ToString() const { return root_->ToString(); }
  gdv_expression* handle() const { return handle_; }

 protected:
  friend class TreeExprBuilder;
  Expression(gdv_expression* h, NodePtr root, FieldPtr result)
      : handle_(h), root_(std::move(root)), result_(std::move(result)) {}
  gdv_expression* handle_;
  NodePtr root_;
  FieldPtr result_;
};
using ExpressionPtr = std::shared_ptr<Expression>;
using ExpressionVector = std::vector<ExpressionPtr>;

}  // namespace gandiva
