// Expression trees: the host-side mirror of gandiva::Node / Expression / Condition
// (signatures pinned by pyarrow/includes/libgandiva.pxd:27-41,98-103; ToString() renderings
// pinned by pyarrow/tests/test_gandiva.py:381-393).  Trees are immutable and shared.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "gdv_types.h"

namespace gdv {

enum class NodeKind { kField, kLiteral, kFunction, kIf, kBoolean, kIn };

struct Literal {
  // One 128-bit payload covers every fixed-width type (decimal128 uses both words,
  // two's complement, lo first); var-len literals use `bytes`.
  uint64_t lo = 0;
  uint64_t hi = 0;
  std::string bytes;
  bool is_null = false;
};

class Node;
using NodePtr = std::shared_ptr<Node>;
using NodeVector = std::vector<NodePtr>;

class Node {
 public:
  Node(NodeKind k, DataType t) : kind_(k), type_(t) {}
  virtual ~Node() = default;
  NodeKind kind() const { return kind_; }
  const DataType& return_type() const { return type_; }
  virtual std::string ToString() const = 0;
  // Unambiguous serialisation for the plan caches: ToString() is pinned by the reference's
  // tests (no parentheses around nested AND/OR, string literals unescaped), so two different
  // trees can render alike; this one length-prefixes names and literal bytes and brackets
  // every node.
  virtual void AppendKey(std::string* out) const = 0;

 private:
  NodeKind kind_;
  DataType type_;
};

class FieldNode : public Node {
 public:
  explicit FieldNode(Field f) : Node(NodeKind::kField, f.type), field_(std::move(f)) {}
  const Field& field() const { return field_; }
  std::string ToString() const override;
  void AppendKey(std::string* out) const override;

 private:
  Field field_;
};

class LiteralNode : public Node {
 public:
  LiteralNode(DataType t, Literal v) : Node(NodeKind::kLiteral, t), value_(std::move(v)) {}
  const Literal& value() const { return value_; }
  bool is_null() const { return value_.is_null; }
  std::string ToString() const override;
  void AppendKey(std::string* out) const override;

 private:
  Literal value_;
};

class FunctionNode : public Node {
 public:
  FunctionNode(std::string name, NodeVector children, DataType ret)
      : Node(NodeKind::kFunction, ret), name_(std::move(name)), children_(std::move(children)) {}
  const std::string& name() const { return name_; }
  const NodeVector& children() const { return children_; }
  std::string ToString() const override;
  void AppendKey(std::string* out) const override;

 private:
  std::string name_;
  NodeVector children_;
};

// Builds a function node; regular-expression functions whose pattern is a plain literal are rewritten here, once, onto
// the matchers that answer them fastest (round 5):
//   regexp_like / regexp_matches(s, 'lit' | '^lit' | 'lit$' | '^lit$')  ->  like(s, '%lit%' | 'lit%' | '%lit' | 'lit')
//   regexp_replace(s, 'lit', 'to')                                      ->  replace(s, 'lit', 'to')
// where lit is a non-empty literal without regular-expression or LIKE metacharacters and `to` holds no backslash
// (RE2 rewrite syntax).  Anything else stays a regexp_* node: regexp_like / regexp_matches are compiled by the planner
// (gdv_regex.h), regexp_replace beyond a literal pattern is refused there with CodeGenError.
NodePtr MakeFunctionNode(std::string name, NodeVector children, DataType ret);

class IfNode : public Node {
 public:
  IfNode(NodePtr c, NodePtr t, NodePtr e, DataType ret)
      : Node(NodeKind::kIf, ret), cond_(std::move(c)), then_(std::move(t)), else_(std::move(e)) {}
  const NodePtr& condition() const { return cond_; }
  const NodePtr& then_node() const { return then_; }
  const NodePtr& else_node() const { return else_; }
  std::string ToString() const override;
  void AppendKey(std::string* out) const override;

 private:
  NodePtr cond_, then_, else_;
};

class BooleanNode : public Node {
 public:
  enum Op { kAnd, kOr };
  BooleanNode(Op op, NodeVector children)
      : Node(NodeKind::kBoolean, boolean()), op_(op), children_(std::move(children)) {}
  Op op() const { return op_; }
  const NodeVector& children() const { return children_; }
  std::string ToString() const override;
  void AppendKey(std::string* out) const override;

 private:
  Op op_;
  NodeVector children_;
};

// `eval IN (v0, v1, …)`; values share the type of the evaluated child.
class InNode : public Node {
 public:
  InNode(NodePtr eval, DataType value_type, std::vector<Literal> values)
      : Node(NodeKind::kIn, boolean()),
        eval_(std::move(eval)),
        value_type_(value_type),
        values_(std::move(values)) {}
  const NodePtr& eval() const { return eval_; }
  const DataType& value_type() const { return value_type_; }
  const std::vector<Literal>& values() const { return values_; }
  std::string ToString() const override;
  void AppendKey(std::string* out) const override;

 private:
  NodePtr eval_;
  DataType value_type_;
  std::vector<Literal> values_;
};

// Expression = root node + result field; Condition = Expression whose result is
// the fixed field ("cond", bool) (libgandiva.pxd:98-103; test_gandiva.py:104).
class Expression {
 public:
  Expression(NodePtr root, Field result) : root_(std::move(root)), result_(std::move(result)) {}
  const NodePtr& root() const { return root_; }
  const Field& result() const { return result_; }
  std::string ToString() const { return root_->ToString(); }
  std::string CacheKey() const {
    std::string k;
    root_->AppendKey(&k);
    return k + "->" + result_.type.ToString();
  }

 private:
  NodePtr root_;
  Field result_;
};
using ExpressionPtr = std::shared_ptr<Expression>;

std::string LiteralToString(const DataType& t, const Literal& v);

}  // namespace gdv
