#include "gdv_node.h"

#include <cstdio>
#include <cstring>
#include <sstream>

namespace gdv {

// ---------------------------------------------------------------- DataType

int DataType::byte_width() const {
  switch (id) {
    case kUInt8: case kInt8: return 1;
    case kUInt16: case kInt16: return 2;
    case kUInt32: case kInt32: case kFloat: case kDate32: case kTime32: return 4;
    case kUInt64: case kInt64: case kDouble: case kDate64: case kTimestamp: case kTime64: return 8;
    case kDecimal128: return 16;
    default: return 0;
  }
}

static const char* UnitName(int32_t u) {
  switch (u) {
    case kSecond: return "s";
    case kMilli: return "ms";
    case kMicro: return "us";
    default: return "ns";
  }
}

std::string DataType::ToString() const {
  switch (id) {
    case kNA: return "null";
    case kBool: return "bool";
    case kUInt8: return "uint8";
    case kInt8: return "int8";
    case kUInt16: return "uint16";
    case kInt16: return "int16";
    case kUInt32: return "uint32";
    case kInt32: return "int32";
    case kUInt64: return "uint64";
    case kInt64: return "int64";
    case kFloat: return "float";
    case kDouble: return "double";
    case kString: return "string";
    case kBinary: return "binary";
    case kDate32: return "date32[day]";
    case kDate64: return "date64[ms]";
    case kTimestamp: return std::string("timestamp[") + UnitName(precision) + "]";
    case kTime32: return std::string("time32[") + UnitName(precision) + "]";
    case kTime64: return std::string("time64[") + UnitName(precision) + "]";
    case kDecimal128:
      return "decimal128(" + std::to_string(precision) + ", " + std::to_string(scale) + ")";
  }
  return "?";
}

std::string DataType::Suffix() const {
  switch (id) {
    case kBool: return "boolean";
    case kUInt8: return "uint8";
    case kInt8: return "int8";
    case kUInt16: return "uint16";
    case kInt16: return "int16";
    case kUInt32: return "uint32";
    case kInt32: return "int32";
    case kUInt64: return "uint64";
    case kInt64: return "int64";
    case kFloat: return "float32";
    case kDouble: return "float64";
    case kString: return "utf8";
    case kBinary: return "binary";
    case kDate32: return "date32";
    case kDate64: return "date64";
    case kTimestamp: return "timestamp";
    case kTime32: return "time32";
    case kTime64: return "time64";
    case kDecimal128: return "decimal128";
    default: return "na";
  }
}

std::string DataType::CType() const {
  switch (id) {
    case kBool: return "bool";
    case kUInt8: return "gdv_uint8";
    case kInt8: return "gdv_int8";
    case kUInt16: return "gdv_uint16";
    case kInt16: return "gdv_int16";
    case kUInt32: return "gdv_uint32";
    case kInt32: case kDate32: case kTime32: return "gdv_int32";
    case kUInt64: return "gdv_uint64";
    case kInt64: case kDate64: case kTimestamp: case kTime64: return "gdv_int64";
    case kFloat: return "gdv_float32";
    case kDouble: return "gdv_float64";
    case kDecimal128: return "gdv_int128";
    case kString: case kBinary: return "gdv_str";
    default: return "void";
  }
}

std::string Status::ToString() const {
  const char* name = "Unknown";
  switch (code) {
    case kOK: return "OK";
    case kOutOfMemory: name = "Out of memory"; break;
    case kInvalid: name = "Invalid"; break;
    case kNotImplemented: name = "NotImplemented"; break;
    case kCodeGenError: name = "CodeGenError"; break;
    case kExpressionValidationError: name = "ExpressionValidationError"; break;
    case kExecutionError: name = "ExecutionError"; break;
  }
  return std::string(name) + ": " + msg;
}

// ---------------------------------------------------------------- literals

static std::string Int128ToString(uint64_t lo, uint64_t hi) {
  unsigned __int128 v = (static_cast<unsigned __int128>(hi) << 64) | lo;
  bool neg = (hi >> 63) != 0;
  if (neg) v = ~v + 1;
  if (v == 0) return "0";
  std::string s;
  while (v != 0) {
    s.insert(s.begin(), static_cast<char>('0' + static_cast<int>(v % 10)));
    v /= 10;
  }
  return neg ? "-" + s : s;
}

std::string LiteralToString(const DataType& t, const Literal& v) {
  if (v.is_null) return "null";
  std::stringstream ss;
  switch (t.id) {
    case kBool: ss << (v.lo ? 1 : 0); break;
    case kUInt8: case kUInt16: case kUInt32: case kUInt64: ss << v.lo; break;
    case kInt8: ss << static_cast<int>(static_cast<int8_t>(v.lo)); break;
    case kInt16: ss << static_cast<int16_t>(v.lo); break;
    case kInt32: case kDate32: case kTime32: ss << static_cast<int32_t>(v.lo); break;
    case kInt64: case kDate64: case kTimestamp: case kTime64:
      ss << static_cast<int64_t>(v.lo);
      break;
    case kFloat: {
      uint32_t bits = static_cast<uint32_t>(v.lo);
      float f;
      std::memcpy(&f, &bits, 4);
      // decimal rendering loses precision, so the raw bits follow in hex
      // (format pinned by test_gandiva.py:381-382)
      ss << f << " raw(" << std::hex << bits << ")";
      break;
    }
    case kDouble: {
      double d;
      std::memcpy(&d, &v.lo, 8);
      ss << d << " raw(" << std::hex << v.lo << ")";
      break;
    }
    case kString: case kBinary: ss << "'" << v.bytes << "'"; break;
    case kDecimal128:
      ss << Int128ToString(v.lo, v.hi) << "," << t.precision << "," << t.scale;
      break;
    default: ss << "?"; break;
  }
  return ss.str();
}

// ---------------------------------------------------------------- nodes

std::string FieldNode::ToString() const {
  return "(" + return_type().ToString() + ") " + field_.name;
}

std::string LiteralNode::ToString() const {
  return "(const " + return_type().ToString() + ") " + LiteralToString(return_type(), value_);
}

std::string FunctionNode::ToString() const {
  std::string s = return_type().ToString() + " " + name_ + "(";
  bool first = true;
  for (auto& c : children_) {
    if (!first) s += ", ";
    s += c->ToString();
    first = false;
  }
  return s + ")";
}

std::string IfNode::ToString() const {
  return "if (" + cond_->ToString() + ") { " + then_->ToString() + " } else { " +
         else_->ToString() + " }";
}

std::string BooleanNode::ToString() const {
  std::string s;
  bool first = true;
  for (auto& c : children_) {
    if (!first) s += (op_ == kAnd) ? " && " : " || ";
    s += c->ToString();
    first = false;
  }
  return s;
}

std::string InNode::ToString() const {
  std::string s = eval_->ToString() + " IN (";
  bool first = true;
  for (auto& v : values_) {
    if (!first) s += ", ";
    s += LiteralToString(value_type_, v);
    first = false;
  }
  return s + ")";
}

// ---------------------------------------------------------------- cache keys

static void KeyBytes(std::string* out, const std::string& b) {
  *out += std::to_string(b.size());
  *out += ':';
  *out += b;
}

static void KeyLiteral(std::string* out, const DataType& t, const Literal& v) {
  if (v.is_null) {
    *out += "null;";
  } else if (t.is_varlen()) {
    KeyBytes(out, v.bytes);
    *out += ';';
  } else {
    char buf[48];
    snprintf(buf, sizeof(buf), "%llx.%llx;", static_cast<unsigned long long>(v.hi),
             static_cast<unsigned long long>(v.lo));
    *out += buf;
  }
}

void FieldNode::AppendKey(std::string* out) const {
  *out += "F[";
  KeyBytes(out, field_.name);
  *out += ' ' + return_type().ToString() + ']';
}

void LiteralNode::AppendKey(std::string* out) const {
  *out += "L[" + return_type().ToString() + ' ';
  KeyLiteral(out, return_type(), value_);
  *out += ']';
}

void FunctionNode::AppendKey(std::string* out) const {
  *out += "f[";
  KeyBytes(out, name_);
  *out += ' ' + return_type().ToString() + ' ' + std::to_string(children_.size());
  for (auto& c : children_) {
    *out += ' ';
    c->AppendKey(out);
  }
  *out += ']';
}

void IfNode::AppendKey(std::string* out) const {
  *out += "I[" + return_type().ToString() + ' ';
  cond_->AppendKey(out);
  *out += ' ';
  then_->AppendKey(out);
  *out += ' ';
  else_->AppendKey(out);
  *out += ']';
}

void BooleanNode::AppendKey(std::string* out) const {
  *out += (op_ == kAnd) ? "A[" : "O[";
  *out += std::to_string(children_.size());
  for (auto& c : children_) {
    *out += ' ';
    c->AppendKey(out);
  }
  *out += ']';
}

void InNode::AppendKey(std::string* out) const {
  *out += "N[" + value_type_.ToString() + ' ';
  eval_->AppendKey(out);
  *out += ' ' + std::to_string(values_.size()) + ' ';
  for (auto& v : values_) KeyLiteral(out, value_type_, v);
  *out += ']';
}


namespace {
bool PlainRegexLiteral(const std::string& t) {
  if (t.empty()) return false;
  for (unsigned char ch : t)
    if (std::strchr("\\^$.|?*+()[]{}%_", ch) != nullptr || ch == 0) return false;
  return true;
}
}  // namespace

NodePtr MakeFunctionNode(std::string name, NodeVector children, DataType ret) {
  auto literal_text = [&](size_t i, std::string* out) {
    if (i >= children.size() || !children[i] || children[i]->kind() != NodeKind::kLiteral) return false;
    auto& l = static_cast<const LiteralNode&>(*children[i]);
    if (l.is_null() || !l.return_type().is_varlen()) return false;
    *out = l.value().bytes;
    return true;
  };
  std::string pat, to;
  if ((name == "regexp_like" || name == "regexp_matches") && children.size() == 2 && literal_text(1, &pat)) {
    const bool head = !pat.empty() && pat.front() == '^';
    const bool tail = pat.size() > (head ? 1u : 0u) && pat.back() == '$';
    const std::string lit = pat.substr(head ? 1 : 0, pat.size() - (head ? 1 : 0) - (tail ? 1 : 0));
    if (PlainRegexLiteral(lit)) {
      Literal v;
      v.bytes = (head ? "" : "%") + lit + (tail ? "" : "%");
      NodeVector kids{children[0], std::make_shared<LiteralNode>(children[1]->return_type(), v)};
      return std::make_shared<FunctionNode>("like", std::move(kids), ret);
    }
  }
  if (name == "regexp_replace" && children.size() == 3 && literal_text(1, &pat) && literal_text(2, &to) &&
      PlainRegexLiteral(pat) && to.find('\\') == std::string::npos)
    return std::make_shared<FunctionNode>("replace", std::move(children), ret);
  return std::make_shared<FunctionNode>(std::move(name), std::move(children), ret);
}

}  // namespace gdv
