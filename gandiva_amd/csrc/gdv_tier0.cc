// Tier 0 (round 6): expression trees -> the post-fix program gdv_tier0.hip interprets.  See gdv_tier0.h.
#include "gdv_tier0.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>

namespace gdv {
namespace {

using namespace tier0;

int KindOf(const DataType& t) {
  switch (t.id) {
    case gdv::kBool: return kTBool;
    case kInt8: return kTI8;
    case kUInt8: return kTU8;
    case kInt16: return kTI16;
    case kUInt16: return kTU16;
    case kInt32: case kDate32: case kTime32: return kTI32;
    case kUInt32: return kTU32;
    case kInt64: case kDate64: case kTimestamp: case kTime64: return kTI64;
    case kUInt64: return kTU64;
    case kFloat: return kTF32;
    case kDouble: return kTF64;
    default: return -1;
  }
}

struct Builder {
  const Schema& schema;
  const KernelPlan& plan;
  Args* out;
  std::string why;
  int depth = 0, max_depth = 0, nlits = 0;

  bool Fail(const std::string& w) {
    if (why.empty()) why = w;
    return false;
  }
  bool Emit(int op, int a = 0, int b = 0, int c = 0) {
    if (out->ncode >= kMaxCode) return Fail("program longer than " + std::to_string(kMaxCode) + " instructions");
    out->code[out->ncode++] = static_cast<uint32_t>(op) | (static_cast<uint32_t>(a) << 8) | (static_cast<uint32_t>(b) << 16) |
                              (static_cast<uint32_t>(c) << 24);
    return true;
  }
  bool Push() {
    if (++depth > kMaxDepth) return Fail("operand stack deeper than " + std::to_string(kMaxDepth));
    max_depth = std::max(max_depth, depth);
    return true;
  }
  bool Node(const gdv::Node& n) {
    const int tk = KindOf(n.return_type());
    if (tk < 0) return Fail("type " + n.return_type().ToString());
    switch (n.kind()) {
      case NodeKind::kField: {
        auto& f = static_cast<const FieldNode&>(n);
        int idx = -1;
        for (size_t i = 0; i < schema.size(); i++)
          if (schema[i].name == f.field().name) idx = static_cast<int>(i);
        int slot = -1;
        for (size_t k = 0; k < plan.input_fields.size(); k++)
          if (plan.input_fields[k] == idx) slot = static_cast<int>(k);
        if (idx < 0 || slot < 0 || slot > 255) return Fail("field " + f.field().name + " has no input slot");
        const int flags = (plan.input_needs_values[slot] ? 1 : 0) | (plan.input_needs_validity[slot] ? 2 : 0);
        return Push() && Emit(kLoad, slot, tk, flags);
      }
      case NodeKind::kLiteral: {
        auto& l = static_cast<const LiteralNode&>(n);
        if (nlits >= kMaxLits) return Fail("more than " + std::to_string(kMaxLits) + " literals");
        uint64_t v = l.value().lo;
        switch (tk) {  // the interpreter's canonical 64-bit slot
          case kTBool: v &= 1; break;
          case kTI8: v = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int8_t>(v))); break;
          case kTU8: v &= 0xff; break;
          case kTI16: v = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int16_t>(v))); break;
          case kTU16: v &= 0xffff; break;
          case kTI32: v = static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(v))); break;
          case kTU32: case kTF32: v &= 0xffffffffull; break;
          default: break;
        }
        out->lits[nlits] = l.is_null() ? 0 : v;
        return Push() && Emit(kLit, nlits++, l.is_null() ? 1 : 0);
      }
      case NodeKind::kFunction: {
        auto& fn = static_cast<const FunctionNode&>(n);
        const std::string& name = fn.name();
        const size_t nargs = fn.children().size();
        std::vector<int> at(nargs);
        for (size_t i = 0; i < nargs; i++) {
          at[i] = KindOf(fn.children()[i]->return_type());
          if (at[i] < 0) return Fail("type " + fn.children()[i]->return_type().ToString());
        }
        auto args = [&]() {
          for (auto& c : fn.children())
            if (!Node(*c)) return false;
          return true;
        };
        const bool arith = name == "add" || name == "subtract" || name == "multiply";
        if (arith && nargs == 2 && at[0] == at[1] && at[0] == tk && tk != kTBool) {
          if (!args()) return false;
          depth--;
          return Emit(name == "add" ? kAdd : name == "subtract" ? kSub : kMul, 0, tk);
        }
        static const std::map<std::string, int> cmps = {{"equal", kEq}, {"eq", kEq}, {"same", kEq}, {"not_equal", kNe},
                                                        {"less_than", kLt}, {"less_than_or_equal_to", kLe},
                                                        {"greater_than", kGt}, {"greater_than_or_equal_to", kGe}};
        auto cmp = cmps.find(name);
        if (cmp != cmps.end() && nargs == 2 && at[0] == at[1] && tk == kTBool &&
            fn.children()[0]->return_type().id == fn.children()[1]->return_type().id) {
          if (!args()) return false;
          depth--;
          return Emit(kCmp, cmp->second, at[0]);
        }
        if (name == "not" && nargs == 1 && at[0] == kTBool && tk == kTBool) return args() && Emit(kNot);
        if ((name == "isnull" || name == "isnotnull") && nargs == 1 && tk == kTBool)
          return args() && Emit(name == "isnull" ? kIsNull : kIsNotNull);
        // numeric casts the registry holds (gdv_device_lib.hpp: castINT_int64, castBIGINT_int32 / _float32 / _float64,
        // castINT_float32 / _float64, castFLOAT4_int32 / _int64 / _float64, castFLOAT8_int32 / _int64 / _float32)
        if (nargs == 1) {
          const TypeId from = fn.children()[0]->return_type().id, to = n.return_type().id;
          const bool ok = (name == "castBIGINT" && to == kInt64 && (from == kInt32 || from == kFloat || from == kDouble)) ||
                          (name == "castINT" && to == kInt32 && (from == kInt64 || from == kFloat || from == kDouble)) ||
                          (name == "castFLOAT4" && to == kFloat && (from == kInt32 || from == kInt64 || from == kDouble)) ||
                          (name == "castFLOAT8" && to == kDouble && (from == kInt32 || from == kInt64 || from == kFloat));
          if (ok) return args() && Emit(kCast, at[0], tk);
        }
        return Fail("function " + name);
      }
      case NodeKind::kIf: {
        auto& f = static_cast<const IfNode&>(n);
        if (KindOf(f.condition()->return_type()) != kTBool || KindOf(f.then_node()->return_type()) != tk ||
            KindOf(f.else_node()->return_type()) != tk)
          return Fail("if / else over mixed types");
        if (!Node(*f.condition()) || !Node(*f.then_node()) || !Node(*f.else_node())) return false;
        depth -= 2;
        return Emit(kIf);
      }
      case NodeKind::kBoolean: {
        auto& b = static_cast<const BooleanNode&>(n);
        if (b.children().empty()) return Fail("empty AND / OR");
        for (size_t i = 0; i < b.children().size(); i++) {
          if (KindOf(b.children()[i]->return_type()) != kTBool) return Fail("AND / OR over a non-boolean");
          if (!Node(*b.children()[i])) return false;
          if (i > 0) {
            depth--;
            if (!Emit(b.op() == BooleanNode::kAnd ? kAnd2 : kOr2)) return false;
          }
        }
        return true;
      }
      default:
        return Fail("IN expression");
    }
  }
};

}  // namespace

bool BuildTier0Program(const Schema& schema, const std::vector<ExpressionPtr>& exprs, bool filter, const KernelPlan& plan,
                       tier0::Args* out, std::string* why) {
  std::memset(out, 0, sizeof(*out));
  Builder b{schema, plan, out, {}};
  auto fail = [&](const std::string& w) {
    if (why) *why = w.empty() ? b.why : w;
    return false;
  };
  if (plan.layout.total() > kMaxBlock) return fail("argument block of " + std::to_string(plan.layout.total()) + " bytes");
  if (plan.mode != SelectionMode::kNone) return fail("selection-mode plan");
  if (plan.has_varlen_input || plan.has_varlen_output || plan.string_skeleton || plan.wave_tiles) return fail("var-len plan");
  if (plan.opts.rows_word || plan.opts.cast_x86_indefinite) return fail("plan options outside tier 0");
  if (filter ? exprs.size() != 1 : exprs.size() != plan.output_types.size()) return fail("expression count");
  out->n_in = plan.layout.n_in;
  out->filter = filter ? 1 : 0;
  out->subtiles = plan.opts.subtiles;
  for (size_t e = 0; e < exprs.size(); e++) {
    const Node& root = *exprs[e]->root();
    if (!b.Node(root)) return fail("");
    if (filter) {
      if (KindOf(root.return_type()) != kTBool) return fail("condition is not boolean");
      if (!b.Emit(kFilterOut)) return fail("");
    } else {
      const int tk = KindOf(exprs[e]->result().type);
      if (tk < 0 || tk != KindOf(root.return_type()) || e > 255) return fail("output type " + exprs[e]->result().type.ToString());
      if (!b.Emit(kOut, static_cast<int>(e), tk)) return fail("");
    }
    b.depth--;
  }
  return true;
}


std::string DescribeTier0Program(const tier0::Args& prog) {
  static const char* kinds[] = {"bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64"};
  static const char* cmps[] = {"eq", "ne", "lt", "le", "gt", "ge"};
  std::string out;
  for (int pc = 0; pc < prog.ncode; pc++) {
    const uint32_t ins = prog.code[pc];
    const int op = ins & 0xff, a = (ins >> 8) & 0xff, b = (ins >> 16) & 0xff, c = (ins >> 24) & 0xff;
    auto kind = [&](int k) { return std::string(k >= 0 && k <= tier0::kTF64 ? kinds[k] : "?"); };
    switch (op) {
      case tier0::kLoad: out += "load in" + std::to_string(a) + " " + kind(b) + ((c & 1) ? "" : " (values unused)") + ((c & 2) ? "" : " (no validity)"); break;
      case tier0::kLit: out += "lit #" + std::to_string(a) + (b ? " null" : " = 0x" + [&] { char t[24]; snprintf(t, sizeof t, "%llx", static_cast<unsigned long long>(prog.lits[a])); return std::string(t); }()); break;
      case tier0::kAdd: out += "add " + kind(b); break;
      case tier0::kSub: out += "subtract " + kind(b); break;
      case tier0::kMul: out += "multiply " + kind(b); break;
      case tier0::kCmp: out += std::string("compare ") + (a <= tier0::kGe ? cmps[a] : "?") + " " + kind(b); break;
      case tier0::kCast: out += "cast " + kind(a) + " -> " + kind(b); break;
      case tier0::kNot: out += "not"; break;
      case tier0::kIsNull: out += "isnull"; break;
      case tier0::kIsNotNull: out += "isnotnull"; break;
      case tier0::kAnd2: out += "and"; break;
      case tier0::kOr2: out += "or"; break;
      case tier0::kIf: out += "if"; break;
      case tier0::kOut: out += "out" + std::to_string(a) + " " + kind(b); break;
      case tier0::kFilterOut: out += "filter"; break;
      default: out += "?"; break;
    }
    out += "\n";
  }
  return out;
}

static std::atomic<int64_t> g_tier0_launches{0};
int64_t Tier0Launches() { return g_tier0_launches.load(std::memory_order_relaxed); }
void CountTier0Launch() { g_tier0_launches.fetch_add(1, std::memory_order_relaxed); }

}  // namespace gdv
