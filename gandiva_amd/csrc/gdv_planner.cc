#include "gdv_planner.h"

#include "gdv_libtag.h"
#include "gdv_regex.h"
#include "gdv_runtime.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <regex>
#include <sstream>

namespace gdv {

// ------------------------------------------------------------------ options

CodegenOptions CodegenOptions::FromEnv() {
  CodegenOptions o;
  if (const char* s = std::getenv("GDV_U")) {
    // powers of two only: the index-emission kernel walks 64-word groups (gdv_kernels.hip)
    int u = std::max(1, std::min(16, atoi(s)));
    while (u & (u - 1)) u &= u - 1;
    o.subtiles = u;
    o.subtiles_forced = true;
  }
  if (const char* s = std::getenv("GDV_WAVES")) {
    o.waves = std::max(1, std::min(16, atoi(s)));
    o.waves_forced = true;
  }
  if (const char* s = std::getenv("GDV_NT")) o.nontemporal = atoi(s) != 0;
  if (const char* s = std::getenv("GDV_NTLOAD")) o.nt_loads = atoi(s) != 0;
  if (const char* s = std::getenv("GDV_NO_LDS_MIRROR")) o.lds_mirror = atoi(s) == 0;
  o.no_inline_string_args = std::getenv("GDV_NO_INLINE_STRING_ARGS") != nullptr;
  o.no_wave_shape = std::getenv("GDV_NO_WAVE_SHAPE") != nullptr;
  o.wave_bytefree_only = std::getenv("GDV_WAVE_BYTEFREE_ONLY") != nullptr;
  o.ablation = std::getenv("GDV_ABLATION") != nullptr;
  o.runtime_needles = std::getenv("GDV_RUNTIME_NEEDLES") != nullptr;
  o.prepass_rolled = std::getenv("GDV_PREPASS_ROLLED") != nullptr;
  o.no_sel_wave = std::getenv("GDV_NO_SEL_WAVE") != nullptr;
  if (const char* s = std::getenv("GDV_PREPASS_AHEAD")) o.prepass_ahead = atoi(s) != 0;
  if (const char* s = std::getenv("GDV_CAST_X86_INDEFINITE")) o.cast_x86_indefinite = atoi(s) != 0;
  if (const char* s = std::getenv("GDV_SWEEP_GROUP")) o.sweep_group = std::max(1, std::min(8, atoi(s)));
  if (const char* s = std::getenv("GDV_FP_EXPERIMENT")) o.fp_experiment = atoi(s);
  if (const char* s = std::getenv("GDV_FP_K")) o.fp_rounds = std::max(1, std::min(8, atoi(s)));
  if (const char* s = std::getenv("GDV_FP_WINDOW")) o.fp_window_bytes = std::max(0, std::min(16384, atoi(s)));
  return o;
}

std::string CodegenOptions::Key() const {
  return "u" + std::to_string(subtiles) + "w" + std::to_string(waves) + (nontemporal ? "nt" : "") +
         (nt_loads ? "ntl" : "") + (lds_mirror ? "" : "nm") + (subtiles_forced ? "U" : "") + (waves_forced ? "W" : "") +
         (no_inline_string_args ? "ni" : "") + (no_wave_shape ? "nw" : "") + (wave_bytefree_only ? "bf" : "") +
         (ablation ? "abl" : "") + (runtime_needles ? "rn" : "") + (prepass_rolled ? "pr" : "") + (rows_word ? "rw" : "") + (no_sel_wave ? "nsw" : "") + (prepass_ahead ? "" : "npa") + (fp_experiment ? "fx" + std::to_string(fp_experiment) : "") + (fp_rounds != 3 ? "k" + std::to_string(fp_rounds) : "") +
         (fp_window_bytes != 9984 ? "fw" + std::to_string(fp_window_bytes) : "") + (cast_x86_indefinite ? "xi" : "") + (sweep_group != 1 ? "sg" + std::to_string(sweep_group) : "");
}

// ------------------------------------------------------------------ validation

namespace {

// Ablation branches (GDV_ABL masks, set through GDV_RTC_OPT=-DGDV_ABL=<mask>) are experiment
// scaffolding: they are emitted only into kernels planned with GDV_ABLATION=1 in the environment of
// Make.  A product kernel's text does not contain them (round-3 verdict: 27 sites in every kernel).
thread_local bool tl_ablation = false;
struct AblationScope {
  bool prev;
  explicit AblationScope(bool on) : prev(tl_ablation) { tl_ablation = on; }
  ~AblationScope() { tl_ablation = prev; }
};
std::string AblNot(int bit) { return tl_ablation ? "!(GDV_ABL & " + std::to_string(bit) + ") && " : ""; }
std::string AblAnd(int bit) { return tl_ablation ? " && !(GDV_ABL & " + std::to_string(bit) + ")" : ""; }
std::string AblIf(int bit) { return tl_ablation ? "if (!(GDV_ABL & " + std::to_string(bit) + ")) " : ""; }
std::string AblSel(int bit, const std::string& on, const std::string& off) {
  return tl_ablation ? "((GDV_ABL & " + std::to_string(bit) + ") ? " + on + " : " + off + ")" : off;
}
std::string AblDefine() {
  return tl_ablation ? "#ifndef GDV_ABL\n#define GDV_ABL 0  // ablation mask for experiments; 0 = the product\n#endif\n" : "";
}

Status ValidateNode(const Schema& schema, const Node& node);

Status ValidateField(const Schema& schema, const FieldNode& n) {
  for (auto& f : schema) {
    if (f.name == n.field().name) {
      if (f.type != n.field().type) {
        return Status::ValidationError("Field definition in schema " + f.name + ": " +
                                       f.type.ToString() + " different from field in expression " +
                                       n.field().name + ": " + n.field().type.ToString());
      }
      return Status::OK();
    }
  }
  return Status::ValidationError("Field " + n.field().name + " not in schema.");
}

bool ResolveFunction(const FunctionNode& n, const FunctionDef** def, DataType* ret) {
  std::vector<DataType> params;
  for (auto& c : n.children()) params.push_back(c->return_type());
  const FunctionDef* d = FunctionRegistry::Get().Lookup(n.name(), params);
  if (d == nullptr) return false;
  *def = d;
  *ret = d->ret;
  if (d->flags & kDecimalResult) {
    DecimalOp op = DecimalOp::kAdd;
    if (n.name() == "subtract") op = DecimalOp::kSubtract;
    else if (n.name() == "multiply") op = DecimalOp::kMultiply;
    else if (n.name() == "divide") op = DecimalOp::kDivide;
    else if (n.name() == "mod") op = DecimalOp::kMod;
    *ret = DecimalResultType(op, params[0], params[1]);
  }
  return true;
}

Status ValidateFunction(const Schema& schema, const FunctionNode& n) {
  for (auto& c : n.children()) GDV_RETURN_NOT_OK(ValidateNode(schema, *c));
  const FunctionDef* def = nullptr;
  DataType ret;
  if (!ResolveFunction(n, &def, &ret)) {
    return Status::ValidationError("Function " + n.ToString() + " not supported yet. ");
  }
  if (ret != n.return_type()) {
    // decimal results declared by the caller win when only precision/scale differ
    if (!(ret.id == kDecimal128 && n.return_type().id == kDecimal128 &&
          !(def->flags & kDecimalResult))) {
      return Status::ValidationError("Function " + n.name() + " returns " + ret.ToString() +
                                     " but the expression declares " +
                                     n.return_type().ToString());
    }
  }
  if (def->flags & kPatternArg) {
    if (n.children().size() < 2 || n.children()[1]->kind() != NodeKind::kLiteral) {
      return Status::ValidationError("'" + n.name() + "' function requires a literal as the last parameter");
    }
  }
  if (def->flags & kDateFormatArg) {  // [to_date_holder.cc ToDateHolder::Make's two messages, as recalled]
    if (n.children()[1]->kind() != NodeKind::kLiteral)
      return Status::ValidationError("'" + n.name() + "' function requires a literal as the second parameter");
    if (n.children().size() == 3 && n.children()[2]->kind() != NodeKind::kLiteral)
      return Status::ValidationError("'" + n.name() + "' function requires a int literal as the third parameter");
  }
  return Status::OK();
}

Status ValidateNode(const Schema& schema, const Node& node) {
  switch (node.kind()) {
    case NodeKind::kField:
      return ValidateField(schema, static_cast<const FieldNode&>(node));
    case NodeKind::kLiteral:
      return Status::OK();
    case NodeKind::kFunction:
      return ValidateFunction(schema, static_cast<const FunctionNode&>(node));
    case NodeKind::kIf: {
      auto& n = static_cast<const IfNode&>(node);
      GDV_RETURN_NOT_OK(ValidateNode(schema, *n.condition()));
      GDV_RETURN_NOT_OK(ValidateNode(schema, *n.then_node()));
      GDV_RETURN_NOT_OK(ValidateNode(schema, *n.else_node()));
      if (n.condition()->return_type().id != kBool)
        return Status::ValidationError("condition must be of boolean type, found type " +
                                       n.condition()->return_type().ToString());
      if (n.then_node()->return_type() != n.return_type())
        return Status::ValidationError("return type of if " + n.return_type().ToString() +
                                       " and then " + n.then_node()->return_type().ToString() +
                                       " not matching.");
      if (n.else_node()->return_type() != n.return_type())
        return Status::ValidationError("return type of if " + n.return_type().ToString() +
                                       " and else " + n.else_node()->return_type().ToString() +
                                       " not matching.");
      return Status::OK();
    }
    case NodeKind::kBoolean: {
      auto& n = static_cast<const BooleanNode&>(node);
      if (n.children().size() < 2)
        return Status::ValidationError("Boolean expression has " +
                                       std::to_string(n.children().size()) +
                                       " children, expected atleast two");
      for (auto& c : n.children()) {
        GDV_RETURN_NOT_OK(ValidateNode(schema, *c));
        if (c->return_type().id != kBool)
          return Status::ValidationError("Boolean expression has a child with return type " +
                                         c->return_type().ToString() + ", expected return type boolean");
      }
      return Status::OK();
    }
    case NodeKind::kIn: {
      auto& n = static_cast<const InNode&>(node);
      GDV_RETURN_NOT_OK(ValidateNode(schema, *n.eval()));
      if (n.eval()->return_type() != n.value_type())
        // message fragment pinned by test_gandiva.py:160-161
        return Status::ValidationError("Evaluation expression for IN clause returns " +
                                       n.eval()->return_type().ToString() +
                                       " values are of type" + n.value_type().ToString());
      return Status::OK();
    }
  }
  return Status::OK();
}

}  // namespace

Status ValidateExpression(const Schema& schema, const Expression& expr) {
  if (!expr.root()) return Status::ValidationError("Root node cannot be null");
  GDV_RETURN_NOT_OK(ValidateNode(schema, *expr.root()));
  if (expr.root()->return_type() != expr.result().type) {
    return Status::ValidationError("Return type of root node " +
                                   expr.root()->return_type().ToString() +
                                   " does not match that of expression " +
                                   expr.result().type.ToString());
  }
  return Status::OK();
}

// ------------------------------------------------------------------ code generation

namespace {

std::string Hex64(uint64_t v) {
  char buf[32];
  snprintf(buf, sizeof(buf), "0x%llxull", static_cast<unsigned long long>(v));
  return buf;
}

// The kernel's identity is its code: the "// @expr_N = ..." header lines render the expressions
// WITH their literal values (DumpIR shows them), the code below them does not depend on the values.
std::string HashableSource(const std::string& text) {
  std::string out;
  size_t pos = 0;
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos);
    if (eol == std::string::npos) eol = text.size();
    if (text.compare(pos, 9, "// @expr_") != 0) out.append(text, pos, eol - pos + 1);
    pos = eol + 1;
  }
  return out;
}

uint64_t Fnv1a(const std::string& s) { return Fnv1a64(s); }

// A kernel is its generated text AND the device functions that text reaches in the library
// (gdv_libtag.h): their hash is part of the kernel name, so a PMC pass or a cached code object can
// only be attributed to the code that really ran — and an edit of a function the kernel never
// calls leaves its name alone (rounds 1-2 hashed the whole header: any edit renamed every kernel).
std::string LibraryTag(const std::string& kernel_text) { return LibraryIndex::Embedded().TagFor(kernel_text); }

// A value inside the generated row body: a C++ expression plus its validity, split the way
// the reference's ValueValidityPair splits it — the set of input columns whose validity
// words intersect, and an optional per-lane predicate for value-dependent validity
// (if/else, SQL three-valued AND/OR, functions that produce nulls themselves).
struct Val {
  std::string v;
  DataType type;
  std::set<int> vcols;
  std::string vlane;
  // concat results are not a view: they are the list of their argument views, written one
  // after the other by the output copy (piece expression, per-lane "piece present"
  // predicate or "" for always).  Only an output expression or another concat can take one.
  std::vector<std::pair<std::string, std::string>> pieces;
  // A string value that IS the row of input slot `col_slot` (whole, unsliced), read through the
  // static byte map `col_map` (0 none, 1 upper, 2 lower): candidates for the byte-parallel
  // paths (sweep-answered '%needle%', flat output copy).  -1: anything else.
  int col_slot = -1;
  int col_map = 0;
  // reverse(), replace() and castVARCHAR(integer) results are not readable views (GDV_MAP_REVERSE /
  // GDV_MAP_REPLACE / GDV_MAP_DIGITS): like concat results, only the output copy or a concat can
  // take them (anything else gets them through a first stage, StageMaterialisedValues)
  bool opaque = false;
  bool never_null() const { return vcols.empty() && vlane.empty(); }
};

// '%needle%' predicate answered by the byte sweep of input slot `slot` (bytes read through `map`)
struct ContainsHook {
  int slot;
  int map;
  std::string needle;
};

// One var-len output of a projector: its row value as 1+ pieces (views written back to back),
// each with the name of the per-sub-tile register array holding it.
struct VarlenOut {
  int e = 0;                      // output index
  int flat_slot = -1;             // >= 0: the row is input slot flat_slot's whole row ...
  int flat_map = 0;               // ... read through this byte map
  int window = -1;                // >= 0: LDS staging window of this output (non-flat outputs)
  int segment = -1;               // wave shape: index of this output's array of wave-tile totals / bases
  bool reads_views = false;       // staged output whose copies read readable views (candidates for the LDS mirror)
};

class CodeGen {
 public:
  CodeGen(const Schema& schema, SelectionMode mode, const CodegenOptions& opts)
      : schema_(schema), sel_mode_(mode), opts_(opts) {}

  bool selection() const { return sel_mode_ != SelectionMode::kNone; }

  Status Gen(const Node& node, const std::string& active, Val* out);

  // ---- emission helpers
  std::string Tmp(const std::string& ctype, const std::string& rhs) {
    std::string key = ctype + "|" + rhs;
    auto it = cse_.find(key);
    if (it != cse_.end()) return it->second;
    std::string name = "t" + std::to_string(next_tmp_++);
    body_ << "      const " << ctype << " " << name << " = " << rhs << ";\n";
    cse_[key] = name;
    return name;
  }
  void Stmt(const std::string& s) { body_ << "      " << s << "\n"; }

  // conjunction of per-lane predicates; "" stands for "always true"
  static std::string AndExpr(const std::string& a, const std::string& b) {
    if (a.empty() || a == "true") return (b == "true") ? "" : b;
    if (b.empty() || b == "true") return a;
    return "(" + a + " && " + b + ")";
  }

  // same, spelled out: never the empty string (for use as a full expression)
  static std::string AndFull(const std::string& a, const std::string& b) {
    std::string r = AndExpr(a, b);
    return r.empty() ? "true" : r;
  }

  // per-lane validity of a value ("true" when it can never be null)
  std::string LaneValid(const Val& val) {
    std::string cols;
    if (!val.vcols.empty()) {
      if (selection()) {
        for (int k : val.vcols) cols = AndExpr(cols, "b" + std::to_string(k) + "[u]");
      } else {
        cols = Tmp("bool", "gdv_lane_bit(" + WordExpr(val.vcols) + ", lane)");
      }
    }
    std::string r = AndExpr(cols, val.vlane);
    return r.empty() ? "true" : r;
  }

  // wave-uniform AND of the validity words of a set of input columns (row mode only)
  std::string WordExpr(const std::set<int>& cols) {
    if (cols.empty()) return "~0ull";
    std::string s;
    for (int k : cols) {
      if (!s.empty()) s += " & ";
      s += "v" + std::to_string(k);
    }
    if (cols.size() > 1) s = Tmp("gdv_uint64", s);
    return s;
  }

  // ---- literals are kernel ARGUMENTS, not source text (round 2): `a > 499` and `a > 500`, or
  // like '%spark%' and like '%flink%', share one compiled kernel; only the shape (types, list
  // sizes, pattern form and needle length) is compiled in.
  // Fixed-width literal -> 8-byte slot of gdv_args::lit (decimal128: two slots, low word first)
  // (slots are never shared by VALUE — the code must not depend on which constants happen to be
  // equal — only by node identity: a literal node used in several places is one slot, so common
  // sub-expressions built from shared nodes still merge)
  int LitSlot(uint64_t v) {
    lits_.push_back(v);
    return static_cast<int>(lits_.size()) - 1;
  }
  std::string LiteralExpr(const DataType& t, const Literal& v, const void* node) {
    auto it = lit_of_node_.find(node);
    if (it != lit_of_node_.end()) return it->second;
    std::string e = LiteralExprNew(t, v);
    lit_of_node_[node] = e;
    return e;
  }
  static std::string InlineLiteral(const DataType& t, const Literal& v) {
    switch (t.id) {
      case kBool: return v.lo ? "true" : "false";
      case kFloat: return "__uint_as_float(" + Hex64(v.lo & 0xffffffffull) + ")";
      case kDouble: return "__longlong_as_double((long long)" + Hex64(v.lo) + ")";
      case kDecimal128: return "gdv_make_int128(" + Hex64(v.hi) + ", " + Hex64(v.lo) + ")";
      default: {
        uint64_t mask = t.byte_width() >= 8 ? ~0ull : ((1ull << (8 * t.byte_width())) - 1);
        return "((" + t.CType() + ")" + Hex64(v.lo & mask) + ")";
      }
    }
  }
  std::string LiteralExprNew(const DataType& t, const Literal& v) {
    auto slot = [&](uint64_t x) { return "A.lit[" + std::to_string(LitSlot(x)) + "]"; };
    switch (t.id) {
      case kBool: return v.lo ? "true" : "false";
      case kFloat: return "__uint_as_float((gdv_uint32)" + slot(v.lo & 0xffffffffull) + ")";
      case kDouble: return "__longlong_as_double((long long)" + slot(v.lo) + ")";
      case kDecimal128: {
        // two consecutive slots that are never shared with single-word literals
        lits_.push_back(v.lo);
        lits_.push_back(v.hi);
        const std::string i = std::to_string(lits_.size() - 2), j = std::to_string(lits_.size() - 1);
        return "gdv_make_int128(A.lit[" + j + "], A.lit[" + i + "])";
      }
      default: {
        uint64_t mask = t.byte_width() >= 8 ? ~0ull : ((1ull << (8 * t.byte_width())) - 1);
        return "((" + t.CType() + ")" + slot(v.lo & mask) + ")";
      }
    }
  }
  // bytes -> the plan's constant block (device memory, bound through gdv_args::aux0); returns a
  // pointer expression.  Every table starts 16-byte aligned and is readable 8 bytes past its end.
  std::string ByteTable(const std::string& bytes, const char* ctype = "gdv_uint8") {
    while (blob_.size() % 16 != 0) blob_.push_back('\0');
    const size_t off = blob_.size();
    blob_ += bytes;
    blob_.append(8, '\0');  // 8-byte loads may run past the table's end
    return "((const " + std::string(ctype) + "*)(gdv_cst + " + std::to_string(off) + "))";
  }
  std::string StringConstant(const std::string& bytes) {
    std::string t = ByteTable(bytes);
    bool ascii = true;
    for (unsigned char c : bytes) ascii = ascii && c < 0x80;
    return "gdv_make_str(" + t + ", 0, " + std::to_string(bytes.size()) + ", " + t + " + " +
           std::to_string(bytes.size() + 8) + (ascii ? ", GDV_STR_ASCII | GDV_STR_INBUF)" : ", GDV_STR_INBUF)");
  }
  // SQL LIKE pattern -> (literal bytes, token kinds); `escape` < 0 means no escape character
  static Status CompileLike(const std::string& pat, int escape, std::string* bytes,
                            std::string* kinds) {
    for (size_t i = 0; i < pat.size(); i++) {
      unsigned char c = static_cast<unsigned char>(pat[i]);
      if (escape >= 0 && c == static_cast<unsigned char>(escape)) {
        if (i + 1 >= pat.size())
          return Status::Invalid("like pattern must not end with the escape character");
        unsigned char nx = static_cast<unsigned char>(pat[i + 1]);
        if (nx != '%' && nx != '_' && nx != static_cast<unsigned char>(escape))
          return Status::Invalid("invalid escape sequence in like pattern");
        bytes->push_back(static_cast<char>(nx));
        kinds->push_back(0);
        i++;
      } else if (c == '%') {
        if (kinds->empty() || kinds->back() != 2) {  // collapse runs of %
          bytes->push_back(0);
          kinds->push_back(2);
        }
      } else if (c == '_') {
        bytes->push_back(0);
        kinds->push_back(1);
      } else {
        bytes->push_back(static_cast<char>(c));
        kinds->push_back(0);
      }
    }
    return Status::OK();
  }

  // input slots of every var-len field below `node`
  std::set<int> StringSlotsOf(const Node& node) {
    std::set<int> r;
    std::function<void(const Node&)> walk = [&](const Node& n) {
      switch (n.kind()) {
        case NodeKind::kField: {
          auto& f = static_cast<const FieldNode&>(n);
          if (f.return_type().is_varlen()) r.insert(SlotFor(f, true, true));
          break;
        }
        case NodeKind::kFunction:
          for (auto& c : static_cast<const FunctionNode&>(n).children()) walk(*c);
          break;
        case NodeKind::kIf: {
          auto& i = static_cast<const IfNode&>(n);
          walk(*i.condition()); walk(*i.then_node()); walk(*i.else_node());
          break;
        }
        case NodeKind::kBoolean:
          for (auto& c : static_cast<const BooleanNode&>(n).children()) walk(*c);
          break;
        case NodeKind::kIn: walk(*static_cast<const InNode&>(n).eval()); break;
        default: break;
      }
    };
    walk(node);
    return r;
  }

  int SlotFor(const FieldNode& f, bool values, bool validity) {
    int idx = -1;
    for (size_t i = 0; i < schema_.size(); i++)
      if (schema_[i].name == f.field().name) idx = static_cast<int>(i);
    int slot;
    auto it = slot_of_field_.find(idx);
    if (it == slot_of_field_.end()) {
      slot = static_cast<int>(input_fields_.size());
      slot_of_field_[idx] = slot;
      input_fields_.push_back(idx);
      needs_values_.push_back(false);
      needs_validity_.push_back(false);
    } else {
      slot = it->second;
    }
    if (values) needs_values_[slot] = true;
    if (validity) needs_validity_[slot] = true;
    return slot;
  }

  const Schema& schema_;
  SelectionMode sel_mode_;
  CodegenOptions opts_;
  int compact_from_ = 0x7fffffff;  // schema fields from this index on are compact temporaries (selection mode)
  bool no_hooks_ = false;          // pre-pass kernels have no byte sweep: '%needle%' takes the per-row search
  // Wave kernels, round 4: false = the OPTIMISTIC variant (views of swept columns carry GDV_STR_ASCII as
  // a compile-time fact; a byte >= 0x80 raises NOTASCII); true = the EXACT variant the host re-runs such
  // a batch on: the flag is what the byte sweep of the (sub-)tile found, in the pre-pass and in the main
  // kernel alike, and a tile that did hold a byte >= 0x80 reports GDV_ERR_SAWUTF8 (so the host knows
  // when a later batch may go back to the optimistic kernels).
  bool exact_ascii_ = false;
  // selection-mode wave main kernel (round 5): the pre-pass took its lengths from the offsets under the ASCII
  // assumption; the rows, which read their bytes here anyway, verify it (NOTASCII -> the general kernel)
  bool sel_ascii_check_ = false;
  std::set<int> row_ascii_slots_;  // exact variant: inputs whose views take a PER-ROW flag (gdv_with_lead)
  bool bake_needles_ = false;      // wave kernels: '%needle%' bytes are immediates of the kernel text (NeedleConstants)
  bool unroll_rows_ = false;       // the row loop of this kernel is unrolled (small bodies that index registers by u)
  // ... and their lead-byte mask from the sweep's continuation bitmap: LDS bitmap index (behind the hooks' bitmaps)
  int CbIndex(int slot) {
    int j = 0;
    for (int k : ascii_slots_) { if (k == slot) return static_cast<int>(contains_hooks_.size()) + j; j++; }
    return -1;
  }
  int mirror_slot_ = -1;           // wave kernels: the var-len input whose sub-tile spans are swept one at a time
                                   // (main kernel: and mirrored in LDS)
  bool replace_hits_ = false;      // wave kernels: replace() over a whole column row may be answered by the sweep
  int replace_hook_ = -1;          // ... the hook (match bitmap) that does
  std::ostringstream body_;
  std::map<std::string, std::string> cse_;
  int next_tmp_ = 0;
  std::map<int, int> slot_of_field_;
  std::vector<int> input_fields_;
  std::vector<bool> needs_values_, needs_validity_;
  bool can_raise_ = false;
  std::vector<uint64_t> lits_;        // gdv_args::lit
  std::map<const void*, std::string> lit_of_node_;
  std::string blob_;                  // constant block: string literals, patterns, IN tables
  // string plans
  std::vector<ContainsHook> contains_hooks_;
  std::vector<std::string> hook_tables_;  // needle bytes in the constant block
  std::set<int> ascii_slots_;     // input slots whose tile-wide ASCII flag some function consults
  std::vector<VarlenOut> varlen_outs_;
  int HookFor(int slot, int map, const std::string& needle) {
    for (size_t h = 0; h < contains_hooks_.size(); h++)
      if (contains_hooks_[h].slot == slot && contains_hooks_[h].map == map && contains_hooks_[h].needle == needle)
        return static_cast<int>(h);
    contains_hooks_.push_back({slot, map, needle});
    hook_tables_.push_back(ByteTable(needle));
    return static_cast<int>(contains_hooks_.size()) - 1;
  }
};

// functions whose fast path is "the string is pure ASCII" (character index == byte index)
bool WantsAsciiHint(const std::string& name) {
  static const std::set<std::string> k = {"substr", "substring", "left", "right", "char_length", "length",
                                          "lengthUtf8", "castVARCHAR", "locate", "position", "strpos", "like",
                                          "reverse", "lpad", "rpad"};
  return k.count(name) != 0;
}

Status CodeGen::Gen(const Node& node, const std::string& active, Val* out) {
  switch (node.kind()) {
    case NodeKind::kField: {
      auto& f = static_cast<const FieldNode&>(node);
      int slot = SlotFor(f, true, true);
      out->type = f.return_type();
      std::string k = std::to_string(slot);
      if (f.return_type().is_varlen()) {
        out->v = "s" + k;  // per-iteration view built from the two offsets (row phase prologue)
        out->col_slot = slot;
        out->col_map = 0;
      } else if (f.return_type().id == kBool) {
        out->v = selection() ? "x" + k + "[u]" : Tmp("bool", "gdv_lane_bit(d" + k + ", lane)");
      } else {
        out->v = "c" + k + "[u]";
      }
      out->vcols = {slot};
      out->vlane.clear();
      return Status::OK();
    }
    case NodeKind::kLiteral: {
      auto& l = static_cast<const LiteralNode&>(node);
      out->type = l.return_type();
      out->col_slot = -1;
      if (l.return_type().is_varlen()) {
        out->v = StringConstant(l.value().bytes);
        out->vcols.clear();
        out->vlane = l.is_null() ? "false" : "";
        return Status::OK();
      }
      out->v = LiteralExpr(l.return_type(), l.value(), &node);
      out->vcols.clear();
      out->vlane = l.is_null() ? "false" : "";
      return Status::OK();
    }
    case NodeKind::kFunction: {
      auto& fn = static_cast<const FunctionNode&>(node);
      const FunctionDef* def = nullptr;
      DataType ret;
      if (!ResolveFunction(fn, &def, &ret))
        return Status::CodeGenError("Function " + fn.ToString() + " not supported yet. ");
      // Integer literals handed to a function over strings (substr positions, left / right
      // counts, castVARCHAR lengths ...) are part of the query's SHAPE: compiled in, so the
      // position arithmetic folds (C5: +0.2 ms when they were kernel arguments).  Everything
      // else — comparison constants, arithmetic operands, IN lists, LIKE needles — is an argument.
      bool over_strings = false;
      for (auto& c : fn.children()) over_strings |= c->return_type().is_varlen();
      std::vector<Val> args(fn.children().size());
      for (size_t i = 0; i < args.size(); i++) {
        const Node& child = *fn.children()[i];
        if (over_strings && child.kind() == NodeKind::kLiteral && !child.return_type().is_varlen() &&
            !opts_.no_inline_string_args) {
          auto& l = static_cast<const LiteralNode&>(child);
          args[i].type = l.return_type();
          args[i].v = InlineLiteral(l.return_type(), l.value());
          args[i].vlane = l.is_null() ? "false" : "";
          continue;
        }
        GDV_RETURN_NOT_OK(Gen(child, active, &args[i]));
      }
      if (fn.name() == "regexp_like" || fn.name() == "regexp_matches") {
        // (patterns that are a plain literal became `like` when the tree was built: gdv_node.cc MakeFunctionNode)  Round 5, late:
        // the pattern is compiled to a position automaton here, at Make time; the row walks it with one 64-bit state set
        auto& pat = static_cast<const LiteralNode&>(*fn.children()[1]);
        out->type = fn.return_type();
        out->vcols = args[0].vcols;
        out->vlane = args[0].vlane;
        out->pieces.clear();
        out->col_slot = -1;
        out->col_map = 0;
        out->opaque = false;
        if (pat.is_null()) {
          out->v = "false";
          out->vlane = "false";
          return Status::OK();
        }
        if (!args[0].pieces.empty() || args[0].opaque)
          return Status::CodeGenError("Function " + fn.ToString() + " not supported yet: a concat / lpad / rpad / reverse / replace / "
                                      "castVARCHAR(number) result can only be an output expression or an argument of concat in the HIP backend. ");
        std::string table;
        GDV_RETURN_NOT_OK(CompileRegex(pat.value().bytes, &table));
        out->v = Tmp("bool", "gdv_regex_search(" + args[0].v + ", " + ByteTable(table) + ")");
        return Status::OK();
      }
      if (fn.name().compare(0, 7, "regexp_") == 0)
        return Status::CodeGenError("Function " + fn.ToString() + " not supported yet: the HIP backend takes regexp_replace "
                                    "with a literal pattern and a replacement without backslashes only (no metacharacters, "
                                    "no '%' or '_'). ");
      out->type = fn.return_type();
      out->vcols.clear();
      out->vlane.clear();
      out->pieces.clear();
      out->col_slot = -1;
      out->col_map = 0;
      const bool digest = fn.name().compare(0, 4, "hash") == 0 ? fn.return_type().is_varlen()
                                                                : (fn.name() == "sha256" || fn.name() == "sha1" || fn.name() == "sha" || fn.name() == "md5");
      out->opaque = fn.name() == "reverse" || fn.name() == "replace" || fn.name() == "initcap" || digest ||
                    (fn.name() == "castVARCHAR" && !args[0].type.is_varlen());
      if ((fn.name() == "upper" || fn.name() == "lower") && args.size() == 1 && args[0].col_slot >= 0) {
        out->col_slot = args[0].col_slot;
        out->col_map = fn.name() == "upper" ? 1 : 2;
      }
      if (WantsAsciiHint(fn.name()))
        for (size_t i = 0; i < args.size(); i++)
          if (args[i].type.is_varlen())
            for (int k : StringSlotsOf(*fn.children()[i])) ascii_slots_.insert(k);
      const std::string ctype = out->type.CType();
      const bool is_concat = fn.name() == "concat" || fn.name() == "concatOperator";
      for (auto& a : args)
        if ((!a.pieces.empty() || a.opaque) && !is_concat)
          return Status::CodeGenError("Function " + fn.ToString() +
                                      " not supported yet: a concat / lpad / rpad / reverse / replace / castVARCHAR(number) "
                                      "result can only be an output expression or an argument of concat in the "
                                      "HIP backend. ");
      if (fn.name() == "replace") {
        // replace(text, from, to) with LITERAL from / to: a table in the constant block; the result
        // is materialised by the output copy (GDV_MAP_REPLACE)
        if (fn.children()[1]->kind() != NodeKind::kLiteral || fn.children()[2]->kind() != NodeKind::kLiteral) {
          // round 5: from / to that are not both literals — the same rule with the arguments read through their own views,
          // per row (byte loops: a registry-tail path; literal arguments keep the table and the sweep's match bits)
          std::string lanes;
          for (auto& a : args) {
            out->vcols.insert(a.vcols.begin(), a.vcols.end());
            lanes = AndExpr(lanes, a.vlane);
          }
          out->vlane = lanes;
          can_raise_ = true;
          const std::string guard = AndExpr(AndExpr("live", active), LaneValid(*out));
          out->v = Tmp("gdv_str", guard + " ? gdv_replace_row(ctx, " + args[0].v + ", " + args[1].v + ", " + args[2].v + ") : gdv_empty_str()");
          return Status::OK();
        }
        auto& lf = static_cast<const LiteralNode&>(*fn.children()[1]);
        auto& lt = static_cast<const LiteralNode&>(*fn.children()[2]);
        out->vcols = args[0].vcols;
        if (lf.is_null() || lt.is_null()) {
          out->opaque = false;
          out->vlane = "false";
          out->v = "gdv_empty_str()";
          return Status::OK();
        }
        const std::string& from = lf.value().bytes;
        const std::string& to = lt.value().bytes;
        std::string tab(16, '\0');
        const int32_t fl = static_cast<int32_t>(from.size()), tl = static_cast<int32_t>(to.size());
        std::memcpy(&tab[0], &fl, 4);
        std::memcpy(&tab[4], &tl, 4);
        tab += from;
        tab.append((16 - from.size() % 16) % 16, '\0');
        tab += to;
        out->vlane = args[0].vlane;
        can_raise_ = true;
        const std::string guard = AndExpr(AndExpr("live", active), LaneValid(*out));
        // A 'from' that cannot overlap itself (no proper prefix is a suffix), over a whole column
        // row: the byte sweep marks the match positions of the sub-tile's span (the '%needle%'
        // machinery); the row counts its own bits, the copy walks them.  One such needle per
        // kernel; spans too long for the bitmap (wave-uniform) search per row as before.
        bool self_overlap = false;
        for (size_t k = 1; k < from.size(); k++) self_overlap |= from.compare(0, from.size() - k, from, k, from.size() - k) == 0;
        if (replace_hits_ && !selection() && args[0].col_slot >= 0 && from.size() >= 2 && from.size() <= 8 && !self_overlap) {
          int h = -1;
          for (size_t i = 0; i < contains_hooks_.size(); i++)
            if (contains_hooks_[i].slot == args[0].col_slot && contains_hooks_[i].map == args[0].col_map && contains_hooks_[i].needle == from)
              h = static_cast<int>(i);
          if (replace_hook_ < 0 || replace_hook_ == h) {
            const std::string K = std::to_string(args[0].col_slot), table = ByteTable(tab);
            if (h < 0) {
              // (the needle's bytes are in the replace table itself, 16 bytes in: no table of its own,
              // so the scanner-shaped fallback — which has no such hook — lays out the same constants)
              contains_hooks_.push_back({args[0].col_slot, args[0].col_map, from});
              hook_tables_.push_back("(" + table + " + 16)");
              h = static_cast<int>(contains_hooks_.size()) - 1;
            }
            replace_hook_ = h;
            out->v = Tmp("gdv_str", guard + " ? (hm_ok" + K + " ? gdv_replace_hits(ctx, " + args[0].v + ", " + table + ", hit" +
                                        std::to_string(h) + ", oa" + K + "[u] - sb" + K + ") : gdv_replace(ctx, " + args[0].v + ", " +
                                        table + ")) : gdv_empty_str()");
            return Status::OK();
          }
        }
        out->v = Tmp("gdv_str", guard + " ? gdv_replace(ctx, " + args[0].v + ", " + ByteTable(tab) + ") : gdv_empty_str()");
        return Status::OK();
      }
      if (fn.name() == "lpad" || fn.name() == "rpad") {
        // lpad / rpad(text, n[, fill]) with LITERAL n and fill: two pieces (device library), the
        // fill repeated to n characters laid out once in the constant block
        const Node& nn = *fn.children()[1];
        const Node* fl = fn.children().size() == 3 ? fn.children()[2].get() : nullptr;
        if (nn.kind() != NodeKind::kLiteral || (fl != nullptr && fl->kind() != NodeKind::kLiteral)) {
          // round 5: a length or a fill that is not a literal — the fill is read cyclically through its own view, per row
          std::string lanes;
          for (auto& a : args) {
            out->vcols.insert(a.vcols.begin(), a.vcols.end());
            lanes = AndExpr(lanes, a.vlane);
          }
          can_raise_ = true;
          const std::string fillv = fl != nullptr ? args[2].v : Tmp("gdv_str", StringConstant(" "));
          const std::string guard = AndExpr(AndExpr("live", active), AndExpr(lanes, LaneValid(args[0])));
          const std::string want = Tmp("gdv_int32", guard + " ? (gdv_int32)" + args[1].v + " : 0");
          const std::string text = Tmp("gdv_str", "gdv_pad_text(" + args[0].v + ", " + want + ")");
          const std::string pad = Tmp("gdv_str", "gdv_pad_fill_row(ctx, " + args[0].v + ", " + want + ", " + fillv + ")");
          if (fn.name() == "lpad") {
            out->pieces.emplace_back(pad, "");
            out->pieces.emplace_back(text, "");
          } else {
            out->pieces.emplace_back(text, "");
            out->pieces.emplace_back(pad, "");
          }
          out->vlane = lanes;
          out->v = "gdv_empty_str()";  // never read: consumers use the pieces
          return Status::OK();
        }
        auto& nl = static_cast<const LiteralNode&>(nn);
        const bool null_lit = nl.is_null() || (fl != nullptr && static_cast<const LiteralNode*>(fl)->is_null());
        const int32_t n = null_lit ? 0 : static_cast<int32_t>(nl.value().lo);
        if (n > (1 << 16))
          return Status::CodeGenError("Function " + fn.ToString() +
                                      " not supported yet: pad lengths above 65536 characters. ");
        const std::string fill = fl != nullptr ? static_cast<const LiteralNode*>(fl)->value().bytes : " ";
        // characters of the fill = runs starting at a non-continuation byte
        std::vector<std::string> chars;
        for (unsigned char c : fill) {
          if (chars.empty() || (c & 0xC0) != 0x80) chars.emplace_back();
          chars.back().push_back(static_cast<char>(c));
        }
        std::string tab;
        bool ascii = true;
        for (int32_t k = 0; k < n && !chars.empty(); k++) tab += chars[k % chars.size()];
        for (unsigned char c : tab) ascii = ascii && c < 0x80;
        const std::string N = std::to_string(n);
        const std::string text = Tmp("gdv_str", "gdv_pad_text(" + args[0].v + ", " + N + ")");
        const std::string pad = Tmp("gdv_str", "gdv_pad_fill(" + args[0].v + ", " + N + ", " + ByteTable(tab) + ", " +
                                                   std::to_string(tab.size()) + ", " + (ascii ? "true" : "false") + ")");
        if (fn.name() == "lpad") {
          out->pieces.emplace_back(pad, "");
          out->pieces.emplace_back(text, "");
        } else {
          out->pieces.emplace_back(text, "");
          out->pieces.emplace_back(pad, "");
        }
        out->vcols = args[0].vcols;
        out->vlane = null_lit ? "false" : args[0].vlane;
        out->v = "gdv_empty_str()";  // never read: consumers use the pieces
        return Status::OK();
      }
      if (is_concat) {
        // concat: a null argument is the empty string, the result is never null;
        // concatOperator (||): null if any argument is null
        const bool never_null = fn.name() == "concat";
        std::string lanes;
        for (auto& a : args) {
          const std::string present = never_null ? LaneValid(a) : "";
          if (a.pieces.empty()) {
            out->pieces.emplace_back(a.v, present == "true" ? "" : present);
          } else {
            for (auto& pc : a.pieces) {
              std::string pv = AndExpr(pc.second, present);
              out->pieces.emplace_back(pc.first, pv);
            }
          }
          if (!never_null) {
            out->vcols.insert(a.vcols.begin(), a.vcols.end());
            lanes = AndExpr(lanes, a.vlane);
          }
        }
        out->vlane = lanes;
        out->v = "gdv_empty_str()";  // never read: consumers use the pieces
        return Status::OK();
      }
      if (def->flags & kDateFormatArg) {
        // to_date(s, 'pattern'[, suppress_errors]): the pattern becomes one byte per strptime directive here, at Make
        // time, the way the reference's ToDateHolder converts it once per expression; the row interprets it
        auto& pat = static_cast<const LiteralNode&>(*fn.children()[1]);
        if (pat.is_null()) return Status::Invalid("Invalid date format: null");
        int suppress = 0;
        if (fn.children().size() == 3) {
          auto& sl = static_cast<const LiteralNode&>(*fn.children()[2]);
          suppress = !sl.is_null() && static_cast<int32_t>(sl.value().lo) == 1 ? 1 : 0;
        }
        std::string ops;
        GDV_RETURN_NOT_OK(CompileDateFormat(pat.value().bytes, &ops));
        can_raise_ = true;
        const std::string ov = "ov" + std::to_string(next_tmp_++);
        Stmt("bool " + ov + " = false;");
        const std::string guard = AndExpr(AndExpr("live", active), LaneValid(args[0]));
        out->v = Tmp("gdv_int64", guard + " ? gdv_parse_date(ctx, " + args[0].v + ", " + ByteTable(ops) + ", " + std::to_string(ops.size()) + ", " +
                                      std::to_string(suppress) + ", true, &" + ov + ") : (gdv_int64)0");
        out->vlane = ov;
        return Status::OK();
      }
      if (def->flags & kPatternArg) {
        // like(s, 'pattern'[, 'escape']): the pattern is compiled here, at Make time, the way
        // the reference's LikeHolder compiles it to a regex once per expression
        auto& pat = static_cast<const LiteralNode&>(*fn.children()[1]);
        int escape = -1;
        if (fn.children().size() == 3) {
          if (fn.children()[2]->kind() != NodeKind::kLiteral)
            return Status::ValidationError("'like' function requires a literal as the escape character");
          auto& esc = static_cast<const LiteralNode&>(*fn.children()[2]);
          if (esc.value().bytes.size() != 1)
            return Status::Invalid("The length of escape char in like function must be 1");
          escape = static_cast<unsigned char>(esc.value().bytes[0]);
        }
        if (pat.is_null()) {
          out->v = "false";
          out->vlane = "false";
          return Status::OK();
        }
        std::string pattern = pat.value().bytes;
        if (fn.name() == "ilike") {
          // case-insensitive: the pattern's ASCII letters are lowered here, the string is read through the
          // lower-case byte map (a whole-column argument stays a whole-column view: the sweep still answers '%needle%')
          for (auto& ch : pattern)
            if (ch >= 'A' && ch <= 'Z') ch = static_cast<char>(ch + 32);
          args[0].v = Tmp("gdv_str", "lower_utf8(" + args[0].v + ")");
          if (args[0].col_slot >= 0) args[0].col_map = 2;
        }
        std::string bytes, kinds;
        GDV_RETURN_NOT_OK(CompileLike(pattern, escape, &bytes, &kinds));
        out->vcols = args[0].vcols;
        out->vlane = args[0].vlane;
        // common shapes skip the general matcher: literal | literal% | %literal | %literal%
        const size_t nk = kinds.size();
        const bool lead = nk > 0 && kinds.front() == 2, trail = nk > 0 && kinds.back() == 2;
        const size_t lo = lead ? 1 : 0, hi = nk - ((trail && nk > lo) ? 1 : 0);
        bool plain = true;
        for (size_t i = lo; i < hi; i++) plain = plain && kinds[i] == 0;
        if (plain && !(nk == 1 && lead)) {
          const std::string lit = bytes.substr(lo, hi - lo);
          const char* fnname = lead && trail ? "gdv_like_contains" : lead ? "gdv_like_suffix"
                               : trail ? "gdv_like_prefix" : "gdv_like_equal";
          const std::string per_row = std::string(fnname) + "(" + args[0].v + ", " + ByteTable(lit) + ", " +
                                      std::to_string(lit.size()) + ")";
          if (lead && trail && lit.size() >= 2 && lit.size() <= 8 && args[0].col_slot >= 0 && !selection() && !no_hooks_) {
            // '%needle%' over a whole input row: the byte sweep has marked every match position
            // of the tile's span in an LDS bitmap; the row tests its own byte range.  Spans too
            // long for the bitmap (wave-uniform) take the per-row search.
            const int h = HookFor(args[0].col_slot, args[0].col_map, lit);
            const std::string k = std::to_string(args[0].col_slot);
            out->v = Tmp("bool", AblSel(2, "(ob" + k + "[u] - oa" + k + "[u] > 19)",
                                            "(hm_ok" + k + " ? gdv_range_any(hit" + std::to_string(h) + ", oa" + k + "[u] - sb" + k +
                                                ", ob" + k + "[u] - sb" + k + " - " + std::to_string(lit.size() - 1) + ") : " +
                                                per_row + ")"));
            return Status::OK();
          }
          out->v = Tmp("bool", per_row);
          return Status::OK();
        }
        std::string pb = ByteTable(bytes), pk = ByteTable(kinds);
        out->v = Tmp("bool", "gdv_like(" + args[0].v + ", " + pb + ", " + pk + ", " +
                                 std::to_string(kinds.size()) + ")");
        return Status::OK();
      }
      std::string call = def->symbol + "(";
      bool first = true;
      auto push = [&](const std::string& a) {
        if (!first) call += ", ";
        call += a;
        first = false;
      };
      if (def->flags & kNeedsContext) {
        push("ctx");
        can_raise_ = true;
      }
      if (def->policy == NullPolicy::kNullIfNull) {
        std::string lanes;
        for (auto& a : args) {
          push(a.v);
          if ((def->flags & kDecimalArgs) && a.type.is_decimal()) {
            push(std::to_string(a.type.precision));
            push(std::to_string(a.type.scale));
          }
          out->vcols.insert(a.vcols.begin(), a.vcols.end());
          lanes = AndExpr(lanes, a.vlane);
        }
        if (def->flags & kDecimalArgs) {
          push(std::to_string(out->type.precision));
          push(std::to_string(out->type.scale));
        }
        out->vlane = lanes;
        call += ")";
        if (def->flags & kNeedsContext) {
          // Functions that can raise run only on rows where every argument is valid and
          // the enclosing if/else / short-circuit path is live — otherwise a guarded
          // `if (b != 0) a / b` would raise on the rows it guards against.
          std::string guard = AndExpr(AndExpr("live", active), LaneValid(*out));
          const std::string idle = out->type.is_varlen() ? "gdv_empty_str()" : "(" + ctype + ")0";
          out->v = Tmp(ctype, guard + " ? " + call + " : " + idle);
        } else {
          out->v = Tmp(ctype, call);
        }
      } else if (def->policy == NullPolicy::kNullNever) {
        for (auto& a : args) {
          push(a.v);
          push(LaneValid(a));
        }
        call += ")";
        out->v = Tmp(ctype, call);
      } else {
        for (auto& a : args) {
          push(a.v);
          push(LaneValid(a));
        }
        std::string ov = "ov" + std::to_string(next_tmp_++);
        Stmt("bool " + ov + " = false;");
        push("&" + ov);
        call += ")";
        out->v = Tmp(ctype, call);
        out->vlane = ov;
      }
      return Status::OK();
    }
    case NodeKind::kIf: {
      auto& n = static_cast<const IfNode&>(node);
      Val c, t, e;
      GDV_RETURN_NOT_OK(Gen(*n.condition(), active, &c));
      // a null condition selects the else branch
      std::string take = Tmp("bool", AndFull(LaneValid(c), c.v));
      GDV_RETURN_NOT_OK(Gen(*n.then_node(), AndExpr(active, take), &t));
      GDV_RETURN_NOT_OK(Gen(*n.else_node(), AndExpr(active, "!" + take), &e));
      // `if (c) <materialised value> else NULL` (and its mirror): the value is the branch's, valid only
      // where the branch is taken — what a guarded first-stage expression of a two-stage plan looks
      // like (StageMaterialisedValues), and fine wherever a materialised value is (output, concat)
      {
        const bool t_mat = !t.pieces.empty() || t.opaque, e_mat = !e.pieces.empty() || e.opaque;
        auto null_literal = [](const Node& x) {
          return x.kind() == NodeKind::kLiteral && static_cast<const LiteralNode&>(x).is_null();
        };
        if (t_mat != e_mat && null_literal(t_mat ? *n.else_node() : *n.then_node())) {
          const Val& m = t_mat ? t : e;
          const std::string taken = t_mat ? take : "!" + take;
          *out = m;
          out->type = n.return_type();
          if (!m.vcols.empty()) {  // fold the column validity into the lane predicate next to the guard
            out->vlane = AndExpr(LaneValid(m), taken);
            out->vcols.clear();
          } else {
            out->vlane = AndExpr(m.vlane, taken);
          }
          for (auto& pc : out->pieces) pc.second = AndExpr(pc.second, taken);
          out->col_slot = -1;
          return Status::OK();
        }
      }
      if (!t.pieces.empty() || !e.pieces.empty() || t.opaque || e.opaque)
        return Status::CodeGenError(
            "if/else over a concat / lpad / rpad / reverse / replace / castVARCHAR(number) result is not supported by the HIP "
            "backend yet");
      out->type = n.return_type();
      const std::string ctype = out->type.CType();
      out->pieces.clear();
      out->col_slot = -1;
      out->opaque = false;
      out->v = Tmp(ctype, take + " ? " + t.v + " : " + e.v);
      out->vcols.clear();
      if (t.never_null() && e.never_null()) {
        out->vlane.clear();
      } else {
        out->vlane = Tmp("bool", take + " ? " + LaneValid(t) + " : " + LaneValid(e));
      }
      return Status::OK();
    }
    case NodeKind::kBoolean: {
      // SQL three-valued logic with left-to-right short circuit:
      //   AND: false if any child is (valid, false); else null if any child is null; else true
      //   OR : true  if any child is (valid, true);  else null if any child is null; else false
      auto& n = static_cast<const BooleanNode&>(node);
      const bool is_and = n.op() == BooleanNode::kAnd;
      std::string decided;    // some earlier child already fixed the result
      std::string all_valid;  // every child so far valid
      std::string live_path = active;
      for (auto& child : n.children()) {
        Val c;
        GDV_RETURN_NOT_OK(Gen(*child, live_path, &c));
        std::string cvalid = LaneValid(c);
        std::string hit = AndFull(cvalid, is_and ? "!" + c.v : c.v);
        hit = Tmp("bool", hit);
        decided = decided.empty() ? hit : Tmp("bool", "(" + decided + " || " + hit + ")");
        all_valid = AndExpr(all_valid, cvalid);
        live_path = AndExpr(active, "!" + decided);
      }
      out->type = boolean();
      out->vcols.clear();
      out->col_slot = -1;
      if (all_valid.empty() || all_valid == "true") {
        out->vlane.clear();
        out->v = Tmp("bool", is_and ? "!" + decided : decided);
      } else {
        std::string av = Tmp("bool", all_valid);
        out->vlane = Tmp("bool", "(" + decided + " || " + av + ")");
        // value bit under a null result is defined as false
        out->v = Tmp("bool", is_and ? "(!" + decided + " && " + av + ")" : decided);
      }
      return Status::OK();
    }
    case NodeKind::kIn: {
      auto& n = static_cast<const InNode&>(node);
      Val x;
      GDV_RETURN_NOT_OK(Gen(*n.eval(), active, &x));
      if (!x.pieces.empty() || x.opaque)
        return Status::CodeGenError(
            "IN over a concat / lpad / rpad / reverse / replace / castVARCHAR(number) result is not supported by the HIP backend yet");
      out->pieces.clear();
      out->col_slot = -1;
      out->type = boolean();
      out->vcols = x.vcols;
      out->vlane = x.vlane;
      if (n.value_type().is_varlen()) {
        std::string bytes, offs;
        auto put32 = [&](uint32_t v) { offs.append(reinterpret_cast<const char*>(&v), 4); };
        put32(0);
        for (auto& l : n.values()) {
          bytes += l.bytes;
          put32(static_cast<uint32_t>(bytes.size()));
        }
        const std::string tab = ByteTable(offs, "gdv_int32");
        out->v = Tmp("bool", "gdv_in_strings(" + x.v + ", " + ByteTable(bytes) + ", " + tab + ", " +
                                 std::to_string(n.values().size()) + ")");
        return Status::OK();
      }
      const DataType& vt = n.value_type();
      if (vt.is_decimal()) {
        // 16-byte values: equality against two argument slots each (lists are short in practice)
        if (n.values().size() > 64)
          return Status::CodeGenError("IN over decimal128 with more than 64 values is not supported by the HIP backend yet");
        std::string e;
        for (auto& l : n.values()) {
          lits_.push_back(l.lo);
          lits_.push_back(l.hi);
          const std::string i = std::to_string(lits_.size() - 2), j = std::to_string(lits_.size() - 1);
          if (!e.empty()) e += " || ";
          e += "(" + x.v + " == gdv_make_int128(A.lit[" + j + "], A.lit[" + i + "]))";
        }
        out->v = e.empty() ? std::string("false") : Tmp("bool", e);
        return Status::OK();
      }
      std::vector<uint64_t> vals;
      uint64_t mask = vt.byte_width() >= 8 ? ~0ull : ((1ull << (8 * vt.byte_width())) - 1);
      const bool is_fp = vt.id == kFloat || vt.id == kDouble;
      for (auto& l : n.values()) {
        uint64_t bits = l.lo & mask;
        if (is_fp) {
          // value equality, as a hash set of floats gives it: -0.0 and +0.0 are one value,
          // a NaN equals nothing (the probe adds +0.0, which maps -0.0 to +0.0 and keeps NaNs NaN)
          const bool nan = vt.id == kFloat ? ((bits & 0x7f800000u) == 0x7f800000u && (bits & 0x7fffffu) != 0)
                                           : ((bits & 0x7ff0000000000000ull) == 0x7ff0000000000000ull &&
                                              (bits & 0xfffffffffffffull) != 0);
          if (nan) continue;
          if (bits == (vt.id == kFloat ? 0x80000000ull : 0x8000000000000000ull)) bits = 0;
        }
        vals.push_back(bits);
      }
      std::sort(vals.begin(), vals.end());
      vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
      const std::string probe = is_fp ? "gdv_bits64(" + x.v + " + (" + vt.CType() + ")0)" : "gdv_bits64(" + x.v + ")";
      if (vals.empty()) {
        out->v = "false";
      } else if (vals.size() <= 8) {
        const std::string xb = Tmp("gdv_uint64", probe);
        std::string e;
        for (auto v : vals) {
          if (!e.empty()) e += " || ";
          e += "(" + xb + " == A.lit[" + std::to_string(LitSlot(v)) + "])";
        }
        out->v = Tmp("bool", e);
      } else {
        // sorted table in the constant block + branch-free binary search on the value's bit image
        std::string tab(reinterpret_cast<const char*>(vals.data()), vals.size() * 8);
        out->v = Tmp("bool", "gdv_in_sorted(" + probe + ", " + ByteTable(tab, "gdv_uint64") + ", " +
                                 std::to_string(vals.size()) + ")");
      }
      return Status::OK();
    }
  }
  return Status::CodeGenError("unknown node kind");
}

// Selection mode: the row of input slot k that output slot `row` reads.  Columns of the caller's
// batch are gathered through the selection vector; the temporaries of a two-stage plan (schema
// index >= compact_from_) were produced BY a selection-mode first stage and are compact already.
std::string RowOf(const CodeGen& cg, int k) {
  return cg.input_fields_[k] >= cg.compact_from_ ? "(live ? row : 0)" : "srow[u]";
}

// Assembles the translation unit around the generated row body.
struct Assembler {
  CodeGen& cg;
  KernelPlan* plan;
  std::ostringstream src;
  int sweep_group_ = 1;  // > 1: wave-shaped main kernel whose byte sweep takes this many sub-tiles' spans at a time

  void Header(const std::vector<std::string>& expr_strings) {
    src << "// generated by gandiva_amd (gdv_planner.cc) — fused "
        << (plan->kind == KernelKind::kFilter ? "filter" : "projection") << " kernel for gfx950\n";
    for (size_t i = 0; i < expr_strings.size(); i++)
      src << "// @expr_" << i << " = " << expr_strings[i] << "\n";
    // rows of the batch: selection-mode plans may take the slot count from device memory (aux2: an
    // asynchronous Filter left it there), so a filter -> project chain needs no host round trip
    if ((plan->mode != SelectionMode::kNone || plan->opts.rows_word) && plan->kind != KernelKind::kFilterProject)
      // (clamped to [0, n]: n is the capacity the outputs and the grid were sized for — a stale or
      // foreign count word must not make the kernel write past them)
      src << "#define GDV_ROWS(A) ((A).aux2 != 0 ? gdv_clamp_rows(*(const gdv_int64*)(A).aux2, (A).n) : (A).n)\n";
    else
      src << "#define GDV_ROWS(A) ((A).n)\n";
    if (cg.unroll_rows_) src << "#define GDV_UNROLL_ROWS 1\n";
    src << "#define GDV_U " << plan->opts.subtiles << "\n";
    src << "#define GDV_WAVES " << plan->opts.waves << "\n";
    if (plan->opts.cast_x86_indefinite) src << "#define GDV_CAST_X86_INDEFINITE 1\n";
    if (sweep_group_ > 1)
      // (wave-shaped main kernels, round 6: the byte sweep covers GDV_SG sub-tiles' spans at a time; the LDS mirror and the
      // match bitmaps hold that much)
      src << "#define GDV_SG " << sweep_group_ << "\n#define GDV_SUB_SPAN " << 1024 * sweep_group_ << "\n";
    src << "#include \"gdv_device_lib.hpp\"\n";
    const int nin = std::max<int>(1, plan->input_fields.size());
    const int nout = std::max<int>(1, plan->output_types.size());
    src << "struct gdv_in_slot { const void* data; gdv_bitmap valid; gdv_bitmap bits; const gdv_int32* offsets; };\n";
    src << "struct gdv_out_slot { void* data; gdv_uint64* valid; gdv_int32* offsets; gdv_int64 cap; };\n";
    src << "struct gdv_args {\n"
        << "  gdv_int64 n; gdv_uint32* err; const void* sel; gdv_uint64* mask; gdv_uint32* counts;\n"
        << "  gdv_int64 aux0, aux1, aux2;\n"
        << "  gdv_in_slot in[" << nin << "];\n"
        << "  gdv_out_slot out[" << nout << "];\n"
        << "  gdv_uint64 lit[" << std::max<size_t>(1, cg.lits_.size()) << "];  // fixed-width literals of the plan\n"
        << "};\n";
    plan->literals = cg.lits_;
    plan->const_block = cg.blob_;
    plan->layout.n_lit = static_cast<int>(cg.lits_.size());
  }
};

std::string SelCType(SelectionMode m) {
  switch (m) {
    case SelectionMode::kUInt16: return "gdv_uint16";
    case SelectionMode::kUInt32: return "gdv_uint32";
    default: return "gdv_uint64";
  }
}

// Output bitmap words are accumulated per wave tile: word u is deposited into lane u of an
// accumulator register, so the tile's GDV_U words leave with one coalesced store.  Outputs
// whose word expressions are textually identical share one accumulator.
struct WordAccumulators {
  std::map<std::string, std::string> by_expr;  // word expression -> accumulator name
  std::vector<std::string> names;
  std::string Get(CodeGen& cg, const std::string& word_expr) {
    auto it = by_expr.find(word_expr);
    if (it != by_expr.end()) return it->second;
    std::string name = "acc" + std::to_string(names.size());
    names.push_back(name);
    by_expr[word_expr] = name;
    cg.Stmt(name + " = gdv_deposit_word(" + name + ", u, " + word_expr + ", lane);");
    return name;
  }
};

std::string WordStore(const std::string& acc, const std::string& dst, bool nontemporal = false) {
  return std::string("  if (") + AblNot(2) + std::string("lane < GDV_U && (FULL || wbase + lane < ((n + 63) >> 6))) ") +
         (nontemporal ? "GDV_WORD_ST_NT(" : "GDV_WORD_ST(") + dst +
         " + wbase + lane, " + acc + ");\n";
}

Status Assemble(CodeGen& cg, KernelPlan* plan, const std::vector<std::string>& expr_strings,
                const WordAccumulators& accs, const std::string& decls_before_loop,
                const std::string& epilogue_after_loop) {
  plan->input_fields = cg.input_fields_;
  plan->input_needs_values = cg.needs_values_;
  plan->input_needs_validity = cg.needs_validity_;
  plan->can_raise = cg.can_raise_;
  plan->layout.n_in = static_cast<int>(plan->input_fields.size());
  plan->layout.n_out = static_cast<int>(plan->output_types.size());
  const bool sel = cg.selection();
  const int nin = plan->layout.n_in;
  const std::string body = cg.body_.str();

  Assembler as{cg, plan, {}};
  as.Header(expr_strings);
  std::ostringstream& s = as.src;
  s << AblDefine();
  s << "#define GDV_OUT(e, v) if (live) " << (plan->opts.nontemporal ? "gdv_stnt" : "gdv_st") << "(out##e, row, (v))\n";

  s << "template <bool FULL>\n"
    << "GDV_DEV void gdv_tile(const gdv_args& A, const gdv_int64 wbase, const int lane) {\n"
    << "  gdv_ctx ctx{A.err};\n"
    << "  (void)ctx;\n"
    << "  const gdv_uint8* const gdv_cst = (const gdv_uint8*)A.aux0;  // the plan's constant block\n"
    << "  (void)gdv_cst;\n"
    << "  const gdv_int64 n = GDV_ROWS(A);\n"
    << "  const gdv_int64 rbase = wbase * 64;\n";
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (t.id != kBool && cg.needs_values_[k])
      s << "  const " << t.CType() << "* __restrict__ in" << k << " = (const " << t.CType() << "*)A.in[" << k
        << "].data;\n";
  }
  for (size_t e = 0; e < plan->output_types.size(); e++) {
    const DataType& t = plan->output_types[e];
    if (t.id != kBool)
      s << "  " << t.CType() << "* __restrict__ out" << e << " = (" << t.CType() << "*)A.out[" << e << "].data;\n";
  }
  if (sel)
    s << "  const " << SelCType(cg.sel_mode_) << "* __restrict__ selv = (const " << SelCType(cg.sel_mode_)
      << "*)A.sel;\n";

  // ---- phase 1: every load of the tile, no control flow in between
  s << "  // ---- phase 1: all loads of this wave's GDV_U sub-tiles, issued back to back\n";
  if (sel) s << "  gdv_int64 srow[GDV_U];\n";
  std::ostringstream bitmap_loads;  // one vector load per column: lane u <-> word u
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (t.id == kBool) {
      if (cg.needs_values_[k]) {
        if (sel) s << "  bool x" << k << "[GDV_U];\n";
        else bitmap_loads << "  const gdv_uint64 dw" << k << " = " << AblSel(1, "~0ull", "gdv_bitmap_tile(A.in[" + std::to_string(k) + "].bits, wbase, lane, GDV_U)") << ";\n";
      }
    } else if (cg.needs_values_[k]) {
      s << "  " << t.CType() << " c" << k << "[GDV_U];\n";
    }
    if (cg.needs_validity_[k]) {
      if (sel) s << "  bool b" << k << "[GDV_U];\n";
      else bitmap_loads << "  const gdv_uint64 vw" << k << " = " << AblSel(1, "~0ull", "gdv_bitmap_tile(A.in[" + std::to_string(k) + "].valid, wbase, lane, GDV_U)") << ";\n";
    }
  }
  // Bitmap words BEHIND the value loads (round 2): issued first, the compiler consumed them first
  // and waited for them before most value loads were even issued (7 of 32 in the C3 predicate
  // kernel); a hand-written copy of that kernel with every load in flight ran 0.45 ms faster
  // (tools/hbm_ceiling.hip, profiles/r02_k1_k2_experiments.txt).
  s << bitmap_loads.str();
  const std::string ld = plan->opts.nt_loads ? "gdv_ldnt" : "gdv_ld";
  {
    s << "#pragma unroll\n  for (int u = 0; u < GDV_U; u++) {\n"
      << "    const gdv_int64 row = rbase + u * 64 + lane;\n"
      << "    const bool live = FULL || row < n;\n"
      << "    (void)live;\n";
    if (sel) {
      s << "    srow[u] = live ? (gdv_int64)selv[row] : 0;\n";
      for (int k = 0; k < nin; k++) {
        const DataType& t = cg.schema_[plan->input_fields[k]].type;
        if (t.id == kBool) {
          if (cg.needs_values_[k]) s << "    x" << k << "[u] = gdv_bitmap_bit(A.in[" << k << "].bits, " << RowOf(cg, k) << ");\n";
        } else if (cg.needs_values_[k]) {
          s << "    c" << k << "[u] = gdv_ld(in" << k << ", " << RowOf(cg, k) << ");\n";
        }
        if (cg.needs_validity_[k]) s << "    b" << k << "[u] = gdv_bitmap_bit(A.in[" << k << "].valid, " << RowOf(cg, k) << ");\n";
      }
    } else {
      for (int k = 0; k < nin; k++) {
        const DataType& t = cg.schema_[plan->input_fields[k]].type;
        if (t.id != kBool && cg.needs_values_[k])
          s << "    c" << k << "[u] = live ? " << ld << "(in" << k << ", row) : (" << t.CType() << ")0;\n";
      }
    }
    s << "  }\n";
  }

  // ---- phase 2: row body
  s << "  // ---- phase 2: fused expression bodies (value for every row, validity per word)\n";
  for (auto& a : accs.names) s << "  gdv_uint64 " << a << " = 0;\n";
  s << decls_before_loop;
  s << "#pragma unroll\n  for (int u = 0; u < GDV_U; u++) {\n"
    << "    {\n";
  s << "      const gdv_int64 row = rbase + u * 64 + lane;\n"
    << "      const bool live = FULL || row < n;\n"
    << "      const gdv_uint64 livemask = FULL ? ~0ull : __ballot(live);\n";
  s << "      (void)livemask; (void)row; (void)live;\n";
  if (!sel) {
    for (int k = 0; k < nin; k++) {
      const DataType& t = cg.schema_[plan->input_fields[k]].type;
      if (t.id == kBool && cg.needs_values_[k])
        s << "      const gdv_uint64 d" << k << " = " << "gdv_tile_word(dw" + std::to_string(k) + ", u)" << ";\n";
      if (cg.needs_validity_[k]) s << "      const gdv_uint64 v" << k << " = " << "gdv_tile_word(vw" + std::to_string(k) + ", u)" << ";\n";
    }
  }
  s << body;
  s << "    }\n  }\n";
  s << epilogue_after_loop;
  s << "}\n\n";
  // ---- kernel: grid-stride over workgroup tiles; wave w of a workgroup owns GDV_U
  // consecutive 64-row sub-tiles, so a workgroup tile is a contiguous run of
  // 64*GDV_U*GDV_WAVES rows and (for GDV_U*GDV_WAVES = 16) exactly one 128-byte line of
  // each bitmap.
  const std::string attrs;
  s << "GDV_DEV void gdv_kernel_body(const gdv_args& A, const gdv_int64 first_group, const gdv_int64 num_groups) {\n"
    << "  const int lane = threadIdx.x & 63;\n"
    << "  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
    << "  const gdv_int64 n = GDV_ROWS(A);\n"
    << "  const gdv_int64 nwords = (n + 63) >> 6;\n"
    << "  const gdv_int64 nfull = n / (64 * GDV_U);                // full wave tiles\n"
    << "  const gdv_int64 nwt = (nwords + GDV_U - 1) / GDV_U;      // all wave tiles\n"
    << "  for (gdv_int64 wt = first_group * GDV_WAVES + wave; wt < nfull; wt += num_groups * GDV_WAVES)\n"
    << "    gdv_tile<true>(A, wt * GDV_U, lane);\n"
    // The single partial wave tile is handled after the loop, not in an if/else next to
    // the full-tile body: side by side, the compiler hoists the two bodies' common bitmap
    // loads above the branch and serialises them in front of the value loads.
    << "  if (nwt > nfull && wave == (int)(nfull % GDV_WAVES) &&\n"
    << "      first_group == (gdv_int64)((nfull / GDV_WAVES) % num_groups))\n"
    << "    gdv_tile<false>(A, nfull * GDV_U, lane);\n"
    << "}\n"
    << "extern \"C\" __global__ void " << attrs << "__launch_bounds__(GDV_WAVES * 64) GDV_KERNEL_NAME(const gdv_args A) {\n"
    << "  gdv_kernel_body(A, (gdv_int64)blockIdx.x, (gdv_int64)gridDim.x);\n"
    << "}\n";
  // Many small batches in ONE launch (round 3: the reference is fed 4K-64K-row batches, where a
  // launch per batch is all overhead): blockIdx.y picks the batch, its argument block comes from a
  // table in device memory instead of the kernel-argument segment.  Row mode projections only.
  if (!sel && plan->kind == KernelKind::kProject)
    s << "extern \"C\" __global__ void " << attrs << "__launch_bounds__(GDV_WAVES * 64) GDV_KERNEL_NAME_many(const gdv_args* __restrict__ table) {\n"
      << "  gdv_kernel_body(table[blockIdx.y], (gdv_int64)blockIdx.x, (gdv_int64)gridDim.x);\n"
      << "}\n";
  // Small batches of a filter: predicate, offsets scan and index emission by ONE workgroup in ONE
  // launch (gdv_small_filter_finish); blockIdx.y picks the batch.  The batch's argument block gives
  // the index buffer in aux1, the index width in `sel`, where the count goes in aux2.
  if (plan->kind == KernelKind::kFilter)
    s << "GDV_DEV void gdv_small_filter(const gdv_args& A) {\n"
      << "  __shared__ gdv_uint32 lds_offsets[GDV_SMALL_MAX_TILES];\n"
      << "  __shared__ gdv_uint16 lds_stage[GDV_WAVES * 64 * 64];\n"
      << "  gdv_kernel_body(A, 0, 1);  // this workgroup runs the predicate over every wave tile of its batch\n"
      << "  __syncthreads();           // match words and counts are this workgroup's own: visible after the barrier\n"
      << "  gdv_small_filter_finish(A.mask, A.counts, A.n, (void*)A.aux1, (gdv_int32)(gdv_int64)A.sel, (gdv_int64*)A.aux2,\n"
      << "                          lds_offsets, lds_stage, threadIdx.x & 63, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)),\n"
      << "                          GDV_WAVES, GDV_U);\n"
      << "}\n"
      << "extern \"C\" __global__ void __launch_bounds__(GDV_WAVES * 64) GDV_KERNEL_NAME_small(const gdv_args* __restrict__ table) {\n"
      << "  gdv_small_filter(table[blockIdx.y]);\n"
      << "}\n"
      // (one batch: its argument block travels in the kernel-argument segment — no table to upload)
      << "extern \"C\" __global__ void __launch_bounds__(GDV_WAVES * 64) GDV_KERNEL_NAME_small1(const gdv_args A) {\n"
      << "  gdv_small_filter(A);\n"
      << "}\n";

  std::string text = s.str();
  uint64_t h = Fnv1a(HashableSource(text) + LibraryTag(text));
  char name[64];
  snprintf(name, sizeof(name), "gdv_k_%016llx", static_cast<unsigned long long>(h));
  plan->kernel_name = name;
  for (size_t pos = text.find("GDV_KERNEL_NAME"); pos != std::string::npos; pos = text.find("GDV_KERNEL_NAME", pos))
    text.replace(pos, strlen("GDV_KERNEL_NAME"), plan->kernel_name);
  plan->source = text;
  plan->ir = text;
  plan->has_many_entry = !sel && plan->kind == KernelKind::kProject;
  plan->has_small_entry = plan->kind == KernelKind::kFilter;
  return Status::OK();
}


// ---- pieces of the string skeletons shared by the scanner shape and the wave shape
// pointers of the tile function: var-len / fixed-width inputs, outputs, the selection vector
void EmitStringPointersAndLoads(std::ostringstream& s, CodeGen& cg, KernelPlan* plan, bool with_outputs,
                                bool wave_shape = false) {
  const bool sel = cg.selection();
  const int nin = plan->layout.n_in;
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (t.is_varlen() && cg.needs_values_[k]) {
      s << "  const gdv_uint8* __restrict__ sd" << k << " = (const gdv_uint8*)A.in[" << k << "].data;\n"
        << "  const gdv_int32* __restrict__ so" << k << " = A.in[" << k << "].offsets;\n"
        << "  const gdv_uint8* slim" << k << " = sd" << k << " + A.in[" << k << "].bits.nwords;\n";
    } else if (t.id != kBool && cg.needs_values_[k]) {
      s << "  const " << t.CType() << "* __restrict__ in" << k << " = (const " << t.CType() << "*)A.in[" << k
        << "].data;\n";
    }
  }
  for (size_t e = 0; with_outputs && e < plan->output_types.size(); e++) {
    const DataType& t = plan->output_types[e];
    if (t.is_varlen()) {
      s << "  gdv_uint8* __restrict__ outd" << e << " = (gdv_uint8*)A.out[" << e << "].data;\n"
        << "  gdv_int32* __restrict__ outo" << e << " = A.out[" << e << "].offsets;\n";
    } else if (t.id != kBool) {
      s << "  " << t.CType() << "* __restrict__ out" << e << " = (" << t.CType() << "*)A.out[" << e << "].data;\n";
    }
  }
  if (sel)
    s << "  const " << SelCType(cg.sel_mode_) << "* __restrict__ selv = (const " << SelCType(cg.sel_mode_)
      << "*)A.sel;\n";

  // ---- loads: offsets, fixed-width values, validity / bool words
  s << "  // ---- loads of this wave's GDV_U sub-tiles, issued back to back\n";
  if (sel) s << "  gdv_int64 srow[GDV_U];\n";
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (t.id == kBool) {
      if (cg.needs_values_[k]) {
        if (sel) s << "  bool x" << k << "[GDV_U];\n";
        else s << "  const gdv_uint64 dw" << k << " = gdv_bitmap_tile(A.in[" << k << "].bits, wbase, lane, GDV_U);\n";
      }
    } else if (t.is_varlen()) {
      if (cg.needs_values_[k]) {
        // (wave shape: only the start offsets are loaded; a row's end is the next lane's start)
        // (... of a CONTIGUOUS run of rows: under a selection vector both ends are gathered)
        if (wave_shape && !sel) s << "  gdv_int32 oa" << k << "[GDV_U];\n";
        else s << "  gdv_int32 oa" << k << "[GDV_U], ob" << k << "[GDV_U];\n";
      }
    } else if (cg.needs_values_[k]) {
      s << "  " << t.CType() << " c" << k << "[GDV_U];\n";
    }
    if (cg.needs_validity_[k]) {
      if (sel) s << "  bool b" << k << "[GDV_U];\n";
      else s << "  const gdv_uint64 vw" << k << " = gdv_bitmap_tile(A.in[" << k << "].valid, wbase, lane, GDV_U);\n";
    }
  }
  s << "#pragma unroll\n  for (int u = 0; u < GDV_U; u++) {\n"
    << "    const gdv_int64 row = rbase + u * 64 + lane;\n"
    << "    const bool live = row < n;\n"
    << "    (void)live;\n";
  if (sel) {
    s << "    srow[u] = live ? (gdv_int64)selv[row] : 0;\n";
    for (int k = 0; k < nin; k++) {
      const DataType& t = cg.schema_[plan->input_fields[k]].type;
      if (t.id == kBool) {
        if (cg.needs_values_[k]) s << "    x" << k << "[u] = gdv_bitmap_bit(A.in[" << k << "].bits, " << RowOf(cg, k) << ");\n";
      } else if (t.is_varlen()) {
        if (cg.needs_values_[k])
          s << "    oa" << k << "[u] = so" << k << "[" << RowOf(cg, k) << "]; ob" << k << "[u] = so" << k << "[" << RowOf(cg, k) << " + 1];\n";
      } else if (cg.needs_values_[k]) {
        s << "    c" << k << "[u] = gdv_ld(in" << k << ", " << RowOf(cg, k) << ");\n";
      }
      if (cg.needs_validity_[k]) s << "    b" << k << "[u] = gdv_bitmap_bit(A.in[" << k << "].valid, " << RowOf(cg, k) << ");\n";
    }
  } else {
    for (int k = 0; k < nin; k++) {
      const DataType& t = cg.schema_[plan->input_fields[k]].type;
      if (t.is_varlen()) {
        // rows past the end take the closing offset: length 0, and the span stays contiguous
        if (cg.needs_values_[k] && wave_shape)
          s << "    oa" << k << "[u] = so" << k << "[live ? row : n];\n";
        else if (cg.needs_values_[k])
          s << "    oa" << k << "[u] = so" << k << "[live ? row : n]; ob" << k << "[u] = so" << k
            << "[row + 1 < n ? row + 1 : n];\n";
      } else if (t.id != kBool && cg.needs_values_[k]) {
        s << "    c" << k << "[u] = live ? " << (plan->opts.nt_loads ? "gdv_ldnt" : "gdv_ld") << "(in" << k
          << ", row) : (" << t.CType() << ")0;\n";
      }
    }
  }
  s << "  }\n";

}

// the rolled row loop: prologue (this sub-tile's inputs picked at index 0), the fused body, rotation
// of the per-sub-tile registers; the caller appends its own rotations and closes the loop ("  }\n")
void EmitStringRowLoop(std::ostringstream& s, CodeGen& cg, KernelPlan* plan, bool wave_shape = false,
                       const std::string& at_top = std::string()) {
  const bool sel = cg.selection();
  const int nin = plan->layout.n_in;
  // The row loop is NOT unrolled: the per-sub-tile registers are read and written through
  // gdv_pick / gdv_put (selects on the wave-uniform u), so the fused body exists once — a
  // quarter of the code, the compile time and the VGPRs of the unrolled form.
  s << "GDV_ROW_LOOP\n  for (int u = 0; u < GDV_U; u++) {\n"
    << at_top
    << "    {\n"
    << "      const gdv_int64 row = rbase + u * 64 + lane;\n"
    << "      const bool live = row < n;\n"
    << "      const gdv_uint64 livemask = __ballot(live);\n"
    << "      (void)livemask; (void)row;\n";
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (t.id == kBool) {
      if (cg.needs_values_[k] && sel) s << "      const bool x" << k << "_u = x" << k << "[0];\n";
    } else if (t.is_varlen()) {
      if (cg.needs_values_[k] && wave_shape && !sel)
        s << "      const gdv_int32 oa" << k << "_u = oa" << k << "[0];\n"
          << "      const gdv_int32 ob" << k << "_u = gdv_next_lane_i32(oa" << k << "_u, u + 1 < GDV_U ? __builtin_amdgcn_readfirstlane(oa"
          << k << "[GDV_U > 1 ? 1 : 0]) : sp1" << k << ", lane);\n";
      else if (cg.needs_values_[k])
        s << "      const gdv_int32 oa" << k << "_u = oa" << k << "[0], ob" << k << "_u = ob" << k << "[0];\n";
      if (cg.needs_values_[k] && cg.row_ascii_slots_.count(k))
        // exact variant: ASCII is a fact about THIS row (conservatively: about the 16-byte pieces it touches)
        s << "      const gdv_str s" << k << " = gdv_with_lead(gdv_make_str(sd" << k << ", oa" << k << "_u, ob" << k << "_u, slim" << k
          << ", sfl" << k << "), hi8_" << k << ", hm_ok" << k << ", cb" << k << ", oa" << k << "_u - sb" << k << ");\n";
      else if (cg.needs_values_[k])
        s << "      const gdv_str s" << k << " = gdv_make_str(sd" << k << ", oa" << k << "_u, ob" << k << "_u, slim" << k
          << ", sfl" << k << ");\n";
      if (cg.needs_values_[k] && cg.sel_ascii_check_ && cg.ascii_slots_.count(k))
        s << "      nasc" << k << " |= __ballot(live && !gdv_row_is_ascii(s" << k << "));\n";
    } else if (cg.needs_values_[k]) {
      s << "      const " << t.CType() << " c" << k << "_u = c" << k << "[0];\n";
    }
    if (cg.needs_validity_[k] && sel) s << "      const bool b" << k << "_u = b" << k << "[0];\n";
  }
  if (!sel) {
    for (int k = 0; k < nin; k++) {
      const DataType& t = cg.schema_[plan->input_fields[k]].type;
      if (t.id == kBool && cg.needs_values_[k])
        s << "      const gdv_uint64 d" << k << " = " << "gdv_tile_word(dw" + std::to_string(k) + ", u)" << ";\n";
      if (cg.needs_validity_[k]) s << "      const gdv_uint64 v" << k << " = " << "gdv_tile_word(vw" + std::to_string(k) + ", u)" << ";\n";
    }
  }
  {
    // the body addresses per-sub-tile inputs as NAME[u]: here they are the NAME_u picked above
    static const std::regex per_u("\\b(oa|ob|c|x|b)([0-9]+)\\[u\\]");
    s << std::regex_replace(cg.body_.str(), per_u, "$1$2_u");
  }
  s << "    }\n    // next sub-tile to the front\n";
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (t.id == kBool) {
      if (cg.needs_values_[k] && sel) s << "    gdv_rot(x" << k << ");\n";
    } else if (t.is_varlen()) {
      if (cg.needs_values_[k]) s << "    gdv_rot(oa" << k << ");" << (wave_shape && !sel ? "" : " gdv_rot(ob" + std::to_string(k) + ");") << "\n";
    } else if (cg.needs_values_[k]) {
      s << "    gdv_rot(c" << k << ");\n";
    }
    if (cg.needs_validity_[k] && sel) s << "    gdv_rot(b" << k << ");\n";
  }
}

// The '%needle%' of a sweep hook as the matcher's three constants.  Wave-shaped kernels (round 4) carry
// them in the kernel TEXT: the needle is a literal of the plan, and as immediates its bytes cost no
// scalar loads, no registers across the row loop and let the compiler fold the first-byte splats (the
// round-3 verdict priced the runtime needle among the 0.2 ms the generic emitter paid over its
// prototype).  Plans that differ in the needle are different kernels there; the scanner shape keeps the
// needle a run-time constant (one code object for every pattern of a given length).
std::string NeedleConstants(CodeGen& cg, int h, uint64_t mask) {
  const ContainsHook& hk = cg.contains_hooks_[h];
  std::ostringstream s;
  const std::string H = std::to_string(h);
  if (cg.bake_needles_) {
    uint64_t v = 0;
    std::memcpy(&v, hk.needle.data(), std::min<size_t>(8, hk.needle.size()));
    v &= mask;
    s << "  const gdv_uint64 nd" << H << " = " << Hex64(v) << ";  // the needle: a literal of the plan\n"
      << "  const gdv_uint32 ns0_" << H << " = " << Hex64((v & 0xff) * 0x01010101ull) << ", ns1_" << H << " = "
      << Hex64(((v >> 8) & 0xff) * 0x01010101ull) << ";\n";
  } else {
    s << "  const gdv_uint64 nd" << H << " = gdv_load8_raw(" << cg.hook_tables_[h] << ") & " << Hex64(mask)
      << ";  // the needle: a runtime constant\n"
      << "  const gdv_uint32 ns0_" << H << " = (gdv_uint32)(nd" << H << " & 0xffull) * 0x01010101u, ns1_" << H << " = (gdv_uint32)((nd" << H
      << " >> 8) & 0xffull) * 0x01010101u;\n";
  }
  return s.str();
}

// byte sweep of every var-len input (scanner shape): tile-wide ASCII flag, '%needle%' match
// bitmaps, flat outputs.  (Wave-shaped kernels sweep one sub-tile at a time: EmitWaveSweep.)
void EmitStringSweep(std::ostringstream& s, CodeGen& cg, KernelPlan* plan) {
  const bool sel = cg.selection();
  const int nin = plan->layout.n_in;
  const int nhook = static_cast<int>(cg.contains_hooks_.size());
  // ---- sweep: lanes over the bytes of each var-len input's span
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (!(t.is_varlen() && cg.needs_values_[k])) continue;
    std::vector<int> hooks;
    for (int h = 0; h < nhook; h++)
      if (cg.contains_hooks_[h].slot == k) hooks.push_back(h);
    const bool want_ascii = cg.ascii_slots_.count(k) != 0;
    std::vector<const VarlenOut*> flats;  // outputs that are this column's (mapped) bytes
    for (auto& vo : cg.varlen_outs_)
      if (vo.flat_slot == k) flats.push_back(&vo);
    const std::string K = std::to_string(k);
    if (sel) {
      s << "  const gdv_int32 sfl" << K << " = 0;\n";
      continue;
    }
    // one wave-uniform range test per tile makes every 8-byte read of these rows unchecked
    s << "  const gdv_int32 inb" << K << " = sd" << K << " + __builtin_amdgcn_readlane(ob" << K
      << "[GDV_U - 1], 63) + 8 <= slim" << K << " ? GDV_STR_INBUF : 0;\n";
    if (!flats.empty())
      s << "  const gdv_int32 so0_" << K << " = so" << K << "[0];  // the batch's first offset (flat outputs rebase by it)\n";
    if (hooks.empty() && !want_ascii && flats.empty()) {
      s << "  const gdv_int32 sfl" << K << " = inb" << K << ";\n";
      continue;
    }
    s << "  // ---- byte sweep of input " << k << ": the wave tile's rows are one contiguous span\n"
      << "  const gdv_int32 sp0" << K << " = __builtin_amdgcn_readfirstlane(oa" << K << "[0]);\n"
      << "  const gdv_int32 sp1" << K << " = __builtin_amdgcn_readlane(ob" << K << "[GDV_U - 1], 63);\n"
      << "  const gdv_int32 sb" << K << " = sp0" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + sp0" << K << ") & 15);\n"
      << "  const bool hm_ok" << K << " = sp1" << K << " - sb" << K << " <= GDV_SPAN_MAX;\n"
      << "  (void)hm_ok" << K << ";\n"
      << "  gdv_uint64 sacc" << K << " = 0;\n";
    for (int h : hooks) {
      const ContainsHook& hk = cg.contains_hooks_[h];
      const uint64_t mask = hk.needle.size() >= 8 ? ~0ull : ((1ull << (8 * hk.needle.size())) - 1);
      s << "  gdv_uint64* const hit" << h << " = lds_hit + " << h << " * GDV_HIT_WORDS;\n"
        << NeedleConstants(cg, h, mask);
    }
    s << "  for (gdv_int32 c = sb" << K << "; c < " << AblSel(64, "sb" + K, "sp1" + K) << "; c += 1024) {\n"
      << "    const gdv_int32 a = c + 16 * lane;\n"
      << "    gdv_uint64 w[2] = {0ull, 0ull};\n"
      << "    if (a < sp1" << K << ") __builtin_memcpy(w, __builtin_assume_aligned(sd" << K << " + a, 16), 16);\n"
      << "    sacc" << K << " |= w[0] | w[1];\n";
    if (!hooks.empty())
      s << "    gdv_uint64 tail = 0;  // lane 63's halo: the first 8 bytes of the next step\n"
        << "    if (lane == 63 && a + 16 < sp1" << K << ") tail = gdv_load8_raw(sd" << K << " + a + 16);\n";
    for (int h : hooks) {
      const ContainsHook& hk = cg.contains_hooks_[h];
      const uint64_t mask = hk.needle.size() >= 8 ? ~0ull : ((1ull << (8 * hk.needle.size())) - 1);
      const std::string H = std::to_string(h), M = std::to_string(hk.map);
      s << "    {\n"
        << "      const gdv_uint64 lo = gdv_map8(w[0], " << M << "), hi = gdv_map8(w[1], " << M << ");\n"
        << "      gdv_uint64 nx = gdv_next_lane(lo);\n"
        << "      if (lane == 63) nx = gdv_map8(tail, " << M << ");\n"
        << "      const gdv_uint32 m = " << (tl_ablation ? "(GDV_ABL & 1) ? (gdv_uint32)(lo >> 60) : " : "") << "gdv_match8(lo, hi, nd" << H << ", " << Hex64(mask) << ", ns0_" << H << ", ns1_" << H << ") |\n"
        << "                           (gdv_match8(hi, nx, nd" << H << ", " << Hex64(mask) << ", ns0_" << H << ", ns1_" << H
        << ") << 8);\n"
        << "      if (hm_ok" << K << " && a < sp1" << K << ") ((gdv_uint16*)hit" << H << ")[(a - sb" << K
        << ") >> 4] = (gdv_uint16)m;\n"
        << "    }\n";
    }
    s << "  }\n";
    // optimistic flat outputs: their place in the output is known from the input offsets alone, so
    // the span is copied right here, while the sweep's lines are still in L2 / L1.  (Moving the copy
    // behind the post of the tile totals, "into the shadow" of the scanner hand-off, measured
    // slower: 1.90 vs 1.78 ms, same box, profiles/r02_c5_tuning.txt.)
    for (auto* vo : flats)
      s << "  if (" << AblNot(8) << "optflat && (gdv_int64)sp1" << K << " - so0_" << K << " <= A.out[" << vo->e << "].cap)\n"
        << "    gdv_flat_copy(outd" << vo->e << " + (sp0" << K << " - so0_" << K << "), sd" << K << " + sp0" << K << ", sp1" << K
        << " - sp0" << K << ", " << vo->flat_map << ", lane);\n";
    if (want_ascii)
      s << "  const gdv_int32 sfl" << K << " = inb" << K << " | (__ballot((sacc" << K
        << " & GDV_B80) != 0) == 0 ? GDV_STR_ASCII : 0);\n";
    else
      s << "  const gdv_int32 sfl" << K << " = inb" << K << ";\n";
    if (!hooks.empty()) s << "  __builtin_amdgcn_wave_barrier();\n";
  }
}

// The byte sweep of wave-shaped kernels, one SUB-TILE (64 rows) at a time, at the top of the row
// loop: lanes over the bytes of the sub-tile's span — 16 B per lane and step, coalesced, the first
// step of the NEXT sub-tile already in flight while this one's rows are evaluated.  Per piece:
//   * tile-wide ASCII check (optimistic: the row bodies were compiled for ASCII; a byte >= 0x80
//     raises NOTASCII after the loop and the host re-runs the batch on the general kernel),
//   * '%needle%' match bits -> LDS bitmap of the sub-tile's span,
//   * flat outputs leave straight from the registers (a piece is stored by the sub-tile that holds
//     its last byte's predecessor: pieces that straddle two sub-tiles are stored exactly once),
//   * the bytes themselves -> the LDS MIRROR of the span, which the rows' staged copies read.
// Why per sub-tile and not per wave tile as in the first wave-shaped kernels: by the time the rows
// of a 512-row tile re-read their bytes (8-byte loads at the row's offset) the lines had left the
// XCD's L2 — 0.6 GB of extra fabric reads on C5, 1.40 x the algorithmic traffic
// (profiles/r03_c5_traffic.txt).  A sub-tile's span is small enough to keep in LDS (GDV_SUB_SPAN =
// 32 bytes per row; longer spans — wave-uniform — read HBM as before), so nothing is read twice.
// Exact variant, kernels that have no other reason to read column K's bytes (a ByteFree pre-pass):
// a bare sweep of the wave tile's span — 16 B per lane and step, OR-reduced — gives the tile's ASCII flag.
// Needs sp0K / sp1K (the span) and sdK / slimK in scope; defines inbK? no: sflK only.
std::string ExactAsciiTileFlag(const std::string& K) {
  std::ostringstream s;
  s << "  gdv_uint64 sacc" << K << " = 0;\n"
    << "  for (gdv_int32 c = sp0" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + sp0" << K << ") & 15); c < sp1" << K << "; c += 1024) {\n"
    << "    const gdv_int32 a = c + 16 * lane;\n"
    << "    gdv_uint64 w[2] = {0ull, 0ull};\n"
    << "    if (a < sp1" << K << ") __builtin_memcpy(w, __builtin_assume_aligned(sd" << K << " + a, 16), 16);\n"
    << "    sacc" << K << " |= w[0] | w[1];\n"
    << "  }\n"
    << "  const gdv_int32 sfl" << K << " = (sd" << K << " + sp1" << K << " + 8 <= slim" << K << " ? GDV_STR_INBUF : 0) |\n"
    << "                       (__ballot((sacc" << K << " & GDV_B80) != 0) == 0 ? GDV_STR_ASCII : 0);\n";
  return s.str();
}

struct WaveSweepText {
  std::string prologue;   // before the row loop
  std::string per_sub;    // top of the row loop's body (u = the sub-tile)
  std::string epilogue;   // after the row loop
};
// sg (round 6): sub-tiles whose spans ONE sweep covers (GDV_SG; 1 = one sub-tile at a time, rounds 3-5).  A 64-row span of
// 12-byte rows fills three quarters of a 1024-byte step; four of them fill three steps exactly.
void EmitWaveSweep(CodeGen& cg, KernelPlan* plan, int mirror_slot, bool prepass, WaveSweepText* out, int sg = 1) {
  std::ostringstream s, b, e;
  const bool grouped = sg > 1;
  const std::string SG = std::to_string(sg);
  const int nin = plan->layout.n_in;
  const int nhook = static_cast<int>(cg.contains_hooks_.size());
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (!(t.is_varlen() && cg.needs_values_[k])) continue;
    std::vector<int> hooks;
    for (int h = 0; h < nhook; h++)
      if (cg.contains_hooks_[h].slot == k) hooks.push_back(h);
    const bool want_ascii = cg.ascii_slots_.count(k) != 0;
    std::vector<const VarlenOut*> flats;  // outputs that are this column's (mapped) bytes
    for (auto& vo : cg.varlen_outs_)
      if (vo.flat_slot == k) flats.push_back(&vo);
    const bool mirror = mirror_slot == k && !prepass;  // (a pre-pass needs the match bits only)
    const std::string K = std::to_string(k);
    if (prepass && mirror_slot != k) {
      // a pre-pass sweeps nothing but the column whose replace() counts matches in the bitmap; views
      // carry the flags the main kernel will give them (the optimistic ASCII flag where consulted)
      s << "  const gdv_int32 sp1" << K << " = so" << K << "[last_tile ? n : rbase + 64 * GDV_U];\n";
      if (want_ascii && cg.exact_ascii_) {
        s << "  const gdv_int32 sp0" << K << " = so" << K << "[rbase];\n" << ExactAsciiTileFlag(K);
      } else {
        s << "  const gdv_int32 sfl" << K << " = (sd" << K << " + sp1" << K << " + 8 <= slim" << K << " ? GDV_STR_INBUF : 0)"
          << (want_ascii ? " | GDV_STR_ASCII" : "") << ";\n";
      }
      continue;
    }
    // the tile's span: its ends come from two scalar loads; one wave-uniform range test per tile
    // makes every 8-byte read of these rows unchecked
    s << "  const gdv_int32 sp0" << K << " = so" << K << "[rbase];\n"
      << "  const gdv_int32 sp1" << K << " = so" << K << "[last_tile ? n : rbase + 64 * GDV_U];\n"
      << "  const gdv_int32 inb" << K << " = sd" << K << " + sp1" << K << " + 8 <= slim" << K << " ? GDV_STR_INBUF : 0;\n";
    if (!flats.empty())
      s << "  const gdv_int32 so0_" << K << " = so" << K << "[0];  // the batch's first offset (flat outputs rebase by it)\n";
    if (hooks.empty() && !want_ascii && flats.empty()) {
      s << "  const gdv_int32 sfl" << K << " = inb" << K << ";\n";
      continue;
    }
    s << "  // ---- byte sweep of input " << k << ", one sub-tile at a time (inside the row loop)\n"
      << "  gdv_uint64 sacc" << K << " = 0;\n";
    for (int h : hooks) {
      const ContainsHook& hk = cg.contains_hooks_[h];
      const uint64_t mask = hk.needle.size() >= 8 ? ~0ull : ((1ull << (8 * hk.needle.size())) - 1);
      s << "  gdv_uint64* const hit" << h << " = lds_hit + " << h << " * GDV_HIT_WORDS;\n"
        << NeedleConstants(cg, h, mask);
    }
    if (mirror)
      s << "  gdv_lds_u8* const mir" << K << " = (gdv_lds_u8*)lds_in;  // LDS mirror of the current sub-tile's span\n";
    // the first piece of sub-tile 0 (every later sub-tile's first piece is loaded one iteration ahead)
    s << "  gdv_uint64 wn" << K << "[2] = {0ull, 0ull};\n"
      << (hooks.empty() ? "" : "  gdv_uint64 tn" + K + " = 0;  // lane 63's halo (the 8 bytes behind its piece), loaded WITH the piece\n")
      << "  {\n"
      << "    const gdv_int32 e0 = GDV_U > " << SG << " ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > " << SG << " ? " << SG << " : 0]) : sp1" << K << ";\n"
      << "    const gdv_int32 b0 = sp0" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + sp0" << K << ") & 15);\n"
      << "    if (" << AblNot(64) << "b0 + 16 * lane < e0) __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K
      << " + b0 + 16 * lane, 16), 16);\n"
      << (hooks.empty() ? "" : "    if (" + AblNot(64) + "lane == 63 && b0 + 1024 < e0) tn" + K + " = gdv_load8_raw(sd" + K + " + b0 + 1024);\n")
      << "  }\n";
    // the two ragged ends of the tile's span (whole 16-byte pieces that overlap their neighbours)
    for (auto* vo : flats)
      s << "  const gdv_int32 fcap" << vo->e << " = (gdv_int32)(A.out[" << vo->e << "].cap > 0x7fffffff ? 0x7fffffff : A.out[" << vo->e
        << "].cap);\n";
    for (auto* vo : flats)
      s << "  " << AblIf(8) << "gdv_sweep_edges(outd" << vo->e << ", sd" << K << ", sp0" << K << ", sp1" << K << ", so0_" << K
        << ", " << vo->flat_map << ", A.out[" << vo->e << "].cap, lane);\n";
    if (want_ascii && cg.exact_ascii_)
      // exact variant: the flag of the CURRENT sub-tile, set by its sweep at the top of the row loop
      s << "  const gdv_int32 sfl" << K << " = inb" << K << ";  // (the rows' views take a per-row flag: gdv_row_has_high)\n"
        << "  gdv_uint64 sawhi" << K << " = 0;  // OR of every byte swept so far (reported as SAWUTF8)\n"
        << "  bool hi8_" << K << " = false;  // the current sub-tile's span holds a byte >= 0x80\n"
        << "  gdv_uint64* const cb" << K << " = lds_hit + " << cg.CbIndex(k) << " * GDV_HIT_WORDS;  // continuation-byte bitmap of the sub-tile's span\n";
    else if (want_ascii)
      // optimistic ASCII (the pre-pass computed the lengths under it): the flag is a compile-time
      // fact for the row bodies — every general UTF-8 path folds away
      s << "  const gdv_int32 sfl" << K << " = inb" << K << " | GDV_STR_ASCII;\n";
    else
      s << "  const gdv_int32 sfl" << K << " = inb" << K << ";\n";

    // ---- per sub-tile
    if (grouped) {
      // the span of a GROUP of GDV_SG sub-tiles, swept when its first sub-tile comes up; ssK / seK / sbK / hm_okK stay what
      // they are for the group's other sub-tiles (the rows address the mirror and the bitmaps relative to sbK)
      s << "  gdv_int32 ss" << K << " = 0, se" << K << " = 0, sb" << K << " = 0;\n"
        << "  bool hm_ok" << K << " = false;\n"
        << "  (void)ss" << K << "; (void)hm_ok" << K << ";\n";
      b << "    // byte sweep of the span of this group of " << SG << " sub-tiles of input " << k << " (every " << SG << "th iteration)\n"
        << "    if ((u & (" << SG << " - 1)) == 0) {\n"
        << "    ss" << K << " = __builtin_amdgcn_readfirstlane(oa" << K << "[0]);\n"
        << "    se" << K << " = u + " << SG << " < GDV_U ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > " << SG << " ? " << SG << " : 0]) : sp1" << K << ";\n"
        << "    sb" << K << " = ss" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + ss" << K << ") & 15);\n"
        << "    hm_ok" << K << " = se" << K << " - sb" << K << " <= GDV_SUB_SPAN;  // wave-uniform: the span fits the LDS bitmap / mirror\n";
    } else {
      b << "    // byte sweep of this sub-tile's span of input " << k << "\n"
        << "    const gdv_int32 ss" << K << " = __builtin_amdgcn_readfirstlane(oa" << K << "[0]);\n"
        << "    const gdv_int32 se" << K << " = u + 1 < GDV_U ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > 1 ? 1 : 0]) : sp1" << K << ";\n"
        << "    const gdv_int32 sb" << K << " = ss" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + ss" << K << ") & 15);\n"
        << "    const bool hm_ok" << K << " = se" << K << " - sb" << K << " <= GDV_SUB_SPAN;  // wave-uniform: the span fits the LDS bitmap / mirror\n"
        << "    (void)hm_ok" << K << ";\n";
    }
    b << "    for (gdv_int32 c = sb" << K << "; c < " << AblSel(64, "sb" + K, "se" + K) << "; c += 1024) {\n"
      << "      const gdv_int32 a = c + 16 * lane;\n"
      << "      const gdv_uint64 w[2] = {wn" << K << "[0], wn" << K << "[1]};\n"
      << "      wn" << K << "[0] = 0ull; wn" << K << "[1] = 0ull;\n"
      << (hooks.empty() ? "" : "      const gdv_uint64 tail = tn" + K + ";  // (lane 63 only) — loaded one step ahead like the piece: nothing here waits for a load it has just issued\n"
                               "      tn" + K + " = 0ull;\n")
      << "      if (a + 1024 < se" << K << ") __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K << " + a + 1024, 16), 16);\n"
      << (hooks.empty() ? "" : "      if (lane == 63 && a + 1024 + 16 < se" + K + ") tn" + K + " = gdv_load8_raw(sd" + K + " + a + 1024 + 16);\n")
      << "      sacc" << K << " |= w[0] | w[1];\n";
    if (want_ascii && cg.exact_ascii_) {
      cg.row_ascii_slots_.insert(k);
      b << "      { const gdv_uint64 hbw = __ballot(((w[0] | w[1]) & GDV_B80) != 0);\n"
        << "        const gdv_uint32 cm = hbw != 0 ? gdv_cont_mask16(w[0], w[1]) : 0u;  // (wave-uniform branch: ASCII steps skip the packing)\n"
        << "        if (hm_ok" << K << " && a < se" << K << ") ((gdv_uint16*)cb" << K << ")[(a - sb" << K << ") >> 4] = (gdv_uint16)cm; }\n";
    }
    for (int h : hooks) {
      const ContainsHook& hk = cg.contains_hooks_[h];
      const uint64_t mask = hk.needle.size() >= 8 ? ~0ull : ((1ull << (8 * hk.needle.size())) - 1);
      const std::string H = std::to_string(h), M = std::to_string(hk.map);
      b << "      {\n"
        << "        const gdv_uint64 lo = gdv_map8(w[0], " << M << "), hi = gdv_map8(w[1], " << M << ");\n"
        << "        gdv_uint64 nx = gdv_next_lane(lo);\n"
        << "        if (lane == 63) nx = gdv_map8(tail, " << M << ");\n"
        << "        const gdv_uint32 m = " << (tl_ablation ? "(GDV_ABL & 1) ? (gdv_uint32)(lo >> 60) : " : "") << "gdv_match8(lo, hi, nd" << H << ", " << Hex64(mask) << ", ns0_" << H << ", ns1_" << H << ") |\n"
        << "                             (gdv_match8(hi, nx, nd" << H << ", " << Hex64(mask) << ", ns0_" << H << ", ns1_" << H
        << ") << 8);\n"
        << "        if (hm_ok" << K << " && a < se" << K << ") ((gdv_uint16*)hit" << H << ")[(a - sb" << K
        << ") >> 4] = (gdv_uint16)m;\n"
        << "      }\n";
    }
    if (mirror)
      b << "      if (hm_ok" << K << " && a < se" << K << ") __builtin_memcpy(mir" << K << " + (a - sb" << K << "), w, 16);\n";
    // a piece is stored by the sub-tile in whose span it ENDS (a + 16 <= se): the piece that
    // straddles two sub-tiles is the next one's first piece; the tile's own ends: gdv_sweep_edges
    for (auto* vo : flats)
      b << "      " << AblIf(8) << "gdv_sweep_store32(outd" << vo->e << ", a - so0_" << K
        << ", w, " << vo->flat_map << ", a >= sp0" << K << " && a + 16 <= se" << K << ", fcap" << vo->e << ");\n";
    const std::string SG2 = std::to_string(2 * sg);
    b << "    }\n"
      << "    if (u + " << SG << " < GDV_U) {  // the first piece of the next " << (grouped ? "group's" : "sub-tile's") << " span\n"
      << "      const gdv_int32 e2 = u + " << SG2 << " < GDV_U ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > " << SG2 << " ? " << SG2 << " : 0]) : sp1" << K << ";\n"
      << "      const gdv_int32 nb = se" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + se" << K << ") & 15);\n"
      << "      if (" << AblNot(64) << "nb + 16 * lane < e2) __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K
      << " + nb + 16 * lane, 16), 16);\n"
      << (hooks.empty() ? "" : "      if (" + AblNot(64) + "lane == 63 && nb + 1024 < e2) tn" + K + " = gdv_load8_raw(sd" + K + " + nb + 1024);\n")
      << "    }\n";
    if (want_ascii && cg.exact_ascii_)
      // (the sweep of a sub-tile covers whole 16-byte pieces: a few bytes of the neighbouring rows may
      // clear the flag needlessly — the general paths are exact for ASCII rows too)
      b << "    hi8_" << K << " = __ballot((sacc" << K << " & GDV_B80) != 0) != 0;  // this sub-tile's span holds a byte >= 0x80\n"
        << "    sawhi" << K << " |= sacc" << K << ";\n"
        << "    sacc" << K << " = 0;\n";
    if (!hooks.empty() || mirror || (want_ascii && cg.exact_ascii_)) b << "    __builtin_amdgcn_wave_barrier();\n";
    if (grouped) b << "    }  // (group's first sub-tile)\n";

    // ---- after the loop
    if (want_ascii && cg.exact_ascii_ && !prepass)
      e << "  if (__ballot((sawhi" << K << " & GDV_B80) != 0) != 0 && lane == 0) gdv_raise_bits(A.err, GDV_ERR_SAWUTF8);\n";
    else if (want_ascii && !prepass)  // (the main kernel raises it)
      e << "  if (__ballot((sacc" << K << " & GDV_B80) != 0) != 0 && lane == 0) gdv_raise_bits(A.err, GDV_ERR_NOTASCII);\n";
  }
  out->prologue = s.str();
  out->per_sub = b.str();
  out->epilogue = e.str();
}

// The byte sweep of wave-shaped kernels WITHOUT an LDS mirror: the whole wave tile's span in one
// go, before the row loop (full 1024-byte steps: cheaper per byte than the per-sub-tile sweep,
// whose steps are three quarters full on average).  Taken when no staged copy would read the
// mirror — flat-only plans, outputs that are not readable views (reverse, replace, digits).
void EmitWaveTileSweep(std::ostringstream& s, CodeGen& cg, KernelPlan* plan, std::string* epilogue) {
  std::ostringstream e;
  const int nin = plan->layout.n_in;
  const int nhook = static_cast<int>(cg.contains_hooks_.size());
  for (int k = 0; k < nin; k++) {
    const DataType& t = cg.schema_[plan->input_fields[k]].type;
    if (!(t.is_varlen() && cg.needs_values_[k])) continue;
    std::vector<int> hooks;
    for (int h = 0; h < nhook; h++)
      if (cg.contains_hooks_[h].slot == k) hooks.push_back(h);
    const bool want_ascii = cg.ascii_slots_.count(k) != 0;
    std::vector<const VarlenOut*> flats;  // outputs that are this column's (mapped) bytes
    for (auto& vo : cg.varlen_outs_)
      if (vo.flat_slot == k) flats.push_back(&vo);
    const std::string K = std::to_string(k);
    if (cg.selection()) {
      // selected rows are not one span of bytes: nothing to sweep, no tile-wide fact about them — every row
      // function takes its general (UTF-8-exact, range-checked) path, as in the scanner-shaped kernel
      // ... except the OPTIMISTIC one the pre-pass made: a function that consults the ASCII flag gets it set here too
      // (so both kernels compute the same lengths) and every row, which reads its bytes in this kernel anyway, checks it
      if (want_ascii) {
        s << "  const gdv_int32 sfl" << K << " = GDV_STR_ASCII;\n"
          << "  gdv_uint64 nasc" << K << " = 0;  // rows that turned out to hold a byte >= 0x80\n";
        e << "  if (nasc" << K << " != 0 && lane == 0) gdv_raise_bits(A.err, GDV_ERR_NOTASCII);\n";
      } else {
        s << "  const gdv_int32 sfl" << K << " = 0;\n";
      }
      continue;
    }
    // the span's ends come from two scalar loads (the sweep does not wait for the offsets' vector
    // loads); one wave-uniform range test per tile makes every 8-byte read of these rows unchecked
    s << "  const gdv_int32 sp0" << K << " = so" << K << "[rbase];\n"
      << "  const gdv_int32 sp1" << K << " = so" << K << "[last_tile ? n : rbase + 64 * GDV_U];\n"
      << "  const gdv_int32 inb" << K << " = sd" << K << " + sp1" << K << " + 8 <= slim" << K << " ? GDV_STR_INBUF : 0;\n";
    if (!flats.empty())
      s << "  const gdv_int32 so0_" << K << " = so" << K << "[0];  // the batch's first offset (flat outputs rebase by it)\n";
    if (hooks.empty() && !want_ascii && flats.empty()) {
      s << "  const gdv_int32 sfl" << K << " = inb" << K << ";\n";
      continue;
    }
    s << "  // ---- byte sweep of input " << k << ": the wave tile's rows are one contiguous span\n"
      << "  const gdv_int32 sb" << K << " = sp0" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + sp0" << K << ") & 15);\n"
      << "  const bool hm_ok" << K << " = sp1" << K << " - sb" << K << " <= GDV_SPAN_MAX;\n"
      << "  (void)hm_ok" << K << ";\n"
      << "  gdv_uint64 sacc" << K << " = 0;\n";
    for (int h : hooks) {
      const ContainsHook& hk = cg.contains_hooks_[h];
      const uint64_t mask = hk.needle.size() >= 8 ? ~0ull : ((1ull << (8 * hk.needle.size())) - 1);
      s << "  gdv_uint64* const hit" << h << " = lds_hit + " << h << " * GDV_HIT_WORDS;\n"
        << NeedleConstants(cg, h, mask);
    }
    for (auto* vo : flats)
      s << "  const gdv_int32 fcap" << vo->e << " = (gdv_int32)(A.out[" << vo->e << "].cap > 0x7fffffff ? 0x7fffffff : A.out[" << vo->e
        << "].cap);\n";
    // software-pipelined: the next step's 16 bytes — and lane 63's halo, the 8 bytes behind its
    // piece — are in flight while this step's are matched / stored
    s << "  gdv_uint64 wn" << K << "[2] = {0ull, 0ull};\n"
      << (hooks.empty() ? "" : "  gdv_uint64 tn" + K + " = 0;\n")
      << "  if (sb" << K << " + 16 * lane < sp1" << K << ") __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K << " + sb" << K << " + 16 * lane, 16), 16);\n"
      << (hooks.empty() ? "" : "  if (lane == 63 && sb" + K + " + 1024 < sp1" + K + ") tn" + K + " = gdv_load8_raw(sd" + K + " + sb" + K + " + 1024);\n")
      << "  for (gdv_int32 c = sb" << K << "; c < " << AblSel(64, "sb" + K, "sp1" + K) << "; c += 1024) {\n"
      << "    const gdv_int32 a = c + 16 * lane;\n"
      << "    const gdv_uint64 w[2] = {wn" << K << "[0], wn" << K << "[1]};\n"
      << "    wn" << K << "[0] = 0ull; wn" << K << "[1] = 0ull;\n"
      << (hooks.empty() ? "" : "    const gdv_uint64 tail = tn" + K + ";  // (lane 63 only)\n    tn" + K + " = 0ull;\n")
      << "    if (a + 1024 < sp1" << K << ") __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K << " + a + 1024, 16), 16);\n"
      << (hooks.empty() ? "" : "    if (lane == 63 && a + 1024 + 16 < sp1" + K + ") tn" + K + " = gdv_load8_raw(sd" + K + " + a + 1024 + 16);\n")
      << "    sacc" << K << " |= w[0] | w[1];\n";
    for (int h : hooks) {
      const ContainsHook& hk = cg.contains_hooks_[h];
      const uint64_t mask = hk.needle.size() >= 8 ? ~0ull : ((1ull << (8 * hk.needle.size())) - 1);
      const std::string H = std::to_string(h), M = std::to_string(hk.map);
      s << "    {\n"
        << "      const gdv_uint64 lo = gdv_map8(w[0], " << M << "), hi = gdv_map8(w[1], " << M << ");\n"
        << "      gdv_uint64 nx = gdv_next_lane(lo);\n"
        << "      if (lane == 63) nx = gdv_map8(tail, " << M << ");\n"
        << "      const gdv_uint32 m = " << (tl_ablation ? "(GDV_ABL & 1) ? (gdv_uint32)(lo >> 60) : " : "") << "gdv_match8(lo, hi, nd" << H << ", " << Hex64(mask) << ", ns0_" << H << ", ns1_" << H << ") |\n"
        << "                           (gdv_match8(hi, nx, nd" << H << ", " << Hex64(mask) << ", ns0_" << H << ", ns1_" << H
        << ") << 8);\n"
        << "      if (hm_ok" << K << " && a < sp1" << K << ") ((gdv_uint16*)hit" << H << ")[(a - sb" << K
        << ") >> 4] = (gdv_uint16)m;\n"
        << "    }\n";
    }
    // flat outputs leave straight from the sweep's registers; a piece's bytes outside this wave's
    // span [sp0, sp1) belong to the neighbouring tiles
    for (auto* vo : flats)
      s << "    " << AblIf(8) << "gdv_sweep_store32(outd" << vo->e << ", a - so0_" << K
        << ", w, " << vo->flat_map << ", a >= sp0" << K << " && a + 16 <= sp1" << K << ", fcap" << vo->e << ");\n";
    s << "  }\n";
    for (auto* vo : flats)
      s << "  " << AblIf(8) << "gdv_sweep_edges(outd" << vo->e << ", sd" << K << ", sp0" << K << ", sp1" << K << ", so0_" << K
        << ", " << vo->flat_map << ", A.out[" << vo->e << "].cap, lane);\n";
    if (want_ascii && cg.exact_ascii_) {
      // exact variant: the tile's flag is what its sweep found
      s << "  const bool hi8_" << K << " = __ballot((sacc" << K << " & GDV_B80) != 0) != 0;\n"
        << "  const gdv_int32 sfl" << K << " = inb" << K << " | (hi8_" << K << " ? 0 : GDV_STR_ASCII);\n";
      e << "  if (hi8_" << K << " && lane == 0) gdv_raise_bits(A.err, GDV_ERR_SAWUTF8);\n";
    } else if (want_ascii) {
      // optimistic ASCII (the pre-pass computed the lengths under it): the flag is a compile-time
      // fact for the row bodies — every general UTF-8 path folds away — and a tile that breaks it
      // raises NOTASCII: the host re-runs the batch on the exact variant of these kernels
      s << "  const gdv_int32 sfl" << K << " = inb" << K << " | GDV_STR_ASCII;\n";
      e << "  if (__ballot((sacc" << K << " & GDV_B80) != 0) != 0 && lane == 0) gdv_raise_bits(A.err, GDV_ERR_NOTASCII);\n";
    } else {
      s << "  const gdv_int32 sfl" << K << " = inb" << K << ";\n";
    }
    if (!hooks.empty()) s << "  __builtin_amdgcn_wave_barrier();\n";
  }
  *epilogue = e.str();
}

// ------------------------------------------------------------------ string plans
// Kernels that read or write var-len columns use their own skeleton (round 2):
//   tile     one workgroup = GDV_WAVES waves x GDV_U sub-tiles x 64 rows; a wave's rows occupy ONE
//            contiguous span of each var-len input's data buffer
//   sweep    lanes over the BYTES of that span: tile-wide ASCII flag, '%needle%' match bitmaps
//   rows     lane = row: the fused expression bodies; var-len results are kept as views
//   offsets  per-wave DPP scan of the lengths -> workgroup totals -> ONE granule posted to the
//            scanner wave (workgroup 0), ONE granule polled for the tile's exclusive prefix
//   bytes    staged in LDS while waiting, flushed coalesced; or streamed flat when the output
//            IS the (mapped) input span
// Single launch, inputs read once (round 1: two passes, 1.43 x the algorithmic traffic).
Status AssembleStrings(CodeGen& cg, KernelPlan* plan, const std::vector<std::string>& expr_strings,
                       const WordAccumulators& accs, const std::string& decls_before_loop,
                       const std::string& epilogue_after_loop) {
  plan->input_fields = cg.input_fields_;
  plan->input_needs_values = cg.needs_values_;
  plan->input_needs_validity = cg.needs_validity_;
  plan->can_raise = cg.can_raise_;
  plan->string_skeleton = true;
  for (size_t k = 0; k < plan->input_fields.size(); k++)
    plan->has_varlen_input |= cg.schema_[plan->input_fields[k]].type.is_varlen() && cg.needs_values_[k];
  plan->layout.n_in = static_cast<int>(plan->input_fields.size());
  plan->layout.n_out = static_cast<int>(plan->output_types.size());
  const int nv = static_cast<int>(cg.varlen_outs_.size());
  const int ng = (nv + 1) / 2;
  int nstage = 0;                              // LDS staging windows per wave
  for (auto& vo : cg.varlen_outs_) nstage = std::max(nstage, vo.window + 1);
  const int nhook = static_cast<int>(cg.contains_hooks_.size());
  plan->num_varlen_outputs = nv;
  for (auto& vo : cg.varlen_outs_) plan->has_flat_output |= vo.flat_slot >= 0;

  Assembler as{cg, plan, {}};
  as.Header(expr_strings);
  std::ostringstream& s = as.src;
  s << "#define GDV_NV " << nv << "\n#define GDV_NG " << ng << "\n#define GDV_NSTAGE " << std::max(nstage, 1)
    << "\n#define GDV_NHOOK " << std::max(nhook, 1) << "\n"
    << "#define GDV_HIT_WORDS (GDV_SPAN_MAX / 64 + 4)\n"
    << "constexpr bool FULL = false;  // string tiles test `live` at run time (one code path)\n"
    << "#define GDV_OPTFLAT GDV_OPTFLAT_VALUE\n"
    << "#define GDV_STAGE_COPY(dst, v) gdv_stage_copy(dst, v)\n"
    << AblDefine()
    << "#define GDV_OUT(e, v) if (live) "
    << (plan->opts.nontemporal ? "gdv_stnt" : "gdv_st") << "(out##e, row, (v))\n";

  s << "GDV_DEV void gdv_tile(const gdv_args& A, const gdv_int64 tile, const gdv_int64 ntiles, const int lane,\n"
    << "                      const int wave, gdv_uint8* lds_out, gdv_uint64* lds_hit, gdv_uint32 (*lds_tot)[GDV_NV > 0 ? GDV_NV : 1],\n"
    << "                      gdv_uint64* lds_base) {\n"
    << "  (void)lds_out; (void)lds_hit; (void)lds_tot; (void)lds_base; (void)ntiles;\n"
    << "  gdv_ctx ctx{A.err};\n"
    << "  (void)ctx;\n"
    << "  const gdv_uint8* const gdv_cst = (const gdv_uint8*)A.aux0;  // the plan's constant block\n"
    << "  (void)gdv_cst;\n"
    << "  const gdv_int64 n = GDV_ROWS(A);\n"
    << "  const gdv_int64 wbase = (tile * GDV_WAVES + wave) * GDV_U;\n"
    << "  const gdv_int64 rbase = wbase * 64;\n"
    << "  constexpr bool optflat = GDV_OPTFLAT != 0;  // flat outputs: offsets = input offsets, bytes copied after the sweep\n"
    << "  (void)optflat;\n";
  if (nv == 0) s << "  if (rbase >= n) return;  // nothing but dead rows (no workgroup barrier below)\n";
  EmitStringPointersAndLoads(s, cg, plan, true);
  EmitStringSweep(s, cg, plan);

  // ---- row phase
  s << "  // ---- rows: fused expression bodies (value for every row, validity per word)\n";
  for (auto& a : accs.names) s << "  gdv_uint64 " << a << " = 0;\n";
  s << decls_before_loop;
  if (nv > 0)
    s << "  bool need_direct = false;\n"
      << "  // pass 0: lengths, offsets, staged / flat bytes.  pass 1 (rare): rows of outputs whose bytes\n"
      << "  // neither fit the LDS window nor are a flat span are recomputed and copied straight to HBM.\n"
      << "  for (int pass = 0; pass < 2; pass++) {\n"
      << "  if (pass == 1 && !need_direct) break;\n";
  else
    s << "  constexpr int pass = 0;\n  (void)pass;\n";
  EmitStringRowLoop(s, cg, plan);
  for (auto& vo : cg.varlen_outs_) s << "    gdv_rot(lc" << vo.e << ");\n";
  s << "  }\n";
  if (nv > 0) s << "  if (pass == 1) break;\n";
  for (auto& vo : cg.varlen_outs_)
    if (vo.flat_slot >= 0)
      s << "  if (optflat && fb" << vo.e << " != 0 && lane == 0) gdv_raise_bits(A.err, GDV_ERR_NOTFLAT);\n";
  s << epilogue_after_loop;

  // ---- var-len outputs
  if (nv > 0) {
    // every var-len output is a flat candidate: the optimistic variant needs no totals, no scanner
    // and no barrier at all — offsets and bytes are already out
    bool all_flat = true;
    for (auto& vo : cg.varlen_outs_) all_flat = all_flat && vo.flat_slot >= 0;
    if (all_flat) s << "  if (!optflat) {  // (all outputs flat: this whole block exists in the general variant only)\n";
    s << "  // ---- var-len outputs: workgroup totals -> one granule to the scanner\n"
      << "  if (lane == 0) {\n";
    for (int v = 0; v < nv; v++)
      s << "    lds_tot[wave][" << v << "] = (gdv_uint32)run" << cg.varlen_outs_[v].e << ";\n";
    s << "  }\n  __syncthreads();\n"
      << "  gdv_uint64 before[GDV_NV], all[GDV_NV];\n"
      << "#pragma unroll\n  for (int v = 0; v < GDV_NV; v++) { before[v] = 0; all[v] = 0; }\n"
      << "#pragma unroll\n  for (int w = 0; w < GDV_WAVES; w++) {\n"
      << "#pragma unroll\n    for (int v = 0; v < GDV_NV; v++) {\n"
      << "      const gdv_uint32 t = lds_tot[w][v];\n      all[v] += t;\n      before[v] += w < wave ? t : 0u;\n    }\n  }\n"
      << "  gdv_uint64* const lb_agg = A.mask;\n  gdv_uint64* const lb_pre = A.mask + (gdv_int64)GDV_NG * ntiles;\n"
      << "  if (threadIdx.x == 0) {\n";
    for (int g = 0; g < ng; g++)
      s << "    gdv_lb_post(lb_agg, ntiles, tile, " << g << ", all[" << 2 * g << "], "
        << (2 * g + 1 < nv ? "all[" + std::to_string(2 * g + 1) + "]" : std::string("0ull")) << ");\n";
    s << "  }\n";
    s << "  if (threadIdx.x == 0) {\n"
      << "#pragma unroll\n    for (int g = 0; g < GDV_NG; g++) lds_base[g] = " << AblSel(32, "(gdv_uint64)tile * 4000", "gdv_lb_wait(lb_pre, ntiles, tile, g, A.err)") << ";\n"
      << "  }\n  __syncthreads();\n";
    for (int v = 0; v < nv; v++) {
      const VarlenOut& vo = cg.varlen_outs_[v];
      const std::string E = std::to_string(vo.e);
      s << (vo.flat_slot >= 0 ? "  if (!optflat) {\n" : "  {\n")
        << "    const gdv_int64 base = (gdv_int64)((lds_base[" << v / 2 << "] >> " << 31 * (v % 2)
        << ") & GDV_LB_M31) + (gdv_int64)before[" << v << "];\n"
        << "    const bool fits = run" << E << " < 0x7fffffff && base + run" << E << " <= A.out[" << E << "].cap;\n"
        << "#pragma unroll\n    for (int u = 0; u < GDV_U; u++) {\n"
        << "      const gdv_int64 row = rbase + u * 64 + lane;\n"
        << "      if (row < n) outo" << E << "[row] = (gdv_int32)(base + lc" << E << "[u]);\n"
        << "    }\n"
        << "    if (fits) {\n";
      if (vo.flat_slot >= 0) {
        const std::string K = std::to_string(vo.flat_slot);
        s << "      if (fb" << E << " == 0) {  // no row dropped: the output IS the mapped input span\n"
          << "        gdv_flat_copy(outd" << E << " + base, sd" << K << " + __builtin_amdgcn_readfirstlane(oa" << K
          << "[0]), run" << E << ", " << vo.flat_map << ", lane);\n"
          << "      } else {\n";
      } else if (vo.window >= 0) {
        s << "      if (run" << E << " <= GDV_OUT_WIN) {\n"
          << "        " << AblIf(16) << "gdv_flush_out(outd" << E << " + base, win" << E << ", run" << E << ", lane);\n"
          << "      } else {\n";
      } else {
        s << "      {\n";
      }
      s << "        dir" << E << " = true;\n        dbase" << E << " = base;\n        need_direct = true;\n"
        << "      }\n    }\n  }\n";
    }
    s << "  if ((gdv_int64)gridDim.x - 1 < ntiles) __syncthreads();  // serial-safe launches only: the LDS hand-off words are reused by the next tile\n";
    if (all_flat) s << "  }\n";
    s
      << "  }  // pass\n";
  }
  s << "}\n\n";

  // ---- kernel
  s << "#ifndef GDV_STRING_KERNEL_ATTR\n#define GDV_STRING_KERNEL_ATTR\n#endif\n"
    << "extern \"C\" __global__ void GDV_STRING_KERNEL_ATTR __launch_bounds__(GDV_WAVES * 64) GDV_KERNEL_NAME(const gdv_args A) {\n"
    << "  const int lane = threadIdx.x & 63;\n"
    << "  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
    << "  __shared__ __attribute__((aligned(16))) gdv_uint8 gdv_lds_out[GDV_WAVES][GDV_NSTAGE * (GDV_OUT_WIN + 16)];\n"
    << "  __shared__ __attribute__((aligned(16))) gdv_uint64 gdv_lds_hit[GDV_WAVES][GDV_NHOOK * GDV_HIT_WORDS];\n"
    << "  __shared__ gdv_uint32 gdv_lds_tot[GDV_WAVES][GDV_NV > 0 ? GDV_NV : 1];\n"
    << "  __shared__ gdv_uint64 gdv_lds_base[GDV_NG > 0 ? GDV_NG : 1];\n"
    << "  const gdv_int64 ntiles = (GDV_ROWS(A) + 64 * GDV_U * GDV_WAVES - 1) / (64 * GDV_U * GDV_WAVES);\n";
  if (nv > 0) {
    s << "  // workgroup 0 is the scanner of the tile totals; workers are workgroups 1..\n"
      << "  if (blockIdx.x == 0) {\n"
      << "    if (wave == 0) {\n"
      << "      gdv_uint64* const totals = (gdv_uint64*)A.counts;\n"
      << "      " << (plan->has_flat_output && [&] { for (auto& vo : cg.varlen_outs_) if (vo.flat_slot < 0) return false; return true; }() ? "if (!GDV_OPTFLAT) " : "")
      << "gdv_scanner<GDV_NG>(A.mask, A.mask + (gdv_int64)GDV_NG * ntiles, ntiles, totals, A.err, lane);\n"
      << "      if (lane == 0) {\n";
    for (int v = 0; v < nv; v++) {
      const VarlenOut& vo = cg.varlen_outs_[v];
      if (vo.flat_slot >= 0)
        s << "        if (GDV_OPTFLAT) { const gdv_int32* so = A.in[" << vo.flat_slot << "].offsets; totals[" << v
          << "] = (gdv_uint64)(so[GDV_ROWS(A)] - so[0]); }\n";
      s << "        A.out[" << vo.e << "].offsets[GDV_ROWS(A)] = (gdv_int32)(totals[" << v
        << "] > GDV_LB_M31 ? GDV_LB_M31 : totals[" << v << "]);\n";
    }
    s << "      }\n    }\n    return;\n  }\n"
      << "  for (gdv_int64 tile = (gdv_int64)blockIdx.x - 1; tile < ntiles; tile += (gdv_int64)gridDim.x - 1)\n";
  } else {
    s << "  for (gdv_int64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x)\n";
  }
  s << "    gdv_tile(A, tile, ntiles, lane, wave, gdv_lds_out[wave], gdv_lds_hit[wave], gdv_lds_tot, gdv_lds_base);\n"
    << "}\n";

  // Two variants of the same text: GDV_OPTFLAT = 1 (flat outputs taken optimistically; the one
  // that runs) and, for plans that have flat outputs, GDV_OPTFLAT = 0 (every output through the
  // scan; compiled only if a batch ever raises NOTFLAT).
  auto finish = [&](const std::string& tmpl, const char* optflat, std::string* name_out, std::string* src_out) {
    std::string text = tmpl;
    size_t p0 = text.find("GDV_OPTFLAT_VALUE");
    text.replace(p0, strlen("GDV_OPTFLAT_VALUE"), optflat);
    uint64_t h = Fnv1a(HashableSource(text) + LibraryTag(text));
    char name[64];
    snprintf(name, sizeof(name), "gdv_k_%016llx", static_cast<unsigned long long>(h));
    size_t pos = text.find("GDV_KERNEL_NAME");
    text.replace(pos, strlen("GDV_KERNEL_NAME"), name);
    *name_out = name;
    *src_out = text;
  };
  const std::string tmpl = s.str();
  finish(tmpl, plan->has_flat_output ? "1" : "0", &plan->kernel_name, &plan->source);
  if (plan->has_flat_output) finish(tmpl, "0", &plan->kernel_name_general, &plan->source_general);
  plan->ir = plan->source;
  return Status::OK();
}

// ------------------------------------------------------------------ string plans, wave shape (round 3)
// Var-len plans whose output lengths are a function of the input OFFSETS (and of fixed-width
// inputs) once the bytes are assumed ASCII — substr / left / right / upper / lower / concat /
// castVARCHAR over columns and literals: C5 — need no hand-off inside the kernel at all:
//   pre-pass  (kPrepass) the same row bodies reduced to their lengths, views built from the offsets
//             only: one byte total per wave tile and output -> `counts`
//   scan      ScanReduce / Spine / Apply over the wave-tile totals (gdv_kernels.hip) -> `mask`
//   main      (kMain) every WAVE is an independent tile: its output base is one scalar load; no
//             scanner workgroup, no look-back, no workgroup barrier, no LDS shared between waves;
//             flat outputs leave straight from the byte sweep's registers
// Measured on the hand-written prototype (tools/proto/k4h_proto.hip, profiles/r03_k4_experiments.txt):
// 1.82 ms (scanner shape) -> 0.90-0.97 ms on C5.  The ASCII assumption is checked by the sweep: a
// tile that breaks it raises NOTASCII and the host re-runs the batch on the scanner-shaped kernel.
enum class WaveKind { kMain, kPrepass };

Status AssembleStringsWave(CodeGen& cg, KernelPlan* plan, const std::vector<std::string>& expr_strings,
                           const WordAccumulators& accs, const std::string& decls_before_loop,
                           const std::string& decls_in_pass, const std::string& after_row_loop,
                           const std::string& epilogue_after_loop, WaveKind kind, bool has_direct_pass) {
  plan->input_fields = cg.input_fields_;
  plan->input_needs_values = cg.needs_values_;
  plan->input_needs_validity = cg.needs_validity_;
  plan->can_raise = cg.can_raise_;
  plan->string_skeleton = true;
  plan->wave_tiles = true;
  for (size_t k = 0; k < plan->input_fields.size(); k++)
    plan->has_varlen_input |= cg.schema_[plan->input_fields[k]].type.is_varlen() && cg.needs_values_[k];
  plan->layout.n_in = static_cast<int>(plan->input_fields.size());
  plan->layout.n_out = static_cast<int>(plan->output_types.size());
  const bool prepass = kind == WaveKind::kPrepass;
  cg.sel_ascii_check_ = cg.selection() && !prepass;
  const int nv = static_cast<int>(cg.varlen_outs_.size());
  int nstage = 0;
  for (auto& vo : cg.varlen_outs_) nstage = std::max(nstage, vo.window + 1);
  // (a pre-pass has hooks only for a swept replace(); the exact variant adds one continuation-byte bitmap
  // per input whose ASCII flag is consulted)
  const int ncb = cg.exact_ascii_ ? static_cast<int>(cg.ascii_slots_.size()) : 0;
  const int nhook = static_cast<int>(cg.contains_hooks_.size()) + ncb;
  plan->num_varlen_outputs = prepass ? 0 : nv;
  for (auto& vo : cg.varlen_outs_) plan->has_flat_output |= vo.flat_slot >= 0;
  const int nin = plan->layout.n_in;
  const int mirror_slot = cg.mirror_slot_;  // (decided with the tile shape, PlanProjectorShape)
  if (prepass && mirror_slot < 0 && !cg.exact_ascii_ && !plan->opts.prepass_rolled && !cg.selection())
    // optimistic pre-pass without a sweep: the body is a few integer operations per row (every general
    // UTF-8 path folds away under the compile-time ASCII flag) — unrolled, the eight sub-tiles' offsets
    // are consumed from their registers without the rotation of the rolled loop
    cg.unroll_rows_ = true;

  Assembler as{cg, plan, {}};
  // round 6: the main kernel's per-sub-tile sweep (the one with the LDS mirror) takes GDV_SG sub-tiles' spans at a time
  if (!prepass && mirror_slot >= 0 && !cg.selection() && plan->opts.sweep_group > 1 &&
      plan->opts.subtiles % plan->opts.sweep_group == 0 && plan->opts.subtiles > plan->opts.sweep_group)
    as.sweep_group_ = plan->opts.sweep_group;
  as.Header(expr_strings);
  std::ostringstream& s = as.src;
  s << "// " << (prepass ? (cg.exact_ascii_ ? "pre-pass: byte totals per wave tile (exact variant: ASCII flags from a sweep of the bytes)"
                                            : "pre-pass: byte totals per wave tile from the offsets alone (optimistic ASCII)")
                         : (cg.exact_ascii_ ? "wave shape, exact variant: ASCII flags per (sub-)tile from the byte sweep"
                                            : "wave shape: independent wave tiles, output bases from the pre-pass + scan"))
    << (cg.selection() ? " (rows = the slots of a selection vector: gathered, no byte sweep)" : "")
    << "\n#define GDV_NV " << nv << "\n#define GDV_NSTAGE " << std::max(nstage, 1)
    << "\n#define GDV_NHOOK " << (prepass && mirror_slot < 0 && ncb > 0 && plan->opts.prepass_ahead ? "(" + std::to_string(nhook) + " * GDV_U)" : std::to_string(std::max(nhook, 1))) << "\n"
    << (mirror_slot >= 0 || (prepass && ncb > 0) ? "#define GDV_HIT_WORDS (GDV_SUB_SPAN / 64 + 4)  // match bits of ONE sub-tile's span\n"
                                                 : "#define GDV_HIT_WORDS (GDV_SPAN_MAX / 64 + 4)\n")
    << "constexpr bool FULL = false;  // string tiles test `live` at run time (one code path)\n"
    << AblDefine()
    << "#define GDV_OUT(e, v) if (live) " << (plan->opts.nontemporal ? "gdv_stnt" : "gdv_st") << "(out##e, row, (v))\n";
  if (prepass) {
    // (no staged copies in a pre-pass)
  } else if (mirror_slot >= 0) {
    // staged copies read the row's bytes from the LDS mirror of the sub-tile's span when the view
    // lies inside it (any view of that column does; literals, other columns: HBM as before); a
    // replace() value answered by the sweep is copied from there along its marked positions
    const std::string M = std::to_string(mirror_slot);
    const std::string where = "mir" + M + ", sd" + M + " + sb" + M + ", hm_ok" + M + " ? se" + M + " - sb" + M + " : 0";
    if (cg.replace_hook_ >= 0)
      s << "#define GDV_STAGE_COPY(dst, v) gdv_stage_copy_mirh(dst, v, " << where << ", hit" << cg.replace_hook_ << ")\n";
    else
      s << "#define GDV_STAGE_COPY(dst, v) gdv_stage_copy_mir(dst, v, " << where << ")\n";
  } else {
    s << "#define GDV_STAGE_COPY(dst, v) gdv_stage_copy(dst, v)\n";
  }

  s << "GDV_DEV void gdv_tile(const gdv_args& A, const gdv_int64 wt, const int lane, const int wave,\n"
    << "                      gdv_uint8* lds_out, gdv_uint64* lds_hit, gdv_uint8* lds_in) {\n"
    << "  (void)lds_out; (void)lds_hit; (void)lds_in; (void)wave;\n"
    << "  gdv_ctx ctx{A.err};\n"
    << "  (void)ctx;\n"
    << "  const gdv_uint8* const gdv_cst = (const gdv_uint8*)A.aux0;  // the plan's constant block\n"
    << "  (void)gdv_cst;\n"
    << "  const gdv_int64 n = GDV_ROWS(A);\n"
    << "  const gdv_int64 wbase = wt * GDV_U;\n"
    << "  const gdv_int64 rbase = wbase * 64;\n"
    << (cg.selection() && !prepass ? [&] {
         // an EMPTY selection whose count sits in device memory still launches: offsets[0] = 0 is then nobody's row
         std::string z;
         for (size_t e = 0; e < plan->output_types.size(); e++)
           if (plan->output_types[e].is_varlen())
             z += "  if (n <= 0 && wt == 0 && lane == 0) A.out[" + std::to_string(e) + "].offsets[0] = 0;\n";
         return z;
       }() : std::string())
    << "  if (rbase >= n) return;  // (no barrier anywhere below: waves are independent)\n"
    << "  const bool last_tile = rbase + 64 * GDV_U >= n;  // the wave tile that holds the batch's last row\n"
    << "  const gdv_int64 seg_stride = A.aux1;  // wave-tile totals / bases: one array of seg_stride entries per scanned output\n"
    << "  (void)last_tile; (void)seg_stride;\n";
  EmitStringPointersAndLoads(s, cg, plan, !prepass, /*wave_shape=*/true);
  WaveSweepText sweep;
  if (prepass && mirror_slot < 0) {
    // views carry the flags the main kernel will give them — the optimistic ASCII flag where a
    // function consults it — so both kernels compute the same lengths.  Outputs whose length is a
    // function of the offsets (substr, left, concat ...) read no byte here; others (replace, rtrim,
    // an if over like ...) read the rows' bytes a first time.
    for (int k = 0; k < nin; k++) {
      const DataType& t = cg.schema_[plan->input_fields[k]].type;
      if (!(t.is_varlen() && cg.needs_values_[k])) continue;
      if (cg.selection()) {
        // gathered rows.  Functions that consult the ASCII flag get it OPTIMISTICALLY — their lengths then follow from
        // the offsets and this pre-pass reads no byte; the main kernel checks every row it copies
        s << "  const gdv_int32 sfl" << k << " = " << (cg.ascii_slots_.count(k) ? "GDV_STR_ASCII" : "0") << ";\n";
        continue;
      }
      s << "  const gdv_int32 sp1" << k << " = so" << k << "[last_tile ? n : rbase + 64 * GDV_U];\n";
      if (cg.exact_ascii_ && cg.ascii_slots_.count(k)) {
        // exact variant: the lengths depend on the bytes now — the pre-pass sweeps every sub-tile's span
        // (at the top of the row loop) for the pieces that hold a byte >= 0x80; rows take a per-row flag
        const std::string K = std::to_string(k);
        cg.row_ascii_slots_.insert(k);
        s << "  const gdv_int32 inb" << K << " = sd" << K << " + sp1" << K << " + 8 <= slim" << K << " ? GDV_STR_INBUF : 0;\n"
          << "  const gdv_int32 sfl" << K << " = inb" << K << ";\n";
        if (plan->opts.prepass_ahead) {
          // round 5: every sub-tile's span is swept HERE, before the row loop — the first 1024-byte piece of all GDV_U spans
          // is requested back to back (GDV_U loads in flight per lane where the pipelined form below has one; the sweep is
          // a dozen instructions, unrolling IT is cheap — unrolling the row body was not), each span's continuation
          // bytes go to a bitmap of its own, bit u of hiw = sub-tile u's span holds a byte >= 0x80
          const std::string CB = std::to_string(cg.CbIndex(k));
          s << "  gdv_uint32 hiw" << K << " = 0;\n"
            << "  {\n"
            << "    gdv_int32 sx[GDV_U + 1];  // the sub-tiles' first bytes (wave-uniform); sx[GDV_U] = the tile's end\n"
            << "#pragma unroll\n"
            << "    for (int u = 0; u < GDV_U; u++) sx[u] = __builtin_amdgcn_readfirstlane(oa" << K << "[u]);\n"
            << "    sx[GDV_U] = sp1" << K << ";\n"
            << "    gdv_uint64 pw[GDV_U][2];\n"
            << "#pragma unroll\n"
            << "    for (int u = 0; u < GDV_U; u++) {\n"
            << "      const gdv_int32 a = sx[u] - (gdv_int32)((gdv_uint64)(sd" << K << " + sx[u]) & 15) + 16 * lane;\n"
            << "      pw[u][0] = 0ull; pw[u][1] = 0ull;\n"
            << "      if (a < sx[u + 1]) __builtin_memcpy(pw[u], __builtin_assume_aligned(sd" << K << " + a, 16), 16);\n"
            << "    }\n"
            << "#pragma unroll\n"
            << "    for (int u = 0; u < GDV_U; u++) {\n"
            << "      const gdv_int32 sb = sx[u] - (gdv_int32)((gdv_uint64)(sd" << K << " + sx[u]) & 15), se = sx[u + 1];\n"
            << "      const bool fits = se - sb <= GDV_SUB_SPAN;\n"
            << "      gdv_uint64* const cb = lds_hit + (" << CB << " * GDV_U + u) * GDV_HIT_WORDS;\n"
            << "      gdv_uint64 sacc = 0, w0 = pw[u][0], w1 = pw[u][1];\n"
            << "      for (gdv_int32 c = sb; c < se; c += 1024) {\n"
            << "        const gdv_int32 a = c + 16 * lane;\n"
            << "        if (c != sb) {  // a span longer than one step (rows of more than 16 bytes on average): loaded as it comes\n"
            << "          gdv_uint64 t[2] = {0ull, 0ull};\n"
            << "          if (a < se) __builtin_memcpy(t, __builtin_assume_aligned(sd" << K << " + a, 16), 16);\n"
            << "          w0 = t[0]; w1 = t[1];\n"
            << "        }\n"
            << "        sacc |= w0 | w1;\n"
            << "        const gdv_uint64 hbw = __ballot(((w0 | w1) & GDV_B80) != 0);\n"
            << "        const gdv_uint32 cm = hbw != 0 ? gdv_cont_mask16(w0, w1) : 0u;\n"
            << "        if (fits && a < se) ((gdv_uint16*)cb)[(a - sb) >> 4] = (gdv_uint16)cm;\n"
            << "      }\n"
            << "      if (__ballot((sacc & GDV_B80) != 0) != 0) hiw" << K << " |= 1u << u;\n"
            << "    }\n"
            << "  }\n"
            << "  __builtin_amdgcn_wave_barrier();\n";
          std::ostringstream b;
          b << "    // exact pre-pass: this sub-tile's continuation-byte bitmap (filled before the loop)\n"
            << "    const gdv_int32 ss" << K << " = __builtin_amdgcn_readfirstlane(oa" << K << "[0]);\n"
            << "    const gdv_int32 se" << K << " = u + 1 < GDV_U ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > 1 ? 1 : 0]) : sp1" << K << ";\n"
            << "    const gdv_int32 sb" << K << " = ss" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + ss" << K << ") & 15);\n"
            << "    const bool hm_ok" << K << " = se" << K << " - sb" << K << " <= GDV_SUB_SPAN;  // wave-uniform: the span fits the LDS bitmap\n"
            << "    const gdv_uint64* const cb" << K << " = lds_hit + (" << CB << " * GDV_U + u) * GDV_HIT_WORDS;\n"
            << "    const bool hi8_" << K << " = ((hiw" << K << " >> u) & 1u) != 0;\n";
          sweep.per_sub += b.str();
          continue;
        }
        // software-pipelined like the main kernel's sweep: the first 1024-byte step of the NEXT sub-tile's
        // span is requested before this sub-tile's rows are looked at (unrolling the loop to have all
        // eight in flight measured slower: 0.81 vs 0.52 ms at 10^8 rows — the general UTF-8 paths are
        // inlined into every copy of the body)
        s << "  gdv_uint64 wn" << K << "[2] = {0ull, 0ull};\n"
          << "  {\n"
          << "    const gdv_int32 ss = __builtin_amdgcn_readfirstlane(oa" << K << "[0]);\n"
          << "    const gdv_int32 se = GDV_U > 1 ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > 1 ? 1 : 0]) : sp1" << K << ";\n"
          << "    const gdv_int32 a = ss - (gdv_int32)((gdv_uint64)(sd" << K << " + ss) & 15) + 16 * lane;\n"
          << "    if (a < se) __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K << " + a, 16), 16);\n"
          << "  }\n";
        std::ostringstream b;
        b << "    // exact pre-pass: the continuation bytes of this sub-tile's span of input " << k << " -> LDS bitmap\n"
          << "    const gdv_int32 ss" << K << " = __builtin_amdgcn_readfirstlane(oa" << K << "[0]);\n"
          << "    const gdv_int32 se" << K << " = u + 1 < GDV_U ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > 1 ? 1 : 0]) : sp1" << K << ";\n"
          << "    const gdv_int32 sb" << K << " = ss" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + ss" << K << ") & 15);\n"
          << "    const bool hm_ok" << K << " = se" << K << " - sb" << K << " <= GDV_SUB_SPAN;  // wave-uniform: the span fits the LDS bitmap\n"
          << "    gdv_uint64* const cb" << K << " = lds_hit + " << cg.CbIndex(k) << " * GDV_HIT_WORDS;\n"
          << "    gdv_uint64 sacc" << K << " = 0;\n"
          << "    for (gdv_int32 c = sb" << K << "; c < se" << K << "; c += 1024) {\n"
          << "      const gdv_int32 a = c + 16 * lane;\n"
          << "      const gdv_uint64 w[2] = {wn" << K << "[0], wn" << K << "[1]};\n"
          << "      wn" << K << "[0] = 0ull; wn" << K << "[1] = 0ull;\n"
          << "      if (a + 1024 < se" << K << ") __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K << " + a + 1024, 16), 16);\n"
          << "      sacc" << K << " |= w[0] | w[1];\n"
          << "      const gdv_uint64 hbw = __ballot(((w[0] | w[1]) & GDV_B80) != 0);\n"
          << "      const gdv_uint32 cm = hbw != 0 ? gdv_cont_mask16(w[0], w[1]) : 0u;\n"
          << "      if (hm_ok" << K << " && a < se" << K << ") ((gdv_uint16*)cb" << K << ")[(a - sb" << K << ") >> 4] = (gdv_uint16)cm;\n"
          << "    }\n"
          << "    if (u + 1 < GDV_U) {  // the first piece of the next sub-tile's span\n"
          << "      const gdv_int32 e2 = u + 2 < GDV_U ? __builtin_amdgcn_readfirstlane(oa" << K << "[GDV_U > 2 ? 2 : 0]) : sp1" << K << ";\n"
          << "      const gdv_int32 nb = se" << K << " - (gdv_int32)((gdv_uint64)(sd" << K << " + se" << K << ") & 15);\n"
          << "      if (nb + 16 * lane < e2) __builtin_memcpy(wn" << K << ", __builtin_assume_aligned(sd" << K << " + nb + 16 * lane, 16), 16);\n"
          << "    }\n"
          << "    const bool hi8_" << K << " = __ballot((sacc" << K << " & GDV_B80) != 0) != 0;\n"
          << "    __builtin_amdgcn_wave_barrier();\n";
        sweep.per_sub += b.str();
      } else
        s << "  const gdv_int32 sfl" << k << " = (sd" << k << " + sp1" << k << " + 8 <= slim" << k << " ? GDV_STR_INBUF : 0)"
          << (cg.ascii_slots_.count(k) ? " | GDV_STR_ASCII" : "") << ";\n";
    }
  } else if (mirror_slot >= 0) {
    EmitWaveSweep(cg, plan, mirror_slot, prepass, &sweep, as.sweep_group_);
    s << sweep.prologue;
  } else {
    EmitWaveTileSweep(s, cg, plan, &sweep.epilogue);
  }

  s << "  // ---- rows: fused expression bodies (value for every row, validity per word)\n";
  for (auto& a : accs.names) s << "  gdv_uint64 " << a << " = 0;\n";
  s << decls_before_loop;
  if (has_direct_pass)
    s << "  bool need_direct = false;\n"
      << "  // pass 0: offsets + bytes staged in LDS.  pass 1 (rare): outputs whose bytes do not fit the\n"
      << "  // LDS window are recomputed and copied straight to HBM.\n"
      << "  for (int pass = 0; pass < 2; pass++) {\n"
      << "  if (pass == 1 && !need_direct) break;\n";
  else
    s << "  constexpr int pass = 0;\n  (void)pass;\n";
  s << decls_in_pass;
  if (!prepass)
    // every load issued so far (offsets, validity words, the tile's base, the first piece) is waited
    // for HERE, once: left to the compiler, the wait lands at the value's first use inside the loop
    // as a vmcnt(0) that every later iteration pays again — stalling on the previous sub-tile's
    // stores and on the piece it has just prefetched
    s << "  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)\n";
  EmitStringRowLoop(s, cg, plan, /*wave_shape=*/true, sweep.per_sub);
  s << "  }\n";
  if (has_direct_pass) s << "  if (pass == 1) break;\n";
  s << sweep.epilogue;
  s << after_row_loop;
  if (!prepass && !has_direct_pass)
    // the epilogue's pointers (validity words, closing offsets, totals) are read from the argument
    // block HERE, not hoisted above the row loop where they would sit in — or be spilled from —
    // scalar registers for the whole tile: the block's address goes through an opaque zero
    s << "  {\n  gdv_int64 gdv_z = 0;\n  asm volatile(\"\" : \"+s\"(gdv_z));\n"
      << "  const gdv_args& A_late = *(const gdv_args*)((const gdv_uint8*)&A + gdv_z);\n"
      << "  {\n  const gdv_args& A = A_late;\n"
      << epilogue_after_loop << "  }\n  }\n";
  else
    s << epilogue_after_loop;
  if (has_direct_pass) s << "  }  // pass\n";
  s << "}\n\n";

  s << "#ifndef GDV_STRING_KERNEL_ATTR\n#define GDV_STRING_KERNEL_ATTR\n#endif\n"
    << "extern \"C\" __global__ void GDV_STRING_KERNEL_ATTR __launch_bounds__(GDV_WAVES * 64) GDV_KERNEL_NAME(const gdv_args A) {\n"
    << "  const int lane = threadIdx.x & 63;\n"
    << "  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n";
  if (prepass)
    // (a pre-pass tile is a few loads and one store: waves walk several tiles, grid-stride)
    s << (mirror_slot >= 0 || ncb > 0 ? "  __shared__ __attribute__((aligned(16))) gdv_uint64 gdv_lds_hit[GDV_WAVES][GDV_NHOOK * GDV_HIT_WORDS];\n" : "")
      << "  const gdv_int64 nwt = (GDV_ROWS(A) + 64 * GDV_U - 1) / (64 * GDV_U);\n"
      << "  for (gdv_int64 wt = (gdv_int64)blockIdx.x * GDV_WAVES + wave; wt < nwt; wt += (gdv_int64)gridDim.x * GDV_WAVES)\n"
      << "    gdv_tile(A, wt, lane, wave, nullptr, " << (mirror_slot >= 0 || ncb > 0 ? "gdv_lds_hit[wave]" : "nullptr") << ", nullptr);\n";
  else
    s << "  __shared__ __attribute__((aligned(16))) gdv_uint8 gdv_lds_out[GDV_WAVES][GDV_NSTAGE * (GDV_OUT_WIN + 16)];\n"
      << "  __shared__ __attribute__((aligned(16))) gdv_uint64 gdv_lds_hit[GDV_WAVES][GDV_NHOOK * GDV_HIT_WORDS];\n"
      << (mirror_slot >= 0 ? "  __shared__ __attribute__((aligned(16))) gdv_uint8 gdv_lds_in[GDV_WAVES][GDV_SUB_SPAN + 32];\n" : "")
      << "  gdv_tile(A, (gdv_int64)blockIdx.x * GDV_WAVES + wave, lane, wave, gdv_lds_out[wave], gdv_lds_hit[wave], "
      << (mirror_slot >= 0 ? "gdv_lds_in[wave]" : "nullptr") << ");\n";
  s << "}\n";

  std::string text = s.str();
  uint64_t h = Fnv1a(HashableSource(text) + LibraryTag(text));
  char name[64];
  snprintf(name, sizeof(name), "gdv_k_%016llx", static_cast<unsigned long long>(h));
  plan->kernel_name = name;
  size_t pos = text.find("GDV_KERNEL_NAME");
  text.replace(pos, strlen("GDV_KERNEL_NAME"), plan->kernel_name);
  plan->source = text;
  plan->ir = text;
  return Status::OK();
}

}  // namespace

namespace {

enum class StringShape { kScanner, kWaveMain, kWavePrepass, kWaveMainExact, kWavePrepassExact };

// Is the value of `n` — and its validity — computable without reading a single var-len BYTE once
// every string is assumed ASCII?  (offsets, validity bitmaps, fixed-width values and literals are
// free.)  Outputs with that property let a plan take the wave shape: their byte totals come from a
// pre-pass over the offsets.
bool ByteFree(const Node& n) {
  static const std::set<std::string> over_strings = {
      "substr", "substring", "left", "right", "upper", "lower", "octet_length", "bit_length", "char_length",
      "length", "lengthUtf8", "castVARCHAR", "concat", "concatOperator", "reverse", "initcap", "lpad", "rpad", "isnull", "hashSHA256", "sha256",
      "hashSHA1", "sha1", "sha", "hashMD5", "md5",
      "isnotnull"};
  switch (n.kind()) {
    case NodeKind::kField:
    case NodeKind::kLiteral:
      return true;
    case NodeKind::kFunction: {
      auto& fn = static_cast<const FunctionNode&>(n);
      bool takes_strings = false;
      for (auto& c : fn.children()) {
        if (!ByteFree(*c)) return false;
        takes_strings |= c->return_type().is_varlen();
      }
      return !takes_strings || over_strings.count(fn.name()) != 0;
    }
    case NodeKind::kIf: {
      auto& i = static_cast<const IfNode&>(n);
      return ByteFree(*i.condition()) && ByteFree(*i.then_node()) && ByteFree(*i.else_node());
    }
    case NodeKind::kBoolean:
      for (auto& c : static_cast<const BooleanNode&>(n).children())
        if (!ByteFree(*c)) return false;
      return true;
    case NodeKind::kIn: {
      auto& in = static_cast<const InNode&>(n);
      return !in.value_type().is_varlen() && ByteFree(*in.eval());
    }
  }
  return false;
}

// One shape of a projector's kernel.  kWavePrepass generates only the expressions listed in
// `only` (the scanned var-len outputs of the main kernel, in that order = segment order).
Status PlanProjectorShape(const Schema& schema, const std::vector<ExpressionPtr>& exprs, SelectionMode mode,
                          const CodegenOptions& opts, StringShape shape, const std::vector<int>* only,
                          KernelPlan* plan, std::vector<VarlenOut>* varlen_outs, int compact_from) {
  plan->kind = KernelKind::kProject;
  plan->mode = mode;
  plan->opts = opts;
  plan->compact_from = compact_from;
  CodeGen cg(schema, mode, opts);
  cg.compact_from_ = compact_from;
  cg.exact_ascii_ = shape == StringShape::kWaveMainExact || shape == StringShape::kWavePrepassExact;
  if (shape == StringShape::kWaveMainExact) shape = StringShape::kWaveMain;
  if (shape == StringShape::kWavePrepassExact) shape = StringShape::kWavePrepass;
  cg.no_hooks_ = shape == StringShape::kWavePrepass;
  cg.bake_needles_ = shape != StringShape::kScanner && !opts.runtime_needles;
  cg.replace_hits_ = shape != StringShape::kScanner && opts.lds_mirror && mode == SelectionMode::kNone;
  WordAccumulators accs;
  std::ostringstream after_loop, before_loop, in_pass, after_rows;
  std::vector<std::string> strings;
  int num_staged = 0, num_scanned = 0;
  const bool wave = shape != StringShape::kScanner;
  const bool prepass = shape == StringShape::kWavePrepass;
  if (!prepass)
    for (auto& e : exprs) plan->has_varlen_output |= e->result().type.is_varlen();
  // 64 lengths below 2^25 cannot wrap the 32-bit scan; otherwise the total is taken on 16-bit
  // halves and saturates (the host rejects outputs of 2 GiB or more)
  auto tile_total = [](const std::string& ln, const std::string& inc) {
    return "(__builtin_expect(__ballot(" + ln + " >= (1 << 25)) != 0, 0) ? gdv_tile_total(" + ln +
           ") : (gdv_uint32)gdv_wave_last(" + inc + "))";
  };
  for (size_t e = 0; e < exprs.size(); e++) {
    if (only != nullptr && std::find(only->begin(), only->end(), static_cast<int>(e)) == only->end()) continue;
    Val v;
    cg.Stmt("// @expr_" + std::to_string(e));
    GDV_RETURN_NOT_OK(cg.Gen(*exprs[e]->root(), "", &v));
    const DataType& t = exprs[e]->result().type;
    if (!prepass) plan->output_types.push_back(t);
    strings.push_back(exprs[e]->ToString());
    const std::string E = std::to_string(e);
    if (t.is_varlen()) {
      // The row's bytes are one view, or the pieces of a concat written back to back; lengths are
      // prefix-summed inside the wave on the DPP data path.
      const std::string ok = cg.Tmp("bool", CodeGen::AndFull("live", cg.LaneValid(v)));
      VarlenOut vo;
      vo.e = static_cast<int>(e);
      const bool flat_cand = v.pieces.empty() && v.col_slot >= 0 && !cg.selection();
      const bool has_window = !flat_cand && num_staged < 3;  // LDS staging windows per wave
      std::vector<std::pair<std::string, std::string>> pieces = v.pieces;
      if (pieces.empty()) pieces.emplace_back(v.v, "");
      std::string total;
      std::vector<std::string> pv;
      auto emit_pieces = [&] {
        for (size_t q = 0; q < pieces.size(); q++) {
          const std::string name = "pv" + E + "_" + std::to_string(q);
          pv.push_back(name);
          cg.Stmt("gdv_str " + name + " = " + pieces[q].first + ";");
          cg.Stmt("if (!(" + CodeGen::AndFull(ok, pieces[q].second) + ")) " + name + ".len = 0;");
          total += (q ? " + " : "") + name + ".len";
        }
      };
      if (prepass) {
        // lengths only: one byte total per wave tile -> counts[segment][wave tile]
        emit_pieces();
        const std::string S = std::to_string(num_scanned++);
        // (lengths add up per lane across the sub-tiles; ONE wave reduction per tile.  Exact whenever
        // the tile stays below 2^31 bytes — the main kernel's running total then agrees — and at
        // least 2^31-1 otherwise, which the host rejects)
        in_pass << "  gdv_int32 run" << E << " = 0;  // this lane's bytes over the sub-tiles (saturates at 2^31-1)\n";
        cg.Stmt("run" + E + " = gdv_sat_add31(run" + E + ", " + total + ");");
        after_rows << "  { const gdv_uint32 t = gdv_tile_total(run" << E << ");\n"
                   << "    if (lane == 0) A.counts[" << S << " * seg_stride + wt] = t > 0x7fffffffu ? 0x7fffffffu : t; }\n";
        cg.varlen_outs_.push_back(vo);
        continue;
      }
      if (wave && flat_cand) {
        // the output IS the (mapped) input span: offsets = the input's, rebased; bytes leave from the
        // byte sweep.  Only wrong if a NULL row carries bytes: NOTFLAT, the host re-runs (general kernel).
        vo.flat_slot = v.col_slot;
        vo.flat_map = v.col_map;
        const std::string K = std::to_string(v.col_slot);
        const int vidx = static_cast<int>(cg.varlen_outs_.size());
        before_loop << "  gdv_uint64 fb" << E << " = 0;  // rows that drop bytes of the input span (nulls with a length)\n";
        cg.Stmt("if (live) outo" + E + "[row] = oa" + K + "[u] - so0_" + K + ";");
        cg.Stmt("fb" + E + " |= __ballot(!" + ok + " && ob" + K + "[u] > oa" + K + "[u]);");
        after_rows << "  if (fb" << E << " != 0 && lane == 0) gdv_raise_bits(A.err, GDV_ERR_NOTFLAT);\n"
                   << "  if (last_tile && lane == 0) {  // closing offset + byte total of a flat output\n"
                   << "    outo" << E << "[n] = sp1" << K << " - so0_" << K << ";\n"
                   << "    ((gdv_uint64*)A.counts)[" << vidx << "] = (gdv_uint64)(sp1" << K << " - so0_" << K << ");\n"
                   << "  }\n";
        cg.varlen_outs_.push_back(vo);
      } else if (wave) {
        // scanned output: the wave tile's base comes from the pre-pass + scan (one scalar load), so
        // bytes can leave as soon as they are staged.  The LDS window STREAMS: when the next
        // sub-tile's rows would not fit behind what is staged, the staged bytes are flushed
        // (coalesced, to their final place) and the window starts over at that sub-tile — outputs
        // of up to GDV_OUT_WIN / 64 bytes per row on average (32 at 4 sub-tiles, 64 at 8) never
        // touch the per-row direct copy, whatever the tile's total is (rounds 1-2 and the scanner
        // shape: a tile above 8 bytes per row takes a second row pass with scattered stores — what
        // made upper(concat(s, '-', s)) cost 4.1 ms at 5 * 10^7 rows).  A single sub-tile wider than
        // the whole window is copied row by row, in place.
        emit_pieces();
        const std::string S = std::to_string(num_scanned++);
        vo.segment = num_scanned - 1;
        before_loop << "  const gdv_int64 base" << E << " = (gdv_int64)A.mask[" << S << " * seg_stride + wt];\n"
                    << "  gdv_int32 run" << E << " = 0;   // bytes this wave tile has produced so far (saturates at 2^31-1)\n"
                    << "  gdv_int32 wb" << E << " = 0;    // tile-relative position of the window's first byte\n";
        cg.Stmt("const gdv_int32 ln" + E + "_u = " + total + ";");
        cg.Stmt("const gdv_int32 inc" + E + " = gdv_wave_scan_incl(ln" + E + "_u);");
        cg.Stmt("const gdv_int32 loc" + E + " = run" + E + " + inc" + E + " - ln" + E + "_u;  // where the row's bytes start inside the wave tile");
        cg.Stmt("const gdv_int32 sub0_" + E + " = run" + E + ";");
        cg.Stmt("{ const gdv_uint32 t = " + tile_total("ln" + E + "_u", "inc" + E) + ";");
        cg.Stmt("  run" + E + " = gdv_sat_add31(run" + E + ", (gdv_int32)(t > 0x7fffffffu ? 0x7fffffffu : t)); }");
        cg.Stmt("if (live) outo" + E + "[row] = (gdv_int32)(base" + E + " + loc" + E + ");");
        if (cg.selection())
          // the slot count may live in device memory (GDV_ROWS reads it): the host cannot know where the closing
          // offset belongs, so the wave tile that holds the last slot writes it
          after_rows << "  if (last_tile && lane == 0) outo" << E << "[n] = (gdv_int32)(base" << E << " + run" << E << ");\n";
        // nothing at or past the caller's capacity is written (the grand total says what was needed)
        cg.Stmt("const bool fit" + E + " = run" + E + " < 0x7fffffff && base" + E + " + run" + E + " <= A.out[" + E + "].cap;");
        if (has_window) {
          vo.window = num_staged++;
          vo.reads_views = !v.opaque;  // (concat results: their pieces are mostly views of columns)
          before_loop << "  gdv_uint8* const win" << E << " = lds_out + " << vo.window << " * (GDV_OUT_WIN + 16);\n";
          cg.Stmt("if (run" + E + " - wb" + E + " > GDV_OUT_WIN) {  // wave-uniform: the window is full");
          cg.Stmt("  if (fit" + E + " && sub0_" + E + " > wb" + E + ") gdv_flush_out(outd" + E + " + base" + E + " + wb" + E + ", win" + E + ", sub0_" + E +
                  " - wb" + E + ", lane);");
          cg.Stmt("  wb" + E + " = sub0_" + E + ";");
          cg.Stmt("}");
          cg.Stmt("if (run" + E + " - wb" + E + " <= GDV_OUT_WIN) {");
          cg.Stmt("  gdv_int32 at = loc" + E + " - wb" + E + ";");
          for (auto& name : pv) {
            cg.Stmt("  if (" + AblNot(4) + name + ".len > 0) GDV_STAGE_COPY((gdv_lds_u8*)(win" + E + " + at), " + name + ");");
            cg.Stmt("  at += " + name + ".len;");
          }
          cg.Stmt("} else if (fit" + E + ") {  // this sub-tile alone is wider than the window: row by row, in place");
        } else {
          cg.Stmt("if (fit" + E + ") {  // (no LDS window left for this output: row by row, in place)");
        }
        cg.Stmt("  gdv_uint8* at = outd" + E + " + base" + E + " + loc" + E + ";");
        for (auto& name : pv) {
          cg.Stmt("  if (" + name + ".len > 0) gdv_str_copy(at, " + name + ");");
          cg.Stmt("  at += " + name + ".len;");
        }
        if (has_window) {
          cg.Stmt("  wb" + E + " = run" + E + ";  // nothing of it is staged");
          cg.Stmt("}");
          after_rows << "  if (run" << E << " > wb" << E << " && run" << E << " < 0x7fffffff && base" << E << " + run" << E << " <= A.out[" << E
                     << "].cap" << AblAnd(16) << ")\n"
                     << "    gdv_flush_out(outd" << E << " + base" << E << " + wb" << E << ", win" << E << ", run" << E << " - wb" << E << ", lane);\n";
        } else {
          cg.Stmt("}");
        }
        cg.varlen_outs_.push_back(vo);
      } else {
        emit_pieces();
        before_loop << "  gdv_int32 lc" << E << "[GDV_U] = {};  // where each row's bytes start inside the wave tile\n"
                    << "  gdv_int32 run" << E << " = 0;  // bytes this wave tile produces (saturates at 2^31-1)\n"
                    << "  bool dir" << E << " = false;  // second row pass: copy straight to HBM at dbase" << E << "\n"
                    << "  gdv_int64 dbase" << E << " = 0;\n";
        if (flat_cand) {
          vo.flat_slot = v.col_slot;
          vo.flat_map = v.col_map;
          const std::string K = std::to_string(v.col_slot);
          before_loop << "  gdv_uint64 fb" << E << " = 0;  // rows that drop bytes of the input span (nulls with a length)\n";
          // optimistic flat variant of the kernel (compile-time): the bytes are copied after the
          // sweep and the offsets are the input's, rebased: nothing of this output is scanned
          cg.Stmt("if (optflat) {");
          cg.Stmt("  if (pass == 0) {");
          cg.Stmt("    if (live) outo" + E + "[row] = oa" + K + "[u] - so0_" + K + ";");
          cg.Stmt("    fb" + E + " |= __ballot(!" + ok + " && ob" + K + "[u] > oa" + K + "[u]);");
          cg.Stmt("  }");
          cg.Stmt("} else {");
        }
        cg.Stmt("const gdv_int32 ln" + E + "_u = " + total + ";");
        cg.Stmt("if (pass == 0) {");
        cg.Stmt("  const gdv_int32 inc = gdv_wave_scan_incl(ln" + E + "_u);");
        cg.Stmt("  lc" + E + "[0] = run" + E + " + inc - ln" + E + "_u;");
        // 64 lengths below 2^25 cannot wrap the 32-bit scan; otherwise the total is taken on
        // 16-bit halves and saturates (the host rejects outputs of 2 GiB or more)
        cg.Stmt("  const gdv_uint32 t = __builtin_expect(__ballot(ln" + E + "_u >= (1 << 25)) != 0, 0) ? gdv_tile_total(ln" + E +
                "_u) : (gdv_uint32)gdv_wave_last(inc);");
        cg.Stmt("  run" + E + " = gdv_sat_add31(run" + E + ", (gdv_int32)(t > 0x7fffffffu ? 0x7fffffffu : t));");
        if (flat_cand) {
          const std::string K = std::to_string(v.col_slot);
          cg.Stmt("  fb" + E + " |= __ballot(!" + ok + " && ob" + K + "[u] > oa" + K + "[u]);");
        }
        if (has_window) {
          vo.window = num_staged++;
          before_loop << "  gdv_uint8* const win" << E << " = lds_out + " << vo.window << " * (GDV_OUT_WIN + 16);\n";
          // stage while the view is at hand (rows that fall outside the window are skipped: the
          // tile then takes the second, direct pass)
          cg.Stmt("  gdv_int32 at = lc" + E + "[0];");
          for (auto& name : pv) {
            cg.Stmt("  if (" + AblNot(4) + name + ".len > 0 && at + " + name + ".len <= GDV_OUT_WIN) GDV_STAGE_COPY((gdv_lds_u8*)(win" + E +
                    " + at), " + name + ");");
            cg.Stmt("  at += " + name + ".len;");
          }
        }
        cg.Stmt("} else if (dir" + E + ") {");
        cg.Stmt("  gdv_uint8* at = outd" + E + " + dbase" + E + " + lc" + E + "[0];");
        for (auto& name : pv) {
          cg.Stmt("  if (" + name + ".len > 0) gdv_str_copy(at, " + name + ");");
          cg.Stmt("  at += " + name + ".len;");
        }
        cg.Stmt("}");
        if (flat_cand) cg.Stmt("}");
        cg.varlen_outs_.push_back(vo);
      }
    } else if (t.id == kBool) {
      std::string acc = accs.Get(cg, "__ballot(" + CodeGen::AndExpr("live", v.v) + ")");
      after_loop << WordStore(acc, "((gdv_uint64*)A.out[" + E + "].data)");
    } else {
      cg.Stmt("GDV_OUT(" + E + ", (" + t.CType() + ")" + v.v + ");");
    }
    if (prepass) continue;
    // validity word of the 64 rows of this sub-tile
    std::string word;
    if (cg.selection()) {
      word = "__ballot(" + CodeGen::AndExpr("live", cg.LaneValid(v)) + ")";
    } else {
      word = "(" + cg.WordExpr(v.vcols) + " & livemask)";
      if (!v.vlane.empty()) word = "(" + word + " & __ballot(live && " + v.vlane + "))";
    }
    after_loop << WordStore(accs.Get(cg, word), "A.out[" + E + "].valid");
  }
  // Loads in flight: aim for >= 8 KiB of input values per wave tile (64 lanes x GDV_U rows x
  // input bytes/row), within a budget of 512 input bytes per lane.  Wide plans (C2: 32 B/row,
  // ten outputs) stay at 4 — measured optimum, more sub-tiles cost occupancy — narrow plans
  // (C1: 12 B/row) go to 16 (+3 % measured).
  bool string_plan = plan->has_varlen_output || wave;
  for (size_t k = 0; k < cg.input_fields_.size(); k++)
    string_plan |= schema[cg.input_fields_[k]].type.is_varlen() && cg.needs_values_[k];
  if (string_plan) {
    // workgroup tile = 4 waves x 4 sub-tiles x 64 rows.  Sub-tiles cost registers, not code (the
    // row loop is rolled): 8 were better while the kernel carried 130 VGPRs either way; with the
    // branch-free range test and the compile-time flat variant 4 sub-tiles fit 95 VGPRs (5 waves
    // per SIMD) and win: 1.70 vs 1.83 ms (profiles/r02_c5_tuning.txt)
    if (!plan->opts.subtiles_forced) {
      if (shape == StringShape::kScanner) plan->opts.subtiles = 4;
      if (shape == StringShape::kWaveMain) {
        // wave shape: a tile costs a fixed prologue (scalar loads, sweep set-up, ends of the span,
        // flush), so 8 sub-tiles per wave beat 4 (C5: 1.15 vs 1.29 ms, profiles/r03_c5_tuning.txt)
        // — as long as the wave's LDS (staging windows of 8 B per row, the match bitmaps and the
        // mirror of ONE sub-tile's span) leaves room for six workgroups per CU
        const int windows = num_staged, hooks = static_cast<int>(cg.contains_hooks_.size());
        // LDS mirror (and with it the per-sub-tile sweep): the first swept var-len input, when some
        // staged output's copies would read it
        cg.mirror_slot_ = -1;
        bool readers = cg.replace_hook_ >= 0 && !cg.selection();  // (rows of a swept replace() are copied from the mirror)
        for (auto& vo : cg.varlen_outs_) readers |= vo.window >= 0 && vo.reads_views;
        for (size_t k = 0; opts.lds_mirror && readers && !cg.selection() && k < cg.input_fields_.size() && cg.mirror_slot_ < 0; k++) {
          if (!(schema[cg.input_fields_[k]].type.is_varlen() && cg.needs_values_[k])) continue;
          bool swept = cg.ascii_slots_.count(static_cast<int>(k)) != 0;
          for (auto& h : cg.contains_hooks_) swept |= h.slot == static_cast<int>(k);
          for (auto& vo : cg.varlen_outs_) swept |= vo.flat_slot == static_cast<int>(k);
          if (swept) cg.mirror_slot_ = static_cast<int>(k);
        }
        // (the mirror must hold the column whose matches the bitmap marks)
        if (opts.lds_mirror && cg.replace_hook_ >= 0) cg.mirror_slot_ = cg.contains_hooks_[cg.replace_hook_].slot;
        const int lds_u8 = windows * (8 * 64 * 8 + 16) +
                           (cg.mirror_slot_ >= 0 ? hooks * (2048 / 64 + 4) * 8 + 2048 + 32 : hooks * ((8 * 64 * 32) / 64 + 4) * 8);
        plan->opts.subtiles = lds_u8 <= 6656 ? 8 : 4;
      }
      // (kWavePrepass: the caller passes the main kernel's tile)
      if (shape == StringShape::kWavePrepass)  // a pre-pass sweeps only for a replace() that counts its matches in the bitmap
        cg.mirror_slot_ = cg.replace_hook_ >= 0 ? cg.contains_hooks_[cg.replace_hook_].slot : -1;
    }
    if (!plan->opts.waves_forced) plan->opts.waves = 4;
    if (varlen_outs != nullptr) *varlen_outs = cg.varlen_outs_;
    if (wave)
      return AssembleStringsWave(cg, plan, strings, accs, before_loop.str(), in_pass.str(), after_rows.str(),
                                 after_loop.str(), prepass ? WaveKind::kPrepass : WaveKind::kMain,
                                 /*has_direct_pass=*/false);
    return AssembleStrings(cg, plan, strings, accs, before_loop.str(), after_loop.str());
  }
  if (wave) return Status::CodeGenError("internal: wave shape asked for a plan without var-len columns");
  if (!plan->opts.subtiles_forced) {
    int in_bytes = 0;
    bool any_varlen = false;
    for (size_t k = 0; k < cg.input_fields_.size(); k++) {
      const DataType& t = schema[cg.input_fields_[k]].type;
      any_varlen |= t.is_varlen();
      if (cg.needs_values_[k]) in_bytes += std::max(1, t.byte_width());
    }
    if (!any_varlen && in_bytes > 0) {
      int u = 4;
      while (u < 16 && 64 * u * in_bytes < 8192 && 2 * u * in_bytes <= 512) u <<= 1;
      // Round 6: SIXTEEN sub-tiles per wave wherever the loaded values fit the registers (in_bytes x 16 <= 512: 128 VGPRs) and
      // no element is wider than 8 bytes.  A wave then reads and writes 64 x 16 consecutive elements of every stream — 4-8 KiB
      // per stream and wave instead of 1-2 — and the DRAM sees longer runs per stream between the 14 interleaved ones:
      // C2 4.75-4.82 ms against 4.96-5.10 (U = 4) on one box, 4.78-4.90 against 4.96-5.14 on another; C1 0.733-0.735 against
      // 0.776-0.780 (U = 4) and 0.81-0.85 (U = 8, what the rule above chose for it); decimal128 columns (C4): no difference,
      // they stay at 4 (profiles/r06_tile_shape.txt).  Occupancy drops to two or three waves per SIMD: it does not matter here.
      int max_width = 0;
      for (size_t k = 0; k < cg.input_fields_.size(); k++)
        if (cg.needs_values_[k]) max_width = std::max(max_width, schema[cg.input_fields_[k]].type.byte_width());
      for (auto& e : exprs) max_width = std::max(max_width, e->result().type.byte_width());
      if (in_bytes * 16 <= 512 && max_width <= 8) u = 16;
      plan->opts.subtiles = u;
      // ... and ONE workgroup per CU for such a plan when it reads at least twice what it writes (row mode): with sixteen
      // sub-tiles a wave has 16 x in_bytes x 64 bytes in flight, four waves keep a CU's share of the HBM busy, and every further
      // resident workgroup only interleaves more read streams at the DRAM — C1 0.705-0.711 ms against 0.743-0.757 at eight per
      // CU (and 0.745-0.754 at two) in three alternations on one box, 0.673-0.709 against 0.706-0.738 in four on another; write-dominated plans (C2) and 4-sub-tile plans (C4)
      // measure the same at 1 / 2 / 8 and keep the default (profiles/r06_grid_density.txt)
      int out_bytes = 0;
      for (auto& e : exprs) out_bytes += std::max(1, e->result().type.byte_width());
      if (u == 16 && mode == SelectionMode::kNone && in_bytes >= 2 * out_bytes) plan->grid_blocks_per_cu = 1;
    }
  }
  return Assemble(cg, plan, strings, accs, before_loop.str(), after_loop.str());
}


}  // namespace

Status PlanProjector(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                     SelectionMode mode, const CodegenOptions& opts, KernelPlan* plan, int compact_from) {
  AblationScope ablation_scope(opts.ablation);
  if (exprs.empty()) return Status::Invalid("Expressions cannot be empty");
  for (auto& e : exprs) {
    if (!e) return Status::Invalid("Expression cannot be null");
    GDV_RETURN_NOT_OK(ValidateExpression(schema, *e));
  }
  // Wave shape (pre-pass + scan + independent wave tiles): row mode and at least one var-len
  // output.  Outputs whose length follows from the offsets under the ASCII assumption (ByteFree)
  // cost a pre-pass over the offsets only; the others make the pre-pass read the bytes a first
  // time — still faster than the scanner shape, whose hand-off, occupancy and second row pass for
  // outputs above 8 bytes per row cost more (replace at 5 * 10^7 rows: 3.3 ms there).  Selection-
  // mode plans — and the re-run of a batch that breaks an assumption — take the scanner shape.
  // Round 5: selection-mode plans take the wave shape too (rows = slots, gathered through the selection vector;
  // no byte sweep, no optimistic assumption: the row functions run their general paths in the pre-pass and in the
  // main kernel alike) — rounds 2-4 sent them to the scanner shape.  GDV_NO_SEL_WAVE=1: as before.
  bool wave_ok = (mode == SelectionMode::kNone || !opts.no_sel_wave) && !opts.no_wave_shape;
  bool any_varlen_out = false;
  const bool bytefree_only = opts.wave_bytefree_only;
  for (auto& e : exprs) {
    if (!e->result().type.is_varlen()) continue;
    any_varlen_out = true;
    if (bytefree_only) wave_ok = wave_ok && ByteFree(*e->root());
  }
  if (!(wave_ok && any_varlen_out))
    return PlanProjectorShape(schema, exprs, mode, opts, StringShape::kScanner, nullptr, plan, nullptr, compact_from);

  KernelPlan fast, slow;
  std::vector<VarlenOut> vouts;
  GDV_RETURN_NOT_OK(PlanProjectorShape(schema, exprs, mode, opts, StringShape::kWaveMain, nullptr, &fast, &vouts, compact_from));
  GDV_RETURN_NOT_OK(PlanProjectorShape(schema, exprs, mode, opts, StringShape::kScanner, nullptr, &slow, nullptr, compact_from));
  // one argument block serves both kernels: the two generations must have bound the same inputs,
  // literals and constants (they run the same tree walk; checked, not assumed)
  if (fast.input_fields != slow.input_fields || fast.input_needs_values != slow.input_needs_values ||
      fast.input_needs_validity != slow.input_needs_validity || fast.literals != slow.literals ||
      fast.const_block != slow.const_block) {
    *plan = slow;
    return Status::OK();
  }
  std::vector<int> scanned;
  for (auto& vo : vouts)
    if (vo.flat_slot < 0) scanned.push_back(vo.e);
  if (static_cast<int>(scanned.size()) > kMaxWaveSegments) {
    *plan = slow;
    return Status::OK();
  }
  *plan = fast;
  plan->can_raise = fast.can_raise || slow.can_raise;
  // the fallback: the scanner-shaped kernel with every output on the general path
  plan->kernel_name_general = slow.has_flat_output ? slow.kernel_name_general : slow.kernel_name;
  plan->source_general = slow.has_flat_output ? slow.source_general : slow.source;
  plan->general_subtiles = slow.opts.subtiles;
  plan->general_waves = slow.opts.waves;
  plan->wave_segments.clear();
  for (auto& vo : vouts) plan->wave_segments.push_back(vo.flat_slot < 0 ? vo.segment : -1);
  if (!scanned.empty()) {
    auto pre = std::make_shared<KernelPlan>();
    CodegenOptions popts = fast.opts;  // same tile: a wave tile of the pre-pass IS a wave tile of the main kernel
    GDV_RETURN_NOT_OK(PlanProjectorShape(schema, exprs, mode, popts, StringShape::kWavePrepass, &scanned, pre.get(), nullptr, compact_from));
    if (pre->opts.subtiles != fast.opts.subtiles) {
      *plan = slow;
      return Status::OK();
    }
    plan->prepass = pre;
    plan->ir = fast.source + pre->source;  // (the main kernel first: its name is the plan's)
  }
  // The exact variant (round 4): only where some function consults the ASCII flag — i.e. where the
  // optimistic kernels can raise NOTASCII at all.  Same tree walk, same inputs / literals / constants
  // (checked): the host hands it the optimistic kernels' argument blocks.
  // (selection-mode plans: a batch that breaks the assumption takes the scanner-shaped general kernel)
  if (mode == SelectionMode::kNone && fast.source.find("GDV_ERR_NOTASCII") != std::string::npos) {
    auto ex = std::make_shared<KernelPlan>();
    std::vector<VarlenOut> evouts;
    Status st = PlanProjectorShape(schema, exprs, mode, fast.opts, StringShape::kWaveMainExact, nullptr, ex.get(), &evouts, compact_from);
    bool ok = st.ok() && ex->opts.subtiles == fast.opts.subtiles && ex->opts.waves == fast.opts.waves &&
              ex->input_fields == fast.input_fields && ex->input_needs_values == fast.input_needs_values &&
              ex->input_needs_validity == fast.input_needs_validity && ex->literals == fast.literals &&
              ex->const_block == fast.const_block && evouts.size() == vouts.size();
    for (size_t i = 0; ok && i < vouts.size(); i++)
      ok = evouts[i].flat_slot == vouts[i].flat_slot && evouts[i].segment == vouts[i].segment;
    if (ok && plan->prepass) {
      auto epre = std::make_shared<KernelPlan>();
      st = PlanProjectorShape(schema, exprs, mode, fast.opts, StringShape::kWavePrepassExact, &scanned, epre.get(), nullptr, compact_from);
      const KernelPlan& p0 = *plan->prepass;
      ok = st.ok() && epre->opts.subtiles == fast.opts.subtiles && epre->input_fields == p0.input_fields &&
           epre->input_needs_values == p0.input_needs_values && epre->input_needs_validity == p0.input_needs_validity &&
           epre->literals == p0.literals && epre->const_block == p0.const_block &&
           epre->layout.total() == p0.layout.total();
      ex->prepass = epre;
    }
    if (ok) {
      plan->exact = ex;
      plan->can_raise = plan->can_raise || ex->can_raise;
    }
  }
  return Status::OK();
}

Status PlanFilter(const Schema& schema, const ExpressionPtr& condition,
                  const CodegenOptions& opts, KernelPlan* plan) {
  AblationScope ablation_scope(opts.ablation);
  if (!condition) return Status::Invalid("Condition cannot be null");
  GDV_RETURN_NOT_OK(ValidateExpression(schema, *condition));
  if (condition->root()->return_type().id != kBool)
    return Status::ValidationError("Filter condition must be of type boolean");
  plan->kind = KernelKind::kFilter;
  plan->mode = SelectionMode::kNone;
  plan->opts = opts;
  CodeGen cg(schema, SelectionMode::kNone, opts);
  WordAccumulators accs;
  Val v;
  cg.Stmt("// @expr_0 (filter condition)");
  GDV_RETURN_NOT_OK(cg.Gen(*condition->root(), "", &v));
  // a null predicate does not select the row
  std::string pass = CodeGen::AndExpr(cg.LaneValid(v), v.v);
  cg.Stmt("const gdv_uint64 fm = __ballot(" + CodeGen::AndExpr("live", pass) + ");");
  cg.Stmt("fcount += (gdv_uint32)__popcll(fm);");
  std::string acc = accs.Get(cg, "fm");
  // A predicate kernel keeps nothing but its input values live, so it can afford many more
  // loads in flight per wave than a projection: measured on C3 (2 x int64, 10^9 rows) the
  // predicate pass goes from 5.2 TB/s at GDV_U = 4 to 5.9 TB/s at 16 (profiles/r01_c3_sweep).
  // Budget: <= 512 bytes of input values per lane, i.e. <= 128 VGPRs of loads.
  if (!plan->opts.subtiles_forced) {
    int in_bytes = 0;
    for (size_t k = 0; k < cg.input_fields_.size(); k++)
      if (cg.needs_values_[k]) in_bytes += std::max(4, schema[cg.input_fields_[k]].type.byte_width());
    int u = 16;
    while (u > 4 && u * std::max(in_bytes, 1) > 512) u >>= 1;
    plan->opts.subtiles = u;
    // (Round 6: ONE workgroup per CU — what the read skeleton prefers — measured 2.64-2.70 ms per Filter::Evaluate against
    // 2.69-2.80 at eight per CU in five alternations on one box and 2.97-2.98 against 2.92-2.94 in four on another: the grid
    // stays at the engine's default, profiles/r06_grid_density.txt)
  }
  std::ostringstream after;
  // the match words are written once and read by the index-emission kernel much later: non-temporal
  // (C3, same box: 2.74 / 2.99 ms plain vs 2.69 / 2.67 ms; validity words of projections measured
  // the other way round — C2 4.90 vs 5.0-5.1 ms — and stay plain)
  after << WordStore(acc, "A.mask", true);
  // one selected-row count per wave tile feeds the offsets scan (gdv_kernels.hip)
  after << "  if (lane == 0) A.counts[wbase / GDV_U] = fcount;\n";
  bool string_plan = false;
  for (size_t k = 0; k < cg.input_fields_.size(); k++)
    string_plan |= schema[cg.input_fields_[k]].type.is_varlen() && cg.needs_values_[k];
  if (string_plan) {
    // the index-emission kernel walks groups of 64 match words: sub-tiles stay a power of two
    plan->opts.subtiles = 4;
    if (!plan->opts.waves_forced) plan->opts.waves = 4;
    return AssembleStrings(cg, plan, {condition->ToString()}, accs, "  gdv_uint32 fcount = 0;\n", after.str());
  }
  return Assemble(cg, plan, {condition->ToString()}, accs, "  gdv_uint32 fcount = 0;\n",
                  after.str());
}

// ------------------------------------------------------------------ fused filter -> project (K2F)
//
// Filter::Evaluate followed by a selection-mode Projector::Evaluate reads most lines of the
// predicate's columns twice (at a selectivity of 1/8 nearly every 128-byte line holds a selected
// row: 2.86 + 2.94 ms at 10^9 rows, profiles/r03_filter_project_chain.txt).  This plan does both in
// ONE pass over the batch:
//   1  all loads of the wave tile (predicate AND projection columns), no control flow in between
//   2  the predicate: one 64-bit match word per sub-tile, the tile's selected-row count
//   3  output base of the workgroup tile: the waves' counts meet in LDS, ONE wave looks back over
//      the earlier workgroup tiles (gdv_fp_lookback)
//   4  the projections, evaluated from the registers phase 1 filled; functions that can raise run on
//      selected rows only; the values of selected rows are stored at  base + rank  (compacted in
//      the output: a wave's stores cover one contiguous run of elements), their validity / bool
//      bits are compacted per sub-tile (gdv_compact_word) and leave as whole bitmap words; the
//      row indices (the SelectionVector) are written the same way when the plan asks for them.
// The result equals Filter::Evaluate + Projector::Evaluate(batch, selection_vector) bit for bit.
// Fixed-width (and bool) outputs over fixed-width columns; anything else -> CodeGenError and the
// callers chain the two operators as before.
//
// Round 5 — the WINDOWED shape (the default; the round-4 shape above stays as its `exact` variant for
// batches that select most of their rows).  Round 4 held every loaded value in registers across the
// look-back (111 VGPRs: two 8-wave workgroups per CU) and stored the selected rows from there — 8 of 64
// lanes per store instruction at a selectivity of 1/8.  Now predicate and projections run in ONE loop
// BEFORE the look-back and the selected rows' values (and row indices) are written, at their rank inside
// the wave tile, to a wave-private LDS window of GDV_FP_CAP rows; validity / bool bits are accumulated at
// wave-local bit positions.  Nothing wide is live across the barrier + look-back; afterwards the window
// leaves with full-width stores at pos0 + i and the local bitmap words are shifted into place
// (gdv_bits_flush_local).  A wave tile that selects more than GDV_FP_CAP rows re-reads the sub-tiles that
// did not fit (rolled loop, from L2 / Infinity Cache) and stores those rows directly; the engine moves a
// FilterProject whose batches select more than that to the direct kernel.
namespace {
enum class FpShape { kDirect, kWindow };

Status PlanFilterProjectShape(const Schema& schema, const ExpressionPtr& condition, const std::vector<ExpressionPtr>& exprs,
                              SelectionMode index_mode, const CodegenOptions& opts, FpShape shape, KernelPlan* plan) {
  AblationScope ablation_scope(opts.ablation);
  const bool win = shape == FpShape::kWindow;
  plan->kind = KernelKind::kFilterProject;
  plan->mode = index_mode;  // width of the emitted row indices; kNone: no SelectionVector output
  plan->opts = opts;
  CodeGen cg(schema, SelectionMode::kNone, opts);
  const std::string sel_t = SelCType(index_mode);
  const bool with_index = index_mode != SelectionMode::kNone;
  // ---- the predicate (row mode; a null predicate does not select the row)
  Val c;
  cg.Stmt("// @expr_0 (filter condition)");
  GDV_RETURN_NOT_OK(cg.Gen(*condition->root(), "", &c));
  const std::string pass = CodeGen::AndExpr(cg.LaneValid(c), c.v);
  cg.Stmt("const gdv_uint64 fmw = __ballot(" + CodeGen::AndExpr("live", pass) + ");");
  if (win) cg.Stmt("const gdv_uint32 run = fcount;  // selected rows of this wave tile before this sub-tile");
  cg.Stmt("fcount += (gdv_uint32)__popcll(fmw);");
  if (!win) cg.Stmt("fm = gdv_deposit_word(fm, u, fmw, lane);  // lane u keeps sub-tile u's match word (2 VGPRs, not 2 x GDV_U SGPRs)");
  const std::string cond_body = cg.body_.str();
  // ---- the projections.  Direct shape: a loop of its own after the look-back — the predicate's temporaries
  // are out of scope, common sub-expressions are shared among the projections only.  Windowed shape: the same
  // loop iteration as the predicate; and once more, predicate included, for wave tiles whose selected rows did not
  // all fit the window (they are read again: nothing of the tile is kept in registers across the look-back).
  cg.body_.str("");
  if (!win) cg.cse_.clear();
  std::vector<std::string> strings{condition->ToString()};
  std::map<std::string, int> bitmap_of;   // compacted-word expression -> accumulator index
  std::vector<std::string> flushes;       // after the row loop: one per output bitmap
  const std::string first_bit = win ? "0" : "pos0";
  const std::string at_bit = win ? "(gdv_int64)run" : "pos0 + run";
  const std::string flush_fn = win ? "gdv_bits_flush_local" : "gdv_bits_flush";
  auto bits_acc = [&](const std::string& word_expr) {
    auto it = bitmap_of.find(word_expr);
    if (it != bitmap_of.end()) return it->second;
    const int k = static_cast<int>(bitmap_of.size());
    bitmap_of[word_expr] = k;
    cg.Stmt("bacc" + std::to_string(k) + " = gdv_bits_append(bacc" + std::to_string(k) + ", " + first_bit + ", " + at_bit +
            ", gdv_compact_word(" + word_expr + ", fmu, below, cnt, lane), cnt, lane);");
    return k;
  };
  std::string proj_body, tail_body;
  int window_bytes_per_row = index_mode == SelectionMode::kUInt16 ? 2 : index_mode == SelectionMode::kUInt32 ? 4
                             : index_mode == SelectionMode::kUInt64 ? 8 : 0;
  for (int pass_no = 0; pass_no < (win ? 2 : 1); pass_no++) {
    const bool tail = pass_no == 1;
    if (tail) {
      cg.body_.str("");
      cg.cse_.clear();
      Val c2;
      cg.Stmt("// @expr_0 (filter condition), again");
      GDV_RETURN_NOT_OK(cg.Gen(*condition->root(), "", &c2));
      cg.Stmt("const gdv_uint64 fmu = __ballot(" + CodeGen::AndExpr("live", CodeGen::AndExpr(cg.LaneValid(c2), c2.v)) + ");");
      cg.Stmt("const int cnt = (int)__popcll(fmu);");
      cg.Stmt("if (run + (gdv_uint32)cnt > (gdv_uint32)GDV_FP_CAP) {  // wave-uniform: a selected row of this sub-tile is beyond the window");
      cg.Stmt("const bool fsel = (fmu >> lane) & 1;");
      cg.Stmt("const int slot = (int)run + gdv_rank_below(fmu);");
      cg.Stmt("const bool ftail = fsel && slot >= GDV_FP_CAP;");
      cg.Stmt("const gdv_int64 opos = pos0 + slot;");
      cg.Stmt("(void)opos; (void)ftail;");
      if (with_index) cg.Stmt("if (ftail) selv[opos] = (" + sel_t + ")row;");
    }
    for (size_t e = 0; e < exprs.size(); e++) {
      Val v;
      cg.Stmt("// @expr_" + std::to_string(e + 1));
      GDV_RETURN_NOT_OK(cg.Gen(*exprs[e]->root(), tail ? "ftail" : "fsel", &v));
      if (!v.pieces.empty() || v.opaque) return Status::CodeGenError("fused filter-project: materialised values take the chain");
      const DataType& t = exprs[e]->result().type;
      const std::string E = std::to_string(e);
      if (!tail) {
        plan->output_types.push_back(t);
        strings.push_back(exprs[e]->ToString());
      }
      if (t.id == kBool) {
        if (!tail) {
          const int k = bits_acc("__ballot(" + CodeGen::AndExpr("live", v.v) + ")");
          flushes.push_back("  " + flush_fn + "((gdv_uint64*)A.out[" + E + "].data, bacc" + std::to_string(k) + ", pos0, (gdv_int64)fcount, lane);\n");
        }
      } else if (tail) {
        cg.Stmt("if (ftail) out" + E + "[opos] = (" + t.CType() + ")" + v.v + ";");
      } else if (win) {
        cg.Stmt("if (fwin) win" + E + "[slot] = (" + t.CType() + ")" + v.v + ";");
        window_bytes_per_row += t.byte_width();
      } else {
        cg.Stmt("if (fsel) out" + E + "[opos] = (" + t.CType() + ")" + v.v + ";");
      }
      if (tail) continue;
      std::string word = "(" + cg.WordExpr(v.vcols) + " & livemask)";
      if (!v.vlane.empty()) word = "(" + word + " & __ballot(live && " + v.vlane + "))";
      const int k = bits_acc(word);
      flushes.push_back("  " + flush_fn + "(A.out[" + E + "].valid, bacc" + std::to_string(k) + ", pos0, (gdv_int64)fcount, lane);\n");
    }
    if (tail) {
      cg.Stmt("}");
      cg.Stmt("run += (gdv_uint32)cnt;");
    }
    (tail ? tail_body : proj_body) = cg.body_.str();
  }
  for (size_t k = 0; k < cg.input_fields_.size(); k++)
    if (schema[cg.input_fields_[k]].type.is_varlen())
      return Status::CodeGenError("fused filter-project: var-len columns take the filter + projector chain");

  plan->input_fields = cg.input_fields_;
  plan->input_needs_values = cg.needs_values_;
  plan->input_needs_validity = cg.needs_validity_;
  plan->can_raise = true;  // (the look-back's stall bit travels in the error word)
  plan->exprs_raise = cg.can_raise_;
  plan->layout.n_in = static_cast<int>(plan->input_fields.size());
  plan->layout.n_out = static_cast<int>(plan->output_types.size());
  const int nin = plan->layout.n_in;
  // loads in flight: as a predicate kernel, within 512 bytes of input values per lane
  if (!plan->opts.subtiles_forced) {
    int in_bytes = 0;
    for (int k = 0; k < nin; k++)
      if (cg.needs_values_[k]) in_bytes += std::max(4, schema[cg.input_fields_[k]].type.byte_width());
    int u = 16;
    while (u > 2 && u * std::max(in_bytes, 1) > 384) u >>= 1;
    plan->opts.subtiles = u;
  }
  // 8 waves per workgroup: half as many look-back participants as 4 (measured at 10^9 rows, C3 shape:
  // 16 x 4: 4.45 / 4.10 ms with / without the selection vector, 16 x 8: 4.26 / 4.05, 16 x 16: 4.89 / 4.19,
  // 8 x 8: 4.58 / 4.40 — profiles/r04_filter_project.txt)
  if (!plan->opts.waves_forced) plan->opts.waves = 8;
  // Windowed shape: a wave tile is GDV_FP_K ROUNDS of GDV_U sub-tiles (contiguous rows) — the look-back, which costs
  // 0.8 ms of the 4.3 at 10^9 rows whatever the shape of the stores (profiles/r05_filter_project_no_lookback.txt), is paid
  // once per K x 8192 rows.  The bitmap accumulators hold one local word per lane: K x U <= 63.
  // The window: GDV_FP_CAP rows per wave tile, every windowed output + the index at its own width, at most half
  // the wave tile's rows (78 KB per 8-wave workgroup at the default: two workgroups per CU, which is what the
  // kernel's registers allow anyway).
  int rounds = 1, cap = 0;
  if (win) {
    if (window_bytes_per_row == 0) return Status::CodeGenError("fused filter-project: nothing to window (bitmap outputs only)");
    rounds = std::max(1, std::min(opts.fp_rounds, 63 / plan->opts.subtiles));
    const int rows_wave = 64 * plan->opts.subtiles * rounds;
    cap = std::min(rows_wave / 2, opts.fp_window_bytes / window_bytes_per_row) / 64 * 64;
    if (cap < 128) return Status::CodeGenError("fused filter-project: rows too wide for the LDS window");
  }

  Assembler as{cg, plan, {}};
  as.Header(strings);
  std::ostringstream& s = as.src;
  s << "#define GDV_FP_K " << rounds << "  // rounds of GDV_U sub-tiles per wave tile\n";
  if (win) s << "#define GDV_FP_CAP " << cap << "  // rows of a wave tile's LDS window\n";
  s << "template <bool FULL>\n"
    << "GDV_DEV void gdv_fused_tile(const gdv_args& A, const gdv_int64 tile, const int lane, const int wave,\n"
    << "                            gdv_uint32* wg_cnt, gdv_uint64* wg_excl";
  if (win) {
    for (size_t e = 0; e < plan->output_types.size(); e++)
      if (plan->output_types[e].id != kBool) s << ", " << plan->output_types[e].CType() << "* win" << e;
    if (with_index) s << ", " << sel_t << "* widx";
  }
  s << ") {\n"
    << "  gdv_ctx ctx{A.err};\n"
    << "  (void)ctx;\n"
    << "  const gdv_uint8* const gdv_cst = (const gdv_uint8*)A.aux0;  // the plan's constant block\n"
    << "  (void)gdv_cst;\n"
    << "  const gdv_int64 n = GDV_ROWS(A);\n"
    << "  const gdv_int64 wfirst = (tile * GDV_WAVES + wave) * (GDV_FP_K * GDV_U);  // this wave's first 64-row word\n";
  for (int k = 0; k < nin; k++) {
    const DataType& t = schema[plan->input_fields[k]].type;
    if (t.id != kBool && cg.needs_values_[k])
      s << "  const " << t.CType() << "* __restrict__ in" << k << " = (const " << t.CType() << "*)A.in[" << k << "].data;\n";
  }
  for (size_t e = 0; e < plan->output_types.size(); e++) {
    const DataType& t = plan->output_types[e];
    if (t.id != kBool)
      s << "  " << t.CType() << "* __restrict__ out" << e << " = (" << t.CType() << "*)A.out[" << e << "].data;\n";
  }
  if (with_index) s << "  " << sel_t << "* __restrict__ selv = (" << sel_t << "*)A.sel;\n";
  const std::string ld = plan->opts.nt_loads ? "gdv_ldnt" : "gdv_ld";
  // phase 1 of one round: all loads of GDV_U sub-tiles, issued back to back
  auto emit_loads = [&] {
    s << "  // ---- phase 1: all loads of this round's GDV_U sub-tiles, issued back to back\n";
    for (int k = 0; k < nin; k++) {
      const DataType& t = schema[plan->input_fields[k]].type;
      if (t.id == kBool) {
        if (cg.needs_values_[k]) s << "  const gdv_uint64 dw" << k << " = gdv_bitmap_tile(A.in[" << k << "].bits, wbase, lane, GDV_U);\n";
      } else if (cg.needs_values_[k]) {
        s << "  " << t.CType() << " c" << k << "[GDV_U];\n";
      }
      if (cg.needs_validity_[k]) s << "  const gdv_uint64 vw" << k << " = gdv_bitmap_tile(A.in[" << k << "].valid, wbase, lane, GDV_U);\n";
    }
    s << "#pragma unroll\n  for (int u = 0; u < GDV_U; u++) {\n"
      << "    const gdv_int64 row = rbase + u * 64 + lane;\n"
      << "    const bool live = FULL || row < n;\n"
      << "    (void)live;\n";
    for (int k = 0; k < nin; k++) {
      const DataType& t = schema[plan->input_fields[k]].type;
      if (t.id != kBool && cg.needs_values_[k])
        s << "    c" << k << "[u] = live ? " << ld << "(in" << k << ", row) : (" << t.CType() << ")0;\n";
    }
    s << "  }\n";
  };
  auto row_prologue = [&](const std::string& words_suffix) {
    s << "      const gdv_int64 row = rbase + u * 64 + lane;\n"
      << "      const bool live = FULL || row < n;\n"
      << "      const gdv_uint64 livemask = FULL ? ~0ull : __ballot(live);\n"
      << "      (void)livemask; (void)row; (void)live;\n";
    for (int k = 0; k < nin; k++) {
      const DataType& t = schema[plan->input_fields[k]].type;
      if (t.id == kBool && cg.needs_values_[k]) s << "      const gdv_uint64 d" << k << " = gdv_tile_word(dw" << words_suffix << k << ", u);\n";
      if (cg.needs_validity_[k]) s << "      const gdv_uint64 v" << k << " = gdv_tile_word(vw" << words_suffix << k << ", u);\n";
    }
  };
  s << "  gdv_uint32 fcount = 0;\n";
  if (win) {
    for (size_t k = 0; k < bitmap_of.size(); k++) s << "  gdv_uint64 bacc" << k << " = 0;  // output bitmap words at wave-local bit positions\n";
    s << (rounds > 1 ? "#pragma unroll 1\n" : "")
      << "  for (int kb = 0; kb < GDV_FP_K; kb++) {\n"
      << "  const gdv_int64 wbase = wfirst + kb * GDV_U;\n"
      << "  const gdv_int64 rbase = wbase * 64;\n";
  } else {
    s << "  gdv_uint64 fm = 0;\n"
      << "  const gdv_int64 wbase = wfirst;\n"
      << "  const gdv_int64 rbase = wbase * 64;\n";
  }
  emit_loads();
  s << "  // ---- phase 2: the predicate -> one match word per sub-tile"
    << (win ? "; the projections of the selected rows -> the wave's LDS window, at their rank in the wave tile\n" : "\n")
    << "#pragma unroll\n  for (int u = 0; u < GDV_U; u++) {\n    {\n";
  row_prologue("");
  if (!win) s << "      const bool fsel = true;  // (the predicate itself runs on every live row)\n      (void)fsel;\n";
  s << cond_body;
  if (win) {
    s << "      const gdv_uint64 fmu = fmw;\n"
      << "      const int cnt = (int)__popcll(fmu);\n"
      << "      const bool fsel = (fmu >> lane) & 1;\n"
      << "      const int below = gdv_rank_below(fmu);\n"
      << "      const int slot = (int)run + below;  // this lane's rank among the wave tile's selected rows\n"
      << "      const bool fwin = fsel && slot < GDV_FP_CAP;\n"
      << "      (void)cnt; (void)below; (void)slot; (void)fwin;\n";
    if (with_index) s << "      if (fwin) widx[slot] = (" << sel_t << ")row;\n";
    s << proj_body;
  }
  s << "    }\n  }\n";
  if (win) s << "  }  // round\n";
  s << "  // ---- phase 3: output base of this wave: the workgroup's waves meet in LDS, wave 0 looks back\n"
    << "  if (lane == 0) wg_cnt[wave] = fcount;\n"
    << "  __syncthreads();\n"
    << "  gdv_uint32 before = 0, wg_total = 0;\n"
    << "#pragma unroll\n  for (int w = 0; w < GDV_WAVES; w++) {\n"
    << "    const gdv_uint32 cw = wg_cnt[w];\n"
    << "    wg_total += cw;\n"
    << "    if (w < wave) before += cw;\n"
    << "  }\n"
    << "  if (wave == 0) {\n"
    << (opts.fp_experiment == 1
            // EXPERIMENT (GDV_FP_EXPERIMENT=1, never the product: the outputs land at the tile's own first row): what the
            // kernel costs WITHOUT the look-back — every workgroup tile takes tile x rows-per-tile as its base
            ? "    const gdv_uint64 e = (gdv_uint64)tile * (GDV_WAVES * GDV_FP_K * GDV_U * 64); (void)wg_total;\n"
            : "    const gdv_uint64 e = gdv_fp_lookback(A.mask, tile, wg_total, lane, A.err);\n")
    << "    if (lane == 0) {\n"
    << "      *wg_excl = e;\n"
    << "      if (tile == (gdv_int64)gridDim.x - 1) *(gdv_int64*)A.counts = (gdv_int64)(e + wg_total);  // the batch's selected-row count\n"
    << "    }\n"
    << "  }\n"
    << "  __syncthreads();\n"
    << "  const gdv_int64 pos0 = (gdv_int64)*wg_excl + before;  // output position of this wave's first selected row\n";
  if (win) {
    const std::string st = plan->opts.nontemporal ? "gdv_stnt" : "gdv_st";
    s << "  // ---- phase 4: the window leaves with full-width stores (lane i -> output position pos0 + i)\n"
      << "  const int in_window = (int)(fcount < (gdv_uint32)GDV_FP_CAP ? fcount : (gdv_uint32)GDV_FP_CAP);\n"
      << "  for (int i = lane; i < in_window; i += 64) {\n";
    for (size_t e = 0; e < plan->output_types.size(); e++)
      if (plan->output_types[e].id != kBool) s << "    " << st << "(out" << e << ", pos0 + i, win" << e << "[i]);\n";
    if (with_index) s << "    " << st << "(selv, pos0 + i, widx[i]);\n";
    s << "  }\n"
      << "  // ---- a wave tile that selected more than GDV_FP_CAP rows: its rows are read AGAIN, one sub-tile at a time, the\n"
      << "  //      predicate is evaluated again and the rows beyond the window are stored directly (bits were all appended above)\n"
      << "  if (fcount > (gdv_uint32)GDV_FP_CAP) {\n"
      << "    gdv_uint32 run = 0;\n"
      << "#pragma unroll 1\n    for (int kb = 0; kb < GDV_FP_K; kb++) {\n"
      << "    const gdv_int64 wbase = wfirst + kb * GDV_U;\n"
      << "    const gdv_int64 rbase = wbase * 64;\n";
    for (int k = 0; k < nin; k++) {
      const DataType& t = schema[plan->input_fields[k]].type;
      if (t.id == kBool && cg.needs_values_[k]) s << "    const gdv_uint64 dwt" << k << " = gdv_bitmap_tile(A.in[" << k << "].bits, wbase, lane, GDV_U);\n";
      if (cg.needs_validity_[k]) s << "    const gdv_uint64 vwt" << k << " = gdv_bitmap_tile(A.in[" << k << "].valid, wbase, lane, GDV_U);\n";
    }
    s << "#pragma unroll 1\n    for (int u = 0; u < GDV_U; u++) {\n";
    row_prologue("t");
    for (int k = 0; k < nin; k++) {
      const DataType& t = schema[plan->input_fields[k]].type;
      if (t.id != kBool && cg.needs_values_[k])
        s << "      const gdv_one<" << t.CType() << "> c" << k << "{live ? gdv_ld(in" << k << ", row) : (" << t.CType() << ")0};  // (answers c" << k << "[u])\n";
    }
    s << tail_body
      << "    }\n    }\n  }\n";
  } else {
    s << "  // ---- phase 4: the projections of the selected rows, stored compacted\n"
      << "  gdv_int64 run = 0;\n";
    for (size_t k = 0; k < bitmap_of.size(); k++) s << "  gdv_uint64 bacc" << k << " = 0;\n";
    s << "#pragma unroll\n  for (int u = 0; u < GDV_U; u++) {\n    {\n";
    row_prologue("");
    s << "      const gdv_uint64 fmu = gdv_tile_word(fm, u);\n"
      << "      const int cnt = (int)__popcll(fmu);\n"
      << "      const bool fsel = (fmu >> lane) & 1;\n"
      << "      const int below = gdv_rank_below(fmu);\n"
      << "      const gdv_int64 opos = pos0 + run + below;  // where this lane's row lands if it is selected\n"
      << "      (void)opos; (void)cnt; (void)below;\n";
    if (with_index) s << "      if (fsel) selv[opos] = (" << sel_t << ")row;\n";
    s << proj_body
      << "      run += cnt;\n"
      << "    }\n  }\n";
  }
  for (auto& f : flushes) s << f;
  s << "}\n\n"
    << "extern \"C\" __global__ void __launch_bounds__(GDV_WAVES * 64) GDV_KERNEL_NAME(const gdv_args A) {\n"
    << "  __shared__ gdv_uint32 wg_cnt[GDV_WAVES];\n"
    << "  __shared__ gdv_uint64 wg_excl;\n";
  std::string win_args;
  if (win) {
    for (size_t e = 0; e < plan->output_types.size(); e++)
      if (plan->output_types[e].id != kBool) {
        s << "  __shared__ " << plan->output_types[e].CType() << " win" << e << "[GDV_WAVES * GDV_FP_CAP];\n";
        win_args += ", win" + std::to_string(e) + " + wave * GDV_FP_CAP";
      }
    if (with_index) {
      s << "  __shared__ " << sel_t << " widx[GDV_WAVES * GDV_FP_CAP];\n";
      win_args += ", widx + wave * GDV_FP_CAP";
    }
  }
  s << "  const int lane = threadIdx.x & 63;\n"
    << "  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
    << "  // one workgroup tile = GDV_WAVES x GDV_FP_K x GDV_U x 64 rows; tile order = row order.  Round 6: the tile comes from an\n"
    << "  // atomic TICKET (one agent-scope fetch-add per workgroup, the word behind the count and the error word), not from\n"
    << "  // blockIdx: the look-back waits only for tiles with LOWER tickets, whose workgroups have already started — no\n"
    << "  // deadlock whatever order the dispatcher starts workgroups in.  One fetch-add + one barrier per 24576 rows: 0.09 ms\n"
    << "  // of 3.44 at 10^9 rows (profiles/r06_ticket_cost.txt; a workgroup walking several tiles — next ticket drawn ahead, or\n"
    << "  // behind the look-back — measured 4.3 / 3.65 ms: a drawn-but-unstarted tile stalls every tile behind it).\n"
    << (opts.fp_experiment == 3
            // EXPERIMENT (GDV_FP_EXPERIMENT=3; tools only): round 5's tile = blockIdx, to price the ticket
            ? "  const gdv_int64 tile = (gdv_int64)blockIdx.x;\n"
            : std::string(opts.fp_experiment == 2
                              // GDV_FP_EXPERIMENT=2 (tests): workgroups with LOW block indices arrive late — every group of 256
                              // consecutive block indices takes its tickets in (roughly) reversed order; results must not change
                              ? "  if (threadIdx.x == 0) for (int z = 0; z < (int)(255u - (blockIdx.x & 255u)) * 4; z++) __builtin_amdgcn_s_sleep(32);\n"
                              : "") +
                  "  __shared__ gdv_uint32 wg_ticket;\n"
                  "  if (threadIdx.x == 0) wg_ticket = __hip_atomic_fetch_add((gdv_uint32*)A.counts + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
                  "  __syncthreads();\n"
                  "  const gdv_int64 tile = (gdv_int64)wg_ticket;\n")
    << "  // (workgroup-uniform branch: the barriers inside are reached by every wave of the workgroup)\n"
    << "  if ((tile + 1) * (GDV_WAVES * GDV_FP_K * GDV_U * 64) <= GDV_ROWS(A)) gdv_fused_tile<true>(A, tile, lane, wave, wg_cnt, &wg_excl" << win_args << ");\n"
    << "  else gdv_fused_tile<false>(A, tile, lane, wave, wg_cnt, &wg_excl" << win_args << ");\n"
    << "}\n";
  std::string text = s.str();
  uint64_t h = Fnv1a(HashableSource(text) + LibraryTag(text));
  char name[64];
  snprintf(name, sizeof(name), "gdv_k_%016llx", static_cast<unsigned long long>(h));
  plan->kernel_name = name;
  for (size_t pos = text.find("GDV_KERNEL_NAME"); pos != std::string::npos; pos = text.find("GDV_KERNEL_NAME", pos))
    text.replace(pos, strlen("GDV_KERNEL_NAME"), plan->kernel_name);
  plan->source = text;
  plan->ir = text;
  plan->fp_window_rows = cap;
  plan->fp_rounds = rounds;
  return Status::OK();
}
}  // namespace

Status PlanFilterProject(const Schema& schema, const ExpressionPtr& condition, const std::vector<ExpressionPtr>& exprs,
                         SelectionMode index_mode, const CodegenOptions& opts, KernelPlan* plan) {
  if (!condition) return Status::Invalid("Condition cannot be null");
  if (exprs.empty()) return Status::Invalid("Expressions cannot be empty");
  GDV_RETURN_NOT_OK(ValidateExpression(schema, *condition));
  if (condition->root()->return_type().id != kBool)
    return Status::ValidationError("Filter condition must be of type boolean");
  for (auto& e : exprs) {
    if (!e) return Status::Invalid("Expression cannot be null");
    GDV_RETURN_NOT_OK(ValidateExpression(schema, *e));
    if (e->result().type.is_varlen())
      return Status::CodeGenError("fused filter-project: var-len outputs take the filter + projector chain");
  }
  // the direct shape always exists; the windowed one is the main kernel wherever its window fits
  auto direct = std::make_shared<KernelPlan>();
  GDV_RETURN_NOT_OK(PlanFilterProjectShape(schema, condition, exprs, index_mode, opts, FpShape::kDirect, direct.get()));
  if (opts.fp_window_bytes > 0) {
    KernelPlan windowed;
    Status st = PlanFilterProjectShape(schema, condition, exprs, index_mode, opts, FpShape::kWindow, &windowed);
    // the engine launches either kernel with the windowed plan's argument block: the direct plan's literals and
    // constant block must be a prefix of it (they are: the windowed body generates the same trees first, then the
    // tail's copies) — if that ever stops holding, the plan keeps the direct shape alone
    const bool prefix = st.ok() && direct->literals.size() <= windowed.literals.size() &&
                        std::equal(direct->literals.begin(), direct->literals.end(), windowed.literals.begin()) &&
                        windowed.const_block.compare(0, direct->const_block.size(), direct->const_block) == 0 &&
                        direct->input_fields == windowed.input_fields && direct->opts.subtiles == windowed.opts.subtiles &&
                        direct->opts.waves == windowed.opts.waves;
    if (st.ok() && prefix) {
      *plan = std::move(windowed);
      plan->exact = direct;
      return Status::OK();
    }
    if (!st.ok() && st.code != kCodeGenError) return st;
  }
  *plan = std::move(*direct);
  return Status::OK();
}

// ------------------------------------------------------------------ two-stage plans

namespace {

bool MaterialisesBytes(const Node& n) {
  if (n.kind() != NodeKind::kFunction) return false;
  auto& fn = static_cast<const FunctionNode&>(n);
  const std::string& f = fn.name();
  if (f == "concat" || f == "concatOperator" || f == "lpad" || f == "rpad" || f == "reverse" || f == "replace" || f == "initcap" ||
      f == "hashSHA256" || f == "sha256" || f == "hashSHA1" || f == "sha1" || f == "sha" || f == "hashMD5" || f == "md5")
    return true;
  return f == "castVARCHAR" && !fn.children().empty() && !fn.children()[0]->return_type().is_varlen();
}

struct Stager {
  const Schema& schema;
  StagedExpressions* out;
  std::map<std::string, NodePtr> field_of;  // hoisted sub-tree (cache key) -> its temporary field

  // `guard` (may be null): the condition under which the caller's tree evaluates `n` at all — the
  // enclosing if/else branches and short-circuit AND / OR children.  The first stage evaluates
  // every row, so a guarded sub-tree is hoisted as `if (guard) n else NULL`: functions that can
  // raise (castVARCHAR(a / b, n), replace ...) then run only where the caller's tree would have
  // run them (round-2 advisor: `if (b != 0) upper(castVARCHAR(a / b, 10)) else 'x'` raised).
  NodePtr Hoist(const NodePtr& sub, const NodePtr& guard) {
    NodePtr n = sub;
    if (guard) {
      Literal null_value;
      null_value.is_null = true;
      n = std::make_shared<IfNode>(guard, sub, std::make_shared<LiteralNode>(sub->return_type(), null_value),
                                   sub->return_type());
    }
    std::string key;
    n->AppendKey(&key);
    auto it = field_of.find(key);
    if (it != field_of.end()) return it->second;
    Field f;
    f.name = "__gdv_stage" + std::to_string(out->pre.size());
    for (bool clash = true; clash;) {  // fields are bound by name: stay clear of the caller's
      clash = false;
      for (auto& g : out->schema) clash = clash || g.name == f.name;
      if (clash) f.name += "_";
    }
    f.type = n->return_type();
    f.nullable = true;
    out->pre.push_back(std::make_shared<Expression>(n, f));
    out->schema.push_back(f);
    NodePtr field = std::make_shared<FieldNode>(f);
    field_of[key] = field;
    return field;
  }
  static NodePtr AndGuard(const NodePtr& a, const NodePtr& b) {
    if (!a) return b;
    if (!b) return a;
    return std::make_shared<BooleanNode>(BooleanNode::kAnd, NodeVector{a, b});
  }
  static NodePtr Test(const char* fn, const NodePtr& x) {  // istrue / isnottrue / isnotfalse: never null
    return std::make_shared<FunctionNode>(fn, NodeVector{x}, boolean());
  }
  // `takes_bytes`: the parent is an output root or a concat — it can take a materialising child as it is
  NodePtr Rewrite(const NodePtr& n, bool takes_bytes, const NodePtr& guard) {
    if (MaterialisesBytes(*n) && !takes_bytes) return Hoist(n, guard);  // (its own sub-tree is the first stage's business)
    switch (n->kind()) {
      case NodeKind::kFunction: {
        auto& fn = static_cast<const FunctionNode&>(*n);
        const bool is_concat = fn.name() == "concat" || fn.name() == "concatOperator";
        NodeVector kids;
        bool changed = false;
        for (auto& c : fn.children()) {
          kids.push_back(Rewrite(c, is_concat, guard));
          changed |= kids.back() != c;
        }
        return changed ? std::make_shared<FunctionNode>(fn.name(), kids, fn.return_type()) : n;
      }
      case NodeKind::kIf: {
        // guards are built from the caller's ORIGINAL condition: it must be evaluable by the first
        // stage, which knows nothing of the temporaries of this one
        auto& i = static_cast<const IfNode&>(*n);
        // (`if (c) <materialised> else NULL` is something the kernel takes as it is wherever it takes
        // a materialised value — CodeGen::Gen, kIf — which is also what a guarded hoist looks like:
        // its branch inherits `takes_bytes`, or the first stage would hoist it again, for ever)
        auto null_literal = [](const NodePtr& x) {
          return x->kind() == NodeKind::kLiteral && static_cast<const LiteralNode&>(*x).is_null();
        };
        NodePtr c = Rewrite(i.condition(), false, guard);
        NodePtr t = Rewrite(i.then_node(), takes_bytes && null_literal(i.else_node()),
                            AndGuard(guard, Test("istrue", i.condition())));
        NodePtr e = Rewrite(i.else_node(), takes_bytes && null_literal(i.then_node()),
                            AndGuard(guard, Test("isnottrue", i.condition())));
        if (c == i.condition() && t == i.then_node() && e == i.else_node()) return n;
        return std::make_shared<IfNode>(c, t, e, i.return_type());
      }
      case NodeKind::kBoolean: {
        // left-to-right short circuit: child k of an AND runs while no earlier child was (valid,
        // false); of an OR, while none was (valid, true)
        auto& b = static_cast<const BooleanNode&>(*n);
        const char* still = b.op() == BooleanNode::kAnd ? "isnotfalse" : "isnottrue";
        NodeVector kids;
        bool changed = false;
        NodePtr g = guard;
        for (auto& c : b.children()) {
          kids.push_back(Rewrite(c, false, g));
          changed |= kids.back() != c;
          g = AndGuard(g, Test(still, c));
        }
        return changed ? std::make_shared<BooleanNode>(b.op(), kids) : n;
      }
      case NodeKind::kIn: {
        auto& in = static_cast<const InNode&>(*n);
        NodePtr e = Rewrite(in.eval(), false, guard);
        return e == in.eval() ? n : std::make_shared<InNode>(e, in.value_type(), in.values());
      }
      default:
        return n;
    }
  }
};

}  // namespace

void StageMaterialisedValues(const Schema& schema, const std::vector<ExpressionPtr>& exprs,
                             StagedExpressions* out) {
  out->pre.clear();
  out->main.clear();
  out->schema = schema;
  Stager st{schema, out, {}};
  for (auto& e : exprs) {
    if (!e || !e->root()) {
      out->main.push_back(e);
      continue;
    }
    NodePtr root = st.Rewrite(e->root(), true, nullptr);
    out->main.push_back(root == e->root() ? e : std::make_shared<Expression>(root, e->result()));
  }
}

}  // namespace gdv
