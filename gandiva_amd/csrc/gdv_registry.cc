#include "gdv_registry.h"

#include <cstdio>
#include <cstdlib>

#include <algorithm>

namespace gdv {

std::string FunctionDef::SignatureString() const {
  std::string s = ret.ToString() + " " + name + "(";
  for (size_t i = 0; i < params.size(); i++) {
    if (i) s += ", ";
    s += params[i].ToString();
  }
  return s + ")";
}

void FunctionRegistry::Add(FunctionDef def) {
  // one definition per (name, parameter types): a second one would be listed twice by
  // GetRegisteredFunctionSignatures and silently shadowed in Lookup
  auto range = by_name_.equal_range(def.name);
  for (auto it = range.first; it != range.second; ++it) {
    if (defs_[it->second].params == def.params) {
      fprintf(stderr, "gandiva_amd: function %s registered twice with the same parameter types\n", def.name.c_str());
      std::abort();
    }
  }
  by_name_.emplace(def.name, defs_.size());
  defs_.push_back(std::move(def));
}

const FunctionRegistry& FunctionRegistry::Get() {
  static FunctionRegistry reg;
  return reg;
}

static bool ParamMatches(const DataType& want, const DataType& got) {
  if (want.id != got.id) return false;
  if (want.id == kDecimal128) return true;
  if (want.id == kTimestamp || want.id == kTime32 || want.id == kTime64)
    return want.precision == got.precision;
  return true;
}

const FunctionDef* FunctionRegistry::Lookup(const std::string& name,
                                            const std::vector<DataType>& params) const {
  auto range = by_name_.equal_range(name);
  for (auto it = range.first; it != range.second; ++it) {
    const FunctionDef& d = defs_[it->second];
    if (d.params.size() != params.size()) continue;
    bool ok = true;
    for (size_t i = 0; i < params.size() && ok; i++) ok = ParamMatches(d.params[i], params[i]);
    if (ok) return &d;
  }
  return nullptr;
}

namespace {

std::string Sym(const std::string& name, const std::vector<DataType>& params) {
  std::string s = name;
  for (auto& p : params) s += "_" + p.Suffix();
  return s;
}

}  // namespace

FunctionRegistry::FunctionRegistry() {
  const std::vector<DataType> ints = {int8(),  int16(),  int32(),  int64(),
                                      uint8(), uint16(), uint32(), uint64()};
  const std::vector<DataType> floats = {float32(), float64()};
  std::vector<DataType> numerics = ints;
  numerics.insert(numerics.end(), floats.begin(), floats.end());
  const std::vector<DataType> dates = {date32(), date64(), timestamp(), time32(), time64()};

  auto add = [&](const std::string& name, std::vector<DataType> params, DataType ret,
                 NullPolicy policy = NullPolicy::kNullIfNull, uint32_t flags = 0,
                 std::string symbol = "") {
    FunctionDef d;
    d.name = name;
    d.params = std::move(params);
    d.ret = ret;
    d.policy = policy;
    d.flags = flags;
    d.symbol = symbol.empty() ? Sym(name, d.params) : symbol;
    Add(std::move(d));
  };

  // arithmetic
  for (auto& t : numerics) {
    add("add", {t, t}, t);
    add("subtract", {t, t}, t);
    add("multiply", {t, t}, t);
    add("divide", {t, t}, t, NullPolicy::kNullIfNull, kNeedsContext);
  }
  add("mod", {int64(), int32()}, int32());
  add("mod", {int64(), int64()}, int64());
  add("mod", {int32(), int32()}, int32());
  add("mod", {float64(), float64()}, float64(), NullPolicy::kNullIfNull, kNeedsContext);
  // aliases of the reference registry (recollection: mod = modulo, power = pow, locate = position)
  add("modulo", {int64(), int32()}, int32(), NullPolicy::kNullIfNull, 0, Sym("mod", {int64(), int32()}));
  add("modulo", {int64(), int64()}, int64(), NullPolicy::kNullIfNull, 0, Sym("mod", {int64(), int64()}));
  add("modulo", {int32(), int32()}, int32(), NullPolicy::kNullIfNull, 0, Sym("mod", {int32(), int32()}));
  add("modulo", {float64(), float64()}, float64(), NullPolicy::kNullIfNull, kNeedsContext,
      Sym("mod", {float64(), float64()}));
  for (auto& t : {int32(), int64(), float32(), float64()}) {
    add("negative", {t}, t);
    add("abs", {t}, t);
    add("greatest", {t, t}, t);
    add("least", {t, t}, t);
  }

  // relational
  std::vector<DataType> comparable = numerics;
  comparable.push_back(boolean());
  comparable.insert(comparable.end(), dates.begin(), dates.end());
  for (auto& t : comparable) {
    for (const char* op : {"equal", "not_equal", "less_than", "less_than_or_equal_to",
                           "greater_than", "greater_than_or_equal_to"}) {
      add(op, {t, t}, boolean());
    }
    // aliases of the reference registry
    add("eq", {t, t}, boolean(), NullPolicy::kNullIfNull, 0, Sym("equal", {t, t}));
    add("same", {t, t}, boolean(), NullPolicy::kNullIfNull, 0, Sym("equal", {t, t}));
  }
  add("not", {boolean()}, boolean());

  // null handling: value functions see validity
  std::vector<DataType> all_fixed = comparable;
  for (auto& t : all_fixed) {
    add("isnull", {t}, boolean(), NullPolicy::kNullNever, 0, "gdv_isnull");
    add("isnotnull", {t}, boolean(), NullPolicy::kNullNever, 0, "gdv_isnotnull");
    add("is_distinct_from", {t, t}, boolean(), NullPolicy::kNullNever, 0, "gdv_is_distinct_from");
    add("is_not_distinct_from", {t, t}, boolean(), NullPolicy::kNullNever, 0,
        "gdv_is_not_distinct_from");
  }
  for (auto& t : numerics) {
    add("isnumeric", {t}, boolean(), NullPolicy::kNullNever, 0, "gdv_isnotnull");
  }

  // bitwise, boolean tests, nvl
  for (auto& t : {int32(), int64(), uint32(), uint64()}) {
    add("bitwise_and", {t, t}, t);
    add("bitwise_or", {t, t}, t);
    add("bitwise_xor", {t, t}, t);
    add("bitwise_not", {t}, t);
  }
  for (const char* f : {"istrue", "isfalse", "isnottrue", "isnotfalse"})
    add(f, {boolean()}, boolean(), NullPolicy::kNullNever);
  for (auto& t : all_fixed) add("nvl", {t, t}, t, NullPolicy::kNullInternal, 0, "gdv_nvl");

  // casts
  add("castBIGINT", {int32()}, int64());
  add("castINT", {int64()}, int32());
  add("castFLOAT4", {int32()}, float32());
  add("castFLOAT4", {int64()}, float32());
  add("castFLOAT4", {float64()}, float32());
  add("castFLOAT8", {int32()}, float64());
  add("castFLOAT8", {int64()}, float64());
  add("castFLOAT8", {float32()}, float64());
  add("castBIGINT", {float32()}, int64());
  add("castBIGINT", {float64()}, int64());
  add("castINT", {float32()}, int32());
  add("castINT", {float64()}, int32());
  add("castDATE", {int64()}, date64());
  add("castDATE", {date32()}, date64());
  add("castDATE", {timestamp()}, date64());
  add("castDATE32", {date64()}, date32());
  add("castTIMESTAMP", {int64()}, timestamp());
  add("castTIMESTAMP", {date64()}, timestamp());
  add("castBIGINT", {date64()}, int64());
  add("castBIGINT", {timestamp()}, int64());

  // extended math
  for (const char* f : {"cbrt", "exp", "log", "log10", "sqrt", "floor", "ceil", "round",
                        "truncate"}) {
    add(f, {float64()}, float64());
  }
  add("power", {float64(), float64()}, float64());
  add("pow", {float64(), float64()}, float64(), NullPolicy::kNullIfNull, 0, Sym("power", {float64(), float64()}));
  add("log", {float64(), float64()}, float64(), NullPolicy::kNullIfNull, kNeedsContext);

  // hash family: never null, null input hashes to the seed
  std::vector<DataType> hashable = numerics;
  hashable.push_back(boolean());
  for (auto& t : {date32(), date64(), timestamp(), time32()}) hashable.push_back(t);
  for (auto& t : {utf8(), binary()}) hashable.push_back(t);  // MurmurHash3 over the bytes
  for (auto& t : hashable) {
    add("hash", {t}, int32(), NullPolicy::kNullNever, 0, Sym("hash32", {t}));
    add("hash32", {t}, int32(), NullPolicy::kNullNever);
    add("hash32AsDouble", {t}, int32(), NullPolicy::kNullNever, 0, Sym("hash32", {t}));
    add("hash64", {t}, int64(), NullPolicy::kNullNever);
    add("hash64AsDouble", {t}, int64(), NullPolicy::kNullNever, 0, Sym("hash64", {t}));
    add("hash32", {t, int32()}, int32(), NullPolicy::kNullNever);
    add("hash32AsDouble", {t, int32()}, int32(), NullPolicy::kNullNever, 0,
        Sym("hash32", {t, int32()}));
    add("hash64", {t, int64()}, int64(), NullPolicy::kNullNever);
    add("hash64AsDouble", {t, int64()}, int64(), NullPolicy::kNullNever, 0,
        Sym("hash64", {t, int64()}));
  }

  // date / time
  for (auto& t : {date32(), date64(), timestamp()}) {
    for (const char* f : {"extractYear", "extractMonth", "extractDay", "extractQuarter",
                          "extractDoy", "extractDow", "extractHour", "extractMinute",
                          "extractSecond", "extractEpoch", "extractDecade", "extractCentury",
                          "extractMillennium"}) {
      add(f, {t}, int64());
    }
  }
  for (const char* f : {"extractHour", "extractMinute", "extractSecond"}) {
    add(f, {time32()}, int64());
  }
  for (auto& t : {date64(), timestamp()}) {
    // round 4 (registry tail): unit starts, ISO week, end of month
    for (const char* f : {"date_trunc_Second", "date_trunc_Minute", "date_trunc_Hour", "date_trunc_Day", "date_trunc_Week",
                          "date_trunc_Month", "date_trunc_Quarter", "date_trunc_Year", "date_trunc_Decade",
                          "date_trunc_Century", "date_trunc_Millennium"})
      add(f, {t}, t);
    add("extractWeek", {t}, int64());
    add("weekofyear", {t}, int64(), NullPolicy::kNullIfNull, 0, Sym("extractWeek", {t}));
    add("last_day", {t}, date64());
    for (const char* f : {"timestampaddSecond", "timestampaddMinute", "timestampaddHour",
                          "timestampaddDay", "timestampaddWeek", "timestampaddMonth",
                          "timestampaddQuarter", "timestampaddYear"}) {
      add(f, {int64(), t}, t);
    }
    add("date_add", {t, int64()}, t);
    add("date_sub", {t, int64()}, t);
    add("date_add", {t, int32()}, t);
    add("date_sub", {t, int32()}, t);
    for (const char* f : {"timestampdiffSecond", "timestampdiffMinute", "timestampdiffHour",
                          "timestampdiffDay", "timestampdiffWeek", "timestampdiffMonth",
                          "timestampdiffQuarter", "timestampdiffYear"}) {
      add(f, {t, t}, int32());
    }
    add("datediff", {t, t}, int32());
    add("date_diff", {t, t}, int32(), NullPolicy::kNullIfNull, 0, Sym("datediff", {t, t}));
  }
  // round 5: to_date(text, 'pattern'[, suppress_errors]) — the holder's pattern is compiled by the planner;
  // to_timestamp / to_time over numbers (seconds since the epoch)
  add("to_date", {utf8(), utf8()}, date64(), NullPolicy::kNullInternal, kNeedsContext | kDateFormatArg, "gdv_parse_date");
  add("to_date", {utf8(), utf8(), int32()}, date64(), NullPolicy::kNullInternal, kNeedsContext | kDateFormatArg, "gdv_parse_date");
  for (auto& t : {int32(), int64(), float32(), float64()}) {
    add("to_timestamp", {t}, timestamp());
    add("to_time", {t}, time32());
  }
  // decimal128: precision/scale are wildcards in the parameter match
  {
    const DataType dec = decimal128(38, 0);  // enumerated like the reference's decimal128()
    for (const char* f : {"add", "subtract", "multiply"})
      add(f, {dec, dec}, dec, NullPolicy::kNullIfNull, kDecimalResult | kDecimalArgs);
    for (const char* f : {"divide", "mod"})
      add(f, {dec, dec}, dec, NullPolicy::kNullIfNull, kDecimalResult | kDecimalArgs | kNeedsContext);
    for (const char* f : {"equal", "not_equal", "less_than", "less_than_or_equal_to", "greater_than",
                          "greater_than_or_equal_to"})
      add(f, {dec, dec}, boolean(), NullPolicy::kNullIfNull, kDecimalArgs);
    add("negative", {dec}, dec, NullPolicy::kNullIfNull, kDecimalArgs);
    add("abs", {dec}, dec, NullPolicy::kNullIfNull, kDecimalArgs);
    // round / truncate (trunc) / ceil / floor (round 5): the result's precision and scale are the expression's own
    for (const char* f : {"round", "truncate", "trunc"}) {
      const std::string base = std::string(f) == "round" ? "round" : "truncate";
      add(f, {dec}, dec, NullPolicy::kNullIfNull, kDecimalArgs, base + "_decimal128");
      add(f, {dec, int32()}, dec, NullPolicy::kNullIfNull, kDecimalArgs, base + "_decimal128_int32");
    }
    add("ceil", {dec}, dec, NullPolicy::kNullIfNull, kDecimalArgs);
    add("floor", {dec}, dec, NullPolicy::kNullIfNull, kDecimalArgs);
    add("castDECIMAL", {int64()}, dec, NullPolicy::kNullIfNull, kDecimalArgs);
    add("castDECIMAL", {int32()}, dec, NullPolicy::kNullIfNull, kDecimalArgs);
    add("castDECIMAL", {dec}, dec, NullPolicy::kNullIfNull, kDecimalArgs);
    add("castFLOAT8", {dec}, float64(), NullPolicy::kNullIfNull, kDecimalArgs);
    add("castBIGINT", {dec}, int64(), NullPolicy::kNullIfNull, kDecimalArgs);
    add("castVARCHAR", {dec, int64()}, utf8(), NullPolicy::kNullIfNull, kDecimalArgs | kVarlenResult | kNeedsContext);
    add("isnull", {dec}, boolean(), NullPolicy::kNullNever, 0, "gdv_isnull");
    add("isnotnull", {dec}, boolean(), NullPolicy::kNullNever, 0, "gdv_isnotnull");
  }
  // utf8 / binary
  for (auto& t : {utf8(), binary()}) {
    for (const char* op : {"equal", "not_equal", "less_than", "less_than_or_equal_to",
                           "greater_than", "greater_than_or_equal_to"})
      add(op, {t, t}, boolean(), NullPolicy::kNullIfNull, 0, std::string(op) + "_utf8_utf8");
    add("isnull", {t}, boolean(), NullPolicy::kNullNever, 0, "gdv_isnull");
    add("isnotnull", {t}, boolean(), NullPolicy::kNullNever, 0, "gdv_isnotnull");
    add("octet_length", {t}, int32(), NullPolicy::kNullIfNull, 0, "octet_length_utf8");
    add("bit_length", {t}, int32(), NullPolicy::kNullIfNull, 0, "bit_length_utf8");
  }
  add("starts_with", {utf8(), utf8()}, boolean());
  add("ends_with", {utf8(), utf8()}, boolean());
  add("char_length", {utf8()}, int32());
  add("length", {utf8()}, int32(), NullPolicy::kNullIfNull, 0, "char_length_utf8");
  add("lengthUtf8", {binary()}, int32(), NullPolicy::kNullIfNull, 0, "char_length_utf8");
  add("like", {utf8(), utf8()}, boolean(), NullPolicy::kNullIfNull, kPatternArg, "gdv_like");
  add("like", {utf8(), utf8(), utf8()}, boolean(), NullPolicy::kNullIfNull, kPatternArg, "gdv_like");
  // Regular expressions (round 5).  regexp_like / regexp_matches: gdv_node.h MakeFunctionNode rewrites 'lit', '^lit', 'lit$',
  // '^lit$' onto like when the tree is built (the sweep's match bits answer those); every other pattern is compiled by the
  // planner into a position automaton of at most 63 positions (gdv_regex.h lists the syntax; what it does not take is refused
  // with the reason).  regexp_replace: a literal pattern and a replacement without backslashes only (-> replace).  [recalled:
  // the lineage's holders compile the pattern with RE2 once per expression]
  add("regexp_like", {utf8(), utf8()}, boolean(), NullPolicy::kNullIfNull, kPatternArg, "gdv_regex_search");
  add("regexp_matches", {utf8(), utf8()}, boolean(), NullPolicy::kNullIfNull, kPatternArg, "gdv_regex_search");
  add("regexp_replace", {utf8(), utf8(), utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult, "gdv_regexp_unsupported");
  // ilike (round 4): like without regard to the case of ASCII letters — both sides are read through the
  // lower-case byte map, so every fast path of like (prefix / suffix / equality / '%needle%' answered by
  // the byte sweep) serves it.  The lineage folds case through RE2 (Unicode simple folding); letters
  // outside ASCII compare exactly here: recollection + a stated limit, "parity unpinned".
  add("ilike", {utf8(), utf8()}, boolean(), NullPolicy::kNullIfNull, kPatternArg, "gdv_like");
  add("upper", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("lower", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("substr", {utf8(), int64(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("substring", {utf8(), int64(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult,
      "substr_utf8_int64_int64");
  add("substr", {utf8(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("substring", {utf8(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult,
      "substr_utf8_int64");
  // concat (null argument = empty string, never null) and || (null if any argument is null),
  // 2..6 arguments; planned as a list of pieces, not as a call (gdv_planner.cc)
  for (int nargs = 2; nargs <= 6; nargs++) {
    std::vector<DataType> ps(nargs, utf8());
    add("concat", ps, utf8(), NullPolicy::kNullNever, kVarlenResult, "gdv_concat");
    add("concatOperator", ps, utf8(), NullPolicy::kNullIfNull, kVarlenResult, "gdv_concat");
  }
  add("left", {utf8(), int32()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("right", {utf8(), int32()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("castVARCHAR", {utf8(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult | kNeedsContext);
  add("castVARCHAR", {int32(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult | kNeedsContext);
  add("castVARCHAR", {int64(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult | kNeedsContext);
  // round 5: shortest round-trip digits in the Java-compatible layout (gandiva/formatting_utils.h, as recalled)
  add("castVARCHAR", {float32(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult | kNeedsContext);
  add("castVARCHAR", {float64(), int64()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult | kNeedsContext);
  add("reverse", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult | kNeedsContext);
  add("initcap", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  // message digests (round 5): lower-case hex text, never null (a NULL hashes as the empty message)
  for (auto& t : std::vector<DataType>{utf8(), binary(), int32(), int64(), float32(), float64()}) {
    for (const char* f : {"hashSHA256", "sha256"}) add(f, {t}, utf8(), NullPolicy::kNullNever, kVarlenResult, Sym("hashSHA256", {t}));
    for (const char* f : {"hashSHA1", "sha1", "sha"}) add(f, {t}, utf8(), NullPolicy::kNullNever, kVarlenResult, Sym("hashSHA1", {t}));
    for (const char* f : {"hashMD5", "md5"}) add(f, {t}, utf8(), NullPolicy::kNullNever, kVarlenResult, Sym("hashMD5", {t}));
  }
  add("replace", {utf8(), utf8(), utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult | kNeedsContext,
      "gdv_replace");  // planned by gdv_planner.cc (literal from / to: a table + the sweep's match bits; otherwise per row)
  // lpad / rpad: planned as two pieces (gdv_planner.cc); a length / fill that is not a literal: the fill read cyclically, per row
  for (const char* f : {"lpad", "rpad"}) {
    add(f, {utf8(), int32()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult, "gdv_pad");
    add(f, {utf8(), int32(), utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult, "gdv_pad");
  }
  add("locate", {utf8(), utf8()}, int32(), NullPolicy::kNullIfNull, kNeedsContext);
  add("locate", {utf8(), utf8(), int32()}, int32(), NullPolicy::kNullIfNull, kNeedsContext);
  add("position", {utf8(), utf8()}, int32(), NullPolicy::kNullIfNull, kNeedsContext, Sym("locate", {utf8(), utf8()}));
  add("position", {utf8(), utf8(), int32()}, int32(), NullPolicy::kNullIfNull, kNeedsContext,
      Sym("locate", {utf8(), utf8(), int32()}));
  add("strpos", {utf8(), utf8()}, int32(), NullPolicy::kNullIfNull, kNeedsContext);
  add("ascii", {utf8()}, int32());
  add("castINT", {utf8()}, int32(), NullPolicy::kNullIfNull, kNeedsContext);
  add("castBIGINT", {utf8()}, int64(), NullPolicy::kNullIfNull, kNeedsContext);
  add("ltrim", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("rtrim", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("btrim", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult);
  add("trim", {utf8()}, utf8(), NullPolicy::kNullIfNull, kVarlenResult, "btrim_utf8");
  add("datediff", {date32(), date32()}, int32());
  add("date_diff", {date32(), date32()}, int32(), NullPolicy::kNullIfNull, 0,
      "datediff_date32_date32");
}

// ------------------------------------------------------------------ decimal result types

DataType DecimalResultType(DecimalOp op, const DataType& a, const DataType& b) {
  const int32_t kMaxPrecision = 38;
  const int32_t kMinAdjustedScale = 6;
  int32_t p1 = a.precision, s1 = a.scale, p2 = b.precision, s2 = b.scale;
  int32_t scale = 0, precision = 0;
  switch (op) {
    case DecimalOp::kAdd:
    case DecimalOp::kSubtract:
      scale = std::max(s1, s2);
      precision = std::max(p1 - s1, p2 - s2) + scale + 1;
      break;
    case DecimalOp::kMultiply:
      scale = s1 + s2;
      precision = p1 + p2 + 1;
      break;
    case DecimalOp::kDivide:
      scale = std::max(kMinAdjustedScale, s1 + p2 + 1);
      precision = p1 - s1 + s2 + scale;
      break;
    case DecimalOp::kMod:
      scale = std::max(s1, s2);
      precision = std::min(p1 - s1, p2 - s2) + scale;
      break;
  }
  if (precision > kMaxPrecision) {
    int32_t delta = precision - kMaxPrecision;
    int32_t min_scale = std::min(scale, kMinAdjustedScale);
    precision = kMaxPrecision;
    scale = std::max(scale - delta, min_scale);
  }
  return decimal128(precision, scale);
}

}  // namespace gdv
