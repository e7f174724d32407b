// Schema / expression trees from protobuf BYTES: the `build` half of the reference's JNI boundary
// (SURVEY.md §2 rows 18 / 20, §8 f4).  The Java side of Gandiva serialises its Schema, ExpressionList
// and Condition with protobuf (java: GandivaTypes, generated from proto/Types.proto) and hands the
// bytes to JNI buildProjector / buildFilter, where the C++ side walks the generated message classes
// and calls TreeExprBuilder.  There is no protoc (nor libprotobuf headers) in this image, so the
// subset of the wire format those messages use — varints, 32/64-bit fixed fields, length-delimited
// nested messages — is decoded by hand here.
//
// The MESSAGE LAYOUT below is a restatement from memory of the lineage's Types.proto (the
// reference mount holds no source): field numbers and enum values are listed so a maintainer can
// diff them against the real file.  "Parity unpinned" — the wire encoding itself is the public
// protobuf encoding and is exercised against hand-encoded messages (tests/test_proto_build.py).
//
//   enum GandivaType { NONE=0 BOOL=1 UINT8=2 INT8=3 UINT16=4 INT16=5 UINT32=6 INT32=7 UINT64=8
//                      INT64=9 HALF_FLOAT=10 FLOAT=11 DOUBLE=12 UTF8=13 BINARY=14
//                      FIXED_SIZE_BINARY=15 DATE32=16 DATE64=17 TIMESTAMP=18 TIME32=19 TIME64=20
//                      INTERVAL=21 DECIMAL=22 LIST=23 STRUCT=24 UNION=25 DICTIONARY=26 MAP=27 }
//   enum TimeUnit    { SEC=0 MILLISEC=1 MICROSEC=2 NANOSEC=3 }
//   ExtGandivaType   { type=1 width=2 precision=3 scale=4 dateUnit=5 timeUnit=6 timeZone=7 intervalType=8 }
//   Field            { name=1 type=2 nullable=3 children=4 }
//   Schema           { columns=1 (repeated Field) }
//   TreeNode         { fieldNode=1 fnNode=2 ifNode=6 andNode=7 orNode=8 nullNode=11 intNode=12
//                      floatNode=13 longNode=14 booleanNode=15 doubleNode=16 stringNode=17
//                      binaryNode=18 decimalNode=19 inNode=21 }
//   FieldNode        { field=1 }
//   FunctionNode     { functionName=1 inArgs=2 (repeated TreeNode) returnType=3 }
//   IfNode           { cond=1 thenNode=2 elseNode=3 returnType=4 }
//   AndNode / OrNode { args=1 (repeated TreeNode) }
//   NullNode         { type=1 }
//   IntNode / LongNode / BooleanNode { value=1 (varint) }   FloatNode { value=1 (fixed32) }
//   DoubleNode       { value=1 (fixed64) }   StringNode / BinaryNode { value=1 (bytes) }
//   DecimalNode      { value=1 (string of digits) precision=2 scale=3 }
//   InNode           { node=1 intValues=2 longValues=3 stringValues=4 binaryValues=5 }
//   IntConstants { intValues=1 (repeated IntNode) }  LongConstants { longValues=1 }
//   StringConstants { stringValues=1 }  BinaryConstants { binaryValues=1 }
//   ExpressionRoot   { root=1 resultType=2 (Field) }
//   ExpressionList   { exprs=2 (repeated ExpressionRoot) }
//   Condition        { root=1 }
#include "gdv_proto.h"

#include <cstring>

namespace gdv {

namespace {

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;

  bool done() const { return p >= end || !ok; }
  uint64_t Varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) { ok = false; return 0; }
      const uint8_t b = *p++;
      v |= static_cast<uint64_t>(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;  // more than 10 bytes
    return 0;
  }
  // next field: number + wire type; payload of length-delimited fields as a sub-reader
  bool Next(int* field, int* wire) {
    if (done()) return false;
    const uint64_t key = Varint();
    if (!ok) return false;
    *field = static_cast<int>(key >> 3);
    *wire = static_cast<int>(key & 7);
    if (*field == 0) ok = false;
    return ok;
  }
  Reader Sub() {
    const uint64_t n = Varint();
    if (!ok || n > static_cast<uint64_t>(end - p)) { ok = false; return Reader{p, p}; }
    Reader r{p, p + n};
    p += n;
    return r;
  }
  std::string Bytes() {
    Reader r = Sub();
    return ok ? std::string(reinterpret_cast<const char*>(r.p), static_cast<size_t>(r.end - r.p)) : std::string();
  }
  uint32_t Fixed32() {
    if (end - p < 4) { ok = false; return 0; }
    uint32_t v;
    std::memcpy(&v, p, 4);
    p += 4;
    return v;
  }
  uint64_t Fixed64() {
    if (end - p < 8) { ok = false; return 0; }
    uint64_t v;
    std::memcpy(&v, p, 8);
    p += 8;
    return v;
  }
  void Skip(int wire) {  // unknown fields are skipped, as protobuf readers do
    switch (wire) {
      case 0: (void)Varint(); break;
      case 1: (void)Fixed64(); break;
      case 2: (void)Sub(); break;
      case 5: (void)Fixed32(); break;
      default: ok = false;  // groups (3, 4) are not used by these messages
    }
  }
};

Status Malformed(const char* what) { return Status::Invalid(std::string("malformed protobuf message: ") + what); }

Status DecodeType(Reader r, DataType* out) {
  int gtype = 0, precision = 0, scale = 0, time_unit = -1;
  int f, w;
  while (r.Next(&f, &w)) {
    if (f == 1 && w == 0) gtype = static_cast<int>(r.Varint());
    else if (f == 3 && w == 0) precision = static_cast<int32_t>(r.Varint());
    else if (f == 4 && w == 0) scale = static_cast<int32_t>(r.Varint());
    else if (f == 6 && w == 0) time_unit = static_cast<int>(r.Varint());
    else r.Skip(w);
  }
  if (!r.ok) return Malformed("ExtGandivaType");
  // type parameters are checked here, as the JNI side rejects what it cannot map: a time unit is
  // SEC / MILLISEC / MICROSEC / NANOSEC (0..3), time32 counts seconds or milliseconds, time64
  // microseconds or nanoseconds, a decimal128 has 1 <= precision <= 38 and 0 <= scale <= precision
  if ((gtype == 18 || gtype == 19 || gtype == 20) && time_unit != -1 && (time_unit < 0 || time_unit > 3))
    return Status::Invalid("malformed protobuf message: time unit " + std::to_string(time_unit) + " is not one of 0..3");
  if (gtype == 19 && time_unit != -1 && time_unit != kSecond && time_unit != kMilli)
    return Status::Invalid("malformed protobuf message: time32 counts seconds or milliseconds");
  if (gtype == 20 && time_unit != -1 && time_unit != kMicro && time_unit != kNano)
    return Status::Invalid("malformed protobuf message: time64 counts microseconds or nanoseconds");
  if (gtype == 22 && (precision < 1 || precision > 38 || scale < 0 || scale > precision))
    return Status::Invalid("malformed protobuf message: decimal128(" + std::to_string(precision) + ", " +
                           std::to_string(scale) + ") is outside 1 <= precision <= 38, 0 <= scale <= precision");
  switch (gtype) {
    case 1: *out = boolean(); break;
    case 2: *out = uint8(); break;
    case 3: *out = int8(); break;
    case 4: *out = uint16(); break;
    case 5: *out = int16(); break;
    case 6: *out = uint32(); break;
    case 7: *out = int32(); break;
    case 8: *out = uint64(); break;
    case 9: *out = int64(); break;
    case 11: *out = float32(); break;
    case 12: *out = float64(); break;
    case 13: *out = utf8(); break;
    case 14: *out = binary(); break;
    case 16: *out = date32(); break;
    case 17: *out = date64(); break;
    case 18: *out = timestamp(time_unit >= 0 ? time_unit : kMilli); break;
    case 19: *out = time32(time_unit >= 0 ? time_unit : kMilli); break;
    case 20: *out = time64(time_unit >= 0 ? time_unit : kMicro); break;
    case 22: *out = decimal128(precision, scale); break;
    default:
      return Status::NotImplemented("GandivaType " + std::to_string(gtype) + " is not evaluated by the HIP backend");
  }
  return Status::OK();
}

Status DecodeField(Reader r, Field* out) {
  out->nullable = true;
  bool have_type = false;
  int f, w;
  while (r.Next(&f, &w)) {
    if (f == 1 && w == 2) out->name = r.Bytes();
    else if (f == 2 && w == 2) { GDV_RETURN_NOT_OK(DecodeType(r.Sub(), &out->type)); have_type = true; }
    else if (f == 3 && w == 0) out->nullable = r.Varint() != 0;
    else if (f == 4 && w == 2) return Status::NotImplemented("nested fields are not evaluated by the HIP backend");
    else r.Skip(w);
  }
  if (!r.ok || !have_type) return Malformed("Field");
  return Status::OK();
}

Status DecodeNode(Reader r, int depth, NodePtr* out);

Status DecodeChildren(Reader& r, int field_no, int depth, NodeVector* kids, DataType* ret, int ret_field,
                      std::string* name) {
  int f, w;
  while (r.Next(&f, &w)) {
    if (name != nullptr && f == 1 && w == 2) *name = r.Bytes();
    else if (f == field_no && w == 2) {
      NodePtr k;
      GDV_RETURN_NOT_OK(DecodeNode(r.Sub(), depth + 1, &k));
      kids->push_back(std::move(k));
    } else if (ret != nullptr && f == ret_field && w == 2) {
      GDV_RETURN_NOT_OK(DecodeType(r.Sub(), ret));
    } else {
      r.Skip(w);
    }
  }
  return r.ok ? Status::OK() : Malformed("node");
}

Literal Fixed(uint64_t lo) {
  Literal l;
  l.lo = lo;
  return l;
}

// decimal text ("-123456", digits only after the sign) -> 128-bit two's complement
bool ParseDecimalDigits(const std::string& text, Literal* out) {
  size_t i = 0;
  bool neg = false;
  if (i < text.size() && (text[i] == '-' || text[i] == '+')) neg = text[i++] == '-';
  if (i >= text.size()) return false;
  unsigned __int128 v = 0;
  int digits = 0;
  for (; i < text.size(); i++) {
    if (text[i] < '0' || text[i] > '9') return false;
    if (v != 0 || text[i] != '0') digits++;
    if (digits > 38) return false;  // beyond decimal128: the accumulator would wrap silently
    v = v * 10 + static_cast<unsigned>(text[i] - '0');
  }
  if (neg) v = ~v + 1;
  out->lo = static_cast<uint64_t>(v);
  out->hi = static_cast<uint64_t>(v >> 64);
  return true;
}

Status DecodeIn(Reader r, int depth, NodePtr* out) {
  NodePtr eval;
  DataType vt;
  std::vector<Literal> values;
  bool have_values = false;
  int f, w;
  while (r.Next(&f, &w)) {
    if (f == 1 && w == 2) {
      GDV_RETURN_NOT_OK(DecodeNode(r.Sub(), depth + 1, &eval));
    } else if (f >= 2 && f <= 5 && w == 2) {
      // XConstants { repeated XNode values = 1 }, XNode { value = 1 }
      have_values = true;
      vt = f == 2 ? int32() : f == 3 ? int64() : f == 4 ? utf8() : binary();
      Reader list = r.Sub();
      int lf, lw;
      while (list.Next(&lf, &lw)) {
        if (lf != 1 || lw != 2) { list.Skip(lw); continue; }
        Reader item = list.Sub();
        int vf, vw;
        Literal lit;
        while (item.Next(&vf, &vw)) {
          if (vf == 1 && vw == 0 && f <= 3) {
            const uint64_t raw = item.Varint();
            lit.lo = f == 2 ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(raw))) : raw;
          } else if (vf == 1 && vw == 2 && f >= 4) {
            lit.bytes = item.Bytes();
          } else {
            item.Skip(vw);
          }
        }
        if (!item.ok) return Malformed("IN constant");
        values.push_back(std::move(lit));
      }
      if (!list.ok) return Malformed("IN constants");
    } else {
      r.Skip(w);
    }
  }
  if (!r.ok || !eval || !have_values) return Malformed("InNode");
  *out = std::make_shared<InNode>(eval, vt, std::move(values));
  return Status::OK();
}

Status DecodeNode(Reader r, int depth, NodePtr* out) {
  if (depth > 200) return Status::Invalid("expression tree nested too deeply");
  int f, w;
  if (!r.Next(&f, &w) || w != 2) return Malformed("TreeNode");
  Reader body = r.Sub();
  if (!r.ok) return Malformed("TreeNode");
  switch (f) {
    case 1: {  // FieldNode { field = 1 }
      int bf, bw;
      Field fld;
      bool have = false;
      while (body.Next(&bf, &bw)) {
        if (bf == 1 && bw == 2) { GDV_RETURN_NOT_OK(DecodeField(body.Sub(), &fld)); have = true; }
        else body.Skip(bw);
      }
      if (!body.ok || !have) return Malformed("FieldNode");
      *out = std::make_shared<FieldNode>(fld);
      return Status::OK();
    }
    case 2: {  // FunctionNode
      NodeVector kids;
      DataType ret;
      std::string name;
      GDV_RETURN_NOT_OK(DecodeChildren(body, 2, depth, &kids, &ret, 3, &name));
      *out = MakeFunctionNode(name, std::move(kids), ret);
      return Status::OK();
    }
    case 6: {  // IfNode { cond = 1, then = 2, else = 3, returnType = 4 }
      NodePtr c, t, e;
      DataType ret;
      int bf, bw;
      while (body.Next(&bf, &bw)) {
        if (bw == 2 && bf >= 1 && bf <= 3) {
          NodePtr k;
          GDV_RETURN_NOT_OK(DecodeNode(body.Sub(), depth + 1, &k));
          (bf == 1 ? c : bf == 2 ? t : e) = std::move(k);
        } else if (bw == 2 && bf == 4) {
          GDV_RETURN_NOT_OK(DecodeType(body.Sub(), &ret));
        } else {
          body.Skip(bw);
        }
      }
      if (!body.ok || !c || !t || !e) return Malformed("IfNode");
      *out = std::make_shared<IfNode>(c, t, e, ret);
      return Status::OK();
    }
    case 7:
    case 8: {  // AndNode / OrNode { args = 1 }
      NodeVector kids;
      GDV_RETURN_NOT_OK(DecodeChildren(body, 1, depth, &kids, nullptr, 0, nullptr));
      *out = std::make_shared<BooleanNode>(f == 7 ? BooleanNode::kAnd : BooleanNode::kOr, std::move(kids));
      return Status::OK();
    }
    case 11: {  // NullNode { type = 1 }
      DataType t;
      int bf, bw;
      bool have = false;
      while (body.Next(&bf, &bw)) {
        if (bf == 1 && bw == 2) { GDV_RETURN_NOT_OK(DecodeType(body.Sub(), &t)); have = true; }
        else body.Skip(bw);
      }
      if (!body.ok || !have) return Malformed("NullNode");
      Literal l;
      l.is_null = true;
      *out = std::make_shared<LiteralNode>(t, l);
      return Status::OK();
    }
    case 12: case 13: case 14: case 15: case 16: case 17: case 18: {
      // IntNode / FloatNode / LongNode / BooleanNode / DoubleNode / StringNode / BinaryNode { value = 1 }
      Literal l;
      int bf, bw;
      while (body.Next(&bf, &bw)) {
        if (bf != 1) { body.Skip(bw); continue; }
        if ((f == 12 || f == 14 || f == 15) && bw == 0) {
          const uint64_t raw = body.Varint();
          l.lo = f == 12 ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(raw))) : f == 15 ? (raw != 0) : raw;
        } else if (f == 13 && bw == 5) {
          l.lo = body.Fixed32();
        } else if (f == 16 && bw == 1) {
          l.lo = body.Fixed64();
        } else if ((f == 17 || f == 18) && bw == 2) {
          l.bytes = body.Bytes();
        } else {
          return Malformed("literal node");
        }
      }
      if (!body.ok) return Malformed("literal node");
      const DataType t = f == 12 ? int32() : f == 13 ? float32() : f == 14 ? int64() : f == 15 ? boolean()
                         : f == 16 ? float64() : f == 17 ? utf8() : binary();
      *out = std::make_shared<LiteralNode>(t, l);
      return Status::OK();
    }
    case 19: {  // DecimalNode { value = 1 (digits), precision = 2, scale = 3 }
      std::string digits;
      int32_t precision = 0, scale = 0;
      int bf, bw;
      while (body.Next(&bf, &bw)) {
        if (bf == 1 && bw == 2) digits = body.Bytes();
        else if (bf == 2 && bw == 0) precision = static_cast<int32_t>(body.Varint());
        else if (bf == 3 && bw == 0) scale = static_cast<int32_t>(body.Varint());
        else body.Skip(bw);
      }
      Literal l;
      if (!body.ok || !ParseDecimalDigits(digits, &l)) return Malformed("DecimalNode");
      if (precision < 1 || precision > 38 || scale < 0 || scale > precision)
        return Status::Invalid("malformed protobuf message: DecimalNode precision / scale out of range");
      {  // the value must fit the declared precision
        unsigned __int128 mag = (static_cast<unsigned __int128>(l.hi) << 64) | l.lo;
        if (static_cast<__int128>(mag) < 0) mag = ~mag + 1;
        unsigned __int128 lim = 1;
        for (int k = 0; k < precision; k++) lim *= 10;
        if (mag >= lim) return Status::Invalid("malformed protobuf message: DecimalNode value exceeds its precision");
      }
      *out = std::make_shared<LiteralNode>(decimal128(precision, scale), l);
      return Status::OK();
    }
    case 21:
      return DecodeIn(body, depth, out);
    default:
      return Status::NotImplemented("TreeNode field " + std::to_string(f) + " is not known to the HIP backend");
  }
}

}  // namespace

Status DecodeSchema(const uint8_t* data, size_t size, Schema* out) {
  if (data == nullptr && size > 0) return Status::Invalid("null schema bytes");
  Reader r{data, data + size};
  out->clear();
  int f, w;
  while (r.Next(&f, &w)) {
    if (f == 1 && w == 2) {
      Field fld;
      GDV_RETURN_NOT_OK(DecodeField(r.Sub(), &fld));
      out->push_back(std::move(fld));
    } else {
      r.Skip(w);
    }
  }
  return r.ok ? Status::OK() : Malformed("Schema");
}

Status DecodeExpressionList(const uint8_t* data, size_t size, std::vector<ExpressionPtr>* out) {
  if (data == nullptr && size > 0) return Status::Invalid("null expression bytes");
  Reader r{data, data + size};
  out->clear();
  int f, w;
  while (r.Next(&f, &w)) {
    if (f != 2 || w != 2) { r.Skip(w); continue; }
    Reader er = r.Sub();  // ExpressionRoot { root = 1, resultType = 2 }
    NodePtr root;
    Field result;
    bool have_result = false;
    int ef, ew;
    while (er.Next(&ef, &ew)) {
      if (ef == 1 && ew == 2) GDV_RETURN_NOT_OK(DecodeNode(er.Sub(), 0, &root));
      else if (ef == 2 && ew == 2) { GDV_RETURN_NOT_OK(DecodeField(er.Sub(), &result)); have_result = true; }
      else er.Skip(ew);
    }
    if (!er.ok || !root || !have_result) return Malformed("ExpressionRoot");
    out->push_back(std::make_shared<Expression>(root, result));
  }
  return r.ok ? Status::OK() : Malformed("ExpressionList");
}

Status DecodeCondition(const uint8_t* data, size_t size, ExpressionPtr* out) {
  if (data == nullptr && size > 0) return Status::Invalid("null condition bytes");
  Reader r{data, data + size};
  NodePtr root;
  int f, w;
  while (r.Next(&f, &w)) {
    if (f == 1 && w == 2) GDV_RETURN_NOT_OK(DecodeNode(r.Sub(), 0, &root));
    else r.Skip(w);
  }
  if (!r.ok || !root) return Malformed("Condition");
  Field cond;
  cond.name = "cond";
  cond.type = boolean();
  *out = std::make_shared<Expression>(root, cond);
  return Status::OK();
}

}  // namespace gdv
