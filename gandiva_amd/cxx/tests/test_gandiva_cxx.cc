// C++ acceptance test of the drop-in gandiva:: API (what a C++ caller of the reference
// writes).  `--host-only`: tree building / ToString / validation (no GPU).  Without the
// flag it also runs the reference lineage's KATs end to end on the GPU:
//   if(a > b) a else b            (pyarrow/tests/test_gandiva.py:24-63)
//   less_than(a, 1000.0) filter   (:93-114)
//   filter -> UINT32 selection -> project with a null (:329-373)
#include <cstdio>
#include <cstring>
#include <iostream>

#include "arrow/api.h"
#include "gandiva/expression_registry.h"
#include "gandiva/filter.h"
#include "gandiva/projector.h"
#include "gandiva/tree_expr_builder.h"

using namespace gandiva;

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } \
  } while (0)
#define CHECK_OK(expr)                                                     \
  do {                                                                     \
    arrow::Status _s = (expr);                                             \
    if (!_s.ok()) { std::printf("FAIL %s:%d  %s -> %s\n", __FILE__, __LINE__, #expr, _s.ToString().c_str()); failures++; } \
  } while (0)

template <typename B, typename T>
std::shared_ptr<arrow::Array> MakeArr(const std::vector<T>& v, const std::vector<bool>& valid = {}) {
  B b;
  for (size_t i = 0; i < v.size(); i++) {
    if (!valid.empty() && !valid[i]) (void)b.AppendNull();
    else (void)b.Append(v[i]);
  }
  return b.Finish().ValueOrDie();
}

int main(int argc, char** argv) {
  const bool host_only = argc > 1 && !std::strcmp(argv[1], "--host-only");
  auto fa = arrow::field("a", arrow::int32()), fb = arrow::field("b", arrow::int32()),
       fc = arrow::field("c", arrow::int32());
  auto na = TreeExprBuilder::MakeField(fa), nb = TreeExprBuilder::MakeField(fb),
       nc = TreeExprBuilder::MakeField(fc);
  CHECK(na->return_type()->Equals(arrow::int32()));
  auto gt = TreeExprBuilder::MakeFunction("greater_than", {na, nb}, arrow::boolean());
  auto ifn = TreeExprBuilder::MakeIf(gt, na, nb, arrow::int32());
  CHECK(ifn->ToString() == "if (bool greater_than((int32) a, (int32) b)) { (int32) a } else { (int32) b }");
  CHECK(TreeExprBuilder::MakeLiteral(int64_t(2))->ToString() == "(const int64) 2");
  CHECK(TreeExprBuilder::MakeLiteral(2.0)->ToString().rfind("(const double) 2 raw(", 0) == 0);
  auto expr = TreeExprBuilder::MakeExpression(ifn, arrow::field("res", arrow::int32()));
  CHECK(expr->result()->type()->Equals(arrow::int32()));
  auto schema = arrow::schema({fa, fb, fc});
  CHECK(GetRegisteredFunctionSignatures().size() > 100);

  // validation errors keep the reference's status code (ExpressionValidationError = 41)
  {
    auto bad = TreeExprBuilder::MakeExpression(
        TreeExprBuilder::MakeFunction("no_such_fn", {na}, arrow::int32()), arrow::field("r", arrow::int32()));
    std::shared_ptr<Projector> p;
    arrow::Status s = Projector::Make(schema, {bad}, &p);
    CHECK(!s.ok());
    CHECK(host_only || s.code() == arrow::StatusCode::ExpressionValidationError);
  }
  if (host_only) {
    std::shared_ptr<Projector> p;
    arrow::Status s = Projector::Make(schema, {expr}, &p);
    // without a HIP device Make must fail loudly (no CPU fallback); with one it succeeds
    std::printf("Projector::Make on this host: %s\n", s.ToString().c_str());
    std::printf(failures ? "FAILED\n" : "OK (host-only)\n");
    return failures ? 1 : 0;
  }

  auto pool = arrow::default_memory_pool();
  {  // test_tree_exp_builder
    std::shared_ptr<Projector> p;
    CHECK_OK(Projector::Make(schema, {expr}, &p));
    CHECK(p->DumpIR().find("@expr_") != std::string::npos);
    auto batch = arrow::RecordBatch::Make(schema, 4, {MakeArr<arrow::Int32Builder, int32_t>({10, 12, -20, 5}),
                                                      MakeArr<arrow::Int32Builder, int32_t>({5, 15, 15, 17}),
                                                      MakeArr<arrow::Int32Builder, int32_t>({0, 0, 0, 0})});
    ArrayVector out;
    CHECK_OK(p->Evaluate(*batch, pool, &out));
    CHECK(out.size() == 1 && out[0]->Equals(MakeArr<arrow::Int32Builder, int32_t>({10, 15, 15, 17})));
  }
  {  // test_filter
    auto fx = arrow::field("x", arrow::float64());
    auto s2 = arrow::schema({fx});
    std::vector<double> v(10000);
    for (int i = 0; i < 10000; i++) v[i] = i;
    auto cond = TreeExprBuilder::MakeCondition(TreeExprBuilder::MakeFunction(
        "less_than", {TreeExprBuilder::MakeField(fx), TreeExprBuilder::MakeLiteral(1000.0)}, arrow::boolean()));
    std::shared_ptr<Filter> f;
    CHECK_OK(Filter::Make(s2, cond, &f));
    std::shared_ptr<SelectionVector> sel;
    CHECK_OK(SelectionVector::MakeInt32(10000, pool, &sel));
    auto batch = arrow::RecordBatch::Make(s2, 10000, {MakeArr<arrow::DoubleBuilder, double>(v)});
    CHECK_OK(f->Evaluate(*batch, sel));
    CHECK(sel->GetNumSlots() == 1000);
    auto arr = std::static_pointer_cast<arrow::UInt32Array>(sel->ToArray());
    bool ok = arr->length() == 1000;
    for (int i = 0; ok && i < 1000; i++) ok = arr->Value(i) == static_cast<uint32_t>(i);
    CHECK(ok);
    // caller-side contract: too small a selection vector is Invalid
    std::shared_ptr<SelectionVector> small;
    CHECK_OK(SelectionVector::MakeInt32(10, pool, &small));
    CHECK(f->Evaluate(*batch, small).IsInvalid());
  }
  {  // test_filter_project
    auto batch = arrow::RecordBatch::Make(
        schema, 6,
        {MakeArr<arrow::Int32Builder, int32_t>({10, 12, -20, 5, 21, 29}),
         MakeArr<arrow::Int32Builder, int32_t>({5, 15, 15, 17, 12, 3}),
         MakeArr<arrow::Int32Builder, int32_t>({1, 25, 11, 30, -21, 0}, {true, true, true, true, true, false})});
    auto fcond = TreeExprBuilder::MakeCondition(gt);
    auto pcond = TreeExprBuilder::MakeFunction("less_than", {nb, nc}, arrow::boolean());
    auto e = TreeExprBuilder::MakeExpression(TreeExprBuilder::MakeIf(pcond, nb, nc, arrow::int32()),
                                             arrow::field("res", arrow::int32()));
    std::shared_ptr<Filter> f;
    std::shared_ptr<Projector> p;
    CHECK_OK(Filter::Make(schema, fcond, &f));
    CHECK_OK(Projector::Make(schema, {e}, SelectionVector::MODE_UINT32,
                             ConfigurationBuilder::DefaultConfiguration(), &p));
    std::shared_ptr<SelectionVector> sel;
    CHECK_OK(SelectionVector::MakeInt32(6, pool, &sel));
    CHECK_OK(f->Evaluate(*batch, sel));
    ArrayVector out;
    CHECK_OK(p->Evaluate(*batch, sel.get(), pool, &out));
    CHECK(out.size() == 1 && out[0]->Equals(MakeArr<arrow::Int32Builder, int32_t>({1, -21, 0}, {true, true, false})));
  }
  {  // strings: like + upper through the C++ API (var-len output path)
    auto fs = arrow::field("s", arrow::utf8());
    auto s3 = arrow::schema({fs});
    auto ns = TreeExprBuilder::MakeField(fs);
    auto like = TreeExprBuilder::MakeExpression(
        TreeExprBuilder::MakeFunction("like", {ns, TreeExprBuilder::MakeStringLiteral("%spark%")}, arrow::boolean()),
        arrow::field("b", arrow::boolean()));
    auto up = TreeExprBuilder::MakeExpression(TreeExprBuilder::MakeFunction("upper", {ns}, arrow::utf8()),
                                              arrow::field("u", arrow::utf8()));
    std::shared_ptr<Projector> p;
    CHECK_OK(Projector::Make(s3, {like, up}, &p));
    auto batch = arrow::RecordBatch::Make(
        s3, 4, {MakeArr<arrow::StringBuilder, std::string>({"park", "sparkle", "bright spark and fire", "spark"})});
    ArrayVector out;
    CHECK_OK(p->Evaluate(*batch, pool, &out));
    CHECK(out.size() == 2 && out[0]->Equals(MakeArr<arrow::BooleanBuilder, bool>({false, true, true, true})));
    CHECK(out[1]->Equals(MakeArr<arrow::StringBuilder, std::string>({"PARK", "SPARKLE", "BRIGHT SPARK AND FIRE", "SPARK"})));
  }
  std::printf(failures ? "FAILED\n" : "OK\n");
  return failures ? 1 : 0;
}
