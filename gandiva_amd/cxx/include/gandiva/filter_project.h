// gandiva/filter_project.h — NOT part of the reference's API: an addition of this backend (round 4).
// The reference's callers chain  Filter::Evaluate -> SelectionVector -> Projector::Evaluate(batch, sv, ...)
// (pyarrow/tests/test_gandiva.py:329-373); that reads the predicate's columns twice.  FilterProject is
// the same computation as ONE operator: a single kernel evaluates the condition and writes the
// projections of the selected rows, compacted (include/gandiva_amd.h, gdv_filter_project_*).  Plans the
// fused kernel does not take (var-len columns / outputs) run the chain internally — same results.
#pragma once
#include "gandiva/arrow.h"
#include "gandiva/condition.h"
#include "gandiva/configuration.h"
#include "gandiva/filter.h"
#include "gandiva/projector.h"
#include "gandiva/selection_vector.h"

struct gdv_filter_project;

namespace gandiva {

class FilterProject {
 public:
  ~FilterProject();
  // selection_vector_mode: MODE_NONE = only the projected columns; MODE_UINT16/32/64 = Evaluate also
  // fills a selection vector of that type
  static Status Make(SchemaPtr schema, ConditionPtr condition, const ExpressionVector& exprs,
                     SelectionVector::Mode selection_vector_mode, std::shared_ptr<Configuration> configuration,
                     std::shared_ptr<FilterProject>* out);
  // Appends one array per expression, each holding the selected rows only; out_selection (may be
  // null with MODE_NONE; else allocated by the caller, max slots >= batch.num_rows()) is filled.
  Status Evaluate(const arrow::RecordBatch& batch, arrow::MemoryPool* pool, ArrayVector* output,
                  std::shared_ptr<SelectionVector> out_selection = nullptr) const;
  bool fused() const { return handle_ != nullptr; }
  std::string DumpIR();

 private:
  FilterProject() = default;
  gdv_filter_project* handle_ = nullptr;
  std::shared_ptr<Filter> filter_;        // the chain, when the plan is not fused
  std::shared_ptr<Projector> projector_;
  SchemaPtr schema_;
  FieldVector output_fields_;
  SelectionVector::Mode mode_ = SelectionVector::MODE_NONE;
};

}  // namespace gandiva
