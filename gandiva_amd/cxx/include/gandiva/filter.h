// gandiva/filter.h (pyarrow/includes/libgandiva.pxd:242-256).
#pragma once
#include "gandiva/arrow.h"
#include "gandiva/condition.h"
#include "gandiva/configuration.h"
#include "gandiva/selection_vector.h"

struct gdv_filter;

namespace gandiva {

class Filter {
 public:
  ~Filter();
  static Status Make(SchemaPtr schema, ConditionPtr condition, std::shared_ptr<Filter>* filter);
  static Status Make(SchemaPtr schema, ConditionPtr condition,
                     std::shared_ptr<Configuration> configuration, std::shared_ptr<Filter>* filter);
  // out_selection is allocated by the caller (max slots >= batch.num_rows()), filled here
  Status Evaluate(const arrow::RecordBatch& batch,
                  std::shared_ptr<SelectionVector> out_selection);
  std::string DumpIR();

 private:
  friend class ShardedFilter;
  Filter(gdv_filter* h, SchemaPtr schema) : handle_(h), schema_(std::move(schema)) {}
  gdv_filter* handle_;
  SchemaPtr schema_;
};

}  // namespace gandiva
