// gandiva/projector.h (pyarrow/includes/libgandiva.pxd:214-240).  Evaluate launches the fused
// HIP kernel; batches whose buffers are HBM-resident (arrow::Buffer::is_cpu() == false) are
// used in place and the outputs are allocated from the same arrow::MemoryManager.
#pragma once
#include "gandiva/arrow.h"
#include "gandiva/configuration.h"
#include "gandiva/node.h"
#include "gandiva/selection_vector.h"

struct gdv_projector;

namespace gandiva {

class Projector {
 public:
  ~Projector();
  static Status Make(SchemaPtr schema, const ExpressionVector& exprs,
                     std::shared_ptr<Projector>* projector);
  static Status Make(SchemaPtr schema, const ExpressionVector& exprs,
                     std::shared_ptr<Configuration> configuration,
                     std::shared_ptr<Projector>* projector);
  static Status Make(SchemaPtr schema, const ExpressionVector& exprs,
                     SelectionVector::Mode selection_vector_mode,
                     std::shared_ptr<Configuration> configuration,
                     std::shared_ptr<Projector>* projector);

  Status Evaluate(const arrow::RecordBatch& batch, arrow::MemoryPool* pool,
                  ArrayVector* output) const;
  Status Evaluate(const arrow::RecordBatch& batch, const SelectionVector* selection_vector,
                  arrow::MemoryPool* pool, ArrayVector* output) const;
  std::string DumpIR();

 private:
  Projector(gdv_projector* h, SchemaPtr schema, FieldVector outs, SelectionVector::Mode mode)
      : handle_(h), schema_(std::move(schema)), output_fields_(std::move(outs)), mode_(mode) {}
  gdv_projector* handle_;
  SchemaPtr schema_;
  FieldVector output_fields_;
  SelectionVector::Mode mode_;
};

}  // namespace gandiva
