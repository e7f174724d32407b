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
  // Caller-allocated outputs — what the JNI layer and any caller that keeps its result buffers
  // (HBM-resident ones included) use.  [M]: restated from the lineage's projector.h as recalled; the
  // .pxd does not bind these overloads.  output[i] is the ArrayData of expression i: type = the
  // expression's result type, length = the number of output rows, buffers = {validity, data} for
  // fixed-width types (validity >= ceil(rows / 8) bytes — 8 * ceil(rows / 64) for device buffers —,
  // data >= rows * width) and {validity, offsets, data} for utf8 / binary (offsets >= (rows + 1) * 4
  // bytes; the data buffer's size is the capacity: too small -> Invalid naming the bytes needed).
  // All buffers must be mutable and live in the same memory domain as the batch.
  Status Evaluate(const arrow::RecordBatch& batch, const ArrayDataVector& output) const;
  Status Evaluate(const arrow::RecordBatch& batch, const SelectionVector* selection_vector,
                  const ArrayDataVector& output) const;
  std::string DumpIR();

 private:
  friend class ShardedProjector;
  Projector(gdv_projector* h, SchemaPtr schema, FieldVector outs, SelectionVector::Mode mode)
      : handle_(h), schema_(std::move(schema)), output_fields_(std::move(outs)), mode_(mode) {}
  gdv_projector* handle_;
  SchemaPtr schema_;
  FieldVector output_fields_;
  SelectionVector::Mode mode_;
};

}  // namespace gandiva
