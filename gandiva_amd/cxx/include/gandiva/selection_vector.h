// gandiva/selection_vector.h — row positions selected by a Filter
// (pyarrow/includes/libgandiva.pxd:43-71; call order pyarrow/gandiva.pyx:261-280).
#pragma once
#include "gandiva/arrow.h"

namespace gandiva {

class SelectionVector {
 public:
  enum Mode : int { MODE_NONE = 0, MODE_UINT16 = 1, MODE_UINT32 = 2, MODE_UINT64 = 3, MODE_MAX = 3 };

  static Status MakeInt16(int64_t max_slots, arrow::MemoryPool* pool,
                          std::shared_ptr<SelectionVector>* selection_vector);
  static Status MakeInt32(int64_t max_slots, arrow::MemoryPool* pool,
                          std::shared_ptr<SelectionVector>* selection_vector);
  static Status MakeInt64(int64_t max_slots, arrow::MemoryPool* pool,
                          std::shared_ptr<SelectionVector>* selection_vector);
  // over a caller-owned (possibly HBM-resident) buffer
  static Status Make(Mode mode, int64_t max_slots, std::shared_ptr<arrow::Buffer> buffer,
                     std::shared_ptr<SelectionVector>* selection_vector);

  uint64_t GetIndex(int64_t index) const;
  void SetIndex(int64_t index, uint64_t value);
  int64_t GetMaxSlots() const { return max_slots_; }
  int64_t GetNumSlots() const { return num_slots_; }
  void SetNumSlots(int64_t n) { num_slots_ = n; }
  Mode GetMode() const { return mode_; }
  arrow::Buffer& GetBuffer() const { return *buffer_; }
  // UInt16Array / UInt32Array / UInt64Array over the first GetNumSlots() entries
  ArrayPtr ToArray() const;
  int index_bytes() const { return mode_ == MODE_UINT16 ? 2 : mode_ == MODE_UINT32 ? 4 : 8; }

 private:
  SelectionVector(Mode m, int64_t max_slots, std::shared_ptr<arrow::Buffer> b)
      : mode_(m), max_slots_(max_slots), buffer_(std::move(b)) {}
  Mode mode_;
  int64_t max_slots_;
  int64_t num_slots_ = 0;
  std::shared_ptr<arrow::Buffer> buffer_;
};

}  // namespace gandiva
