// gandiva/arrow.h — Arrow aliases used across the gandiva:: API
// (pyarrow/includes/libgandiva.pxd:105-108 pins `gandiva::ArrayVector`).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "arrow/api.h"

namespace gandiva {
using Status = arrow::Status;
using ArrayPtr = std::shared_ptr<arrow::Array>;
using DataTypePtr = std::shared_ptr<arrow::DataType>;
using DataTypeVector = std::vector<DataTypePtr>;
using FieldPtr = std::shared_ptr<arrow::Field>;
using FieldVector = std::vector<FieldPtr>;
using SchemaPtr = std::shared_ptr<arrow::Schema>;
using ArrayVector = std::vector<ArrayPtr>;
using ArrayDataPtr = std::shared_ptr<arrow::ArrayData>;
using ArrayDataVector = std::vector<ArrayDataPtr>;
}  // namespace gandiva
