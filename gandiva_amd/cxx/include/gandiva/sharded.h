// gandiva/sharded.h — NOT part of the reference's API: an addition of this backend (round 6).
// The reference's Projector::Evaluate / Filter::Evaluate are ONE call per batch (pyarrow/includes/libgandiva.pxd:218-226,
// 246-248); on a node with several GPUs these two classes keep it one call: the batch is cut into row ranges on 1024-row
// bounds (gdv_shard_bounds), every range is evaluated on a GPU of its own by a host thread of its own inside the library,
// nothing is exchanged between the ranges (include/gandiva_amd.h, gdv_*_evaluate_sharded / _host_sharded).
#pragma once
#include <vector>

#include "gandiva/arrow.h"
#include "gandiva/filter.h"
#include "gandiva/projector.h"
#include "gandiva/selection_vector.h"

namespace gandiva {

// rows [*lo, *hi) of shard `shard` of `num_shards` over `num_rows` rows
void ShardBounds(int64_t num_rows, int num_shards, int shard, int64_t* lo, int64_t* hi);
// devices the library can be pointed at (gdv_device_count)
int DeviceCount();

class ShardedProjector {
 public:
  // devices: the library's device numbering (gdv_set_device); empty = every device of the node
  static Status Make(SchemaPtr schema, const ExpressionVector& exprs, std::vector<int> devices,
                     std::shared_ptr<Configuration> configuration, std::shared_ptr<ShardedProjector>* out);
  // ONE host-resident batch: sliced by the library, every slice staged through its own GPU, ONE array per expression
  // appended to `output` (allocated from `pool`).  Plans with utf8 / binary outputs run on devices[0] alone.
  Status Evaluate(const arrow::RecordBatch& batch, arrow::MemoryPool* pool, ArrayVector* output) const;
  // Device-resident shards: shards[s] holds rows [lo_s, hi_s) of the logical batch in the memory of devices[s]
  // (non-CPU arrow::Buffers); (*outputs)[s] receives shard s's arrays, allocated from that shard's MemoryManager.
  Status Evaluate(const std::vector<std::shared_ptr<arrow::RecordBatch>>& shards, std::vector<ArrayVector>* outputs) const;
  const std::vector<int>& devices() const { return devices_; }
  const std::shared_ptr<Projector>& projector() const { return projector_; }

 private:
  std::shared_ptr<Projector> projector_;
  std::vector<int> devices_;
};

class ShardedFilter {
 public:
  static Status Make(SchemaPtr schema, ConditionPtr condition, std::vector<int> devices,
                     std::shared_ptr<Configuration> configuration, std::shared_ptr<ShardedFilter>* out);
  // ONE host-resident batch -> the global, ascending selection vector (host buffer, max slots >= batch.num_rows())
  Status Evaluate(const arrow::RecordBatch& batch, std::shared_ptr<SelectionVector> out_selection) const;
  // Device-resident shards: out_selections[s] (device buffer on devices[s], max slots >= the shard's rows) receives
  // lo_s + local positions — the vectors concatenate, in shard order, into the globally ascending one; *total = their sum
  Status Evaluate(const std::vector<std::shared_ptr<arrow::RecordBatch>>& shards,
                  const std::vector<std::shared_ptr<SelectionVector>>& out_selections, int64_t* total) const;
  const std::vector<int>& devices() const { return devices_; }

 private:
  std::shared_ptr<Filter> filter_;
  std::vector<int> devices_;
};

}  // namespace gandiva
