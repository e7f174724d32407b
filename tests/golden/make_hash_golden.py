"""Generates tests/golden/hash64_f64.json from the CPU oracle (run once; committed so the
oracle and the HIP library cannot drift together unnoticed)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pyarrow as pa
import gandiva_amd as gandiva
from oracle import oracle
b = gandiva.TreeExprBuilder()
d = pa.array([0.0, 1.0, -1.0, 42.0], type=pa.float64())
batch = pa.RecordBatch.from_arrays([d], names=["d"])
f = b.make_field(batch.schema.field(0))
h = oracle.project_one(b.make_function("hash64", [f], pa.int64()), pa.int64(), batch).to_pylist()
h32 = oracle.project_one(b.make_function("hash32", [f], pa.int32()), pa.int32(), batch).to_pylist()
json.dump({"hash64_of_0_1_-1_42": h, "hash32_of_0_1_-1_42": h32},
          open(os.path.join(os.path.dirname(__file__), "hash64_f64.json"), "w"), indent=1)
print(h, h32)
