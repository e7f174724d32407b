"""What rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE is pointed at (tools/fused_fp_traffic.sh): three evaluations of the fused
filter -> project operator and three of the Filter + selection-mode Projector chain, 10^9 int64 rows x 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
db = W.c3_device_batch(n)
b = gandiva.TreeExprBuilder()
a, c = (b.make_field(W.c3_schema().field(i)) for i in range(2))
expr = b.make_expression(b.make_function("add", [a, c], pa.int64()), pa.field("s", pa.int64()))
idx = torch.empty(n, dtype=torch.int32, device="cuda")
fp = gandiva.make_filter_project(W.c3_schema(), W.c3_condition(), [expr], "int32")
flt = gandiva.make_filter(W.c3_schema(), W.c3_condition())
proj = gandiva.make_projector(W.c3_schema(), [expr], None, "UINT32")

outs = None
for _ in range(3):
    outs, _ = fp.evaluate_device(db, outputs=outs, indices=idx)
torch.cuda.synchronize()
pouts = None
for _ in range(3):
    sel = flt.evaluate_device(db, "int32", out=idx)
    pouts = proj.evaluate_device(db, selection=sel, outputs=pouts)
torch.cuda.synchronize()
print("selected", sel.num_slots)
