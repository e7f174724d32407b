"""Run one variant of the C5 string workload N times (device-resident) — for rocprofv3
kernel traces:  python tools/c5_variant.py identity|like|substr|upper|all3 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

which = sys.argv[1] if len(sys.argv) > 1 else "all3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(os.environ.get("C5_ROWS", 100_000_000))
db = W.c5_device_batch(n)
ex = W.c5_expressions()
b = gandiva.TreeExprBuilder()
s = b.make_field(W.c5_schema().field(0))
variants = {"like": [ex[0]], "substr": [ex[1]], "upper": [ex[2]],
            "identity": [b.make_expression(s, pa.field("id", pa.string()))], "all3": ex}
proj = gandiva.make_projector(W.c5_schema(), variants[which], None)
outs = proj.evaluate_device(db)
for _ in range(reps):
    proj.evaluate_device(db, outputs=outs)
torch.cuda.synchronize()
