"""Fused filter -> project at C3 scale (10^9 int64 rows x 2, a > 499 AND b < 250, project a + b):
device ms per Evaluate with / without the selection vector, next to the filter alone.  The tile shape
can be forced through GDV_U / GDV_WAVES (read at Make)."""
import os, sys, time
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
out = torch.empty(n, dtype=torch.int32, device="cuda")


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


res = {}
if os.environ.get("GDV_U") is None and os.environ.get("GDV_WAVES") is None:
    flt = gandiva.make_filter(W.c3_schema(), W.c3_condition())
    res["filter alone"] = timed(lambda: flt.evaluate_device(db, "int32", out=out, sync=False))
for label, dtype in (("fused + selection vector", "int32"), ("fused, projection only", None)):
    fp = gandiva.make_filter_project(W.c3_schema(), W.c3_condition(), [expr], dtype)
    state = {"o": None}

    def run():
        state["o"], _ = fp.evaluate_device(db, outputs=state["o"], indices=out if dtype else None, sync=False)
    res[label] = timed(run)
shape = f"U={os.environ.get('GDV_U', 'auto')} W={os.environ.get('GDV_WAVES', 'auto')}"
print(f"{shape:16s} " + "   ".join(f"{k}: {v:6.3f} ms" for k, v in res.items()), flush=True)
