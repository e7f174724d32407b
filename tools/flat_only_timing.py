"""Times single-output flat string plans (upper / lower / the column itself) at C5's batch:
with every var-len output flat the optimistic variant has no scanner hand-off at all."""
import sys, time
import pyarrow as pa, torch
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
db = W.c5_device_batch(n)
sch = W.c5_schema()
b = gandiva.TreeExprBuilder()
f = b.make_field(sch.field(0))
plans = {
    "upper": [b.make_expression(b.make_function("upper", [f], pa.string()), pa.field("u", pa.string()))],
    "upper+lower": [b.make_expression(b.make_function("upper", [f], pa.string()), pa.field("u", pa.string())),
                    b.make_expression(b.make_function("lower", [f], pa.string()), pa.field("l", pa.string()))],
    "c5": W.c5_expressions(),
}
off = db.columns[0].offsets.view(torch.int32)
total = int(off[n]) - int(off[0])
for name, ex in plans.items():
    p = gandiva.make_projector(sch, ex, None)
    outs = p.evaluate_device(db)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); outs = p.evaluate_device(db, outs) if False else p.evaluate_device(db); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name}: n={n} bytes={total} median {ts[len(ts)//2]:.3f} ms min {ts[0]:.3f} ms")
