"""Device time per Evaluate of the round-2 registry-tail functions on C5's column (utf8, lengths
4..20) — HIP events around evaluate_device, median of 7, HBM-resident inputs and outputs."""
import sys
import pyarrow as pa, torch
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
db = W.c5_device_batch(n)
sch = W.c5_schema()
b = gandiva.TreeExprBuilder()
s = b.make_field(sch.field(0))
STR, I32, I64 = pa.string(), pa.int32(), pa.int64()
lit = lambda v, t=STR: b.make_literal(v, t)
ex = lambda name, node, t=STR: [b.make_expression(node, pa.field(name, t))]
plans = {
    "reverse(s)": ex("r", b.make_function("reverse", [s], STR)),
    "lpad(s, 16, '*')": ex("l", b.make_function("lpad", [s, lit(16, I32), lit("*")], STR)),
    "replace(s, 'spark', 'flink')": ex("p", b.make_function("replace", [s, lit("spark"), lit("flink")], STR)),
    "castVARCHAR(char_length(s), 10)": ex("c", b.make_function("castVARCHAR", [b.make_function("castBIGINT", [b.make_function("char_length", [s], I32)], I64), lit(10, I64)], STR)),
    "upper(concat(s, '-', s))  [two stages]": ex("u", b.make_function("upper", [b.make_function("concat", [s, lit("-"), s], STR)], STR)),
    "locate('spark', s)": ex("k", b.make_function("locate", [lit("spark"), s], I32), I32),
}
for name, e in plans.items():
    p = gandiva.make_projector(sch, e, None)
    p.evaluate_device(db); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); outs = p.evaluate_device(db); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name}: n={n} median {ts[3]:.3f} ms ({n / ts[3] / 1e6:.1f} G rows/s)")

# the two-stage plan again, asynchronously (gdv_projector_evaluate_async: both stages + the gate on the stream,
# nothing read back): 8 evaluations back to back between two events
name = "upper(concat(s, '-', s))  [two stages]"
p = gandiva.make_projector(sch, plans[name], None)
outs = p.evaluate_device(db)
cap = int(outs[0].data.numel())
outs, res = p.evaluate_device_async(db, capacity_bytes=cap)
torch.cuda.synchronize()
assert int(res[0]) == 0, int(res[0])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(8):          # (8 evaluations in flight hold 8 sets of temporaries: the first round allocates them)
    outs, res = p.evaluate_device_async(db, outputs=outs)
torch.cuda.synchronize()
e0.record()
for _ in range(8):
    outs, res = p.evaluate_device_async(db, outputs=outs)
e1.record()
torch.cuda.synchronize()
assert int(res[0]) == 0
print(f"{name}, asynchronous: n={n} {e0.elapsed_time(e1) / 8:.3f} ms per Evaluate, 8 back to back")
e0.record()
for _ in range(8):
    p.evaluate_device(db, outputs=None)
e1.record()
torch.cuda.synchronize()
print(f"{name}, synchronous:  n={n} {e0.elapsed_time(e1) / 8:.3f} ms per Evaluate, 8 back to back")
