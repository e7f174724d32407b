"""Filter -> SelectionVector -> Projector chain (pyarrow/tests/test_gandiva.py:329-373 shape) at
C3 scale, device-resident: time of each stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
db = W.c3_device_batch(n)
flt = gandiva.make_filter(W.c3_schema(), W.c3_condition())
b = gandiva.TreeExprBuilder()
a, c = (b.make_field(W.c3_schema().field(i)) for i in range(2))
expr = b.make_expression(b.make_function("add", [a, c], pa.int64()), pa.field("s", pa.int64()))
proj = gandiva.make_projector(W.c3_schema(), [expr], None, "UINT32")
out = torch.empty(n, dtype=torch.int32, device="cuda")
sel = flt.evaluate_device(db, "int32", out=out)
outs = proj.evaluate_device(db, selection=sel)
torch.cuda.synchronize()
for name, fn in (("filter", lambda: flt.evaluate_device(db, "int32", out=out)),
                 ("project(selection)", lambda: proj.evaluate_device(db, selection=sel, outputs=outs))):
    t = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    print(f"{name:20s} {ms:7.3f} ms   selected {sel.num_slots} of {n}")
k = sel.num_slots
want = (db.columns[0].data.view(torch.int64) + db.columns[1].data.view(torch.int64))[out[:k].long()]
assert torch.equal(outs[0].data[:8 * k].view(torch.int64), want)
print("gather result verified against torch")

# round 3: the same chain with the slot count left on the device (gdv_filter_evaluate_async +
# gdv_projector_evaluate_selected): no hipStreamSynchronize between filter and projector
outs_cap = None
def chain_async():
    global outs_cap
    s = flt.evaluate_device(db, "int32", out=out, sync=False)
    outs_cap = proj.evaluate_device(db, selection=s, outputs=outs_cap, sync=False)
    return s
s = chain_async()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    s = chain_async()
torch.cuda.synchronize()
ms_async = (time.perf_counter() - t) / 5 * 1e3
def chain_sync():
    s = flt.evaluate_device(db, "int32", out=out)
    proj.evaluate_device(db, selection=s, outputs=outs)
chain_sync()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    chain_sync()
torch.cuda.synchronize()
ms_sync = (time.perf_counter() - t) / 5 * 1e3
print(f"filter -> project chain: {ms_sync:7.3f} ms with the count read back between the two calls, "
      f"{ms_async:7.3f} ms with the count left in HBM (zero host synchronisations inside the chain)")
assert s.num_slots == k
assert torch.equal(outs_cap[0].data[:8 * k].view(torch.int64), want)
print("asynchronous chain verified against torch")

# round 4: the same result from ONE kernel (gdv_filter_project_*: predicate + look-back + compacted projection)
fp = gandiva.make_filter_project(W.c3_schema(), W.c3_condition(), [expr], "int32")
assert fp.fused
fo, fsel = fp.evaluate_device(db, indices=out)
torch.cuda.synchronize()
assert fsel.num_slots == k
assert torch.equal(fo[0].data[:8 * k].view(torch.int64), want)
for label, sync in (("synchronous", True), ("asynchronous", False)):
    t = time.perf_counter()
    for _ in range(5):
        fo, fsel = fp.evaluate_device(db, outputs=fo, indices=out, sync=sync)
    torch.cuda.synchronize()
    print(f"fused filter-project ({label}): {(time.perf_counter() - t) / 5 * 1e3:7.3f} ms   (chain above: {ms_sync:.3f} / {ms_async:.3f} ms)")
fp0 = gandiva.make_filter_project(W.c3_schema(), W.c3_condition(), [expr], None)
fo0, _ = fp0.evaluate_device(db)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    fo0, _ = fp0.evaluate_device(db, outputs=fo0, sync=False)
torch.cuda.synchronize()
print(f"fused filter-project without the selection vector: {(time.perf_counter() - t) / 5 * 1e3:7.3f} ms")
assert torch.equal(fo0[0].data[:8 * k].view(torch.int64), want)
print("fused results verified against torch")
