"""Why does the C2 kernel reach 0.93 of the shape-matched streaming ceiling on one box and 0.80 on another?
The ceiling kernel (4 input + 10 output streams, no arithmetic) on 1 GiB scratch streams, on 2 GiB scratch
streams and on the product's OWN buffers, next to the product kernel, on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gandiva_amd as gandiva
from gandiva_amd import workloads as W
rows = 1 << 28
db = W.c2_device_batch(rows)
proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
outs = proj.evaluate_device(db)
def product():
    for _ in range(3):
        proj.evaluate_device(db, outputs=outs, sync=False)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        proj.evaluate_device(db, outputs=outs, sync=False)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    return ms, W.C2_BYTES_PER_ROW * rows / ms / 1e6
for rnd in range(3):
    ms, gbs = product()
    print(f"round {rnd}: product {ms:.3f} ms = {gbs:.0f} GB/s", flush=True)
    print("   ceiling, 1 GiB scratch streams:", bench.stream_ceiling(4, 10, 1 << 30), flush=True)
    print("   ceiling, 2 GiB scratch streams:", bench.stream_ceiling(4, 10, 1 << 31), flush=True)
    print("   ceiling, the product's buffers:", bench.stream_ceiling_on([c.data for c in db.columns], [o.data for o in outs], rows), flush=True)
    outs = proj.evaluate_device(db)
