"""Round 6: does the placement of the INPUT columns matter too?  C2 with its outputs on a fixed spread set; the four inputs copied
onto (a) four neighbours in allocation order, (b) four buffers spread over the pool, several picks each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

rows = 1 << 28
db = W.c2_device_batch(rows)
proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
outs = proj.evaluate_device(db)
valid = [o.validity for o in outs]
del outs
torch.cuda.empty_cache()
npool = 56
bufs = [torch.empty(rows * 8, dtype=torch.uint8, device="cuda") for _ in range(npool)]
out_idx = list(range(0, 40, 4))                     # a spread set (every 4th of the first 40)
cols_out = [gandiva.DeviceColumn(t, rows, valid[e], bufs[i]) for e, (i, t) in enumerate(zip(out_idx, proj._out_types))]


def kernel_ms(in_idx):
    if in_idx is None:
        b = db
    else:
        cols = []
        for c, i in zip(db.columns, in_idx):
            bufs[i].copy_(c.data)
            cols.append(gandiva.DeviceColumn(c.type, rows, c.validity, bufs[i]))
        b = gandiva.DeviceBatch(db.schema, cols, rows)
    for _ in range(3):
        proj.evaluate_device(b, outputs=cols_out, sync=False)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        proj.evaluate_device(b, outputs=cols_out, sync=False)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / 4


free = [i for i in range(npool) if i not in out_idx]
print("outputs on", out_idx)
print(f"inputs where torch put them (four consecutive allocations made BEFORE the pool's 56): kernel {kernel_ms(None):.3f} ms")
for label, picks in (("neighbours", [free[k:k + 4] for k in (0, 8, 20, 30, 36)]),
                     ("spread", [[free[k], free[k + 11], free[k + 22], free[k + 33]] for k in (0, 2, 4, 6, 8)]),
                     ("beyond the outputs' span", [[41 + k, 45 + k, 49 + k, 53 + k] for k in (0, 1, 2)] + [[44, 45, 46, 47], [52, 53, 54, 55]])):
    for p in picks:
        print(f"inputs on {label:26s} {p}: kernel {kernel_ms(p):.3f} ms", flush=True)
print(f"inputs where torch put them, again: kernel {kernel_ms(None):.3f} ms")
