"""Ablation of the C5 string workload: time each expression alone (device-resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = 100_000_000
db = W.c5_device_batch_philox(n)
ex = W.c5_expressions()
b = gandiva.TreeExprBuilder()
s = b.make_field(W.c5_schema().field(0))
ident = b.make_expression(s, pa.field("id", pa.string()))
length = b.make_expression(b.make_function("octet_length", [s], pa.int32()), pa.field("len", pa.int32()))
variants = {"like": [ex[0]], "substr": [ex[1]], "upper": [ex[2]], "identity": [ident], "octet_length": [length],
            "like+substr": ex[:2], "all3": ex}
for name, exprs in variants.items():
    proj = gandiva.make_projector(W.c5_schema(), exprs, None)
    outs = proj.evaluate_device(db)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        proj.evaluate_device(db, outputs=outs)
    torch.cuda.synchronize()
    print(f"{name:14s} {(time.perf_counter() - t) / 5 * 1e3:7.3f} ms")
