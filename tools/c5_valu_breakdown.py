"""Round 6, verdict item 3: where do the C5 kernels' instructions go?  Run under rocprofv3 --pmc (tools/c5_valu_breakdown.sh):
each variant of the C5 projection (single expressions and combinations) is evaluated a few times; the shell script then
sums SQ_INSTS_VALU / SALU / LDS / SQ_WAVES per kernel and this file's printed kernel names tie kernels to variants."""
import os, sys, re, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(os.environ.get("C5_ROWS", "100000000"))
db = W.c5_device_batch_philox(n)
ex = W.c5_expressions()
b = gandiva.TreeExprBuilder()
s = b.make_field(W.c5_schema().field(0))
ident = b.make_expression(s, pa.field("id", pa.string()))
length = b.make_expression(b.make_function("octet_length", [s], pa.int32()), pa.field("len", pa.int32()))
variants = {"like": [ex[0]], "substr": [ex[1]], "upper": [ex[2]], "identity": [ident], "octet_length": [length],
            "like+substr": ex[:2], "like+upper": [ex[0], ex[2]], "substr+upper": ex[1:], "all3": ex}
only = os.environ.get("C5_VARIANTS")
for name, exprs in variants.items():
    if only and name not in only.split(","):
        continue
    proj = gandiva.make_projector(W.c5_schema(), exprs, None)
    outs = proj.evaluate_device(db)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        proj.evaluate_device(db, outputs=outs)
    e.record()
    torch.cuda.synchronize()
    names = []
    for m in re.findall(r"extern \"C\" __global__ void [^\n]*?(gdv_k_[0-9a-f]{16})\(", proj.llvm_ir):
        if m not in names:
            names.append(m)
    print(f"VARIANT {name:14s} {a.elapsed_time(e) / 4:7.3f} ms  kernels {' '.join(names)}", flush=True)
