"""PCIe-inclusive rate of the host-buffer path (GDV_MEM_HOST) for C2 — reported in DESIGN.md,
never the bench value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W
n = 1 << 24
batch = W.c2_batch(n)
proj = gandiva.make_projector(batch.schema, W.c2_expressions(), None)
proj.evaluate(batch)
t = time.perf_counter(); reps = 3
for _ in range(reps):
    proj.evaluate(batch)
el = (time.perf_counter() - t) / reps
print(f"host path C2 {n} rows: {el*1e3:.1f} ms/evaluate, {n/el/1e6:.1f} M rows/s, {113.75*n/el/1e9:.1f} GB/s moved")
