"""Round 5, verdict item 2: the lineage's canonical chain with a STRING projection —
Filter (predicate on an int64 column) -> SelectionVector -> Projector(upper(s) | substr(s, 2, 5)) in selection mode —
at 10^8 rows (C5's column), selectivity 1/8 and 1/2.  Device-resident, HIP events, per-stage and whole-chain times.

  python tools/filter_string_chain.py [rows]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
sbatch = W.c5_device_batch(n)
g = torch.Generator(device="cuda")
g.manual_seed(31)
key = torch.empty(n, dtype=torch.int64, device="cuda").random_(0, 1000, generator=g)
schema = pa.schema([pa.field("s", pa.string()), pa.field("k", pa.int64())])
db = gandiva.DeviceBatch(schema, [sbatch.columns[0], gandiva.DeviceColumn(pa.int64(), n, None, key.view(torch.uint8))], n)
b = gandiva.TreeExprBuilder()
fs, fk = b.make_field(schema.field(0)), b.make_field(schema.field(1))
exprs = {"upper(s)": [b.make_expression(b.make_function("upper", [fs], pa.string()), pa.field("u", pa.string()))],
         "substr(s, 2, 5)": [b.make_expression(b.make_function("substr", [fs, b.make_literal(2, pa.int64()), b.make_literal(5, pa.int64())],
                                                               pa.string()), pa.field("t", pa.string()))],
         "like + substr + upper (C5's three)": None}
idx = torch.empty(n, dtype=torch.int32, device="cuda")


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


print(f"# {n} rows; strings: C5's column (lengths 4..20); key: int64 U[0,1000)", flush=True)
for thr, label in ((874, "1/8"), (499, "1/2")):
    cond = b.make_condition(b.make_function("greater_than", [fk, b.make_literal(thr, pa.int64())], pa.bool_()))
    flt = gandiva.make_filter(schema, cond)
    sel = flt.evaluate_device(db, "int32", out=idx)
    torch.cuda.synchronize()
    k = sel.num_slots
    t_filter = timed(lambda: flt.evaluate_device(db, "int32", out=idx, sync=False))
    for name, ex in exprs.items():
        if ex is None:
            ex = W.c5_expressions()   # built over C5's schema: field "s" resolves by name here as well
        proj = gandiva.make_projector(schema, ex, pa.default_memory_pool(), selection_mode="UINT32")
        state = {"o": None}

        def run():
            state["o"] = proj.evaluate_device(db, selection=sel, outputs=state["o"], sync=False)
        try:
            t_proj = timed(run)
        except Exception as err:   # noqa: BLE001
            print(f"selectivity {label}: {name}: {type(err).__name__}: {err}", flush=True)
            continue
        print(f"selectivity {label} ({k} rows selected): filter {t_filter:.3f} ms + selection-mode {name} {t_proj:.3f} ms "
              f"= {t_filter + t_proj:.3f} ms per chain   [path hint {proj.path_hint}]", flush=True)
