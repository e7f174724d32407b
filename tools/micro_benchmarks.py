"""The reference's own micro-benchmarks (upstream cpp/integ/micro_benchmarks.cc: TimedTestAdd3,
TimedTestBigNested, TimedTestExtractYear, TimedTestFilterAdd2, TimedTestFilterLike,
TimedTestInExpr, TimedTestMultiOr — shapes recalled; the reference printed
elapsed milliseconds and published none), run against this backend the way the reference
runs them: batches of 1M rows pushed through ONE Projector/Filter.

Two columns per benchmark: host Arrow buffers in and out (what the reference's callers do;
PCIe-bound here) and HBM-resident batches (the path this library is built for).

    python tools/micro_benchmarks.py [batches=20]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa
import torch
import gandiva_amd as gandiva
from oracle import oracle
import bench

BATCH = 1 << 20
CORES = bench.effective_cores()
BENCH_TREE = {}
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(7)
b = gandiva.TreeExprBuilder()
I32, I64, BOOL, STR = pa.int32(), pa.int64(), pa.bool_(), pa.string()


def ints(t, lo, hi):
    return pa.array(rng.integers(lo, hi, BATCH), t)


def strings(words, extra=8):
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    out = []
    picks = rng.integers(0, len(words) * 3, BATCH)
    tails = rng.integers(0, 26, (BATCH, extra))
    for p, t in zip(picks, tails):
        s = "".join(letters[t])
        out.append(words[p] if p < len(words) else s)
    return pa.array(out, STR)


def run(name, schema, batch, projector=None, flt=None):
    dev = gandiva.DeviceBatch.from_arrow(batch)
    def host():
        return projector.evaluate(batch) if projector else flt.evaluate(batch, None, "int32")
    def resident():
        return projector.evaluate_device(dev, outputs=outs) if projector else flt.evaluate_device(dev, "int32", out=sel)
    outs = projector.evaluate_device(dev) if projector else None
    sel = torch.empty(BATCH, dtype=torch.int32, device="cuda") if flt else None
    res = []
    for fn in (host, resident):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(NB):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) * 1e3)
    rows = NB * BATCH
    # the CPU restatement on this host's usable cores, same batch, same expression
    exprs_or_cond = BENCH_TREE[name]
    def cpu():
        if projector:
            return oracle.project(exprs_or_cond, batch, threads=CORES)
        return oracle.filter_indices(exprs_or_cond, batch, "int32", threads=CORES)
    cpu()
    t = time.perf_counter()
    reps = max(2, NB // 4)
    for _ in range(reps):
        cpu()
    cpu_ms = (time.perf_counter() - t) * 1e3
    print(f"{name:22s} {res[0]:9.1f} ms {rows / res[0] / 1e3:8.1f} Mrows/s   {res[1]:9.2f} ms {rows / res[1] / 1e3:9.1f} Mrows/s"
          f"   {reps * BATCH / cpu_ms / 1e3:8.1f} Mrows/s")


print(f"# {NB} batches x {BATCH} rows per benchmark; one Make, then Evaluate per batch")
print(f"{'benchmark':22s} {'host buffers in/out':>24s}   {'HBM-resident batches':>26s}   {'CPU restatement, ' + str(CORES) + ' threads':>20s}")

# TimedTestAdd3: x + (y + z) over int64
sc = pa.schema([("x", I64), ("y", I64), ("z", I64)])
f = [b.make_field(x) for x in sc]
e = b.make_expression(b.make_function("add", [f[0], b.make_function("add", [f[1], f[2]], I64)], I64), pa.field("r", I64))
batch = pa.RecordBatch.from_arrays([ints(I64, 0, 1 << 40) for _ in range(3)], schema=sc)
BENCH_TREE["TimedTestAdd3"] = [e]
run("TimedTestAdd3", sc, batch, projector=gandiva.make_projector(sc, [e], None))

# TimedTestBigNested: if (a < 10) 10 else if (a < 20) 20 ... else 200
sc = pa.schema([("a", I32)])
a = b.make_field(sc.field(0))
node = b.make_literal(200, I32)
for top in range(190, 0, -10):
    node = b.make_if(b.make_function("less_than", [a, b.make_literal(top, I32)], BOOL), b.make_literal(top, I32), node, I32)
batch = pa.RecordBatch.from_arrays([ints(I32, 0, 210)], schema=sc)
BENCH_TREE["TimedTestBigNested"] = [b.make_expression(node, pa.field("r", I32))]
run("TimedTestBigNested", sc, batch, projector=gandiva.make_projector(sc, BENCH_TREE["TimedTestBigNested"], None))

# TimedTestExtractYear: extractYear(date64)
sc = pa.schema([("d", pa.date64())])
d = b.make_field(sc.field(0))
batch = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 20000, BATCH) * 86400000, pa.date64())], schema=sc)
e = b.make_expression(b.make_function("extractYear", [d], I64), pa.field("y", I64))
BENCH_TREE["TimedTestExtractYear"] = [e]
run("TimedTestExtractYear", sc, batch, projector=gandiva.make_projector(sc, [e], None))

# TimedTestFilterAdd2: filter (f0 + f1 < f2) over int32
sc = pa.schema([("f0", I32), ("f1", I32), ("f2", I32)])
f = [b.make_field(x) for x in sc]
cond = b.make_condition(b.make_function("less_than", [b.make_function("add", [f[0], f[1]], I32), f[2]], BOOL))
batch = pa.RecordBatch.from_arrays([ints(I32, 0, 1 << 20) for _ in range(3)], schema=sc)
BENCH_TREE["TimedTestFilterAdd2"] = cond
run("TimedTestFilterAdd2", sc, batch, flt=gandiva.make_filter(sc, cond))

# TimedTestFilterLike / InExpr / MultiOr over a utf8 column
words = ["yellow", "green", "blue", "red", "orange", "purple"]
sc = pa.schema([("s", STR)])
s = b.make_field(sc.field(0))
batch = pa.RecordBatch.from_arrays([strings(words)], schema=sc)
cond = b.make_condition(b.make_function("like", [s, b.make_literal("%yellow%", STR)], BOOL))
BENCH_TREE["TimedTestFilterLike"] = cond
run("TimedTestFilterLike", sc, batch, flt=gandiva.make_filter(sc, cond))
vals = words[:4] + ["magenta", "cyan"]
cond = b.make_condition(b.make_in_expression(s, vals, STR))
BENCH_TREE["TimedTestInExpr"] = cond
run("TimedTestInExpr", sc, batch, flt=gandiva.make_filter(sc, cond))
ors = b.make_or([b.make_function("equal", [s, b.make_literal(v, STR)], BOOL) for v in vals])
BENCH_TREE["TimedTestMultiOr"] = b.make_condition(ors)
run("TimedTestMultiOr", sc, batch, flt=gandiva.make_filter(sc, BENCH_TREE["TimedTestMultiOr"]))
