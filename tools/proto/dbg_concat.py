import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, pyarrow as pa
import gandiva_amd as gandiva
from oracle import oracle
import test_strings as T
n = 1000
batch = T._position_batch(n, n + 5)
b = gandiva.TreeExprBuilder()
s, t = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
exprs = T._concat_exprs(b, s, t)
def run(ex, tag):
    got = gandiva.make_projector(batch.schema, ex, None).evaluate(batch)
    want = oracle.project(ex, batch)
    for g, w, e in zip(got, want, ex):
        gl, wl = g.cast(pa.binary()).to_pylist(), w.cast(pa.binary()).to_pylist()
        bad = [i for i in range(n) if gl[i] != wl[i]]
        print(tag, e.result().name, "bad rows:", len(bad), bad[:6])
        for i in bad[:3]:
            print("   row", i, "s=", batch.column(0)[i].as_py(), "t=", batch.column(1)[i].as_py(), "got", gl[i], "want", wl[i])
run(exprs, "all6")
run(exprs[:1], "only concat2")
run(exprs[:3], "first3")
run([exprs[0], exprs[3]], "concat2+pipes3")
