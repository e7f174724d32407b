"""Generates tests/golden/c{1..5}.arrow: the BASELINE configs at 1000 rows (inputs + expected
outputs in one Arrow IPC file each).

There is no reference implementation to generate vectors from, so the
expected outputs come from the CPU oracle and are accepted into the fixture ONLY where an
independent engine reproduces them: pyarrow.compute for C1, C2, C3, C5 and the first C4
output, Python's decimal module for the second C4 output.  Once committed, the fixtures pin
BOTH the oracle and the HIP path (tests/test_golden.py): neither can drift silently.
Run from the repo root:  python tests/golden/make_golden.py
"""
import decimal
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.compute as pc  # noqa: E402
from gandiva_amd import workloads as W  # noqa: E402
from oracle import oracle  # noqa: E402

N = 1000
HERE = os.path.dirname(os.path.abspath(__file__))


def write(name, batch, outputs, out_names):
    cols = list(batch.columns) + list(outputs)
    names = [f"in_{n}" for n in batch.schema.names] + [f"out_{n}" for n in out_names]
    t = pa.Table.from_arrays(cols, names=names)
    with pa.OSFile(os.path.join(HERE, name + ".arrow"), "wb") as f, pa.ipc.new_file(f, t.schema) as w:
        w.write_table(t)
    print(name, t.num_rows, "rows", t.num_columns, "columns")


def same(a, b):
    if pa.types.is_floating(a.type):
        va, vb = a.to_numpy(zero_copy_only=False), b.to_numpy(zero_copy_only=False)
        return a.is_valid().equals(b.is_valid()) and np.array_equal(np.nan_to_num(va), np.nan_to_num(vb))
    return a.equals(b)


# C1
b1 = W.c1_batch(N)
o1 = oracle.project(W.c1_expressions(), b1)
a, b, c = b1.columns
assert same(o1[0], pc.multiply(pc.add(a, b), c))
write("c1", b1, o1, ["r"])

# C2
b2 = W.c2_batch(N)
o2 = oracle.project(W.c2_expressions(), b2)
a, b, c, d = b2.columns
ind = [pc.add(a, b), pc.subtract(a, b), pc.multiply(a, b), pc.add(c, d), pc.multiply(c, d),
       pc.multiply(pc.add(a, b), c), pc.multiply(pc.subtract(a, b), d),
       pc.add(pc.multiply(a, b), pc.multiply(c, d)), pc.multiply(pc.add(a, b), pc.subtract(c, d)),
       pc.multiply(pc.multiply(pc.multiply(a, b), c), d)]
for x, y in zip(o2, ind):
    assert same(x, y)
write("c2", b2, o2, [f"e{i}" for i in range(10)])

# C3 (10 % nulls): selection vector
b3 = W.c3_batch(N, 0.1)
sel = oracle.filter_indices(W.c3_condition(), b3, "int32")
a, b = b3.columns
mask = pc.fill_null(pc.and_kleene(pc.greater(a, W.C3_K1), pc.less(b, W.C3_K2)), False)
assert sel.equals(pc.indices_nonzero(mask).cast(pa.uint32()))
padded = pa.concat_arrays([sel, pa.nulls(N - len(sel), pa.uint32())])
write("c3", b3, [padded], ["selection_padded_with_nulls"])

# C4
b4 = W.c4_batch(N, 0.1)
o4 = oracle.project(W.c4_expressions(), b4)
ep, disc, tax, ship = b4.columns
one = pa.scalar(decimal.Decimal("1.00"), pa.decimal128(15, 2))
assert o4[0].equals(pc.multiply(ep, pc.subtract(one, disc)).cast(o4[0].type))
ctx = decimal.Context(prec=80, rounding=decimal.ROUND_HALF_UP)
want = []
for e, dd, t in zip(ep.to_pylist(), disc.to_pylist(), tax.to_pylist()):
    if e is None or dd is None or t is None:
        want.append(None)
        continue
    v = ctx.multiply(ctx.multiply(e, ctx.subtract(decimal.Decimal(1), dd)), ctx.add(decimal.Decimal(1), t))
    want.append(v.quantize(decimal.Decimal("0.000001"), rounding=decimal.ROUND_HALF_UP, context=ctx))
assert o4[1].to_pylist() == want
assert o4[2].equals(pc.subtract(pa.scalar(W.C4_DATE_1998_12_01, pa.int32()), ship.cast(pa.int32())))
write("c4", b4, o4, ["disc_price", "charge", "days"])

# C5
b5 = W.c5_batch(N, 0.1)
o5 = oracle.project(W.c5_expressions(), b5)
s = b5.column(0)
assert o5[0].equals(pc.match_like(s, "%spark%"))
assert o5[1].equals(pc.utf8_slice_codeunits(s, 1, 6))
assert o5[2].equals(pc.utf8_upper(s))
write("c5", b5, o5, ["is_spark", "sub", "up"])
