"""The oracle's TREE semantics against an independent engine: random expression trees
(tests/test_fuzz_trees.py's generator, restricted to operators pyarrow.compute also has) are
evaluated by the CPU oracle and by a small interpreter over pyarrow.compute kernels —
wrapping integer arithmetic, IEEE float arithmetic, SQL comparisons with null propagation,
Kleene AND/OR, if/else taking the else branch on a null condition, null tests, casts.
CPU only; this is what pins the oracle's null-propagation through nested trees beyond the
nine reference KATs."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import test_fuzz_trees as F
from oracle import oracle
from helpers import assert_bit_exact

ARROW_OPS = {"add", "subtract", "multiply", "negative", "abs", "equal", "not_equal", "less_than",
             "less_than_or_equal_to", "greater_than", "greater_than_or_equal_to", "isnull", "isnotnull",
             "not", "castBIGINT", "castFLOAT8", "castFLOAT4", "bitwise_and", "bitwise_or", "bitwise_xor",
             "istrue", "isfalse", "isnottrue", "isnotfalse", "is_distinct_from", "is_not_distinct_from"}
WRAP = {"add": pc.add, "subtract": pc.subtract, "multiply": pc.multiply}
CMP = {"equal": pc.equal, "not_equal": pc.not_equal, "less_than": pc.less, "less_than_or_equal_to": pc.less_equal,
       "greater_than": pc.greater, "greater_than_or_equal_to": pc.greater_equal}


def _arr(x, n, t):
    return x if isinstance(x, (pa.Array, pa.ChunkedArray)) else pa.array([x.as_py()] * n, t)


def arrow_eval(node, batch):
    n, k = batch.num_rows, node.kind
    if k == "field":
        return batch.column(batch.schema.get_field_index(node.desc["name"]))
    if k == "literal":
        return pa.array([None if node.desc["is_null"] else node.desc["value"]] * n, node.dtype)
    kids = [arrow_eval(c, batch) for c in node.desc.get("children", [])]
    if k == "if":
        take = pc.fill_null(kids[0], False)           # a null condition selects the else branch
        return pc.if_else(take, kids[1], kids[2])
    if k in ("and", "or"):
        out = kids[0]
        for c in kids[1:]:
            out = pc.and_kleene(out, c) if k == "and" else pc.or_kleene(out, c)
        return out
    f = node.desc["name"]
    a = kids[0]
    if f in WRAP:
        return WRAP[f](a, kids[1])                    # the unchecked kernels wrap on overflow
    if f in CMP:
        return CMP[f](a, kids[1])
    if f == "negative":
        return pc.negate(a)
    if f == "abs":
        return pc.abs(a)
    if f == "isnull":
        return pc.is_null(a)
    if f == "isnotnull":
        return pc.is_valid(a)
    if f == "not":
        return pc.invert(a)
    if f in ("castBIGINT", "castFLOAT8", "castFLOAT4"):
        return pc.cast(a, node.dtype, safe=False)
    if f.startswith("bitwise_"):
        return {"bitwise_and": pc.bit_wise_and, "bitwise_or": pc.bit_wise_or, "bitwise_xor": pc.bit_wise_xor}[f](a, kids[1])
    if f == "istrue":
        return pc.fill_null(a, False)
    if f == "isfalse":
        return pc.fill_null(pc.invert(a), False)
    if f == "isnottrue":
        return pc.invert(pc.fill_null(a, False))
    if f == "isnotfalse":
        return pc.invert(pc.fill_null(pc.invert(a), False))
    if f in ("is_distinct_from", "is_not_distinct_from"):
        b = kids[1]
        both_null = pc.and_(pc.is_null(a), pc.is_null(b))
        differ = pc.fill_null(pc.not_equal(a, b), True)      # exactly one null -> distinct
        distinct = pc.and_(differ, pc.invert(both_null))
        return distinct if f == "is_distinct_from" else pc.invert(distinct)
    raise NotImplementedError(f)


@pytest.fixture(autouse=True)
def _restrict_generator(monkeypatch):
    monkeypatch.setattr(F, "EXACT", ARROW_OPS)


@pytest.mark.parametrize("seed", range(40))
def test_oracle_trees_match_pyarrow_compute(seed):
    exprs, cond = F._expressions(1000 + seed)
    batch = F._batch(seed, 777)
    got = oracle.project(exprs, batch)
    for g, e in zip(got, exprs):
        want = arrow_eval(e.root(), batch)
        if isinstance(want, pa.ChunkedArray):
            want = want.combine_chunks()
        assert_bit_exact(g, want.cast(g.type), f"seed {seed}: {e}")
    sel = oracle.filter_indices(cond, batch, "int32")
    keep = pc.fill_null(arrow_eval(cond.root(), batch), False)
    assert sel.to_pylist() == np.flatnonzero(np.asarray(keep)).tolist(), f"seed {seed}: {cond}"
