"""The reference lineage's own known-answer tests, value for value.

Source of every expected value: pyarrow/tests/test_gandiva.py (the only golden vectors for
this path present in the container — SURVEY.md §4 / §8c).  Each KAT is run twice:
  * against the CPU oracle (pins the oracle; runs everywhere)
  * against the HIP path through the C ABI (`-m gpu`)
The test bodies mirror the originals; only `import pyarrow.gandiva` became `gandiva_amd`.
"""
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from oracle import oracle


# ------------------------------------------------------------------ KAT definitions
# each returns (kind, schema-batch, expression(s) or condition, expected, extra)

def kat_tree_exp_builder():  # test_gandiva.py:24-63
    builder = gandiva.TreeExprBuilder()
    field_a, field_b = pa.field('a', pa.int32()), pa.field('b', pa.int32())
    node_a, node_b = builder.make_field(field_a), builder.make_field(field_b)
    condition = builder.make_function("greater_than", [node_a, node_b], pa.bool_())
    if_node = builder.make_if(condition, node_a, node_b, pa.int32())
    expr = builder.make_expression(if_node, pa.field('res', pa.int32()))
    a = pa.array([10, 12, -20, 5], type=pa.int32())
    b = pa.array([5, 15, 15, 17], type=pa.int32())
    batch = pa.RecordBatch.from_arrays([a, b], names=['a', 'b'])
    return "project", batch, [expr], [pa.array([10, 15, 15, 17], type=pa.int32())]


def kat_table():  # test_gandiva.py:66-90
    table = pa.Table.from_arrays([pa.array([1.0, 2.0]), pa.array([3.0, 4.0])], ['a', 'b'])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    node_b = builder.make_field(table.schema.field("b"))
    s = builder.make_function("add", [node_a, node_b], pa.float64())
    expr = builder.make_expression(s, pa.field("c", pa.float64()))
    return "project", table.to_batches()[0], [expr], [pa.array([4.0, 6.0])]


def kat_filter():  # test_gandiva.py:93-114
    table = pa.Table.from_arrays([pa.array([1.0 * i for i in range(10000)])], ['a'])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    thousand = builder.make_literal(1000.0, pa.float64())
    cond = builder.make_function("less_than", [node_a, thousand], pa.bool_())
    condition = builder.make_condition(cond)
    return "filter", table.to_batches()[0], condition, pa.array(range(1000), type=pa.uint32())


def kat_in_utf8():  # test_gandiva.py:117-129
    arr = pa.array(["ga", "an", "nd", "di", "iv", "va"])
    table = pa.Table.from_arrays([arr], ["a"])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    cond = builder.make_in_expression(node_a, ["an", "nd"], pa.string())
    return "filter", table.to_batches()[0], builder.make_condition(cond), \
        pa.array([1, 2], type=pa.uint32())


def kat_regex():  # test_gandiva.py:295-316
    elements = ["park", "sparkle", "bright spark and fire", "spark"]
    data = pa.array(elements, type=pa.string())
    table = pa.Table.from_arrays([data], names=['a'])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    regex = builder.make_literal("%spark%", pa.string())
    like = builder.make_function("like", [node_a, regex], pa.bool_())
    expr = builder.make_expression(like, pa.field("b", pa.bool_()))
    return "project", table.to_batches()[0], [expr], [pa.array([False, True, True, True], type=pa.bool_())]


def kat_in_int32():  # test_gandiva.py:131-140
    arr = pa.array([3, 1, 4, 1, 5, 9, 2, 6, 5, 4])
    table = pa.Table.from_arrays([arr.cast(pa.int32())], ["a"])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    cond = builder.make_in_expression(node_a, [1, 5], pa.int32())
    return "filter", table.to_batches()[0], builder.make_condition(cond), \
        pa.array([1, 3, 4, 8], type=pa.uint32())


def kat_in_int64():  # test_gandiva.py:142-151
    arr = pa.array([3, 1, 4, 1, 5, 9, 2, 6, 5, 4])
    table = pa.Table.from_arrays([arr], ["a"])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    cond = builder.make_in_expression(node_a, [1, 5], pa.int64())
    return "filter", table.to_batches()[0], builder.make_condition(cond), \
        pa.array([1, 3, 4, 8], type=pa.uint32())


def kat_boolean():  # test_gandiva.py:228-252
    table = pa.Table.from_arrays([
        pa.array([1., 31., 46., 3., 57., 44., 22.]),
        pa.array([5., 45., 36., 73., 83., 23., 76.])], ['a', 'b'])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    node_b = builder.make_field(table.schema.field("b"))
    fifty = builder.make_literal(50.0, pa.float64())
    eleven = builder.make_literal(11.0, pa.float64())
    cond_1 = builder.make_function("less_than", [node_a, fifty], pa.bool_())
    cond_2 = builder.make_function("greater_than", [node_a, node_b], pa.bool_())
    cond_3 = builder.make_function("less_than", [node_b, eleven], pa.bool_())
    cond = builder.make_or([builder.make_and([cond_1, cond_2]), cond_3])
    return "filter", table.to_batches()[0], builder.make_condition(cond), \
        pa.array([0, 2, 5], type=pa.uint32())


def kat_filter_project():  # test_gandiva.py:329-373
    array0 = pa.array([10, 12, -20, 5, 21, 29], pa.int32())
    array1 = pa.array([5, 15, 15, 17, 12, 3], pa.int32())
    array2 = pa.array([1, 25, 11, 30, -21, None], pa.int32())
    table = pa.Table.from_arrays([array0, array1, array2], ['a', 'b', 'c'])
    builder = gandiva.TreeExprBuilder()
    node_a = builder.make_field(table.schema.field("a"))
    node_b = builder.make_field(table.schema.field("b"))
    node_c = builder.make_field(table.schema.field("c"))
    filter_condition = builder.make_condition(
        builder.make_function("greater_than", [node_a, node_b], pa.bool_()))
    project_condition = builder.make_function("less_than", [node_b, node_c], pa.bool_())
    if_node = builder.make_if(project_condition, node_b, node_c, pa.int32())
    expr = builder.make_expression(if_node, pa.field("res", pa.int32()))
    return "filter_project", table.to_batches()[0], (filter_condition, [expr]), \
        [pa.array([1, -21, None], pa.int32())]


KATS = [kat_tree_exp_builder, kat_table, kat_filter, kat_in_utf8, kat_in_int32, kat_in_int64,
        kat_boolean, kat_regex, kat_filter_project]


# ------------------------------------------------------------------ oracle pins (CPU)

@pytest.mark.parametrize("kat", KATS, ids=lambda k: k.__name__)
def test_oracle_matches_reference_kat(kat):
    kind, batch, what, expected = kat()
    if kind == "project":
        got = oracle.project(what, batch)
        for g, e in zip(got, expected):
            assert g.equals(e)
    elif kind == "filter":
        assert oracle.filter_indices(what, batch, "int32").equals(expected)
    else:
        cond, exprs = what
        sel = oracle.filter_indices(cond, batch, "int32")
        got = oracle.project(exprs, oracle.take_rows(batch, sel.to_numpy()))
        for g, e in zip(got, expected):
            assert g.equals(e)


# ------------------------------------------------------------------ HIP path (GPU)

@pytest.mark.gpu
@pytest.mark.parametrize("kat", KATS, ids=lambda k: k.__name__)
def test_hip_matches_reference_kat(kat):
    kind, batch, what, expected = kat()
    pool = pa.default_memory_pool()
    if kind == "project":
        config = gandiva.Configuration(dump_ir=True)
        projector = gandiva.make_projector(batch.schema, what, pool, "NONE", config)
        # Gandiva generates compute kernel function named `@expr_X` (test_gandiva.py:54-55)
        assert projector.llvm_ir.find("@expr_") != -1
        got = projector.evaluate(batch)
        for g, e in zip(got, expected):
            assert g.equals(e)
    elif kind == "filter":
        flt = gandiva.make_filter(batch.schema, what, gandiva.Configuration(dump_ir=True))
        assert flt.llvm_ir.find("@expr_") != -1
        result = flt.evaluate(batch, pool)
        assert result.to_array().equals(expected)
    else:
        cond, exprs = what
        flt = gandiva.make_filter(batch.schema, cond)
        projector = gandiva.make_projector(batch.schema, exprs, pool, "UINT32")
        selection_vector = flt.evaluate(batch, pool)
        r, = projector.evaluate(batch, selection_vector)
        assert r.equals(expected[0])
