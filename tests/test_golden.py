"""Committed golden fixtures (tests/golden/c1..c5.arrow, made by tests/golden/make_golden.py:
oracle outputs accepted only where pyarrow.compute / Python decimal reproduce them).
CPU: the oracle must reproduce them; GPU: the HIP path must reproduce them."""
import os

import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from oracle import oracle
from helpers import assert_bit_exact

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = {
    "c1": (W.c1_schema, W.c1_expressions),
    "c2": (W.c2_schema, W.c2_expressions),
    "c4": (W.c4_schema, W.c4_expressions),
    "c5": (W.c5_schema, W.c5_expressions),
}


def load(name):
    t = pa.ipc.open_file(os.path.join(HERE, name + ".arrow")).read_all().combine_chunks()
    ins = [c for c in t.column_names if c.startswith("in_")]
    outs = [c for c in t.column_names if c.startswith("out_")]
    return ([t.column(c).chunk(0) for c in ins], [t.column(c).chunk(0) for c in outs])


def check(got, want):
    for g, w in zip(got, want):
        if pa.types.is_string(w.type):
            assert g.equals(w)
        else:
            assert_bit_exact(g, w)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_reproduces_golden_projection(name):
    schema, exprs = CONFIGS[name]
    ins, outs = load(name)
    batch = pa.RecordBatch.from_arrays(ins, schema=schema())
    check(oracle.project(exprs(), batch), outs)


def test_oracle_reproduces_golden_selection():
    ins, outs = load("c3")
    batch = pa.RecordBatch.from_arrays(ins, schema=W.c3_schema())
    want = outs[0].drop_null()
    assert oracle.filter_indices(W.c3_condition(), batch, "int32").equals(want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_hip_reproduces_golden_projection(name):
    schema, exprs = CONFIGS[name]
    ins, outs = load(name)
    batch = pa.RecordBatch.from_arrays(ins, schema=schema())
    check(gandiva.make_projector(batch.schema, exprs(), None).evaluate(batch), outs)


@pytest.mark.gpu
def test_hip_reproduces_golden_selection():
    ins, outs = load("c3")
    batch = pa.RecordBatch.from_arrays(ins, schema=W.c3_schema())
    got = gandiva.make_filter(batch.schema, W.c3_condition()).evaluate(batch, None).to_array()
    assert got.equals(outs[0].drop_null())
