"""Round 4: host buffers the GPU addresses in place (gdv_host_register / gdv_host_alloc).  A GDV_MEM_HOST
evaluation whose buffers lie in a registered range copies nothing — `gdv_host_staged_bytes` says so — and
returns what the staged path returns."""
import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from helpers import assert_bit_exact
from oracle import oracle


def test_the_host_memory_entry_points_are_exported():
    from gandiva_amd import _capi
    lib = _capi.lib()
    for name in ("gdv_host_register", "gdv_host_unregister", "gdv_host_alloc", "gdv_host_free", "gdv_host_staged_bytes"):
        assert hasattr(lib, name)
    assert gandiva.host_staged_bytes() >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 4099, 16384, 70_001])
def test_c2_in_an_arena_is_evaluated_in_place(n):
    batch = W.c2_batch(n)
    exprs = W.c2_expressions()
    proj = gandiva.make_projector(batch.schema, exprs, pa.default_memory_pool())
    want = oracle.project(exprs, batch)
    arena = gandiva.HostArena(64 << 20)
    placed = arena.place(batch)
    before = gandiva.host_staged_bytes()
    got = proj.evaluate(placed, arena=arena)
    assert gandiva.host_staged_bytes() == before, "a buffer of the call went through the staging block"
    for g, w in zip(got, want):
        assert_bit_exact(g, w, "arena in, arena out")
    # outputs in the arena, inputs where pyarrow put them: only the inputs are staged
    before = gandiva.host_staged_bytes()
    got = proj.evaluate(batch, arena=arena)
    moved = gandiva.host_staged_bytes() - before
    assert 0 < moved <= sum(b.size for a in batch.columns for b in a.buffers() if b is not None) + 64 * batch.num_columns
    for g, w in zip(got, want):
        assert_bit_exact(g, w, "pageable in, arena out")
    # and the other way round
    got = proj.evaluate(placed)
    for g, w in zip(got, want):
        assert_bit_exact(g, w, "arena in, pageable out")


@pytest.mark.gpu
def test_registered_memory_of_the_caller_sliced_batches_bool_outputs_and_the_filter():
    n = 50_021
    rng = np.random.default_rng(3)
    a = pa.array(rng.integers(-1000, 1000, n), pa.int64(), mask=rng.random(n) < 0.1)
    b = pa.array(rng.integers(-1000, 1000, n).astype(np.int32), pa.int32(), mask=rng.random(n) < 0.1)
    f = pa.array(rng.random(n) < 0.5, pa.bool_(), mask=rng.random(n) < 0.1)
    batch = pa.RecordBatch.from_arrays([a, b, f], names=["a", "b", "f"])
    tb = gandiva.TreeExprBuilder()
    fa, fb, ff = (tb.make_field(batch.schema.field(i)) for i in range(3))
    lt = tb.make_function("less_than", [fa, tb.make_function("castBIGINT", [fb], pa.int64())], pa.bool_())
    exprs = [tb.make_expression(lt, pa.field("lt", pa.bool_())),
             tb.make_expression(tb.make_and([lt, ff]), pa.field("both", pa.bool_())),
             tb.make_expression(tb.make_function("add", [fa, tb.make_function("castBIGINT", [fb], pa.int64())], pa.int64()),
                                pa.field("s", pa.int64()))]
    cond = tb.make_condition(tb.make_or([lt, ff]))
    proj = gandiva.make_projector(batch.schema, exprs, None)
    flt = gandiva.make_filter(batch.schema, cond)
    block = np.zeros(32 << 20, dtype=np.uint8)          # the caller's own memory
    arena = gandiva.HostArena.over(block)
    placed = arena.place(batch)
    for lo, ln in ((0, n), (5, n - 5), (64, 1000), (4097, 30_000), (n - 1, 1)):
        sl, ref = placed.slice(lo, ln), batch.slice(lo, ln)
        before = gandiva.host_staged_bytes()
        got = proj.evaluate(sl, arena=arena)
        assert gandiva.host_staged_bytes() == before
        for g, w in zip(got, oracle.project(exprs, ref)):
            assert_bit_exact(g, w, f"slice [{lo}, {lo + ln})")
        sel = flt.evaluate(sl, None, "int32")
        assert sel.to_array().equals(oracle.filter_indices(cond, ref, "int32"))
    # once the registration is gone the same addresses are staged again — and the results do not change
    from gandiva_amd import _capi
    import ctypes
    assert _capi.lib().gdv_host_unregister(ctypes.c_void_p(block.ctypes.data)) == 0
    arena._base = None                       # (nothing left for the arena to undo)
    before = gandiva.host_staged_bytes()
    got = proj.evaluate(placed)
    assert gandiva.host_staged_bytes() > before
    for g, w in zip(got, oracle.project(exprs, batch)):
        assert_bit_exact(g, w, "staged")
    assert _capi.lib().gdv_host_unregister(ctypes.c_void_p(block.ctypes.data)) != 0   # not registered any more


@pytest.mark.gpu
def test_var_len_columns_in_registered_memory_are_read_in_place():
    """Round 5: the offsets and the bytes of a utf8 column that lies in registered host memory are bound in place
    (round 4 staged them); var-len OUTPUTS still come back through the staging block."""
    n = 20_011
    batch = W.c5_batch(n, 0.1)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    fixed = [b.make_expression(b.make_function("like", [s, b.make_literal("%spark%", pa.string())], pa.bool_()), pa.field("m", pa.bool_())),
             b.make_expression(b.make_function("char_length", [s], pa.int32()), pa.field("n", pa.int32())),
             b.make_expression(b.make_function("hash64", [s], pa.int64()), pa.field("h", pa.int64()))]
    arena = gandiva.HostArena(64 << 20)
    placed = arena.place(batch)
    proj = gandiva.make_projector(batch.schema, fixed, None)
    proj.evaluate(placed, arena=arena)                       # (first call: whatever Make-time uploads there are)
    before = gandiva.host_staged_bytes()
    got = proj.evaluate(placed, arena=arena)
    assert gandiva.host_staged_bytes() == before, "a var-len column in an arena must not be staged"
    for g, w in zip(got, oracle.project(fixed, batch)):
        assert_bit_exact(g, w, "var-len input in place, fixed-width outputs")
    # sliced (array offset) and multi-byte text
    sl = arena.place(W.c5_batch(n, 0.1, non_ascii_fraction=0.05)).slice(777, 9000)
    for g, w in zip(proj.evaluate(sl, arena=arena), oracle.project(fixed, sl)):
        assert_bit_exact(g, w, "sliced, multi-byte")
    # var-len outputs: inputs in place, outputs staged, same answers
    exprs = W.c5_expressions()
    p5 = gandiva.make_projector(batch.schema, exprs, None)
    for g, w in zip(p5.evaluate(placed, arena=arena), oracle.project(exprs, batch)):
        assert_bit_exact(g, w, "C5 through an arena")
