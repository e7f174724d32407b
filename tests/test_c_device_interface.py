"""Arrow C Device Data Interface import and export: batches produced
elsewhere are handed to Evaluate as `struct ArrowDeviceArray` — exported by pyarrow itself for
CPU memory, assembled by hand (ctypes) over torch HBM tensors for ARROW_DEVICE_ROCM — and
results are handed on the same way (imported back by pyarrow for CPU memory; read through
the raw struct for HBM)."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from oracle import oracle
from helpers import assert_bit_exact

pytestmark = pytest.mark.gpu


from gandiva_amd._capi import ArrowArray, ArrowDeviceArray, ArrowSchema, release_c_struct


def test_cpu_device_array_exported_by_pyarrow():
    batch = W.c2_batch(50021)
    exprs = W.c2_expressions()
    proj = gandiva.make_projector(batch.schema, exprs, None)
    c_arr, c_schema = ArrowDeviceArray(), ArrowSchema()
    batch._export_to_c_device(C.addressof(c_arr), C.addressof(c_schema))
    try:
        assert c_arr.device_type == 1  # ARROW_DEVICE_CPU
        assert c_arr.array.n_children == 4 and c_arr.array.length == batch.num_rows
        got = proj.evaluate_device_array(C.addressof(c_arr), batch.num_rows, on_device=False)
    finally:
        if c_arr.array.release:
            C.CFUNCTYPE(None, C.c_void_p)(c_arr.array.release)(C.addressof(c_arr.array))
        if c_schema.release:
            C.CFUNCTYPE(None, C.c_void_p)(c_schema.release)(C.addressof(c_schema))
    for g, w in zip(got, oracle.project(exprs, batch)):
        assert_bit_exact(g, w)


def test_rocm_device_array_over_torch_tensors_is_zero_copy():
    import torch
    n = 70003
    batch = W.c2_batch(n)
    exprs = W.c2_expressions()
    proj = gandiva.make_projector(batch.schema, exprs, None)
    dbatch = gandiva.DeviceBatch.from_arrow(batch)
    keep = []
    children = (C.POINTER(ArrowArray) * 4)()
    for i, col in enumerate(dbatch.columns):
        bufs = (C.c_void_p * 2)(col.validity.data_ptr(), col.data.data_ptr())
        child = ArrowArray(n, -1, 0, 2, 0, bufs, None, None, C.c_void_p(1), None)
        keep += [bufs, child]
        children[i] = C.pointer(child)
    top_bufs = (C.c_void_p * 1)(None)
    dev = ArrowDeviceArray()
    dev.array = ArrowArray(n, 0, 0, 1, 4, top_bufs, children, None, C.c_void_p(1), None)
    dev.device_id, dev.device_type, dev.sync_event = 0, 10, None  # ARROW_DEVICE_ROCM
    outs = proj.evaluate_device_array(C.addressof(dev), n, on_device=True)
    torch.cuda.synchronize()
    for o, w in zip(outs, oracle.project(exprs, batch)):
        assert_bit_exact(o.to_arrow(), w)
    # a filter over the same hand-over
    cond_batch = W.c3_batch(n, 0.1)
    dcb = gandiva.DeviceBatch.from_arrow(cond_batch)
    ch2 = (C.POINTER(ArrowArray) * 2)()
    for i, col in enumerate(dcb.columns):
        bufs = (C.c_void_p * 2)(col.validity.data_ptr() if col.validity is not None else None, col.data.data_ptr())
        child = ArrowArray(n, -1, 0, 2, 0, bufs, None, None, C.c_void_p(1), None)
        keep += [bufs, child]
        ch2[i] = C.pointer(child)
    dev2 = ArrowDeviceArray()
    dev2.array = ArrowArray(n, 0, 0, 1, 2, top_bufs, ch2, None, C.c_void_p(1), None)
    dev2.device_id, dev2.device_type = 0, 10
    flt = gandiva.make_filter(cond_batch.schema, W.c3_condition())
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    count = C.c_int64()
    from gandiva_amd import _capi
    rc = _capi.lib().gdv_filter_evaluate_device_array(flt._h, C.addressof(dev2), 2, C.c_void_p(out.data_ptr()),
                                                      n, C.byref(count), None)
    assert rc == 0, _capi.last_error()
    want = oracle.filter_indices(W.c3_condition(), cond_batch, "int32")
    assert np.array_equal(out[:count.value].cpu().numpy().view(np.uint32), want.to_numpy())


def test_bad_device_arrays_are_rejected():
    batch = W.c1_batch(100)
    proj = gandiva.make_projector(batch.schema, W.c1_expressions(), None)
    dev = ArrowDeviceArray()
    dev.array.release = None
    with pytest.raises(pa.ArrowInvalid, match="released"):
        proj.evaluate_device_array(C.addressof(dev), 100, on_device=False)
    dev.array.release = C.c_void_p(1)
    dev.array.n_children = 1
    dev.device_type = 1
    with pytest.raises(pa.ArrowInvalid, match="children"):
        proj.evaluate_device_array(C.addressof(dev), 100, on_device=False)


# ------------------------------------------------------------------ export

def _rocm_device_array(dbatch, n, keep):
    children = (C.POINTER(ArrowArray) * len(dbatch.columns))()
    for i, col in enumerate(dbatch.columns):
        if col.offsets is not None:
            bufs = (C.c_void_p * 3)(col.validity.data_ptr() if col.validity is not None else None,
                                    col.offsets.data_ptr(), col.data.data_ptr())
        else:
            bufs = (C.c_void_p * 2)(col.validity.data_ptr() if col.validity is not None else None,
                                    col.data.data_ptr())
        child = ArrowArray(n, -1, 0, len(bufs), 0, bufs, None, None, C.c_void_p(1), None)
        keep += [bufs, child]
        children[i] = C.pointer(child)
    top_bufs = (C.c_void_p * 1)(None)
    dev = ArrowDeviceArray()
    dev.array = ArrowArray(n, 0, 0, 1, len(dbatch.columns), top_bufs, children, None, C.c_void_p(1), None)
    dev.device_id, dev.device_type, dev.sync_event = 0, 10, None
    keep += [top_bufs, children]
    return dev


def _read_device_child(child, t, n):
    """Copies the buffers of one exported ROCm child back and rebuilds a pyarrow array."""
    from gandiva_amd import _capi
    lib = _capi.lib()

    def fetch(ptr, nbytes):
        buf = pa.allocate_buffer(max(nbytes, 1))
        if nbytes:
            assert lib.gdv_memcpy_d2h(C.c_void_p(buf.address), C.c_void_p(ptr), nbytes) == 0
        return buf
    validity = fetch(child.buffers[0], (n + 7) // 8)
    if pa.types.is_string(t) or pa.types.is_binary(t):
        offsets = fetch(child.buffers[1], (n + 1) * 4)
        total = int(np.frombuffer(offsets, dtype=np.int32)[n])
        return pa.Array.from_buffers(t, n, [validity, offsets, fetch(child.buffers[2], total)])
    width = (n + 7) // 8 if pa.types.is_boolean(t) else n * (t.bit_width // 8)
    return pa.Array.from_buffers(t, n, [validity, fetch(child.buffers[1], width)])


@pytest.mark.parametrize("workload", ["c2", "c4", "c5"])
def test_cpu_export_is_importable_by_pyarrow(workload):
    batch = {"c2": W.c2_batch, "c4": W.c4_batch, "c5": W.c5_batch}[workload](30011)
    exprs = {"c2": W.c2_expressions, "c4": W.c4_expressions, "c5": W.c5_expressions}[workload]()
    proj = gandiva.make_projector(batch.schema, exprs, None)
    c_in, c_in_schema = ArrowDeviceArray(), ArrowSchema()
    batch._export_to_c_device(C.addressof(c_in), C.addressof(c_in_schema))
    try:
        out, schema = proj.evaluate_export(C.addressof(c_in))
    finally:
        release_c_struct(c_in)
        release_c_struct(c_in_schema)
    assert out.device_type == 1 and not out.sync_event
    got = pa.RecordBatch._import_from_c_device(C.addressof(out), C.addressof(schema))  # takes ownership
    assert not out.array.release and not schema.release
    assert got.num_rows == batch.num_rows
    assert got.schema.names == [e.result().name for e in exprs]
    for g, w, e in zip(got.columns, oracle.project(exprs, batch), exprs):
        assert g.type == w.type
        assert_bit_exact(g, w, str(e))


def test_export_regrows_varlen_buffers():
    """No var-len input: the first capacity guess (64 bytes) is always too small."""
    n = 5000
    rng = np.random.default_rng(3)
    batch = pa.RecordBatch.from_arrays([pa.array(rng.integers(-5, 5, n), pa.int64())], names=["a"])
    b = gandiva.TreeExprBuilder()
    a = b.make_field(batch.schema.field(0))
    pos = b.make_function("greater_than", [a, b.make_literal(0, pa.int64())], pa.bool_())
    node = b.make_if(pos, b.make_literal("a rather long positive label", pa.string()),
                     b.make_literal("neg", pa.string()), pa.string())
    exprs = [b.make_expression(node, pa.field("label", pa.string()))]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    c_in, c_in_schema = ArrowDeviceArray(), ArrowSchema()
    batch._export_to_c_device(C.addressof(c_in), C.addressof(c_in_schema))
    try:
        out, schema = proj.evaluate_export(C.addressof(c_in))
    finally:
        release_c_struct(c_in)
        release_c_struct(c_in_schema)
    got = pa.RecordBatch._import_from_c_device(C.addressof(out), C.addressof(schema))
    assert_bit_exact(got.column(0), oracle.project(exprs, batch)[0])


def test_rocm_export_stays_in_hbm_and_children_outlive_the_parent():
    n = 40009
    batch = W.c5_batch(n)
    exprs = W.c5_expressions()
    proj = gandiva.make_projector(batch.schema, exprs, None)
    keep = []
    dev = _rocm_device_array(gandiva.DeviceBatch.from_arrow(batch), n, keep)
    out, schema = proj.evaluate_export(C.addressof(dev))
    assert out.device_type == 10 and out.sync_event          # ARROW_DEVICE_ROCM + hipEvent_t*
    assert out.array.n_children == len(exprs) and out.array.length == n
    fmts = [C.cast(C.cast(schema.children, C.POINTER(C.c_void_p))[i], C.POINTER(ArrowSchema)).contents.format
            for i in range(len(exprs))]
    assert fmts == [b"b", b"u", b"u"]
    want = oracle.project(exprs, batch)
    for i, (w, e) in enumerate(zip(want, exprs)):
        child = out.array.children[i].contents
        assert child.n_buffers == (3 if pa.types.is_string(w.type) else 2)
        assert_bit_exact(_read_device_child(child, w.type, n), w, str(e))
    # move the last child out, release the parent, the child's buffers must stay alive
    moved = ArrowArray()
    C.memmove(C.addressof(moved), C.addressof(out.array.children[2].contents), C.sizeof(ArrowArray))
    out.array.children[2].contents.release = None
    release_c_struct(out)
    assert not out.array.release
    assert_bit_exact(_read_device_child(moved, want[2].type, n), want[2])
    release_c_struct(moved)
    release_c_struct(schema)
    assert not moved.release and not schema.release
