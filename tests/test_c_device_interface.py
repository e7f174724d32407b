"""Arrow C Device Data Interface import: batches produced elsewhere are
handed to Evaluate as `struct ArrowDeviceArray` — exported by pyarrow itself for CPU memory,
assembled by hand (ctypes) over torch HBM tensors for ARROW_DEVICE_ROCM."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from oracle import oracle
from helpers import assert_bit_exact

pytestmark = pytest.mark.gpu


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64),
    ("n_buffers", C.c_int64), ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowDeviceArray(C.Structure):
    _fields_ = [("array", ArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32),
                ("sync_event", C.c_void_p), ("reserved", C.c_int64 * 3)]


class ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p),
                ("flags", C.c_int64), ("n_children", C.c_int64), ("children", C.c_void_p),
                ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


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
