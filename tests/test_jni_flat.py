"""JNI-shaped flat entry points: the buffers of a batch as two long[]
arrays (addresses, sizes) in the order the reference's Java side flattens them — validity,
[offsets,] data per field — and the outputs the same way."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import _capi, workloads as W
from oracle import oracle
from helpers import assert_bit_exact

pytestmark = pytest.mark.gpu


def _flatten(batch):
    addrs, sizes, keep = [], [], []
    for col in batch.columns:
        bufs = col.buffers()
        assert col.offset == 0
        for b in bufs:
            addrs.append(b.address if b is not None else 0)
            sizes.append(b.size if b is not None else 0)
        keep.append(bufs)
    n = len(addrs)
    return (C.c_int64 * n)(*addrs), (C.c_int64 * n)(*sizes), n, keep


def test_projector_flat_matches_oracle_fixed_and_varlen():
    n = 20011
    batch = W.c5_batch(n, 0.1)
    exprs = W.c5_expressions()   # bool, utf8, utf8
    proj = gandiva.make_projector(batch.schema, exprs, None)
    addrs, sizes, nb, keep = _flatten(batch)
    vb = (n + 7) // 8
    bufs = [np.zeros(vb, np.uint8), np.zeros(vb, np.uint8),                                # like
            np.zeros(vb, np.uint8), np.zeros(n + 1, np.int32), np.zeros(16, np.uint8),     # substr (too small)
            np.zeros(vb, np.uint8), np.zeros(n + 1, np.int32), np.zeros(16, np.uint8)]     # upper (too small)
    lib = _capi.lib()

    def call():
        oa = (C.c_int64 * len(bufs))(*[b.ctypes.data for b in bufs])
        osz = (C.c_int64 * len(bufs))(*[b.nbytes for b in bufs])
        rc = lib.gdv_projector_evaluate_flat(proj._h, n, addrs, sizes, nb, 0, 0, 0, oa, osz, len(bufs), 0)
        return rc, list(osz)
    rc, osz = call()
    assert rc == 4 and osz[4] > 16 and osz[7] > 16      # GDV_INVALID + bytes needed reported
    bufs[4], bufs[7] = np.zeros(osz[4], np.uint8), np.zeros(osz[7], np.uint8)
    rc, osz = call()
    assert rc == 0, _capi.last_error()
    got = [pa.Array.from_buffers(pa.bool_(), n, [pa.py_buffer(bufs[0]), pa.py_buffer(bufs[1])]),
           pa.Array.from_buffers(pa.string(), n, [pa.py_buffer(bufs[2]), pa.py_buffer(bufs[3]), pa.py_buffer(bufs[4])]),
           pa.Array.from_buffers(pa.string(), n, [pa.py_buffer(bufs[5]), pa.py_buffer(bufs[6]), pa.py_buffer(bufs[7])])]
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, str(e))


def test_filter_flat_then_projector_flat_with_selection():
    n = 50021
    batch = W.c3_batch(n, 0.1)
    cond = W.c3_condition()
    flt = gandiva.make_filter(batch.schema, cond)
    addrs, sizes, nb, keep = _flatten(batch)
    idx = np.zeros(n, np.uint32)
    count = C.c_int64()
    lib = _capi.lib()
    rc = lib.gdv_filter_evaluate_flat(flt._h, n, addrs, sizes, nb, 2, idx.ctypes.data, idx.nbytes, C.byref(count), 0)
    assert rc == 0, _capi.last_error()
    want = oracle.filter_indices(cond, batch, "int32")
    assert np.array_equal(idx[:count.value], want.to_numpy())
    # project a + b over the selected rows
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    ex = [b.make_expression(b.make_function("add", [fa, fb], pa.int64()), pa.field("s", pa.int64()))]
    proj = gandiva.make_projector(batch.schema, ex, None, "UINT32")
    k = count.value
    outs = [np.zeros((k + 7) // 8, np.uint8), np.zeros(k, np.int64)]
    oa = (C.c_int64 * 2)(*[o.ctypes.data for o in outs])
    osz = (C.c_int64 * 2)(*[o.nbytes for o in outs])
    rc = lib.gdv_projector_evaluate_flat(proj._h, n, addrs, sizes, nb, 2, idx.ctypes.data, k, oa, osz, 2, 0)
    assert rc == 0, _capi.last_error()
    got = pa.Array.from_buffers(pa.int64(), k, [pa.py_buffer(outs[0]), pa.py_buffer(outs[1])])
    assert_bit_exact(got, oracle.take_rows(oracle.project(ex, batch)[0], want))
    # wrong buffer count is rejected
    assert lib.gdv_projector_evaluate_flat(proj._h, n, addrs, sizes, nb - 1, 2, idx.ctypes.data, k, oa, osz, 2, 0) == 4
