"""f4, the `build` half of the JNI boundary (round 3): Schema / ExpressionList / Condition arrive as
protobuf BYTES (what the reference's Java side hands to JNI buildProjector / buildFilter), are
decoded by the hand-written wire decoder of gandiva_amd/csrc/gdv_proto.cc, and the handles are
evaluated through the flat long[] entry points — the whole JNI-shaped path without a JVM.
The encoder (tests/proto_encode.py) is test infrastructure; the message layout both sides assume is
a restatement from memory of the lineage's Types.proto (no source in the reference mount)."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import _capi
from oracle import oracle
import proto_encode as P
from test_reference_kats import KATS
import test_fuzz_trees as F


def _describe(schema_bytes, body, is_condition):
    lib = _capi.lib()
    p = lib.gdv_proto_describe(schema_bytes, len(schema_bytes), body, len(body), 1 if is_condition else 0)
    assert p, _capi.last_error()
    return _capi.take_string(p)


@pytest.mark.parametrize("kat", KATS, ids=lambda k: k.__name__)
def test_decoded_kat_trees_render_like_the_built_ones(kat):
    kind, batch, what, _ = kat()
    sb = P.schema(batch.schema)
    if kind == "filter_project":
        cond, exprs = what
    else:
        cond, exprs = (what, None) if kind == "filter" else (None, what)
    if exprs is not None:
        text = _describe(sb, P.expression_list(exprs), False)
        for f in batch.schema:
            assert f"field {f.name}: {f.type}" in text
        for e in exprs:
            assert f"expr {e.result().name}: {e.result().type} = {e}" in text, text
    if cond is not None:
        assert f"condition {cond}" in _describe(sb, P.condition(cond), True)


@pytest.mark.parametrize("seed", range(10))
def test_decoded_random_trees_render_like_the_built_ones(seed):
    """Every node kind: functions, if/else, AND/OR, IN lists, literals of every type the fuzz
    generator knows, negative integers (10-byte varints), nulls."""
    exprs, cond = F._expressions(seed) if hasattr(F, "_expressions") else (None, None)
    if exprs is None:
        pytest.skip("numeric tree generator not exposed")
    schema = F._batch(seed, 4).schema if hasattr(F, "_batch") else None
    sb = P.schema(schema)
    try:
        body = P.expression_list(exprs)
    except NotImplementedError as e:
        pytest.skip(str(e))
    text = _describe(sb, body, False)
    for e in exprs:
        assert f" = {e}\n" in text, (str(e), text)
    try:
        cbody = P.condition(cond)
    except NotImplementedError as e:
        pytest.skip(str(e))
    assert f"condition {cond}" in _describe(sb, cbody, True)


@pytest.mark.parametrize("seed", range(6))
def test_decoded_string_trees_render_like_the_built_ones(seed):
    exprs, cond = F._string_expressions(seed)
    sb = P.schema(F._string_batch(0, 1).schema)
    text = _describe(sb, P.expression_list(exprs), False)
    for e in exprs:
        assert f" = {e}\n" in text, (str(e), text)
    assert f"condition {cond}" in _describe(sb, P.condition(cond), True)


def test_malformed_bytes_are_refused_not_crashed_on():
    lib = _capi.lib()
    batch = KATS[0]()[1]
    sb, eb = P.schema(batch.schema), P.expression_list(KATS[0]()[2])
    out = C.c_void_p()
    rng = np.random.default_rng(0)
    for cut in range(0, len(eb), 3):      # truncations
        rc = lib.gdv_projector_make_from_proto(sb, len(sb), eb[:cut], cut, 0, None, C.byref(out))
        assert rc != 0 or cut == 0 or out.value, (cut, rc)
        if rc == 0 and out.value:
            lib.gdv_projector_free(out)
    for _ in range(300):                  # bit flips
        raw = bytearray(eb)
        raw[int(rng.integers(0, len(raw)))] ^= 1 << int(rng.integers(0, 8))
        p = lib.gdv_proto_describe(sb, len(sb), bytes(raw), len(raw), 0)
        if p:
            lib.gdv_free_string(p)
    assert lib.gdv_projector_make_from_proto(sb, len(sb), b"\x12\x02\x0a\x00", 4, 0, None, C.byref(out)) != 0
    assert "malformed" in _capi.last_error() or "protobuf" in _capi.last_error()


def _flat(batch):
    addrs, sizes, keep = [], [], []
    for col in batch.columns:
        for b in col.buffers():
            addrs.append(b.address if b is not None else 0)
            sizes.append(b.size if b is not None else 0)
        keep.append(col)
    n = len(addrs)
    return (C.c_int64 * n)(*addrs), (C.c_int64 * n)(*sizes), n


def _project_flat(handle, batch, types, sel=None):
    """gdv_projector_evaluate_flat over host buffers: returns pyarrow arrays (fixed-width / bool)."""
    lib = _capi.lib()
    addrs, sizes, nb = _flat(batch)
    rows = batch.num_rows if sel is None else len(sel)
    bufs = []
    for t in types:
        bufs.append(np.zeros((rows + 7) // 8 + 8, np.uint8))
        bufs.append(np.zeros((rows + 7) // 8 + 8 if t == pa.bool_() else rows * t.bit_width // 8 + 8, np.uint8))
    oa = (C.c_int64 * len(bufs))(*[b.ctypes.data for b in bufs])
    osz = (C.c_int64 * len(bufs))(*[b.nbytes for b in bufs])
    if sel is None:
        rc = lib.gdv_projector_evaluate_flat(handle, batch.num_rows, addrs, sizes, nb, 0, 0, 0, oa, osz, len(bufs), 0)
    else:
        rc = lib.gdv_projector_evaluate_flat(handle, batch.num_rows, addrs, sizes, nb, 2, sel.ctypes.data, len(sel),
                                             oa, osz, len(bufs), 0)
    assert rc == 0, _capi.last_error()
    return [pa.Array.from_buffers(t, rows, [pa.py_buffer(bufs[2 * i]), pa.py_buffer(bufs[2 * i + 1])])
            for i, t in enumerate(types)]


@pytest.mark.gpu
@pytest.mark.parametrize("kat", KATS, ids=lambda k: k.__name__)
def test_nine_reference_kats_through_build_from_proto_and_evaluate_flat(kat):
    kind, batch, what, expected = kat()
    lib = _capi.lib()
    sb = P.schema(batch.schema)
    if kind == "filter_project":
        cond, exprs = what
    else:
        cond, exprs = (what, None) if kind == "filter" else (None, what)
    sel = None
    if cond is not None:
        cb = P.condition(cond)
        fh = C.c_void_p()
        assert lib.gdv_filter_make_from_proto(sb, len(sb), cb, len(cb), None, C.byref(fh)) == 0, _capi.last_error()
        addrs, sizes, nb = _flat(batch)
        idx = np.zeros(batch.num_rows, np.uint32)
        count = C.c_int64()
        assert lib.gdv_filter_evaluate_flat(fh, batch.num_rows, addrs, sizes, nb, 2, idx.ctypes.data, idx.nbytes,
                                            C.byref(count), 0) == 0, _capi.last_error()
        sel = idx[:count.value].copy()
        lib.gdv_filter_free(fh)
        if kind == "filter":
            assert pa.array(sel, pa.uint32()).equals(expected)
            return
    eb = P.expression_list(exprs)
    ph = C.c_void_p()
    mode = 2 if sel is not None else 0   # SV_INT32 / SV_NONE
    assert lib.gdv_projector_make_from_proto(sb, len(sb), eb, len(eb), mode, None, C.byref(ph)) == 0, _capi.last_error()
    got = _project_flat(ph, batch, [e.result().type for e in exprs], sel)
    lib.gdv_projector_free(ph)
    for g, w in zip(got, expected):
        assert g.equals(w), (g, w)


def test_type_parameters_in_the_bytes_are_validated():
    """Round-3 advisor: any integer used to pass as a time unit or as a decimal precision / scale, a
    decimal literal of more than 38 digits wrapped silently, and gdv_proto_describe took a negative
    length for a huge one.  All refused with a Status now."""
    lib = _capi.lib()

    def schema_of(type_bytes):
        return P.ld(1, P.ld(1, b"x") + P.ld(2, type_bytes) + P.vi(3, 1))

    def describe(sb, body=b""):
        p = lib.gdv_proto_describe(sb, len(sb), body, len(body), 0)
        if p:
            return _capi.take_string(p)
        return None
    ok = [P.vi(1, 18) + P.vi(6, 1), P.vi(1, 18) + P.vi(6, 3), P.vi(1, 19) + P.vi(6, 0), P.vi(1, 20) + P.vi(6, 3),
          P.vi(1, 22) + P.vi(3, 38) + P.vi(4, 38), P.vi(1, 22) + P.vi(3, 1) + P.vi(4, 0)]
    for t in ok:
        assert describe(schema_of(t)) is not None, _capi.last_error()
    bad = [P.vi(1, 18) + P.vi(6, 4), P.vi(1, 18) + P.vi(6, 77), P.vi(1, 19) + P.vi(6, 2), P.vi(1, 20) + P.vi(6, 1),
           P.vi(1, 22) + P.vi(3, 0) + P.vi(4, 0), P.vi(1, 22) + P.vi(3, 39) + P.vi(4, 2),
           P.vi(1, 22) + P.vi(3, 10) + P.vi(4, 11), P.vi(1, 22) + P.vi(3, 10) + P.vi(4, -1)]
    for t in bad:
        assert describe(schema_of(t)) is None, t
        assert "malformed" in _capi.last_error()
    # decimal literals: DecimalNode { value = 1 (digits), precision = 2, scale = 3 } is TreeNode field 19
    sb = schema_of(P.vi(1, 22) + P.vi(3, 38) + P.vi(4, 2))

    def expr_with_decimal(digits, precision, scale):
        node = P.ld(19, P.ld(1, digits.encode()) + P.vi(2, precision) + P.vi(3, scale))
        rt = P.ld(2, P.vi(1, 22) + P.vi(3, 38) + P.vi(4, 2))
        return P.ld(2, P.ld(1, node) + P.ld(2, P.ld(1, b"r") + rt + P.vi(3, 1)))   # ExpressionList.exprs = 2
    assert describe(sb, expr_with_decimal("9" * 38, 38, 2)) is not None, _capi.last_error()
    assert describe(sb, expr_with_decimal("-" + "9" * 38, 38, 2)) is not None, _capi.last_error()
    assert describe(sb, expr_with_decimal("000" + "9" * 38, 38, 2)) is not None      # leading zeros are not digits
    for digits, p, s in (("1" + "0" * 38, 38, 2), ("9" * 60, 38, 2), ("12345", 4, 2), ("1", 0, 0), ("1", 39, 0), ("1", 5, 6)):
        assert describe(sb, expr_with_decimal(digits, p, s)) is None, (digits, p, s)
    # negative lengths
    assert not lib.gdv_proto_describe(sb, -1, b"", 0, 0)
    assert not lib.gdv_proto_describe(sb, len(sb), b"", -5, 0)


@pytest.mark.gpu
def test_the_filter_project_kat_as_one_fused_operator_built_from_proto_bytes():
    """pyarrow/tests/test_gandiva.py:329-373 (filter -> UINT32 selection -> project, one null) through
    gdv_filter_project_make_from_proto + gdv_filter_project_evaluate: ONE kernel, same answer."""
    kat = [k for k in KATS if k()[0] == "filter_project"][0]
    _, batch, (cond, exprs), expected = kat()
    lib = _capi.lib()
    sb, cb, eb = P.schema(batch.schema), P.condition(cond), P.expression_list(exprs)
    fp = C.c_void_p()
    assert lib.gdv_filter_project_make_from_proto(sb, len(sb), cb, len(cb), eb, len(eb), 2, None, C.byref(fp)) == 0, \
        _capi.last_error()
    n = batch.num_rows
    cols = (_capi.gdv_column_t * batch.num_columns)(*[gandiva.gandiva._column_of_array(a) for a in batch.columns])
    outs = (_capi.gdv_out_column_t * len(exprs))()
    keep = []
    for i, e in enumerate(exprs):
        v, d = np.zeros(64, np.uint8), np.zeros(n * 8 + 64, np.uint8)
        keep.append((v, d))
        outs[i].validity, outs[i].validity_size = v.ctypes.data, v.nbytes
        outs[i].data, outs[i].data_size = d.ctypes.data, d.nbytes
    idx = np.zeros(n, np.uint32)
    count = C.c_int64()
    assert lib.gdv_filter_project_evaluate(fp, n, cols, batch.num_columns, outs, len(exprs), C.c_void_p(idx.ctypes.data), n,
                                           C.byref(count), None, 0, None, 0) == 0, _capi.last_error()
    k = count.value
    for (v, d), e, w in zip(keep, exprs, expected):
        got = pa.Array.from_buffers(e.result().type, k, [pa.py_buffer(v), pa.py_buffer(d)])
        assert got.equals(w), (got, w)
    lib.gdv_filter_project_free(fp)
