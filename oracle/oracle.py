"""Python driver of the CPU oracle (oracle/gdv_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of gdv_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by gandiva_amd/.

The oracle consumes the plain-Python description every ``gandiva_amd.Node`` carries
(kind / name / children / value), NOT the native handles: the two evaluations share no code
below the tree description.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "gdv_oracle.c")
_LIB = os.path.join(_HERE, "libgdv_oracle.so")
_lib = None


def build(force=False):
    """gcc -O3 -march=native the restatement into oracle/libgdv_oracle.so."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fno-math-errno", "-ffp-contract=off",
                               "-fPIC", "-shared", "-o", _LIB, _SRC, "-lm", "-lpthread"])
    return _LIB


class _Column(C.Structure):
    _fields_ = [("type", C.c_int32), ("validity", C.c_void_p), ("data", C.c_void_p),
                ("offset", C.c_int64), ("precision", C.c_int32), ("scale", C.c_int32),
                ("offsets", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        try:
            l = C.CDLL(build())
        except OSError:
            # a library built on another host with -march=native: rebuild here
            l = C.CDLL(build(force=True))
        l.gdv_oracle_project.restype = C.c_int
        l.gdv_oracle_project.argtypes = [C.c_char_p, C.POINTER(_Column), C.c_int, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_int]
        l.gdv_oracle_bitmap_to_selection.restype = C.c_int64
        l.gdv_oracle_bitmap_to_selection.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                     C.c_void_p, C.c_int64]
        l.gdv_oracle_force_generic.restype = None
        l.gdv_oracle_force_generic.argtypes = [C.c_int]
        l.gdv_oracle_project_str.restype = C.c_int64
        l.gdv_oracle_project_str.argtypes = [C.c_char_p, C.POINTER(_Column), C.c_int, C.c_int64,
                                             C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        _lib = l
    return _lib


_TYPE_IDS = [
    (pa.types.is_boolean, 1), (pa.types.is_uint8, 2), (pa.types.is_int8, 3), (pa.types.is_uint16, 4),
    (pa.types.is_int16, 5), (pa.types.is_uint32, 6), (pa.types.is_int32, 7), (pa.types.is_uint64, 8),
    (pa.types.is_int64, 9), (pa.types.is_float32, 11), (pa.types.is_float64, 12),
    (pa.types.is_string, 13), (pa.types.is_binary, 14),
    (pa.types.is_date32, 16), (pa.types.is_date64, 17), (pa.types.is_timestamp, 18),
    (pa.types.is_time32, 19), (pa.types.is_time64, 20), (pa.types.is_decimal128, 23),
]
_PACK = {1: "<B", 2: "<B", 3: "<b", 4: "<H", 5: "<h", 6: "<I", 7: "<i", 8: "<Q", 9: "<q",
         11: "<f", 12: "<d", 16: "<i", 17: "<q", 18: "<q", 19: "<i", 20: "<q"}


def type_id(t):
    for pred, tid in _TYPE_IDS:
        if pred(t):
            return tid
    raise NotImplementedError(f"oracle: type {t} not restated")


def _bits(tid, value):
    if tid == 23:
        return int(value) & ((1 << 128) - 1)
    raw = struct.pack(_PACK[tid], value)
    return int.from_bytes(raw, "little")


def type_token(t):
    tid = type_id(t)
    return f"{tid}:{t.precision}:{t.scale}" if tid == 23 else str(tid)


def serialize(node, schema):
    """gandiva_amd.Node description -> oracle program text."""
    k = node.kind
    if k == "field":
        return f"F {schema.get_field_index(node.desc['name'])}"
    if k == "literal":
        tid = type_id(node.dtype)
        tok = type_token(node.dtype)
        if tid in (13, 14):
            raw = b"" if node.desc["is_null"] else node.desc["value"]
            return f"S {tok} {int(node.desc['is_null'])} {len(raw)} {raw.hex() or '-'}"
        if node.desc["is_null"]:
            return f"L {tok} 1 0 0"
        bits = _bits(tid, node.desc['value'])
        return f"L {tok} 0 {bits & ((1 << 64) - 1):x} {bits >> 64:x}"
    if k == "function":
        kids = node.desc["children"]
        return " ".join([f"C {node.desc['name']} {type_token(node.dtype)} {len(kids)}"] +
                        [serialize(c, schema) for c in kids])
    if k == "if":
        return " ".join([f"I {type_token(node.dtype)}"] + [serialize(c, schema) for c in node.desc["children"]])
    if k in ("and", "or"):
        kids = node.desc["children"]
        return " ".join([f"{'A' if k == 'and' else 'O'} {len(kids)}"] + [serialize(c, schema) for c in kids])
    if k == "in" and type_id(node.desc["value_type"]) in (13, 14):
        vals = node.desc["values"]
        parts = [f"M {len(vals)}"]
        for v in vals:
            parts += [str(len(v)), v.hex() or "-"]
        return " ".join(parts + [serialize(node.desc["children"][0], schema)])
    if k == "in":
        tid = type_id(node.desc["value_type"])
        vals = [f"{_bits(tid, v):x}" for v in node.desc["values"]]
        return " ".join([f"N {tid} {len(vals)}"] + vals + [serialize(node.desc["children"][0], schema)])
    raise NotImplementedError(k)


def _columns(batch):
    cols = (_Column * max(batch.num_columns, 1))()
    keep = []
    for i, arr in enumerate(batch.columns):
        try:
            tid = type_id(arr.type)
        except NotImplementedError:
            tid = 0  # unreferenced columns of unsupported types are fine
        bufs = arr.buffers()
        cols[i].type = tid
        cols[i].validity = bufs[0].address if bufs[0] is not None else None
        if tid in (13, 14):
            cols[i].offsets = bufs[1].address
            cols[i].data = bufs[2].address if bufs[2] is not None else None
        else:
            cols[i].data = bufs[1].address if len(bufs) > 1 and bufs[1] is not None else None
        cols[i].offset = arr.offset
        if tid == 23:
            cols[i].precision, cols[i].scale = arr.type.precision, arr.type.scale
        keep.append(bufs)
    return cols, keep


class OracleError(Exception):
    pass


def _raise(err):
    if err & 1:
        raise OracleError("divide by zero error")
    if err & 4:
        raise OracleError("invalid argument")
    raise OracleError(f"oracle error bits {err:#x}")


def cast_indefinite(on):
    """float -> integer casts of NaN / out-of-range values: the x86 "indefinite integer" (True) or saturation (False, default)."""
    lib().gdv_oracle_cast_indefinite(int(bool(on)))


def force_generic(on):
    """Disable (True) / enable (False) the float64 fast path: for the agreement test."""
    lib().gdv_oracle_force_generic(int(bool(on)))


def project_one(root, result_type, batch, threads=1, out=None):
    """Evaluate one expression tree over the batch -> pyarrow.Array (host).
    `out` = (validity uint8 array, data uint8 array) to reuse across calls (timing runs):
    they are re-zeroed here, so repeated passes do not pay first-touch page faults."""
    n = batch.num_rows
    tid = type_id(result_type)
    if tid in (13, 14):
        return _project_str(root, result_type, batch)
    width = 0 if tid == 1 else result_type.bit_width // 8
    vbytes = (n + 7) // 8
    if out is None:
        validity = np.zeros(max(vbytes, 1), dtype=np.uint8)
        data = np.zeros(max(vbytes if tid == 1 else n * width, 1), dtype=np.uint8)
    else:
        validity, data = out
        validity[:] = 0
        if tid == 1:
            data[:] = 0
    cols, keep = _columns(batch)
    prog = serialize(root, batch.schema).encode()
    err = lib().gdv_oracle_project(prog, cols, batch.num_columns, n, data.ctypes.data,
                                   validity.ctypes.data, threads)
    if err:
        _raise(err)
    return pa.Array.from_buffers(result_type, n, [pa.py_buffer(validity), pa.py_buffer(data)])


def _project_str(root, result_type, batch):
    n = batch.num_rows
    cols, keep = _columns(batch)
    prog = serialize(root, batch.schema).encode()
    cap = 1 << 16
    while True:
        validity = np.zeros(max((n + 7) // 8, 1), dtype=np.uint8)
        offsets = np.zeros(n + 1, dtype=np.int32)
        data = np.zeros(max(cap, 1), dtype=np.uint8)
        total = lib().gdv_oracle_project_str(prog, cols, batch.num_columns, n, offsets.ctypes.data,
                                             data.ctypes.data, cap, validity.ctypes.data)
        if total <= -0x1000:
            _raise(-total - 0x1000)
        if total < 0:
            raise OracleError("oracle string evaluation failed")
        if total <= cap:
            break
        cap = int(total)
    return pa.Array.from_buffers(result_type, n, [pa.py_buffer(validity), pa.py_buffer(offsets),
                                                  pa.py_buffer(data[:max(total, 0)])])


def project(expressions, batch, threads=1, out=None):
    """Reference execution shape: one pass over the batch PER expression."""
    return [project_one(e.root(), e.result().type, batch, threads, None if out is None else out[i])
            for i, e in enumerate(expressions)]


def alloc_outputs(expressions, n):
    """Pre-touched output buffers for repeated timing passes of `project`."""
    outs = []
    for e in expressions:
        t = e.result().type
        tid = type_id(t)
        vbytes = max((n + 7) // 8, 1)
        data = np.ones(max(vbytes if tid == 1 else n * (t.bit_width // 8), 1), dtype=np.uint8)
        outs.append((np.ones(vbytes, dtype=np.uint8), data))
    return outs


def filter_indices(condition, batch, dtype="int32", threads=1):
    """Filter::Evaluate restated: condition -> (value, validity) bitmaps -> ascending indices."""
    res = project_one(condition.root(), pa.bool_(), batch, threads)
    n = batch.num_rows
    t = pa.type_for_alias(dtype) if isinstance(dtype, str) else dtype
    w = t.bit_width // 8
    np_t = {2: np.uint16, 4: np.uint32, 8: np.uint64}[w]
    out = np.zeros(max(n, 1), dtype=np_t)
    validity, bits = res.buffers()
    k = lib().gdv_oracle_bitmap_to_selection(bits.address, validity.address, n, w,
                                             out.ctypes.data, n)
    if k < 0:
        raise OracleError("selection vector too small")
    return pa.array(out[:k], type={2: pa.uint16(), 4: pa.uint32(), 8: pa.uint64()}[w])


def take_rows(batch, indices):
    """Rows a selection vector picks (for Projector-with-selection parity)."""
    return batch.take(pa.array(np.asarray(indices).astype(np.int64)))
