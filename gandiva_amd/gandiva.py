"""Python mirror of ``pyarrow.gandiva`` on top of the gandiva_amd C ABI.

Same names, argument meaning and error behaviour as the reference lineage's Python binding
(pyarrow/gandiva.pyx: TreeExprBuilder :296-627, make_projector :629-674, make_filter
:677-705, Projector.evaluate :199-226, Filter.evaluate :247-280, SelectionVector :127-160,
Configuration :707-742, get_registered_function_signatures :745-764), so that parity tests
read like pyarrow/tests/test_gandiva.py.  All evaluation happens in HIP kernels behind
``gdv_projector_evaluate`` / ``gdv_filter_evaluate``; this module only marshals Arrow
buffers.  Two extra entry points expose the HBM-resident path the benchmark measures:
``DeviceBatch`` and ``Projector.evaluate_device`` / ``Filter.evaluate_device``.
"""
import ctypes as C
import struct

import numpy as np
import pyarrow as pa

from . import _capi
from ._capi import gdv_type_t, gdv_column_t, gdv_out_column_t, gdv_selection_t, gdv_config_t

GDV_MEM_HOST, GDV_MEM_DEVICE = 0, 1
GDV_EVAL_ASYNC = 1

_TIME_UNITS = {"s": 0, "ms": 1, "us": 2, "ns": 3}
_TIME_UNIT_NAMES = {v: k for k, v in _TIME_UNITS.items()}


class GandivaError(pa.lib.ArrowException):
    """CodeGenError / ExpressionValidationError / ExecutionError (arrow status 40/41/42)."""


def _raise_status(code):
    msg = _capi.last_error()
    if code == 4:
        raise pa.ArrowInvalid(msg)
    if code == 10:
        raise pa.ArrowNotImplementedError(msg)
    if code == 1:
        raise pa.ArrowMemoryError(msg)
    raise GandivaError(msg)


def _check(code):
    if code != 0:
        _raise_status(code)


# ---------------------------------------------------------------------------- types

def ensure_type(dtype):
    if dtype is None:
        raise TypeError("dtype must not be None")
    if isinstance(dtype, pa.DataType):
        return dtype
    if isinstance(dtype, str):
        return pa.type_for_alias(dtype)
    raise TypeError(f"cannot interpret {dtype!r} as a DataType")


def to_gdv_type(t):
    t = ensure_type(t)
    if pa.types.is_boolean(t): return gdv_type_t(1, 0, 0)
    if pa.types.is_uint8(t): return gdv_type_t(2, 0, 0)
    if pa.types.is_int8(t): return gdv_type_t(3, 0, 0)
    if pa.types.is_uint16(t): return gdv_type_t(4, 0, 0)
    if pa.types.is_int16(t): return gdv_type_t(5, 0, 0)
    if pa.types.is_uint32(t): return gdv_type_t(6, 0, 0)
    if pa.types.is_int32(t): return gdv_type_t(7, 0, 0)
    if pa.types.is_uint64(t): return gdv_type_t(8, 0, 0)
    if pa.types.is_int64(t): return gdv_type_t(9, 0, 0)
    if pa.types.is_float32(t): return gdv_type_t(11, 0, 0)
    if pa.types.is_float64(t): return gdv_type_t(12, 0, 0)
    if pa.types.is_string(t): return gdv_type_t(13, 0, 0)
    if pa.types.is_binary(t): return gdv_type_t(14, 0, 0)
    if pa.types.is_date32(t): return gdv_type_t(16, 0, 0)
    if pa.types.is_date64(t): return gdv_type_t(17, 0, 0)
    if pa.types.is_timestamp(t): return gdv_type_t(18, _TIME_UNITS[t.unit], 0)
    if pa.types.is_time32(t): return gdv_type_t(19, _TIME_UNITS[t.unit], 0)
    if pa.types.is_time64(t): return gdv_type_t(20, _TIME_UNITS[t.unit], 0)
    if pa.types.is_decimal128(t): return gdv_type_t(23, t.precision, t.scale)
    raise pa.ArrowNotImplementedError(f"type {t} is not supported by gandiva_amd")


def from_gdv_type(g):
    simple = {1: pa.bool_(), 2: pa.uint8(), 3: pa.int8(), 4: pa.uint16(), 5: pa.int16(),
              6: pa.uint32(), 7: pa.int32(), 8: pa.uint64(), 9: pa.int64(), 11: pa.float32(),
              12: pa.float64(), 13: pa.string(), 14: pa.binary(), 16: pa.date32(),
              17: pa.date64()}
    if g.id in simple: return simple[g.id]
    if g.id == 18: return pa.timestamp(_TIME_UNIT_NAMES[g.precision])
    if g.id == 19: return pa.time32(_TIME_UNIT_NAMES[g.precision])
    if g.id == 20: return pa.time64(_TIME_UNIT_NAMES[g.precision])
    if g.id == 23: return pa.decimal128(g.precision, g.scale)
    raise ValueError(f"unknown gdv type id {g.id}")


_FIXED_PACK = {1: "<B", 2: "<B", 3: "<b", 4: "<H", 5: "<h", 6: "<I", 7: "<i", 8: "<Q", 9: "<q",
               11: "<f", 12: "<d", 16: "<i", 17: "<q", 18: "<q", 19: "<i", 20: "<q"}


def _pack_fixed(gt, value):
    if gt.id == 23:
        v = int(value) & ((1 << 128) - 1)
        return v.to_bytes(16, "little")
    return struct.pack(_FIXED_PACK[gt.id], value)


# ---------------------------------------------------------------------------- tree nodes

class Node:
    """A node of an expression tree (gandiva::Node).  Keeps a plain-Python description
    (kind / name / children / dtype / value) next to the native handle; tests hand that
    description to the CPU oracle, the library never reads it."""

    def __init__(self, handle, kind, dtype, **desc):
        if not handle:
            raise GandivaError(_capi.last_error())
        self._h = handle
        self.kind = kind
        self.dtype = dtype
        self.desc = desc

    def __del__(self):
        try:
            _capi.lib().gdv_node_free(self._h)
        except Exception:
            pass

    def __str__(self):
        return _capi.take_string(_capi.lib().gdv_node_to_string(self._h))

    def return_type(self):
        return from_gdv_type(_capi.lib().gdv_node_return_type(self._h))


class Expression:
    def __init__(self, handle, root, result_field):
        if not handle:
            raise GandivaError(_capi.last_error())
        self._h = handle
        self._root = root
        self._result = result_field

    def __del__(self):
        try:
            _capi.lib().gdv_expression_free(self._h)
        except Exception:
            pass

    def __str__(self):
        return _capi.take_string(_capi.lib().gdv_expression_to_string(self._h))

    def root(self):
        return self._root

    def result(self):
        return self._result


class Condition(Expression):
    pass


def _node_array(nodes):
    for n in nodes:
        if not isinstance(n, Node):
            raise TypeError(f"expected a gandiva Node, got {type(n).__name__}")
    arr = (C.c_void_p * max(len(nodes), 1))(*[n._h for n in nodes])
    return arr


class TreeExprBuilder:
    """Mirror of pyarrow.gandiva.TreeExprBuilder (gandiva::TreeExprBuilder,
    libgandiva.pxd:110-212)."""

    def make_literal(self, value, dtype):
        t = ensure_type(dtype)
        gt = to_gdv_type(t)
        lib = _capi.lib()
        if gt.id in (13, 14):
            if gt.id == 13:
                if not isinstance(value, str):
                    raise TypeError(f"expected str for {t}, got {type(value).__name__}")
                raw = value.encode("utf-8")
            else:
                if not isinstance(value, (bytes, bytearray)):
                    raise TypeError(f"expected bytes for {t}, got {type(value).__name__}")
                raw = bytes(value)
            h = lib.gdv_node_literal_bytes(gt, raw, len(raw), 0)
            return Node(h, "literal", t, value=raw, is_null=False)
        if gt.id == 1:
            if not isinstance(value, (bool, np.bool_)):
                raise TypeError(f"expected bool for {t}, got {type(value).__name__}")
        elif gt.id in (11, 12):
            if isinstance(value, (str, bytes)) or not isinstance(value, (int, float, np.number)):
                raise TypeError(f"expected a number for {t}, got {type(value).__name__}")
            value = float(value)
        elif gt.id == 23:
            import decimal
            if isinstance(value, decimal.Decimal):
                value = int(value.scaleb(t.scale).to_integral_exact())
            elif not isinstance(value, (int, np.integer)) or isinstance(value, bool):
                raise TypeError(f"expected a Decimal or an unscaled integer for {t}")
            value = int(value)
        else:
            if isinstance(value, (bool, str, bytes, float)) and not isinstance(value, (int, np.integer)):
                raise TypeError(f"expected an integer for {t}, got {type(value).__name__}")
            if not isinstance(value, (int, np.integer)):
                raise TypeError(f"expected an integer for {t}, got {type(value).__name__}")
            value = int(value)
        try:
            raw = _pack_fixed(gt, value)
        except struct.error as e:
            raise TypeError(str(e))
        h = lib.gdv_node_literal(gt, raw, 0)
        return Node(h, "literal", t, value=value, is_null=False)

    def make_null(self, dtype):
        t = ensure_type(dtype)
        gt = to_gdv_type(t)
        lib = _capi.lib()
        if gt.id in (13, 14):
            h = lib.gdv_node_literal_bytes(gt, None, 0, 1)
        else:
            h = lib.gdv_node_literal(gt, None, 1)
        return Node(h, "literal", t, value=None, is_null=True)

    def make_field(self, field):
        if not isinstance(field, pa.Field):
            raise TypeError("make_field expects a pyarrow.Field")
        h = _capi.lib().gdv_node_field(field.name.encode(), to_gdv_type(field.type))
        return Node(h, "field", field.type, name=field.name)

    def make_function(self, name, children, return_type):
        children = list(children)
        arr = _node_array(children)
        t = ensure_type(return_type)
        h = _capi.lib().gdv_node_function(name.encode(), arr, len(children), to_gdv_type(t))
        return Node(h, "function", t, name=name, children=children)

    def make_if(self, condition, this_node, else_node, return_type):
        for n in (condition, this_node, else_node):
            if not isinstance(n, Node):
                raise TypeError("make_if expects gandiva Nodes")
        t = ensure_type(return_type)
        h = _capi.lib().gdv_node_if(condition._h, this_node._h, else_node._h, to_gdv_type(t))
        return Node(h, "if", t, children=[condition, this_node, else_node])

    def make_and(self, children):
        children = list(children)
        arr = _node_array(children)
        return Node(_capi.lib().gdv_node_and(arr, len(children)), "and", pa.bool_(),
                    children=children)

    def make_or(self, children):
        children = list(children)
        arr = _node_array(children)
        return Node(_capi.lib().gdv_node_or(arr, len(children)), "or", pa.bool_(),
                    children=children)

    def make_in_expression(self, node, values, dtype):
        if not isinstance(node, Node):
            raise TypeError("make_in_expression expects a gandiva Node")
        t = ensure_type(dtype)
        gt = to_gdv_type(t)
        values = list(values)
        lib = _capi.lib()
        if gt.id in (13, 14):
            raws = [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in values]
            arr = (C.c_char_p * max(len(raws), 1))(*raws)
            lens = (C.c_int64 * max(len(raws), 1))(*[len(r) for r in raws])
            h = lib.gdv_node_in_bytes(node._h, gt, arr, lens, len(raws))
            return Node(h, "in", pa.bool_(), children=[node], values=raws, value_type=t)
        ints = [_to_storage_int(v, t) for v in values]
        raw = b"".join(_pack_fixed(gt, v) for v in ints)
        h = lib.gdv_node_in(node._h, gt, raw, len(ints))
        return Node(h, "in", pa.bool_(), children=[node], values=ints, value_type=t)

    def make_expression(self, root_node, return_field):
        if not isinstance(root_node, Node):
            raise TypeError("make_expression expects a gandiva Node")
        if not isinstance(return_field, pa.Field):
            raise TypeError("make_expression expects a pyarrow.Field")
        h = _capi.lib().gdv_expression_new(root_node._h, return_field.name.encode(),
                                           to_gdv_type(return_field.type))
        return Expression(h, root_node, return_field)

    def make_condition(self, condition):
        if not isinstance(condition, Node):
            raise TypeError("make_condition expects a gandiva Node")
        h = _capi.lib().gdv_condition_new(condition._h)
        return Condition(h, condition, pa.field("cond", pa.bool_()))


def _to_storage_int(v, t):
    """IN-list value -> the integer stored in the Arrow values buffer."""
    if pa.types.is_decimal(t):
        import decimal
        return int(decimal.Decimal(v).scaleb(t.scale).to_integral_exact())
    if pa.types.is_floating(t):
        return float(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    return int(pa.scalar(v, type=t).cast(pa.int64() if t.bit_width == 64 else pa.int32()).as_py())


# ---------------------------------------------------------------------------- configuration

class Configuration:
    def __init__(self, optimize=True, dump_ir=False):
        self.optimize = bool(optimize)
        self.dump_ir = bool(dump_ir)

    def _c(self):
        return gdv_config_t(int(self.optimize), int(self.dump_ir))


_SEL_MODES = {"NONE": 0, "UINT16": 1, "UINT32": 2, "UINT64": 3}
_SEL_DTYPE = {1: (pa.uint16(), np.uint16), 2: (pa.uint32(), np.uint32), 3: (pa.uint64(), np.uint64)}


def _selection_mode(name):
    up = str(name).upper()
    if up not in _SEL_MODES:
        raise ValueError(f"Invalid value for Selection Mode: {name!r}")
    return _SEL_MODES[up]


def _make_schema(schema):
    if not isinstance(schema, pa.Schema):
        raise TypeError("expected a pyarrow.Schema")
    lib = _capi.lib()
    h = lib.gdv_schema_new()
    for f in schema:
        _check(lib.gdv_schema_add_field(h, f.name.encode(), to_gdv_type(f.type), int(f.nullable)))
    return h


# ---------------------------------------------------------------------------- batches

def _buf(b):
    return (b.address, b.size) if b is not None else (None, 0)


def _column_of_array(arr):
    """pyarrow.Array (host) -> gdv_column_t, plus the objects that must stay alive."""
    bufs = arr.buffers()
    col = gdv_column_t()
    col.validity, col.validity_size = _buf(bufs[0])
    if pa.types.is_string(arr.type) or pa.types.is_binary(arr.type):
        col.offsets, col.offsets_size = _buf(bufs[1])
        col.data, col.data_size = _buf(bufs[2])
        if col.data is None:
            col.data, col.data_size = 0, 0
    else:
        col.data, col.data_size = _buf(bufs[1])
    col.offset = arr.offset
    return col


class DeviceColumn:
    """One HBM-resident Arrow array: torch uint8 tensors own the buffers."""

    def __init__(self, type, length, validity, data, offsets=None, offset=0):
        self.type, self.length = type, length
        self.validity, self.data, self.offsets, self.offset = validity, data, offsets, offset
        # Outputs of a selection-mode evaluation whose slot count was still on the device: `length` is
        # the CAPACITY the buffers were sized for, rows past the real count were never written.  The
        # count tensor travels with the column; `num_rows` / `to_arrow` read it (and wait) when asked.
        self.count_tensor = None

    @property
    def num_rows(self):
        """Rows that hold results (reads the device-resident count — and waits — if there is one)."""
        if self.count_tensor is not None:
            self.length = min(self.length, int(self.count_tensor.item()))
            self.count_tensor = None
        return self.length

    def _c(self):
        col = gdv_column_t()
        if self.validity is not None:
            col.validity, col.validity_size = self.validity.data_ptr(), self.validity.numel()
        if self.data is not None:
            col.data, col.data_size = self.data.data_ptr(), self.data.numel()
        if self.offsets is not None:
            col.offsets, col.offsets_size = self.offsets.data_ptr(), self.offsets.numel()
        col.offset = self.offset
        return col

    def to_arrow(self):
        """Copy back to a host pyarrow.Array."""
        bufs = [None if self.validity is None else pa.py_buffer(self.validity.cpu().numpy())]
        if self.offsets is not None:
            bufs.append(pa.py_buffer(self.offsets.cpu().numpy()))
        used = getattr(self, "data_used", None)
        pending = getattr(self, "result", None)
        if pending is not None and self.offsets is not None:
            # an asynchronous evaluation left the byte total on the device (gdv_projector_evaluate_async)
            if int(pending[0]) != 0:
                raise GandivaError("the asynchronous evaluation did not complete these outputs (status "
                                   f"{int(pending[0])}): evaluate the batch with evaluate_device")
            used = int(pending[1 + self.result_index])
            if used > self.data.numel():
                raise GandivaError(f"var-len output needs {used} bytes, the buffer holds {self.data.numel()}")
        data = self.data if used is None else self.data[:used]
        bufs.append(pa.py_buffer(data.cpu().numpy()))
        return pa.Array.from_buffers(self.type, self.num_rows, bufs, offset=self.offset)


def _pad64(n):
    return (n + 63) // 64 * 64


class HostArena:
    """Page-locked host memory the GPUs address directly (``gdv_host_alloc``), handed out as pyarrow
    buffers by a bump allocator.  A host ``evaluate`` whose batch was placed with ``place`` and whose outputs
    come from the arena (``evaluate(batch, arena=...)``) copies nothing: the kernel reads and writes the
    arrays where they are.  ``HostArena.over(ndarray)`` registers memory the caller owns instead
    (``gdv_host_register``).  Buffers keep the arena alive; ``reset()`` recycles the space once they are gone."""

    def __init__(self, nbytes, _foreign=None):
        lib = _capi.lib()
        self._foreign = _foreign
        self._size, self._used, self._base = int(nbytes), 0, None
        if _foreign is None:
            p = C.c_void_p()
            _check(lib.gdv_host_alloc(self._size, C.byref(p)))
            self._base = p.value
        else:
            _check(lib.gdv_host_register(C.c_void_p(_foreign.ctypes.data), self._size))
            self._base = _foreign.ctypes.data

    @staticmethod
    def over(array):
        """Register a writable, contiguous numpy array the caller owns; unregistered when the arena goes."""
        return HostArena(array.nbytes, _foreign=array)

    def __del__(self):
        if getattr(self, "_base", None) is not None:
            try:
                lib = _capi.lib()
                if self._foreign is None:
                    lib.gdv_host_free(C.c_void_p(self._base))
                else:
                    lib.gdv_host_unregister(C.c_void_p(self._base))
            except Exception:   # (interpreter shutdown: the library may be gone already)
                pass
            self._base = None

    def reset(self):
        self._used = 0

    def allocate(self, nbytes):
        """A 64-byte aligned, 64-byte padded pyarrow buffer of ``nbytes`` bytes inside the arena."""
        at = (self._base + self._used + 63) // 64 * 64 - self._base
        # (16 zeroed bytes behind every buffer: the string kernels sweep var-len bytes in whole 16-byte pieces —
        # in place, since round 5 — and what follows the last byte must not look like text)
        room = _pad64(max(int(nbytes), 1) + 16)
        if at + room > self._size:
            raise MemoryError(f"HostArena: {nbytes} bytes do not fit ({self._size - at} left)")
        self._used = at + room
        C.memset(self._base + at + int(nbytes), 0, room - int(nbytes))
        return pa.foreign_buffer(self._base + at, int(nbytes), base=self)

    def place(self, batch):
        """A copy of the record batch (flat columns) whose buffers live in the arena."""
        cols = []
        for arr in batch.columns:
            bufs = []
            for b in arr.buffers():
                if b is None:
                    bufs.append(None)
                    continue
                nb = self.allocate(_pad64(b.size))
                C.memmove(nb.address, b.address, b.size)
                bufs.append(nb.slice(0, b.size))
            cols.append(pa.Array.from_buffers(arr.type, len(arr), bufs, arr.null_count, arr.offset))
        return pa.RecordBatch.from_arrays(cols, schema=batch.schema)


def host_staged_bytes():
    """``gdv_host_staged_bytes``: bytes host evaluations have copied through staging blocks so far."""
    return int(_capi.lib().gdv_host_staged_bytes())


class _PoolBlock:
    """`nbytes` of a DevicePool buffer, seen by torch through __cuda_array_interface__ (zero-copy)."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class DevicePool:
    """gdv_device_pool_*: library-owned, placement-aware HBM for buffers that are streamed together (include/gandiva_amd.h).
    ``reserve_outputs`` hands a Projector's output columns for batches of ``rows`` rows out of the best of several
    candidate placements (probed with a write sweep by the library) and ``release`` puts them back INTO THE POOL, where
    the next ``reserve_outputs`` of the same shape finds them: a good placement, found once, is kept."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_capi.lib().gdv_device_pool_create(C.byref(self._h)))
        self.last_probe = None

    def close(self):
        if self._h:
            _capi.lib().gdv_device_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _tensor(self, ptr, nbytes):
        import torch
        return torch.as_tensor(_PoolBlock(ptr, nbytes), device="cuda")

    def reserve_set(self, count, nbytes, candidates=8):
        """-> (pointers, {"rates_gbs": [...], "kept": index}): `count` buffers of `nbytes` each, the best of up to `candidates` placements."""
        ptrs = (C.c_void_p * count)()
        rates = (C.c_double * max(candidates, 1))()
        tried, kept = C.c_int(0), C.c_int(0)
        _check(_capi.lib().gdv_device_pool_reserve_set(self._h, count, nbytes, candidates, ptrs, rates, C.byref(tried), C.byref(kept)))
        return [int(p) for p in ptrs], {"rates_gbs": [round(rates[i], 1) for i in range(tried.value)], "kept": kept.value}

    def alloc(self, nbytes):
        p = C.c_void_p()
        _check(_capi.lib().gdv_device_pool_alloc(self._h, nbytes, C.byref(p)))
        return int(p.value)

    def free(self, ptr):
        _check(_capi.lib().gdv_device_pool_free(self._h, C.c_void_p(ptr)))

    def trim(self):
        _check(_capi.lib().gdv_device_pool_trim(self._h))

    def bytes_held(self):
        used = C.c_int64(0)
        total = _capi.lib().gdv_device_pool_bytes(self._h, C.byref(used))
        return total, used.value

    def reserve_outputs(self, projector, rows, candidates=8):
        """DeviceColumns for every (fixed-width / bool) output of ``projector`` over ``rows`` rows.  Outputs of one size
        are reserved as ONE set (they are written together); the validity bitmaps, a 64th of the traffic, are plain pool
        allocations.  ``last_probe`` holds what the library measured."""
        lib = _capi.lib()
        sizes = []
        for i, t in enumerate(projector._out_types):
            if pa.types.is_string(t) or pa.types.is_binary(t):
                raise TypeError("DevicePool.reserve_outputs: fixed-width outputs only")
            vb, db = C.c_int64(), C.c_int64()
            _check(lib.gdv_projector_output_sizes(projector._h, i, rows, GDV_MEM_DEVICE, vb, db))
            sizes.append((_pad64(max(vb.value, 1)), _pad64(max(db.value, 1))))
        data_ptr = [None] * len(sizes)
        self.last_probe = []
        for nbytes in sorted({d for _, d in sizes}, reverse=True):
            members = [i for i, (_, d) in enumerate(sizes) if d == nbytes]
            for lo in range(0, len(members), 32):
                group = members[lo:lo + 32]
                ptrs, probe = self.reserve_set(len(group), nbytes, candidates)
                probe["buffers"], probe["bytes_each"] = len(group), nbytes
                self.last_probe.append(probe)
                for i, p in zip(group, ptrs):
                    data_ptr[i] = p
        cols = []
        for i, t in enumerate(projector._out_types):
            vbytes, dbytes = sizes[i]
            col = DeviceColumn(t, rows, self._tensor(self.alloc(vbytes), vbytes), self._tensor(data_ptr[i], dbytes))
            col._pool_ptrs = (col.validity.data_ptr(), data_ptr[i])
            cols.append(col)
        return cols

    def release(self, columns):
        """The columns' buffers go back into the pool (they stay allocated there).  The tensors must not be used afterwards."""
        for c in columns:
            for p in getattr(c, "_pool_ptrs", ()):
                self.free(p)
            c._pool_ptrs = ()


class DeviceBatch:
    """A record batch resident in the HBM of the current device (Arrow layout, buffers
    padded to 64 bytes).  Build with ``DeviceBatch.from_arrow`` (uploads) or directly from
    DeviceColumns produced on the GPU."""

    def __init__(self, schema, columns, num_rows):
        self.schema, self.columns, self.num_rows = schema, columns, num_rows

    @staticmethod
    def from_arrow(batch, device="cuda"):
        import torch
        cols = []
        for arr in batch.columns:
            bufs = arr.buffers()

            def up(b):
                if b is None:
                    return None
                host = np.frombuffer(b, dtype=np.uint8)
                t = torch.zeros(_pad64(max(host.size, 1)), dtype=torch.uint8, device=device)
                if host.size:
                    t[:host.size] = torch.from_numpy(host.copy()).to(device)
                return t
            if pa.types.is_string(arr.type) or pa.types.is_binary(arr.type):
                cols.append(DeviceColumn(arr.type, len(arr), up(bufs[0]), up(bufs[2]) if bufs[2] is not None else torch.zeros(64, dtype=torch.uint8, device=device), up(bufs[1]), arr.offset))
            else:
                cols.append(DeviceColumn(arr.type, len(arr), up(bufs[0]), up(bufs[1]), None, arr.offset))
        return DeviceBatch(batch.schema, cols, batch.num_rows)


class SelectionVector:
    """gandiva::SelectionVector (libgandiva.pxd:43-71): host numpy indices or, on the
    device path, a torch tensor of indices plus the slot count."""

    def __init__(self, mode, indices, num_slots, device=False, count_tensor=None):
        self.mode, self.indices, self.device = mode, indices, device
        self._num_slots = num_slots
        # asynchronous device filter: the slot count is still in HBM (torch int64 tensor of one element)
        self.count_tensor = count_tensor

    @property
    def num_slots(self):
        """The slot count (reads it back from the device — and waits — if the filter was asynchronous)."""
        if self._num_slots is None:
            self._num_slots = int(self.count_tensor.item())
        return self._num_slots

    @property
    def pending(self):
        """True while the slot count has not been brought to the host."""
        return self._num_slots is None

    def to_array(self):
        atype, ntype = _SEL_DTYPE[self.mode]
        if self.device:
            host = self.indices[:self.num_slots].cpu().numpy()
        else:
            host = self.indices[:self.num_slots]
        return pa.array(host.astype(ntype, copy=False), type=atype)

    def _c(self):
        s = gdv_selection_t()
        s.mode = self.mode
        s.num_slots = self.num_slots
        s.indices = self.indices.data_ptr() if self.device else self.indices.ctypes.data
        return s


def _check_batch(batch, schema):
    if not isinstance(batch, pa.RecordBatch):
        raise TypeError("expected a pyarrow.RecordBatch")
    if not batch.schema.equals(schema, check_metadata=False):
        raise pa.ArrowInvalid("Schema in RecordBatch must match schema in Make()")


# ---------------------------------------------------------------------------- Projector

class Projector:
    def __init__(self, handle, schema, mode, exprs):
        self._h = handle
        self._schema = schema
        self._mode = mode
        self._exprs = exprs
        lib = _capi.lib()
        self._out_types = [from_gdv_type(lib.gdv_projector_output_type(handle, i))
                           for i in range(lib.gdv_projector_num_outputs(handle))]

    def __del__(self):
        try:
            _capi.lib().gdv_projector_free(self._h)
        except Exception:
            pass

    @property
    def llvm_ir(self):
        """DumpIR(): the generated HIP source of the fused kernel (contains `@expr_N`)."""
        return _capi.take_string(_capi.lib().gdv_projector_dump_ir(self._h))

    @property
    def path_hint(self):
        """gdv_projector_path_hint: 0 optimistic kernels, 1 exact wave variant, 2 scanner-shaped general kernel."""
        return _capi.lib().gdv_projector_path_hint(self._h)

    def evaluate(self, batch, selection=None, arena=None):
        """Host-buffer path: stages the batch through HBM, returns host pyarrow arrays.  ``arena`` (a
        ``HostArena``): the outputs are allocated there and written by the kernel in place."""
        alloc = pa.allocate_buffer if arena is None else arena.allocate
        _check_batch(batch, self._schema)
        lib = _capi.lib()
        cols = (gdv_column_t * max(batch.num_columns, 1))(*[_column_of_array(a) for a in batch.columns])
        out_rows = selection.num_slots if selection is not None else batch.num_rows
        n_out = len(self._out_types)
        outs = (gdv_out_column_t * n_out)()
        holders = [None] * n_out
        varlen = [pa.types.is_string(t) or pa.types.is_binary(t) for t in self._out_types]
        # first guess for var-len byte capacity: the bytes of all var-len inputs
        guess = 64 + sum(a.buffers()[2].size for a in batch.columns
                         if (pa.types.is_string(a.type) or pa.types.is_binary(a.type))
                         and a.buffers()[2] is not None)
        for i, t in enumerate(self._out_types):
            vb, db = C.c_int64(), C.c_int64()
            _check(lib.gdv_projector_output_sizes(self._h, i, out_rows, GDV_MEM_HOST, vb, db))
            v = alloc(_pad64(max(vb.value, 1)))
            d = alloc(_pad64((db.value or guess) if varlen[i] else max(db.value, 1)))
            o = alloc(_pad64((out_rows + 1) * 4)) if varlen[i] else None
            holders[i] = [v, d, o]
            outs[i].validity, outs[i].validity_size = v.address, v.size
            outs[i].data, outs[i].data_size = d.address, d.size
            if o is not None:
                outs[i].offsets, outs[i].offsets_size = o.address, o.size
        sel_c = None
        if selection is not None:
            if selection.device:
                raise TypeError("device selection vector passed to the host evaluate()")
            s = selection._c()
            sel_c = C.byref(s)
        for attempt in range(2):
            caps = [outs[i].data_size for i in range(n_out)]
            rc = lib.gdv_projector_evaluate(self._h, batch.num_rows, cols, batch.num_columns, sel_c,
                                            outs, n_out, GDV_MEM_HOST, None, 0)
            grown = False
            if rc == 4 and attempt == 0:
                # a var-len byte buffer was too small: data_size now holds the bytes needed
                for i in range(n_out):
                    if varlen[i] and outs[i].data_size > caps[i]:
                        d = alloc(_pad64(outs[i].data_size))
                        holders[i][1] = d
                        outs[i].data, outs[i].data_size = d.address, d.size
                        grown = True
                    elif varlen[i]:
                        outs[i].data_size = caps[i]
            if not grown:
                _check(rc)
                break
        result = []
        for i, t in enumerate(self._out_types):
            v, d, o = holders[i]
            if varlen[i]:
                result.append(pa.Array.from_buffers(t, out_rows, [v, o, d.slice(0, outs[i].data_size)]))
            else:
                result.append(pa.Array.from_buffers(t, out_rows, [v, d]))
        return result

    def evaluate_device_array(self, array_address, num_rows, on_device):
        """Evaluate a batch handed over through the Arrow C Device Data Interface:
        `array_address` is the address of a `struct ArrowDeviceArray` (struct array, one child
        per schema field).  ARROW_DEVICE_ROCM arrays are used in place and the outputs are
        DeviceColumns; CPU arrays take the staged path and return pyarrow arrays."""
        return _evaluate_device_array(self, array_address, num_rows, on_device)

    def evaluate_export(self, array_address, selection=None, stream=None):
        """Evaluate a batch handed over as `struct ArrowDeviceArray` and hand the results on
        the same way (Arrow C Device Data Interface export): returns the ctypes structs
        ``(ArrowDeviceArray, ArrowSchema)``; buffers are allocated by the library (HBM for
        ARROW_DEVICE_ROCM inputs, host memory otherwise) and owned by the consumer, who
        releases both through their release callbacks (`_capi.release_c_struct`)."""
        out, schema = _capi.ArrowDeviceArray(), _capi.ArrowSchema()
        sel_c = None
        if selection is not None:
            s = selection._c()
            sel_c = C.byref(s)
        _check(_capi.lib().gdv_projector_evaluate_export(
            self._h, C.c_void_p(array_address), sel_c, C.c_void_p(stream or 0),
            C.addressof(out), C.addressof(schema)))
        return out, schema

    def evaluate_device(self, dbatch, selection=None, outputs=None, stream=None, sync=True):
        """HBM-resident path (zero-copy): inputs are a DeviceBatch, outputs DeviceColumns
        (allocated here unless ``outputs`` from a previous call are passed back in)."""
        import torch
        lib = _capi.lib()
        cols = (gdv_column_t * max(len(dbatch.columns), 1))(*[c._c() for c in dbatch.columns])
        # a selection whose count is still on the device: outputs and launch are sized for its capacity
        pending = selection is not None and selection.device and selection.pending
        if pending:
            out_rows = selection.indices.numel()
        else:
            out_rows = selection.num_slots if selection is not None else dbatch.num_rows
        n_out = len(self._out_types)
        varlen = [pa.types.is_string(t) or pa.types.is_binary(t) for t in self._out_types]
        if outputs is None:
            outputs = []
            guess = 64 + sum(c.data.numel() for c in dbatch.columns if c.offsets is not None)
            for i, t in enumerate(self._out_types):
                vb, db = C.c_int64(), C.c_int64()
                _check(lib.gdv_projector_output_sizes(self._h, i, out_rows, GDV_MEM_DEVICE, vb, db))
                outputs.append(DeviceColumn(
                    t, out_rows,
                    torch.empty(_pad64(max(vb.value, 1)), dtype=torch.uint8, device="cuda"),
                    # (var-len: db = what earlier batches of this projector produced per row, 0 before the first)
                    torch.empty(_pad64((db.value or guess) if varlen[i] else max(db.value, 1)), dtype=torch.uint8,
                                device="cuda"),
                    torch.empty(_pad64((out_rows + 1) * 4), dtype=torch.uint8, device="cuda")
                    if varlen[i] else None))
        outs = (gdv_out_column_t * n_out)()
        for i, o in enumerate(outputs):
            outs[i].validity, outs[i].validity_size = o.validity.data_ptr(), o.validity.numel()
            outs[i].data, outs[i].data_size = o.data.data_ptr(), o.data.numel()
            if o.offsets is not None:
                outs[i].offsets, outs[i].offsets_size = o.offsets.data_ptr(), o.offsets.numel()
        sel_c = None
        if selection is not None:
            s = gdv_selection_t()
            s.mode = selection.mode
            s.num_slots = out_rows
            s.indices = selection.indices.data_ptr() if selection.device else selection.indices.ctypes.data
            sel_c = C.byref(s)
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        for attempt in range(2):
            caps = [outs[i].data_size for i in range(n_out)]
            if pending:
                rc = lib.gdv_projector_evaluate_selected(self._h, dbatch.num_rows, cols, len(dbatch.columns), sel_c,
                                                         C.c_void_p(selection.count_tensor.data_ptr()), outs, n_out,
                                                         C.c_void_p(stream), 0 if sync else GDV_EVAL_ASYNC)
            else:
                rc = lib.gdv_projector_evaluate(self._h, dbatch.num_rows, cols, len(dbatch.columns), sel_c,
                                                outs, n_out, GDV_MEM_DEVICE, C.c_void_p(stream),
                                                0 if sync else GDV_EVAL_ASYNC)
            grown = False
            if rc == 4 and attempt == 0:
                for i in range(n_out):
                    if varlen[i] and outs[i].data_size > caps[i]:
                        outputs[i].data = torch.empty(_pad64(outs[i].data_size), dtype=torch.uint8,
                                                      device="cuda")
                        outs[i].data, outs[i].data_size = outputs[i].data.data_ptr(), outputs[i].data.numel()
                        grown = True
                    elif varlen[i]:
                        outs[i].data_size = caps[i]
            if not grown:
                _check(rc)
                break
        for i in range(n_out):
            if varlen[i]:
                outputs[i].data_used = outs[i].data_size
            # (pending selection: the columns were sized for its capacity; they trim themselves to the
            # real count when it is first asked for)
            outputs[i].count_tensor = selection.count_tensor if pending else None
            outputs[i].length = out_rows
        return outputs


def _evaluate_device_async(self, dbatch, selection=None, outputs=None, capacity_bytes=None, stream=None):
    """Var-len plans without a host synchronisation (gdv_projector_evaluate_async): everything is enqueued and
    the call returns ``(outputs, result)`` — ``result`` is a torch int64 tensor of 1 + num_outputs elements that
    will hold, once the stream has passed: [0] the device status (0 = outputs complete; anything else: discard
    them and call evaluate_device), [1 + e] the bytes output e produced (above its capacity: buffer too small).
    ``selection`` may be a pending device SelectionVector (filter.evaluate_device(sync=False)): its count is
    read on the device.  ``capacity_bytes``: byte capacity of every var-len output (default: the capacity
    hint of earlier batches, else the bytes of the var-len inputs)."""
    import torch
    lib = _capi.lib()
    cols = (gdv_column_t * max(len(dbatch.columns), 1))(*[c._c() for c in dbatch.columns])
    pending = selection is not None and selection.device and selection.pending
    if selection is not None and not selection.device:
        raise TypeError("evaluate_device_async takes a device selection vector")
    out_rows = dbatch.num_rows if selection is None else (selection.indices.numel() if pending else selection.num_slots)
    n_out = len(self._out_types)
    varlen = [pa.types.is_string(t) or pa.types.is_binary(t) for t in self._out_types]
    if outputs is None:
        outputs = []
        guess = 64 + sum(c.data.numel() for c in dbatch.columns if c.offsets is not None)
        for i, t in enumerate(self._out_types):
            vb, db = C.c_int64(), C.c_int64()
            _check(lib.gdv_projector_output_sizes(self._h, i, out_rows, GDV_MEM_DEVICE, vb, db))
            dbytes = (capacity_bytes or db.value or guess) if varlen[i] else max(db.value, 1)
            outputs.append(DeviceColumn(t, out_rows, torch.empty(_pad64(max(vb.value, 1)), dtype=torch.uint8, device="cuda"),
                                        torch.empty(_pad64(dbytes), dtype=torch.uint8, device="cuda"),
                                        torch.empty(_pad64((out_rows + 1) * 4), dtype=torch.uint8, device="cuda") if varlen[i] else None))
    outs = (gdv_out_column_t * n_out)()
    for i, o in enumerate(outputs):
        outs[i].validity, outs[i].validity_size = o.validity.data_ptr(), o.validity.numel()
        outs[i].data, outs[i].data_size = o.data.data_ptr(), o.data.numel()
        if o.offsets is not None:
            outs[i].offsets, outs[i].offsets_size = o.offsets.data_ptr(), o.offsets.numel()
    sel_c, cnt_ptr = None, None
    if selection is not None:
        s = gdv_selection_t()
        s.mode, s.num_slots, s.indices = selection.mode, out_rows, selection.indices.data_ptr()
        sel_c = C.byref(s)
        if pending:
            cnt_ptr = C.c_void_p(selection.count_tensor.data_ptr())
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    result = torch.zeros(1 + n_out, dtype=torch.int64, device="cuda")
    _check(lib.gdv_projector_evaluate_async(self._h, dbatch.num_rows, cols, len(dbatch.columns), sel_c, cnt_ptr, outs, n_out,
                                            C.c_void_p(stream), C.c_void_p(result.data_ptr())))
    for i, o in enumerate(outputs):
        o.length = out_rows
        o.count_tensor = selection.count_tensor if pending else None
        o.result, o.result_index = result, i   # (the byte total is result[1 + i] once the stream has passed)
    return outputs, result


Projector.evaluate_device_async = _evaluate_device_async


def _evaluate_device_many(self, dbatches, outputs=None, stream=None, sync=True):
    """Many HBM-resident batches in one call (gdv_projector_evaluate_many): row-mode plans with
    fixed-width outputs run them all in ONE launch.  Returns a list (one per batch) of lists of
    DeviceColumns; pass it back as ``outputs`` to reuse the buffers."""
    import torch
    lib = _capi.lib()
    n_out = len(self._out_types)
    if any(pa.types.is_string(t) or pa.types.is_binary(t) for t in self._out_types):
        raise NotImplementedError("evaluate_device_many: fixed-width outputs only")
    if outputs is None:
        outputs = []
        for db in dbatches:
            cols = []
            for i, t in enumerate(self._out_types):
                vb, dbytes = C.c_int64(), C.c_int64()
                _check(lib.gdv_projector_output_sizes(self._h, i, db.num_rows, GDV_MEM_DEVICE, vb, dbytes))
                cols.append(DeviceColumn(t, db.num_rows,
                                         torch.empty(_pad64(max(vb.value, 1)), dtype=torch.uint8, device="cuda"),
                                         torch.empty(_pad64(max(dbytes.value, 1)), dtype=torch.uint8, device="cuda"), None))
            outputs.append(cols)
    nb = len(dbatches)
    batches = (_capi.gdv_batch_t * max(nb, 1))()
    keep = []
    for b, db in enumerate(dbatches):
        cols = (gdv_column_t * max(len(db.columns), 1))(*[c._c() for c in db.columns])
        outs = (gdv_out_column_t * n_out)()
        for i, o in enumerate(outputs[b]):
            outs[i].validity, outs[i].validity_size = o.validity.data_ptr(), o.validity.numel()
            outs[i].data, outs[i].data_size = o.data.data_ptr(), o.data.numel()
        keep.append((cols, outs))
        batches[b].num_rows = db.num_rows
        batches[b].cols, batches[b].num_cols = cols, len(db.columns)
        batches[b].outs, batches[b].num_outs = outs, n_out
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    _check(lib.gdv_projector_evaluate_many(self._h, batches, nb, C.c_void_p(stream), 0 if sync else GDV_EVAL_ASYNC))
    return outputs


Projector.evaluate_device_many = _evaluate_device_many


def _evaluate_device_array(projector, array_address, num_rows, on_device):
    """Shared body of Projector.evaluate_device_array: fixed-width outputs only."""
    lib = _capi.lib()
    n_out = len(projector._out_types)
    outs = (gdv_out_column_t * n_out)()
    mem = GDV_MEM_DEVICE if on_device else GDV_MEM_HOST
    holders = []
    for i, t in enumerate(projector._out_types):
        if pa.types.is_string(t) or pa.types.is_binary(t):
            raise pa.ArrowNotImplementedError("var-len outputs: use evaluate / evaluate_device")
        vb, db = C.c_int64(), C.c_int64()
        _check(lib.gdv_projector_output_sizes(projector._h, i, num_rows, mem, vb, db))
        if on_device:
            import torch
            v = torch.empty(_pad64(max(vb.value, 1)), dtype=torch.uint8, device="cuda")
            d = torch.empty(_pad64(max(db.value, 1)), dtype=torch.uint8, device="cuda")
            outs[i].validity, outs[i].validity_size = v.data_ptr(), v.numel()
            outs[i].data, outs[i].data_size = d.data_ptr(), d.numel()
            holders.append(DeviceColumn(t, num_rows, v, d))
        else:
            v = pa.allocate_buffer(_pad64(max(vb.value, 1)))
            d = pa.allocate_buffer(_pad64(max(db.value, 1)))
            outs[i].validity, outs[i].validity_size = v.address, v.size
            outs[i].data, outs[i].data_size = d.address, d.size
            holders.append((t, v, d))
    _check(lib.gdv_projector_evaluate_device_array(projector._h, C.c_void_p(array_address), None, outs,
                                                   n_out, None, 0))
    if on_device:
        return holders
    return [pa.Array.from_buffers(t, num_rows, [v, d]) for t, v, d in holders]


def make_projector(schema, children, pool=None, selection_mode="NONE", configuration=None):
    children = list(children)
    for c in children:
        if not isinstance(c, Expression):
            raise TypeError("make_projector expects gandiva Expressions")
    mode = _selection_mode(selection_mode)
    lib = _capi.lib()
    sh = _make_schema(schema)
    try:
        arr = (C.c_void_p * max(len(children), 1))(*[c._h for c in children])
        out = C.c_void_p()
        cfg = (configuration or Configuration())._c()
        _check(lib.gdv_projector_make(sh, arr, len(children), mode, C.byref(cfg), C.byref(out)))
    finally:
        lib.gdv_schema_free(sh)
    return Projector(out, schema, mode, children)


# ---------------------------------------------------------------------------- Filter

class Filter:
    def __init__(self, handle, schema, condition):
        self._h = handle
        self._schema = schema
        self._condition = condition

    def __del__(self):
        try:
            _capi.lib().gdv_filter_free(self._h)
        except Exception:
            pass

    @property
    def llvm_ir(self):
        return _capi.take_string(_capi.lib().gdv_filter_dump_ir(self._h))

    @staticmethod
    def _mode_of(dtype):
        t = ensure_type(dtype)
        if pa.types.is_int16(t) or pa.types.is_uint16(t): return 1
        if pa.types.is_int32(t) or pa.types.is_uint32(t): return 2
        if pa.types.is_int64(t) or pa.types.is_uint64(t): return 3
        raise ValueError("'dtype' of the selection vector should be one of 'int16', 'int32' and 'int64'.")

    def evaluate(self, batch, pool=None, dtype="int32"):
        _check_batch(batch, self._schema)
        mode = self._mode_of(dtype)
        lib = _capi.lib()
        cols = (gdv_column_t * max(batch.num_columns, 1))(*[_column_of_array(a) for a in batch.columns])
        idx = np.empty(max(batch.num_rows, 1), dtype=_SEL_DTYPE[mode][1])
        count = C.c_int64(0)
        _check(lib.gdv_filter_evaluate(self._h, batch.num_rows, cols, batch.num_columns, mode,
                                       idx.ctypes.data, batch.num_rows, C.byref(count),
                                       GDV_MEM_HOST, None))
        return SelectionVector(mode, idx, count.value, device=False)

    def set_tuning(self, key, value):
        """gdv_filter_set_tuning: "chunks" (1..64) / "small_filter" (0 / 1) — tests and measurements."""
        _check(_capi.lib().gdv_filter_set_tuning(self._h, key.encode(), int(value)))

    def evaluate_device(self, dbatch, dtype="int32", out=None, stream=None, sync=True):
        """HBM-resident filter.  sync=False: everything is enqueued and the call returns at once; the
        SelectionVector's slot count stays on the device (``count_tensor``) until someone asks for
        ``num_slots`` — a selection-mode Projector.evaluate_device takes it from there directly."""
        import torch
        mode = self._mode_of(dtype)
        lib = _capi.lib()
        cols = (gdv_column_t * max(len(dbatch.columns), 1))(*[c._c() for c in dbatch.columns])
        tdt = {1: torch.int16, 2: torch.int32, 3: torch.int64}[mode]
        if out is None:
            out = torch.empty(max(dbatch.num_rows, 1), dtype=tdt, device="cuda")
        count = C.c_int64(0)
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        if not sync:
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            _check(lib.gdv_filter_evaluate_async(self._h, dbatch.num_rows, cols, len(dbatch.columns), mode,
                                                 C.c_void_p(out.data_ptr()), out.numel(),
                                                 C.c_void_p(cnt.data_ptr()), C.c_void_p(stream)))
            return SelectionVector(mode, out, None, device=True, count_tensor=cnt)
        _check(lib.gdv_filter_evaluate(self._h, dbatch.num_rows, cols, len(dbatch.columns), mode,
                                       C.c_void_p(out.data_ptr()), out.numel(), C.byref(count),
                                       GDV_MEM_DEVICE, C.c_void_p(stream)))
        return SelectionVector(mode, out, count.value, device=True)


def _filter_evaluate_device_many(self, dbatches, dtype="int32", stream=None):
    """Many small HBM-resident batches in one launch (gdv_filter_evaluate_many): returns one device
    SelectionVector per batch."""
    import torch
    mode = self._mode_of(dtype)
    lib = _capi.lib()
    tdt = {1: torch.int16, 2: torch.int32, 3: torch.int64}[mode]
    nb = len(dbatches)
    batches = (_capi.gdv_filter_batch_t * max(nb, 1))()
    keep, outs = [], []
    for b, db in enumerate(dbatches):
        cols = (gdv_column_t * max(len(db.columns), 1))(*[c._c() for c in db.columns])
        out = torch.empty(max(db.num_rows, 1), dtype=tdt, device="cuda")
        keep.append(cols)
        outs.append(out)
        batches[b].num_rows, batches[b].cols, batches[b].num_cols = db.num_rows, cols, len(db.columns)
        batches[b].out_indices, batches[b].max_slots = out.data_ptr(), out.numel()
    counts = (C.c_int64 * max(nb, 1))()
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    _check(lib.gdv_filter_evaluate_many(self._h, batches, nb, mode, counts, None, C.c_void_p(stream), 0))
    return [SelectionVector(mode, outs[b], counts[b], device=True) for b in range(nb)]


Filter.evaluate_device_many = _filter_evaluate_device_many


def make_filter(schema, condition, configuration=None):
    if not isinstance(condition, Condition):
        raise TypeError("make_filter expects a gandiva Condition")
    lib = _capi.lib()
    sh = _make_schema(schema)
    try:
        out = C.c_void_p()
        cfg = (configuration or Configuration())._c()
        _check(lib.gdv_filter_make(sh, condition._h, C.byref(cfg), C.byref(out)))
    finally:
        lib.gdv_schema_free(sh)
    return Filter(out, schema, condition)


# ---------------------------------------------------------------------------- Filter -> Projector, fused

class FilterProject:
    """Filter.evaluate -> SelectionVector -> Projector.evaluate(batch, selection) as ONE operator
    (gdv_filter_project_*): a single kernel reads the batch once and writes the projections of the
    selected rows, compacted, plus the selection vector itself when ``index_dtype`` is given.  Plans
    the fused kernel does not take (var-len columns / outputs) are evaluated as the chain of a Filter
    and a selection-mode Projector — same results, same interface (``fused`` tells which)."""

    def __init__(self, handle, schema, condition, exprs, mode, chain=None):
        self._h, self._schema, self._condition, self._exprs, self._mode = handle, schema, condition, exprs, mode
        self._chain = chain  # (Filter, Projector) when the plan is not fused
        if handle is not None:
            lib = _capi.lib()
            self._out_types = [from_gdv_type(lib.gdv_filter_project_output_type(handle, i))
                               for i in range(lib.gdv_filter_project_num_outputs(handle))]
        else:
            self._out_types = chain[1]._out_types

    def __del__(self):
        try:
            if self._h is not None:
                _capi.lib().gdv_filter_project_free(self._h)
        except Exception:
            pass

    @property
    def fused(self):
        return self._h is not None

    @property
    def llvm_ir(self):
        if self._h is None:
            return self._chain[0].llvm_ir + self._chain[1].llvm_ir
        return _capi.take_string(_capi.lib().gdv_filter_project_dump_ir(self._h))

    def set_tuning(self, key, value):
        """gdv_filter_project_set_tuning: "kernel" (-1 follow the selectivity / 0 windowed / 1 direct)."""
        if self._h is not None:
            _check(_capi.lib().gdv_filter_project_set_tuning(self._h, key.encode(), int(value)))

    @property
    def kernel_shape(self):
        """0: the windowed kernel runs next (selected rows staged in LDS); 1: the direct one (recent batches selected
        more rows than the window holds); -1: the plan has one shape only, or is a chain."""
        return -1 if self._h is None else _capi.lib().gdv_filter_project_kernel_shape(self._h)

    def evaluate(self, batch):
        """Host buffers in, host pyarrow arrays out: ``(arrays, selection_vector or None)``."""
        _check_batch(batch, self._schema)
        if self._h is None:
            flt, proj = self._chain
            sel = flt.evaluate(batch, None, {1: "int16", 2: "int32", 3: "int64"}[self._mode or 2])
            return proj.evaluate(batch, sel), (sel if self._mode else None)
        lib = _capi.lib()
        n = batch.num_rows
        cols = (gdv_column_t * max(batch.num_columns, 1))(*[_column_of_array(a) for a in batch.columns])
        n_out = len(self._out_types)
        outs = (gdv_out_column_t * n_out)()
        holders = []
        for i, t in enumerate(self._out_types):
            v = pa.allocate_buffer(_pad64(max((n + 7) // 8, 1)))
            d = pa.allocate_buffer(_pad64(max((n + 7) // 8 if pa.types.is_boolean(t) else n * t.bit_width // 8, 1)))
            holders.append((v, d))
            outs[i].validity, outs[i].validity_size = v.address, v.size
            outs[i].data, outs[i].data_size = d.address, d.size
        idx = None
        if self._mode:
            idx = np.empty(max(n, 1), dtype=_SEL_DTYPE[self._mode][1])
        count = C.c_int64(0)
        _check(lib.gdv_filter_project_evaluate(self._h, n, cols, batch.num_columns, outs, n_out,
                                               C.c_void_p(idx.ctypes.data if idx is not None else 0), n, C.byref(count),
                                               None, GDV_MEM_HOST, None, 0))
        k = count.value
        arrays = [pa.Array.from_buffers(t, k, [v, d]) for t, (v, d) in zip(self._out_types, holders)]
        return arrays, (SelectionVector(self._mode, idx, k) if self._mode else None)

    def evaluate_device(self, dbatch, outputs=None, indices=None, stream=None, sync=True):
        """HBM-resident: returns ``(DeviceColumns, SelectionVector or None)``.  The columns are sized for
        ``dbatch.num_rows`` rows (the count is known only afterwards) and carry the count: ``num_rows`` /
        ``to_arrow`` trim to it.  sync=False (plans that cannot raise): nothing waits, the count stays on the
        device until someone asks."""
        import torch
        if self._h is None:
            flt, proj = self._chain
            sel = flt.evaluate_device(dbatch, {1: "int16", 2: "int32", 3: "int64"}[self._mode or 2], out=indices,
                                      stream=stream, sync=sync)
            return proj.evaluate_device(dbatch, selection=sel, outputs=outputs, stream=stream, sync=sync), (sel if self._mode else None)
        lib = _capi.lib()
        n = dbatch.num_rows
        cols = (gdv_column_t * max(len(dbatch.columns), 1))(*[c._c() for c in dbatch.columns])
        n_out = len(self._out_types)
        vbytes = (n + 63) // 64 * 8
        if outputs is None:
            outputs = [DeviceColumn(t, n, torch.empty(_pad64(max(vbytes, 1)), dtype=torch.uint8, device="cuda"),
                                    torch.empty(_pad64(max(vbytes if pa.types.is_boolean(t) else n * t.bit_width // 8, 1)),
                                                dtype=torch.uint8, device="cuda")) for t in self._out_types]
        outs = (gdv_out_column_t * n_out)()
        for i, o in enumerate(outputs):
            outs[i].validity, outs[i].validity_size = o.validity.data_ptr(), o.validity.numel()
            outs[i].data, outs[i].data_size = o.data.data_ptr(), o.data.numel()
        if self._mode and indices is None:
            indices = torch.empty(max(n, 1), dtype={1: torch.int16, 2: torch.int32, 3: torch.int64}[self._mode], device="cuda")
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        count = C.c_int64(0)
        _check(lib.gdv_filter_project_evaluate(self._h, n, cols, len(dbatch.columns), outs, n_out,
                                               C.c_void_p(indices.data_ptr() if self._mode else 0),
                                               indices.numel() if self._mode else 0, C.byref(count),
                                               C.c_void_p(cnt.data_ptr()), GDV_MEM_DEVICE, C.c_void_p(stream),
                                               0 if sync else GDV_EVAL_ASYNC))
        pending = count.value < 0
        for o in outputs:
            o.length = n if pending else count.value
            o.count_tensor = cnt if pending else None
        sel = None
        if self._mode:
            sel = SelectionVector(self._mode, indices, None if pending else count.value, device=True, count_tensor=cnt)
        self.last_count_tensor = cnt
        return outputs, sel


def make_filter_project(schema, condition, children, index_dtype=None, configuration=None):
    """One operator for filter -> project.  ``index_dtype`` ("int16" / "int32" / "int64"): also emit the
    selection vector.  Fused where the plan allows it, the Filter + Projector chain otherwise."""
    if not isinstance(condition, Condition):
        raise TypeError("make_filter_project expects a gandiva Condition")
    mode = 0 if index_dtype is None else Filter._mode_of(index_dtype)
    lib = _capi.lib()
    sh = _make_schema(schema)
    try:
        out = C.c_void_p()
        cfg = (configuration or Configuration())._c()
        arr = (C.c_void_p * len(children))(*[e._h for e in children])
        rc = lib.gdv_filter_project_make(sh, condition._h, arr, len(children), mode, C.byref(cfg), C.byref(out))
    finally:
        lib.gdv_schema_free(sh)
    if rc == 40:  # CodeGenError: not a fused shape -> the chain
        flt = make_filter(schema, condition, configuration)
        proj = make_projector(schema, children, None, {0: "UINT32", 1: "UINT16", 2: "UINT32", 3: "UINT64"}[mode], configuration)
        return FilterProject(None, schema, condition, children, mode, chain=(flt, proj))
    _check(rc)
    return FilterProject(out, schema, condition, children, mode)


# ---------------------------------------------------------------------------- registry

class FunctionSignature:
    def __init__(self, name, return_type, param_types):
        self._name, self._ret, self._params = name, return_type, param_types

    def return_type(self):
        return self._ret

    def param_types(self):
        return list(self._params)

    def name(self):
        return self._name

    def __repr__(self):
        return f"FunctionSignature({self._ret} {self._name}({', '.join(map(str, self._params))}))"


# ---- devices (round 3): every call runs on the calling thread's device context
def device_count():
    """Devices the library can be pointed at (physical HIP devices, or more with virtual devices)."""
    return _capi.lib().gdv_device_count()


def physical_device_count():
    return _capi.lib().gdv_physical_device_count()


def set_virtual_devices(n):
    """Device ids up to n: id d runs on physical device d % physical_device_count() with a context
    (code objects, buffer pool, streams) of its own — the N-device code path on fewer GPUs."""
    _check(_capi.lib().gdv_set_virtual_devices(int(n)))


def set_device(device):
    """Select the calling THREAD's device (also makes it the thread's current HIP device, which
    torch follows).  One host thread per device drives all GPUs of a node from one process."""
    _check(_capi.lib().gdv_set_device(int(device)))


def get_device():
    return _capi.lib().gdv_get_device()


def get_registered_function_signatures():
    lib = _capi.lib()
    out = []
    for i in range(lib.gdv_registry_size()):
        name = C.c_char_p()
        ret = gdv_type_t()
        params = (gdv_type_t * 8)()
        n = C.c_int()
        _check(lib.gdv_registry_get(i, C.byref(name), C.byref(ret), params, 8, C.byref(n)))
        out.append(FunctionSignature(name.value.decode(), from_gdv_type(ret),
                                     [from_gdv_type(params[j]) for j in range(n.value)]))
    return out
