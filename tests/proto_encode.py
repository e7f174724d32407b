"""A hand-written protobuf ENCODER for the subset of the reference's Types.proto that the JNI
boundary ships (Schema, ExpressionList, Condition) — test infrastructure: it turns the plain-Python
description every gandiva_amd Node carries into the bytes the Java side would serialise, so that
gdv_projector_make_from_proto / gdv_filter_make_from_proto can be driven without protoc.
Message layout: as restated in gandiva_amd/csrc/gdv_proto.cc."""
import struct

import pyarrow as pa

_GTYPE = {pa.bool_(): 1, pa.uint8(): 2, pa.int8(): 3, pa.uint16(): 4, pa.int16(): 5, pa.uint32(): 6,
          pa.int32(): 7, pa.uint64(): 8, pa.int64(): 9, pa.float32(): 11, pa.float64(): 12, pa.string(): 13,
          pa.binary(): 14, pa.date32(): 16, pa.date64(): 17}
_UNIT = {"s": 0, "ms": 1, "us": 2, "ns": 3}


def varint(v):
    v &= (1 << 64) - 1          # negative int32 / int64 values travel as 10-byte varints
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def key(field, wire):
    return varint((field << 3) | wire)


def ld(field, payload):                      # length-delimited
    return key(field, 2) + varint(len(payload)) + payload


def vi(field, value):
    return key(field, 0) + varint(value)


def ext_type(t):
    if pa.types.is_decimal(t):
        return vi(1, 22) + vi(3, t.precision) + vi(4, t.scale)
    if pa.types.is_timestamp(t):
        return vi(1, 18) + vi(6, _UNIT[t.unit])
    if pa.types.is_time32(t):
        return vi(1, 19) + vi(6, _UNIT[t.unit])
    if pa.types.is_time64(t):
        return vi(1, 20) + vi(6, _UNIT[t.unit])
    return vi(1, _GTYPE[t])


def field(f):
    return ld(1, f.name.encode()) + ld(2, ext_type(f.type)) + vi(3, 1 if f.nullable else 0)


def schema(s):
    return b"".join(ld(1, field(f)) for f in s)


def _literal(node):
    t, v = node.dtype, node.desc["value"]
    if node.desc.get("is_null"):
        return ld(11, ld(1, ext_type(t)))
    if t == pa.int32():
        return ld(12, vi(1, v))
    if t == pa.float32():
        return ld(13, key(1, 5) + struct.pack("<f", v))
    if t == pa.int64():
        return ld(14, vi(1, v))
    if t == pa.bool_():
        return ld(15, vi(1, 1 if v else 0))
    if t == pa.float64():
        return ld(16, key(1, 1) + struct.pack("<d", v))
    if t == pa.string():
        return ld(17, ld(1, v))
    if t == pa.binary():
        return ld(18, ld(1, v))
    if pa.types.is_decimal(t):
        return ld(19, ld(1, str(int(v)).encode()) + vi(2, t.precision) + vi(3, t.scale))
    raise NotImplementedError(f"literal of type {t} has no node in Types.proto")


def tree(node):
    k = node.kind
    if k == "field":
        return ld(1, ld(1, field(pa.field(node.desc["name"], node.dtype))))
    if k == "literal":
        return _literal(node)
    if k == "function":
        body = ld(1, node.desc["name"].encode())
        body += b"".join(ld(2, tree(c)) for c in node.desc["children"])
        return ld(2, body + ld(3, ext_type(node.dtype)))
    if k == "if":
        c, t, e = node.desc["children"]
        return ld(6, ld(1, tree(c)) + ld(2, tree(t)) + ld(3, tree(e)) + ld(4, ext_type(node.dtype)))
    if k in ("and", "or"):
        return ld(7 if k == "and" else 8, b"".join(ld(1, tree(c)) for c in node.desc["children"]))
    if k == "in":
        vt = node.desc["value_type"]
        vals = node.desc["values"]
        body = ld(1, tree(node.desc["children"][0]))
        if vt == pa.int32():
            body += ld(2, b"".join(ld(1, vi(1, v)) for v in vals))
        elif vt == pa.int64():
            body += ld(3, b"".join(ld(1, vi(1, v)) for v in vals))
        elif vt == pa.string():
            body += ld(4, b"".join(ld(1, ld(1, v)) for v in vals))
        elif vt == pa.binary():
            body += ld(5, b"".join(ld(1, ld(1, v)) for v in vals))
        else:
            raise NotImplementedError(f"IN over {vt} has no constants message in Types.proto")
        return ld(21, body)
    raise NotImplementedError(k)


def expression_list(exprs):
    return b"".join(ld(2, ld(1, tree(e.root())) + ld(2, field(e.result()))) for e in exprs)


def condition(cond):
    return ld(1, tree(cond.root()))
