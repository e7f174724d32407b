"""The product's device function library compiled for the HOST (tests/host_devlib/): its
per-row functions are plain C++, so with the GPU intrinsics stubbed they run on a CPU-only
machine.  This file drives them over dense random inputs and compares with exact arithmetic /
the oracle — a second line of defence for the scalar semantics of the device code that needs
no GPU (wave-level helpers are stubbed here and covered by the GPU parity suite)."""
import ctypes as C
import decimal
import os
import subprocess

import numpy as np
import pyarrow as pa
import pytest

import test_decimal as D

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_devlib", "host_devlib.cc")
LIB = os.path.join(HERE, "host_devlib", "libhost_devlib.so")


@pytest.fixture(scope="module")
def hostlib():
    hdr = os.path.join(HERE, "..", "gandiva_amd", "csrc", "gdv_device_lib.hpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unused-function", "-Wno-unused-variable", SRC, "-o", LIB])
    return C.CDLL(LIB)


def _raw128(arr):
    """16-byte little-endian values of a decimal128 array (nulls as 1: the kernels never call a
    raising function on a null row, the plain loop here does)."""
    out = np.zeros(len(arr) * 2, dtype=np.uint64)
    for i, v in enumerate(arr.to_pylist()):
        if v is None:
            out[2 * i] = 1
            continue
        iv = int(v.scaleb(arr.type.scale, D.CTX)) & ((1 << 128) - 1)
        out[2 * i], out[2 * i + 1] = iv & ((1 << 64) - 1), iv >> 64
    return out


def _from_raw128(raw, t, valid):
    vals = []
    for i, ok in enumerate(valid):
        if not ok:
            vals.append(None)
            continue
        iv = int(raw[2 * i]) | (int(raw[2 * i + 1]) << 64)
        if iv >> 127:
            iv -= 1 << 128
        vals.append(decimal.Decimal(iv).scaleb(-t.scale, D.CTX))
    return vals


@pytest.mark.parametrize("seed", range(16))
def test_device_decimal_functions_are_exact_on_dense_digits(hostlib, seed):
    rng = np.random.default_rng(1700 + seed)
    ta = pa.decimal128(*D.DENSE_TYPES[int(rng.integers(0, len(D.DENSE_TYPES)))])
    tb = pa.decimal128(*D.DENSE_TYPES[int(rng.integers(0, len(D.DENSE_TYPES)))])
    n = 300
    a, b = D._dense_decimals(rng, ta, n), D._nonzero(D._dense_decimals(rng, tb, n))
    xa, xb = _raw128(a), _raw128(b)
    valid = [x is not None and y is not None for x, y in zip(a.to_pylist(), b.to_pylist())]
    for code, op in enumerate(("add", "subtract", "multiply", "divide", "mod")):
        rt = D._result_type(op, ta, tb)
        out = np.zeros(2 * n, dtype=np.uint64)
        err = hostlib.host_decimal_binary(code, xa.ctypes.data_as(C.c_void_p), ta.precision, ta.scale,
                                          xb.ctypes.data_as(C.c_void_p), tb.precision, tb.scale,
                                          rt.precision, rt.scale, out.ctypes.data_as(C.c_void_p), C.c_long(n))
        assert err == 0
        ref = D._python_divmod_expected if op in ("divide", "mod") else D._python_expected
        want = ref(op, a.to_pylist(), b.to_pylist(), rt)
        assert _from_raw128(out, rt, valid) == want, f"{op} {ta} {tb} -> {rt}"


def _half_away(v, scale):
    q = decimal.Decimal(1).scaleb(-scale)
    return v.quantize(q, rounding=decimal.ROUND_HALF_UP, context=D.CTX)


@pytest.mark.parametrize("seed", range(12))
def test_device_decimal_casts_and_compares_on_dense_digits(hostlib, seed):
    rng = np.random.default_rng(2600 + seed)
    ta = pa.decimal128(*D.DENSE_TYPES[int(rng.integers(0, len(D.DENSE_TYPES)))])
    tb = pa.decimal128(*D.DENSE_TYPES[int(rng.integers(0, len(D.DENSE_TYPES)))])
    n = 300
    a, b = D._dense_decimals(rng, ta, n, 0.0), D._dense_decimals(rng, tb, n, 0.0)
    xa, xb = _raw128(a), _raw128(b)
    av, bv = a.to_pylist(), b.to_pylist()
    # decimal -> decimal(tb): rescale, round half away from zero, 0 when it needs more digits
    out = np.zeros(2 * n, dtype=np.uint64)
    hostlib.host_decimal_cast(xa.ctypes.data_as(C.c_void_p), ta.precision, ta.scale, tb.precision, tb.scale,
                              out.ctypes.data_as(C.c_void_p), C.c_long(n))
    lim = decimal.Decimal(10) ** (tb.precision - tb.scale)
    want = [decimal.Decimal(0).scaleb(-tb.scale) if abs(_half_away(v, tb.scale)) >= lim else _half_away(v, tb.scale)
            for v in av]
    assert _from_raw128(out, tb, [True] * n) == want, f"cast {ta} -> {tb}"
    # int64 -> decimal(ta)
    ints = rng.integers(-2**62, 2**62, n) // (10 ** rng.integers(0, 18, n))
    out = np.zeros(2 * n, dtype=np.uint64)
    hostlib.host_decimal_from_int64(ints.ctypes.data_as(C.c_void_p), ta.precision, ta.scale,
                                    out.ctypes.data_as(C.c_void_p), C.c_long(n))
    lim = decimal.Decimal(10) ** (ta.precision - ta.scale)
    want = [decimal.Decimal(0).scaleb(-ta.scale) if abs(decimal.Decimal(int(v))) >= lim
            else decimal.Decimal(int(v)).scaleb(0).quantize(decimal.Decimal(1).scaleb(-ta.scale), context=D.CTX) for v in ints]
    assert _from_raw128(out, ta, [True] * n) == want, f"int64 -> {ta}"
    # decimal -> int64 (round half away; values that fit)
    small = [v for v in av if abs(v) < decimal.Decimal(2) ** 62]
    if small:
        arr = pa.array(small, type=ta)
        res = np.zeros(len(small), dtype=np.int64)
        hostlib.host_decimal_to_int64(_raw128(arr).ctypes.data_as(C.c_void_p), ta.precision, ta.scale,
                                      res.ctypes.data_as(C.c_void_p), C.c_long(len(small)))
        assert res.tolist() == [int(_half_away(v, 0)) for v in small], f"{ta} -> int64"
    # three-way compare across scales
    cmp = np.zeros(n, dtype=np.int8)
    hostlib.host_decimal_compare(xa.ctypes.data_as(C.c_void_p), ta.precision, ta.scale,
                                 xb.ctypes.data_as(C.c_void_p), tb.precision, tb.scale,
                                 cmp.ctypes.data_as(C.c_void_p), C.c_long(n))
    assert cmp.tolist() == [(x > y) - (x < y) for x, y in zip(av, bv)], f"compare {ta} {tb}"


# ------------------------------------------------------------------ utf8 functions on the host build

import gandiva_amd as gandiva
from oracle import oracle
import test_strings as S


def _col(arr):
    """(offsets int32, data uint8 padded by 8 zero bytes, true size) of a utf8 array, nulls emptied."""
    vals = ["" if v is None else v for v in arr.to_pylist()]
    raw = [v.encode() for v in vals]
    off = np.zeros(len(raw) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(r) for r in raw])
    data = np.frombuffer(b"".join(raw) + b"\0" * 16, dtype=np.uint8).copy()
    return off, data, int(off[-1])


def _lit(text):
    raw = text.encode()
    return np.frombuffer(raw + b"\0" * 8, dtype=np.uint8).copy(), len(raw)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("seed", range(6))
def test_device_string_functions_match_oracle_on_host(hostlib, seed):
    rng = np.random.default_rng(4100 + seed)
    n = 1500
    s_arr = S._strings(rng, n, null_fraction=0.0)
    t_arr = pa.array([["a", "spark", "rk", "é", "", "Sp", "bright spark and fire"][int(rng.integers(0, 7))]
                      for _ in range(n)], pa.string())
    batch = pa.RecordBatch.from_arrays([s_arr, t_arr], names=["s", "t"])
    b = gandiva.TreeExprBuilder()
    s, t = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    off, data, size = _col(s_arr)
    offt, datat, sizet = _col(t_arr)
    i64, i32, STR, BOOL = pa.int64(), pa.int32(), pa.string(), pa.bool_()
    lit = lambda v: b.make_literal(v, STR)

    def oracle_of(node, typ):
        return oracle.project_one(node, typ, batch).to_pylist()

    # predicates against literals, also through the upper() byte map
    for fn, name, needle in [(0, "%{}%", "spark"), (0, "%{}%", "a"), (0, "%{}%", "日本"), (1, "{}%", "spa"),
                             (2, "%{}", "rk"), (3, "{}", "spark"), (0, "%{}%", "xxxxxxxxxxxx")]:
        lb, ll = _lit(needle)
        for mp, wrap in ((0, lambda x: x), (1, lambda x: b.make_function("upper", [x], STR))):
            pat = name.format(needle.upper() if mp else needle)
            lb2, ll2 = _lit(needle.upper() if mp else needle)
            out = np.zeros(n, dtype=np.uint8)
            hostlib.host_str_pred_lit(fn, _p(off), _p(data), C.c_long(size), C.c_long(n), _p(lb2), ll2, mp, _p(out))
            want = oracle_of(b.make_function("like", [wrap(s), lit(pat)], BOOL), BOOL)
            assert out.astype(bool).tolist() == want, (fn, pat, mp)
    for fn, op, v in [(4, "equal", "spark"), (5, "less_than", "park"), (6, "starts_with", "spa"),
                      (7, "ends_with", "rk"), (8, "greater_than_or_equal_to", "a_b%c"), (4, "equal", ""),
                      (5, "less_than", "ünïcödé spark"), (4, "equal", "x" * 300)]:
        lb, ll = _lit(v)
        out = np.zeros(n, dtype=np.uint8)
        hostlib.host_str_pred_lit(fn, _p(off), _p(data), C.c_long(size), C.c_long(n), _p(lb), ll, 0, _p(out))
        assert out.astype(bool).tolist() == oracle_of(b.make_function(op, [s, lit(v)], BOOL), BOOL), (op, v)
    # column against column
    for fn, op in [(0, "equal"), (1, "less_than"), (2, "starts_with"), (3, "ends_with")]:
        out = np.zeros(n, dtype=np.uint8)
        hostlib.host_str_pred_col(fn, _p(off), _p(data), C.c_long(size), _p(offt), _p(datat), C.c_long(sizet),
                                  C.c_long(n), _p(out))
        assert out.astype(bool).tolist() == oracle_of(b.make_function(op, [s, t], BOOL), BOOL), op
    # integer-valued functions
    lb, ll = _lit("ar")
    for fn, node, typ in [(0, b.make_function("octet_length", [s], i32), i32),
                          (1, b.make_function("char_length", [s], i32), i32),
                          (2, b.make_function("hash32", [s], i32), i32),
                          (3, b.make_function("hash64", [s], i64), i64),
                          (4, b.make_function("ascii", [s], i32), i32),
                          (5, b.make_function("locate", [lit("ar"), s], i32), i32)]:
        out = np.zeros(n, dtype=np.int64)
        hostlib.host_str_int(fn, _p(off), _p(data), C.c_long(size), C.c_long(n), _p(lb), ll, 0, _p(out))
        assert out.tolist() == oracle_of(node, typ), fn
    out = np.zeros(n, dtype=np.int64)   # hash of the upper-cased view
    hostlib.host_str_int(3, _p(off), _p(data), C.c_long(size), C.c_long(n), _p(lb), ll, 1, _p(out))
    assert out.tolist() == oracle_of(b.make_function("hash64", [b.make_function("upper", [s], STR)], i64), i64)
    # views, materialised by gdv_str_copy
    K = lambda v, tt=i64: b.make_literal(v, tt)
    cases = [(0, 0, 0, 0, s), (0, 0, 0, 1, b.make_function("upper", [s], STR)),
             (0, 0, 0, 2, b.make_function("lower", [s], STR)),
             (2, 0, 0, 0, b.make_function("ltrim", [s], STR)), (3, 0, 0, 0, b.make_function("rtrim", [s], STR)),
             (4, 0, 0, 1, b.make_function("upper", [b.make_function("btrim", [s], STR)], STR)),
             (7, 7, 0, 0, b.make_function("castVARCHAR", [s, K(7)], STR)), (8, 3, 0, 0, b.make_function("substr", [s, K(3)], STR))]
    for frm, cnt in [(2, 5), (1, 1), (-3, 2), (0, 4), (5, 100), (50, 2), (-400, 3), (9, 30)]:
        cases.append((1, frm, cnt, 0, b.make_function("substr", [s, K(frm), K(cnt)], STR)))
        cases.append((1, frm, cnt, 1, b.make_function("upper", [b.make_function("substr", [s, K(frm), K(cnt)], STR)], STR)))
    for k in (0, 1, 3, 100, -1, -4, -100):
        cases.append((5, k, 0, 0, b.make_function("left", [s, K(k, i32)], STR)))
        cases.append((6, k, 0, 0, b.make_function("right", [s, K(k, i32)], STR)))
    for fn, a1, a2, mp, node in cases:
        out_off = np.zeros(n + 1, dtype=np.int32)
        out_data = np.zeros(size + 64, dtype=np.uint8)
        total = hostlib.host_str_view(fn, _p(off), _p(data), C.c_long(size), C.c_long(n), C.c_longlong(a1),
                                      C.c_longlong(a2), mp, _p(out_off), _p(out_data))
        got = [bytes(out_data[out_off[i]:out_off[i + 1]]).decode() for i in range(n)]
        assert total == out_off[-1]
        assert got == oracle_of(node, STR), (fn, a1, a2, mp)
    # IN over a literal list
    lits = ["spark", "park", "", "日本語テキスト", "bright spark and fire", "x" * 300]
    raw = b"".join(v.encode() for v in lits)
    loffs = np.zeros(len(lits) + 1, dtype=np.int32)
    loffs[1:] = np.cumsum([len(v.encode()) for v in lits])
    lbytes = np.frombuffer(raw + b"\0" * 8, dtype=np.uint8).copy()
    out = np.zeros(n, dtype=np.uint8)
    hostlib.host_str_in(_p(off), _p(data), C.c_long(size), C.c_long(n), _p(lbytes), _p(loffs), len(lits), _p(out))
    assert out.astype(bool).tolist() == oracle_of(b.make_in_expression(s, lits, STR), BOOL)


def test_device_text_to_integer_casts_on_host(hostlib):
    texts = S.NUMBER_TEXTS
    arr = pa.array(texts, pa.string())
    off, data, size = _col(arr)
    lb, ll = _lit("")
    for fn, bits, marker in ((7, 32, 1 << 40), (8, 64, -1234567890123456789)):
        out = np.zeros(len(texts), dtype=np.int64)
        hostlib.host_str_int(fn, _p(off), _p(data), C.c_long(size), C.c_long(len(texts)), _p(lb), ll, 0, _p(out))
        want = [marker if S._python_parse(t, bits) == "error" else S._python_parse(t, bits) for t in texts]
        assert out.tolist() == want, bits


def _compile_like(pattern, escape=None):
    """(bytes, kinds) as CodeGen::CompileLike builds them: kind 0 literal byte, 1 '_', 2 '%' (runs collapsed)."""
    raw = pattern.encode()
    esc = escape.encode()[0] if escape else None
    pb, pk, i = bytearray(), bytearray(), 0
    while i < len(raw):
        c = raw[i]
        if esc is not None and c == esc:
            pb.append(raw[i + 1]); pk.append(0); i += 2
            continue
        if c == ord("%"):
            if not pk or pk[-1] != 2:
                pb.append(0); pk.append(2)
        elif c == ord("_"):
            pb.append(0); pk.append(1)
        else:
            pb.append(c); pk.append(0)
        i += 1
    return (np.frombuffer(bytes(pb) + b"\0" * 8, dtype=np.uint8).copy(),
            np.frombuffer(bytes(pk) + b"\0" * 8, dtype=np.uint8).copy(), len(pk))


@pytest.mark.parametrize("seed", range(4))
def test_device_general_like_matcher_on_host(hostlib, seed):
    rng = np.random.default_rng(5200 + seed)
    n = 2000
    s_arr = S._strings(rng, n, null_fraction=0.0)
    batch = pa.RecordBatch.from_arrays([s_arr], names=["s"])
    off, data, size = _col(s_arr)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    for pattern, escape in [("s_ark%", None), ("%a%b%", None), ("_%_", None), ("%_ark", None), ("a_b%c", None),
                            ("%", None), ("", None), ("__", None), ("%é%", None), ("_本%", None), ("%spa_k%fire", None),
                            ("100#%", "#"), ("a\\_b\\%c", "\\"), ("%#_%", "#"), ("%%%a%%", None)]:
        pb, pk, plen = _compile_like(pattern, escape)
        out = np.zeros(n, dtype=np.uint8)
        hostlib.host_str_like(_p(off), _p(data), C.c_long(size), C.c_long(n), _p(pb), _p(pk), plen, 0, _p(out))
        args = [s, b.make_literal(pattern, pa.string())] + ([b.make_literal(escape, pa.string())] if escape else [])
        want = oracle.project_one(b.make_function("like", args, pa.bool_()), pa.bool_(), batch).to_pylist()
        assert out.astype(bool).tolist() == want, (pattern, escape)


def test_device_decimal_functions_reproduce_the_c4_golden_fixture(hostlib):
    """The TPC-H Q1 projections of BASELINE config C4, operator by operator, through the
    host-built device functions: must reproduce tests/golden/c4.arrow."""
    import test_golden as G
    ins, outs = G.load("c4")
    ep, disc, tax, _ = ins
    n = len(ep)

    def call(code, xa, ta, xb, tb, rt):
        out = np.zeros(2 * n, dtype=np.uint64)
        err = hostlib.host_decimal_binary(code, _p(xa), ta[0], ta[1], _p(xb), tb[0], tb[1], rt[0], rt[1], _p(out),
                                          C.c_long(n))
        assert err == 0
        return out
    one = np.zeros(2 * n, dtype=np.uint64)
    one[0::2] = 100                                                       # 1.00 as decimal(15,2)
    one_minus = call(1, one, (15, 2), _raw128(disc), (15, 2), (16, 2))
    disc_price = call(2, _raw128(ep), (15, 2), one_minus, (16, 2), (32, 4))
    one_plus = call(0, one, (15, 2), _raw128(tax), (15, 2), (16, 2))
    charge = call(2, disc_price, (32, 4), one_plus, (16, 2), (38, 6))
    for raw, want in ((disc_price, outs[0]), (charge, outs[1])):
        valid = [v is not None for v in want.to_pylist()]
        assert _from_raw128(raw, want.type, valid) == want.to_pylist()


# ---------------------------------------------------------------- numeric / date / hash scalars (round 3)
def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _project_one(name, args_arrays, args_types, out_type, lits=()):
    """The oracle's answer for fn(col0, col1, ..., *lits) over the given arrays."""
    import gandiva_amd as gandiva
    from oracle import oracle
    fields = [pa.field(f"c{i}", t) for i, t in enumerate(args_types)]
    batch = pa.RecordBatch.from_arrays(args_arrays, schema=pa.schema(fields))
    b = gandiva.TreeExprBuilder()
    node = b.make_function(name, [b.make_field(f) for f in fields] + [b.make_literal(v, t) for v, t in lits], out_type)
    return oracle.project_one(node, out_type, batch)


EDGE_F64 = [0.0, -0.0, 0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 0.49999999999999994, -0.49999999999999994, 4503599627370495.5,
            4503599627370497.0, -4503599627370497.0, 9007199254740993.0, 1e18, -1e18, 9.3e18, -9.3e18, 1e300, -1e300,
            2147483647.5, -2147483648.5, 2147483646.5, float("inf"), float("-inf"), float("nan"), 5e-324, 123456.5, -7.5]


@pytest.mark.parametrize("seed", range(3))
def test_device_round_and_float_casts_on_host(hostlib, seed):
    """round() and the float -> integer casts follow trunc(x +- 0.5) with saturation (round 3): the
    device functions on the CPU against the oracle, dense random values plus the edges of the rule."""
    rng = np.random.default_rng(5100 + seed)
    x = np.concatenate([np.array(EDGE_F64), rng.normal(0, 1e3, 2000), rng.uniform(-3, 3, 2000).round(1),
                        (rng.integers(-10**6, 10**6, 2000) + 0.5), rng.uniform(-1e19, 1e19, 500)]).astype(np.float64)
    n = len(x)
    arr = pa.array(x, pa.float64())
    outd, outi = np.zeros(n, np.float64), np.zeros(n, np.int64)
    for op, name, t in ((0, "round", pa.float64()), (1, "truncate", pa.float64()), (2, "castBIGINT", pa.int64()),
                        (3, "castINT", pa.int32())):
        hostlib.host_f64_op(op, _p(x), C.c_long(n), _p(outd), _p(outi))
        want = _project_one(name, [arr], [pa.float64()], t)
        if op < 2:
            got = pa.array(outd, pa.float64())
            w, g = np.array(want.to_pylist(), np.float64), np.array(got.to_pylist(), np.float64)
            assert np.array_equal(w.view(np.uint64)[~np.isnan(w)], g.view(np.uint64)[~np.isnan(w)]) and \
                np.array_equal(np.isnan(w), np.isnan(g)), name
        else:
            finite = np.isfinite(x)          # (NaN / inf -> integer: not defined by the rule; not compared)
            assert np.array(want.to_pylist(), np.int64)[finite].tolist() == outi[finite].tolist(), name


def test_device_timestamp_extraction_on_host(hostlib):
    rng = np.random.default_rng(77)
    ts = np.concatenate([np.array([0, -1, 1, 86399999, 86400000, -86400000, -86400001, 951782400000, 951868800000,
                                   -62135596800000, 253402300799999, 4102444800000, -2208988800000]),
                         rng.integers(-62135596800000, 253402300799999, 5000)]).astype(np.int64)
    n = len(ts)
    arr = pa.array(ts, pa.timestamp("ms"))
    out = np.zeros(n, np.int64)
    for op, name in enumerate(["extractYear", "extractMonth", "extractDay", "extractHour", "extractMinute", "extractSecond",
                               "extractDoy", "extractDow", "extractQuarter", "extractEpoch", "extractDecade", "extractCentury",
                               "extractMillennium"]):
        hostlib.host_extract_timestamp(op, _p(ts), C.c_long(n), _p(out))
        want = _project_one(name, [arr], [pa.timestamp("ms")], pa.int64())
        assert want.to_pylist() == out.tolist(), name


def test_device_hashes_on_host(hostlib):
    rng = np.random.default_rng(78)
    n = 3000
    iv = np.concatenate([np.array([0, 1, -1, 2**63 - 1, -2**63]), rng.integers(-2**62, 2**62, n - 5)]).astype(np.int64)
    fv = np.concatenate([np.array([0.0, -0.0, 1.0, float("inf"), 1e-300]), rng.normal(0, 1e6, n - 5)]).astype(np.float64)
    h32, h64 = np.zeros(n, np.int32), np.zeros(n, np.int64)
    for is_f, vals, t in ((0, iv, pa.int64()), (1, fv, pa.float64())):
        hostlib.host_hash_fixed(is_f, _p(vals), C.c_long(n), _p(h32), _p(h64))
        arr = pa.array(vals, t)
        assert _project_one("hash32", [arr], [t], pa.int32()).to_pylist() == h32.tolist()
        assert _project_one("hash64", [arr], [t], pa.int64()).to_pylist() == h64.tolist()
    words = ["".join(rng.choice(list("abcdefgh é日0123456789"), size=int(k))) for k in rng.integers(0, 70, n)]
    sarr = pa.array(words, pa.string())
    off = np.frombuffer(sarr.buffers()[1], np.int32)[: n + 1].copy()
    raw = sarr.buffers()[2]
    size = int(off[-1])
    data = np.concatenate([np.frombuffer(raw, np.uint8)[:size], np.zeros(16, np.uint8)])
    hostlib.host_hash_utf8(_p(off), _p(data), C.c_long(size), C.c_long(n), _p(h32), _p(h64))
    assert _project_one("hash32", [sarr], [pa.string()], pa.int32()).to_pylist() == h32.tolist()
    assert _project_one("hash64", [sarr], [pa.string()], pa.int64()).to_pylist() == h64.tolist()


def test_device_text_to_integer_casts_raise_and_hexadecimal_on_host(hostlib):
    """castBIGINT / castINT(text): blanks trimmed, then arrow::internal::ParseValue's rules incl.
    hexadecimal (round 3) — device functions on the CPU against the oracle, and rejected texts raise."""
    good64 = ["0", "-0", "  42 ", "9223372036854775807", "-9223372036854775808", "0x10", "0XfF", "0x7fffffffffffffff",
              "0xffffffffffffffff", "000123", "-000", " 0x0 ", "1", "-1"]
    good32 = ["0", "2147483647", "-2147483648", "0x7fffffff", "0xffffffff", "  -12  ", "0X1a"]
    bad = ["", " ", "abc", "+7", "+0", "1.5", "9223372036854775808", "-9223372036854775809", "0x", "0x1g", "1 2", "--1", "0x10000000000000000",
           "١٢", "1e3"]
    for wide, good, t, name in ((1, good64, pa.int64(), "castBIGINT"), (0, good32, pa.int32(), "castINT")):
        texts = good + (bad if wide else bad + ["2147483648", "-2147483649", "0x100000000"])
        n = len(texts)
        sarr = pa.array(texts, pa.string())
        off = np.frombuffer(sarr.buffers()[1], np.int32)[: n + 1].copy()
        size = int(off[-1])
        data = np.concatenate([np.frombuffer(sarr.buffers()[2], np.uint8)[:size], np.zeros(16, np.uint8)])
        out, flag = np.zeros(n, np.int64), np.zeros(n, np.uint8)
        hostlib.host_parse_int(wide, _p(off), _p(data), C.c_long(size), C.c_long(n), _p(out), _p(flag))
        assert flag[: len(good)].tolist() == [0] * len(good), [g for g, f in zip(good, flag) if f]
        assert flag[len(good):].tolist() == [1] * (n - len(good)), [b_ for b_, f in zip(texts[len(good):], flag[len(good):]) if not f]
        want = _project_one(name, [pa.array(good, pa.string())], [pa.string()], t).to_pylist()
        assert want == out[: len(good)].tolist(), name


def test_random_like_patterns_device_matcher_oracle_and_re2_agree(hostlib):
    """LIKE patterns drawn at random over literals, '%', '_' and multi-byte characters: the device library's general matcher
    (host build), the oracle's dynamic program and pyarrow.compute.match_like (RE2 underneath) give the same answers"""
    import pyarrow.compute as pc
    rng = np.random.default_rng(8)
    alpha = ["a", "b", "%", "_", "é", "ab", "%", "_", "語", "-", "a%", "_b"]
    words = ["", "a", "b", "ab", "ba", "aab", "abab", "é", "aé", "éa", "語", "a語b", "-", "a-b", "abba", "bab", "\n", "a\nb"]
    texts = ["".join(words[int(rng.integers(0, len(words)))] for _ in range(int(rng.integers(0, 4)))) for _ in range(400)] + words
    arr = pa.array(texts, pa.string())
    batch = pa.RecordBatch.from_arrays([arr], names=["s"])
    off, data, size = _col(arr)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    for _ in range(150):
        pat = "".join(alpha[int(rng.integers(0, len(alpha)))] for _ in range(int(rng.integers(0, 6))))
        want = pc.match_like(arr, pat).to_pylist()
        got = oracle.project_one(b.make_function("like", [s, b.make_literal(pat, pa.string())], pa.bool_()), pa.bool_(), batch).to_pylist()
        assert got == want, pat
        pb, pk, plen = _compile_like(pat, None)
        out = np.zeros(len(texts), dtype=np.uint8)
        hostlib.host_str_like(_p(off), _p(data), C.c_long(size), C.c_long(len(texts)), _p(pb), _p(pk), plen, 0, _p(out))
        assert out.astype(bool).tolist() == want, pat
