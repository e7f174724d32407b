"""decimal128 (BASELINE config C4).  Parity status: UNPINNED — the mounted reference has no
decimal code at all; result-type rules and round-half-up follow the
Arrow-era reference from memory.  What is checked:
  CPU: the oracle against Python's `decimal` module (exact arithmetic, ROUND_HALF_UP) and
       against pyarrow.compute where no scale adjustment happens;
  GPU: the HIP path bit-exact against the oracle, including scale-reduced multiplies,
       mixed scales, negatives, nulls, overflow -> 0."""
import decimal
import os

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from oracle import oracle
from helpers import assert_bit_exact

CTX = decimal.Context(prec=100, rounding=decimal.ROUND_HALF_UP)


def _dec_array(rng, t, n, digits, null_fraction=0.1):
    lim = 10 ** digits
    vals = [int(rng.integers(-lim, lim)) if digits <= 18 else
            int(rng.integers(-10**18, 10**18)) * 10 ** (digits - 18) + int(rng.integers(0, 10**9))
            for _ in range(n)]
    mask = rng.random(n) < null_fraction
    py = [None if m else decimal.Decimal(v).scaleb(-t.scale, CTX) for v, m in zip(vals, mask)]
    return pa.array(py, type=t)


def _result_type(op, a, b):
    p1, s1, p2, s2 = a.precision, a.scale, b.precision, b.scale
    if op in ("add", "subtract"):
        s = max(s1, s2); p = max(p1 - s1, p2 - s2) + s + 1
    elif op == "multiply":
        s = s1 + s2; p = p1 + p2 + 1
    elif op == "divide":
        s = max(6, s1 + p2 + 1); p = p1 - s1 + s2 + s
    else:  # mod
        s = max(s1, s2); p = min(p1 - s1, p2 - s2) + s
    if p > 38:
        delta = p - 38
        s = max(s - delta, min(s, 6)); p = 38
    return pa.decimal128(p, s)


def _python_expected(op, xs, ys, rt):
    q = decimal.Decimal(1).scaleb(-rt.scale)
    lim = decimal.Decimal(10) ** (38 - rt.scale)
    out = []
    for x, y in zip(xs, ys):
        if x is None or y is None:
            out.append(None); continue
        v = {"add": CTX.add, "subtract": CTX.subtract, "multiply": CTX.multiply}[op](x, y)
        v = v.quantize(q, rounding=decimal.ROUND_HALF_UP, context=CTX)
        out.append(decimal.Decimal(0).quantize(q) if abs(v) >= lim else v)
    return out


CASES = [  # (type a, digits a, type b, digits b)
    (pa.decimal128(15, 2), 12, pa.decimal128(15, 2), 12),
    (pa.decimal128(10, 0), 9, pa.decimal128(12, 5), 11),
    (pa.decimal128(38, 10), 30, pa.decimal128(38, 4), 25),   # multiply cuts the scale, rounds
    (pa.decimal128(30, 8), 28, pa.decimal128(20, 12), 18),
    (pa.decimal128(38, 6), 36, pa.decimal128(38, 6), 36),    # multiply overflows -> 0
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[2]}")
def test_oracle_matches_python_decimal(case):
    ta, da, tb, db = case
    rng = np.random.default_rng(da * 100 + db)
    n = 400
    a, bcol = _dec_array(rng, ta, n, da), _dec_array(rng, tb, n, db)
    batch = pa.RecordBatch.from_arrays([a, bcol], names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    for op in ("add", "subtract", "multiply"):
        rt = _result_type(op, ta, tb)
        got = oracle.project_one(b.make_function(op, [fa, fb], rt), rt, batch)
        want = _python_expected(op, a.to_pylist(), bcol.to_pylist(), rt)
        assert got.to_pylist() == want, f"{op} {ta} {tb} -> {rt}"
    for op, fn in (("less_than", lambda x, y: x < y), ("equal", lambda x, y: x == y),
                   ("greater_than_or_equal_to", lambda x, y: x >= y)):
        got = oracle.project_one(b.make_function(op, [fa, fb], pa.bool_()), pa.bool_(), batch)
        want = [None if x is None or y is None else fn(x, y) for x, y in zip(a.to_pylist(), bcol.to_pylist())]
        assert got.to_pylist() == want, op


def test_oracle_c4_matches_arrow_and_python():
    batch = W.c4_batch(3000, 0.1)
    out = oracle.project(W.c4_expressions(), batch)
    ep, disc, tax, ship = batch.columns
    one = pa.scalar(decimal.Decimal("1.00"), pa.decimal128(15, 2))
    assert out[0].equals(pc.multiply(ep, pc.subtract(one, disc)).cast(out[0].type))
    want = []
    for e, d, t in zip(ep.to_pylist(), disc.to_pylist(), tax.to_pylist()):
        if e is None or d is None or t is None:
            want.append(None); continue
        v = CTX.multiply(CTX.multiply(e, CTX.subtract(decimal.Decimal(1), d)), CTX.add(decimal.Decimal(1), t))
        want.append(v.quantize(decimal.Decimal("0.000001"), rounding=decimal.ROUND_HALF_UP, context=CTX))
    assert out[1].to_pylist() == want
    days = pc.subtract(pa.scalar(W.C4_DATE_1998_12_01, pa.int32()), ship.cast(pa.int32()))
    assert out[2].equals(days)


def _nonzero(arr):
    one = decimal.Decimal(1).scaleb(-arr.type.scale)
    return pa.array([one if v is not None and v == 0 else v for v in arr.to_pylist()], type=arr.type)


def _python_divmod_expected(op, xs, ys, rt):
    q = decimal.Decimal(1).scaleb(-rt.scale)
    lim = decimal.Decimal(10) ** (38 - rt.scale)
    out = []
    for x, y in zip(xs, ys):
        if x is None or y is None:
            out.append(None); continue
        if op == "divide":
            v = CTX.divide(x, y)
            # CTX.divide is already rounded to 100 digits: redo exactly with integers
            sx, sy = -x.as_tuple().exponent, -y.as_tuple().exponent
            ix, iy = int(x.scaleb(sx, CTX)), int(y.scaleb(sy, CTX))   # CTX: the default context would round to 28 digits
            num = abs(ix) * 10 ** (rt.scale - sx + sy)
            quo, rem = divmod(num, abs(iy))
            if 2 * rem >= abs(iy):
                quo += 1
            v = decimal.Decimal(quo if (ix < 0) == (iy < 0) else -quo).scaleb(-rt.scale, CTX)
        else:
            v = CTX.remainder(x, y)          # sign of the dividend, like C
            v = v.quantize(q, context=CTX)
        out.append(decimal.Decimal(0).quantize(q) if abs(v) >= lim else v)
    return out


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[2]}")
def test_oracle_divide_mod_match_python(case):
    ta, da, tb, db = case
    rng = np.random.default_rng(da * 100 + db + 7)
    n = 400
    a, bcol = _dec_array(rng, ta, n, da), _nonzero(_dec_array(rng, tb, n, db))
    batch = pa.RecordBatch.from_arrays([a, bcol], names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    for op in ("divide", "mod"):
        rt = _result_type(op, ta, tb)
        got = oracle.project_one(b.make_function(op, [fa, fb], rt), rt, batch)
        want = _python_divmod_expected(op, a.to_pylist(), bcol.to_pylist(), rt)
        assert got.to_pylist() == want, f"{op} {ta} {tb} -> {rt}"


def _dense_decimals(rng, t, n, null_fraction=0.05):
    """Every digit random (the CASES generator leaves runs of zeros in wide values)."""
    vals = []
    for _ in range(n):
        digits = int(rng.integers(1, t.precision + 1))
        v = int("".join(str(int(d)) for d in rng.integers(0, 10, digits)))
        v = -v if rng.random() < 0.5 else v
        vals.append(None if rng.random() < null_fraction else decimal.Decimal(v).scaleb(-t.scale, CTX))
    return pa.array(vals, type=t)


DENSE_TYPES = [(38, 0), (38, 10), (38, 37), (20, 5), (10, 2), (1, 0), (30, 15)]


@pytest.mark.parametrize("seed", range(12))
def test_oracle_decimal_ops_on_dense_random_digits(seed):
    """All five operators over fully random 1..38-digit operands and extreme (precision, scale)
    pairs, against exact integer / `decimal` arithmetic."""
    rng = np.random.default_rng(900 + seed)
    ta = pa.decimal128(*DENSE_TYPES[int(rng.integers(0, len(DENSE_TYPES)))])
    tb = pa.decimal128(*DENSE_TYPES[int(rng.integers(0, len(DENSE_TYPES)))])
    a, bcol = _dense_decimals(rng, ta, 250), _nonzero(_dense_decimals(rng, tb, 250))
    batch = pa.RecordBatch.from_arrays([a, bcol], names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    for op in ("add", "subtract", "multiply", "divide", "mod"):
        rt = _result_type(op, ta, tb)
        got = oracle.project_one(b.make_function(op, [fa, fb], rt), rt, batch)
        ref = _python_divmod_expected if op in ("divide", "mod") else _python_expected
        assert got.to_pylist() == ref(op, a.to_pylist(), bcol.to_pylist(), rt), f"{op} {ta} {tb} -> {rt}"


def test_oracle_decimal_divide_by_zero_raises():
    t = pa.decimal128(10, 2)
    batch = pa.RecordBatch.from_arrays(
        [pa.array([decimal.Decimal("1.00"), None], type=t), pa.array([decimal.Decimal("0.00"), decimal.Decimal("0.00")], type=t)],
        names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    rt = _result_type("divide", t, t)
    with pytest.raises(Exception, match="divide by zero"):
        oracle.project_one(b.make_function("divide", [fa, fb], rt), rt, batch)
    # a null dividend over a zero divisor is a null row, not an error
    ok = oracle.project_one(b.make_function("divide", [fa, fb], rt), rt, batch.slice(1))
    assert ok.to_pylist() == [None]


# ------------------------------------------------------------------ GPU parity

@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1000, 100003])
@pytest.mark.parametrize("nulls", [0.0, 0.1])
def test_hip_c4_matches_oracle(n, nulls):
    batch = W.c4_batch(n, nulls)
    exprs = W.c4_expressions()
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, str(e))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[2]}")
def test_hip_decimal_ops_match_oracle(case):
    ta, da, tb, db = case
    rng = np.random.default_rng(da * 100 + db + 1)
    n = 3000
    a, bcol = _dec_array(rng, ta, n, da), _dec_array(rng, tb, n, db)
    i64 = pa.array(rng.integers(-10**6, 10**6, n))
    batch = pa.RecordBatch.from_arrays([a, bcol, i64], names=["a", "b", "i"])
    b = gandiva.TreeExprBuilder()
    fa, fb, fi = (b.make_field(batch.schema.field(k)) for k in range(3))
    exprs = []
    for op in ("add", "subtract", "multiply"):
        rt = _result_type(op, ta, tb)
        exprs.append(b.make_expression(b.make_function(op, [fa, fb], rt), pa.field(op, rt)))
    for op in ("equal", "not_equal", "less_than", "less_than_or_equal_to", "greater_than",
               "greater_than_or_equal_to"):
        exprs.append(b.make_expression(b.make_function(op, [fa, fb], pa.bool_()), pa.field(op, pa.bool_())))
    exprs.append(b.make_expression(b.make_function("castFLOAT8", [fa], pa.float64()), pa.field("f", pa.float64())))
    exprs.append(b.make_expression(b.make_function("castDECIMAL", [fi], pa.decimal128(20, 4)),
                                   pa.field("d", pa.decimal128(20, 4))))
    exprs.append(b.make_expression(b.make_function("castDECIMAL", [fa], pa.decimal128(38, 1)),
                                   pa.field("r", pa.decimal128(38, 1))))
    exprs.append(b.make_expression(b.make_function("abs", [fa], ta), pa.field("abs", ta)))
    sel = b.make_if(b.make_function("less_than", [fa, fb], pa.bool_()), fa, b.make_function("negative", [fa], ta), ta)
    exprs.append(b.make_expression(sel, pa.field("sel", ta)))
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, str(e))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[2]}")
def test_hip_decimal_divide_mod_match_oracle(case):
    ta, da, tb, db = case
    rng = np.random.default_rng(da * 100 + db + 9)
    n = 3000
    a, bcol = _dec_array(rng, ta, n, da), _nonzero(_dec_array(rng, tb, n, db))
    batch = pa.RecordBatch.from_arrays([a, bcol], names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    exprs = []
    for op in ("divide", "mod"):
        rt = _result_type(op, ta, tb)
        exprs.append(b.make_expression(b.make_function(op, [fa, fb], rt), pa.field(op, rt)))
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, str(e))
    # and against exact integer arithmetic
    for g, e, op in zip(got, exprs, ("divide", "mod")):
        assert g.to_pylist() == _python_divmod_expected(op, a.to_pylist(), bcol.to_pylist(), g.type), op


@pytest.mark.gpu
def test_hip_decimal_divide_by_zero():
    t = pa.decimal128(10, 2)
    D = decimal.Decimal
    batch = pa.RecordBatch.from_arrays(
        [pa.array([D("1.00"), None, D("3.00")], type=t), pa.array([D("0.00"), D("0.00"), D("2.00")], type=t)],
        names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    for op in ("divide", "mod"):
        rt = _result_type(op, t, t)
        p = gandiva.make_projector(batch.schema, [b.make_expression(b.make_function(op, [fa, fb], rt), pa.field("r", rt))], None)
        with pytest.raises(Exception, match="divide by zero"):
            p.evaluate(batch)
        got, = p.evaluate(batch.slice(1))
        want = [None, D("1.50000000000000") if op == "divide" else D("1.00")]
        assert got.to_pylist() == [None if w is None else w.quantize(D(1).scaleb(-rt.scale)) for w in want]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_hip_decimal_ops_on_dense_random_digits(seed):
    rng = np.random.default_rng(900 + seed)
    ta = pa.decimal128(*DENSE_TYPES[int(rng.integers(0, len(DENSE_TYPES)))])
    tb = pa.decimal128(*DENSE_TYPES[int(rng.integers(0, len(DENSE_TYPES)))])
    a, bcol = _dense_decimals(rng, ta, 2500), _nonzero(_dense_decimals(rng, tb, 2500))
    batch = pa.RecordBatch.from_arrays([a, bcol], names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    exprs = []
    for op in ("add", "subtract", "multiply", "divide", "mod"):
        rt = _result_type(op, ta, tb)
        exprs.append(b.make_expression(b.make_function(op, [fa, fb], rt), pa.field(op, rt)))
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, str(e))


def _in_batch():
    import decimal as D
    t = pa.decimal128(20, 4)
    vals = [D.Decimal("1.5000"), D.Decimal("-99999999999999.9999"), None, D.Decimal("0.0000"), D.Decimal("7.2500"),
            D.Decimal("1234567890123456.7891")] * 40
    f = [0.0, -0.0, float("nan"), 1.5, -2.25, None] * 40
    return pa.RecordBatch.from_arrays([pa.array(vals, type=t), pa.array(f, type=pa.float64())], names=["d", "f"]), t


def _in_exprs(b, batch, t):
    d, f = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    return [b.make_expression(b.make_in_expression(d, ["1.5", "-99999999999999.9999", "42"], t), pa.field("din", pa.bool_())),
            b.make_expression(b.make_in_expression(f, [-0.0, float("nan"), -2.25], pa.float64()), pa.field("fin", pa.bool_()))]


def test_oracle_in_over_decimal_and_float_uses_value_equality():
    """IN over decimal128 (16-byte values) and over float64: -0.0 and +0.0 are one value, a NaN in
    the list or in the column matches nothing (what a hash set of doubles does)."""
    batch, t = _in_batch()
    b = gandiva.TreeExprBuilder()
    got = oracle.project(_in_exprs(b, batch, t), batch)
    import decimal as D
    want_d = [None if v is None else v in (D.Decimal("1.5"), D.Decimal("-99999999999999.9999"), D.Decimal(42))
              for v in batch.column(0).to_pylist()]
    want_f = [None if v is None else (v == 0.0 or v == -2.25) for v in batch.column(1).to_pylist()]
    assert got[0].to_pylist() == want_d
    assert got[1].to_pylist() == want_f


@pytest.mark.gpu
def test_hip_in_over_decimal_and_float_matches_oracle():
    batch, t = _in_batch()
    b = gandiva.TreeExprBuilder()
    exprs = _in_exprs(b, batch, t)
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, str(e))
    cond = b.make_condition(exprs[0].root())
    sel = gandiva.make_filter(batch.schema, cond).evaluate(batch, None)
    assert sel.to_array().equals(oracle.filter_indices(cond, batch, "int32"))


# ------------------------------------------------------------------ round 5: round / truncate / ceil / floor
# [recalled: precompiled/decimal_ops.cc Round / Truncate / Ceil / Floor; the result's (precision, scale) is the
# expression's own, chosen by the caller as in the lineage].  Second engine: Python's decimal quantize under
# ROUND_HALF_UP / ROUND_DOWN / ROUND_CEILING / ROUND_FLOOR, then the declared type's precision check.

ROUNDINGS = {"round": decimal.ROUND_HALF_UP, "truncate": decimal.ROUND_DOWN, "trunc": decimal.ROUND_DOWN,
             "ceil": decimal.ROUND_CEILING, "floor": decimal.ROUND_FLOOR}


def _python_rounded(fn, vals, ks, rt):
    out = []
    for v, k in zip(vals, ks):
        if v is None or k is None:
            out.append(None)
            continue
        if k < -38:
            r = decimal.Decimal(0)
        else:
            r = v if -v.as_tuple().exponent <= k else v.quantize(decimal.Decimal(1).scaleb(-k, CTX), rounding=ROUNDINGS[fn], context=CTX)
        r = r.quantize(decimal.Decimal(1).scaleb(-rt.scale, CTX), rounding=decimal.ROUND_HALF_UP, context=CTX)
        if abs(int(r.scaleb(rt.scale, CTX))) >= 10 ** rt.precision:
            r = decimal.Decimal(0).scaleb(-rt.scale, CTX)
        out.append(r)
    return out


def _rounding_cases(rng):
    """(function, input type, k column or None, declared result type)"""
    cases = []
    for (p, s) in DENSE_TYPES + [(12, 4), (9, 9)]:
        t = pa.decimal128(p, s)
        for fn in ("round", "truncate", "ceil", "floor"):
            cases.append((fn, t, False, pa.decimal128(min(38, p - s + 1) if p > s else 1, 0)))   # to an integer: one more digit
        for fn in ("round", "trunc"):
            cases.append((fn, t, True, pa.decimal128(38, s)))      # k per row, result kept at the input's scale
            cases.append((fn, t, True, pa.decimal128(38, max(s - 2, 0))))
    return cases


def _rounding_batch(rng, t, n):
    x = _dense_decimals(rng, t, n)
    ks = rng.integers(-6, t.scale + 4, n).astype(np.int32)
    edge = [0, t.scale, t.scale + 1, -1, -38, -39][:n]
    ks[:len(edge)] = edge
    kmask = rng.random(n) < 0.05
    return pa.RecordBatch.from_arrays([x, pa.array(ks, pa.int32(), mask=kmask)], names=["x", "k"])


def _rounding_node(b, batch, fn, with_k, rt):
    fx, fk = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    return b.make_function(fn, [fx, fk] if with_k else [fx], rt)


@pytest.mark.parametrize("seed", range(4))
def test_oracle_decimal_rounding_matches_python_decimal(seed):
    rng = np.random.default_rng(4100 + seed)
    b = gandiva.TreeExprBuilder()
    for fn, t, with_k, rt in _rounding_cases(rng):
        batch = _rounding_batch(rng, t, 200)
        got = oracle.project_one(_rounding_node(b, batch, fn, with_k, rt), rt, batch)
        ks = batch.column(1).to_pylist() if with_k else [0] * len(batch)
        assert got.to_pylist() == _python_rounded(fn, batch.column(0).to_pylist(), ks, rt), f"{fn} {t} k={with_k} -> {rt}"


def test_device_decimal_rounding_on_the_host():
    import ctypes as C
    from test_device_lib_on_host import LIB, SRC, HERE, _raw128, _from_raw128
    import subprocess
    hdr = os.path.join(HERE, "..", "gandiva_amd", "csrc", "gdv_device_lib.hpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unused-function", "-Wno-unused-variable", SRC, "-o", LIB])
    lib = C.CDLL(LIB)
    rng = np.random.default_rng(77)
    for fn, t, with_k, rt in _rounding_cases(rng):
        batch = _rounding_batch(rng, t, 150)
        x = batch.column(0)
        raw = _raw128(x)
        ks = np.asarray(batch.column(1).fill_null(0), dtype=np.int32)
        out = np.zeros(len(x) * 2, np.uint64)
        mode = {"round": 0, "truncate": 1, "trunc": 1, "ceil": 2, "floor": 3}[fn]
        lib.host_decimal_round(mode, raw.ctypes.data_as(C.c_void_p), t.precision, t.scale,
                               ks.ctypes.data_as(C.c_void_p) if with_k else None, rt.precision, rt.scale,
                               out.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
        valid = [v is not None for v in x.to_pylist()]
        got = _from_raw128(out, rt, valid)
        got = got.to_pylist() if hasattr(got, "to_pylist") else list(got)
        want = _python_rounded(fn, x.to_pylist(), ks.tolist() if with_k else [0] * len(x), rt)
        assert got == want, f"{fn} {t} k={with_k} -> {rt}"


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 1000, 30_011])
def test_hip_decimal_rounding_matches_oracle(n):
    rng = np.random.default_rng(n)
    b = gandiva.TreeExprBuilder()
    for t in (pa.decimal128(38, 10), pa.decimal128(12, 4), pa.decimal128(20, 5)):
        batch = _rounding_batch(rng, t, n)
        cases = [c for c in _rounding_cases(rng) if c[1] == t]
        exprs = [b.make_expression(_rounding_node(b, batch, fn, with_k, rt), pa.field(f"r{i}", rt))
                 for i, (fn, _, with_k, rt) in enumerate(cases)]
        got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
        for g, w, c in zip(got, oracle.project(exprs, batch), cases):
            assert g.equals(w), str(c)
