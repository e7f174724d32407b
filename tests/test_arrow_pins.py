"""Second, independent pin for the decimal128 and calendar semantics (round-1 verdict item 7).

The reference itself cannot be built or imported here, but the primitives its
Arrow-era decimal and date functions are built on do ship in the image with pyarrow:
`arrow::BasicDecimal256` (IncreaseScaleBy / ReduceScaleBy(round) / Divide / FitsInPrecision,
libarrow.so) and the vendored Hinnant `date.h`.  tests/cxx_pins/arrow_pins.cc restates the
operators ONLY in terms of those primitives; here the oracle is compared with it on the dense
38-digit generator and on random dates.  Three independent engines now agree on these
functions: the oracle (C), Python `decimal` / `datetime` (tests/test_decimal.py,
tests/test_oracle_crosscheck.py) and Arrow's own C++ primitives (this file).
What this does NOT pin: which rounding / overflow rule the reference chose — that remains
recollection (oracle header: "parity unpinned")."""
import ctypes as C
import os
import subprocess

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from oracle import oracle
from test_decimal import DENSE_TYPES, _dense_decimals, _nonzero, _result_type

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cxx_pins")
PA = os.path.dirname(pa.__file__)


@pytest.fixture(scope="module")
def pins():
    so = os.path.join(HERE, "libarrow_pins.so")
    src = os.path.join(HERE, "arrow_pins.cc")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        lib = [f for f in os.listdir(PA) if f.startswith("libarrow.so.")][0]
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-fPIC", "-shared", "-I", os.path.join(PA, "include"),
                               src, "-o", so, "-L", PA, "-l:" + lib, "-Wl,-rpath," + PA])
    return C.CDLL(so)


def _raw16(arr):
    return np.frombuffer(arr.buffers()[1], dtype=np.uint8)[arr.offset * 16:(arr.offset + len(arr)) * 16].copy()


@pytest.mark.parametrize("seed", range(10))
def test_oracle_decimal_ops_match_arrow_basic_decimal256(pins, seed):
    rng = np.random.default_rng(4200 + seed)
    ta = pa.decimal128(*DENSE_TYPES[int(rng.integers(0, len(DENSE_TYPES)))])
    tb = pa.decimal128(*DENSE_TYPES[int(rng.integers(0, len(DENSE_TYPES)))])
    n = 400
    a = _dense_decimals(rng, ta, n, null_fraction=0.0)
    bcol = _nonzero(_dense_decimals(rng, tb, n, null_fraction=0.0))
    batch = pa.RecordBatch.from_arrays([a, bcol], names=["a", "b"])
    b = gandiva.TreeExprBuilder()
    fa, fb = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    xa, xb = _raw16(a), _raw16(bcol)
    for code, op in enumerate(("add", "subtract", "multiply", "divide")):
        rt = _result_type(op, ta, tb)
        got = oracle.project_one(b.make_function(op, [fa, fb], rt), rt, batch)
        out = np.zeros(16 * n, dtype=np.uint8)
        pins.pin_decimal_binary(code, xa.ctypes.data_as(C.c_void_p), ta.scale, xb.ctypes.data_as(C.c_void_p),
                                tb.scale, rt.scale, out.ctypes.data_as(C.c_void_p), C.c_long(n))
        want = pa.Array.from_buffers(rt, n, [None, pa.py_buffer(out)])
        assert got.to_pylist() == want.to_pylist(), f"{op} {ta} {tb} -> {rt}"


def test_oracle_calendar_functions_match_vendored_date_h(pins):
    rng = np.random.default_rng(77)
    n = 5000
    days = np.concatenate([rng.integers(-200_000, 200_000, n - 6),
                           np.array([0, -1, 59, 60, 11016, -25567])]).astype(np.int64)  # incl. 1970-03-01, 2000-02-29
    ms = days * 86_400_000 + rng.integers(0, 86_400_000, n)
    ts = pa.array(ms, type=pa.int64()).cast(pa.timestamp("ms"))
    batch = pa.RecordBatch.from_arrays([ts], names=["t"])
    b = gandiva.TreeExprBuilder()
    t = b.make_field(batch.schema.field(0))
    out = [np.zeros(n, dtype=np.int32) for _ in range(5)]
    pins.pin_civil(days.ctypes.data_as(C.c_void_p), C.c_long(n), *[o.ctypes.data_as(C.c_void_p) for o in out])
    for name, want in zip(("extractYear", "extractMonth", "extractDay", "extractDoy", "extractDow"), out):
        got = oracle.project_one(b.make_function(name, [t], pa.int64()), pa.int64(), batch)
        assert np.array_equal(got.to_numpy(), want.astype(np.int64)), name
    for months in (1, -1, 12, -13, 25, 1200):
        got = oracle.project_one(b.make_function("timestampaddMonth", [b.make_literal(months, pa.int64()), t],
                                                 pa.timestamp("ms")), pa.timestamp("ms"), batch)
        want = np.zeros(n, dtype=np.int64)
        pins.pin_add_months(ms.ctypes.data_as(C.c_void_p), C.c_long(n), C.c_int(months), want.ctypes.data_as(C.c_void_p))
        assert np.array_equal(got.cast(pa.int64()).to_numpy(), want), f"timestampaddMonth {months}"


def test_oracle_integer_text_matches_arrow_formatter_and_parser(pins):
    """castVARCHAR(integer, n) against arrow::internal::StringFormatter<Int64Type>, castINT /
    castBIGINT(text) against arrow::internal::ParseValue on the blank-trimmed text — the Arrow
    primitives the reference lineage is believed to call (recollection), compiled from the headers
    and libarrow that ship with pyarrow."""
    import test_strings as S
    rng = np.random.default_rng(5)
    vals = np.concatenate([np.array([0, 1, -1, 9, 10, -10, 2**31 - 1, -2**31, 2**63 - 1, -2**63, 10**18, -10**18], np.int64),
                           rng.integers(-2**63, 2**63 - 1, 3000, dtype=np.int64) >> rng.integers(0, 63, 3000)])
    out = np.zeros(24 * len(vals), np.uint8)
    lens = np.zeros(len(vals), np.int32)
    pins.pin_format_int64(vals.ctypes.data_as(C.c_void_p), C.c_long(len(vals)), out.ctypes.data_as(C.c_void_p),
                          lens.ctypes.data_as(C.c_void_p))
    arrow_text = [bytes(out[24 * i:24 * i + lens[i]]).decode() for i in range(len(vals))]
    b = gandiva.TreeExprBuilder()
    batch = pa.RecordBatch.from_arrays([pa.array(vals, pa.int64())], names=["x"])
    x = b.make_field(batch.schema.field(0))
    for n in (30, 7, 1, 0):
        node = b.make_function("castVARCHAR", [x, b.make_literal(n, pa.int64())], pa.string())
        assert oracle.project_one(node, pa.string(), batch).to_pylist() == [t[:n] for t in arrow_text], n
    # text -> integer: same accept / reject decision and same value as Arrow's parser
    # (round 3: hexadecimal included — the oracle and the device code now follow ParseValue there too,
    # the divergence list of round 2 is empty)
    hexes = ["0x10", "0X1f", "0xFFFFFFFF", "0x7fffffff", "0x80000000", "0xffffffffffffffff", "0x8000000000000000",
             "0x123456789", "0x11111111111111111", "0x", "0xg", " 0x1A ", "-0x10", "0x-1", "00x10", "0x 1", "0x0000000000000001"]
    texts = S.NUMBER_TEXTS + [str(int(v)) for v in vals[:200]] + hexes + ["+0", "-00", "1e3", "١", " -12 ", "--1", "9" * 19, "-" + "9" * 19]
    for name, typ, bits in (("castINT", pa.int32(), 32), ("castBIGINT", pa.int64(), 64)):
        for text in texts:
            raw = text.strip(" ").encode()  # the stub trims blanks, then hands the text to ParseValue
            got = C.c_longlong(0)
            ok = pins.pin_parse_int(raw, len(raw), bits, C.byref(got))
            tb = pa.RecordBatch.from_arrays([pa.array([text], pa.string())], names=["s"])
            node = b.make_function(name, [b.make_field(tb.schema.field(0))], typ)
            if False:
                pass
            elif ok:
                assert oracle.project_one(node, typ, tb).to_pylist() == [got.value], (name, text)
            else:
                with pytest.raises(Exception, match="invalid argument"):
                    oracle.project_one(node, typ, tb)
