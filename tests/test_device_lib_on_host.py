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
