"""Shared comparison helpers for the parity tests."""
import numpy as np
import pyarrow as pa


def validity_np(arr):
    """bool numpy array: True where the slot is valid."""
    if arr.null_count == 0 and arr.buffers()[0] is None:
        return np.ones(len(arr), dtype=bool)
    buf = arr.buffers()[0]
    bits = np.unpackbits(np.frombuffer(buf, dtype=np.uint8), bitorder="little")
    return bits[arr.offset: arr.offset + len(arr)].astype(bool)


def values_bits_np(arr):
    """Raw bit image of every slot (uint8/16/32/64 view; bool -> 0/1)."""
    t = arr.type
    if pa.types.is_boolean(t):
        bits = np.unpackbits(np.frombuffer(arr.buffers()[1], dtype=np.uint8), bitorder="little")
        return bits[arr.offset: arr.offset + len(arr)]
    w = t.bit_width // 8
    if w == 16:  # decimal128: compare as (lo, hi) pairs packed into one structured value
        raw = np.frombuffer(arr.buffers()[1], dtype=np.dtype([("lo", np.uint64), ("hi", np.uint64)]))
        return raw[arr.offset: arr.offset + len(arr)]
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]
    raw = np.frombuffer(arr.buffers()[1], dtype=dt)
    return raw[arr.offset: arr.offset + len(arr)]


def assert_bit_exact(got, want, what=""):
    """Same type, same length, same validity bitmap, bit-identical values at valid slots
    (values under nulls are unspecified in Arrow and not compared)."""
    assert got.type == want.type, f"{what}: type {got.type} != {want.type}"
    assert len(got) == len(want), f"{what}: length {len(got)} != {len(want)}"
    if len(got) == 0:
        return
    gv, wv = validity_np(got), validity_np(want)
    if not np.array_equal(gv, wv):
        bad = np.flatnonzero(gv != wv)
        raise AssertionError(f"{what}: validity differs at {len(bad)} rows, first {bad[:8]}")
    if pa.types.is_string(got.type) or pa.types.is_binary(got.type):
        # var-len: the bytes of every valid row (offsets themselves may legitimately differ
        # under nulls); cast to binary so invalid UTF-8 cannot hide behind a decode error
        g, w = got.cast(pa.binary()).to_pylist(), want.cast(pa.binary()).to_pylist()
        if g != w:
            bad = [i for i, (x, y) in enumerate(zip(g, w)) if x != y]
            raise AssertionError(f"{what}: bytes differ at {len(bad)} rows, first {bad[:8]}: "
                                 f"{g[bad[0]]!r} vs {w[bad[0]]!r}")
        return
    gb, wb = values_bits_np(got)[wv], values_bits_np(want)[wv]
    if pa.types.is_floating(got.type):
        # every NaN is the same value: sign and payload of a generated NaN are not part of
        # IEEE-754 arithmetic semantics (x86 SSE yields -qNaN for inf-inf, CDNA4 +qNaN)
        ft = np.float32 if pa.types.is_float32(got.type) else np.float64
        gnan, wnan = np.isnan(gb.view(ft)), np.isnan(wb.view(ft))
        if not np.array_equal(gnan, wnan):
            bad = np.flatnonzero(gnan != wnan)
            raise AssertionError(f"{what}: NaN-ness differs at {len(bad)} valid rows, first {bad[:8]}")
        gb, wb = gb[~gnan], wb[~wnan]
    if not np.array_equal(gb, wb):
        bad = np.flatnonzero(gb != wb)
        raise AssertionError(f"{what}: values differ at {len(bad)} valid rows, first idx {bad[:8]}: "
                             f"{gb[bad[:4]]} vs {wb[bad[:4]]}")


def ulp_distance(g, w):
    """max distance in units of last place between two float64 numpy arrays."""
    both_nan = np.isnan(g) & np.isnan(w)
    gi = np.ascontiguousarray(g).view(np.int64)
    wi = np.ascontiguousarray(w).view(np.int64)
    gi = np.where(gi < 0, np.int64(-2**63) - gi, gi)
    wi = np.where(wi < 0, np.int64(-2**63) - wi, wi)
    d = np.abs(gi - wi)
    d[both_nan] = 0
    return int(d.max(initial=0))


def assert_within_ulp(got, want, ulps=1, what=""):
    """Floating point results of library math functions: <= `ulps` ULP at valid slots."""
    gv, wv = validity_np(got), validity_np(want)
    assert np.array_equal(gv, wv), f"{what}: validity differs"
    g = np.asarray(got.to_numpy(zero_copy_only=False), dtype=np.float64)[wv]
    w = np.asarray(want.to_numpy(zero_copy_only=False), dtype=np.float64)[wv]
    both_nan = np.isnan(g) & np.isnan(w)
    gi = g.view(np.int64)
    wi = w.view(np.int64)
    # map to a monotonic integer line
    gi = np.where(gi < 0, np.int64(-2**63) - gi, gi)
    wi = np.where(wi < 0, np.int64(-2**63) - wi, wi)
    d = np.abs(gi - wi)
    d[both_nan] = 0
    assert d.max(initial=0) <= ulps, f"{what}: max ulp distance {d.max()}"


def random_array(rng, t, n, null_fraction=0.1, special=True):
    """Random pyarrow array of fixed-width type t with edge values mixed in."""
    if pa.types.is_boolean(t):
        vals = rng.integers(0, 2, n).astype(bool)
    elif pa.types.is_floating(t):
        dt = np.float32 if pa.types.is_float32(t) else np.float64
        vals = rng.standard_normal(n).astype(dt) * dt(1000.0)
        if special and n >= 8:
            idx = rng.integers(0, n, 8)
            vals[idx] = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, np.finfo(dt).tiny / 4,
                                  np.finfo(dt).max, 1.0], dtype=dt)
    elif pa.types.is_integer(t):
        info = np.iinfo(t.to_pandas_dtype())
        small = max(info.min, -1000), min(info.max, 1000)
        vals = rng.integers(small[0], small[1], n, dtype=np.int64).astype(t.to_pandas_dtype())
        if special and n >= 4:
            idx = rng.integers(0, n, 4)
            vals[idx] = np.array([info.min, info.max, 0, 1], dtype=t.to_pandas_dtype())
    elif pa.types.is_date32(t):
        return pa.array(rng.integers(-30000, 60000, n).astype(np.int32), type=pa.int32(),
                        mask=_mask(rng, n, null_fraction)).cast(t)
    elif pa.types.is_date64(t):
        days = rng.integers(-30000, 60000, n).astype(np.int64)
        return pa.array(days * 86400000, type=pa.int64(), mask=_mask(rng, n, null_fraction)).cast(t)
    elif pa.types.is_timestamp(t):
        ms = rng.integers(-2_000_000_000_000, 4_000_000_000_000, n).astype(np.int64)
        return pa.array(ms, type=pa.int64(), mask=_mask(rng, n, null_fraction)).cast(t)
    else:
        raise NotImplementedError(str(t))
    return pa.array(vals, type=t, mask=_mask(rng, n, null_fraction))


def _mask(rng, n, null_fraction):
    if null_fraction <= 0:
        return None
    if null_fraction >= 1:
        return np.ones(n, dtype=bool)
    return rng.random(n) < null_fraction
