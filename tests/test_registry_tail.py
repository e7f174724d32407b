"""Round 4, registry tail: castVARCHAR(decimal128, n).  The text is Arrow's Decimal128::ToString(scale) —
pinned here against pyarrow's own decimal -> string cast (the same C++ routine the reference calls through
gdv_fn_dec_to_string) for the oracle AND for the product's device function compiled for the host; the GPU
tests then hold the kernels against the oracle."""
import ctypes as C
import decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from helpers import assert_bit_exact
from oracle import oracle
from test_device_lib_on_host import hostlib, _raw128  # noqa: F401  (fixture)

WIDE = decimal.Context(prec=80)
TYPES = [(38, 0), (10, 2), (38, 10), (38, 38), (20, 19), (15, 8), (38, 37), (5, 5), (12, 12), (19, 0), (20, 1), (7, 7)]


def decimals(p, s, seed, extra=300, null_every=11):
    rng = np.random.default_rng(seed)
    vals = [0, 1, -1, 7, -7, 10 ** p - 1, -(10 ** p - 1), 10 ** (p - 1), 123456789, -123456789012345678, 10 ** 19, 10 ** 19 - 1,
            -(10 ** 20), 5 * 10 ** 18, 10 ** 18, 99, -100, 1000000, 12345678901234567890123456789012345678]
    vals += [int(rng.integers(-10 ** 18, 10 ** 18)) * int(rng.integers(1, 10 ** 18)) for _ in range(extra)]
    vals += [int(rng.integers(-10 ** 6, 10 ** 6)) for _ in range(extra // 2)]
    vals += [int(rng.integers(-10 ** 18, 10 ** 18)) * 10 ** int(rng.integers(0, 20)) for _ in range(extra // 2)]
    vals = [v for v in vals if abs(v) < 10 ** p]
    py = [None if null_every and i % null_every == 3 else decimal.Decimal(v).scaleb(-s, context=WIDE) for i, v in enumerate(vals)]
    return pa.array(py, pa.decimal128(p, s))


def cast_expr(field, n):
    b = gandiva.TreeExprBuilder()
    return b.make_expression(b.make_function("castVARCHAR", [b.make_field(field), b.make_literal(n, pa.int64())], pa.string()),
                             pa.field("t", pa.string()))


def arrow_text(arr, n):
    full = pc.cast(arr, pa.string())
    return pc.utf8_slice_codeunits(full, 0, n) if n > 0 else pc.if_else(pc.is_valid(full), "", None)


@pytest.mark.parametrize("p,s", TYPES)
def test_oracle_text_of_a_decimal_is_arrows(p, s):
    arr = decimals(p, s, 100 * p + s)
    batch = pa.RecordBatch.from_arrays([arr], names=["d"])
    for n in (100, 44, 7, 1, 0):
        got = oracle.project([cast_expr(batch.schema.field(0), n)], batch)[0]
        assert got.equals(arrow_text(arr, n).cast(pa.string())), (p, s, n)


@pytest.mark.parametrize("p,s", TYPES)
def test_device_function_on_the_host_writes_arrows_text(hostlib, p, s):  # noqa: F811
    arr = decimals(p, s, 7 * p + s, null_every=0)
    raw = _raw128(arr)
    n = len(arr)
    for cut in (100, 20, 3):
        off = np.zeros(n + 1, dtype=np.int32)
        data = np.zeros(64 * n + 64, dtype=np.uint8)
        err = hostlib.host_cast_varchar_decimal(raw.ctypes.data_as(C.c_void_p), p, s, C.c_long(n), C.c_longlong(cut),
                                                off.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p))
        assert err == 0
        got = [bytes(data[off[i]:off[i + 1]]).decode() for i in range(n)]
        assert got == arrow_text(arr, cut).to_pylist(), (p, s, cut)


def test_a_negative_length_is_an_error_and_the_signature_is_registered():
    sigs = [s for s in gandiva.get_registered_function_signatures() if s.name() == "castVARCHAR"]
    assert any(pa.types.is_decimal(s.param_types()[0]) for s in sigs)
    arr = decimals(10, 2, 5)
    batch = pa.RecordBatch.from_arrays([arr], names=["d"])
    with pytest.raises(Exception):
        oracle.project([cast_expr(batch.schema.field(0), -1)], batch)


@pytest.mark.gpu
@pytest.mark.parametrize("p,s", TYPES)
def test_cast_decimal_to_text_on_the_gpu(p, s):
    arr = decimals(p, s, 31 * p + s, extra=3000)
    batch = pa.RecordBatch.from_arrays([arr], names=["d"])
    for n in (100, 9):
        e = cast_expr(batch.schema.field(0), n)
        proj = gandiva.make_projector(batch.schema, [e], None)
        got = proj.evaluate(batch)[0]
        assert_bit_exact(got, oracle.project([e], batch)[0], f"decimal128({p},{s}) -> text, cut {n}")
        assert got.equals(arrow_text(arr, n).cast(pa.string()))
        dev = proj.evaluate_device(gandiva.DeviceBatch.from_arrow(batch))[0].to_arrow()
        assert_bit_exact(dev, got, "HBM-resident")


@pytest.mark.gpu
def test_text_of_a_decimal_feeds_other_string_functions_and_selection_mode():
    arr = decimals(20, 4, 77, extra=2000)
    k = pa.array(np.arange(len(arr)) % 7, pa.int64())
    batch = pa.RecordBatch.from_arrays([arr, k], names=["d", "k"])
    b = gandiva.TreeExprBuilder()
    fd, fk = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    txt = b.make_function("castVARCHAR", [fd, b.make_literal(30, pa.int64())], pa.string())
    exprs = [b.make_expression(b.make_function("concat", [b.make_literal("[", pa.string()), txt, b.make_literal("]", pa.string())], pa.string()),
                               pa.field("c", pa.string())),
             b.make_expression(b.make_function("like", [txt, b.make_literal("%.12%", pa.string())], pa.bool_()), pa.field("l", pa.bool_())),
             b.make_expression(b.make_function("length", [txt], pa.int32()), pa.field("n", pa.int32())),
             b.make_expression(b.make_function("castVARCHAR", [b.make_function("multiply", [fd, fd], pa.decimal128(38, 6)), fk], pa.string()),
                               pa.field("sq", pa.string()))]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    for g, w in zip(proj.evaluate(batch), oracle.project(exprs, batch)):
        assert_bit_exact(g, w, "two-stage consumers of the text")
    cond = b.make_condition(b.make_function("greater_than", [fk, b.make_literal(3, pa.int64())], pa.bool_()))
    sel = gandiva.make_filter(batch.schema, cond).evaluate(batch, None, "int32")
    psel = gandiva.make_projector(batch.schema, exprs, None, "UINT32")
    taken = oracle.take_rows(batch, sel.to_array().to_numpy())
    for g, w in zip(psel.evaluate(batch, sel), oracle.project(exprs, taken)):
        assert_bit_exact(g, w, "selection mode")
    with pytest.raises(Exception):
        gandiva.make_projector(batch.schema, [cast_expr(batch.schema.field(0), -2)], None).evaluate(batch)


# ------------------------------------------------------------------ date_trunc_*, extractWeek / weekofyear, last_day

UNITS = ["Second", "Minute", "Hour", "Day", "Week", "Month", "Quarter", "Year", "Decade", "Century", "Millennium"]


def instants(seed, n=4000):
    rng = np.random.default_rng(seed)
    fixed = [0, -1, 1, 86399999, 86400000, -86400000, -86400001, 951782400000, 951868800000, 4102444800000, -2208988800000,
             1609459199999, 1609459200000, 1230768000000, 1262217600000, 1293753600000, 978307200000 - 1, 978307200000,
             -62135596800000, 253402300799999]
    return np.concatenate([np.array(fixed), rng.integers(-62135596800000, 253402300799999, n),
                           rng.integers(-3 * 10 ** 12, 5 * 10 ** 12, n)]).astype(np.int64)


def independent_answer(unit, ms):
    """pyarrow.compute where it has the operation, the calendar by hand elsewhere."""
    arr = pa.array(ms, pa.timestamp("ms"))
    if unit in ("Second", "Minute", "Hour", "Day"):
        # upstream's DATE_TRUNC_FIXED_UNIT is (millis / N) * N with C++ division: towards ZERO, so instants before
        # 1970 go up.  pyarrow floors; mirrored around zero it is the same rule.
        up = -pc.floor_temporal(pa.array(-ms, pa.timestamp("ms")), unit=unit.lower()).cast(pa.int64()).to_numpy()
        down = pc.floor_temporal(arr, unit=unit.lower()).cast(pa.int64()).to_numpy()
        return np.where(ms < 0, up, down)
    if unit in ("Month", "Quarter", "Year"):
        return pc.floor_temporal(arr, unit=unit.lower()).cast(pa.int64()).to_numpy()
    if unit == "Week":
        return pc.floor_temporal(arr, unit="week", week_starts_monday=True).cast(pa.int64()).to_numpy()
    if unit == "IsoWeek":
        return pc.iso_week(arr).to_numpy()
    years = pc.year(arr).to_numpy()
    if unit == "Last":
        months = pc.month(arr).to_numpy()
        nxt = np.array([np.datetime64(f"{y + (m == 12):04d}-{m % 12 + 1:02d}-01", "D") for y, m in zip(years, months)])
        return (nxt - np.timedelta64(1, "D")).astype("datetime64[ms]").astype(np.int64)
    start = {"Decade": (years - 1) // 10 * 10 + 1, "Century": (years - 1) // 100 * 100 + 1, "Millennium": (years - 1) // 1000 * 1000 + 1}[unit]
    return np.array([np.datetime64(f"{y:04d}-01-01", "ms") for y in start]).astype(np.int64)


def date_expr(name, field, out_type):
    b = gandiva.TreeExprBuilder()
    return b.make_expression(b.make_function(name, [b.make_field(field)], out_type), pa.field("r", out_type))


def date_cases():
    return ([(f"date_trunc_{u}", u, None) for u in UNITS] + [("extractWeek", "IsoWeek", pa.int64()), ("weekofyear", "IsoWeek", pa.int64()),
                                                             ("last_day", "Last", pa.date64())])


@pytest.mark.parametrize("t", [pa.timestamp("ms"), pa.date64()])
def test_oracle_unit_starts_iso_weeks_and_month_ends_agree_with_pyarrow_and_the_calendar(t):
    ms = instants(5)
    if pa.types.is_date64(t):
        ms = ms // 86400000 * 86400000
    ms = ms[ms >= -62135596800000 + 86400000 * 400]          # (year 1 onwards: numpy / pyarrow calendars agree there)
    batch = pa.RecordBatch.from_arrays([pa.array(ms, pa.int64()).cast(t)], names=["t"])
    for name, unit, rt in date_cases():
        got = oracle.project([date_expr(name, batch.schema.field(0), rt or t)], batch)[0]
        assert got.cast(pa.int64()).to_numpy().tolist() == independent_answer(unit, ms).tolist(), (name, str(t))


def test_device_date_functions_on_the_host(hostlib):  # noqa: F811
    ms = instants(6)
    ms = ms[ms >= -62135596800000 + 86400000 * 400]
    out = np.zeros(len(ms), np.int64)
    for op, unit in enumerate(UNITS + ["IsoWeek", "Last"]):
        hostlib.host_date_trunc_timestamp(op, ms.ctypes.data_as(C.c_void_p), C.c_long(len(ms)), out.ctypes.data_as(C.c_void_p))
        assert out.tolist() == independent_answer(unit, ms).tolist(), unit


@pytest.mark.gpu
@pytest.mark.parametrize("t", [pa.timestamp("ms"), pa.date64()])
def test_date_trunc_week_and_last_day_on_the_gpu(t):
    ms = instants(7, n=30_000)
    if pa.types.is_date64(t):
        ms = ms // 86400000 * 86400000
    mask = np.arange(len(ms)) % 13 == 5
    batch = pa.RecordBatch.from_arrays([pa.array(ms, pa.int64(), mask=mask).cast(t)], names=["t"])
    exprs = [date_expr(name, batch.schema.field(0), rt or t) for name, _, rt in date_cases()]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    for g, w, (name, _, _) in zip(proj.evaluate(batch), oracle.project(exprs, batch), date_cases()):
        assert_bit_exact(g, w, name)


# ------------------------------------------------------------------ round 5: initcap
# [recalled: string_ops.cc initcap_utf8 — "any character is considered as space, except if it is alphanumeric"]: the
# first letter of a word in upper case, the other letters in lower case, digits are word characters.  ASCII letters
# only on this backend (bytes >= 0x80: copied as they are, word characters — upstream maps them through utf8proc: a
# stated divergence, oracle header).  Second engines: a regular-expression restatement over bytes, and
# pyarrow.compute.utf8_title wherever the two rules coincide (ASCII text whose words hold no digit).

import re  # noqa: E402

STR = pa.string()
WORDS = ["hello", "WORLD", "mIxEd", "a", "Z", "x1y2", "42", "9lives", "o'neil", "snake_case", "kebab-case", "dotted.name",
         "tab\tsep", "two  spaces", " lead", "trail ", "", "ÉCOLE", "straße", "naïve café", "日本語 text", "aéb CÉD", "éx"]


def _initcap_by_regex(t):
    if t is None:
        return None
    def word(m):
        w = m.group(0)
        head = w[:1].upper() if w[0] < 0x80 else w[:1]
        return head + w[1:].lower()          # bytes.lower() / .upper() touch ASCII letters only
    return re.sub(rb"[A-Za-z0-9\x80-\xff]+", word, t.encode()).decode()


def _initcap_cases(seed, n):
    rng = np.random.default_rng(seed)
    seps = [" ", "  ", "-", "_", ".", ",", "'", "/", "\t", "1", ""]
    vals = []
    for _ in range(n):
        k = int(rng.integers(0, 6))
        t = "".join(str(rng.choice(WORDS)) + str(rng.choice(seps)) for _ in range(k))
        vals.append(None if rng.random() < 0.1 else t)
    return vals


def _initcap_exprs(b, s):
    f = lambda name, *a: b.make_function(name, list(a), STR)  # noqa: E731
    lit = lambda v, t=STR: b.make_literal(v, t)  # noqa: E731
    return [("initcap", f("initcap", s)), ("initcap(upper)", f("initcap", f("upper", s))),
            ("initcap(substr)", f("initcap", f("substr", s, lit(3, pa.int64()), lit(12, pa.int64())))),
            ("concat(initcap, '!')", f("concat", f("initcap", s), lit("!"))),
            ("upper(initcap) [two-stage]", f("upper", f("initcap", s)))]


def test_oracle_initcap_matches_the_regex_restatement_and_arrows_title_where_the_rules_coincide():
    vals = WORDS + _initcap_cases(3, 3000)
    batch = pa.RecordBatch.from_arrays([pa.array(vals, STR)], names=["s"])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    e = b.make_expression(b.make_function("initcap", [s], STR), pa.field("r", STR))
    got = oracle.project([e], batch)[0].to_pylist()
    assert got == [_initcap_by_regex(v) for v in vals]
    # pyarrow.compute.utf8_title starts a new word after anything that is not a LETTER (digits included) and knows
    # Unicode: same answers on ASCII text whose words carry no digit
    same = [v for v in vals if v is not None and v.isascii() and not re.search(r"\d", v)]
    title = pc.utf8_title(pa.array(same, STR)).to_pylist()
    assert [_initcap_by_regex(v) for v in same] == title
    assert gandiva.get_registered_function_signatures and any(
        sig.name() == "initcap" for sig in gandiva.get_registered_function_signatures())


def test_device_initcap_on_the_host(hostlib):  # noqa: F811
    vals = [v for v in WORDS + _initcap_cases(4, 2000) if v is not None]
    arr = pa.array(vals, STR)
    off = np.frombuffer(arr.buffers()[1], np.int32)[: len(vals) + 1].copy()
    raw = np.frombuffer(arr.buffers()[2], np.uint8) if arr.buffers()[2] is not None else np.zeros(0, np.uint8)
    size = int(off[-1])
    data = np.concatenate([raw[:size], np.zeros(64, np.uint8)])
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for mp, pre in ((0, lambda t: t), (1, lambda t: t.encode().upper().decode()), (2, lambda t: t.encode().lower().decode())):
        out_off, out_data = np.zeros(len(vals) + 1, np.int32), np.zeros(size + 64, np.uint8)
        hostlib.host_str_initcap(p(off), p(data), C.c_long(size), C.c_long(len(vals)), mp, p(out_off), p(out_data))
        got = [bytes(out_data[out_off[i]:out_off[i + 1]]).decode() for i in range(len(vals))]
        assert got == [_initcap_by_regex(pre(v)) for v in vals], mp


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1000, 40_003])
def test_initcap_on_the_gpu(n):
    vals = (WORDS + _initcap_cases(n, n))[:n] if n >= len(WORDS) else _initcap_cases(n, n)
    batch = pa.RecordBatch.from_arrays([pa.array(vals, STR)], names=["s"])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    exprs = [b.make_expression(node, pa.field(f"r{i}", STR)) for i, (_, node) in enumerate(_initcap_exprs(b, s))]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    for g, w, (name, _) in zip(proj.evaluate(batch), oracle.project(exprs, batch), _initcap_exprs(b, s)):
        assert_bit_exact(g, w, name)
    # under a selection vector (the wave-shaped selection-mode kernels)
    keep = np.arange(0, n, 3, dtype=np.uint32)
    sel = gandiva.SelectionVector(2, keep, len(keep))
    psel = gandiva.make_projector(batch.schema, exprs[:4], None, "UINT32")
    for g, w in zip(psel.evaluate(batch, sel), oracle.project(exprs[:4], oracle.take_rows(batch, keep))):
        assert_bit_exact(g, w, "selection mode")


# ------------------------------------------------------------------ round 5: regular expressions, the literal subset
# [recalled: regexp_like / regexp_matches = RE2::PartialMatch, regexp_replace = RE2::GlobalReplace].  The tree builder
# rewrites the literal subset onto like / replace (gdv_node.h MakeFunctionNode); everything else is CodeGenError.
# Second engine: Python's re (search / sub), whose semantics on metacharacter-free patterns are RE2's.

REGEX_OK = ["spark", "^spark", "spark$", "^spark$", "a", "^a", "k$", "ar", "é", "^日本", "x-y", "it's"]
REGEX_REFUSED = ["sp.rk", "spa*", "(spark)", "[sp]ark", "a|b", "^", "$", "", "50%", "a_b", "back\\slash", "x{2}", "q?"]
REPLACEMENTS = [("spark", "flink"), ("ar", ""), ("é", "e"), ("a", "AAA"), ("x-y", "-")]


def _regex_batch(n, seed=9):
    rng = np.random.default_rng(seed)
    base = W.c5_batch(n, 0.1).column(0).to_pylist()
    extra = ["spark", "sparks fly", "a spark", "x-y", "it's", "é", "日本語", "", "park", "sparkspark", "café x-y ar"]
    vals = [extra[int(rng.integers(0, len(extra)))] if rng.random() < 0.2 else v for v in base]
    return pa.RecordBatch.from_arrays([pa.array(vals, STR)], names=["s"])


def _regex_exprs(b, s):
    out = []
    for k, pat in enumerate(REGEX_OK):
        for fn in ("regexp_like", "regexp_matches"):
            out.append((f"{fn}({pat!r})", b.make_expression(b.make_function(fn, [s, b.make_literal(pat, STR)], pa.bool_()),
                                                            pa.field(f"m{k}{fn[-1]}", pa.bool_())), ("search", pat, None)))
    for k, (pat, to) in enumerate(REPLACEMENTS):
        out.append((f"regexp_replace({pat!r}, {to!r})",
                    b.make_expression(b.make_function("regexp_replace", [s, b.make_literal(pat, STR), b.make_literal(to, STR)], STR),
                                      pa.field(f"r{k}", STR)), ("sub", pat, to)))
    return out


def test_oracle_regexp_subset_matches_pythons_re():
    batch = _regex_batch(4000)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    cases = _regex_exprs(b, s)
    got = oracle.project([e for _, e, _ in cases], batch)
    vals = batch.column(0).to_pylist()
    for (name, _, (kind, pat, to)), g in zip(cases, got):
        rx = re.compile(pat)
        want = [None if v is None else (rx.search(v) is not None if kind == "search" else rx.sub(to, v)) for v in vals]
        assert g.to_pylist() == want, name


def test_regexps_beyond_the_literal_subset_are_refused_not_guessed(monkeypatch, tmp_path):
    """The rewritten forms compile for gfx950 (they ARE like / replace plans); a pattern with a metacharacter, a LIKE
    wildcard, nothing at all, or a replacement with a backslash reaches the planner as regexp_* and is refused there
    with CodeGenError — and the oracle refuses the same inputs."""
    from test_planner_cpu import _precompile
    batch = _regex_batch(64)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    ok = [e for _, e, _ in _regex_exprs(b, s)]
    files = _precompile(monkeypatch, tmp_path, batch.schema, exprs=ok[:6] + ok[-2:])
    text = "".join(open(tmp_path / f).read() for f in files)
    assert "regexp_" not in text.split("#include")[1]        # nothing of the regexp names is left below the @expr header
    assert "gdv_like" in text or "gdv_range_any" in text
    from gandiva_amd import _capi, gandiva as gg
    for pat in REGEX_REFUSED:
        for fn, extra, t in (("regexp_like", [], pa.bool_()), ("regexp_replace", [b.make_literal("z", STR)], STR)):
            e = b.make_expression(b.make_function(fn, [s, b.make_literal(pat, STR)] + extra, t), pa.field("r", t))
            sh = gg._make_schema(batch.schema)
            try:
                arr = (C.c_void_p * 1)(e._h)
                assert _capi.lib().gdv_precompile_projector(sh, arr, 1, 0) == 40, (fn, pat, _capi.last_error())
                assert "literal subset" in _capi.last_error()
            finally:
                _capi.lib().gdv_schema_free(sh)
            with pytest.raises(Exception):
                oracle.project([e], batch)
    e = b.make_expression(b.make_function("regexp_replace", [s, b.make_literal("a", STR), b.make_literal("\\1", STR)], STR), pa.field("r", STR))
    with pytest.raises(Exception):
        oracle.project([e], batch)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 1000, 50_003])
def test_regexp_subset_on_the_gpu(n):
    batch = _regex_batch(n, seed=n)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    cases = _regex_exprs(b, s)
    exprs = [e for _, e, _ in cases]
    # (one projector per group: a kernel takes one swept replace() needle and a few '%needle%' hooks)
    for lo in range(0, len(exprs), 6):
        part = exprs[lo:lo + 6]
        got = gandiva.make_projector(batch.schema, part, None).evaluate(batch)
        for g, w, (name, _, _) in zip(got, oracle.project(part, batch), cases[lo:lo + 6]):
            assert_bit_exact(g, w, name)
    cond = b.make_condition(b.make_function("regexp_like", [s, b.make_literal("^spark", STR)], pa.bool_()))
    sel = gandiva.make_filter(batch.schema, cond).evaluate(batch, None, "int32")
    assert sel.to_array().equals(oracle.filter_indices(cond, batch, "int32"))


# ------------------------------------------------------------------ round 5: hashSHA256 / hashSHA1 / hashMD5
# The digests are public standards: pinned against Python's hashlib.  [recalled] a number is hashed as the 8 bytes of
# (double)value, a NULL as the empty message, the result is lower-case hex and never null.

import hashlib  # noqa: E402
import struct  # noqa: E402

DIGESTS = [("hashSHA256", "sha256", hashlib.sha256, 0), ("hashSHA1", "sha1", hashlib.sha1, 1), ("hashMD5", "md5", hashlib.md5, 2)]
DIGEST_TEXTS = ["", "a", "abc", "message digest", "x" * 55, "y" * 56, "z" * 63, "w" * 64, "v" * 65, "u" * 119, "t" * 120, "s" * 1000,
                "The quick brown fox jumps over the lazy dog", "日本語テキスト", "é" * 40]


def _digest_batch(n, seed=5):
    rng = np.random.default_rng(seed)
    base = W.c5_batch(n, 0.1).column(0).to_pylist()
    vals = [(DIGEST_TEXTS[int(rng.integers(0, len(DIGEST_TEXTS)))] if rng.random() < 0.3 else v) for v in base]
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 12, n)
    x[:min(4, n)] = [0.0, -0.0, 1.0, -1.5][:n]
    k = rng.integers(-2 ** 40, 2 ** 40, n)
    m = rng.random(n) < 0.1
    return pa.RecordBatch.from_arrays([pa.array(vals, STR), pa.array(x, pa.float64(), mask=m), pa.array(k, pa.int64(), mask=m),
                                       pa.array((k % 1000).astype(np.int32), pa.int32())], names=["s", "x", "k", "i"])


def _digest_exprs(b, batch):
    s, x, k, i32 = (b.make_field(batch.schema.field(j)) for j in range(4))
    out = []
    for name, alias, fn, _ in DIGESTS:
        out += [(f"{name}(s)", b.make_function(name, [s], STR), ("s", fn)), (f"{alias}(upper(s))", b.make_function(alias, [b.make_function("upper", [s], STR)], STR), ("S", fn)),
                (f"{name}(x)", b.make_function(name, [x], STR), ("x", fn)), (f"{alias}(k)", b.make_function(alias, [k], STR), ("k", fn)),
                (f"{name}(i)", b.make_function(name, [i32], STR), ("i", fn))]
    return out


def _digest_want(batch, what, fn):
    col = {"s": 0, "S": 0, "x": 1, "k": 2, "i": 3}[what]
    want = []
    for v in batch.column(col).to_pylist():
        if v is None:
            msg = b""
        elif what == "s":
            msg = v.encode()
        elif what == "S":
            msg = v.encode().upper()
        else:
            msg = struct.pack("<d", float(v))
        want.append(fn(msg).hexdigest())
    return want


def test_oracle_digests_match_hashlib():
    batch = _digest_batch(1500)
    b = gandiva.TreeExprBuilder()
    cases = _digest_exprs(b, batch)
    exprs = [b.make_expression(node, pa.field(f"h{j}", STR)) for j, (_, node, _) in enumerate(cases)]
    for (name, _, (what, fn)), g in zip(cases, oracle.project(exprs, batch)):
        assert g.null_count == 0 and g.to_pylist() == _digest_want(batch, what, fn), name


def test_device_digests_on_the_host(hostlib):  # noqa: F811
    batch = _digest_batch(600, seed=8)
    arr = batch.column(0)
    vals = arr.to_pylist()
    filled = pa.array([v if v is not None else "" for v in vals], STR)
    off = np.frombuffer(filled.buffers()[1], np.int32)[: len(vals) + 1].copy()
    size = int(off[-1])
    data = np.concatenate([np.frombuffer(filled.buffers()[2], np.uint8)[:size], np.zeros(64, np.uint8)])
    valid = np.array([v is not None for v in vals], np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    for _, _, fn, algo in DIGESTS:
        nhex = fn(b"").digest_size * 2
        for mp, what in ((0, "s"), (1, "S")):
            out = np.zeros(64 * len(vals), np.uint8)
            hostlib.host_str_digest(algo, p(off), p(data), C.c_long(size), C.c_long(len(vals)), p(valid), mp, p(out))
            got = [bytes(out[64 * i:64 * i + nhex]).decode() for i in range(len(vals))]
            assert got == _digest_want(batch, what, fn), (algo, mp)
        x = np.asarray(batch.column(1).fill_null(0.0), np.float64)
        xv = np.array([v is not None for v in batch.column(1).to_pylist()], np.uint8)
        out = np.zeros(64 * len(x), np.uint8)
        hostlib.host_f64_digest(algo, p(x), C.c_long(len(x)), p(xv), p(out))
        got = [bytes(out[64 * i:64 * i + nhex]).decode() for i in range(len(x))]
        assert got == _digest_want(batch, "x", fn), algo


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1000, 20_011])
def test_digests_on_the_gpu(n):
    batch = _digest_batch(n, seed=n)
    b = gandiva.TreeExprBuilder()
    cases = _digest_exprs(b, batch)
    for lo in range(0, len(cases), 5):     # one projector per algorithm (a wave's LDS staging windows are three)
        part = cases[lo:lo + 5]
        for j in range(0, len(part), 3):
            exprs = [b.make_expression(node, pa.field(f"h{q}", STR)) for q, (_, node, _) in enumerate(part[j:j + 3])]
            got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
            for g, w, (name, _, _) in zip(got, oracle.project(exprs, batch), part[j:j + 3]):
                assert_bit_exact(g, w, name)
    # a digest as the argument of another function: two stages
    s = b.make_field(batch.schema.field(0))
    e = b.make_expression(b.make_function("upper", [b.make_function("substr", [b.make_function("hashMD5", [s], STR), b.make_literal(1, pa.int64()),
                                                                             b.make_literal(8, pa.int64())], STR)], STR), pa.field("u", STR))
    got = gandiva.make_projector(batch.schema, [e], None).evaluate(batch)
    assert_bit_exact(got[0], oracle.project([e], batch)[0], "upper(substr(hashMD5(s), 1, 8))")
