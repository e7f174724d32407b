"""utf8 / binary (BASELINE config C5: like / substr / upper over a var-len column).
Parity status: `like '%spark%'` and IN over strings are pinned by the reference lineage's
KATs (test_gandiva.py:117-129, 295-316, in tests/test_reference_kats.py); everything else is
UNPINNED (Arrow-era additions) and cross-checked against pyarrow.compute on the CPU.
GPU tests compare the HIP path (two-pass var-len outputs) bit-exactly with the oracle."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from oracle import oracle
from helpers import assert_bit_exact

WORDS = ["", "a", "spark", "sparkle", "bright spark and fire", "park", "Sp", "  padded  ",
         "ünïcödé spark", "日本語テキスト", "a_b%c", "100%", "under_score", "MiXeD CaSe 123", "x" * 300]


def _strings(rng, n, null_fraction=0.15):
    vals = [WORDS[i] if rng.random() < 0.7 else
            "".join(rng.choice(list("abspark_%XYZ é"), size=rng.integers(0, 24)))
            for i in rng.integers(0, len(WORDS), n)]
    mask = rng.random(n) < null_fraction
    return pa.array([None if m else v for v, m in zip(vals, mask)], type=pa.string())


def _exprs(b, s):
    def lit(v, t=pa.string()):
        return b.make_literal(v, t)
    i64 = pa.int64()
    out = []

    def add(name, node, t):
        out.append(b.make_expression(node, pa.field(name, t)))
    for i, pat in enumerate(["%spark%", "spark%", "%spark", "s_ark%", "%", "", "a_b%c", "%a%b%", "_%_"]):
        add(f"like{i}", b.make_function("like", [s, lit(pat)], pa.bool_()), pa.bool_())
    add("ilike0", b.make_function("ilike", [s, lit("%SpArK%")], pa.bool_()), pa.bool_())
    add("ilike1", b.make_function("ilike", [s, lit("mixed c_se%")], pa.bool_()), pa.bool_())
    add("ilike_up", b.make_function("ilike", [b.make_function("upper", [s], pa.string()), lit("%spark%")], pa.bool_()), pa.bool_())
    add("like_esc", b.make_function("like", [s, lit("100#%"), lit("#")], pa.bool_()), pa.bool_())
    add("like_esc2", b.make_function("like", [s, lit("a\\_b\\%c"), lit("\\")], pa.bool_()), pa.bool_())
    add("starts", b.make_function("starts_with", [s, lit("spa")], pa.bool_()), pa.bool_())
    add("ends", b.make_function("ends_with", [s, lit("rk")], pa.bool_()), pa.bool_())
    add("eq", b.make_function("equal", [s, lit("spark")], pa.bool_()), pa.bool_())
    add("lt", b.make_function("less_than", [s, lit("park")], pa.bool_()), pa.bool_())
    add("octets", b.make_function("octet_length", [s], pa.int32()), pa.int32())
    add("chars", b.make_function("char_length", [s], pa.int32()), pa.int32())
    add("in", b.make_in_expression(s, ["spark", "park", "", "日本語テキスト"], pa.string()), pa.bool_())
    add("isnull", b.make_function("isnull", [s], pa.bool_()), pa.bool_())
    add("upper", b.make_function("upper", [s], pa.string()), pa.string())
    add("lower", b.make_function("lower", [s], pa.string()), pa.string())
    for k, (f, c) in enumerate([(2, 5), (1, 1), (-3, 2), (0, 4), (5, 100), (50, 2), (-400, 3)]):
        add(f"substr{k}", b.make_function("substr", [s, b.make_literal(f, i64), b.make_literal(c, i64)],
                                          pa.string()), pa.string())
    add("substr_open", b.make_function("substr", [s, b.make_literal(3, i64)], pa.string()), pa.string())
    add("trim", b.make_function("btrim", [s], pa.string()), pa.string())
    add("up_sub", b.make_function("upper", [b.make_function("substr", [s, b.make_literal(2, i64),
                                                                    b.make_literal(3, i64)], pa.string())],
                                  pa.string()), pa.string())
    add("if_str", b.make_if(b.make_function("starts_with", [s, lit("s")], pa.bool_()), s, lit("other"),
                            pa.string()), pa.string())
    add("like_up", b.make_function("like", [b.make_function("upper", [s], pa.string()), lit("%SPARK%")],
                                   pa.bool_()), pa.bool_())
    return out


def test_oracle_strings_match_arrow_compute():
    rng = np.random.default_rng(9)
    s = _strings(rng, 3000)
    batch = pa.RecordBatch.from_arrays([s], names=["s"])
    b = gandiva.TreeExprBuilder()
    f = b.make_field(batch.schema.field(0))
    got = {e.result().name: r for e, r in zip(_exprs(b, f), oracle.project(_exprs(b, f), batch))}
    assert got["like0"].equals(pc.match_like(s, "%spark%"))
    assert got["like1"].equals(pc.match_like(s, "spark%"))
    assert got["like2"].equals(pc.match_like(s, "%spark"))
    assert got["like3"].equals(pc.match_like(s, "s_ark%"))
    assert got["like4"].equals(pc.match_like(s, "%"))
    assert got["like5"].equals(pc.equal(s, ""))
    assert got["like6"].equals(pc.match_like(s, "a_b%c"))
    assert got["like_esc2"].equals(pc.equal(s, "a_b%c"))
    assert got["like7"].equals(pc.match_like(s, "%a%b%"))
    assert got["like8"].equals(pc.greater_equal(pc.utf8_length(s), 2))
    assert got["like_esc"].equals(pc.equal(s, "100%"))
    assert got["starts"].equals(pc.starts_with(s, "spa"))
    assert got["ends"].equals(pc.ends_with(s, "rk"))
    assert got["eq"].equals(pc.equal(s, "spark"))
    assert got["lt"].equals(pc.less(s, "park"))
    assert got["octets"].equals(pc.binary_length(s))
    assert got["chars"].equals(pc.utf8_length(s))
    assert got["in"].equals(pc.is_in(s, value_set=pa.array(["spark", "park", "", "日本語テキスト"]), skip_nulls=True)
                            .filter(pa.array([True] * len(s))) if False else
                            pc.if_else(pc.is_null(s), pa.scalar(None, pa.bool_()),
                                       pc.is_in(s, value_set=pa.array(["spark", "park", "", "日本語テキスト"]))))
    assert got["isnull"].equals(pc.is_null(s))
    assert got["upper"].equals(pc.ascii_upper(s))
    assert got["lower"].equals(pc.ascii_lower(s))
    assert got["substr0"].equals(pc.utf8_slice_codeunits(s, 1, 6))
    assert got["substr1"].equals(pc.utf8_slice_codeunits(s, 0, 1))
    assert got["substr3"].equals(pc.utf8_slice_codeunits(s, 0, 4))
    assert got["substr4"].equals(pc.utf8_slice_codeunits(s, 4, 104))
    assert got["substr_open"].equals(pc.utf8_slice_codeunits(s, 2))
    assert got["trim"].equals(pc.utf8_trim(s, " "))
    assert got["up_sub"].equals(pc.ascii_upper(pc.utf8_slice_codeunits(s, 1, 4)))
    assert got["like_up"].equals(pc.match_like(pc.ascii_upper(s), "%SPARK%"))
    # substr with a negative start counts from the end; out of range -> ""
    py = s.to_pylist()
    want = [None if v is None else (v[len(v) - 3: len(v) - 1] if len(v) >= 3 else "") for v in py]
    assert got["substr2"].to_pylist() == want
    assert got["substr5"].to_pylist() == [None if v is None else v[49:51] for v in py]
    assert got["substr6"].to_pylist() == [None if v is None else "" if len(v) < 400 else v[len(v) - 400:len(v) - 397] for v in py]


def test_upper_lower_differ_from_utf8proc_exactly_on_rows_with_cased_non_ascii_letters():
    """PARITY.md, `upper lower`: this backend maps ASCII letters only (the lineage's earlier precompiled byte loop); the
    Arrow-era lineage maps every cased code point through utf8proc — which pyarrow.compute.utf8_upper / utf8_lower
    (the same utf8proc, linked into this image's libarrow) restate.  The divergence is exactly the rows that hold a
    non-ASCII letter utf8proc would change; everywhere else the two agree byte for byte.  And the reason it stays: the
    simple case map is not length-preserving."""
    rng = np.random.default_rng(12)
    alphabet = list("abcXYZ 09-_") + ["é", "É", "ß", "ñ", "Ω", "ω", "я", "Я", "ı", "ſ", "ÿ", "µ", "日", "本", "€", "ǆ", "ɐ"]
    rows = ["".join(alphabet[int(k)] for k in rng.integers(0, len(alphabet), int(rng.integers(0, 12)))) for _ in range(4000)]
    arr = pa.array(rows + [None], pa.string())
    batch = pa.RecordBatch.from_arrays([arr], names=["s"])
    b = gandiva.TreeExprBuilder()
    f = b.make_field(batch.schema.field(0))
    e = [b.make_expression(b.make_function("upper", [f], pa.string()), pa.field("u", pa.string())),
         b.make_expression(b.make_function("lower", [f], pa.string()), pa.field("l", pa.string()))]
    up, lo = oracle.project(e, batch)
    assert up.equals(pc.ascii_upper(arr)) and lo.equals(pc.ascii_lower(arr))
    want_up, want_lo = pc.utf8_upper(arr).to_pylist(), pc.utf8_lower(arr).to_pylist()
    changes_up = {c for c in alphabet if ord(c[0]) > 127 and pc.utf8_upper(pa.scalar(c)).as_py() != c}
    changes_lo = {c for c in alphabet if ord(c[0]) > 127 and pc.utf8_lower(pa.scalar(c)).as_py() != c}
    assert {"é", "ω", "я", "ı", "ſ", "ÿ", "µ", "ɐ"} <= changes_up and {"É", "Ω", "Я"} <= changes_lo
    differs = 0
    for r, u, l, wu, wl in zip(rows, up.to_pylist(), lo.to_pylist(), want_up, want_lo):
        assert (u != wu) == any(c in changes_up for c in r), r
        assert (l != wl) == any(c in changes_lo for c in r), r
        differs += u != wu
    assert differs > 1000
    # not length-preserving: why `upper` cannot stay on the flat path (output offsets = input offsets) if it followed utf8proc
    lengths = {c: (len(c.encode()), len(pc.utf8_upper(pa.scalar(c)).as_py().encode())) for c in ("ı", "ſ", "ɐ", "é")}
    assert lengths["ı"] == (2, 1) and lengths["ſ"] == (2, 1) and lengths["ɐ"] == (2, 3) and lengths["é"] == (2, 2)


def test_oracle_c5_matches_arrow_compute():
    batch = W.c5_batch(50000, 0.1)
    out = oracle.project(W.c5_expressions(), batch)
    s = batch.column(0)
    assert out[0].equals(pc.match_like(s, "%spark%"))
    assert out[1].equals(pc.utf8_slice_codeunits(s, 1, 6))
    assert out[2].equals(pc.utf8_upper(s))
    assert 0.03 < pc.mean(out[0].cast(pa.int8())).as_py() < 0.07


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 20011])
def test_hip_strings_match_oracle(n):
    rng = np.random.default_rng(n)
    s = _strings(rng, n)
    batch = pa.RecordBatch.from_arrays([s, pa.array(rng.integers(0, 9, n))], names=["s", "k"])
    b = gandiva.TreeExprBuilder()
    exprs = _exprs(b, b.make_field(batch.schema.field(0)))
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    want = oracle.project(exprs, batch)
    for g, w, e in zip(got, want, exprs):
        g.validate(full=True)
        assert g.equals(w), f"{e.result().name}: {e}"


@pytest.mark.gpu
@pytest.mark.parametrize("nulls", [0.0, 0.1])
def test_hip_c5_matches_oracle_host_and_device_paths(nulls):
    import torch
    n = 200003
    batch = W.c5_batch(n, nulls)
    exprs = W.c5_expressions()
    proj = gandiva.make_projector(batch.schema, exprs, None)
    want = oracle.project(exprs, batch)
    for g, w in zip(proj.evaluate(batch), want):
        assert g.equals(w)
    outs = proj.evaluate_device(gandiva.DeviceBatch.from_arrow(batch))
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        assert o.to_arrow().equals(w)
    # sliced input (array offset into the offsets buffer) and string filter + selection vector
    sl = batch.slice(1001, 70007)
    for g, w in zip(proj.evaluate(sl), oracle.project(exprs, sl)):
        assert g.equals(w)
    b = gandiva.TreeExprBuilder()
    f = b.make_field(batch.schema.field(0))
    cond = b.make_condition(b.make_function("like", [f, b.make_literal("%spark%", pa.string())], pa.bool_()))
    flt = gandiva.make_filter(batch.schema, cond)
    sel = flt.evaluate(batch, None)
    assert sel.to_array().equals(oracle.filter_indices(cond, batch, "int32"))
    p2 = gandiva.make_projector(batch.schema, exprs[1:], None, "UINT32")
    got = p2.evaluate(batch, sel)
    want2 = oracle.project(exprs[1:], oracle.take_rows(batch, sel.to_array().to_numpy()))
    for g, w in zip(got, want2):
        assert g.equals(w)


# ------------------------------------------------------------------ hash over var-len values

def _py_murmur3_x64_128_h1(data: bytes, seed: int) -> int:
    """Independent pure-Python MurmurHash3_x64_128, first 64 bits (signed)."""
    M = (1 << 64) - 1
    c1, c2 = 0x87c37b91114253d5, 0x4cf5ad432745937f
    rotl = lambda v, d: ((v << d) | (v >> (64 - d))) & M

    def fmix(k):
        k ^= k >> 33; k = k * 0xff51afd7ed558ccd & M; k ^= k >> 33
        k = k * 0xc4ceb9fe1a85ec53 & M; k ^= k >> 33
        return k
    h1 = h2 = seed & M
    nblocks = len(data) // 16
    for b in range(nblocks):
        k1 = int.from_bytes(data[16 * b: 16 * b + 8], "little")
        k2 = int.from_bytes(data[16 * b + 8: 16 * b + 16], "little")
        k1 = k1 * c1 & M; k1 = rotl(k1, 31); k1 = k1 * c2 & M; h1 ^= k1
        h1 = rotl(h1, 27); h1 = (h1 + h2) & M; h1 = (h1 * 5 + 0x52dce729) & M
        k2 = k2 * c2 & M; k2 = rotl(k2, 33); k2 = k2 * c1 & M; h2 ^= k2
        h2 = rotl(h2, 31); h2 = (h2 + h1) & M; h2 = (h2 * 5 + 0x38495ab5) & M
    tail = data[16 * nblocks:]
    if len(tail) > 8:
        k2 = int.from_bytes(tail[8:], "little")
        k2 = k2 * c2 & M; k2 = rotl(k2, 33); k2 = k2 * c1 & M; h2 ^= k2
    if tail:
        k1 = int.from_bytes(tail[:8], "little")
        k1 = k1 * c1 & M; k1 = rotl(k1, 31); k1 = k1 * c2 & M; h1 ^= k1
    h1 ^= len(data); h2 ^= len(data)
    h1 = (h1 + h2) & M; h2 = (h2 + h1) & M
    h1, h2 = fmix(h1), fmix(h2)
    h1 = (h1 + h2) & M
    return h1 - (1 << 64) if h1 >> 63 else h1


def _hash_exprs(b, s, k):
    out = []

    def add(name, node, t):
        out.append(b.make_expression(node, pa.field(name, t)))
    add("h32", b.make_function("hash32", [s], pa.int32()), pa.int32())
    add("h64", b.make_function("hash64", [s], pa.int64()), pa.int64())
    add("h32s", b.make_function("hash32", [s, b.make_literal(17, pa.int32())], pa.int32()), pa.int32())
    add("h64s", b.make_function("hash64", [s, k], pa.int64()), pa.int64())
    return out


def test_oracle_string_hashes_match_independent_murmur3():
    from sklearn.utils import murmurhash3_32
    rng = np.random.default_rng(5)
    n = 600
    s = _strings(rng, n)
    k = pa.array(rng.integers(-1000, 1000, n), pa.int64())
    batch = pa.RecordBatch.from_arrays([s, k], names=["s", "k"])
    b = gandiva.TreeExprBuilder()
    fs, fk = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    h32, h64, h32s, h64s = oracle.project(_hash_exprs(b, fs, fk), batch)
    assert h32.null_count == 0 and h64.null_count == 0          # hash is never null
    for i, (v, seed) in enumerate(zip(s.to_pylist(), k.to_pylist())):
        if v is None:   # null hashes to the seed
            assert (h32[i].as_py(), h64[i].as_py(), h32s[i].as_py(), h64s[i].as_py()) == (0, 0, 17, seed)
            continue
        raw = v.encode()
        assert h32[i].as_py() == murmurhash3_32(raw, seed=0, positive=False), v
        assert h32s[i].as_py() == murmurhash3_32(raw, seed=17, positive=False), v
        assert h64[i].as_py() == _py_murmur3_x64_128_h1(raw, 0), v
        # the seed is narrowed to int32, then sign-extended into both lanes
        assert h64s[i].as_py() == _py_murmur3_x64_128_h1(raw, seed), v


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1000, 30011])
def test_hip_string_hashes_match_oracle(n):
    from helpers import assert_bit_exact
    rng = np.random.default_rng(n + 77)
    s = _strings(rng, n)
    k = pa.array(rng.integers(-1000, 1000, n), pa.int64())
    batch = pa.RecordBatch.from_arrays([s, k, s.cast(pa.binary())], names=["s", "k", "raw"])
    b = gandiva.TreeExprBuilder()
    fs, fk, fr = (b.make_field(batch.schema.field(i)) for i in range(3))
    exprs = _hash_exprs(b, fs, fk)
    exprs += [b.make_expression(b.make_function("hash64", [fr], pa.int64()), pa.field("hb", pa.int64())),
              b.make_expression(b.make_function("hash32", [b.make_function("upper", [fs], pa.string())],
                                                pa.int32()), pa.field("hup", pa.int32())),
              b.make_expression(b.make_function("hash64", [b.make_function(
                  "substr", [fs, b.make_literal(2, pa.int64()), b.make_literal(20, pa.int64())], pa.string())],
                  pa.int64()), pa.field("hsub", pa.int64()))]
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, str(e))


# ------------------------------------------------------------------ left / right / castVARCHAR / locate / strpos / ascii

def _position_exprs(b, s, t):
    i32, i64 = pa.int32(), pa.int64()
    out = []

    def add(name, node, typ):
        out.append(b.make_expression(node, pa.field(name, typ)))
    for k in (0, 1, 3, 100, -1, -4, -100):
        add(f"left{k}", b.make_function("left", [s, b.make_literal(k, i32)], pa.string()), pa.string())
        add(f"right{k}", b.make_function("right", [s, b.make_literal(k, i32)], pa.string()), pa.string())
    for k in (0, 2, 7, 1000):
        add(f"cast{k}", b.make_function("castVARCHAR", [s, b.make_literal(k, i64)], pa.string()), pa.string())
    for sub in ("spark", "a", "é", "日本", ""):
        add(f"locate_{sub}", b.make_function("locate", [b.make_literal(sub, pa.string()), s], i32), i32)
        add(f"strpos_{sub}", b.make_function("strpos", [s, b.make_literal(sub, pa.string())], i32), i32)
    add("locate_from3", b.make_function("locate", [b.make_literal("a", pa.string()), s, b.make_literal(3, i32)], i32), i32)
    add("locate_col", b.make_function("locate", [t, s], i32), i32)
    add("ascii", b.make_function("ascii", [s], i32), i32)
    add("ascii_up", b.make_function("ascii", [b.make_function("upper", [s], pa.string())], i32), i32)
    return out


def _python_positions(vals, subs):
    """The same expressions in plain Python (str = sequence of characters)."""
    cols = []

    def col(fn):
        cols.append([None if v is None else fn(v) for v in vals])
    for k in (0, 1, 3, 100, -1, -4, -100):
        col(lambda v, k=k: "" if k == 0 else (v[:k] if k > 0 else v[:max(len(v) + k, 0)]))
        col(lambda v, k=k: "" if k == 0 else (v[max(len(v) - k, 0):] if k > 0 else v[min(-k, len(v)):]))
    for k in (0, 2, 7, 1000):
        col(lambda v, k=k: v[:k])
    for sub in ("spark", "a", "é", "日本", ""):
        f = lambda v, sub=sub: 0 if not v or not sub else v.find(sub) + 1
        col(f)
        col(f)
    col(lambda v: 0 if not v else v.find("a", 2) + 1)
    cols.append([None if v is None or t is None else (0 if not v or not t else v.find(t) + 1)
                 for v, t in zip(vals, subs)])
    first = lambda v: 0 if not v else int(np.int8(np.uint8(v.encode()[0])))
    col(first)
    col(lambda v: 0 if not v else int(np.int8(np.uint8(v.encode()[0] - 32 if "a" <= v[0] <= "z" else v.encode()[0]))))
    return cols


def _position_batch(n, seed):
    rng = np.random.default_rng(seed)
    s = _strings(rng, n)
    t = pa.array([None if rng.random() < 0.1 else ["a", "spark", "rk", "é", ""][int(rng.integers(0, 5))]
                  for _ in range(n)], pa.string())
    return pa.RecordBatch.from_arrays([s, t], names=["s", "t"])


def test_oracle_position_functions_match_python():
    batch = _position_batch(700, 9)
    b = gandiva.TreeExprBuilder()
    exprs = _position_exprs(b, b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1)))
    got = oracle.project(exprs, batch)
    want = _python_positions(batch.column(0).to_pylist(), batch.column(1).to_pylist())
    for g, w, e in zip(got, want, exprs):
        assert g.to_pylist() == w, e.result().name


def test_oracle_position_functions_raise_on_bad_arguments():
    batch = _position_batch(10, 1)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    for node in (b.make_function("castVARCHAR", [s, b.make_literal(-1, pa.int64())], pa.string()),
                 b.make_function("locate", [b.make_literal("a", pa.string()), s, b.make_literal(0, pa.int32())],
                                 pa.int32())):
        with pytest.raises(Exception, match="invalid argument"):
            oracle.project_one(node, node.return_type(), batch)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1000, 20011])
def test_hip_position_functions_match_oracle(n):
    from helpers import assert_bit_exact
    batch = _position_batch(n, n + 3)
    b = gandiva.TreeExprBuilder()
    s, t = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    exprs = _position_exprs(b, s, t)
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        assert_bit_exact(g, w, e.result().name)
    for node in (b.make_function("castVARCHAR", [s, b.make_literal(-1, pa.int64())], pa.string()),
                 b.make_function("locate", [b.make_literal("a", pa.string()), s, b.make_literal(0, pa.int32())],
                                 pa.int32())):
        p = gandiva.make_projector(batch.schema, [b.make_expression(node, pa.field("r", node.return_type()))], None)
        with pytest.raises(Exception, match="invalid argument"):
            p.evaluate(batch)


# ------------------------------------------------------------------ concat / ||

def _concat_exprs(b, s, t):
    lit = lambda v: b.make_literal(v, pa.string())
    S = pa.string()
    out = []

    def add(name, node):
        out.append(b.make_expression(node, pa.field(name, S)))
    add("concat2", b.make_function("concat", [s, t], S))
    add("pipes2", b.make_function("concatOperator", [s, t], S))
    add("concat3", b.make_function("concat", [s, lit(" - "), b.make_function("upper", [t], S)], S))
    add("pipes3", b.make_function("concatOperator", [b.make_function("substr", [s, b.make_literal(2, pa.int64()),
                                                                              b.make_literal(3, pa.int64())], S),
                                                 lit("|"), b.make_function("lower", [s], S)], S))
    add("nested", b.make_function("concat", [b.make_function("concatOperator", [s, t], S), lit("."),
                                             b.make_function("concat", [t, t], S)], S))
    add("six", b.make_function("concat", [s, t, s, lit(""), t, lit("日本")], S))
    return out


def _python_concat(ss, ts):
    e = lambda v: "" if v is None else v
    cols = [[e(s) + e(t) for s, t in zip(ss, ts)],
            [None if s is None or t is None else s + t for s, t in zip(ss, ts)],
            [e(s) + " - " + ("" if t is None else _ascii_upper(t)) for s, t in zip(ss, ts)],
            [None if s is None else s[1:4] + "|" + _ascii_lower(s) for s in ss],
            [("" if s is None or t is None else s + t) + "." + e(t) + e(t) for s, t in zip(ss, ts)],
            [e(s) + e(t) + e(s) + e(t) + "日本" for s, t in zip(ss, ts)]]
    return cols


def _ascii_upper(v):
    return "".join(chr(ord(c) - 32) if "a" <= c <= "z" else c for c in v)


def _ascii_lower(v):
    return "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in v)


def test_oracle_concat_matches_python():
    batch = _position_batch(500, 21)
    b = gandiva.TreeExprBuilder()
    exprs = _concat_exprs(b, b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1)))
    got = oracle.project(exprs, batch)
    want = _python_concat(batch.column(0).to_pylist(), batch.column(1).to_pylist())
    for g, w, e in zip(got, want, exprs):
        assert g.to_pylist() == w, e.result().name


def test_concat_results_feed_other_functions_through_a_first_stage():
    from gandiva_amd import _capi, gandiva as gg
    import ctypes as C
    schema = pa.schema([("s", pa.string())])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(schema.field(0))
    cc = b.make_function("concat", [s, s], pa.string())
    e = b.make_expression(b.make_function("like", [cc, b.make_literal("%a%", pa.string())], pa.bool_()),
                          pa.field("r", pa.bool_()))
    arr = (C.c_void_p * 1)(e._h)
    # round 2: the concat is hoisted into a first-stage kernel (tests/test_registry_tail.py) ...
    assert _capi.lib().gdv_precompile_projector(gg._make_schema(schema), arr, 1, 0) == 0, _capi.last_error()
    # ... round 3: also under a selection vector — the first stage is built in the same mode and
    # evaluates the selected rows only (tests/test_registry_tail.py)
    assert _capi.lib().gdv_precompile_projector(gg._make_schema(schema), arr, 1, 2) == 0, _capi.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1000, 20011])
def test_hip_concat_matches_oracle(n):
    from helpers import assert_bit_exact
    batch = _position_batch(n, n + 5)
    b = gandiva.TreeExprBuilder()
    s, t = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    exprs = _concat_exprs(b, s, t)
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, oracle.project(exprs, batch), exprs):
        g.validate(full=True)
        assert_bit_exact(g, w, e.result().name)


# ------------------------------------------------------------------ var-len edge cases

@pytest.mark.gpu
def test_varlen_outputs_with_an_empty_selection():
    batch = W.c5_batch(5000, 0.1)
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    none = b.make_condition(b.make_function("like", [s, b.make_literal("no such thing%", pa.string())], pa.bool_()))
    sv = gandiva.make_filter(batch.schema, none).evaluate(batch, None, "int32")
    assert sv.num_slots == 0
    proj = gandiva.make_projector(batch.schema, W.c5_expressions(), None, "UINT32")
    got = proj.evaluate(batch, sv)
    assert [len(g) for g in got] == [0, 0, 0]
    for g in got:
        g.validate(full=True)


@pytest.mark.gpu
def test_varlen_multi_chunk_tile_scan_matches_oracle():
    """> 4096 wave tiles per output: the three-launch segmented scan (small batches take the
    single-launch form)."""
    n = 3_000_017
    batch = W.c5_batch(n, 0.05)
    exprs = W.c5_expressions()
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w in zip(got, oracle.project(exprs, batch)):
        assert g.equals(w)


@pytest.mark.gpu
def test_varlen_output_over_2_gib_is_rejected():
    import torch
    n = 100_000_000
    db = W.c5_device_batch(n)                    # 1.2 GB of string bytes
    b = gandiva.TreeExprBuilder()
    s = b.make_field(W.c5_schema().field(0))
    big = b.make_expression(b.make_function("concat", [s, s], pa.string()), pa.field("ss", pa.string()))
    proj = gandiva.make_projector(W.c5_schema(), [big], None)
    with pytest.raises(pa.ArrowInvalid, match="2 GiB"):
        proj.evaluate_device(db)
    torch.cuda.synchronize()


# ------------------------------------------------------------------ castINT / castBIGINT from text

NUMBER_TEXTS = ["0", "7", "-7", "  42  ", "-0", "007", "2147483647", "-2147483648", "2147483648", "-2147483649",
                "9223372036854775807", "-9223372036854775808", "9223372036854775808", "-9223372036854775809",
                "", " ", "-", "+5", "1 2", "12a", "a12", "1.5", "١٢", "0000000000000000000000000000000000000123",
                "99999999999999999999999999999999999999999", "- 5", "5-"]


def _python_parse(text, bits):
    import re as _re
    t = text.strip(" ")
    if not _re.fullmatch(r"-?[0-9]+", t):
        return "error"
    v = int(t)
    return v if -(1 << (bits - 1)) <= v < (1 << (bits - 1)) else "error"


@pytest.mark.parametrize("bits,name,typ", [(32, "castINT", pa.int32()), (64, "castBIGINT", pa.int64())])
def test_oracle_text_to_integer_casts_match_python(bits, name, typ):
    b = gandiva.TreeExprBuilder()
    for text in NUMBER_TEXTS:
        batch = pa.RecordBatch.from_arrays([pa.array([text, None], pa.string())], names=["s"])
        node = b.make_function(name, [b.make_field(batch.schema.field(0))], typ)
        want = _python_parse(text, bits)
        if want == "error":
            with pytest.raises(Exception, match="invalid argument"):
                oracle.project_one(node, typ, batch)
        else:
            assert oracle.project_one(node, typ, batch).to_pylist() == [want, None], text


def test_text_to_integer_casts_compile_for_gfx950():
    import ctypes as C
    from gandiva_amd import _capi, gandiva as gg
    schema = pa.schema([("s", pa.string())])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(schema.field(0))
    exprs = [b.make_expression(b.make_function("castINT", [s], pa.int32()), pa.field("i", pa.int32())),
             b.make_expression(b.make_function("castBIGINT", [b.make_function("btrim", [s], pa.string())], pa.int64()),
                               pa.field("l", pa.int64()))]
    arr = (C.c_void_p * len(exprs))(*[e._h for e in exprs])
    assert _capi.lib().gdv_precompile_projector(gg._make_schema(schema), arr, len(exprs), 0) == 0, _capi.last_error()


@pytest.mark.gpu
def test_hip_text_to_integer_casts_match_oracle():
    from helpers import assert_bit_exact
    good = [t for t in NUMBER_TEXTS if _python_parse(t, 32) != "error"]
    rng = np.random.default_rng(12)
    vals = [good[int(rng.integers(0, len(good)))] if rng.random() < 0.5 else str(int(rng.integers(-2**31, 2**31)))
            for _ in range(5000)]
    batch = pa.RecordBatch.from_arrays([pa.array([None if rng.random() < 0.1 else v for v in vals], pa.string())],
                                       names=["s"])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    exprs = [b.make_expression(b.make_function("castINT", [s], pa.int32()), pa.field("i", pa.int32())),
             b.make_expression(b.make_function("castBIGINT", [s], pa.int64()), pa.field("l", pa.int64()))]
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w in zip(got, oracle.project(exprs, batch)):
        assert_bit_exact(g, w)
    bad = pa.RecordBatch.from_arrays([pa.array(["12", "x1"], pa.string())], names=["s"])
    with pytest.raises(Exception, match="invalid argument"):
        gandiva.make_projector(bad.schema, exprs, None).evaluate(bad)


@pytest.mark.gpu
def test_flat_outputs_fall_back_when_null_rows_carry_bytes():
    """`upper(s)` / pass-through outputs are evaluated optimistically flat (offsets = input offsets,
    bytes = mapped input span).  Arrow allows a NULL row to own bytes; then the flat copy would
    emit them: the kernel must notice (NOTFLAT) and the batch is re-run on the general path."""
    from helpers import assert_bit_exact
    rng = np.random.default_rng(31)
    n = 3000
    lens = rng.integers(0, 12, n)
    offsets = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=offsets[1:])
    data = rng.integers(97, 123, int(offsets[-1])).astype(np.uint8)
    nulls = rng.random(n) < 0.2                       # null rows KEEP their bytes
    arr = pa.Array.from_buffers(pa.string(), n, [pa.py_buffer(np.packbits(~nulls, bitorder="little")),
                                                 pa.py_buffer(offsets), pa.py_buffer(data)])
    assert any(nulls[i] and lens[i] > 0 for i in range(n))
    batch = pa.RecordBatch.from_arrays([arr], names=["s"])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    exprs = [b.make_expression(b.make_function("upper", [s], pa.string()), pa.field("up", pa.string())),
             b.make_expression(s, pa.field("same", pa.string())),
             b.make_expression(b.make_function("substr", [s, b.make_literal(2, pa.int64()), b.make_literal(3, pa.int64())],
                                               pa.string()), pa.field("sub", pa.string()))]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    for bt in (batch, batch.slice(1024, 1500)):
        got = proj.evaluate(bt)
        for g, w in zip(got, oracle.project(exprs, bt)):
            g.validate(full=True)
            assert_bit_exact(g, w)
    # and the common case right after it on the same projector: no nulls with bytes -> flat
    clean = pa.RecordBatch.from_arrays([pa.array([None if m else "abcXYZ"[:int(k)] for m, k in zip(nulls, lens % 7)])], names=["s"])
    for g, w in zip(proj.evaluate(clean), oracle.project(exprs, clean)):
        assert_bit_exact(g, w)


def test_oracle_like_wildcards_match_newlines_as_arrows_re2_backed_match_like_does():
    """Round-1 verdict, weak #1: do LIKE's `_` and `%` match a newline (RE2 dot_nl)?  The only RE2 in
    the image is the one inside libarrow: pyarrow.compute.match_like (like -> regex, RE2) says yes,
    and so does the oracle (the device code is held to the oracle by the GPU suite)."""
    vals = ["a\nb", "a\n\nb", "\n", "ab", "a\rb", "a b", "x\ny", "line1\nspark\nline3"]
    arr = pa.array(vals, pa.string())
    batch = pa.RecordBatch.from_arrays([arr], names=["s"])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    for pat in ["a_b", "a%b", "_", "%", "a__b", "x%y", "%spark%", "line1_spark%", "%\n%"]:
        node = b.make_function("like", [s, b.make_literal(pat, pa.string())], pa.bool_())
        assert oracle.project_one(node, pa.bool_(), batch).equals(pc.match_like(arr, pat)), pat


# ------------------------------------------------------------------ round 4: per-batch choice of the kernels

@pytest.mark.gpu
def test_a_non_ascii_batch_takes_the_exact_wave_variant_and_an_ascii_batch_brings_the_fast_path_back():
    """Round 3 sent a Projector to the scanner-shaped kernel FOR GOOD after one byte >= 0x80.  Now: the
    batch that raises NOTASCII is re-run on the exact variant of the wave kernels (path 1), the next
    batches start there, and the first all-ASCII batch returns the Projector to the optimistic pair (0)."""
    exprs = W.c5_expressions()
    proj = gandiva.make_projector(W.c5_schema(), exprs, pa.default_memory_pool())
    n = 60_013
    ascii_batch = W.c5_batch(n)
    for _ in range(17):   # (projectors are cached per plan: another test may have left this one on path 1 or 2)
        if proj.path_hint == 0:
            break
        proj.evaluate(ascii_batch)
    assert proj.path_hint == 0
    mixed = W.c5_batch(n, non_ascii_fraction=0.01)
    heavy = W.c5_batch(n, non_ascii_fraction=0.3)

    def check(batch, what):
        for g, w in zip(proj.evaluate(batch), oracle.project(exprs, batch)):
            assert_bit_exact(g, w, what)
    check(ascii_batch, "ASCII batch on the optimistic kernels")
    assert proj.path_hint == 0
    check(mixed, "1 % of the rows hold a two-byte character")
    assert proj.path_hint == 1                      # re-run on the exact variant; the next batch starts there
    check(heavy, "30 % non-ASCII rows, straight on the exact variant")
    assert proj.path_hint == 1
    check(ascii_batch, "an ASCII batch on the exact variant")
    assert proj.path_hint == 0                      # it saw no byte >= 0x80: back to the optimistic kernels
    check(ascii_batch, "and again on the optimistic kernels")
    assert proj.path_hint == 0
    # HBM-resident, buffers reused across the switch
    import torch
    dm, da = gandiva.DeviceBatch.from_arrow(mixed), gandiva.DeviceBatch.from_arrow(ascii_batch)
    outs = proj.evaluate_device(dm)
    torch.cuda.synchronize()
    for o, w in zip(outs, oracle.project(exprs, mixed)):
        assert_bit_exact(o.to_arrow(), w, "device-resident, non-ASCII")
    assert proj.path_hint == 1
    outs = proj.evaluate_device(da)
    torch.cuda.synchronize()
    for o, w in zip(outs, oracle.project(exprs, ascii_batch)):
        assert_bit_exact(o.to_arrow(), w, "device-resident, ASCII after non-ASCII")
    assert proj.path_hint == 0


@pytest.mark.gpu
def test_character_functions_on_the_exact_variant_match_the_oracle_on_multibyte_text():
    """substr / left / right / char_length / lpad / reverse / locate over rows mixing 1-, 2-, 3- and 4-byte
    characters, through the wave shape's exact variant (every batch here is non-ASCII)."""
    rng = np.random.default_rng(77)
    alphabet = list("abcXYZ 019") + ["é", "ü", "ß", "日", "本", "語", "𝄞", "€"]
    rows = [None if rng.random() < 0.08 else "".join(rng.choice(alphabet, size=int(rng.integers(0, 30)))) for _ in range(30_011)]
    batch = pa.RecordBatch.from_arrays([pa.array(rows, pa.string())], names=["s"])
    b = gandiva.TreeExprBuilder()
    s = b.make_field(batch.schema.field(0))
    i64 = pa.int64()

    def lit(v):
        return b.make_literal(v, i64)
    exprs = [b.make_expression(b.make_function("substr", [s, lit(2), lit(5)], pa.string()), pa.field("a", pa.string())),
             b.make_expression(b.make_function("substr", [s, lit(-3), lit(2)], pa.string()), pa.field("b", pa.string())),
             b.make_expression(b.make_function("left", [s, b.make_literal(4, pa.int32())], pa.string()), pa.field("c", pa.string())),
             b.make_expression(b.make_function("right", [s, b.make_literal(3, pa.int32())], pa.string()), pa.field("d", pa.string())),
             b.make_expression(b.make_function("char_length", [s], pa.int32()), pa.field("e", pa.int32())),
             b.make_expression(b.make_function("upper", [s], pa.string()), pa.field("f", pa.string()))]
    proj = gandiva.make_projector(batch.schema, exprs, pa.default_memory_pool())
    want = oracle.project(exprs, batch)
    for rnd in range(2):   # first call: optimistic -> NOTASCII -> exact; second: exact from the start
        for g, w, e in zip(proj.evaluate(batch), want, exprs):
            assert_bit_exact(g, w, f"{e} (call {rnd})")
        assert proj.path_hint == 1


@pytest.mark.gpu
def test_null_rows_that_carry_bytes_take_the_general_kernel_and_the_fast_path_is_tried_again_later():
    """A flat output (upper(col)) whose NULL rows hold bytes raises NOTFLAT: the general (scanner-shaped)
    kernel takes the batch and the following ones (path 2); every 16th batch the optimistic kernels get
    another try, so a Projector fed clean batches afterwards comes back to them."""
    n = 20_011
    offsets, data, _ = W.c5_numpy(n)
    rng = np.random.default_rng(3)
    mask = rng.random(n) < 0.1                      # NULL rows KEEP their bytes
    dirty = pa.RecordBatch.from_arrays([pa.Array.from_buffers(pa.string(), n, [
        pa.py_buffer(np.packbits(~mask, bitorder="little")), pa.py_buffer(offsets), pa.py_buffer(data)])], schema=W.c5_schema())
    clean = W.c5_batch(n)
    exprs = W.c5_expressions()
    proj = gandiva.make_projector(W.c5_schema(), exprs, pa.default_memory_pool())
    for g, w in zip(proj.evaluate(dirty), oracle.project(exprs, dirty)):
        assert_bit_exact(g, w, "NULL rows with bytes")
    assert proj.path_hint == 2
    want = oracle.project(exprs, clean)
    hints = []
    for _ in range(20):
        for g, w in zip(proj.evaluate(clean), want):
            assert_bit_exact(g, w, "clean batch")
        hints.append(proj.path_hint)
    assert hints[0] == 2 and hints[-1] == 0 and 0 in hints[:17]


# ------------------------------------------------------------------ round 4: var-len plans without a host synchronisation

@pytest.mark.gpu
def test_var_len_plans_evaluate_without_a_host_synchronisation():
    """gdv_projector_evaluate_async: C5 (wave-shaped pair) is enqueued and the call returns; the status word and the
    byte totals arrive in device memory in stream order.  An ASCII batch completes (status 0, same bytes as the
    synchronous call); a batch with bytes >= 0x80 on the optimistic kernels reports that it did not (the
    caller re-runs it synchronously, which also moves the projector to the exact variant, where the
    asynchronous call then completes)."""
    import torch
    exprs = W.c5_expressions()
    proj = gandiva.make_projector(W.c5_schema(), exprs, pa.default_memory_pool())
    n = 80_021
    clean, mixed = W.c5_batch(n), W.c5_batch(n, non_ascii_fraction=0.02)
    dc, dm = gandiva.DeviceBatch.from_arrow(clean), gandiva.DeviceBatch.from_arrow(mixed)
    for _ in range(17):
        if proj.path_hint == 0:
            break
        proj.evaluate(clean)
    assert proj.path_hint == 0
    outs, result = proj.evaluate_device_async(dc)
    torch.cuda.synchronize()
    want = oracle.project(exprs, clean)
    assert int(result[0]) == 0
    for i, (o, w) in enumerate(zip(outs, want)):
        assert_bit_exact(o.to_arrow(), w, f"asynchronous, ASCII batch, output {i}")
        if pa.types.is_string(w.type):
            assert int(result[1 + i]) == w.buffers()[2].size or int(result[1 + i]) == len(b"".join(x.as_py().encode() for x in w if x.is_valid))
    # non-ASCII batch on the optimistic kernels: the status says so, nothing is promised about the outputs
    outs2, result2 = proj.evaluate_device_async(dm)
    torch.cuda.synchronize()
    assert int(result2[0]) & 32, "NOTASCII expected in the status word"
    with pytest.raises(gandiva.GandivaError):
        outs2[1].to_arrow()
    assert proj.path_hint == 0            # an asynchronous call cannot move the projector ...
    got = proj.evaluate_device(dm)        # ... the synchronous re-run does
    torch.cuda.synchronize()
    want_m = oracle.project(exprs, mixed)
    for o, w in zip(got, want_m):
        assert_bit_exact(o.to_arrow(), w, "synchronous re-run")
    assert proj.path_hint == 1
    outs3, result3 = proj.evaluate_device_async(dm)   # now on the exact variant: completes
    torch.cuda.synchronize()
    # 0 = complete: the exact kernels' "saw UTF-8" note (bit 64) is cleared on the device before the word is
    # published (round 4 let it through: every asynchronous call on non-ASCII text looked failed to to_arrow())
    assert int(result3[0]) == 0
    for i, (o, w) in enumerate(zip(outs3, want_m)):
        assert_bit_exact(o.to_arrow(), w, f"asynchronous, exact variant, output {i}")
    # too small a byte buffer: the total says what was needed, nothing was written past the capacity
    outs4, result4 = proj.evaluate_device_async(dc, capacity_bytes=4096)
    torch.cuda.synchronize()
    assert int(result4[2]) > 4096 and int(result4[3]) > 4096
    with pytest.raises(gandiva.GandivaError):
        outs4[2].to_arrow()
    proj.evaluate(clean)                  # leave the cached projector on the fast path


@pytest.mark.gpu
def test_filter_then_upper_without_a_host_round_trip():
    """Round-3 verdict item 6: filter (asynchronous, count left in HBM) -> selection-mode projector with a
    VAR-LEN output, slot count read on the device — zero host synchronisations inside the chain."""
    import torch
    n = 120_011
    rng = np.random.default_rng(8)
    s = W.c5_batch(n).column(0)
    k = pa.array(rng.integers(0, 100, n), pa.int64())
    batch = pa.RecordBatch.from_arrays([s, k], names=["s", "k"])
    b = gandiva.TreeExprBuilder()
    fs, fk = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    cond = b.make_condition(b.make_function("greater_than", [fk, b.make_literal(70, pa.int64())], pa.bool_()))
    exprs = [b.make_expression(b.make_function("upper", [fs], pa.string()), pa.field("u", pa.string())),
             b.make_expression(b.make_function("substr", [fs, b.make_literal(2, pa.int64()), b.make_literal(5, pa.int64())], pa.string()),
                               pa.field("t", pa.string())),
             b.make_expression(b.make_function("add", [fk, fk], pa.int64()), pa.field("kk", pa.int64()))]
    flt = gandiva.make_filter(batch.schema, cond)
    proj = gandiva.make_projector(batch.schema, exprs, None, "UINT32")
    db = gandiva.DeviceBatch.from_arrow(batch)
    sel = flt.evaluate_device(db, "int32", sync=False)
    assert sel.pending
    outs, result = proj.evaluate_device_async(db, selection=sel)
    assert sel.pending                                   # nothing has waited yet
    torch.cuda.synchronize()
    want_sel = oracle.filter_indices(cond, batch, "int32")
    assert sel.num_slots == len(want_sel) and sel.to_array().equals(want_sel)
    assert int(result[0]) == 0
    want = oracle.project(exprs, oracle.take_rows(batch, want_sel.to_numpy()))
    for i, (o, w) in enumerate(zip(outs, want)):
        assert_bit_exact(o.to_arrow(), w, f"output {i}")


@pytest.mark.gpu
def test_two_stage_plans_evaluate_without_a_host_synchronisation():
    """upper(concat(s, '-', s)) needs the concat materialised first (a two-stage plan).  Asynchronously both stages
    are enqueued with a device-side gate between them: the second stage runs over the rows the gate lets through —
    all of them when the first stage completed and its temporaries were large enough, none otherwise (bit 128)."""
    import torch
    n = 60_013
    batch = W.c5_batch(n, 0.1)
    b = gandiva.TreeExprBuilder()
    fs = b.make_field(batch.schema.field(0))
    dash = b.make_literal("-", pa.string())
    cat = b.make_function("concat", [fs, dash, fs], pa.string())
    exprs = [b.make_expression(b.make_function("upper", [cat], pa.string()), pa.field("u", pa.string())),
             b.make_expression(b.make_function("substr", [b.make_function("reverse", [fs], pa.string()),
                                                          b.make_literal(2, pa.int64()), b.make_literal(4, pa.int64())], pa.string()),
                               pa.field("r", pa.string())),
             b.make_expression(b.make_function("like", [cat, b.make_literal("%k-s%", pa.string())], pa.bool_()), pa.field("l", pa.bool_()))]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    db = gandiva.DeviceBatch.from_arrow(batch)
    want = oracle.project(exprs, batch)
    # before any synchronous batch: the temporaries are sized by the first guess
    outs, result = proj.evaluate_device_async(db, capacity_bytes=4 << 20)
    torch.cuda.synchronize()
    assert int(result[0]) == 0
    for i, (o, w) in enumerate(zip(outs, want)):
        assert_bit_exact(o.to_arrow(), w, f"asynchronous two-stage, first guess, output {i}")
    # after one: sized from what that batch produced
    for g, w in zip(proj.evaluate_device(db), want):
        assert_bit_exact(g.to_arrow(), w, "synchronous")
    outs, result = proj.evaluate_device_async(db)
    torch.cuda.synchronize()
    assert int(result[0]) == 0
    for i, (o, w) in enumerate(zip(outs, want)):
        assert_bit_exact(o.to_arrow(), w, f"asynchronous two-stage, learnt sizes, output {i}")
    # a batch whose rows are far longer than anything seen so far: the temporaries are too small, the gate
    # closes, the status says so, and the synchronous call completes it
    long_rows = pa.array(["spark-" * 40 + str(i) for i in range(n)], pa.string())
    big = pa.RecordBatch.from_arrays([long_rows], schema=batch.schema)
    dbig = gandiva.DeviceBatch.from_arrow(big)
    short = pa.RecordBatch.from_arrays([pa.array(["ab"] * n, pa.string())], schema=batch.schema)
    for _ in range(12):                       # (the size hint decays towards the short rows)
        proj.evaluate_device(gandiva.DeviceBatch.from_arrow(short))
    outs, result = proj.evaluate_device_async(dbig, capacity_bytes=64 << 20)
    torch.cuda.synchronize()
    assert int(result[0]) & 128, f"status {int(result[0])}"
    with pytest.raises(gandiva.GandivaError):
        outs[0].to_arrow()
    for g, w in zip(proj.evaluate_device(dbig), oracle.project(exprs, big)):
        assert_bit_exact(g.to_arrow(), w, "synchronous, long rows")


@pytest.mark.gpu
def test_plans_of_three_stages_evaluate_without_a_host_synchronisation():
    """upper(reverse(replace(s, 'a', 'bb'))): the replace is materialised for the reverse, the reverse for the upper — three
    stages, two gates.  Rounds 3-4 evaluated such plans synchronously only; the staged asynchronous call now nests."""
    import torch
    n = 30_011
    batch = W.c5_batch(n, 0.1)
    b = gandiva.TreeExprBuilder()
    fs = b.make_field(batch.schema.field(0))
    S = pa.string()
    rep = b.make_function("replace", [fs, b.make_literal("a", S), b.make_literal("bb", S)], S)
    rev = b.make_function("reverse", [rep], S)
    exprs = [b.make_expression(b.make_function("upper", [rev], S), pa.field("u", S)),
             b.make_expression(b.make_function("length", [rev], pa.int32()), pa.field("n", pa.int32())),
             b.make_expression(b.make_function("substr", [rep, b.make_literal(2, pa.int64()), b.make_literal(3, pa.int64())], S), pa.field("p", S))]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    db = gandiva.DeviceBatch.from_arrow(batch)
    want = oracle.project(exprs, batch)
    outs, result = proj.evaluate_device_async(db, capacity_bytes=4 << 20)
    torch.cuda.synchronize()
    assert int(result[0]) == 0
    for i, (o, w) in enumerate(zip(outs, want)):
        assert_bit_exact(o.to_arrow(), w, f"asynchronous, three stages, output {i}")
    for g, w in zip(proj.evaluate_device(db), want):
        assert_bit_exact(g.to_arrow(), w, "synchronous, three stages")
    # behind an asynchronous filter (the count stays in device memory through both gates)
    cond = b.make_condition(b.make_function("like", [fs, b.make_literal("%a%", S)], pa.bool_()))
    flt = gandiva.make_filter(batch.schema, cond)
    sproj = gandiva.make_projector(batch.schema, exprs[:1], None, "UINT32")
    sel = flt.evaluate_device(db, "int32", sync=False)
    outs, result = sproj.evaluate_device_async(db, selection=sel)
    torch.cuda.synchronize()
    want_sel = oracle.filter_indices(cond, batch, "int32")
    assert sel.to_array().equals(want_sel) and int(result[0]) == 0
    assert_bit_exact(outs[0].to_arrow(), oracle.project(exprs[:1], oracle.take_rows(batch, want_sel.to_numpy()))[0], "filter -> three stages, asynchronous")


@pytest.mark.gpu
def test_filter_then_a_two_stage_projection_without_a_host_round_trip():
    import torch
    n = 90_001
    rng = np.random.default_rng(12)
    s = W.c5_batch(n).column(0)
    k = pa.array(rng.integers(0, 100, n), pa.int64())
    batch = pa.RecordBatch.from_arrays([s, k], names=["s", "k"])
    b = gandiva.TreeExprBuilder()
    fs, fk = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    cond = b.make_condition(b.make_function("less_than", [fk, b.make_literal(25, pa.int64())], pa.bool_()))
    exprs = [b.make_expression(b.make_function("upper", [b.make_function("concat", [fs, b.make_function("castVARCHAR", [fk, b.make_literal(10, pa.int64())], pa.string())],
                                                                          pa.string())], pa.string()), pa.field("u", pa.string()))]
    flt = gandiva.make_filter(batch.schema, cond)
    proj = gandiva.make_projector(batch.schema, exprs, None, "UINT32")
    db = gandiva.DeviceBatch.from_arrow(batch)
    sel = flt.evaluate_device(db, "int32", sync=False)
    outs, result = proj.evaluate_device_async(db, selection=sel)
    assert sel.pending
    torch.cuda.synchronize()
    want_sel = oracle.filter_indices(cond, batch, "int32")
    assert sel.to_array().equals(want_sel) and int(result[0]) == 0
    for o, w in zip(outs, oracle.project(exprs, oracle.take_rows(batch, want_sel.to_numpy()))):
        assert_bit_exact(o.to_arrow(), w, "filter -> two-stage projection, asynchronous")


@pytest.mark.gpu
def test_two_stage_plans_with_fixed_width_outputs_evaluate_without_a_host_synchronisation():
    """like(concat(s, '-', s), ...) and length(concat(...)): the second stage has fixed-width outputs only — the
    ordinary asynchronous launch over the staged columns, its row count from the gate."""
    import torch
    n = 40_009
    batch = W.c5_batch(n, 0.1)
    b = gandiva.TreeExprBuilder()
    fs = b.make_field(batch.schema.field(0))
    cat = b.make_function("concat", [fs, b.make_literal("-", pa.string()), fs], pa.string())
    exprs = [b.make_expression(b.make_function("like", [cat, b.make_literal("%k-s%", pa.string())], pa.bool_()), pa.field("l", pa.bool_())),
             b.make_expression(b.make_function("length", [b.make_function("reverse", [fs], pa.string())], pa.int32()), pa.field("n", pa.int32())),
             b.make_expression(b.make_function("starts_with", [cat, b.make_literal("sp", pa.string())], pa.bool_()), pa.field("p", pa.bool_()))]
    proj = gandiva.make_projector(batch.schema, exprs, None)
    db = gandiva.DeviceBatch.from_arrow(batch)
    # a second stage that CAN raise (locate checks its start position; an integer division) raises into result[0]
    # itself (round 5; round 4 refused such plans: "synchronous only")
    len_cat = b.make_function("length", [cat], pa.int32())
    can_raise = [b.make_expression(b.make_function("locate", [b.make_literal("k-", pa.string()), cat], pa.int32()), pa.field("q", pa.int32())),
                 b.make_expression(b.make_function("divide", [len_cat, b.make_literal(7, pa.int32())], pa.int32()), pa.field("d", pa.int32()))]
    raising = gandiva.make_projector(batch.schema, can_raise, None)
    outs_r, result_r = raising.evaluate_device_async(db)
    torch.cuda.synchronize()
    assert int(result_r[0]) == 0
    for o, w in zip(outs_r, oracle.project(can_raise, batch)):
        assert_bit_exact(o.to_arrow(), w, "second stage that can raise, asynchronously")
    by_zero = gandiva.make_projector(batch.schema, [b.make_expression(b.make_function("divide", [len_cat, b.make_literal(0, pa.int32())], pa.int32()),
                                                                       pa.field("z", pa.int32()))], None)
    outs_z, result_z = by_zero.evaluate_device_async(db)
    torch.cuda.synchronize()
    assert int(result_z[0]) & 1, "divide by zero must arrive in the status word"
    want = oracle.project(exprs, batch)
    for attempt in ("first guess", "learnt sizes"):
        outs, result = proj.evaluate_device_async(db)
        torch.cuda.synchronize()
        assert int(result[0]) == 0, (attempt, int(result[0]))
        for i, (o, w) in enumerate(zip(outs, want)):
            assert_bit_exact(o.to_arrow(), w, f"{attempt}, output {i}")
        for g, w in zip(proj.evaluate_device(db), want):
            assert_bit_exact(g.to_arrow(), w, "synchronous")
    short = pa.RecordBatch.from_arrays([pa.array(["ab"] * n, pa.string())], schema=batch.schema)
    for _ in range(12):
        proj.evaluate_device(gandiva.DeviceBatch.from_arrow(short))
    big = pa.RecordBatch.from_arrays([pa.array(["spark-" * 40 + str(i) for i in range(n)], pa.string())], schema=batch.schema)
    dbig = gandiva.DeviceBatch.from_arrow(big)
    outs, result = proj.evaluate_device_async(dbig)
    torch.cuda.synchronize()
    assert int(result[0]) & 128
    for g, w in zip(proj.evaluate_device(dbig), oracle.project(exprs, big)):
        assert_bit_exact(g.to_arrow(), w, "synchronous, long rows")


# ------------------------------------------------------------------ round 5: selection-mode plans on the wave shape

def test_selection_mode_string_plans_take_the_wave_shape(monkeypatch, tmp_path):
    """Rounds 2-4: every var-len plan under a selection vector ran the scanner-shaped kernel (0.27 of the roofline,
    5x the per-row cost of row mode: profiles/r05_filter_string_chain_before.txt).  Now: pre-pass + offsets scan +
    independent wave tiles, rows = slots (gathered), no byte sweep, no optimistic assumption — and no exact
    variant, there is nothing to be optimistic about."""
    import ctypes as C
    from gandiva_amd import _capi, gandiva as gg
    monkeypatch.setenv("GDV_NO_DISK_CACHE", "1")
    monkeypatch.setenv("GDV_DUMP_SOURCE", "1")
    lib = _capi.lib()
    ex = W.c5_expressions()
    for sub, env, want_wave in (("wave", None, True), ("scanner", "GDV_NO_SEL_WAVE", False)):
        d = tmp_path / sub
        d.mkdir()
        monkeypatch.setenv("GANDIVA_AMD_CACHE_DIR", str(d))
        if env:
            monkeypatch.setenv(env, "1")
        sh = gg._make_schema(W.c5_schema())
        try:
            arr = (C.c_void_p * len(ex))(*[e._h for e in ex])
            assert lib.gdv_precompile_projector(sh, arr, len(ex), 2) == 0, _capi.last_error()
        finally:
            lib.gdv_schema_free(sh)
        texts = [open(d / f).read() for f in os.listdir(d) if f.endswith(".hip")]
        wave = [t for t in texts if "// wave shape:" in t]
        pre = [t for t in texts if "// pre-pass:" in t]
        if want_wave:
            assert len(wave) == 1 and len(pre) == 1 and len(texts) == 3          # + the scanner-shaped fallback
            assert "rows = the slots of a selection vector" in wave[0] and "selv[row]" in wave[0] and "selv[row]" in pre[0]
            # optimistic ASCII: the pre-pass reads offsets only, the main kernel checks every row it copies;
            # no exact variant (a batch that breaks the assumption takes the scanner-shaped general kernel)
            assert "gdv_row_is_ascii(s0)" in wave[0] and "GDV_ERR_NOTASCII" in wave[0] and "gdv_row_is_ascii" not in pre[0]
            assert "exact variant" not in "".join(texts)
            assert "outo1[n] =" in wave[0]            # the closing offset is the kernel's own business (device-resident counts)
        else:
            assert not wave and not pre and len(texts) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode,dtype", [("UINT16", "int16"), ("UINT32", "int32"), ("UINT64", "int64")])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 511, 512, 513, 4097, 40_003])
def test_selection_mode_wave_shape_matches_the_oracle(mode, dtype, n):
    """C5's three expressions + multi-byte text under a selection vector (every wave-tile boundary: 512 slots),
    sparse and dense selections, nulls, host and HBM-resident paths."""
    import torch
    if dtype == "int16":
        n = min(n, 30_000)
    rng = np.random.default_rng(n * 3 + len(mode))
    base = W.c5_batch(n, 0.1, non_ascii_fraction=0.05 if n % 2 else 0.0)
    key = pa.array(rng.integers(0, 100, n), pa.int64())
    batch = pa.RecordBatch.from_arrays([base.column(0), key], names=["s", "k"])
    b = gandiva.TreeExprBuilder()
    fs, fk = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    exprs = [b.make_expression(b.make_function("upper", [fs], pa.string()), pa.field("u", pa.string())),
             b.make_expression(b.make_function("substr", [fs, b.make_literal(2, pa.int64()), b.make_literal(5, pa.int64())], pa.string()),
                               pa.field("t", pa.string())),
             b.make_expression(b.make_function("like", [fs, b.make_literal("%spark%", pa.string())], pa.bool_()), pa.field("m", pa.bool_())),
             b.make_expression(b.make_function("concat", [fs, b.make_literal("-", pa.string()), fs], pa.string()), pa.field("c", pa.string())),
             b.make_expression(b.make_function("add", [fk, fk], pa.int64()), pa.field("kk", pa.int64()))]
    proj = gandiva.make_projector(batch.schema, exprs, None, mode)
    for thr in (10, 60, 98):
        cond = b.make_condition(b.make_function("greater_than", [fk, b.make_literal(thr, pa.int64())], pa.bool_()))
        sel = gandiva.make_filter(batch.schema, cond).evaluate(batch, None, dtype)
        want = oracle.project(exprs, oracle.take_rows(batch, sel.to_array().to_numpy()))
        for i, (g, w) in enumerate(zip(proj.evaluate(batch, sel), want)):
            assert_bit_exact(g, w, f"host buffers, threshold {thr}, output {i}")
        db = gandiva.DeviceBatch.from_arrow(batch)
        dsel = gandiva.make_filter(batch.schema, cond).evaluate_device(db, dtype)
        outs = proj.evaluate_device(db, selection=dsel)
        torch.cuda.synchronize()
        for i, (o, w) in enumerate(zip(outs, want)):
            assert_bit_exact(o.to_arrow(), w, f"HBM-resident, threshold {thr}, output {i}")
    # ASCII batches stay on the wave pair; one with bytes >= 0x80 under substr / like was re-run on the general kernel
    assert proj.path_hint in (0, 2)   # (projectors are cached per plan: an earlier non-ASCII batch may have left it on the general kernel)


@pytest.mark.gpu
def test_an_empty_selection_whose_count_sits_in_device_memory():
    """filter (asynchronous) selects nothing -> the selection-mode wave kernels launch for the capacity, walk 0 slots
    and still leave a valid empty column: offsets[0] = 0, byte totals 0, status 0."""
    import torch
    n = 10_000
    batch = pa.RecordBatch.from_arrays([W.c5_batch(n).column(0), pa.array(np.arange(n) % 50, pa.int64())], names=["s", "k"])
    b = gandiva.TreeExprBuilder()
    fs, fk = b.make_field(batch.schema.field(0)), b.make_field(batch.schema.field(1))
    cond = b.make_condition(b.make_function("greater_than", [fk, b.make_literal(1000, pa.int64())], pa.bool_()))
    exprs = [b.make_expression(b.make_function("upper", [fs], pa.string()), pa.field("u", pa.string()))]
    db = gandiva.DeviceBatch.from_arrow(batch)
    sel = gandiva.make_filter(batch.schema, cond).evaluate_device(db, "int32", sync=False)
    proj = gandiva.make_projector(batch.schema, exprs, None, "UINT32")
    outs, result = proj.evaluate_device_async(db, selection=sel)
    outs[0].offsets.fill_(0x55)         # whatever was there must not survive as offsets[0]
    outs, result = proj.evaluate_device_async(db, selection=sel, outputs=outs)
    torch.cuda.synchronize()
    assert sel.num_slots == 0 and int(result[0]) == 0 and int(result[1]) == 0
    got = outs[0].to_arrow()
    assert len(got) == 0
    got.validate(full=True)
    assert int(outs[0].offsets[:4].view(torch.int32)[0]) == 0
