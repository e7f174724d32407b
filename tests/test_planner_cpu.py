"""Planner properties that need no GPU: the generated kernels are compiled for gfx950 by hipRTC
right here (cross-compile), and their source is dumped next to the code object."""
import ctypes as C
import os
import re

import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import _capi, gandiva as gg, workloads as W


def _precompile(monkeypatch, tmp_path, schema, exprs=None, cond=None):
    monkeypatch.setenv("GDV_NO_DISK_CACHE", "1")
    monkeypatch.setenv("GDV_DUMP_SOURCE", "1")
    monkeypatch.setenv("GANDIVA_AMD_CACHE_DIR", str(tmp_path))
    lib = _capi.lib()
    sh = gg._make_schema(schema)
    try:
        if cond is not None:
            rc = lib.gdv_precompile_filter(sh, cond._h)
        else:
            arr = (C.c_void_p * len(exprs))(*[e._h for e in exprs])
            rc = lib.gdv_precompile_projector(sh, arr, len(exprs), 0)
        assert rc == 0, _capi.last_error()
    finally:
        lib.gdv_schema_free(sh)
    return sorted(f for f in os.listdir(tmp_path) if f.endswith(".hip"))


def _precompile_fp(monkeypatch, tmp_path, schema, cond, exprs, mode):
    """the fused filter-project plan: every kernel it holds is compiled (the windowed shape AND its direct variant)"""
    tmp_path.mkdir(parents=True, exist_ok=True)
    monkeypatch.setenv("GDV_NO_DISK_CACHE", "1")
    monkeypatch.setenv("GDV_DUMP_SOURCE", "1")
    monkeypatch.setenv("GANDIVA_AMD_CACHE_DIR", str(tmp_path))
    lib = _capi.lib()
    sh = gg._make_schema(schema)
    try:
        arr = (C.c_void_p * len(exprs))(*[e._h for e in exprs])
        rc = lib.gdv_precompile_filter_project(sh, cond._h, arr, len(exprs), mode)
        assert rc == 0, _capi.last_error()
    finally:
        lib.gdv_schema_free(sh)
    return sorted(f for f in os.listdir(tmp_path) if f.endswith(".hip"))


def test_plans_that_differ_only_in_constants_share_one_kernel(monkeypatch, tmp_path):
    """Fixed-width literals, IN values and LIKE needles are kernel arguments / constant-block
    bytes: two filters with different constants must generate the SAME source (same kernel name);
    a different needle LENGTH is a different shape and may not."""
    b = gandiva.TreeExprBuilder()
    sch = pa.schema([pa.field("a", pa.int64()), pa.field("x", pa.float64()), pa.field("s", pa.string())])
    a, x, s = (b.make_field(sch.field(i)) for i in range(3))

    def cond(k, f, pat, inl):
        return b.make_condition(b.make_and([
            b.make_function("greater_than", [a, b.make_literal(k, pa.int64())], pa.bool_()),
            b.make_or([b.make_function("less_than", [x, b.make_literal(f, pa.float64())], pa.bool_()),
                       b.make_function("like", [s, b.make_literal(pat, pa.string())], pa.bool_()),
                       b.make_in_expression(a, inl, pa.int64())])]))
    _precompile(monkeypatch, tmp_path, sch, cond=cond(499, 0.25, "%spark%", [1, 5, 9]))
    files = _precompile(monkeypatch, tmp_path, sch, cond=cond(500, -1.5, "%flink%", [500, 2, 777]))
    assert len(files) == 1, files
    files = _precompile(monkeypatch, tmp_path, sch, cond=cond(500, -1.5, "%sparks%", [500, 2, 777]))
    assert len(files) == 2, files
    text = open(tmp_path / files[0]).read()
    assert "499" not in re.sub(r"^// @expr_.*$", "", text, flags=re.M)   # constants only in the header comment


def test_string_projection_builds_wave_prepass_and_general_kernels(monkeypatch, tmp_path):
    """C5 (like / substr / upper): the lengths of its var-len outputs follow from the offsets once
    the bytes are assumed ASCII, so the plan takes the wave shape (round 3): a pre-pass kernel (byte
    totals per wave tile, no byte read), the main kernel of independent wave tiles, and the
    scanner-shaped general kernel a batch that breaks the assumption is re-run on.  All three must
    compile for gfx950 at build time."""
    monkeypatch.delenv("GDV_PRECOMPILE_SKIP_GENERAL", raising=False)
    files = _precompile(monkeypatch, tmp_path, W.c5_schema(), exprs=W.c5_expressions())
    assert len(files) == 5, files
    texts = [open(tmp_path / f).read() for f in files]
    main = [t for t in texts if "// wave shape:" in t]
    pre = [t for t in texts if "// pre-pass:" in t and "exact variant" not in t]
    general = [t for t in texts if "gdv_scanner<GDV_NG>" in t]
    # round 4: the EXACT variant of the pair — what a batch with bytes >= 0x80 is re-run on: ASCII is a
    # fact about each row there (continuation bitmap from the sweep), in the pre-pass and the main kernel
    main_x = [t for t in texts if "// wave shape, exact variant" in t]
    pre_x = [t for t in texts if "// pre-pass:" in t and "exact variant" in t]
    assert len(main) == len(pre) == len(general) == len(main_x) == len(pre_x) == 1
    for t in (main_x[0], pre_x[0]):
        assert "gdv_with_lead(" in t and "gdv_cont_mask16(" in t and "GDV_ERR_NOTASCII" not in t
    assert "GDV_ERR_SAWUTF8" in main_x[0] and "gdv_with_lead(" not in main[0] and "| GDV_STR_ASCII" in main[0]
    assert "@expr_2 = string upper((string) s)" in main[0]           # DumpIR keeps the readable header
    assert "__syncthreads" not in main[0] and "gdv_lb_wait" not in main[0] and "gdv_sweep_store" in main[0]
    assert "GDV_ERR_NOTASCII" in main[0] and "A.mask[0 * seg_stride + wt]" in main[0]
    assert "A.counts[0 * seg_stride + wt]" in pre[0] and "sd0 + a" not in pre[0]   # the pre-pass reads no byte
    assert re.search(r"#define GDV_OPTFLAT (\d)", general[0]).group(1) == "0"


def test_plans_whose_lengths_need_bytes_get_a_byte_reading_prepass(monkeypatch, tmp_path):
    """rtrim / replace / an if over like(): the output length depends on the bytes.  Such plans take
    the wave shape too — their pre-pass reads the rows' bytes a first time (no sweep there: a
    '%needle%' inside it is the per-row search); GDV_WAVE_BYTEFREE_ONLY=1 and GDV_NO_WAVE_SHAPE=1
    restore the scanner shape, which selection-mode plans always take."""
    b = gandiva.TreeExprBuilder()
    sch = W.c5_schema()
    s = b.make_field(sch.field(0))
    exprs = [b.make_expression(b.make_function("rtrim", [s], pa.string()), pa.field("t", pa.string())),
             b.make_expression(b.make_if(b.make_function("like", [s, b.make_literal("%spark%", pa.string())], pa.bool_()),
                                         b.make_function("upper", [s], pa.string()), b.make_literal("-", pa.string()),
                                         pa.string()), pa.field("u", pa.string()))]
    files = _precompile(monkeypatch, tmp_path, sch, exprs=exprs)
    texts = [open(tmp_path / f).read() for f in files]
    texts = [t for t in texts if "exact variant" not in t]   # (round 4: + the exact variant of the wave pair)
    pre = [t for t in texts if "// pre-pass" in t]
    assert len(texts) == 3 and len(pre) == 1, files
    assert "rtrim_utf8" in pre[0] and "gdv_like_contains" in pre[0] and "gdv_range_any" not in pre[0]
    for env in ("GDV_WAVE_BYTEFREE_ONLY", "GDV_NO_WAVE_SHAPE"):
        d = tmp_path / env
        d.mkdir()
        monkeypatch.setenv(env, "1")
        files = _precompile(monkeypatch, d, sch, exprs=exprs)
        assert all("gdv_scanner<GDV_NG>" in open(d / f).read() for f in files), (env, files)
        monkeypatch.delenv(env)


def test_fixed_width_plans_have_one_variant_and_a_literal_free_body(monkeypatch, tmp_path):
    files = _precompile(monkeypatch, tmp_path, W.c4_schema(), exprs=W.c4_expressions())
    assert len(files) == 1
    text = open(tmp_path / files[0]).read()
    assert "A.lit[" in text and "gdv_make_int128(A.lit[" in text


def test_committed_pmc_files_belong_to_the_kernels_this_tree_generates(monkeypatch, tmp_path):
    """profiles/pmc_c2..c5.json carry the name of the kernel they were measured on; a kernel's name
    is a hash of its generated text AND of the device library.  If this fails, the tree has moved
    on since the counters were taken: re-run tools/gpu_evidence.sh (bench.py would print
    traffic = null for the same reason)."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    plans = {"c2": (W.c2_schema(), W.c2_expressions(), None), "c3": (W.c3_schema(), None, W.c3_condition()),
             "c4": (W.c4_schema(), W.c4_expressions(), None), "c5": (W.c5_schema(), W.c5_expressions(), None)}
    stale = []
    for w, (schema, exprs, cond) in plans.items():
        d = tmp_path / w
        d.mkdir()
        dumped = _precompile(monkeypatch, d, schema, exprs, cond)
        want = json.load(open(os.path.join(here, "..", "profiles", f"pmc_{w}.json")))["kernel"]
        if want + ".hip" not in dumped:
            stale.append(f"profiles/pmc_{w}.json was measured on {want}; this tree generates {dumped}")
        # ... and the latest round's rocprofv3 kernel stats of the same workload name a kernel this tree generates
        import glob
        stats = sorted(glob.glob(os.path.join(here, "..", "profiles", f"r[0-9][0-9]_{w}_kernel_stats.csv")))
        if stats:
            text = open(stats[-1]).read()
            if want not in text:
                stale.append(f"{os.path.basename(stats[-1])} was not taken on {want}, the kernel profiles/pmc_{w}.json names")
    if stale:
        msg = "; ".join(stale) + ": re-run tools/gpu_evidence.sh"
        # round 5: stale evidence FAILS.  Round 4 only warned here and shipped four orphaned counter files.
        # Work in progress opts out explicitly (GDV_EVIDENCE_PENDING=1) — the default run does not.
        if os.environ.get("GDV_EVIDENCE_PENDING") == "1":
            import warnings
            warnings.warn(msg)
        else:
            pytest.fail(msg)


def test_registry_aliases_share_their_kernels(monkeypatch, tmp_path):
    """modulo = mod, pow = power, position = locate: an alias resolves to the same device function, so
    the generated kernel (its name is a hash of the text without the `// @expr_` header) is the same
    one — and the oracle gives the same answers under both names."""
    from oracle import oracle
    b = gandiva.TreeExprBuilder()
    sch = pa.schema([pa.field("a", pa.int64()), pa.field("k", pa.int32()), pa.field("x", pa.float64()), pa.field("s", pa.string())])
    a, k, x, s = (b.make_field(sch.field(i)) for i in range(4))
    pairs = [("mod", "modulo", [a, k], pa.int32()), ("power", "pow", [x, x], pa.float64()),
             ("locate", "position", [b.make_literal("ar", pa.string()), s], pa.int32())]
    batch = pa.RecordBatch.from_arrays([pa.array([7, -7, 100, None], pa.int64()), pa.array([3, 3, 0, 5], pa.int32()),
                                        pa.array([2.0, 0.5, None, 3.0]), pa.array(["spark", "art", None, "bar ar"])], schema=sch)
    for name, alias, args, t in pairs:
        d1, d2 = tmp_path / name, tmp_path / alias
        d1.mkdir(); d2.mkdir()
        e1 = b.make_expression(b.make_function(name, args, t), pa.field("r", t))
        e2 = b.make_expression(b.make_function(alias, args, t), pa.field("r", t))
        assert _precompile(monkeypatch, d1, sch, [e1]) == _precompile(monkeypatch, d2, sch, [e2]), (name, alias)
        assert oracle.project([e1], batch)[0].equals(oracle.project([e2], batch)[0])


def test_replace_is_answered_by_the_sweep_only_where_matches_cannot_overlap(monkeypatch, tmp_path):
    """replace(col, from, to): a 'from' of 2..8 bytes that cannot overlap itself is counted in the byte
    sweep's match bitmap in BOTH wave-shaped kernels (pre-pass and main) and copied along it; a needle
    with a border ('aa'), of one byte or of more than eight keeps the per-row search.  Either way the
    plan stays wave-shaped: the scanner-shaped fallback lays out the same constant block (the hook reads
    the needle out of the replace table, it has no table of its own)."""
    b = gandiva.TreeExprBuilder()
    sch = pa.schema([pa.field("s", pa.string())])
    s = b.make_field(sch.field(0))

    def kernels(frm, sub):
        d = tmp_path / sub
        d.mkdir()
        e = b.make_expression(b.make_function("replace", [s, b.make_literal(frm, pa.string()), b.make_literal("flink", pa.string())],
                                              pa.string()), pa.field("r", pa.string()))
        files = _precompile(monkeypatch, d, sch, exprs=[e])
        texts = [open(d / f).read() for f in files]
        assert len(texts) == 3, files                                   # pre-pass + main + scanner-shaped fallback
        return ([t for t in texts if "// pre-pass:" in t][0], [t for t in texts if "// wave shape:" in t][0],
                [t for t in texts if "// pre-pass:" not in t and "// wave shape:" not in t][0])

    pre, main, general = kernels("spark", "a")
    for t in (pre, main):
        assert "gdv_replace_hits(" in t and "gdv_match8(" in t and "GDV_SUB_SPAN" in t
    assert "gdv_stage_copy_mirh(" in main and "gdv_lds_in" in main and "gdv_lds_in" not in pre
    assert "gdv_replace_hits(" not in general and "gdv_replace(" in general
    for k, frm in enumerate(["aa", "abab", "x", "a much longer needle"]):
        pre, main, general = kernels(frm, f"b{k}")
        for t in (pre, main, general):
            assert "gdv_replace_hits(" not in t and "gdv_replace(" in t, frm


def test_the_wave_tile_of_a_fixed_width_projection_follows_the_width_of_its_values(monkeypatch, tmp_path):
    """Round 6 (profiles/r06_tile_shape.txt): 16 sub-tiles of 64 rows per wave where 16 rows of loaded values fit 512 bytes of
    registers per lane and no element is wider than 8 bytes (C1, C2); decimal128 plans (C4) keep 4; predicate kernels 16."""
    def shape(d, w, cond=False):
        files = _precompile(monkeypatch, tmp_path / d, getattr(W, w + "_schema")(),
                            None if cond else getattr(W, w + "_expressions")(), W.c3_condition() if cond else None)
        text = open(os.path.join(tmp_path / d, files[0])).read()
        return int(re.search(r"#define GDV_U (\d+)", text).group(1)), int(re.search(r"#define GDV_WAVES (\d+)", text).group(1))
    for d in ("c1", "c2", "c4", "c3"):
        (tmp_path / d).mkdir()
    assert shape("c1", "c1") == (16, 4)
    assert shape("c2", "c2") == (16, 4)
    assert shape("c4", "c4") == (4, 4)
    assert shape("c3", "c3", cond=True) == (16, 4)
    # forty bytes of inputs per row: 16 sub-tiles would need 640 bytes per lane -> the rule of rounds 1-5
    b = gandiva.TreeExprBuilder()
    sch = pa.schema([pa.field(c, pa.float64()) for c in "pqrst"])
    f = [b.make_field(x) for x in sch]
    s = f[0]
    for x in f[1:]:
        s = b.make_function("add", [s, x], pa.float64())
    (tmp_path / "wide").mkdir()
    files = _precompile(monkeypatch, tmp_path / "wide", sch, [b.make_expression(s, pa.field("o", pa.float64()))])
    text = open(os.path.join(tmp_path / "wide", files[0])).read()
    assert int(re.search(r"#define GDV_U (\d+)", text).group(1)) == 4
