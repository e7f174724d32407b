"""Kernel identity (round 3): a kernel's name hashes its generated text and the device-library
items that text REACHES (gdv_libtag.cc) — not the whole header.  Editing or adding a function a
kernel never calls must leave its name, and every profile taken on it, alone."""
import ctypes as C
import os
import re

import pytest

from gandiva_amd import _capi, workloads as W
from test_planner_cpu import _precompile


def _lib_source():
    return _capi.lib().gdv_device_library_source().decode()


def _tag(text, lib_src=None):
    lib = _capi.lib()
    p = lib.gdv_kernel_library_tag(None if lib_src is None else lib_src.encode(), text.encode())
    s = C.cast(p, C.c_char_p).value.decode()
    lib.gdv_free_string(p)
    return s


def _items(text, lib_src=None):
    lib = _capi.lib()
    p = lib.gdv_kernel_library_items(None if lib_src is None else lib_src.encode(), text.encode())
    s = C.cast(p, C.c_char_p).value.decode()
    lib.gdv_free_string(p)
    return [l for l in s.split("\n") if l]


def _kernel_text(monkeypatch, tmp_path, which):
    if which == "c3":
        files = _precompile(monkeypatch, tmp_path, W.c3_schema(), cond=W.c3_condition())
    else:
        schema, exprs = {"c2": (W.c2_schema, W.c2_expressions), "c4": (W.c4_schema, W.c4_expressions),
                         "c5": (W.c5_schema, W.c5_expressions)}[which]
        files = _precompile(monkeypatch, tmp_path, schema(), exprs=exprs())
    texts = [open(tmp_path / f).read() for f in files]
    main = [t for t in texts if "// wave shape" in t]   # C5: pre-pass / main / general kernel
    return main[0] if main else texts[0]


def test_editing_a_string_function_renames_string_kernels_only(monkeypatch, tmp_path):
    src = _lib_source()
    c2 = _kernel_text(monkeypatch, tmp_path / "c2", "c2") if (tmp_path / "c2").mkdir() is None else None
    c5 = _kernel_text(monkeypatch, tmp_path / "c5", "c5") if (tmp_path / "c5").mkdir() is None else None
    # the float64 projection reaches no string function at all
    reached = _items(c2)
    assert "add_float64_float64" in reached      # round 5: the expansion of GDV_FLOAT_TYPES(GDV_FLOAT_ARITH), by name
    assert "add_int32_int32" not in reached and "less_than_int64_int64" not in reached
    assert not [r for r in reached if r.startswith("<base>") and not r.startswith("<base> #")], reached   # directives only: no instantiation line left in the base
    for name in ("substr_utf8_int64_int64", "upper_utf8", "gdv_like_contains", "gdv_scanner", "gdv_flat_copy"):
        assert name not in reached, name
    assert "substr_utf8_int64_int64" in _items(c5)
    # an edit inside substr: C5's kernel changes identity, C2's does not
    needle = "GDV_DEV gdv_str substr_utf8_int64_int64(gdv_str s, gdv_int64 from, gdv_int64 count) {"
    assert needle in src
    edited = src.replace(needle, needle + " count += 0;")
    assert _tag(c2, edited) == _tag(c2, src) == _tag(c2)
    assert _tag(c5, edited) != _tag(c5, src)
    # a brand-new function nobody calls changes nothing
    added = src + "\nGDV_DEV gdv_str initcap_utf8(gdv_str s) { return s; }\n"
    assert _tag(c2, added) == _tag(c2) and _tag(c5, added) == _tag(c5)
    # comments and blank lines are not code
    commented = src.replace(needle, "// a remark\n\n" + needle)
    assert _tag(c5, commented) == _tag(c5)


def test_a_new_macro_family_renames_nothing_and_an_edited_one_only_its_users(monkeypatch, tmp_path):
    """Round 4 lost its evidence to exactly this: `GDV_DATE_TRUNC(T)` + two instantiation lines were
    added late, the instantiations sat in the always-hashed base, and every kernel — the float64
    projection included — got a new name.  Families are expanded now and their functions attributed
    by name like hand-written ones."""
    src = _lib_source()
    texts = {}
    for w in ("c2", "c3", "c4", "c5"):
        (tmp_path / w).mkdir()
        texts[w] = _kernel_text(monkeypatch, tmp_path / w, w)
    family = ("\n#define GDV_NEW_FAMILY(T) \\\n  GDV_DEV gdv_##T brand_new_##T(gdv_##T a) { return a; } \\\n"
              "  GDV_DEV gdv_##T brand_newer_##T##_##T(gdv_##T a, gdv_##T b) { return add_##T##_##T(a, b); }\n"
              "GDV_NEW_FAMILY(float64)\nGDV_NUMERIC_TYPES(GDV_NEW_FAMILY)\n")
    anchor = "GDV_FLOAT_TYPES(GDV_FLOAT_ARITH)\n"
    assert anchor in src
    added = src.replace(anchor, anchor + family, 1)
    for w, t in texts.items():
        assert _tag(t, added) == _tag(t, src) == _tag(t), w
    assert "brand_new_float64" not in _items(texts["c2"], added)
    assert "brand_new_float64" in _items("x = brand_new_float64(y);", added)
    # a real edit of the float arithmetic family renames the float64 projection, and NOT the int64
    # filter, the decimal projection or the string kernels
    edited = src.replace("GDV_DEV gdv_##T add_##T##_##T(gdv_##T a, gdv_##T b) { return a + b; }",
                         "GDV_DEV gdv_##T add_##T##_##T(gdv_##T a, gdv_##T b) { return b + a; }", 1)
    assert edited != src
    assert _tag(texts["c2"], edited) != _tag(texts["c2"], src)
    for w in ("c3", "c4", "c5"):
        assert _tag(texts[w], edited) == _tag(texts[w], src), w
    # ... and an edit of the relational family renames the filter, not the projection
    edited = src.replace("GDV_DEV bool greater_than_##T##_##T(gdv_##T a, gdv_##T b) { return a > b; }",
                         "GDV_DEV bool greater_than_##T##_##T(gdv_##T a, gdv_##T b) { return b < a; }", 1)
    assert edited != src
    assert _tag(texts["c3"], edited) != _tag(texts["c3"], src)
    assert _tag(texts["c2"], edited) == _tag(texts["c2"], src)


def test_editing_arithmetic_or_core_items_renames_every_kernel(monkeypatch, tmp_path):
    src = _lib_source()
    (tmp_path / "c2").mkdir()
    c2 = _kernel_text(monkeypatch, tmp_path / "c2", "c2")
    edited = src.replace("#define GDV_FLOAT_ARITH(T)", "#define GDV_FLOAT_ARITH(T) /* */ ", 1)
    assert edited != src
    assert _tag(c2, edited) == _tag(c2, src)  # a comment is still not code
    edited = src.replace("GDV_DEV T gdv_ldnt(const T* p, gdv_int64 i) {", "GDV_DEV T gdv_ldnt(const T* p, gdv_int64 i) { (void)i;", 1)
    assert edited != src
    assert _tag(c2, edited) != _tag(c2, src)


def test_kernel_name_carries_the_reached_library_hash(monkeypatch, tmp_path):
    (tmp_path / "c2").mkdir()
    text = _kernel_text(monkeypatch, tmp_path / "c2", "c2")
    m = re.search(r"gdv_k_[0-9a-f]{16}", text)
    assert m
    # the text handed back contains the kernel's own name; the tag is taken on the text with the
    # placeholder, so only check stability: two plans of the same shape agree
    (tmp_path / "again").mkdir()
    again = _kernel_text(monkeypatch, tmp_path / "again", "c2")
    assert re.search(r"gdv_k_[0-9a-f]{16}", again).group(0) == m.group(0)
