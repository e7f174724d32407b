"""The oracle's string-tree semantics against an independent restatement in plain Python
(str / bytes / re), over the random string trees of tests/test_fuzz_trees.py: views (substr,
trims, ASCII upper/lower), LIKE (translated to a regular expression), bytewise comparisons,
starts/ends_with, IN, null tests, Kleene AND/OR, if/else (also over differently mapped
branches), lengths, MurmurHash3 (sklearn's x86_32, a pure-Python x64_128), concat and ||.
CPU only."""
import re

import numpy as np
import pyarrow as pa
import pytest
from sklearn.utils import murmurhash3_32

import test_fuzz_trees as F
from test_strings import _py_murmur3_x64_128_h1
from oracle import oracle

NULL = None


def _upper(v):
    return "".join(chr(ord(c) - 32) if "a" <= c <= "z" else c for c in v)


def _lower(v):
    return "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in v)


def _substr(v, start, count=0x7fffffff):
    if count <= 0 or not v:
        return ""
    n = len(v)
    s = start - 1 if start > 0 else (n + start if start < 0 else 0)
    if s < 0 or s >= n:
        return ""
    return v[s:s + count]


def _like(v, pat):
    rx = "".join(".*" if c == "%" else "." if c == "_" else re.escape(c) for c in pat)
    return re.fullmatch(rx, v, re.DOTALL) is not None


def _and(vals):      # SQL three-valued AND
    if any(v is False for v in vals):
        return False
    return None if any(v is None for v in vals) else True


def _or(vals):
    if any(v is True for v in vals):
        return True
    return None if any(v is None for v in vals) else False


def py_eval(node, row):
    """row: dict column name -> python value; returns the python value (None = null)."""
    k = node.kind
    if k == "field":
        return row[node.desc["name"]]
    if k == "literal":
        if node.desc["is_null"]:
            return None
        v = node.desc["value"]
        return v.decode() if isinstance(v, bytes) else v
    kids = node.desc.get("children", [])
    if k == "if":
        c = py_eval(kids[0], row)
        return py_eval(kids[1], row) if c else py_eval(kids[2], row)
    if k == "and":
        return _and([py_eval(c, row) for c in kids])
    if k == "or":
        return _or([py_eval(c, row) for c in kids])
    if k == "in":
        x = py_eval(kids[0], row)
        vals = [v.decode() if isinstance(v, bytes) else v for v in node.desc["values"]]
        return None if x is None else x in vals
    f = node.desc["name"]
    a = [py_eval(c, row) for c in kids]
    if f == "isnull":
        return a[0] is None
    if f == "isnotnull":
        return a[0] is not None
    if f in ("hash32", "hash64"):
        if a[0] is None:
            return 0
        raw = a[0].encode()
        return murmurhash3_32(raw, seed=0, positive=False) if f == "hash32" else _py_murmur3_x64_128_h1(raw, 0)
    if f == "concat":
        return "".join("" if x is None else x for x in a)
    if any(x is None for x in a):
        return None
    if f == "concatOperator":
        return "".join(a)
    if f == "upper":
        return _upper(a[0])
    if f == "lower":
        return _lower(a[0])
    if f == "ltrim":
        return a[0].lstrip(" ")
    if f == "rtrim":
        return a[0].rstrip(" ")
    if f == "btrim":
        return a[0].strip(" ")
    if f == "substr":
        return _substr(*a)
    if f == "like":
        return _like(a[0], a[1])
    if f == "ilike":
        return _like(_lower(a[0]), _lower(a[1]))
    if f in ("equal", "not_equal", "less_than", "greater_than_or_equal_to"):
        x, y = a[0].encode(), a[1].encode()
        return {"equal": x == y, "not_equal": x != y, "less_than": x < y, "greater_than_or_equal_to": x >= y}[f]
    if f == "starts_with":
        return a[0].startswith(a[1])
    if f == "ends_with":
        return a[0].endswith(a[1])
    if f == "reverse":
        return a[0][::-1]
    if f == "replace":
        return a[0].replace(a[1], a[2]) if a[1] else a[0]
    if f in ("lpad", "rpad"):
        v, n, fill = a[0], a[1], (a[2] if len(a) == 3 else " ")
        if v == "" or n <= 0:
            return ""
        if len(v) >= n or fill == "":
            return v[:n] if len(v) > n else v
        pad = (fill * n)[:n - len(v)]
        return v + pad if f == "rpad" else pad + v
    if f == "castVARCHAR":
        return str(a[0])[:a[1]]
    if f == "octet_length":
        return len(a[0].encode())
    if f == "char_length":
        return len(a[0])
    if f == "greater_than":
        return a[0] > a[1]
    raise NotImplementedError(f)


@pytest.mark.parametrize("seed", range(40))
def test_oracle_string_trees_match_plain_python(seed):
    exprs, cond = F._string_expressions(100 + seed)
    batch = F._string_batch(seed, 300)
    rows = [dict(zip(batch.schema.names, vals)) for vals in zip(*[c.to_pylist() for c in batch.columns])]
    got = oracle.project(exprs, batch)
    for g, e in zip(got, exprs):
        want = [py_eval(e.root(), r) for r in rows]
        assert g.to_pylist() == want, f"seed {seed}: {e}"
    sel = oracle.filter_indices(cond, batch, "int32").to_pylist()
    assert sel == [i for i, r in enumerate(rows) if py_eval(cond.root(), r) is True], f"seed {seed}: {cond}"


@pytest.mark.parametrize("seed", range(30))
def test_oracle_trees_with_materialised_values_match_plain_python(seed):
    """reverse / replace / lpad / rpad / castVARCHAR(integer) / concat nested at any depth."""
    exprs, cond = F._tail_expressions(100 + seed)
    batch = F._string_batch(seed, 300)
    rows = [dict(zip(batch.schema.names, vals)) for vals in zip(*[c.to_pylist() for c in batch.columns])]
    got = oracle.project(exprs, batch)
    for g, e in zip(got, exprs):
        want = [py_eval(e.root(), r) for r in rows]
        assert g.to_pylist() == want, f"seed {seed}: {e}"
    sel = oracle.filter_indices(cond, batch, "int32").to_pylist()
    assert sel == [i for i, r in enumerate(rows) if py_eval(cond.root(), r) is True], f"seed {seed}: {cond}"
