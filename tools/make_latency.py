"""Cold Make() latency per BASELINE plan (hipRTC compile, no disk cache) and the cost of a warm
re-Make with a different literal (round-1 verdict item 4).  No GPU needed for the cold numbers."""
import ctypes as C
import os
import sys
import time

os.environ["GDV_NO_DISK_CACHE"] = "1"
os.environ["GDV_PRECOMPILE_SKIP_GENERAL"] = "1"  # what Make compiles: the general variant of a string plan is built on demand
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa  # noqa: E402
import gandiva_amd as gandiva  # noqa: E402
from gandiva_amd import _capi, gandiva as gg, workloads as W  # noqa: E402

lib = _capi.lib()


def pp(schema, exprs, mode=0):
    sh = gg._make_schema(schema)
    arr = (C.c_void_p * len(exprs))(*[e._h for e in exprs])
    t = time.time()
    rc = lib.gdv_precompile_projector(sh, arr, len(exprs), mode)
    dt = time.time() - t
    assert rc == 0, _capi.last_error()
    return dt


def pf(schema, cond):
    sh = gg._make_schema(schema)
    t = time.time()
    rc = lib.gdv_precompile_filter(sh, cond._h)
    dt = time.time() - t
    assert rc == 0, _capi.last_error()
    return dt


print("# cold plan + hipRTC compile to a gfx950 code object, seconds (GDV_NO_DISK_CACHE=1)")
for w in ("c1", "c2", "c4", "c5"):
    print(w, round(pp(getattr(W, w + "_schema")(), getattr(W, w + "_expressions")()), 3))
print("c3", round(pf(W.c3_schema(), W.c3_condition()), 3))
try:
    import torch
    if torch.cuda.is_available():
        b = gandiva.TreeExprBuilder()
        s = W.c3_schema()
        a, bb = b.make_field(s.field(0)), b.make_field(s.field(1))
        for k1 in (499, 500, 501):
            cond = b.make_condition(b.make_and([
                b.make_function("greater_than", [a, b.make_literal(k1, pa.int64())], pa.bool_()),
                b.make_function("less_than", [bb, b.make_literal(250, pa.int64())], pa.bool_())]))
            t = time.time()
            gandiva.make_filter(s, cond)
            print(f"Filter::Make a > {k1}: {1e3 * (time.time() - t):.2f} ms")
except ImportError:
    pass
