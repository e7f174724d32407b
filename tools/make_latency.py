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


# ---- round 6: tier 0.  Make of an UNSEEN tree (fresh trees: other columns / constants' shapes than anything compiled above;
# no disk cache), the first Evaluate (interpreted), and the same Evaluate once the background compiler has delivered.
def tier0_section():
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        return
    print("# tier 0 (round 6): Make of an unseen tree returns once the plan and its post-fix program exist; hipRTC runs behind")
    rng = np.random.default_rng(0)
    n = 1 << 22
    cases = []
    b = gandiva.TreeExprBuilder()
    # a C2-like projection over other column names (a new plan: nothing of it is cached)
    sch = pa.schema([pa.field(c, pa.float64()) for c in "pqrs"])
    p, q, r, s = (b.make_field(sch.field(i)) for i in range(4))
    f = lambda name, x, y: b.make_function(name, [x, y], pa.float64())  # noqa: E731
    exprs = [b.make_expression(f("add", f("multiply", p, q), f("subtract", r, s)), pa.field("e0", pa.float64())),
             b.make_expression(f("multiply", f("add", p, r), f("add", q, s)), pa.field("e1", pa.float64())),
             b.make_expression(f("subtract", f("multiply", p, p), f("multiply", q, q)), pa.field("e2", pa.float64()))]
    batch = pa.RecordBatch.from_arrays([pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.1) for _ in range(4)], schema=sch)
    cases.append(("projector, 3 float64 expressions over 4 columns", sch, exprs, None, batch))
    sch2 = pa.schema([pa.field("k", pa.int32()), pa.field("v", pa.int64())])
    k, v = b.make_field(sch2.field(0)), b.make_field(sch2.field(1))
    cond = b.make_condition(b.make_or([b.make_function("less_than", [k, b.make_literal(17, pa.int32())], pa.bool_()),
                                       b.make_function("greater_than", [v, b.make_literal(900, pa.int64())], pa.bool_())]))
    batch2 = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 1000, n, dtype=np.int32)), pa.array(rng.integers(0, 1000, n))], schema=sch2)
    cases.append(("filter, k < 17 OR v > 900 (int32, int64)", sch2, None, cond, batch2))
    for name, schema, ex, cd, bt in cases:
        db = gandiva.DeviceBatch.from_arrow(bt)
        torch.cuda.synchronize()
        t = time.perf_counter()
        op = gandiva.make_projector(schema, ex, None) if ex else gandiva.make_filter(schema, cd)
        make_ms = 1e3 * (time.perf_counter() - t)
        run = (lambda: op.evaluate_device(db)) if ex else (lambda: op.evaluate_device(db, "int32"))
        before = lib.gdv_tier0_launches()
        t = time.perf_counter()
        first = run()
        torch.cuda.synchronize()
        first_ms = 1e3 * (time.perf_counter() - t)
        interpreted = lib.gdv_tier0_launches() - before
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 60:
            before = lib.gdv_tier0_launches()
            run()
            if lib.gdv_tier0_launches() == before:
                break
            time.sleep(0.02)
        arrived_s = time.perf_counter() - t0

        def timed(fn, reps=10):
            fn(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t) / reps
        spec_ms = timed(run)
        print(f"{name}: Make {make_ms:.2f} ms; first Evaluate over {n} rows {first_ms:.2f} ms ({'interpreted' if interpreted else 'specialised'}); "
              f"specialised kernel in use {arrived_s:.2f} s after the first Evaluate; Evaluate on it {spec_ms:.3f} ms")
    # steady state of the interpreter itself, next to the specialised kernel, on the C2 / C3 shapes
    print("# the interpreter's own rate (GDV_FORCE_TIER0=1 in a subprocess) next to the specialised kernels, 2^24 rows:")
    import subprocess
    code = ("import sys, time, torch; sys.path.insert(0, %r); import gandiva_amd as g; from gandiva_amd import workloads as W\n"
            "db = W.c2_device_batch(1 << 24); p = g.make_projector(W.c2_schema(), W.c2_expressions(), None); o = p.evaluate_device(db)\n"
            "d3 = W.c3_device_batch(1 << 24); f = g.make_filter(W.c3_schema(), W.c3_condition()); out = torch.empty(1 << 24, dtype=torch.int32, device='cuda'); f.evaluate_device(d3, 'int32', out=out)\n"
            "def tm(fn):\n    fn(); torch.cuda.synchronize(); t = time.perf_counter()\n    for _ in range(20): fn()\n    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / 20\n"
            "print('C2 shape %%.3f ms, C3 shape %%.3f ms' %% (tm(lambda: p.evaluate_device(db, outputs=o, sync=False)), tm(lambda: f.evaluate_device(d3, 'int32', out=out))))\n"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for label, env in (("tier 0 (interpreted)", {"GDV_FORCE_TIER0": "1"}), ("specialised", {"GDV_NO_TIER0": "1"})):
        e = dict(os.environ, **env)
        e.pop("GDV_NO_DISK_CACHE", None)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        print(f"{label}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")


try:
    tier0_section()
except ImportError:
    pass
