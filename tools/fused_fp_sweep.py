"""Round 5: the fused filter -> project kernel's shapes side by side on ONE box, in ONE process (C3 scale: 10^9 int64
rows x 2, a > 499 AND b < k2, project a + b, uint32 selection vector).  Every variant is a fresh Make under its own
GDV_FP_WINDOW / GDV_U / GDV_WAVES (read at Make); HIP events, 8 Evaluates each after one untimed.

  python tools/fused_fp_sweep.py [rows] [k2 ...]      k2 = 250 -> selectivity 1/8 (default), 1000 -> 1/2
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
k2s = [int(x) for x in sys.argv[2:]] or [250]
db = W.c3_device_batch(n)
b = gandiva.TreeExprBuilder()
a, c = (b.make_field(W.c3_schema().field(i)) for i in range(2))
expr = b.make_expression(b.make_function("add", [a, c], pa.int64()), pa.field("s", pa.int64()))
idx = torch.empty(n, dtype=torch.int32, device="cuda")


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def condition(k2):
    f = [b.make_field(W.c3_schema().field(i)) for i in range(2)]
    return b.make_condition(b.make_and([b.make_function("greater_than", [f[0], b.make_literal(499, pa.int64())], pa.bool_()),
                                        b.make_function("less_than", [f[1], b.make_literal(k2, pa.int64())], pa.bool_())]))


VARIANTS = [("direct (round 4)      U16 W8", {"GDV_FP_WINDOW": "0"}),
            ("window, K=1          U16 W8", {"GDV_FP_K": "1"}),
            ("window, K=2          U16 W8", {"GDV_FP_K": "2"}),
            ("window, K=3 (default)      ", {}),
            ("window, K=3          U16 W4", {"GDV_WAVES": "4"}),
            ("window, K=3, 6 KB window   ", {"GDV_FP_WINDOW": "6144"}),
            ("window, K=1, NO look-back  ", {"GDV_FP_K": "1", "GDV_FP_EXPERIMENT": "1"}),
            ("window, K=3, NO look-back  ", {"GDV_FP_EXPERIMENT": "1"}),
            ("window, K=7          U8  W8", {"GDV_FP_K": "7", "GDV_U": "8", "GDV_FP_WINDOW": "12288"}),
            ("window, K=3          U16 W16", {"GDV_WAVES": "16", "GDV_FP_WINDOW": "4992"})]
only = os.environ.get("FP_VARIANTS")
for k2 in k2s:
    flt = gandiva.make_filter(W.c3_schema(), condition(k2))
    t_f = timed(lambda: flt.evaluate_device(db, "int32", out=idx, sync=False))
    print(f"# rows {n}, a > 499 AND b < {k2}: filter alone {t_f:.3f} ms", flush=True)
    for vi, (label, env) in enumerate(VARIANTS):
        if only and str(vi) not in only.split(","):
            continue
        for k in ("GDV_FP_WINDOW", "GDV_U", "GDV_WAVES", "GDV_FP_EXPERIMENT", "GDV_FP_K"):
            os.environ.pop(k, None)
        os.environ.update(env)
        row = []
        for dtype in ("int32", None):
            fp = gandiva.make_filter_project(W.c3_schema(), condition(k2), [expr], dtype)
            state = {"o": None}

            def run():
                state["o"], _ = fp.evaluate_device(db, outputs=state["o"], indices=idx if dtype else None, sync=False)
            t = timed(run)
            row.append(f"{'with selection vector' if dtype else 'projection only'} {t:6.3f} ms (shape {fp.kernel_shape})")
            del fp, state
        print(f"{label:28s} " + "   ".join(row), flush=True)
