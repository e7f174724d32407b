"""Evaluate latency vs batch size (device-resident, sync after every call): where launch and
host overheads — not HBM — set the time.  python tools/latency_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W


def timeit(fn, reps=200):
    fn(); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


print(f"{'rows':>10} {'C2 project us':>14} {'C3 filter us':>13} {'C5 strings us':>14} {'C2 host-path us':>16}")
for lg in (10, 12, 14, 16, 18, 20, 22, 24):
    n = 1 << lg
    d2 = W.c2_device_batch(n)
    p2 = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
    o2 = p2.evaluate_device(d2)
    t2 = timeit(lambda: p2.evaluate_device(d2, outputs=o2))
    d3 = W.c3_device_batch(n)
    f3 = gandiva.make_filter(W.c3_schema(), W.c3_condition())
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    t3 = timeit(lambda: f3.evaluate_device(d3, "int32", out=out))
    d5 = W.c5_device_batch(n)
    p5 = gandiva.make_projector(W.c5_schema(), W.c5_expressions(), None)
    o5 = p5.evaluate_device(d5)
    t5 = timeit(lambda: p5.evaluate_device(d5, outputs=o5), reps=100)
    hb = W.c2_batch(n) if lg <= 22 else None
    th = timeit(lambda: p2.evaluate(hb), reps=20) if hb is not None else float("nan")
    print(f"{n:>10} {t2:>14.1f} {t3:>13.1f} {t5:>14.1f} {th:>16.1f}")
