"""Thread scaling of the CPU restatement on this host (what bench.py's cpu_baseline uses)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gandiva_amd import workloads as W
from oracle import oracle

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
rows = 1 << 24
batch = W.c2_batch(rows)
ex = W.c2_expressions()
outs = oracle.alloc_outputs(ex, rows)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > 2 * (os.cpu_count() or 1):
        break
    oracle.project(ex, batch, threads=th, out=outs)
    t = time.perf_counter()
    reps = 0
    while time.perf_counter() - t < 2.0:
        oracle.project(ex, batch, threads=th, out=outs)
        reps += 1
    el = time.perf_counter() - t
    print(f"threads {th:4d}: {rows * reps / el / 1e6:9.1f} M rows/s")
