"""bench.py's CPU-side legs (no GPU): the argument parser, the per-workload tables and the short cpu_baseline of every
sub-workload of the driver's line — the oracle ("port") timed on a bounded sample, as the task's measurement contract asks."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_are_the_drivers_invocation_and_every_workload_is_described(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.workload, a.scaling, a.no_extras) == (1, "c2", "weak", False)
    assert a.extras.split(",") == ["c1", "c3", "k2f", "c5", "c4"] and a.pool_candidates == 8 and a.placements == 1
    for w in ["c2"] + a.extras.split(","):
        assert w in bench.WORKLOAD_TEXT and w in bench.DEFAULT_ROWS and w in bench.STRONG_ROWS and w in bench.CEILING_SHAPE
    assert bench.DEFAULT_ROWS["c2"] == 1 << 28 and bench.DEFAULT_ROWS["c3"] == 10**9 and bench.DEFAULT_ROWS["c5"] == 10**8
    assert bench.STRONG_ROWS["c4"] == 6 * 10**9 and bench.DEFAULT_ROWS["c4"] * 8 == bench.STRONG_ROWS["c4"]


def test_short_cpu_baselines_run_on_bounded_samples():
    bench = _bench()
    for name in ("c1", "c3", "k2f", "c4", "c5"):
        b = bench.short_cpu_baseline(name, 0.2)
        assert b["kind"] == "port" and b["unit"] == "million rows/s" and b["value"] > 0 and b["cores"] >= 1, (name, b)
        assert "passes x" in b["sample"]
