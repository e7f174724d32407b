"""In-process multi-device evaluation (round 3): one process, one host thread per device context,
ONE logical batch row-sharded with gdv_shard_bounds — through the C ABI from C++
(tests/cxx/multi_device_test.cc) and through the Python mirror with threads."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import _capi, shard, workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build(tmp_path):
    exe = str(tmp_path / "multi_device_test")
    lib_dir = os.path.join(ROOT, "gandiva_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread",
                           "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "cxx", "multi_device_test.cc"),
                           "-o", exe, "-L", lib_dir, "-lgandiva_amd", "-Wl,-rpath," + lib_dir])
    return exe


def test_multi_device_test_builds_and_shard_bounds_agree_with_the_python_helper(tmp_path):
    _build(tmp_path)
    lib = _capi.lib()
    rng = np.random.default_rng(1)
    for _ in range(300):
        rows = int(rng.integers(0, 10_000_000))
        n = int(rng.integers(1, 17))
        prev = 0
        for r in range(n):
            lo, hi = C.c_int64(-1), C.c_int64(-1)
            assert lib.gdv_shard_bounds(rows, n, r, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == shard.shard_bounds(rows, n, r)
            assert lo.value == prev and lo.value % 1024 == 0 or lo.value == rows
            prev = hi.value
        assert prev == rows
    lo, hi = C.c_int64(0), C.c_int64(0)
    assert lib.gdv_shard_bounds(10, 0, 0, C.byref(lo), C.byref(hi)) != 0
    assert lib.gdv_shard_bounds(10, 2, 2, C.byref(lo), C.byref(hi)) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows", [(2, 300_000), (8, 300_000), (8, 1_000_003), (3, 5_000)])
def test_one_batch_sharded_over_n_device_contexts_equals_the_unsharded_evaluation(tmp_path, n, rows):
    dump = tmp_path / "dump"
    dump.mkdir()
    r = subprocess.run([_build(tmp_path), str(n), str(rows), str(dump)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "multi-device ok" in r.stdout
    _compare_the_dump_with_the_oracle(dump, rows)


def _compare_the_dump_with_the_oracle(dump, rows):
    """The C++ test proved sharded == unsharded; here the unsharded results are held against VALUES:
    the batch is rebuilt from the dumped Arrow buffers and the three plans are evaluated by the oracle."""
    from oracle import oracle
    from helpers import assert_bit_exact

    def raw(name):
        return np.fromfile(str(dump / name), dtype=np.uint8)

    def column(name, t):
        v = raw(name + ".validity")
        validity = pa.py_buffer(v) if v.size else None
        if t == pa.string():
            return pa.Array.from_buffers(t, rows, [validity, pa.py_buffer(raw(name + ".offsets")), pa.py_buffer(raw(name + ".data"))])
        return pa.Array.from_buffers(t, rows, [validity, pa.py_buffer(raw(name + ".data"))])
    f64, i64, st = pa.float64(), pa.int64(), pa.string()
    schema = pa.schema([("a", f64), ("b", f64), ("k1", i64), ("k2", i64), ("s", st)])
    batch = pa.RecordBatch.from_arrays([column("a", f64), column("b", f64), column("k1", i64), column("k2", i64),
                                        column("s", st)], schema=schema)
    bld = gandiva.TreeExprBuilder()
    fa, fb, fk1, fk2, fs = (bld.make_field(schema.field(i)) for i in range(5))
    c2 = [bld.make_expression(bld.make_function("add", [fa, fb], f64), pa.field("e0", f64)),
          bld.make_expression(bld.make_function("multiply", [fa, fb], f64), pa.field("e1", f64)),
          bld.make_expression(bld.make_function("multiply", [bld.make_function("subtract", [fa, fb], f64), fa], f64),
                              pa.field("e2", f64))]
    for e, w in enumerate(oracle.project(c2, batch)):
        got = pa.Array.from_buffers(f64, rows, [pa.py_buffer(raw(f"c2_{e}.validity")), pa.py_buffer(raw(f"c2_{e}.data"))])
        assert_bit_exact(got, w, f"C2-shaped expression {e}")
    cond = bld.make_condition(bld.make_and([
        bld.make_function("greater_than", [fk1, bld.make_literal(499, i64)], pa.bool_()),
        bld.make_function("less_than", [fk2, bld.make_literal(250, i64)], pa.bool_())]))
    want = oracle.filter_indices(cond, batch, "int32").to_numpy()
    assert np.array_equal(raw("c3.indices").view(np.uint32), want.view(np.uint32))
    c5 = [bld.make_expression(bld.make_function("like", [fs, bld.make_literal("%spark%", st)], pa.bool_()),
                              pa.field("m", pa.bool_())),
          bld.make_expression(bld.make_function("substr", [fs, bld.make_literal(2, i64), bld.make_literal(5, i64)], st),
                              pa.field("sub", st)),
          bld.make_expression(bld.make_function("upper", [fs], st), pa.field("up", st))]
    for e, w in enumerate(oracle.project(c5, batch)):
        v = pa.py_buffer(raw(f"c5_{e}.validity"))
        if e == 0:
            got = pa.Array.from_buffers(pa.bool_(), rows, [v, pa.py_buffer(raw("c5_0.data"))])
        else:
            got = pa.Array.from_buffers(st, rows, [v, pa.py_buffer(raw(f"c5_{e}.offsets")), pa.py_buffer(raw(f"c5_{e}.data"))])
        assert_bit_exact(got, w, f"C5-shaped expression {e}")


@pytest.mark.gpu
def test_python_threads_each_on_its_own_device_context():
    """The Python mirror: gandiva.set_device(r) per thread, one shared Projector handle evaluated on
    every context (the handle is loaded onto a context the first time it runs there); the shards'
    concatenation equals the oracle on the whole batch."""
    import torch
    from oracle import oracle
    from helpers import assert_bit_exact
    n_dev = 4
    gandiva.set_virtual_devices(max(n_dev, gandiva.physical_device_count()))
    batch = W.c2_batch(200_003)
    exprs = W.c2_expressions()
    proj = gandiva.make_projector(batch.schema, exprs, None)
    want = oracle.project(exprs, batch)
    got = [None] * n_dev
    errors = []

    def work(r):
        try:
            gandiva.set_device(r)
            assert gandiva.get_device() == r
            part, lo = shard.shard_record_batch(batch, n_dev, r)
            if part.num_rows:
                dev = proj.evaluate_device(gandiva.DeviceBatch.from_arrow(part))
                torch.cuda.synchronize()
                got[r] = [d.to_arrow() for d in dev]
            else:
                got[r] = []
        except Exception as e:  # surfaced below
            errors.append((r, e))
    threads = [threading.Thread(target=work, args=(r,)) for r in range(n_dev)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for e in range(len(exprs)):
        parts = [g[e] for g in got if g]
        assert_bit_exact(pa.concat_arrays(parts), want[e], f"expression {e}")
