"""The reference lineage's Python acceptance suite, UNMODIFIED: pyarrow/tests/test_gandiva.py
is imported from the installed pyarrow and its test functions are called as they are, with
`pyarrow.gandiva` provided by pyarrow's own gandiva.pyx compiled against gandiva_amd's C++
API (libgandiva.so -> libgandiva_amd.so -> HIP kernels).  Functions that only build trees run
on the CPU; the ones that evaluate need the GPU."""
import importlib

import pytest


@pytest.fixture(scope="module")
def ref_tests():
    from gandiva_amd import pyarrow_gandiva
    pyarrow_gandiva.load()
    return importlib.import_module("pyarrow.tests.test_gandiva")


HOST_ONLY = ["test_literals", "test_to_string", "test_rejects_none",
             "test_get_registered_function_signatures"]
NEED_GPU = ["test_tree_exp_builder", "test_table", "test_filter", "test_in_expr", "test_boolean",
            "test_regex", "test_filter_project"]


@pytest.mark.parametrize("name", HOST_ONLY)
def test_reference_python_test_host(ref_tests, name):
    getattr(ref_tests, name)()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NEED_GPU)
def test_reference_python_test_gpu(ref_tests, name):
    getattr(ref_tests, name)()


def test_every_reference_test_is_accounted_for(ref_tests):
    names = {n for n in dir(ref_tests) if n.startswith("test_")}
    # test_in_expr_todo is skipped upstream too ("Gandiva C++ did not have *real* binary,
    # time and date support")
    assert names == set(HOST_ONLY) | set(NEED_GPU) | {"test_in_expr_todo"}


def test_the_host_memory_pool_is_a_pyarrow_memory_pool():
    import pyarrow as pa
    from gandiva_amd import pyarrow_gandiva
    pool = pyarrow_gandiva.host_memory_pool(4 << 20)
    assert isinstance(pool, pa.MemoryPool) and pool.backend_name == "gandiva_amd-host"
    assert pool is pyarrow_gandiva.host_memory_pool(4 << 20)      # one pool per chunk size, never deleted
    assert pyarrow_gandiva.host_staged_bytes() >= 0


@pytest.mark.gpu
def test_pyarrow_gandiva_on_a_host_memory_pool_copies_nothing():
    """pyarrow.gandiva, unmodified, over arrays and outputs that live in a gandiva::HostMemoryPool: the kernels read
    the columns and write the results where they are (no staging copy in either direction)."""
    import numpy as np
    import pyarrow as pa
    from gandiva_amd import pyarrow_gandiva
    gandiva = pyarrow_gandiva.load()
    pool = pyarrow_gandiva.host_memory_pool(16 << 20)
    n = 100_003
    rng = np.random.default_rng(4)
    a_np, b_np = rng.normal(0, 100, n), rng.normal(0, 100, n)
    mask = rng.random(n) < 0.1
    # (from Python lists: pa.array over a numpy array wraps numpy's memory instead of allocating from the pool)
    a = pa.array([None if m else float(x) for x, m in zip(a_np, mask)], pa.float64(), memory_pool=pool)
    b = pa.array(b_np.tolist(), pa.float64(), memory_pool=pool)
    batch = pa.RecordBatch.from_arrays([a, b], names=["a", "b"])
    builder = gandiva.TreeExprBuilder()
    na, nb = builder.make_field(batch.schema.field(0)), builder.make_field(batch.schema.field(1))
    s = builder.make_function("add", [builder.make_function("multiply", [na, nb], pa.float64()), nb], pa.float64())
    projector = gandiva.make_projector(batch.schema, [builder.make_expression(s, pa.field("r", pa.float64()))], pool)
    before = pyarrow_gandiva.host_staged_bytes()
    r, = projector.evaluate(batch)
    assert pyarrow_gandiva.host_staged_bytes() == before, "a buffer went through the staging block"
    want = pa.array(a_np * b_np + b_np, pa.float64(), mask=mask)
    assert r.equals(want)
    # the same batch in pyarrow's default pool is staged — and gives the same answer
    batch2 = pa.RecordBatch.from_arrays([pa.array(a_np, pa.float64(), mask=mask), pa.array(b_np, pa.float64())], names=["a", "b"])
    r2, = projector.evaluate(batch2)
    assert pyarrow_gandiva.host_staged_bytes() > before and r2.equals(want)
