"""`pyarrow.gandiva` — the reference lineage's own Cython binding — on the HIP backend.

pyarrow ships gandiva.pyx but no compiled module (the wheel is built without Gandiva).
`load()` builds that file against gandiva_amd's C++ API if needed
(gandiva_amd/cxx/build_pyarrow_gandiva.py) and registers the result as `pyarrow.gandiva`, so
existing code — `import pyarrow.gandiva as gandiva` — runs unchanged on MI355X.

Quick start (host batches, the fast way).  With pyarrow's default pool every host batch is STAGED through a
page-locked block (two memcpys + two DMA copies: 103-133 us per 16K-row batch); arrays and outputs that live in
`host_memory_pool()` are read and written in place by the kernel (51 us):

    from gandiva_amd import pyarrow_gandiva
    gandiva = pyarrow_gandiva.load()                     # = import pyarrow.gandiva as gandiva
    pool = pyarrow_gandiva.host_memory_pool()            # page-locked, GPU-addressable arrow MemoryPool
    a = pa.array(values, pa.float64(), memory_pool=pool) # put the columns there (builders, IPC readers take a pool too)
    batch = pa.RecordBatch.from_arrays([a], names=["a"])
    proj = gandiva.make_projector(batch.schema, exprs, pool)   # outputs are allocated from `pool`
    out, = proj.evaluate(batch)                          # nothing is staged: pyarrow_gandiva.host_staged_bytes() stays put

HBM-resident batches (`gandiva_amd.DeviceBatch`, the Arrow C Device Data Interface) skip the host link altogether.
"""
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def load(build=True):
    if "pyarrow.gandiva" in sys.modules:
        return sys.modules["pyarrow.gandiva"]
    try:
        import torch  # noqa: F401  (one HIP runtime per process: see _capi.lib)
    except ImportError:
        pass
    import pyarrow  # noqa: F401
    sys.path.insert(0, os.path.join(_HERE, "cxx"))
    try:
        import build_pyarrow_gandiva as b
    finally:
        sys.path.pop(0)
    path = b.build() if build else os.path.join(b.OUT_DIR, "gandiva" + __import__("sysconfig").get_config_var("EXT_SUFFIX"))
    loader = importlib.machinery.ExtensionFileLoader("pyarrow.gandiva", path)
    spec = importlib.util.spec_from_file_location("pyarrow.gandiva", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pyarrow.gandiva"] = mod
    try:
        loader.exec_module(mod)
    except BaseException:
        del sys.modules["pyarrow.gandiva"]
        raise
    import pyarrow as pa
    pa.gandiva = mod
    return mod


def host_memory_pool(chunk_bytes=64 << 20):
    """A `pyarrow.MemoryPool` over page-locked host memory the GPUs address directly (gandiva::HostMemoryPool).
    `pa.array(x, memory_pool=pool)` puts a column there, `pyarrow.gandiva.make_projector(schema, exprs, pool)`
    the outputs: such batches are evaluated in place, nothing is staged."""
    return _host_pool_module().host_memory_pool(chunk_bytes)


def host_staged_bytes():
    return _host_pool_module().host_staged_bytes()


def _host_pool_module():
    name = "gandiva_amd._pyarrow_host_pool"
    if name in sys.modules:
        return sys.modules[name]
    load()
    sys.path.insert(0, os.path.join(_HERE, "cxx"))
    try:
        import build_pyarrow_gandiva as b
    finally:
        sys.path.pop(0)
    path = os.path.join(b.OUT_DIR, "host_pool" + __import__("sysconfig").get_config_var("EXT_SUFFIX"))
    loader = importlib.machinery.ExtensionFileLoader("host_pool", path)
    spec = importlib.util.spec_from_file_location("host_pool", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    sys.modules[name] = mod
    return mod
