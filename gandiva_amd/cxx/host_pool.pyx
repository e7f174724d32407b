# cython: language_level=3
# distutils: language = c++
"""pyarrow.MemoryPool objects over gandiva::HostMemoryPool (gandiva/host_memory.h): page-locked host memory
the GPUs address directly.  Arrays created in such a pool (`pa.array(x, memory_pool=pool)`) and the outputs
of a projector made with it (`pyarrow.gandiva.make_projector(schema, exprs, pool)`) are evaluated in place —
no staging copy in either direction.  Built next to pyarrow's own gandiva.pyx by build_pyarrow_gandiva.py."""
from libc.stdint cimport int64_t
from pyarrow.lib cimport MemoryPool
from pyarrow.includes.libarrow cimport CMemoryPool


cdef extern from "gandiva/host_memory.h" namespace "gandiva" nogil:
    cdef cppclass CHostMemoryPool "gandiva::HostMemoryPool"(CMemoryPool):
        CHostMemoryPool(int64_t chunk_bytes) except +
    int64_t CHostStagedBytes "gandiva::HostStagedBytes"()


cdef class HostMemoryPool(MemoryPool):
    """The C++ pool lives as long as the process: buffers hold a raw pointer to it, not a reference to this
    object, so it is never deleted (use `host_memory_pool()`, which hands out one pool per chunk size)."""
    cdef CHostMemoryPool* hp

    def __init__(self, int64_t chunk_bytes=64 << 20):
        self.hp = new CHostMemoryPool(chunk_bytes)
        self.init(self.hp)


_pools = {}


def host_memory_pool(chunk_bytes=64 << 20):
    pool = _pools.get(chunk_bytes)
    if pool is None:
        pool = _pools[chunk_bytes] = HostMemoryPool(chunk_bytes)
    return pool


def host_staged_bytes():
    """Bytes host-buffer evaluations have copied through staging blocks so far (process-wide)."""
    return CHostStagedBytes()
