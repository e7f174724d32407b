"""gandiva_amd — MI355X-native Projector / Filter evaluator for Arrow record batches.

Package layout (only what the hot path needs):
  csrc/        C++ core, HIP device-function library, AOT kernels, C ABI
  _capi.py     ctypes declarations of include/gandiva_amd.h
  gandiva.py   mirror of the reference lineage's `pyarrow.gandiva` Python API
  shard.py     row-range sharding across the GPUs of a node (one process per GPU, or one
               thread per device context inside one process)
"""
from .gandiva import (  # noqa: F401
    Condition, Configuration, DeviceBatch, DeviceColumn, DevicePool, Expression, Filter, FilterProject, FunctionSignature,
    GandivaError, HostArena, Node, Projector, SelectionVector, TreeExprBuilder, host_staged_bytes,
    device_count, get_device, get_registered_function_signatures, make_filter, make_filter_project, make_projector,
    physical_device_count, set_device, set_virtual_devices,
)

__version__ = "0.1.0"
