"""ctypes binding of the C ABI declared in include/gandiva_amd.h.

Nothing here computes: it loads libgandiva_amd.so (built in-tree by
``make -C gandiva_amd/csrc`` / ``__graft_entry__.build()``) and declares prototypes.  A
missing library is a hard error — there is no Python or CPU fallback for evaluation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GANDIVA_AMD_LIB: another build of the same library — the sanitizer build of tools/sanitize_cpu_suite.sh)
LIB_PATH = os.environ.get("GANDIVA_AMD_LIB") or os.path.join(_HERE, "libgandiva_amd.so")


class gdv_type_t(C.Structure):
    _fields_ = [("id", C.c_int32), ("precision", C.c_int32), ("scale", C.c_int32)]


class gdv_config_t(C.Structure):
    _fields_ = [("optimize", C.c_int32), ("dump_ir", C.c_int32)]


class gdv_column_t(C.Structure):
    _fields_ = [
        ("validity", C.c_void_p), ("validity_size", C.c_int64),
        ("data", C.c_void_p), ("data_size", C.c_int64),
        ("offsets", C.c_void_p), ("offsets_size", C.c_int64),
        ("offset", C.c_int64),
    ]


class gdv_out_column_t(C.Structure):
    _fields_ = [
        ("validity", C.c_void_p), ("validity_size", C.c_int64),
        ("data", C.c_void_p), ("data_size", C.c_int64),
        ("offsets", C.c_void_p), ("offsets_size", C.c_int64),
    ]


class gdv_batch_t(C.Structure):
    _fields_ = [("num_rows", C.c_int64), ("cols", C.POINTER(gdv_column_t)), ("num_cols", C.c_int),
                ("outs", C.POINTER(gdv_out_column_t)), ("num_outs", C.c_int)]


class gdv_filter_batch_t(C.Structure):
    _fields_ = [("num_rows", C.c_int64), ("cols", C.POINTER(gdv_column_t)), ("num_cols", C.c_int),
                ("out_indices", C.c_void_p), ("max_slots", C.c_int64)]


class gdv_selection_t(C.Structure):
    _fields_ = [("mode", C.c_int32), ("indices", C.c_void_p), ("num_slots", C.c_int64)]


class gdv_shard_t(C.Structure):
    _fields_ = [("device", C.c_int32), ("cols", C.POINTER(gdv_column_t)), ("outs", C.POINTER(gdv_out_column_t)),
                ("out_indices", C.c_void_p), ("max_slots", C.c_int64), ("num_selected", C.c_int64)]


# every symbol include/gandiva_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
PROTOTYPES = [
    ("gdv_last_error", C.c_char_p, []),
    ("gdv_version", C.c_char_p, []),
    ("gdv_free_string", None, [_P]),
    ("gdv_schema_new", _P, []),
    ("gdv_schema_add_field", C.c_int, [_P, C.c_char_p, gdv_type_t, C.c_int]),
    ("gdv_schema_num_fields", C.c_int, [_P]),
    ("gdv_schema_free", None, [_P]),
    ("gdv_node_field", _P, [C.c_char_p, gdv_type_t]),
    ("gdv_node_literal", _P, [gdv_type_t, _P, C.c_int]),
    ("gdv_node_literal_bytes", _P, [gdv_type_t, C.c_char_p, C.c_int64, C.c_int]),
    ("gdv_node_function", _P, [C.c_char_p, C.POINTER(_P), C.c_int, gdv_type_t]),
    ("gdv_node_if", _P, [_P, _P, _P, gdv_type_t]),
    ("gdv_node_and", _P, [C.POINTER(_P), C.c_int]),
    ("gdv_node_or", _P, [C.POINTER(_P), C.c_int]),
    ("gdv_node_in", _P, [_P, gdv_type_t, _P, C.c_int]),
    ("gdv_node_in_bytes", _P, [_P, gdv_type_t, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int]),
    ("gdv_node_to_string", _P, [_P]),
    ("gdv_node_return_type", gdv_type_t, [_P]),
    ("gdv_node_free", None, [_P]),
    ("gdv_expression_new", _P, [_P, C.c_char_p, gdv_type_t]),
    ("gdv_condition_new", _P, [_P]),
    ("gdv_expression_to_string", _P, [_P]),
    ("gdv_expression_result_type", gdv_type_t, [_P]),
    ("gdv_expression_free", None, [_P]),
    ("gdv_projector_make", C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.POINTER(gdv_config_t), C.POINTER(_P)]),
    ("gdv_projector_num_outputs", C.c_int, [_P]),
    ("gdv_projector_output_type", gdv_type_t, [_P, C.c_int]),
    ("gdv_projector_output_sizes", C.c_int, [_P, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("gdv_projector_evaluate", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int, C.POINTER(gdv_selection_t), C.POINTER(gdv_out_column_t), C.c_int, C.c_int, _P, C.c_uint32]),
    ("gdv_projector_evaluate_selected", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int, C.POINTER(gdv_selection_t), _P, C.POINTER(gdv_out_column_t), C.c_int, _P, C.c_uint32]),
    ("gdv_projector_evaluate_async", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int, C.POINTER(gdv_selection_t), _P, C.POINTER(gdv_out_column_t), C.c_int, _P, _P]),
    ("gdv_projector_evaluate_many", C.c_int, [_P, C.POINTER(gdv_batch_t), C.c_int, _P, C.c_uint32]),
    ("gdv_projector_dump_ir", _P, [_P]),
    ("gdv_projector_path_hint", C.c_int, [_P]),
    ("gdv_projector_free", None, [_P]),
    ("gdv_filter_make", C.c_int, [_P, _P, C.POINTER(gdv_config_t), C.POINTER(_P)]),
    ("gdv_filter_evaluate", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int64), C.c_int, _P]),
    ("gdv_filter_evaluate_async", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int, C.c_int, _P, C.c_int64, _P, _P]),
    ("gdv_projector_make_from_proto", C.c_int, [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_int, C.POINTER(gdv_config_t), C.POINTER(_P)]),
    ("gdv_filter_make_from_proto", C.c_int, [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.POINTER(gdv_config_t), C.POINTER(_P)]),
    ("gdv_proto_describe", _P, [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_int]),
    ("gdv_filter_evaluate_many", C.c_int, [_P, C.POINTER(gdv_filter_batch_t), C.c_int, C.c_int, C.POINTER(C.c_int64), _P, _P, C.c_uint32]),
    ("gdv_filter_dump_ir", _P, [_P]),
    ("gdv_filter_free", None, [_P]),
    ("gdv_filter_set_tuning", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("gdv_filter_project_make", C.c_int, [_P, _P, C.POINTER(_P), C.c_int, C.c_int, C.POINTER(gdv_config_t), C.POINTER(_P)]),
    ("gdv_filter_project_make_from_proto", C.c_int, [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_int, C.POINTER(gdv_config_t), C.POINTER(_P)]),
    ("gdv_filter_project_num_outputs", C.c_int, [_P]),
    ("gdv_filter_project_output_type", gdv_type_t, [_P, C.c_int]),
    ("gdv_filter_project_evaluate", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int, C.POINTER(gdv_out_column_t), C.c_int, _P, C.c_int64, C.POINTER(C.c_int64), _P, C.c_int, _P, C.c_uint32]),
    ("gdv_filter_project_dump_ir", _P, [_P]),
    ("gdv_filter_project_kernel_shape", C.c_int, [_P]),
    ("gdv_filter_project_set_tuning", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("gdv_filter_project_free", None, [_P]),
    ("gdv_precompile_filter_project", C.c_int, [_P, _P, C.POINTER(_P), C.c_int, C.c_int]),
    ("gdv_compile_regex", C.c_int, [C.c_char_p, C.c_int64, _P]),
    ("gdv_compile_date_format", C.c_int, [C.c_char_p, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int64)]),
    ("gdv_registry_size", C.c_int, []),
    ("gdv_registry_get", C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(gdv_type_t), C.POINTER(gdv_type_t), C.c_int, C.POINTER(C.c_int)]),
    ("gdv_device_stream_ceiling", C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("gdv_device_stream_ceiling_on", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("gdv_device_count", C.c_int, []),
    ("gdv_physical_device_count", C.c_int, []),
    ("gdv_set_virtual_devices", C.c_int, [C.c_int]),
    ("gdv_set_device", C.c_int, [C.c_int]),
    ("gdv_get_device", C.c_int, []),
    ("gdv_shard_bounds", C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("gdv_projector_evaluate_sharded", C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.POINTER(gdv_shard_t), C.c_int, C.c_uint32]),
    ("gdv_filter_evaluate_sharded", C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.POINTER(gdv_shard_t), C.c_int, C.c_uint32,
                                              C.POINTER(C.c_int64)]),
    ("gdv_filter_gather_sharded", C.c_int, [C.POINTER(gdv_shard_t), C.c_int, C.c_int, C.c_int, _P, C.c_int64]),
    ("gdv_projector_evaluate_host_sharded", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int,
                                                      C.POINTER(gdv_out_column_t), C.c_int, C.POINTER(C.c_int32), C.c_int]),
    ("gdv_filter_evaluate_host_sharded", C.c_int, [_P, C.c_int64, C.POINTER(gdv_column_t), C.c_int, C.c_int, _P, C.c_int64,
                                                   C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int]),
    ("gdv_device_pool_create", C.c_int, [C.POINTER(_P)]),
    ("gdv_device_pool_destroy", None, [_P]),
    ("gdv_device_pool_reserve_set", C.c_int, [_P, C.c_int, C.c_int64, C.c_int, C.POINTER(_P), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]),
    ("gdv_device_pool_alloc", C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    ("gdv_device_pool_free", C.c_int, [_P, _P]),
    ("gdv_device_pool_trim", C.c_int, [_P]),
    ("gdv_device_pool_bytes", C.c_int64, [_P, C.POINTER(C.c_int64)]),
    ("gdv_device_num_cus", C.c_int, []),
    ("gdv_device_arch", C.c_char_p, []),
    ("gdv_device_alloc", C.c_int, [C.c_int64, C.POINTER(_P)]),
    ("gdv_device_free", C.c_int, [_P]),
    ("gdv_memcpy_h2d", C.c_int, [_P, _P, C.c_int64]),
    ("gdv_memcpy_d2h", C.c_int, [_P, _P, C.c_int64]),
    ("gdv_device_synchronize", C.c_int, []),
    ("gdv_host_register", C.c_int, [_P, C.c_int64]),
    ("gdv_host_unregister", C.c_int, [_P]),
    ("gdv_host_alloc", C.c_int, [C.c_int64, C.POINTER(_P)]),
    ("gdv_host_free", C.c_int, [_P]),
    ("gdv_host_staged_bytes", C.c_int64, []),
    ("gdv_device_hbm_ceilings", C.c_int, [C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("gdv_projector_evaluate_device_array", C.c_int, [_P, _P, C.POINTER(gdv_selection_t), C.POINTER(gdv_out_column_t), C.c_int, _P, C.c_uint32]),
    ("gdv_filter_evaluate_device_array", C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    ("gdv_projector_evaluate_flat", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int,
                                              C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int64),
                                              C.POINTER(C.c_int64), C.c_int, C.c_int]),
    ("gdv_filter_evaluate_flat", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int,
                                           C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_int]),
    ("gdv_projector_evaluate_export", C.c_int, [_P, _P, C.POINTER(gdv_selection_t), _P, _P, _P]),
    ("gdv_tier0_program", _P, [_P, C.POINTER(_P), C.c_int, C.c_int]),
    ("gdv_tier0_launches", C.c_int64, []),
    ("gdv_shutdown", None, []),
    ("gdv_precompile_projector", C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int]),
    ("gdv_precompile_filter", C.c_int, [_P, _P]),
    ("gdv_kernel_library_tag", _P, [C.c_char_p, C.c_char_p]),
    ("gdv_kernel_library_items", _P, [C.c_char_p, C.c_char_p]),
    ("gdv_device_library_source", C.c_char_p, []),
]



# ABI-stable structs of the Arrow C data / C device data interfaces (arrow/c/abi.h)
class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64),
    ("n_buffers", C.c_int64), ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))),
    ("dictionary", C.POINTER(ArrowArray)), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowDeviceArray(C.Structure):
    _fields_ = [("array", ArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32),
                ("sync_event", C.c_void_p), ("reserved", C.c_int64 * 3)]


class ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p),
                ("flags", C.c_int64), ("n_children", C.c_int64), ("children", C.c_void_p),
                ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


def release_c_struct(struct):
    """Calls the release callback of an ArrowArray / ArrowSchema / ArrowDeviceArray, if set."""
    target = struct.array if isinstance(struct, ArrowDeviceArray) else struct
    if target.release:
        C.CFUNCTYPE(None, C.c_void_p)(target.release)(C.addressof(target))


_lib = None


def lib():
    """The loaded shared library (loaded once; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C gandiva_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`. gandiva_amd has no "
                "CPU/Python fallback.")
        # ONE HIP runtime per process: PyTorch wheels bundle their own libamdhip64 (same
        # SONAME as /opt/rocm's).  If torch is going to be used in this process it must be
        # loaded first, so this library binds to the runtime torch already brought in;
        # loading ours first would leave two runtimes and torch then finds "no HIP GPUs".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, restype, argtypes in PROTOTYPES:
            fn = getattr(l, name)  # AttributeError = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = l
        # the background compiler (tier 0) is stopped before the interpreter goes away: deterministic, whatever order the
        # process's shared libraries are torn down in
        import atexit
        atexit.register(l.gdv_shutdown)
    return _lib


def take_string(ptr):
    """Copy and free a malloc'ed C string returned by the library."""
    if not ptr:
        return None
    s = C.string_at(ptr).decode("utf-8", errors="replace")
    lib().gdv_free_string(ptr)
    return s


def last_error():
    return lib().gdv_last_error().decode("utf-8", errors="replace")
