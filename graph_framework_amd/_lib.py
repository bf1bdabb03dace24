"""Loader for libgf_hip.so (the C ABI of include/gf_hip.h).

There is no CPU fallback: if the library is missing or cannot be loaded the
import of anything that computes fails loudly.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgf_hip.so")
CACHE_DIR = os.path.join(HERE, "kernel_cache")
WORKLOAD_DIR = os.path.join(HERE, "workloads")

GFIR_F32 = 0
GFIR_F64 = 1
GFIR_C32 = 2
GFIR_C64 = 3


class KernelInfo(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_uint32),
                ("num_inputs", ctypes.c_uint32), ("num_outputs", ctypes.c_uint32),
                ("num_setters", ctypes.c_uint32), ("num_tables", ctypes.c_uint32),
                ("num_instructions", ctypes.c_uint32),
                ("vgprs", ctypes.c_uint32), ("agprs", ctypes.c_uint32), ("sgprs", ctypes.c_uint32),
                ("lds_bytes", ctypes.c_uint32), ("scratch_bytes", ctypes.c_uint32),
                ("block_size", ctypes.c_uint32), ("grid_size", ctypes.c_uint32),
                ("from_cache", ctypes.c_uint32), ("segments", ctypes.c_uint32),
                ("converge_batch", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("source_hash", ctypes.c_uint64),
                ("name", ctypes.c_char*64)]


# Every symbol include/gf_hip.h declares: (name, restype, argtypes).
_P, _S, _U64, _U32, _I = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
SYMBOLS = [
    ("gfhip_max_concurrency", _I, []),
    ("gfhip_device_type", ctypes.c_char_p, []),
    ("gfhip_shard_bounds", _I, [_S, _S, _S, ctypes.POINTER(_S), ctypes.POINTER(_S)]),
    ("gfhip_create_context", _P, [_I, _P]),
    ("gfhip_destroy_context", None, [_P]),
    ("gfhip_last_error", ctypes.c_char_p, [_P]),
    ("gfhip_add_kernel", _P, [_P, _P, _S, _S]),
    ("gfhip_compile", _I, [_P]),
    ("gfhip_create_kernel_call", _I, [_P, _P, _P, _P, _P]),
    ("gfhip_run", _I, [_P, _U32]),
    ("gfhip_run_max", _I, [_P, ctypes.POINTER(ctypes.c_double)]),
    ("gfhip_run_max_complex", _I, [_P, ctypes.POINTER(ctypes.c_double)]),
    ("gfhip_reduce_max", _I, [_P, _U64, ctypes.POINTER(ctypes.c_double)]),
    ("gfhip_converge", _I, [_P, ctypes.c_double, _S, ctypes.POINTER(_S), ctypes.POINTER(ctypes.c_double)]),
    ("gfhip_converge_per_ray", _I, [_P, ctypes.c_double, _S, ctypes.POINTER(_S), ctypes.POINTER(ctypes.c_double)]),
    ("gfhip_wait", _I, [_P]),
    ("gfhip_get_flags", _I, [_P, ctypes.POINTER(ctypes.c_uint32)]),
    ("gfhip_copy_to_device", _I, [_P, _U64, _P]),
    ("gfhip_copy_to_host", _I, [_P, _U64, _P]),
    ("gfhip_check_value", _I, [_P, _U64, _S, ctypes.POINTER(ctypes.c_double)]),
    ("gfhip_read_element", _I, [_P, _U64, _S, _P]),
    ("gfhip_set_random_state", _I, [_P, _U64, _P, _S]),
    ("gfhip_get_buffer", _P, [_P, _U64, ctypes.POINTER(_S)]),
    ("gfhip_allocate_buffer", _I, [_P, _U64, _S, _U32]),
    ("gfhip_get_buffer_info", _I, [_P, _U64, ctypes.POINTER(_S), ctypes.POINTER(_U32)]),
    ("gfhip_get_host_buffer", _P, [_P, _U64, ctypes.POINTER(_S)]),
    ("gfhip_set_buffer", _I, [_P, _U64, _P, _S, _U32]),
    ("gfhip_kernel_get_info", _I, [_P, ctypes.POINTER(KernelInfo)]),
    ("gfhip_generate_source", _P, [_P, _S, ctypes.POINTER(_U64)]),
    ("gfhip_generate_piece_source", _I, [_P, _S, _U32, ctypes.POINTER(_P), ctypes.POINTER(_U64)]),
    ("gfhip_export_piece", _I, [_P, _S, _U32, ctypes.POINTER(_P), ctypes.POINTER(_S)]),
    ("gfhip_free_string", None, [_P]),
    ("gfhip_cli_distribution", None, [_U64, _S, _P, _P, _P]),
    ("gfhip_enable_timing", _I, [_P, _I]),
    ("gfhip_kernel_timing", _I, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_U64)]),
    ("gfhip_kernel_timing_samples", _I, [_P, ctypes.POINTER(ctypes.c_double), _S, ctypes.POINTER(_S)]),
]

_lib = None


def load():
    """Load libgf_hip.so and declare every entry point.  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and os.path.basename(LIB_PATH) == "libgf_hip.so":
#  A tree that was checked out without its built artefacts: build the HIP library now if the
#  toolchain is here (seconds).  This is the product's own build, not a fallback path.
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            if os.path.exists(hipcc) and os.path.isdir(os.path.join(HERE, "csrc")):
                import subprocess
                try:
                    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "-s", "HIPCC=" + hipcc])
                except (OSError, subprocess.CalledProcessError):
                    pass
        if not os.path.exists(LIB_PATH):
            raise ImportError("graph_framework_amd: %s is missing — run __graft_entry__.build() "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
#  torch bundles its own HIP runtime; load it first so that the process has ONE libamdhip64
#  (two runtimes in one process cannot both open the device).  torch is only plumbing here.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib
