"""Python mirror of the reference's backend-context interface over libgf_hip.so.

`Context` has the members jit::context forwards to its backend
(graph_framework/jit.hpp:87-338); `Kernel` is the closure
create_kernel_call/create_max_call return (cuda_context.hpp:316-576).  All
computation happens in the HIP library; numpy/torch only carry memory.
"""
import ctypes
import zlib

import numpy as np

from . import _lib

_NP = {_lib.GFIR_F32: np.float32, _lib.GFIR_F64: np.float64, _lib.GFIR_C32: np.complex64, _lib.GFIR_C64: np.complex128}


def key_of(name):
    """Buffer key for a variable name.  The reference keys buffers by leaf_node*
    (cuda_context.hpp:78-80); hosts without node objects key them by name."""
    if isinstance(name, int):
        return name
    data = name.encode()
    return (zlib.crc32(data) << 32) | zlib.adler32(data)


class GfHipError(RuntimeError):
    pass


class Context:
    """One device + one stream, as gpu::cuda_context(index) (cuda_context.hpp:137-148)."""

    def __init__(self, index=0, stream=None):
        self.lib = _lib.load()
        self.handle = self.lib.gfhip_create_context(int(index), ctypes.c_void_p(stream) if stream else None)
        if not self.handle:
            raise GfHipError(self.lib.gfhip_last_error(None).decode())
        self.index = index
        self.kernels = []
        self._keepalive = []

    @staticmethod
    def max_concurrency():
        return _lib.load().gfhip_max_concurrency()

    @staticmethod
    def device_type():
        return _lib.load().gfhip_device_type().decode()

    def close(self):
        if self.handle:
            self.lib.gfhip_destroy_context(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status):
        if status:
            raise GfHipError(self.lib.gfhip_last_error(self.handle).decode())

    def add_kernel(self, gfir, num_rays):
        """jit::context::add_kernel (jit.hpp:118-194) for a serialized work item."""
        if not isinstance(gfir, (bytes, bytearray)):
            with open(gfir, "rb") as f:
                gfir = f.read()
        handle = self.lib.gfhip_add_kernel(self.handle, gfir, len(gfir), int(num_rays))
        if not handle:
            raise GfHipError(self.lib.gfhip_last_error(self.handle).decode())
        kernel = Kernel(self, handle, int(num_rays))
        self.kernels.append(kernel)
        return kernel

    def compile(self):
        """jit::context::compile (jit.hpp:238-244)."""
        self._check(self.lib.gfhip_compile(self.handle))

    def wait(self):
        self._check(self.lib.gfhip_wait(self.handle))

    def flags(self):
        """Status bits raised by kernels (bit 0: fast-division window left), after a drain."""
        value = ctypes.c_uint32()
        self._check(self.lib.gfhip_get_flags(self.handle, ctypes.byref(value)))
        return value.value

    def buffer_info(self, key):
        """(element count, numpy dtype) of a buffer."""
        count, dtype = ctypes.c_size_t(), ctypes.c_uint32()
        self._check(self.lib.gfhip_get_buffer_info(self.handle, key_of(key), ctypes.byref(count), ctypes.byref(dtype)))
        return count.value, _NP[dtype.value]

    def copy_to_device(self, key, host):
        """Whole-buffer H2D copy (jit.hpp:315-320).  The C side copies the buffer's own size,
        so the host array is converted to the buffer's dtype and must hold that many elements."""
        count, dtype = self.buffer_info(key)
        host = np.ascontiguousarray(host, dtype=dtype)
        if host.size < count:
            raise ValueError("copy_to_device(%r): %d host elements for a buffer of %d" % (key, host.size, count))
        self._check(self.lib.gfhip_copy_to_device(self.handle, key_of(key), host.ctypes.data))

    def copy_to_host(self, key, host):
        """Whole-buffer D2H copy (jit.hpp:324-329) into `host`: a C-contiguous numpy array of the
        buffer's dtype with room for the whole buffer."""
        count, dtype = self.buffer_info(key)
        if not isinstance(host, np.ndarray) or host.dtype != dtype or not host.flags["C_CONTIGUOUS"] or host.size < count:
            raise ValueError("copy_to_host(%r): need a contiguous %s array of at least %d elements"
                             % (key, np.dtype(dtype).name, count))
        self._check(self.lib.gfhip_copy_to_host(self.handle, key_of(key), host.ctypes.data))
        return host

    def get_host_buffer(self, key):
        """jit::context::get_buffer (jit.hpp:336): a pinned host mirror of the buffer as a numpy
        view, refreshed by every wait()."""
        count = ctypes.c_size_t()
        pointer = self.lib.gfhip_get_host_buffer(self.handle, key_of(key), ctypes.byref(count))
        if not pointer:
            raise GfHipError(self.lib.gfhip_last_error(self.handle).decode())
        _, dtype = self.buffer_info(key)
        base = ctypes.c_double if dtype in (np.float64, np.complex128) else ctypes.c_float
        parts = 2 if dtype in (np.complex64, np.complex128) else 1
        flat = np.ctypeslib.as_array(ctypes.cast(pointer, ctypes.POINTER(base)), shape=(count.value*parts,))
        return flat.view(dtype) if parts == 2 else flat

    def check_value(self, index, key):
        value = ctypes.c_double()
        self._check(self.lib.gfhip_check_value(self.handle, key_of(key), int(index), ctypes.byref(value)))
        return value.value

    def get_buffer(self, key):
        """(device pointer, element count) of a buffer."""
        count = ctypes.c_size_t()
        pointer = self.lib.gfhip_get_buffer(self.handle, key_of(key), ctypes.byref(count))
        if not pointer:
            raise GfHipError(self.lib.gfhip_last_error(self.handle).decode())
        return pointer, count.value

    def set_buffer(self, key, tensor):
        """Adopt a contiguous torch CUDA tensor (float32/float64) as the buffer for `key`."""
        import torch
        assert tensor.is_cuda and tensor.is_contiguous()
        dtype = {torch.float32: _lib.GFIR_F32, torch.float64: _lib.GFIR_F64}[tensor.dtype]
        self._keepalive.append(tensor)
        self._check(self.lib.gfhip_set_buffer(self.handle, key_of(key), tensor.data_ptr(), tensor.numel(), dtype))

    def enable_timing(self, enable=True, every=1):
        """HIP events around every `every`-th launch of each kernel (see Kernel.timing)."""
        self._check(self.lib.gfhip_enable_timing(self.handle, int(every) if enable else 0))


class Kernel:
    def __init__(self, context, handle, num_rays):
        self.context = context
        self.lib = context.lib
        self.handle = handle
        self.num_rays = num_rays

    def info(self):
        info = _lib.KernelInfo()
        self.context._check(self.lib.gfhip_kernel_get_info(self.handle, ctypes.byref(info)))
        return info

    @property
    def np_dtype(self):
        return _NP[self.info().dtype]

    def create_kernel_call(self, input_keys, output_keys, input_init=None):
        """create_kernel_call (cuda_context.hpp:316-531): bind buffers, allocating and
        uploading `input_init[i]` (numpy array or None) on first sight of a key."""
        info = self.info()
        assert len(input_keys) == info.num_inputs and len(output_keys) == info.num_outputs
        in_keys = (ctypes.c_uint64*max(len(input_keys), 1))(*[key_of(k) for k in input_keys])
        out_keys = (ctypes.c_uint64*max(len(output_keys), 1))(*[key_of(k) for k in output_keys])
        init = (ctypes.c_void_p*max(len(input_keys), 1))()
        counts = (ctypes.c_size_t*max(len(input_keys), 1))()
        keep = []
        for i, key in enumerate(input_keys):
            value = None if input_init is None else input_init[i]
            if value is not None:
                value = np.ascontiguousarray(value, dtype=_NP[info.dtype])
                if value.size < self.num_rays:
                    raise ValueError("initial values of %r: %d elements for %d rays" % (key, value.size, self.num_rays))
                keep.append(value)
                init[i] = value.ctypes.data
#  The value's own length: an input that index_1D/2D nodes read may be longer than the ensemble
#  (hip_context passes buffer.size() the same way); the C side allocates max(num_rays, indexed
#  length, this) and rejects a value shorter than what the index nodes address.
                counts[i] = value.size
        self.context._check(self.lib.gfhip_create_kernel_call(self.handle, in_keys, init, counts, out_keys))

    def set_random_state(self, key, states):
        """Bind the MT19937 states of the item's random_state node (random.hpp:24-130): the bytes of
        1024 mt_state structures, uploaded on first sight of `key`."""
        states = np.ascontiguousarray(states, dtype=np.uint8)
        self.context._check(self.lib.gfhip_set_random_state(self.handle, key_of(key), states.ctypes.data, states.size))

    def run_max_complex(self):
        value = (ctypes.c_double*2)()
        self.context._check(self.lib.gfhip_run_max_complex(self.handle, value))
        return complex(value[0], value[1])

    def run(self, steps=1):
        self.context._check(self.lib.gfhip_run(self.handle, int(steps)))

    def run_max(self):
        value = ctypes.c_double()
        self.context._check(self.lib.gfhip_run_max(self.handle, ctypes.byref(value)))
        return value.value

    def converge(self, tolerance=1.0e-30, max_iterations=1000):
        """workflow::converge_item::run (workflow.hpp:179-205).  Returns (iterations, last max)."""
        iterations = ctypes.c_size_t()
        last = ctypes.c_double()
        self.context._check(self.lib.gfhip_converge(self.handle, float(tolerance), int(max_iterations),
                                                    ctypes.byref(iterations), ctypes.byref(last)))
        return iterations.value, last.value

    def converge_per_ray(self, tolerance=1.0e-30, max_iterations=1000):
        """The converge loop per ray inside one launch (wavefront ballot exit).
        Returns (max iterations over rays, max final residual)."""
        iterations = ctypes.c_size_t()
        last = ctypes.c_double()
        self.context._check(self.lib.gfhip_converge_per_ray(self.handle, float(tolerance), int(max_iterations),
                                                            ctypes.byref(iterations), ctypes.byref(last)))
        return iterations.value, last.value

    def timing_samples(self):
        """Durations (ms) of the sampled launches since the last timing call, in launch order."""
        count = ctypes.c_size_t()
        self.context._check(self.lib.gfhip_kernel_timing_samples(self.handle, None, 0, ctypes.byref(count)))
        values = (ctypes.c_double*max(count.value, 1))()
        self.context._check(self.lib.gfhip_kernel_timing_samples(self.handle, values, count.value, ctypes.byref(count)))
        return [values[i] for i in range(count.value)]

    def timing(self):
        """(average launch ms, launches) since the last call; needs Context.enable_timing()."""
        ms = ctypes.c_double()
        launches = ctypes.c_uint64()
        self.context._check(self.lib.gfhip_kernel_timing(self.handle, ctypes.byref(ms), ctypes.byref(launches)))
        return ms.value, launches.value


def generate_source(gfir):
    """HIP source and cache hash of a serialized work item (no device needed)."""
    lib = _lib.load()
    if not isinstance(gfir, (bytes, bytearray)):
        with open(gfir, "rb") as f:
            gfir = f.read()
    source_hash = ctypes.c_uint64()
    text = lib.gfhip_generate_source(gfir, len(gfir), ctypes.byref(source_hash))
    if not text:
        raise GfHipError(lib.gfhip_last_error(None).decode())
    source = ctypes.string_at(text).decode()
    lib.gfhip_free_string(text)
    return source, source_hash.value


def generate_piece_sources(gfir):
    """[(HIP source, cache hash)] of the kernels an item runs as: one for most items, one per
    segment for items the lowering cuts into segments (csrc/segments.hpp)."""
    lib = _lib.load()
    if not isinstance(gfir, (bytes, bytearray)):
        with open(gfir, "rb") as f:
            gfir = f.read()
    pieces = []
    while True:
        text, source_hash = ctypes.c_void_p(), ctypes.c_uint64()
        if lib.gfhip_generate_piece_source(gfir, len(gfir), len(pieces), ctypes.byref(text), ctypes.byref(source_hash)):
            raise GfHipError(lib.gfhip_last_error(None).decode())
        if not text:
            return pieces
        pieces.append((ctypes.string_at(text).decode(), source_hash.value))
        lib.gfhip_free_string(text)


def export_pieces(gfir):
    """The segments of an item as data (no device): a list of dicts with the piece as GFIR bytes
    and what its symbols and outputs are (include/gf_hip.h, gfhip_export_piece); [] for an item
    that runs as one kernel."""
    import numpy as np
    lib = _lib.load()
    if not isinstance(gfir, (bytes, bytearray)):
        with open(gfir, "rb") as f:
            gfir = f.read()
    pieces = []
    while True:
        block, size = ctypes.c_void_p(), ctypes.c_size_t()
        if lib.gfhip_export_piece(gfir, len(gfir), len(pieces), ctypes.byref(block), ctypes.byref(size)):
            raise GfHipError(lib.gfhip_last_error(None).decode())
        if not block:
            return pieces
        data = ctypes.string_at(block, size.value)
        lib.gfhip_free_string(block)
        symbols, outputs, slots, count = np.frombuffer(data, dtype="<i4", count=4)
        words = np.frombuffer(data, dtype="<i4", count=4 + 2*symbols + 2*outputs)
        at = 4
        fields = {}
        for name, length in (("symbol_state", symbols), ("symbol_slot", symbols), ("output_slot", outputs), ("output_original", outputs)):
            fields[name] = [int(w) for w in words[at:at + length]]
            at += length
        pieces.append(dict(fields, slots=int(slots), pieces=int(count), gfir=data[4*at:]))
