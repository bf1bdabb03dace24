"""ctypes binding of oracle/libgfir_interp.so — the CPU ORACLE that executes GFIR work items.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
The product package never imports this module.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgfir_interp.so")

_lib = None


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("gfir_interp.c", "Makefile")] + [os.path.join(HERE, "..", "include", "gfir.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-s", "libgfir_interp.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.gfi_load.restype = ctypes.c_void_p
        _lib.gfi_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        _lib.gfi_free.argtypes = [ctypes.c_void_p]
        _lib.gfi_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.gfi_setter_inputs.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.gfi_flags.restype = ctypes.c_uint32
        _lib.gfi_flags.argtypes = [ctypes.c_void_p]
        _lib.gfi_random_states.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        _lib.gfi_erfi.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        for s in ("f64", "f32"):
            getattr(_lib, "gfi_run_" + s).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_size_t, ctypes.c_size_t]
            getattr(_lib, "gfi_run_generic_" + s).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                              ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
            f = getattr(_lib, "gfi_run_threads_" + s)
            f.restype = ctypes.c_double
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                          ctypes.c_size_t, ctypes.c_size_t]
    return _lib


def erfi(z):
    """erfi of a complex number as the oracle (and the device) evaluates it."""
    out = (ctypes.c_double*2)()
    lib().gfi_erfi(float(z.real), float(z.imag), out)
    return complex(out[0], out[1])


class Item:
    """One work item executed on the CPU exactly as gpu::cpu_context's kernel loop would."""

    def __init__(self, source):
        if isinstance(source, (bytes, bytearray)):
            data = bytes(source)
        else:
            with open(source, "rb") as f:
                data = f.read()
        self.handle = lib().gfi_load(data, len(data))
        if not self.handle:
            raise ValueError("not a GFIR work item")
        info = (ctypes.c_uint32*6)()
        lib().gfi_info(self.handle, info)
        self.dtype = {0: "f32", 1: "f64", 2: "c32", 3: "c64"}[info[0]]
        self.base = "f64" if info[0] in (1, 3) else "f32"
        self.np_dtype = {0: np.float32, 1: np.float64, 2: np.complex64, 3: np.complex128}[info[0]]
        flags = lib().gfi_flags(self.handle)
        self.safe_math, self.has_random = bool(flags & 1), bool(flags & 0x100)
        self.generic = info[0] >= 2 or self.safe_math or self.has_random
        self.random_states = None
        self.num_inputs, self.num_outputs, self.num_setters = info[1], info[2], info[3]
        self.num_tables, self.num_instructions = info[4], info[5]
        si = (ctypes.c_uint32*max(self.num_setters, 1))()
        lib().gfi_setter_inputs(self.handle, si)
        self.setter_inputs = [si[i] for i in range(self.num_setters)]

    def __del__(self):
        try:
            lib().gfi_free(self.handle)
        except Exception:
            pass

    def _pointers(self, arrays):
        ptrs = (ctypes.c_void_p*max(len(arrays), 1))()
        for i, a in enumerate(arrays):
            assert a.dtype == self.np_dtype and a.flags["C_CONTIGUOUS"]
            ptrs[i] = a.ctypes.data
        return ptrs

    def run(self, columns, steps=1, threads=1):
        """columns: list of num_inputs arrays (updated in place by setters).
        Returns (list of output arrays, wall seconds)."""
        assert len(columns) == self.num_inputs
        if self.generic:
            return self._run_generic(columns, steps)
        n = columns[0].size
        outs = [np.empty(n, dtype=self.np_dtype) for _ in range(self.num_outputs)]
        fn = getattr(lib(), "gfi_run_threads_" + self.dtype)
        secs = fn(self.handle, self._pointers(columns), self._pointers(outs), n, steps, threads)
        return outs, secs

    STATE_BYTES = 2500                  # sizeof(mt_state): 624 words + a 16-bit index, 4-byte aligned (random.hpp:44-52)

    def seed(self, seed=0):
        """random_state_node(1024, seed): the states a kernel with draws works on (returned as the
        bytes hip_context uploads; this item keeps and advances its own copy)."""
        self.random_states = np.zeros(1024*self.STATE_BYTES, dtype=np.uint8)
        lib().gfi_random_states(self.random_states.ctypes.data, 1024, int(seed))
        return self.random_states.copy()

    def _run_generic(self, columns, steps, size=None):
        """Complex / SAFE_MATH / random items: serial loop, elements in order.  Inputs that index
        nodes read may be longer than the ensemble; `size` (default: the shortest input, or the
        first argument of run_sized) is the number of elements the kernel runs over."""
        import time
        n = size if size is not None else (min(c.size for c in columns) if columns else 1)
        outs = [np.zeros(n, dtype=self.np_dtype) for _ in range(self.num_outputs)]
        if self.has_random and self.random_states is None:
            self.seed(0)
        states = self.random_states.ctypes.data if self.random_states is not None else None
        start = time.perf_counter()
        fn = getattr(lib(), "gfi_run_generic_" + self.base)
        for _ in range(steps):
            fn(self.handle, self._pointers(columns), self._pointers(outs), 0, n, states)
        return outs, time.perf_counter() - start

    def run_sized(self, size, columns, steps=1):
        """run() over `size` elements (items without inputs, or with indexed inputs of other lengths)."""
        assert self.generic or all(c.size >= size for c in columns)
        if self.generic:
            return self._run_generic(columns, steps, size)
        return self.run([c[:size] for c in columns], steps)

    def converge(self, columns, tolerance=1.0e-30, max_iterations=1000):
        """workflow::converge_item::run (workflow.hpp:179-205): repeat the kernel until the
        max of its last output stalls.  Returns (iterations, last max, outputs)."""
        def max_kernel():
            outs, _ = self.run(columns)
            values = outs[-1]
            if np.iscomplexobj(values):
#  complex items: the element of largest modulus, the first of equals (cpu_context.hpp:314-318)
                return complex(values[int(np.argmax(np.abs(values)))]), outs
#  std::max_element (cpu_context.hpp:306-322): `m < x` is false for NaN, so a NaN is never
#  selected unless it is the first element.
            if np.isnan(values[0]):
                return float("nan"), outs
            return float(np.fmax.reduce(values)), outs

        big = float(np.finfo(self.np_dtype).max)
        iterations = 0
        max_residual, outs = max_kernel()
        if isinstance(max_residual, complex):
            big = 0.0j                               # std::numeric_limits<std::complex<T>>::max() is T()
        last_max = big
        off_last_max = big
        while (abs(max_residual) > abs(tolerance) and abs(last_max - max_residual) > abs(tolerance)
               and abs(off_last_max - max_residual) > abs(tolerance)):
            took = iterations < max_iterations
            iterations += 1
            if not took:
                break
            last_max = max_residual
            if not iterations % 2:
                off_last_max = max_residual
            max_residual, outs = max_kernel()
        return iterations, max_residual, outs
