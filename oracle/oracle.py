"""ctypes binding of the CPU ORACLE (oracle/libgf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package graph_framework_amd never
imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgf_oracle.so")

_DT = {"f64": (np.float64, ctypes.c_double), "f32": (np.float32, ctypes.c_float)}


def build(force=False):
    """Compile libgf_oracle.so with the committed Makefile (gcc, seconds)."""
    src_time = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("gf_oracle.cpp", "gf_oracle.hpp", "Makefile"))
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < src_time:
        subprocess.check_call(["make", "-C", HERE, "-s", "libgf_oracle.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


SCALAR_NAMES = ("rmin", "dr", "zmin", "dz", "psimin", "dpsi", "ne_scale", "te_scale", "pres_scale")


def pack_tables(tables):
    """tables: mapping with the efit.nc dataset names -> the flat arrays the C side takes."""
    scalars = np.array([float(tables[k]) for k in SCALAR_NAMES], dtype=np.float64)
    psi = np.ascontiguousarray(np.stack([tables["psi_c%d%d" % (a, b)] for a in range(4) for b in range(4)]),
                               dtype=np.float64)
    numr, numz = tables["psi_c00"].shape

    def four(prefix):
        return np.ascontiguousarray(np.stack([tables["%s_c%d" % (prefix, k)] for k in range(4)]), dtype=np.float64)

    te, ne, pres, fpol = four("te"), four("ne"), four("pressure"), four("fpol")
    return scalars, numr, numz, te.shape[1], psi, te, ne, pres, fpol


class Efit:
    """Oracle-side EFIT equilibrium (gfo::efit<T>)."""

    def __init__(self, tables, dtype="f64"):
        self.dtype = dtype
        self.np_dtype, self.c_type = _DT[dtype]
        scalars, numr, numz, numpsi, psi, te, ne, pres, fpol = pack_tables(tables)
        create = getattr(lib(), "gfo_efit_create_" + dtype)
        create.restype = ctypes.c_void_p
        create.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_void_p]*5
        self.handle = ctypes.c_void_p(create(_ptr(scalars), numr, numz, numpsi,
                                             _ptr(psi), _ptr(te), _ptr(ne), _ptr(pres), _ptr(fpol)))

    def _fn(self, name, restype=None):
        f = getattr(lib(), "%s_%s" % (name, self.dtype))
        f.restype = restype
        return f

    def __del__(self):
        try:
            self._fn("gfo_efit_destroy")(self.handle)
        except Exception:
            pass

    def _arr(self, a):
        return np.ascontiguousarray(a, dtype=self.np_dtype)

    def test_kernel(self, x, y, z):
        """efit_test.cpp `test_kernel`: returns bx, by, bz, ne, te, div."""
        x, y, z = self._arr(x), self._arr(y), self._arr(z)
        outs = [np.empty_like(x) for _ in range(6)]
        self._fn("gfo_efit_test_kernel")(self.handle, ctypes.c_size_t(x.size), _ptr(x), _ptr(y), _ptr(z),
                                         *[_ptr(o) for o in outs])
        return outs

    def cold_plasma_D(self, w, kx, ky, kz, x, y, z):
        """Returns D[n] and dD[7][n] (order w, kx, ky, kz, x, y, z)."""
        args = [self._arr(a) for a in (w, kx, ky, kz, x, y, z)]
        n = args[0].size
        D = np.empty(n, dtype=self.np_dtype)
        dD = np.empty((7, n), dtype=self.np_dtype)
        self._fn("gfo_cold_plasma_D")(self.handle, ctypes.c_size_t(n), *[_ptr(a) for a in args], _ptr(D), _ptr(dD))
        return D, dD

    def loss_kernel(self, state, var=1, step=1.0):
        """One Newton iteration in place.  state: dict t,w,x,y,z,kx,ky,kz. Returns residual D*D."""
        n = state["x"].size
        res = np.empty(n, dtype=self.np_dtype)
        self._fn("gfo_loss_kernel")(self.handle, ctypes.c_size_t(n),
                                    *[_ptr(state[k]) for k in ("t", "w", "x", "y", "z", "kx", "ky", "kz")],
                                    _ptr(res), ctypes.c_int(var), self.c_type(step))
        return res

    def newton_solve(self, state, var=1, step=1.0, tolerance=1.0e-30, max_iterations=1000):
        """converge_item::run.  Returns (iterations, last max residual, residual array)."""
        n = state["x"].size
        res = np.empty(n, dtype=self.np_dtype)
        last = self.c_type(0)
        it = self._fn("gfo_newton_solve", ctypes.c_size_t)(
            self.handle, ctypes.c_size_t(n),
            *[_ptr(state[k]) for k in ("t", "w", "x", "y", "z", "kx", "ky", "kz")],
            _ptr(res), ctypes.c_int(var), self.c_type(step), self.c_type(tolerance),
            ctypes.c_size_t(max_iterations), ctypes.byref(last))
        return it, last.value, res

    def rk4_steps(self, state, dt, num_steps=1, threads=1):
        """num_steps `solver_kernel` passes in place.  Returns (residual, wall seconds)."""
        n = state["x"].size
        res = np.empty(n, dtype=self.np_dtype)
        secs = self._fn("gfo_rk4_steps", ctypes.c_double)(
            self.handle, ctypes.c_size_t(n),
            *[_ptr(state[k]) for k in ("t", "w", "x", "y", "z", "kx", "ky", "kz")],
            _ptr(res), self.c_type(dt), ctypes.c_size_t(num_steps), ctypes.c_size_t(threads))
        return res, secs

    def characteristic_field(self):
        it = ctypes.c_size_t(0)
        b0 = self._fn("gfo_characteristic_field", self.c_type)(self.handle, ctypes.byref(it))
        return b0, it.value

    def korc_constants(self, b0):
        larmor, dt = self.c_type(0), self.c_type(0)
        self._fn("gfo_korc_constants")(self.c_type(b0), ctypes.byref(larmor), ctypes.byref(dt))
        return larmor.value, dt.value

    def korc_initialize_gamma(self, p):
        n = p["ux"].size
        self._fn("gfo_korc_initialize_gamma")(ctypes.c_size_t(n), *[_ptr(p[k]) for k in ("ux", "uy", "uz", "gamma")])

    def korc_steps(self, p, b0, num_steps=1, threads=1):
        n = p["x"].size
        return self._fn("gfo_korc_steps", ctypes.c_double)(
            self.handle, self.c_type(b0), ctypes.c_size_t(n),
            *[_ptr(p[k]) for k in ("x", "y", "z", "ux", "uy", "uz", "gamma")],
            ctypes.c_size_t(num_steps), ctypes.c_size_t(threads))


def new_ray_state(n, dtype="f64", **values):
    """SoA ray state {t,w,x,y,z,kx,ky,kz}, each filled with a scalar or array."""
    np_dtype = _DT[dtype][0]
    s = {}
    for k in ("t", "w", "x", "y", "z", "kx", "ky", "kz"):
        s[k] = np.ascontiguousarray(np.broadcast_to(np.asarray(values.get(k, 0.0), dtype=np_dtype), (n,)).copy())
    return s


def new_particle_state(n, dtype="f64", **values):
    np_dtype = _DT[dtype][0]
    p = {}
    for k in ("x", "y", "z", "ux", "uy", "uz", "gamma"):
        p[k] = np.ascontiguousarray(np.broadcast_to(np.asarray(values.get(k, 0.0), dtype=np_dtype), (n,)).copy())
    return p
