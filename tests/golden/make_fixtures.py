#!/usr/bin/env python3
"""Regenerate the .npz fixtures from the reference's own NetCDF-4 test data.

Inputs (data files the reference's tests hold, copied unchanged):
    tests/golden/efit.nc        <- graph_tests/efit.nc       (EFIT spline tables)
    tests/golden/efit_gold.nc   <- graph_tests/efit_gold.nc  (Mathematica gold fields,
                                    used by graph_tests/efit_test.cpp:132-187)

NetCDF-4 files are HDF5 files; they are read here through libhdf5 with ctypes
(no netCDF4 / h5py in the image).  Outputs are plain numpy archives so that
tests need neither HDF5 nor the reference checkout:
    tests/golden/efit_tables.npz, tests/golden/efit_gold.npz

Run:  python tests/golden/make_fixtures.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

_CANDIDATES = ["libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so"]


def _load_hdf5():
    for name in _CANDIDATES:
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libhdf5 not found (tried %s)" % _CANDIDATES)


class H5File:
    """Minimal read-only HDF5 access: whole datasets as float64 arrays."""

    def __init__(self, path):
        self.lib = _load_hdf5()
        lib = self.lib
        hid = ctypes.c_int64
        lib.H5open.restype = ctypes.c_int
        lib.H5Fopen.restype = hid
        lib.H5Fopen.argtypes = [ctypes.c_char_p, ctypes.c_uint, hid]
        lib.H5Dopen2.restype = hid
        lib.H5Dopen2.argtypes = [hid, ctypes.c_char_p, hid]
        lib.H5Dget_space.restype = hid
        lib.H5Dget_space.argtypes = [hid]
        lib.H5Sget_simple_extent_ndims.argtypes = [hid]
        lib.H5Sget_simple_extent_dims.argtypes = [hid, ctypes.c_void_p, ctypes.c_void_p]
        lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, ctypes.c_void_p]
        lib.H5Dclose.argtypes = [hid]
        lib.H5Sclose.argtypes = [hid]
        lib.H5Fclose.argtypes = [hid]
        lib.H5open()
        self.native_double = hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
        self.fid = lib.H5Fopen(path.encode(), 0, 0)  # H5F_ACC_RDONLY, H5P_DEFAULT
        if self.fid < 0:
            raise IOError("cannot open %s" % path)

    def read(self, name):
        lib = self.lib
        did = lib.H5Dopen2(self.fid, name.encode(), 0)
        if did < 0:
            raise KeyError(name)
        sid = lib.H5Dget_space(did)
        ndims = lib.H5Sget_simple_extent_ndims(sid)
        dims = (ctypes.c_uint64 * max(ndims, 1))()
        if ndims > 0:
            lib.H5Sget_simple_extent_dims(sid, dims, None)
        shape = tuple(int(dims[i]) for i in range(ndims))
        out = np.empty(shape, dtype=np.float64)
        status = lib.H5Dread(did, self.native_double, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p))
        lib.H5Sclose(sid)
        lib.H5Dclose(did)
        if status < 0:
            raise IOError("H5Dread failed for %s" % name)
        return out

    def close(self):
        self.lib.H5Fclose(self.fid)


def read_efit(path):
    f = H5File(path)
    out = {}
    for name in ("rmin", "dr", "zmin", "dz", "psimin", "dpsi", "pres_scale", "ne_scale", "te_scale"):
        out[name] = f.read(name)
    for a in range(4):
        for b in range(4):
            out["psi_c%d%d" % (a, b)] = f.read("psi_c%d%d" % (a, b))
        for prefix in ("fpol", "pressure", "te", "ne"):
            out["%s_c%d" % (prefix, a)] = f.read("%s_c%d" % (prefix, a))
    f.close()
    return out


def read_gold(path):
    f = H5File(path)
    out = {name: f.read(name) for name in
           ("r_grid", "z_grid", "bx_grid", "by_grid", "bz_grid", "pressure_grid", "ne_grid", "te_grid")}
    f.close()
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "efit_tables.npz"), **read_efit(os.path.join(HERE, "efit.nc")))
    np.savez_compressed(os.path.join(HERE, "efit_gold.npz"), **read_gold(os.path.join(HERE, "efit_gold.nc")))
    print("wrote efit_tables.npz, efit_gold.npz")
    if os.path.exists("/root/reference/graph_tests/test_erfi.nc"):
        erfi_fixture()
        print("wrote test_erfi.npz")


def erfi_fixture(reference="/root/reference"):
    """graph_tests/test_erfi.nc (the reference-held fixture of special::erfi: a 16 x 16 grid of
    z = x + iy on [-10, 10]^2 with erfi(z) = re + i img) as tests/golden/test_erfi.npz."""
    f = H5File(os.path.join(reference, "graph_tests", "test_erfi.nc"))
    np.savez_compressed(os.path.join(HERE, "test_erfi.npz"), x=f.read("x"), y=f.read("y"), re=f.read("re"), img=f.read("img"))
    f.close()


if __name__ == "__main__":
    sys.exit(main())
