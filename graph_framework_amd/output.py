"""Trajectory output: solver_interface::write_step + output::data_set.

Reference: `write_step` (graph_framework/solver.hpp:418-424) joins the previous writer thread,
waits for the device, then starts a thread that appends one record per variable to
`result<n>.nc` (`data_set::write`, output.hpp:354-400; variables created by
`create_variable`, output.hpp:260-273, shape (time, num_rays, ray_dim), time unlimited).

Here the record leaves the device without stalling the step loop:
  1. a device-side snapshot of the 9 arrays (D2D on the compute stream, ~15 us for 1e6 rays) so
     that later steps may overwrite the state;
  2. the D2H copy of the snapshot into pinned host memory on a second HIP stream, ordered after
     the snapshot by an event, overlapping the following RK4 launches;
  3. a writer thread that waits for that copy and appends the record to the file.
NetCDF-C is not in the image; the file is written through libhdf5 in NetCDF-4's on-disk
conventions, so that the reference's own readers (result_file(filename), output.hpp:77-78;
reference_variable, :218-232; bin_power, graph_driver/xrays.cpp:674) find what they look up:
  * the dimensions `time` (unlimited), `num_rays` and `ray_dim` (output.hpp:61-62, :189-197) as
    HDF5 dimension-scale datasets the way netCDF-C writes a dimension that has no coordinate
    variable (CLASS = "DIMENSION_SCALE", NAME = "This is a netCDF dimension but not a netCDF
    variable.<length>", _Netcdf4Dimid; cf. `numpsi` in graph_tests/efit.nc);
  * every variable (time, num_rays, ray_dim) with the three scales attached (DIMENSION_LIST /
    REFERENCE_LIST through H5DSattach_scale) and _Netcdf4Coordinates;
  * the variable `time` under netCDF-4's name for a variable that shares a dimension's name
    without being its coordinate variable, `_nc4_non_coord_time`;
  * the root attribute _NCProperties.
"""
import ctypes
import threading

import numpy as np

_CANDIDATES = ["libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so"]
_HL_CANDIDATES = ["libhdf5_hl.so", "/opt/conda/lib/libhdf5_hl.so.100", "/opt/conda/lib/libhdf5_hl.so"]
_H5S_SCALAR = 0
_H5T_STR_NULLTERM = 0
_NON_COORDINATE = "_nc4_non_coord_"
_DIMENSION_NAME = "This is a netCDF dimension but not a netCDF variable.%10d"
_H5F_ACC_TRUNC = 2
_H5F_ACC_RDWR = 1
_H5S_SELECT_SET = 0
_H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF


def _hdf5():
    for name in _CANDIDATES:
        try:
            lib = ctypes.CDLL(name)
            break
        except OSError:
            continue
    else:
        raise RuntimeError("libhdf5 not found (tried %s)" % _CANDIDATES)
    hid = ctypes.c_int64
    lib.H5open.restype = ctypes.c_int
    lib.H5Fcreate.restype = hid
    lib.H5Fcreate.argtypes = [ctypes.c_char_p, ctypes.c_uint, hid, hid]
    lib.H5Screate_simple.restype = hid
    lib.H5Screate_simple.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.H5Pcreate.restype = hid
    lib.H5Pcreate.argtypes = [hid]
    lib.H5Pset_chunk.argtypes = [hid, ctypes.c_int, ctypes.c_void_p]
    lib.H5Dcreate2.restype = hid
    lib.H5Dcreate2.argtypes = [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]
    lib.H5Dset_extent.argtypes = [hid, ctypes.c_void_p]
    lib.H5Dget_space.restype = hid
    lib.H5Dget_space.argtypes = [hid]
    lib.H5Sselect_hyperslab.argtypes = [hid, ctypes.c_int] + [ctypes.c_void_p]*4
    lib.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, ctypes.c_void_p]
    lib.H5Fflush.argtypes = [hid, ctypes.c_int]
    for name in ("H5Dclose", "H5Sclose", "H5Pclose", "H5Fclose", "H5Aclose", "H5Tclose"):
        getattr(lib, name).argtypes = [hid]
    lib.H5Screate.restype = hid
    lib.H5Screate.argtypes = [ctypes.c_int]
    lib.H5Tcopy.restype = hid
    lib.H5Tcopy.argtypes = [hid]
    lib.H5Tset_size.argtypes = [hid, ctypes.c_size_t]
    lib.H5Tset_strpad.argtypes = [hid, ctypes.c_int]
    lib.H5Acreate2.restype = hid
    lib.H5Acreate2.argtypes = [hid, ctypes.c_char_p, hid, hid, hid, hid]
    lib.H5Awrite.argtypes = [hid, hid, ctypes.c_void_p]
    lib.H5Fopen.restype = hid
    lib.H5Fopen.argtypes = [ctypes.c_char_p, ctypes.c_uint, hid]
    lib.H5Dopen2.restype = hid
    lib.H5Dopen2.argtypes = [hid, ctypes.c_char_p, hid]
    lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, ctypes.c_void_p]
    lib.H5Sget_simple_extent_dims.argtypes = [hid, ctypes.c_void_p, ctypes.c_void_p]
    lib.H5Lexists.argtypes = [hid, ctypes.c_char_p, hid]
    lib.H5open()
    for name in _HL_CANDIDATES:
        try:
            lib.hl = ctypes.CDLL(name)
            break
        except OSError:
            continue
    else:
        raise RuntimeError("libhdf5_hl not found (tried %s)" % _HL_CANDIDATES)
    lib.hl.H5DSset_scale.argtypes = [hid, ctypes.c_char_p]
    lib.hl.H5DSattach_scale.argtypes = [hid, hid, ctypes.c_uint]
    return lib


class ResultFile:
    """output::result_file + output::data_set (output.hpp:32-400): the dimensions time
    (unlimited), num_rays, ray_dim; variables of shape (time, num_rays, ray_dim); one record
    appended per write."""

    def __init__(self, path, num_rays=None, dtype=np.float64):
        """num_rays given: result_file(filename, num_rays), output.hpp:50-66 — a new file.
        num_rays None: result_file(filename), output.hpp:77-86 — an existing file opened for
        update (the absorption pass and bin_power add their variables to the trajectory file)."""
        self.lib = _hdf5()
        self.lock = threading.Lock()                         # output::sync (output.hpp:21): libhdf5 calls are serialised
        self.dtype = np.dtype(dtype)
        hid = ctypes.c_int64
        native = "H5T_NATIVE_DOUBLE_g" if self.dtype == np.float64 else "H5T_NATIVE_FLOAT_g"
        self.native = hid.in_dll(self.lib, native).value
        self.native_int = hid.in_dll(self.lib, "H5T_NATIVE_INT_g").value
        self.string = hid.in_dll(self.lib, "H5T_C_S1_g").value
        self.scale_type = hid.in_dll(self.lib, "H5T_IEEE_F32BE_g").value
        self.dataset_create = hid.in_dll(self.lib, "H5P_CLS_DATASET_CREATE_ID_g").value
        self.variables = {}
        self.parts = {}
        self.opened = {}
        if num_rays is None:
            self.file = self.lib.H5Fopen(path.encode(), _H5F_ACC_RDWR, 0)
            if self.file < 0:
                raise IOError("cannot open %s" % path)
            self.dimensions = [self._open_dataset(name) for name in ("time", "num_rays", "ray_dim")]
            self.records = self._extent(self.dimensions[0])[0]
            self.num_rays = self._extent(self.dimensions[1])[0]
            self.next_dimid = 3
            self.complex_dimension = None
            if self.lib.H5Lexists(self.file, b"ray_dim_cplx", 0) > 0:
                self.complex_dimension = self._open_dataset("ray_dim_cplx")
                self.next_dimid = 4
            return
        self.num_rays = int(num_rays)
        self.file = self.lib.H5Fcreate(path.encode(), _H5F_ACC_TRUNC, 0, 0)
        if self.file < 0:
            raise IOError("cannot create %s" % path)
        self.records = 0
        self._string_attribute(self.file, "_NCProperties", "version=2,graph_framework_amd=1,hdf5=1.10", None)
#  result_file's constructor (output.hpp:61-64) and data_set's (output.hpp:189-197).
        self.dimensions = [self._dimension("time", None, 0), self._dimension("num_rays", max(self.num_rays, 1), 1),
                           self._dimension("ray_dim", 1, 2)]
        self.next_dimid = 3
        self.complex_dimension = None

    def _open_dataset(self, name):
        dataset = self.lib.H5Dopen2(self.file, name.encode(), 0)
        if dataset < 0:
            raise IOError("no dataset %s in the file" % name)
        return dataset

    def _extent(self, dataset):
        space = self.lib.H5Dget_space(dataset)
        dims = (ctypes.c_uint64*3)()
        self.lib.H5Sget_simple_extent_dims(space, dims, None)
        self.lib.H5Sclose(space)
        return list(dims)

    def _string_attribute(self, where, name, text, size):
        data = text.encode()
        size = size or len(data) + 1
        kind = self.lib.H5Tcopy(self.string)
        self.lib.H5Tset_size(kind, size)
        self.lib.H5Tset_strpad(kind, _H5T_STR_NULLTERM)
        space = self.lib.H5Screate(_H5S_SCALAR)
        attribute = self.lib.H5Acreate2(where, name.encode(), kind, space, 0, 0)
        if attribute < 0:
            raise IOError("cannot create attribute %s" % name)
        buffer = ctypes.create_string_buffer(data, size)
        self.lib.H5Awrite(attribute, kind, buffer)
        self.lib.H5Aclose(attribute)
        self.lib.H5Sclose(space)
        self.lib.H5Tclose(kind)

    def _int_attribute(self, where, name, values):
        values = list(values)
        if len(values) == 1:
            space = self.lib.H5Screate(_H5S_SCALAR)
        else:
            space = self.lib.H5Screate_simple(1, (ctypes.c_uint64*1)(len(values)), None)
        attribute = self.lib.H5Acreate2(where, name.encode(), self.native_int, space, 0, 0)
        if attribute < 0:
            raise IOError("cannot create attribute %s" % name)
        self.lib.H5Awrite(attribute, self.native_int, (ctypes.c_int*len(values))(*values))
        self.lib.H5Aclose(attribute)
        self.lib.H5Sclose(space)

    def _dimension(self, name, length, dimid):
        """A netCDF-4 dimension without a coordinate variable: a dimension-scale dataset of the
        dimension's length (unlimited: extensible, extended with the records)."""
        plist = self.lib.H5Pcreate(self.dataset_create)
        if length is None:
            space = self.lib.H5Screate_simple(1, (ctypes.c_uint64*1)(0), (ctypes.c_uint64*1)(_H5S_UNLIMITED))
            self.lib.H5Pset_chunk(plist, 1, (ctypes.c_uint64*1)(1024))
        else:
            space = self.lib.H5Screate_simple(1, (ctypes.c_uint64*1)(length), None)
        dataset = self.lib.H5Dcreate2(self.file, name.encode(), self.scale_type, space, 0, plist, 0)
        self.lib.H5Pclose(plist)
        self.lib.H5Sclose(space)
        if dataset < 0:
            raise IOError("cannot create dimension %s" % name)
        if self.lib.hl.H5DSset_scale(dataset, None) < 0:            # CLASS = "DIMENSION_SCALE"
            raise IOError("H5DSset_scale failed for %s" % name)
        self._string_attribute(dataset, "NAME", _DIMENSION_NAME % (0 if length is None else length), 64)
        self._int_attribute(dataset, "_Netcdf4Dimid", [dimid])
        return dataset

    @staticmethod
    def _stored_name(name):
#  A variable that shares a dimension's name without being its coordinate variable.
        return _NON_COORDINATE + name if name in ("time", "num_rays", "ray_dim") else name

    def create_variable(self, name, parts=1):
        """data_set::create_variable (output.hpp:260-273): nc_def_var(name, type, {time, num_rays, ray_dim}).
        parts = 2: a variable of a complex data_set, whose last dimension is `ray_dim_cplx` of length 2
        (real, imaginary; output.hpp:215-224)."""
        last = self.dimensions[2]
        last_id = 2
        if parts == 2:
            if self.complex_dimension is None:
                self.complex_dimension = self._dimension("ray_dim_cplx", 2, self.next_dimid)
                self.next_dimid += 1
            last = self.complex_dimension
            last_id = self.next_dimid - 1
        dims = (ctypes.c_uint64*3)(self.records, self.num_rays, parts)
        maxdims = (ctypes.c_uint64*3)(_H5S_UNLIMITED, self.num_rays, parts)
        chunk = (ctypes.c_uint64*3)(1, max(self.num_rays, 1), parts)
        space = self.lib.H5Screate_simple(3, dims, maxdims)
        plist = self.lib.H5Pcreate(self.dataset_create)
        self.lib.H5Pset_chunk(plist, 3, chunk)
        dataset = self.lib.H5Dcreate2(self.file, self._stored_name(name).encode(), self.native, space, 0, plist, 0)
        self.lib.H5Pclose(plist)
        self.lib.H5Sclose(space)
        if dataset < 0:
            raise IOError("cannot create variable %s" % name)
        for index, scale in enumerate((self.dimensions[0], self.dimensions[1], last)):
            if self.lib.hl.H5DSattach_scale(dataset, scale, index) < 0:
                raise IOError("H5DSattach_scale failed for %s" % name)
        self._int_attribute(dataset, "_Netcdf4Coordinates", [0, 1, last_id])
        self.variables[name] = dataset
        self.parts[name] = parts

    def _hyperslab(self, dataset, index, part, parts):
        start = (ctypes.c_uint64*3)(index, 0, part)
        count = (ctypes.c_uint64*3)(1, self.num_rays, parts)
        file_space = self.lib.H5Dget_space(dataset)
        self.lib.H5Sselect_hyperslab(file_space, _H5S_SELECT_SET, start, None, count, None)
        mem_space = self.lib.H5Screate_simple(3, count, None)
        return file_space, mem_space

    def write(self, record, index=None):
        """data_set::write (output.hpp:354-400): one record {variable: array of num_rays} (complex arrays
        for variables created with parts = 2), appended, or at time index `index` (output.hpp:363)."""
        with self.lock:
            self._write(record, index)

    def _write(self, record, index):
        at = self.records if index is None else int(index)
        for name, dataset in self.variables.items():
            parts = self.parts[name]
            values = np.ascontiguousarray(record[name], dtype=np.complex128 if parts == 2 and self.dtype == np.float64
                                          else (np.complex64 if parts == 2 else self.dtype))
            assert values.size == self.num_rays
            if self._extent(dataset)[0] < at + 1:
                self.lib.H5Dset_extent(dataset, (ctypes.c_uint64*3)(at + 1, self.num_rays, parts))
            file_space, mem_space = self._hyperslab(dataset, at, 0, parts)
            status = self.lib.H5Dwrite(dataset, self.native, mem_space, file_space, 0, values.ctypes.data)
            self.lib.H5Sclose(mem_space)
            self.lib.H5Sclose(file_space)
            if status < 0:
                raise IOError("H5Dwrite failed for %s" % name)
        if at + 1 > self.records:
            self.records = at + 1
            self.lib.H5Dset_extent(self.dimensions[0], (ctypes.c_uint64*1)(self.records))    # the length of `time`
        self.lib.H5Fflush(self.file, 1)                      # result.sync_file(), output.hpp:399

    def read(self, name, index, part=0):
        """data_set::read of one referenced variable (output.hpp:285-350, :412-470): the values of `name`
        at time index `index`; part = 1 reads the imaginary part of a complex variable
        (reference_imag_variable)."""
        values = np.empty(self.num_rays, dtype=self.dtype)
        with self.lock:
            if name not in self.opened:
                self.opened[name] = self.variables.get(name) or self._open_dataset(self._stored_name(name))
            dataset = self.opened[name]
            file_space, mem_space = self._hyperslab(dataset, int(index), part, 1)
            status = self.lib.H5Dread(dataset, self.native, mem_space, file_space, 0, values.ctypes.data)
            self.lib.H5Sclose(mem_space)
            self.lib.H5Sclose(file_space)
        if status < 0:
            raise IOError("H5Dread failed for %s" % name)
        return values

    def close(self):
        if self.file is not None:
            extra = [d for name, d in self.opened.items() if name not in self.variables]
            if self.complex_dimension is not None:
                extra.append(self.complex_dimension)
            for dataset in list(self.variables.values()) + self.dimensions + extra:
                self.lib.H5Dclose(dataset)
            self.lib.H5Fclose(self.file)
            self.file = None


#  Variables of the reference's ray files, solver.hpp:338-346.
RAY_VARIABLES = (("time", "t"), ("residual", "residual"), ("w", "w"), ("x", "x"), ("y", "y"), ("z", "z"),
                 ("kx", "kx"), ("ky", "ky"), ("kz", "kz"))


class TrajectoryWriter:
    """write_step for an xrays.Rk4ColdPlasmaEfit whose state lives in torch tensors
    (Rk4ColdPlasmaEfit(..., device_state=True))."""

    def __init__(self, solver, path):
        import torch
        self.torch = torch
        self.solver = solver
        if solver.device is None:
            raise ValueError("construct the solver with device_state=True")
        n = solver.num_rays
        self.file = ResultFile(path, n, solver.np_dtype)
        for name, _ in RAY_VARIABLES:
            self.file.create_variable(name)
        device = next(iter(solver.device.values())).device
        self.snapshot = {key: torch.empty_like(solver.device[key]) for _, key in RAY_VARIABLES}
        self.pinned = {key: torch.empty(n, dtype=solver.device[key].dtype, pin_memory=True)
                       for _, key in RAY_VARIABLES}
        self.copy_stream = torch.cuda.Stream(device=device)
        self.thread = None
        self.error = None

    def write_step(self):
        """Append the current state; returns as soon as the snapshot is enqueued."""
        torch = self.torch
        self.wait()                                     # sync.join(), solver.hpp:419
        compute = self.solver.torch_stream or torch.cuda.current_stream()
        with torch.cuda.stream(compute):                # D2D snapshot, ordered with the step kernels
            for _, key in RAY_VARIABLES:
                self.snapshot[key].copy_(self.solver.device[key], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(compute)
        with torch.cuda.stream(self.copy_stream):       # D2H on the second stream
            self.copy_stream.wait_event(ready)
            for _, key in RAY_VARIABLES:
                self.pinned[key].copy_(self.snapshot[key], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)

        def work():
            try:
                done.synchronize()
                self.file.write({name: self.pinned[key].numpy() for name, key in RAY_VARIABLES})
            except Exception as error:                  # surfaced by the next wait()
                self.error = error

        self.thread = threading.Thread(target=work)
        self.thread.start()

    def wait(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        if self.error is not None:
            error, self.error = self.error, None
            raise error

    def close(self):
        self.wait()
        self.file.close()
