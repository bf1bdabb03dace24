//------------------------------------------------------------------------------
///  @file gf_output.hpp
///  @brief Trajectory files for the C++ host mirror: output::result_file + output::data_set
///  (graph_framework/output.hpp:32-400) as solver_interface::write_step uses them (solver.hpp:418-424).
///
///  The reference writes through NetCDF-C, which the image lacks; like the Python mirror
///  (graph_framework_amd/output.py, whose header states the conventions) this writes NetCDF-4's
///  on-disk form through libhdf5: the dimensions `time` (unlimited), `num_rays`, `ray_dim` as
///  dimension-scale datasets without coordinate variables, every variable of shape
///  (time, num_rays, ray_dim) with the scales attached and `_Netcdf4Coordinates`, a variable that
///  shares a dimension's name stored as `_nc4_non_coord_<name>`, the root attribute `_NCProperties`.
///  libhdf5 and libhdf5_hl are opened at run time (dlopen), so a host that never writes a file does
///  not need them; errors follow the reference (message on stderr, exit(1)).
//------------------------------------------------------------------------------
#ifndef gf_output_hpp
#define gf_output_hpp

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace gf {
namespace output {

typedef int64_t hid_t;
typedef unsigned long long hsize_t;

//  The entry points of libhdf5 / libhdf5_hl this file calls, resolved once.
struct hdf5 {
    void *core = nullptr, *high = nullptr;
    int (*H5open)() = nullptr;
    hid_t (*H5Fcreate)(const char *, unsigned, hid_t, hid_t) = nullptr;
    int (*H5Fflush)(hid_t, int) = nullptr;
    int (*H5Fclose)(hid_t) = nullptr;
    hid_t (*H5Screate)(int) = nullptr;
    hid_t (*H5Screate_simple)(int, const hsize_t *, const hsize_t *) = nullptr;
    int (*H5Sselect_hyperslab)(hid_t, int, const hsize_t *, const hsize_t *, const hsize_t *, const hsize_t *) = nullptr;
    int (*H5Sclose)(hid_t) = nullptr;
    hid_t (*H5Pcreate)(hid_t) = nullptr;
    int (*H5Pset_chunk)(hid_t, int, const hsize_t *) = nullptr;
    int (*H5Pclose)(hid_t) = nullptr;
    hid_t (*H5Dcreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    int (*H5Dset_extent)(hid_t, const hsize_t *) = nullptr;
    hid_t (*H5Dget_space)(hid_t) = nullptr;
    int (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void *) = nullptr;
    int (*H5Dclose)(hid_t) = nullptr;
    hid_t (*H5Tcopy)(hid_t) = nullptr;
    int (*H5Tset_size)(hid_t, size_t) = nullptr;
    int (*H5Tset_strpad)(hid_t, int) = nullptr;
    int (*H5Tclose)(hid_t) = nullptr;
    hid_t (*H5Acreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t) = nullptr;
    int (*H5Awrite)(hid_t, hid_t, const void *) = nullptr;
    int (*H5Aclose)(hid_t) = nullptr;
    int (*H5DSset_scale)(hid_t, const char *) = nullptr;
    int (*H5DSattach_scale)(hid_t, hid_t, unsigned) = nullptr;
    hid_t native_double = -1, native_float = -1, native_int = -1, c_string = -1, scale_type = -1, dataset_create = -1;

    static void fail(const std::string &what) {
        std::cerr << "gf::output: " << what << std::endl;
        exit(1);
    }

    template<typename F> void resolve(void *library, F &target, const char *name) {
        target = reinterpret_cast<F> (dlsym(library, name));
        if (!target) fail(std::string("libhdf5 has no ") + name);
    }

    static void *open_any(const std::vector<const char *> &names) {
        for (auto name : names) {
            if (void *library = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) return library;
        }
        return nullptr;
    }

    hdf5() {
        core = open_any({"libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so"});
        if (!core) fail("libhdf5 not found");
        high = open_any({"libhdf5_hl.so", "/opt/conda/lib/libhdf5_hl.so.100", "/opt/conda/lib/libhdf5_hl.so"});
        if (!high) fail("libhdf5_hl not found");
        resolve(core, H5open, "H5open");
        resolve(core, H5Fcreate, "H5Fcreate");
        resolve(core, H5Fflush, "H5Fflush");
        resolve(core, H5Fclose, "H5Fclose");
        resolve(core, H5Screate, "H5Screate");
        resolve(core, H5Screate_simple, "H5Screate_simple");
        resolve(core, H5Sselect_hyperslab, "H5Sselect_hyperslab");
        resolve(core, H5Sclose, "H5Sclose");
        resolve(core, H5Pcreate, "H5Pcreate");
        resolve(core, H5Pset_chunk, "H5Pset_chunk");
        resolve(core, H5Pclose, "H5Pclose");
        resolve(core, H5Dcreate2, "H5Dcreate2");
        resolve(core, H5Dset_extent, "H5Dset_extent");
        resolve(core, H5Dget_space, "H5Dget_space");
        resolve(core, H5Dwrite, "H5Dwrite");
        resolve(core, H5Dclose, "H5Dclose");
        resolve(core, H5Tcopy, "H5Tcopy");
        resolve(core, H5Tset_size, "H5Tset_size");
        resolve(core, H5Tset_strpad, "H5Tset_strpad");
        resolve(core, H5Tclose, "H5Tclose");
        resolve(core, H5Acreate2, "H5Acreate2");
        resolve(core, H5Awrite, "H5Awrite");
        resolve(core, H5Aclose, "H5Aclose");
        resolve(high, H5DSset_scale, "H5DSset_scale");
        resolve(high, H5DSattach_scale, "H5DSattach_scale");
        H5open();
        auto global = [&] (const char *name) -> hid_t {
            hid_t *value = reinterpret_cast<hid_t *> (dlsym(core, name));
            if (!value) fail(std::string("libhdf5 has no ") + name);
            return *value;
        };
        native_double = global("H5T_NATIVE_DOUBLE_g");
        native_float = global("H5T_NATIVE_FLOAT_g");
        native_int = global("H5T_NATIVE_INT_g");
        c_string = global("H5T_C_S1_g");
        scale_type = global("H5T_IEEE_F32BE_g");
        dataset_create = global("H5P_CLS_DATASET_CREATE_ID_g");
    }

    static hdf5 &library() {
        static hdf5 instance;
        return instance;
    }
};

///  output::sync (output.hpp:21): libhdf5 calls are serialised over all files of the process.
inline std::mutex &sync() {
    static std::mutex lock;
    return lock;
}

//------------------------------------------------------------------------------
///  @brief output::result_file(filename, num_rays) with the variables of one output::data_set<T>.
//------------------------------------------------------------------------------
template<typename T>
class result_file {
    hdf5 &h5 = hdf5::library();
    hid_t file = -1;
    hid_t dimensions[3] = {-1, -1, -1};
    std::vector<std::pair<std::string, hid_t>> variables;
    size_t num_rays;
    size_t records = 0;
    const hsize_t unlimited = ~0ull;

    void string_attribute(const hid_t where, const char *name, const std::string &text, size_t size) {
        if (!size) size = text.size() + 1;
        const hid_t kind = h5.H5Tcopy(h5.c_string);
        h5.H5Tset_size(kind, size);
        h5.H5Tset_strpad(kind, 0);
        const hid_t space = h5.H5Screate(0);
        const hid_t attribute = h5.H5Acreate2(where, name, kind, space, 0, 0);
        if (attribute < 0) hdf5::fail(std::string("cannot create attribute ") + name);
        std::vector<char> buffer(size, '\0');
        std::memcpy(buffer.data(), text.data(), std::min(text.size(), size - 1));
        h5.H5Awrite(attribute, kind, buffer.data());
        h5.H5Aclose(attribute);
        h5.H5Sclose(space);
        h5.H5Tclose(kind);
    }

    void int_attribute(const hid_t where, const char *name, const std::vector<int> &values) {
        const hsize_t count = values.size();
        const hid_t space = values.size() == 1 ? h5.H5Screate(0) : h5.H5Screate_simple(1, &count, nullptr);
        const hid_t attribute = h5.H5Acreate2(where, name, h5.native_int, space, 0, 0);
        if (attribute < 0) hdf5::fail(std::string("cannot create attribute ") + name);
        h5.H5Awrite(attribute, h5.native_int, values.data());
        h5.H5Aclose(attribute);
        h5.H5Sclose(space);
    }

//  A netCDF-4 dimension without a coordinate variable (length 0 = unlimited).
    hid_t dimension(const char *name, const hsize_t length, const bool is_unlimited, const int dimid) {
        const hid_t plist = h5.H5Pcreate(h5.dataset_create);
        hid_t space;
        if (is_unlimited) {
            const hsize_t zero = 0, chunk = 1024;
            space = h5.H5Screate_simple(1, &zero, &unlimited);
            h5.H5Pset_chunk(plist, 1, &chunk);
        } else {
            space = h5.H5Screate_simple(1, &length, nullptr);
        }
        const hid_t dataset = h5.H5Dcreate2(file, name, h5.scale_type, space, 0, plist, 0);
        h5.H5Pclose(plist);
        h5.H5Sclose(space);
        if (dataset < 0) hdf5::fail(std::string("cannot create dimension ") + name);
        if (h5.H5DSset_scale(dataset, nullptr) < 0) hdf5::fail("H5DSset_scale failed");
        char text[80];
        std::snprintf(text, sizeof(text), "This is a netCDF dimension but not a netCDF variable.%10llu", is_unlimited ? 0ull : length);
        string_attribute(dataset, "NAME", text, 64);
        int_attribute(dataset, "_Netcdf4Dimid", {dimid});
        return dataset;
    }

    static std::string stored_name(const std::string &name) {
        return name == "time" || name == "num_rays" || name == "ray_dim" ? "_nc4_non_coord_" + name : name;
    }

public:
///  result_file(filename, num_rays), output.hpp:50-66.
    result_file(const std::string &path, const size_t rays) : num_rays(rays) {
        std::lock_guard<std::mutex> hold(sync());
        file = h5.H5Fcreate(path.c_str(), 2u, 0, 0);
        if (file < 0) hdf5::fail("cannot create " + path);
        string_attribute(file, "_NCProperties", "version=2,graph_framework_amd=1,hdf5=1.10", 0);
        dimensions[0] = dimension("time", 0, true, 0);
        dimensions[1] = dimension("num_rays", num_rays ? num_rays : 1, false, 1);
        dimensions[2] = dimension("ray_dim", 1, false, 2);
    }
    ~result_file() { close(); }
    result_file(const result_file &) = delete;
    result_file &operator=(const result_file &) = delete;

///  data_set::create_variable (output.hpp:260-273): nc_def_var(name, type, {time, num_rays, ray_dim}).
    void create_variable(const std::string &name) {
        std::lock_guard<std::mutex> hold(sync());
        const hsize_t dims[3] = {records, num_rays, 1}, maxdims[3] = {unlimited, num_rays, 1};
        const hsize_t chunk[3] = {1, num_rays ? num_rays : 1, 1};
        const hid_t space = h5.H5Screate_simple(3, dims, maxdims);
        const hid_t plist = h5.H5Pcreate(h5.dataset_create);
        h5.H5Pset_chunk(plist, 3, chunk);
        const hid_t dataset = h5.H5Dcreate2(file, stored_name(name).c_str(), sizeof(T) == 8 ? h5.native_double : h5.native_float,
                                            space, 0, plist, 0);
        h5.H5Pclose(plist);
        h5.H5Sclose(space);
        if (dataset < 0) hdf5::fail("cannot create variable " + name);
        for (unsigned index = 0; index < 3; index++) {
            if (h5.H5DSattach_scale(dataset, dimensions[index], index) < 0) hdf5::fail("H5DSattach_scale failed for " + name);
        }
        int_attribute(dataset, "_Netcdf4Coordinates", {0, 1, 2});
        variables.push_back({name, dataset});
    }

///  data_set::write (output.hpp:354-400): append one record; `values` holds one pointer per variable, in
///  the order the variables were created.
    void write(const std::vector<const T *> &values) {
        std::lock_guard<std::mutex> hold(sync());
        if (values.size() != variables.size()) hdf5::fail("write: one array per variable is needed");
        const hsize_t at = records;
        const hid_t native = sizeof(T) == 8 ? h5.native_double : h5.native_float;
        for (size_t v = 0; v < variables.size(); v++) {
            const hid_t dataset = variables[v].second;
            const hsize_t extent[3] = {at + 1, num_rays, 1};
            h5.H5Dset_extent(dataset, extent);
            const hsize_t start[3] = {at, 0, 0}, count[3] = {1, num_rays, 1};
            const hid_t file_space = h5.H5Dget_space(dataset);
            h5.H5Sselect_hyperslab(file_space, 0, start, nullptr, count, nullptr);
            const hid_t memory_space = h5.H5Screate_simple(3, count, nullptr);
            const int status = h5.H5Dwrite(dataset, native, memory_space, file_space, 0, values[v]);
            h5.H5Sclose(memory_space);
            h5.H5Sclose(file_space);
            if (status < 0) hdf5::fail("H5Dwrite failed for " + variables[v].first);
        }
        records = at + 1;
        const hsize_t length = records;
        h5.H5Dset_extent(dimensions[0], &length);               // the length of `time`
        h5.H5Fflush(file, 1);                                   // result.sync_file(), output.hpp:399
    }

    void close() {
        std::lock_guard<std::mutex> hold(sync());
        if (file < 0) return;
        for (auto &v : variables) h5.H5Dclose(v.second);
        for (auto d : dimensions) h5.H5Dclose(d);
        h5.H5Fclose(file);
        file = -1;
    }

    size_t size() const { return records; }
};

}  // namespace output
}  // namespace gf

#endif /* gf_output_hpp */
