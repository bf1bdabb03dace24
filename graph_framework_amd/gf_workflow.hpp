//------------------------------------------------------------------------------
///  @file gf_workflow.hpp
///  @brief C++ host side over the C ABI of include/gf_hip.h for hosts that have exported work
///  items (GFIR files) instead of the reference's graph front end.
///
///  Same names and call order as the reference:
///    gf::workflow::manager<T>   <- workflow::manager<T, SAFE_MATH>   (workflow.hpp:215-425)
///      add_preitem / add_item / add_converge_item, compile, pre_run, run, wait,
///      copy_to_device, copy_to_host, check_value, get_context
///    gf::solver::ray_solver<T>  <- solver::solver_interface          (solver.hpp:123-430)
///      init(unknown, tolerance, max_iterations), compile, step, sync_host, sync_device,
///      check_residual, write_step (result file: gf_output.hpp)
///  Variables are named by their symbol (the reference keys buffers by leaf_node*); a work item
///  arrives as the bytes of a GFIR file.  Errors follow the reference (message on stderr and
///  exit(1): graph_c_binding.cpp:2355, cuda_context.hpp:55-67).  Header only; link libgf_hip.so.
///  The Python package mirrors the same classes for the tests and bench.py.
//------------------------------------------------------------------------------
#ifndef gf_workflow_hpp
#define gf_workflow_hpp

#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../include/gf_hip.h"
#include "../include/gfir.h"
#include "gf_output.hpp"

namespace gf {

//  Buffer key of a variable name (FNV-1a, as graph_framework_amd/backend.py key_of).
inline uint64_t key_of(const std::string &name) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : name) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

inline std::vector<char> read_item(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        std::cerr << "cannot open work item " << path << std::endl;
        exit(1);
    }
    return std::vector<char> ((std::istreambuf_iterator<char> (f)), std::istreambuf_iterator<char> ());
}

template<typename T> constexpr uint32_t dtype_of() { return sizeof(T) == 8 ? GFIR_F64 : GFIR_F32; }

namespace workflow {

template<typename T> class manager;

//------------------------------------------------------------------------------
///  @brief workflow::work_item (workflow.hpp:22-76).
//------------------------------------------------------------------------------
template<typename T>
class work_item {
protected:
    gfhip_context *context;
    gfhip_kernel *kernel;
    std::vector<std::string> inputs, outputs;
    std::map<std::string, const T *> initial;

public:
    work_item(gfhip_context *ctx, const std::vector<char> &gfir, const std::vector<std::string> &in,
              const std::vector<std::string> &out, const size_t size, const std::map<std::string, const T *> &init) :
    context(ctx), kernel(gfhip_add_kernel(ctx, gfir.data(), gfir.size(), size)), inputs(in), outputs(out), initial(init) {
        if (!kernel) fail("gfhip_add_kernel");
    }
    virtual ~work_item() {}

    void fail(const char *what) const {
        std::cerr << what << ": " << gfhip_last_error(context) << std::endl;
        exit(1);
    }

///  work_item::create_kernel_call (workflow.hpp:52-58).
    void create_kernel_call() {
        std::vector<uint64_t> in_keys, out_keys;
        std::vector<const void *> init;
        for (auto &name : inputs) {
            in_keys.push_back(key_of(name));
            auto found = initial.find(name);
            init.push_back(found == initial.end() ? nullptr : found->second);
        }
        for (auto &name : outputs) out_keys.push_back(key_of(name));
        if (gfhip_create_kernel_call(kernel, in_keys.data(), init.data(), nullptr, out_keys.data())) fail("gfhip_create_kernel_call");
    }

///  work_item::run (workflow.hpp:63-65).
    virtual void run() {
        if (gfhip_run(kernel, 1)) fail("gfhip_run");
    }

    gfhip_kernel *get_kernel() { return kernel; }
};

//------------------------------------------------------------------------------
///  @brief workflow::converge_item (workflow.hpp:129-205).
//------------------------------------------------------------------------------
template<typename T>
class converge_item final : public work_item<T> {
    const T tolerance;
    const size_t max_iterations;

public:
    size_t iterations = 0;
    T last_max = 0;

    converge_item(gfhip_context *ctx, const std::vector<char> &gfir, const std::vector<std::string> &in,
                  const std::vector<std::string> &out, const size_t size, const std::map<std::string, const T *> &init,
                  const T tol, const size_t max_iter) :
    work_item<T> (ctx, gfir, in, out, size, init), tolerance(tol), max_iterations(max_iter) {}

///  converge_item::run (workflow.hpp:179-205): the stall loop on the global max.
    void run() override {
        double last = 0.0;
        if (gfhip_converge(this->kernel, static_cast<double> (tolerance), max_iterations, &iterations, &last)) {
            this->fail("gfhip_converge");
        }
        last_max = static_cast<T> (last);
    }
};

//------------------------------------------------------------------------------
///  @brief workflow::manager<T>(index) (workflow.hpp:215-425).
//------------------------------------------------------------------------------
template<typename T>
class manager {
    gfhip_context *context;
    std::vector<std::unique_ptr<work_item<T>>> preitems, items;

    void check(const int status, const char *what) const {
        if (status) {
            std::cerr << what << ": " << gfhip_last_error(context) << std::endl;
            exit(1);
        }
    }

public:
    explicit manager(const size_t index) : context(gfhip_create_context(static_cast<int> (index), nullptr)) {
        if (!context) {
            std::cerr << gfhip_last_error(nullptr) << std::endl;
            exit(1);
        }
    }
    ~manager() {
        preitems.clear();
        items.clear();
        gfhip_destroy_context(context);
    }
    manager(const manager &) = delete;
    manager &operator=(const manager &) = delete;

    work_item<T> *add_preitem(const std::vector<char> &gfir, const std::vector<std::string> &inputs,
                              const std::vector<std::string> &outputs, const size_t size,
                              const std::map<std::string, const T *> &initial = {}) {
        preitems.emplace_back(new work_item<T> (context, gfir, inputs, outputs, size, initial));
        return preitems.back().get();
    }
    work_item<T> *add_item(const std::vector<char> &gfir, const std::vector<std::string> &inputs,
                           const std::vector<std::string> &outputs, const size_t size,
                           const std::map<std::string, const T *> &initial = {}) {
        items.emplace_back(new work_item<T> (context, gfir, inputs, outputs, size, initial));
        return items.back().get();
    }
    converge_item<T> *add_converge_item(const std::vector<char> &gfir, const std::vector<std::string> &inputs,
                                        const std::vector<std::string> &outputs, const size_t size,
                                        const std::map<std::string, const T *> &initial = {},
                                        const T tolerance = 1.0E-30, const size_t max_iterations = 1000) {
        auto *item = new converge_item<T> (context, gfir, inputs, outputs, size, initial, tolerance, max_iterations);
        items.emplace_back(item);
        return item;
    }

///  manager::compile (workflow.hpp:336-345): build the module, then bind every item.
    void compile() {
        check(gfhip_compile(context), "gfhip_compile");
        for (auto &item : preitems) item->create_kernel_call();
        for (auto &item : items) item->create_kernel_call();
    }
    void pre_run() { for (auto &item : preitems) item->run(); }
    void run() { for (auto &item : items) item->run(); }
    void wait() { check(gfhip_wait(context), "gfhip_wait"); }

    void copy_to_device(const std::string &name, const T *source) {
        check(gfhip_copy_to_device(context, key_of(name), source), "gfhip_copy_to_device");
    }
    void copy_to_host(const std::string &name, T *destination) {
        check(gfhip_copy_to_host(context, key_of(name), destination), "gfhip_copy_to_host");
    }
    T check_value(const size_t index, const std::string &name) {
        double value = 0.0;
        check(gfhip_check_value(context, key_of(name), index, &value), "gfhip_check_value");
        return static_cast<T> (value);
    }
    gfhip_context *get_context() { return context; }
};

}  // namespace workflow

namespace solver {

//------------------------------------------------------------------------------
///  @brief solver::solver_interface (solver.hpp:123-430) over exported work items
///  `<prefix>loss_kernel_<unknown>_<f64|f32>.gfir` and `<prefix>solver_kernel_<f64|f32>.gfir`.
//------------------------------------------------------------------------------
template<typename T>
class ray_solver {
    const std::string directory, prefix;
    const size_t num_rays;
    workflow::manager<T> work;
    workflow::work_item<T> *solver_item = nullptr;
    const std::vector<std::string> names = {"t", "w", "x", "y", "z", "kx", "ky", "kz"};   // solver.hpp:304-313

    std::string item_path(const std::string &item) const {
        return directory + "/" + prefix + item + (sizeof(T) == 8 ? "_f64.gfir" : "_f32.gfir");
    }
    std::map<std::string, const T *> initial() const {
        std::map<std::string, const T *> init;
        for (auto &n : names) init[n] = state.at(n).data();
        return init;
    }

    std::unique_ptr<output::result_file<T>> file;
    std::vector<const T *> mirrors;
    std::thread sync;

public:
///  Host copies of the ray variables (the reference's variable nodes), input order of the kernels.
    std::map<std::string, std::vector<T>> state;
    size_t newton_iterations = 0;

    ray_solver(const std::string &workload_directory, const std::string &workload_prefix, const size_t rays,
               const size_t index = 0) :
    directory(workload_directory), prefix(workload_prefix), num_rays(rays), work(index) {
        for (auto &n : names) state[n].assign(num_rays, static_cast<T> (0));
    }

///  solver_interface::init(x, tolerance, max_iterations) (solver.hpp:254-274) ->
///  dispersion_interface::solve (dispersion.hpp:1452-1475); the unknown is copied back to the host.
    T init(const std::string &unknown, const T tolerance = 1.0E-30, const size_t max_iterations = 1000) {
        auto *item = work.add_converge_item(read_item(item_path("loss_kernel_" + unknown)), names, {"newton_residual"},
                                            num_rays, initial(), tolerance, max_iterations);
        if (gfhip_compile(work.get_context())) item->fail("gfhip_compile");
        item->create_kernel_call();
        item->run();
        newton_iterations = item->iterations;
        work.copy_to_host(unknown, state[unknown].data());
        return item->last_max;
    }

///  solver_interface::compile (solver.hpp:303-349).
    void compile() {
        solver_item = work.add_item(read_item(item_path("solver_kernel")), names, {"residual"}, num_rays, initial());
        if (gfhip_compile(work.get_context())) solver_item->fail("gfhip_compile");
        solver_item->create_kernel_call();
    }

    void step() { solver_item->run(); }                                     // solver.hpp:382
    void sync_host() { for (auto &n : names) work.copy_to_host(n, state[n].data()); }      // :368-377
    void sync_device() { for (auto &n : names) work.copy_to_device(n, state[n].data()); }  // :354-363
    T check_residual(const size_t index) { return work.check_value(index, "residual"); }   // :392

///  The result file of solver_interface's constructor (solver.hpp:220-226, variables created by compile,
///  :338-346): call after compile().  The variables are bound to the context's host mirrors of the state
///  buffers (data_set::create_variable(file, name, node, context) -> context.get_buffer, output.hpp:260-273).
    void open_result_file(const std::string &filename) {
        file.reset(new output::result_file<T> (filename, num_rays));
        const std::vector<std::pair<std::string, std::string>> stored = {{"time", "t"}, {"residual", "residual"}, {"w", "w"},
            {"x", "x"}, {"y", "y"}, {"z", "z"}, {"kx", "kx"}, {"ky", "ky"}, {"kz", "kz"}};
        mirrors.clear();
        for (auto &v : stored) {
            file->create_variable(v.first);
            const T *mirror = static_cast<const T *> (gfhip_get_host_buffer(work.get_context(), key_of(v.second), nullptr));
            if (!mirror) {
                std::cerr << "gfhip_get_host_buffer: " << gfhip_last_error(work.get_context()) << std::endl;
                exit(1);
            }
            mirrors.push_back(mirror);
        }
    }

///  solver_interface::write_step (solver.hpp:418-424): join the previous writer, wait for the device (which
///  refreshes the host mirrors), write the record on a thread of its own while the next steps run.
    void write_step() {
        if (!file) return;
        if (sync.joinable()) sync.join();
        work.wait();
        sync = std::thread([this] { file->write(mirrors); });
    }

///  End of the trace (xrays.cpp:1078-1083): the last writer is joined and the file closed.
    void close_result_file() {
        if (sync.joinable()) sync.join();
        if (file) file->close();
    }
    ~ray_solver() { close_result_file(); }
    T newton_residual(const size_t index) { return work.check_value(index, "newton_residual"); }
    workflow::manager<T> &manager() { return work; }
};

}  // namespace solver
}  // namespace gf

#endif /* gf_workflow_hpp */
