// ---------------------------------------------------------------------------
// hip_context_physics.cpp — gpu::hip_context (graph_framework_amd/hip_context.hpp) under the
// scenarios of graph_tests/solver_test.cpp and graph_tests/physics_test.cpp.
//
// TEST INFRASTRUCTURE; builds to oracle/_ref/hip_context_physics (git-ignored) from the
// reference's expression-graph headers where they lie, links the product library, runs only
// where a GPU is present.  Every scenario of ref_scenarios.hpp is replayed twice — on the
// tape interpreter and on hip_context driven the way jit::context / workflow::manager /
// dispersion_interface::solve drive a backend (a context per Newton solve,
// dispersion.hpp:1452-1475; a context for the solver, solver.hpp:303-349; host variable
// buffers pushed with sync_device, solver.hpp:354-363) — and the two reports are compared.
// They must be identical text (%.17g of every state) except where the graph has an exp node.
//
// Usage: hip_context_physics
// ---------------------------------------------------------------------------
#include "hip_context_driver.hpp"
#include "ref_scenarios.hpp"

namespace {

const char *state_names[8] = {"t", "w", "x", "y", "z", "kx", "ky", "kz"};

//  The tape side: as ray_solver of ref_physics.cpp without the GFIR export.
struct tape_solver {
    ray_variables<T> v;
    equilibrium_base<T> &eq;
    dispersion_interface<T> D;
    const size_t n;
    std::vector<std::vector<T>> cols;
    std::vector<T> residual;
    std::map<int, std::unique_ptr<work_item<T>>> loss;
    std::unique_ptr<work_item<T>> solver;
    std::vector<size_t> newton_iterations;

    tape_solver(const std::string &, const std::string &, equilibrium_base<T> &e, dispersion_function<T> f,
                const size_t num_rays = 1) :
    eq(e), D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, e, f), n(num_rays),
    cols(8, std::vector<T> (num_rays, 0.0)), residual(num_rays, 0.0) {}

    std::vector<T> &column(const char *key) {
        for (int i = 0; i < 8; i++) {
            if (std::string(key) == state_names[i]) return cols[i];
        }
        exit(1);
    }
    void set(const char *key, const T value) { for (auto &e : column(key)) e = value; }
    void set(const char *key, const size_t i, const T value) { column(key)[i] = value; }
    T get(const char *key, const size_t i = 0) { return column(key)[i]; }
    std::vector<T *> pointers() {
        std::vector<T *> p;
        for (auto &c : cols) p.push_back(c.data());
        return p;
    }
    T init(const int var, const T tolerance = 1.0E-30, const size_t max_iterations = 1000) {
        if (!loss.count(var)) loss[var].reset(new work_item<T> (make_loss_kernel(v, D.D, var, static_cast<T> (1.0))));
        T last;
        newton_iterations.push_back(converge(*loss[var], n, pointers(), residual.data(), tolerance, max_iterations, &last));
        return last;
    }
    void compile(const method m, const T dt) {
        switch (m) {
            case method::rk2: solver.reset(new work_item<T> (make_rk2_kernel(v, eq, dt, D))); break;
            case method::rk4: solver.reset(new work_item<T> (make_solver_kernel(v, eq, dt, D))); break;
            default: solver.reset(new work_item<T> (make_split_simplextic_kernel(v, eq, dt, D)));
        }
    }
    void step() { solver->run(n, pointers(), {residual.data()}); }
    T residual_at(const size_t i) { return residual[i]; }
    std::string state() {
        std::ostringstream s;
        s << "[";
        for (size_t i = 0; i < n; i++) {
            s << (i ? ", [" : "[");
            for (int c = 0; c < 8; c++) s << g17(cols[c][i]) << ", ";
            s << g17(residual[i]) << "]";
        }
        s << "]";
        return s.str();
    }
};

//  The device side: solver::solver_interface over gpu::hip_context.
struct hip_ray_solver {
    ray_variables<T> v;
    equilibrium_base<T> &eq;
    dispersion_interface<T> D;
    const size_t n;
    std::vector<leaf<T>> nodes;                         // t w x y z kx ky kz
    std::vector<T> residual;
    std::vector<size_t> newton_iterations;

    std::unique_ptr<gpu::hip_context<T>> solver_context;
    std::unique_ptr<work_item<T>> solver;
    std::function<void()> step_call;
    bool device_ahead = false;                          // the solver context holds newer state than the host
    bool residual_from_step = false;                    // the last residual came from solver_kernel, not from Newton

    hip_ray_solver(const std::string &, const std::string &, equilibrium_base<T> &e, dispersion_function<T> f,
                   const size_t num_rays = 1) :
    eq(e), D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, e, f), n(num_rays), nodes(v.inputs()), residual(num_rays, 0.0) {
        for (auto &node : nodes) {
            graph::variable_cast(node)->set(std::vector<T> (n, static_cast<T> (0.0)));
        }
    }

    leaf<T> node(const char *key) {
        for (int i = 0; i < 8; i++) {
            if (std::string(key) == state_names[i]) return nodes[i];
        }
        exit(1);
    }

//  solver_interface::sync_host / sync_device, solver.hpp:354-377.
    void sync_host() {
        if (solver_context && device_ahead) {
            std::vector<T> host(n);
            for (auto &nd : nodes) {
                solver_context->copy_to_host(nd, host.data());
                graph::variable_cast(nd)->set(host);
            }
        }
        device_ahead = false;
    }
    void sync_device() {
        if (solver_context) {
            for (auto &nd : nodes) {
                solver_context->copy_to_device(nd, graph::variable_cast(nd)->data());
            }
        }
    }

    void set(const char *key, const T value) {
        sync_host();
        graph::variable_cast(node(key))->set(std::vector<T> (n, value));
        sync_device();
    }
    void set(const char *key, const size_t i, const T value) {
        sync_host();
        graph::variable_cast(node(key))->set(i, value);
        sync_device();
    }
    T get(const char *key, const size_t i = 0) {
        sync_host();
        return node(key)->evaluate().at(i);
    }

//  solver_interface::init -> dispersion_interface::solve (dispersion.hpp:1452-1475): its own
//  manager, hence its own context; newton (newton.hpp:34-51) + converge_item::run
//  (workflow.hpp:179-205); the unknown is copied back into the host variable.
    T init(const int var, const T tolerance = 1.0E-30, const size_t max_iterations = 1000) {
        sync_host();
        work_item<T> loss = make_loss_kernel(v, D.D, var, static_cast<T> (1.0));
        graph::input_nodes<T> in;
        graph::map_nodes<T> set;
        to_lists<T> (loss, in, set);

        gpu::hip_context<T> gpu(0);
        std::ostringstream source;
        jit::register_map registers;
        gpu.create_header(source);
        add_kernel<T> (gpu, source, registers, "loss_kernel", in, loss.out_nodes, set, n);
        gpu.create_reduction(source, n);
        gpu.compile(source.str(), {"loss_kernel"}, true);
        jit::texture1d_list tex1d;
        jit::texture2d_list tex2d;
        auto call = gpu.create_kernel_call("loss_kernel", in, loss.out_nodes, graph::shared_random_state<T> (),
                                           n, tex1d, tex2d);
        auto max_kernel = gpu.create_max_call(loss.out_nodes.back(), call);

        size_t iterations = 0;
        T max_residual = max_kernel();
        T last_max = std::numeric_limits<T>::max();
        T off_last_max = std::numeric_limits<T>::max();
        while (std::abs(max_residual) > std::abs(tolerance)                &&
               std::abs(last_max - max_residual) > std::abs(tolerance)     &&
               std::abs(off_last_max - max_residual) > std::abs(tolerance) &&
               iterations++ < max_iterations) {
            last_max = max_residual;
            if (!(iterations%2)) {
                off_last_max = max_residual;
            }
            max_residual = max_kernel();
        }
        newton_iterations.push_back(iterations);

        const leaf<T> unknowns[7] = {v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z};
        std::vector<T> host(n);
        gpu.copy_to_host(unknowns[var], host.data());
        graph::variable_cast(unknowns[var])->set(host);                 // dispersion.hpp:1472
        for (size_t i = 0; i < n; i++) residual[i] = gpu.check_value(i, loss.out_nodes.back());
        residual_from_step = false;
        sync_device();                                                  // solve.sync_device() of the tests
        return max_residual;
    }

//  solver_interface::compile, solver.hpp:303-349.
    void compile(const method m, const T dt) {
        switch (m) {
            case method::rk2: solver.reset(new work_item<T> (make_rk2_kernel(v, eq, dt, D))); break;
            case method::rk4: solver.reset(new work_item<T> (make_solver_kernel(v, eq, dt, D))); break;
            default: solver.reset(new work_item<T> (make_split_simplextic_kernel(v, eq, dt, D)));
        }
        graph::input_nodes<T> in;
        graph::map_nodes<T> set;
        to_lists<T> (*solver, in, set);
        solver_context.reset(new gpu::hip_context<T> (0));
        std::ostringstream source;
        jit::register_map registers;
        solver_context->create_header(source);
        add_kernel<T> (*solver_context, source, registers, "solver_kernel", in, solver->out_nodes, set, n);
        solver_context->compile(source.str(), {"solver_kernel"}, false);
        jit::texture1d_list tex1d;
        jit::texture2d_list tex2d;
        step_call = solver_context->create_kernel_call("solver_kernel", in, solver->out_nodes,
                                                       graph::shared_random_state<T> (), n, tex1d, tex2d);
    }

    void step() {                                                       // solver.hpp:382
        step_call();
        device_ahead = true;
        residual_from_step = true;
    }
    T residual_at(const size_t i) {                                     // check_residual, solver.hpp:392
        return solver_context->check_value(i, solver->out_nodes[0]);
    }

    std::string state() {
        sync_host();
        if (residual_from_step) {
            for (size_t i = 0; i < n; i++) residual[i] = residual_at(i);
        }
        std::ostringstream s;
        s << "[";
        for (size_t i = 0; i < n; i++) {
            s << (i ? ", [" : "[");
            for (int c = 0; c < 8; c++) s << g17(nodes[c]->evaluate().at(i)) << ", ";
            s << g17(residual[i]) << "]";
        }
        s << "]";
        return s.str();
    }
};

}  // namespace

int main() {
    report tape, device;
    run_all<tape_solver> (tape, "");
    run_all<hip_ray_solver> (device, "");
    const std::string a = tape.out.str(), b = device.out.str();
    printf("tape report %zu bytes, hip_context report %zu bytes\n", a.size(), b.size());
    printf("reference assertions hold: tape %s, hip_context %s\n", tape.all_passed ? "all" : "NOT ALL",
           device.all_passed ? "all" : "NOT ALL");

//  Compare scenario by scenario (split at the scenario headers).
    auto split = [] (const std::string &text) {
        std::vector<std::string> parts;
        size_t begin = 0;
        while (true) {
            const size_t next = text.find("\n \"", begin + 1);
            parts.push_back(text.substr(begin, next == std::string::npos ? std::string::npos : next - begin));
            if (next == std::string::npos) break;
            begin = next;
        }
        return parts;
    };
    const auto pa = split(a), pb = split(b);
    bool ok = tape.all_passed && device.all_passed && pa.size() == pb.size();
    size_t identical = 0;
    for (size_t i = 0; i < std::min(pa.size(), pb.size()); i++) {
        const std::string name = pa[i].substr(pa[i].find('"') + 1, pa[i].find('"', pa[i].find('"') + 1) - pa[i].find('"') - 1);
        const bool has_exp = name.find("gaussian_well") != std::string::npos || name.find("solver_test_cold_plasma") != std::string::npos;
        const bool same = pa[i] == pb[i];
        identical += same;
        printf("  %-36s %s\n", name.c_str(), same ? "identical" : (has_exp ? "differs (exp node: ocml vs glibc)" : "DIFFERS"));
        if (!same && !has_exp) {
            ok = false;
            printf("--- tape\n%s\n--- hip_context\n%s\n", pa[i].c_str(), pb[i].c_str());
        }
    }
    printf("%zu of %zu scenarios identical\n", identical, pa.size());
    printf(ok ? "PASS\n" : "FAIL\n");
    return ok ? 0 : 1;
}
