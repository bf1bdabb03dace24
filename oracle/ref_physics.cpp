// ---------------------------------------------------------------------------
// ref_physics.cpp — REFERENCE-BACKED ORACLE, part 2 (test infrastructure; builds to
// oracle/_ref/gf_ref_physics, git-ignored).
//
// Replays, on the reference's own graph layer + the tape interpreter, the scenarios of
//   graph_tests/solver_test.cpp:28-99     (rk2 / rk4 x simple / gaussian_well / cold_plasma
//                                          on gaussian_density: residual stays < 1e-30)
//   graph_tests/physics_test.cpp:24-576   (constant of motion, Bohm-Gross, light wave,
//                                          acoustic wave, O-mode cut-off, cold-plasma
//                                          reflection and cut-offs)
// checks each scenario's own assertion, writes the golden states as JSON on stdout, and
// exports every work item it used as GFIR into <directory> so that the HIP backend can be
// driven through the same scenarios (tests/test_gpu_physics.py).
//
// Usage: gf_ref_physics <directory for .gfir files>  > physics_golden.json
// ---------------------------------------------------------------------------
#include "ref_scenarios.hpp"

namespace {

//  solver::solver_interface over the tape interpreter: state columns on the host.
struct ray_solver {
    ray_variables<T> v;
    equilibrium_base<T> &eq;
    dispersion_interface<T> D;
    const size_t n;
    std::vector<std::vector<T>> cols;               // t w x y z kx ky kz
    std::vector<T> residual;
    std::map<int, std::unique_ptr<work_item<T>>> loss;
    std::unique_ptr<work_item<T>> solver;
    std::string name, directory;
    std::vector<size_t> newton_iterations;

    ray_solver(const std::string &scenario, const std::string &dir, equilibrium_base<T> &e,
               dispersion_function<T> f, const size_t num_rays = 1) :
    eq(e), D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, e, f), n(num_rays),
    cols(8, std::vector<T> (num_rays, 0.0)), residual(num_rays, 0.0), name(scenario), directory(dir) {}

    std::vector<T> &column(const char *key) {
        static const char *names[8] = {"t", "w", "x", "y", "z", "kx", "ky", "kz"};
        for (int i = 0; i < 8; i++) {
            if (std::string(key) == names[i]) return cols[i];
        }
        fprintf(stderr, "no column %s\n", key);
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

//  solver_interface::init(x, tolerance) (solver.hpp:254-274): var 0 w, 1 kx, 2 ky, 3 kz, 4 x, 5 y, 6 z.
    T init(const int var, const T tolerance = 1.0E-30, const size_t max_iterations = 1000) {
        static const char *names[7] = {"w", "kx", "ky", "kz", "x", "y", "z"};
        if (!loss.count(var)) {
            loss[var].reset(new work_item<T> (make_loss_kernel(v, D.D, var, static_cast<T> (1.0))));
            loss[var]->write_gfir("loss_kernel", directory + "/physics_" + name + "_loss_kernel_" + names[var] + "_f64.gfir");
        }
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
        solver->write_gfir("solver_kernel", directory + "/physics_" + name + "_solver_kernel_f64.gfir");
        fprintf(stderr, "%s solver_kernel ", name.c_str());
        solver->code.print_counts(stderr);
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

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: gf_ref_physics <directory for .gfir files>\n");
        return 2;
    }
    report r;
    run_all<ray_solver> (r, argv[1]);
    printf("{%s\n}\n", r.out.str().c_str());
    return r.all_passed ? 0 : 3;
}
