//------------------------------------------------------------------------------
///  @file korc_push.cpp
///  @brief Counterpart of graph_korc/xkorc.cpp:29-154 (run_korc<T>) on the C++ host mirror
///  (gf_workflow.hpp): the characteristic field at the magnetic axis
///  (efit::get_characteristic_field, equilibrium.hpp:1585-1615: a two-unknown Newton, then |B|),
///  the `initialize_gamma` pre-item (xkorc.cpp:66-85) and `num_steps` launches of the push
///  (`step`, xkorc.cpp:87-121).  Prints b0, the Newton iterations and particle 0 after the
///  steps listed on the command line (%.17g) for tests/test_gpu_workflows.py.
///
///  Usage: korc_push <workload directory> <f64|f32> <num_particles> <step> [<step> ...]
//------------------------------------------------------------------------------
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../gf_workflow.hpp"

template<typename T>
static int run_korc(const std::string &directory, const size_t num_particles, const std::vector<size_t> &report) {
    const std::string suffix = sizeof(T) == 8 ? "_f64.gfir" : "_f32.gfir";
    auto item = [&] (const char *name) { return gf::read_item(directory + "/korc_" + name + suffix); };

//  efit::get_characteristic_field: its own manager, one element.
    T b0;
    size_t axis_iterations;
    {
        const T x0 = 1.7, zero = 0.0;
        gf::workflow::manager<T> work(0);
        const std::vector<std::string> axis = {"axis_x", "axis_y", "axis_z"};
        const std::map<std::string, const T *> initial = {{"axis_x", &x0}, {"axis_y", &zero}, {"axis_z", &zero}};
        auto *newton = work.add_converge_item(item("axis_newton"), axis, {"axis_residual"}, 1, initial);
        auto *bmod = work.add_item(item("bmod_at_axis"), axis, {"axis_bmod"}, 1, initial);
        work.compile();
        newton->run();
        bmod->run();
        work.wait();
        b0 = work.check_value(0, "axis_bmod");
        axis_iterations = newton->iterations;
    }
    std::printf("b0 %.17g axis_iterations %zu\n", static_cast<double> (b0), axis_iterations);

//  run_korc: particles x = (1.7, 0, 0), u = (0, 0.99, 0.1) (xkorc.cpp:47-64).
    const std::vector<std::string> particle = {"x", "y", "z", "ux", "uy", "uz", "gamma"};
    const T start[7] = {1.7, 0.0, 0.0, 0.0, 0.99, 0.1, 0.0};
    std::map<std::string, std::vector<T>> host;
    std::map<std::string, const T *> initial;
    for (size_t k = 0; k < particle.size(); k++) {
        host[particle[k]].assign(num_particles, start[k]);
        initial[particle[k]] = host[particle[k]].data();
    }
    gf::workflow::manager<T> work(0);
    work.add_preitem(item("initialize_gamma"), {"ux", "uy", "uz", "gamma"}, {}, num_particles, initial);
    work.add_item(item("step"), particle, {}, num_particles, initial);
    work.compile();
    work.pre_run();
    size_t done = 0;
    for (const size_t target : report) {
        for (; done < target; done++) work.run();
        work.wait();
        std::printf("step %zu", target);
        for (auto &name : particle) {
            work.copy_to_host(name, host[name].data());
            bool uniform = true;
            for (const T v : host[name]) uniform = uniform && v == host[name][0];
            std::printf(" %.17g%s", static_cast<double> (host[name][0]), uniform ? "" : "(not uniform)");
        }
        std::printf("\n");
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: korc_push <workload directory> <f64|f32> <num_particles> <step> [<step> ...]\n");
        return 2;
    }
    std::vector<size_t> report;
    for (int i = 4; i < argc; i++) report.push_back(std::strtoull(argv[i], nullptr, 10));
    const size_t n = std::strtoull(argv[3], nullptr, 10);
    return std::strcmp(argv[2], "f32") ? run_korc<double> (argv[1], n, report) : run_korc<float> (argv[1], n, report);
}
