//------------------------------------------------------------------------------
///  @file solver_check.cpp
///  @brief graph_tests/solver_test.cpp:28-60 on the C++ host mirror (gf_workflow.hpp): Newton
///  init for kx, then `steps` steps of an exported (integrator x dispersion relation x
///  equilibrium) combination; prints the state of every ray after the Newton solve and after
///  each step (%.17g) so that tests/test_gpu_physics.py can compare it with the record of the
///  reference's own graph layer (tests/golden/physics_golden.json).
///
///  Usage: solver_check <workload directory> <prefix> <omega> <kx> <ky> <kz> <steps>
//------------------------------------------------------------------------------
#include <cstdio>
#include <cstdlib>

#include "../gf_workflow.hpp"

int main(int argc, char **argv) {
    if (argc != 8) {
        std::fprintf(stderr, "usage: solver_check <workload directory> <prefix> <omega> <kx> <ky> <kz> <steps>\n");
        return 2;
    }
    const size_t steps = std::strtoull(argv[7], nullptr, 10);
    gf::solver::ray_solver<double> solve(argv[1], argv[2], 1);
    solve.state["w"][0] = std::atof(argv[3]);
    solve.state["kx"][0] = std::atof(argv[4]);
    solve.state["ky"][0] = std::atof(argv[5]);
    solve.state["kz"][0] = std::atof(argv[6]);

    const double tolerance = 1.0E-30;                                       // solver_test.cpp:113
    solve.init("kx", tolerance);
    solve.sync_host();
    auto print = [&] (const double residual) {
        for (const char *name : {"t", "w", "x", "y", "z", "kx", "ky", "kz"}) std::printf("%.17g ", solve.state[name][0]);
        std::printf("%.17g\n", residual);
    };
    std::printf("newton_iterations %zu\n", solve.newton_iterations);
    print(solve.newton_residual(0));
    solve.compile();
    bool holds = true;
    for (size_t i = 0; i < steps; i++) {
        solve.step();
        const double residual = solve.check_residual(0);
        holds = holds && std::abs(residual) < std::abs(tolerance);          // solver_test.cpp:54-58
        solve.sync_host();
        print(residual);
    }
    std::printf("holds %d\n", holds ? 1 : 0);
    return 0;
}
