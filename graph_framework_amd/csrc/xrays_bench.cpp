//------------------------------------------------------------------------------
///  @file xrays_bench.cpp
///  @brief Counterpart of graph_benchmark/xrays_bench.cpp on the C++ host mirror
///  (graph_framework_amd/gf_workflow.hpp) of the reference's solver interface.
///
///  Same structure as the reference benchmark (xrays_bench.cpp:20-116): one host
///  thread per device (gfhip_max_concurrency(), jit.hpp:87), rays split into
///  contiguous batches (batch + 1 for the first NUM_RAYS % threads, :38-51),
///  every ray omega=500, x=2.5, kx=-600 (:62-71), Newton init (`solve.init(kx)`,
///  :89), compile (:92), NUM_TIMES steps (:96-100), sync_host (:101), and the
///  four timers printed as averages over threads (timing.hpp:115-123).
///
///  The work items come from GFIR files (the DAGs the reference front end builds
///  for this case); this program needs neither the reference headers nor Python.
///
///  Usage: xrays_bench <workload directory> [num_rays=100000] [num_times=1000] [output prefix [sub_steps]]
///  With an output prefix every device thread writes `<prefix><thread>.nc` as graph_driver/xrays.cpp does
///  (:1054-1083): the state after the Newton solve, then every `sub_steps` (default 100) steps through
///  solver_interface::write_step (solver.hpp:418-424) — the writer thread overlaps the following steps.
//------------------------------------------------------------------------------
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../gf_workflow.hpp"

namespace {

//  timing::measure_diagnostic_threaded (timing.hpp:67-150).
struct threaded_timer {
    std::string label;
    std::vector<double> seconds;
    std::vector<std::chrono::steady_clock::time_point> starts;
    threaded_timer(const std::string &l, const size_t threads) : label(l), seconds(threads, 0.0), starts(threads) {}
    void start_time(const size_t i) { starts[i] = std::chrono::steady_clock::now(); }
    void end_time(const size_t i) {
        seconds[i] = std::chrono::duration<double> (std::chrono::steady_clock::now() - starts[i]).count();
    }
    void print() const {
        double sum = 0.0, longest = 0.0;
        for (double s : seconds) { sum += s; longest = std::max(longest, s); }
        std::printf("%-14s: average %.6f s  max %.6f s\n", label.c_str(), sum/static_cast<double> (seconds.size()), longest);
    }
};

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        std::cerr << "usage: xrays_bench <workload directory> [num_rays] [num_times]" << std::endl;
        return 2;
    }
    const std::string directory = argv[1];
    const size_t num_rays = argc > 2 ? strtoull(argv[2], nullptr, 10) : 100000;
    const size_t num_times = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1000;
    const std::string output_prefix = argc > 4 ? argv[4] : "";
    const size_t sub_steps = argc > 5 ? strtoull(argv[5], nullptr, 10) : 100;

    const size_t devices = static_cast<size_t> (std::max(gfhip_max_concurrency(), 0));
    if (devices == 0) {
        std::cerr << "no HIP device" << std::endl;
        return 1;
    }
    std::vector<std::thread> threads(std::max<size_t> (std::min(devices, num_rays), 1));
    const size_t batch = num_rays/threads.size();
    const size_t extra = num_rays%threads.size();

    threaded_timer time_setup("Setup Time", threads.size()), time_init("Init Time", threads.size());
    threaded_timer time_compile("Compile Time", threads.size()), time_steps("Time Steps", threads.size());
    std::vector<double> final_x(threads.size()), final_kx(threads.size()), residual(threads.size());
    std::vector<size_t> iterations(threads.size());

    std::cout << gfhip_device_type() << ", " << threads.size() << " device thread(s), " << num_rays << " rays, "
              << num_times << " steps" << std::endl;

    for (size_t i = 0, ie = threads.size(); i < ie; i++) {
        threads[i] = std::thread([&, i] () {
            time_setup.start_time(i);
            const size_t local_num_rays = batch + (extra > i ? 1 : 0);
            size_t first = 0, last = 0;
            if (gfhip_shard_bounds(num_rays, threads.size(), i, &first, &last) || last - first != local_num_rays) {
                std::cerr << "shard split disagrees with xrays_bench.cpp:38-51" << std::endl;
                exit(1);
            }
            gf::solver::ray_solver<double> solve(directory, "", local_num_rays, i);
            solve.state["w"].assign(local_num_rays, 500.0);                 // xrays_bench.cpp:62-71
            solve.state["x"].assign(local_num_rays, 2.5);
            solve.state["kx"].assign(local_num_rays, -600.0);
            time_setup.end_time(i);

            time_init.start_time(i);
            solve.init("kx");
            iterations[i] = solve.newton_iterations;
            time_init.end_time(i);

            time_compile.start_time(i);
            solve.compile();
            time_compile.end_time(i);

            if (!output_prefix.empty()) {
                solve.open_result_file(output_prefix + std::to_string(i) + ".nc");
                solve.write_step();
            }
            time_steps.start_time(i);
            for (size_t j = 0; j < num_times; j++) {
                solve.step();
                if (!output_prefix.empty() && sub_steps && (j + 1)%sub_steps == 0) solve.write_step();
            }
            solve.sync_host();
            time_steps.end_time(i);
            solve.close_result_file();

            final_x[i] = solve.state["x"][local_num_rays - 1];
            final_kx[i] = solve.state["kx"][local_num_rays - 1];
            residual[i] = solve.check_residual(0);
        });
    }
    for (std::thread &t : threads) {
        t.join();
    }

    time_setup.print();
    time_init.print();
    time_compile.print();
    time_steps.print();
    double slowest = 0.0;
    for (double s : time_steps.seconds) slowest = std::max(slowest, s);
    std::printf("ray-steps/s (max over devices, step loop + sync_host): %.6e\n",
                static_cast<double> (num_rays)*static_cast<double> (num_times)/slowest);
    for (size_t i = 0; i < threads.size(); i++) {
        std::printf("device %zu: Newton iterations %zu, x = %.17g, kx = %.17g, residual = %.17g\n",
                    i, iterations[i], final_x[i], final_kx[i], residual[i]);
    }
    return 0;
}
