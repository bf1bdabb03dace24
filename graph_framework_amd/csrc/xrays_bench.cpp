//------------------------------------------------------------------------------
///  @file xrays_bench.cpp
///  @brief Counterpart of graph_benchmark/xrays_bench.cpp on the C ABI of gf_hip.h.
///
///  Same structure as the reference benchmark (xrays_bench.cpp:20-116): one host
///  thread per device (gfhip_max_concurrency(), jit.hpp:87), rays split into
///  contiguous batches (batch + 1 for the first NUM_RAYS % threads, :38-51),
///  every ray omega=500, x=2.5, kx=-600 (:62-71), Newton init (`solve.init(kx)`,
///  :89), compile (:92), NUM_TIMES steps (:96-100), sync_host (:101), and the
///  four timers printed as averages over threads (timing.hpp:115-123).
///
///  The two work items come from GFIR files (the DAGs the reference front end
///  builds for this case); everything else goes through the C ABI, so this
///  program needs neither the reference headers nor Python.
///
///  Usage: xrays_bench <loss_kernel.gfir> <solver_kernel.gfir> [num_rays=100000] [num_times=1000]
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

#include "../../include/gf_hip.h"

namespace {

std::vector<char> read_file(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        std::cerr << "cannot open " << path << std::endl;
        exit(1);
    }
    return std::vector<char> ((std::istreambuf_iterator<char> (f)), std::istreambuf_iterator<char> ());
}

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

void check(gfhip_context *ctx, const int status, const char *what) {
    if (status) {
        std::cerr << what << ": " << gfhip_last_error(ctx) << std::endl;
        exit(1);
    }
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "usage: xrays_bench <loss_kernel.gfir> <solver_kernel.gfir> [num_rays] [num_times]" << std::endl;
        return 2;
    }
    const std::vector<char> loss = read_file(argv[1]);
    const std::vector<char> solver = read_file(argv[2]);
    const size_t num_rays = argc > 3 ? strtoull(argv[3], nullptr, 10) : 100000;
    const size_t num_times = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1000;

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
            const std::vector<double> t(local_num_rays, 0.0), w(local_num_rays, 500.0), x(local_num_rays, 2.5);
            const std::vector<double> y(local_num_rays, 0.0), z(local_num_rays, 0.0), kx(local_num_rays, -600.0);
            const std::vector<double> ky(local_num_rays, 0.0), kz(local_num_rays, 0.0);
            gfhip_context *ctx = gfhip_create_context(static_cast<int> (i), nullptr);
            if (!ctx) {
                std::cerr << gfhip_last_error(nullptr) << std::endl;
                exit(1);
            }
            time_setup.end_time(i);

//  Buffer keys: t w x y z kx ky kz (solver.hpp:304-313), then the two residual outputs.
            const uint64_t keys[8] = {1, 2, 3, 4, 5, 6, 7, 8};
            const void *initial[8] = {t.data(), w.data(), x.data(), y.data(), z.data(), kx.data(), ky.data(), kz.data()};
            const uint64_t newton_residual = 9, step_residual = 10;

            time_init.start_time(i);
            gfhip_kernel *newton = gfhip_add_kernel(ctx, loss.data(), loss.size(), local_num_rays);
            if (!newton) check(ctx, 1, "gfhip_add_kernel(loss_kernel)");
            check(ctx, gfhip_compile(ctx), "gfhip_compile");
            check(ctx, gfhip_create_kernel_call(newton, keys, initial, &newton_residual), "gfhip_create_kernel_call");
            double last_max;
            check(ctx, gfhip_converge(newton, 1.0E-30, 1000, &iterations[i], &last_max), "gfhip_converge");
            time_init.end_time(i);

            time_compile.start_time(i);
            gfhip_kernel *step = gfhip_add_kernel(ctx, solver.data(), solver.size(), local_num_rays);
            if (!step) check(ctx, 1, "gfhip_add_kernel(solver_kernel)");
            check(ctx, gfhip_compile(ctx), "gfhip_compile");
            check(ctx, gfhip_create_kernel_call(step, keys, nullptr, &step_residual), "gfhip_create_kernel_call");
            time_compile.end_time(i);

            time_steps.start_time(i);
            for (size_t j = 0; j < num_times; j++) {
                check(ctx, gfhip_run(step, 1), "gfhip_run");
            }
            std::vector<std::vector<double>> host(8, std::vector<double> (local_num_rays));
            for (size_t k = 0; k < 8; k++) {                        // solve.sync_host(), solver.hpp:368-377
                check(ctx, gfhip_copy_to_host(ctx, keys[k], host[k].data()), "gfhip_copy_to_host");
            }
            time_steps.end_time(i);

            final_x[i] = host[2][local_num_rays - 1];
            final_kx[i] = host[5][local_num_rays - 1];
            check(ctx, gfhip_check_value(ctx, step_residual, 0, &residual[i]), "gfhip_check_value");
            gfhip_destroy_context(ctx);
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
