//------------------------------------------------------------------------------
///  @file korc_push.cpp
///  @brief Counterpart of graph_korc/xkorc.cpp:10-164 (run_korc<T>) on the C++ host mirror
///  (gf_workflow.hpp).
///
///  Same structure as the reference driver: one host thread per device
///  (threads = max(min(max_concurrency, num_particles), 1), xkorc.cpp:16-18), the particles
///  split into `batch + (extra > thread_number ? 1 : 0)` per thread (:20-25), and in every thread
///  the characteristic field at the magnetic axis (efit::get_characteristic_field(thread_number),
///  equilibrium.hpp:1585-1615: a two-unknown Newton, then |B|; :29-31), its own
///  workflow::manager(thread_number) (:74) with the `initialize_gamma` pre-item (:66-85) and the
///  push (`step`, :87-121), `pre_run()` and the step loop (:144-154).
///
///  Prints b0 and the Newton iterations of thread 0, then after each step listed on the command
///  line the particle with global index 0 (%.17g; "(not uniform)" if a shard's particles differ
///  although they started identical) — the lines tests/test_gpu_workflows.py reads — and, with
///  more than one thread or `--perturb`, one line per probed global index (first, middle, last
///  and the two particles either side of every shard boundary) so that a sharded run can be
///  compared with an unsharded one.
///
///  Usage: korc_push [--threads N] [--perturb] <workload directory> <f64|f32> <num_particles> <step> [<step> ...]
///    --threads N   host threads instead of one per device; thread i uses device i % devices
///                  (rehearsal of the multi-device split on a box with fewer devices)
///    --perturb     particle g starts with uz = 0.1 + 1e-3*(g % 997)/997 instead of 0.1, so that
///                  every particle's trajectory depends on its GLOBAL index
//------------------------------------------------------------------------------
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "../gf_workflow.hpp"

namespace {

struct probe_record {
    size_t step, global;
    double value[7];
    bool uniform;
};

template<typename T>
int run_korc(const std::string &directory, const size_t num_particles, const std::vector<size_t> &report,
             const size_t requested_threads, const bool perturb) {
    const std::string suffix = sizeof(T) == 8 ? "_f64.gfir" : "_f32.gfir";
    auto item = [&] (const char *name) { return gf::read_item(directory + "/korc_" + name + suffix); };

    const size_t devices = static_cast<size_t> (std::max(gfhip_max_concurrency(), 0));
    if (devices == 0) {
        std::fprintf(stderr, "no HIP device\n");
        return 1;
    }
//  xkorc.cpp:16-18
    const size_t concurrency = requested_threads ? requested_threads : devices;
    std::vector<std::thread> threads(std::max(std::min(concurrency, num_particles), static_cast<size_t> (1)));
//  xkorc.cpp:20-21
    const size_t batch = num_particles/threads.size();
    const size_t extra = num_particles%threads.size();

//  Global indices to report: the ends, the middle and both sides of every shard boundary.
    std::vector<size_t> probes = {0};
    if (num_particles > 1) {
        probes.push_back(num_particles/2);
        probes.push_back(num_particles - 1);
        for (size_t i = 1, begin = 0; i < threads.size(); i++) {
            begin += batch + (extra > i - 1 ? 1 : 0);
            if (begin > 0 && begin < num_particles) {
                probes.push_back(begin - 1);
                probes.push_back(begin);
            }
        }
    }
    std::sort(probes.begin(), probes.end());
    probes.erase(std::unique(probes.begin(), probes.end()), probes.end());

    std::vector<T> b0(threads.size());
    std::vector<size_t> axis_iterations(threads.size());
    std::vector<probe_record> records;
    std::mutex records_lock;

    for (size_t i = 0, ie = threads.size(); i < ie; i++) {
        threads[i] = std::thread([&, i] () {
            const size_t thread_number = i;
            const size_t device = thread_number%devices;
//  xkorc.cpp:24
            const size_t local_num_particles = batch + (extra > thread_number ? 1 : 0);
            size_t first = 0, last = 0;
            if (gfhip_shard_bounds(num_particles, threads.size(), thread_number, &first, &last) ||
                last - first != local_num_particles) {
                std::fprintf(stderr, "shard split disagrees with xkorc.cpp:20-25\n");
                exit(1);
            }

//  efit::get_characteristic_field(thread_number): its own manager, one element (xkorc.cpp:29-31).
            {
                const T x0 = 1.7, zero = 0.0;
                gf::workflow::manager<T> work(device);
                const std::vector<std::string> axis = {"axis_x", "axis_y", "axis_z"};
                const std::map<std::string, const T *> initial = {{"axis_x", &x0}, {"axis_y", &zero}, {"axis_z", &zero}};
                auto *newton = work.add_converge_item(item("axis_newton"), axis, {"axis_residual"}, 1, initial);
                auto *bmod = work.add_item(item("bmod_at_axis"), axis, {"axis_bmod"}, 1, initial);
                work.compile();
                newton->run();
                bmod->run();
                work.wait();
                b0[thread_number] = work.check_value(0, "axis_bmod");
                axis_iterations[thread_number] = newton->iterations;
            }

//  Particles x = (1.7, 0, 0), u = (0, 0.99, 0.1) (xkorc.cpp:47-64).
            const std::vector<std::string> particle = {"x", "y", "z", "ux", "uy", "uz", "gamma"};
            const T start[7] = {1.7, 0.0, 0.0, 0.0, 0.99, 0.1, 0.0};
            std::map<std::string, std::vector<T>> host;
            std::map<std::string, const T *> initial;
            for (size_t k = 0; k < particle.size(); k++) {
                host[particle[k]].assign(local_num_particles, start[k]);
            }
            if (perturb) {
                for (size_t p = 0; p < local_num_particles; p++) {
                    host["uz"][p] = static_cast<T> (0.1 + 1.0E-3*static_cast<double> ((first + p)%997)/997.0);
                }
            }
            for (size_t k = 0; k < particle.size(); k++) {
                initial[particle[k]] = host[particle[k]].data();
            }
//  xkorc.cpp:74: workflow::manager<T> work(thread_number)
            gf::workflow::manager<T> work(device);
            work.add_preitem(item("initialize_gamma"), {"ux", "uy", "uz", "gamma"}, {}, local_num_particles, initial);
            work.add_item(item("step"), particle, {}, local_num_particles, initial);
            work.compile();
            work.pre_run();
            size_t done = 0;
            for (const size_t target : report) {
                for (; done < target; done++) work.run();
                work.wait();
                bool uniform = true;
                for (auto &name : particle) {
                    work.copy_to_host(name, host[name].data());
                    for (const T v : host[name]) uniform = uniform && v == host[name][0];
                }
                std::lock_guard<std::mutex> hold(records_lock);
                for (const size_t g : probes) {
                    if (g < first || g >= first + local_num_particles) continue;
                    probe_record r;
                    r.step = target;
                    r.global = g;
                    r.uniform = uniform;
                    for (size_t k = 0; k < particle.size(); k++) r.value[k] = static_cast<double> (host[particle[k]][g - first]);
                    records.push_back(r);
                }
            }
        });
    }
    for (std::thread &t : threads) {
        t.join();
    }

    for (size_t i = 1; i < threads.size(); i++) {
        if (b0[i] != b0[0] || axis_iterations[i] != axis_iterations[0]) {
            std::fprintf(stderr, "device threads disagree on the characteristic field\n");
            return 1;
        }
    }
    std::printf("b0 %.17g axis_iterations %zu threads %zu devices %zu\n", static_cast<double> (b0[0]), axis_iterations[0],
                threads.size(), devices);
    std::sort(records.begin(), records.end(), [] (const probe_record &a, const probe_record &b) {
        return a.step != b.step ? a.step < b.step : a.global < b.global;
    });
    const bool listing = threads.size() > 1 || perturb;
    for (const size_t target : report) {
        bool uniform = true;
        for (auto &r : records) {
            if (r.step == target) uniform = uniform && (r.uniform || perturb);
        }
        for (auto &r : records) {
            if (r.step != target || r.global != 0) continue;
            std::printf("step %zu", target);
            for (size_t k = 0; k < 7; k++) std::printf(" %.17g%s", r.value[k], uniform ? "" : "(not uniform)");
            std::printf("\n");
        }
        for (auto &r : records) {
            if (!listing || r.step != target) continue;
            std::printf("probe %zu %zu", target, r.global);
            for (size_t k = 0; k < 7; k++) std::printf(" %.17g", r.value[k]);
            std::printf("\n");
        }
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    size_t threads = 0;
    bool perturb = false;
    int at = 1;
    while (at < argc && !std::strncmp(argv[at], "--", 2)) {
        if (!std::strcmp(argv[at], "--threads") && at + 1 < argc) {
            threads = std::strtoull(argv[at + 1], nullptr, 10);
            at += 2;
        } else if (!std::strcmp(argv[at], "--perturb")) {
            perturb = true;
            at += 1;
        } else {
            break;
        }
    }
    if (argc - at < 4) {
        std::fprintf(stderr, "usage: korc_push [--threads N] [--perturb] <workload directory> <f64|f32> <num_particles> <step> [<step> ...]\n");
        return 2;
    }
    std::vector<size_t> report;
    for (int i = at + 3; i < argc; i++) report.push_back(std::strtoull(argv[i], nullptr, 10));
    const size_t n = std::strtoull(argv[at + 2], nullptr, 10);
    return std::strcmp(argv[at + 1], "f32") ? run_korc<double> (argv[at], n, report, threads, perturb)
                                            : run_korc<float> (argv[at], n, report, threads, perturb);
}
