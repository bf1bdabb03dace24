// ---------------------------------------------------------------------------
// hip_context_demo.cpp — drives gpu::hip_context (graph_framework_amd/hip_context.hpp)
// exactly the way jit::context and workflow::manager drive a backend context, on the
// reference's REAL node graph for the xrays_bench case, and checks the device results
// against the tape interpreter (strict-IEEE evaluation of the same DAG).
//
// TEST INFRASTRUCTURE.  Builds to oracle/_ref/hip_context_demo (git-ignored) from the
// reference's expression-graph headers where they lie; jit.hpp / workflow.hpp themselves
// cannot be compiled here (LLVM JIT), so the few lines of theirs that touch the backend are
// restated below with citations.  The binary travels to the GPU box like any built file.
//
// Usage: hip_context_demo <tables.bin> <num_rays> <num_steps>
// ---------------------------------------------------------------------------
#include "hip_context_driver.hpp"

typedef double T;

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: hip_context_demo <tables.bin> <num_rays> <num_steps>\n");
        return 2;
    }
    const raw_tables raw(argv[1]);
    const size_t n = strtoull(argv[2], nullptr, 10);
    const size_t steps = strtoull(argv[3], nullptr, 10);

//  graph_benchmark/xrays_bench.cpp:53-86.
    efit<T> eq(raw);
    ray_variables<T> v;
    v.t->set(static_cast<T> (0.0));
    auto resize = [n] (leaf<T> node, const T value) {
        graph::variable_cast(node)->set(std::vector<T> (n, value));
    };
    resize(v.t, 0.0); resize(v.w, 500.0); resize(v.x, 2.5); resize(v.y, 0.0); resize(v.z, 0.0);
    resize(v.kx, -600.0); resize(v.ky, 0.0); resize(v.kz, 0.0);
    dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq);
    work_item<T> loss = make_loss_kernel(v, D.D, 1, static_cast<T> (1.0));
    work_item<T> solver = make_solver_kernel(v, eq, static_cast<T> (1.0E-3), D);

    gpu::hip_context<T> gpu(0);
    std::cout << gpu.device_type() << " devices: " << gpu.max_concurrency() << std::endl;
    std::ostringstream source;
    jit::register_map registers;
    gpu.create_header(source);

//  dispersion_interface::solve (dispersion.hpp:1452-1475) and solver_interface::compile
//  (solver.hpp:303-349) on one context.
    graph::input_nodes<T> loss_in, solver_in;
    graph::map_nodes<T> loss_set, solver_set;
    to_lists<T> (loss, loss_in, loss_set);
    to_lists<T> (solver, solver_in, solver_set);
    add_kernel<T> (gpu, source, registers, "loss_kernel", loss_in, loss.out_nodes, loss_set, n);
    add_kernel<T> (gpu, source, registers, "solver_kernel", solver_in, solver.out_nodes, solver_set, n);
    gpu.create_reduction(source, n);
    gpu.compile(source.str(), {"loss_kernel", "solver_kernel"}, true);

    jit::texture1d_list tex1d;
    jit::texture2d_list tex2d;
    auto loss_call = gpu.create_kernel_call("loss_kernel", loss_in, loss.out_nodes,
                                            graph::shared_random_state<T> (), n, tex1d, tex2d);
    auto max_kernel = gpu.create_max_call(loss.out_nodes.back(), loss_call);

//  converge_item::run, workflow.hpp:179-205.
    const T tolerance = 1.0E-30;
    const size_t max_iterations = 1000;
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
    std::vector<T> kx_host(n);
    gpu.copy_to_host(v.kx, kx_host.data());
    graph::variable_cast(v.kx)->set(kx_host);                          // dispersion.hpp:1472

    auto step = gpu.create_kernel_call("solver_kernel", solver_in, solver.out_nodes,
                                       graph::shared_random_state<T> (), n, tex1d, tex2d);
    for (size_t s = 0; s < steps; s++) {
        step();                                                        // solver.hpp:382
    }
    gpu.wait();

//  The same on the tape interpreter.
    std::vector<std::vector<T>> cols = {std::vector<T> (1, 0.0), std::vector<T> (1, 500.0), std::vector<T> (1, 2.5),
                                        std::vector<T> (1, 0.0), std::vector<T> (1, 0.0), std::vector<T> (1, -600.0),
                                        std::vector<T> (1, 0.0), std::vector<T> (1, 0.0)};
    std::vector<T *> pointers;
    for (auto &c : cols) pointers.push_back(c.data());
    T residual, last;
    const size_t ref_iterations = converge(loss, 1, pointers, &residual, tolerance, max_iterations, &last);
    for (size_t s = 0; s < steps; s++) {
        solver.run(1, pointers, {&residual});
    }

    bool ok = iterations == ref_iterations;
    printf("Newton iterations: device %zu, reference %zu\n", iterations, ref_iterations);
    const std::vector<leaf<T>> nodes = v.inputs();
    const char *names[] = {"t", "w", "x", "y", "z", "kx", "ky", "kz"};
    for (size_t i = 0; i < 8; i++) {
        const T device_first = gpu.check_value(0, nodes[i]);
        const T device_last = gpu.check_value(n - 1, nodes[i]);
        const T expected = cols[i][0];
        const bool match = std::abs(device_first - expected) <= 1.0E-6*std::abs(expected) &&
                           device_first == device_last;
        printf("  %-2s device % .17e reference % .17e %s\n", names[i], device_first, expected, match ? "ok" : "MISMATCH");
        ok = ok && match;
    }
    const T device_residual = gpu.check_value(0, solver.out_nodes[0]);
    printf("  residual device % .17e reference % .17e\n", device_residual, residual);
    ok = ok && std::abs(device_residual - residual) <= 1.0E-6*std::abs(residual);
    leaf<T> x_node = nodes[2];
    T *mirror = gpu.get_buffer(x_node);
    ok = ok && mirror[0] == gpu.check_value(0, nodes[2]);
    printf(ok ? "PASS\n" : "FAIL\n");
    return ok ? 0 : 1;
}
