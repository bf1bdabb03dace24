// ---------------------------------------------------------------------------
// ref_vmec.cpp — TEST INFRASTRUCTURE (oracle/Makefile target _ref/gf_ref_vmec; needs /root/reference).
// The VMEC equilibrium (equilibrium.hpp:1868-2650) on the reference graph layer: work items of the
// (cold_plasma x rk4 x vmec) combination as GFIR, and their values on the tape.
//
//   gf_ref_vmec <vmec.bin> info <modes>                               node counts and build times
//   gf_ref_vmec <vmec.bin> field <modes> <in: s u v> <out: bx by bz x y z ne te> [gfir]
//   gf_ref_vmec <vmec.bin> trace <modes> <in: t w s u v ks ku kv> <out> <dt> <steps> <save_every> <newton var|-1> [dir]
// ---------------------------------------------------------------------------
#include <chrono>

#include "ref_builders.hpp"

static std::vector<std::vector<double>> read_columns(const char *path, const size_t cols, size_t &n) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    uint64_t n64;
    if (fread(&n64, 8, 1, f) != 1) { fprintf(stderr, "short read\n"); exit(1); }
    n = n64;
    std::vector<std::vector<double>> out(cols, std::vector<double> (n));
    for (auto &c : out) {
        if (fread(c.data(), 8, n, f) != n) { fprintf(stderr, "short read\n"); exit(1); }
    }
    fclose(f);
    return out;
}

static void write_columns(const char *path, const std::vector<std::vector<double>> &cols) {
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); exit(1); }
    const uint64_t n = cols.empty() ? 0 : cols[0].size();
    fwrite(&n, 8, 1, f);
    for (auto &c : cols) fwrite(c.data(), 8, n, f);
    fclose(f);
}

static double seconds() {
    return std::chrono::duration<double> (std::chrono::steady_clock::now().time_since_epoch()).count();
}

typedef double T;

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: gf_ref_vmec <vmec.bin> info|field|trace <modes> ...\n");
        return 2;
    }
    const raw_vmec raw(argv[1]);
    const std::string cmd = argv[2];
    const size_t modes = strtoull(argv[3], nullptr, 10);
    if (cmd == "info") {
        double t0 = seconds();
        vmec<T> eq(raw, modes);
        ray_variables<T> v;
        dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq);
        fprintf(stderr, "dispersion_interface built in %.1f s\n", seconds() - t0);
        t0 = seconds();
        work_item<T> loss = make_loss_kernel(v, D.D, 1, static_cast<T> (1.0));
        fprintf(stderr, "loss_kernel: %zu statements (%.1f s)\n", loss.code.code.size(), seconds() - t0);
        loss.code.print_counts(stderr);
        t0 = seconds();
        work_item<T> solver = make_solver_kernel(v, eq, static_cast<T> (1.0e-3), D);
        fprintf(stderr, "solver_kernel: %zu statements (%.1f s)\n", solver.code.code.size(), seconds() - t0);
        solver.code.print_counts(stderr);
        return 0;
    }
    if (cmd == "field" && argc >= 6) {
        vmec<T> eq(raw, modes);
        size_t n;
        auto cols = read_columns(argv[4], 3, n);
        auto s = graph::variable<T> (1, "s");
        auto u = graph::variable<T> (1, "u");
        auto w = graph::variable<T> (1, "v");
        auto b = eq.get_magnetic_field(s, u, w);
        eq.set_cache(s, u, w);
        work_item<T> item({s, u, w}, {b->get_x(), b->get_y(), b->get_z(), eq.x_cache, eq.y_cache, eq.z_cache,
                                      eq.get_electron_density(s, u, w), eq.get_electron_temperature(s, u, w)}, {});
        item.code.print_counts(stderr);
        if (argc > 6) item.write_gfir("vmec_field_kernel", argv[6]);
        std::vector<std::vector<T>> outs(8, std::vector<T> (n));
        std::vector<T *> in, out;
        for (auto &c : cols) in.push_back(c.data());
        for (auto &c : outs) out.push_back(c.data());
        item.run(n, in, out);
        write_columns(argv[5], outs);
        return 0;
    }
//  The (cold_plasma x rk4 x vmec) combination of graph_driver/xrays.cpp:382: Newton init of one unknown,
//  then RK4 steps in flux coordinates (x, y, z = s, u, v; k = k_s e^s + k_u e^u + k_v e^v), as cmd_trace of
//  ref_driver.cpp does on EFIT.  out: per saved step 9 columns (8 state + residual); with `dir` the two items
//  are written as GFIR (vmec<modes>_loss_kernel_<unknown>, vmec<modes>_solver_kernel).
    if (cmd == "trace" && argc >= 10) {
        vmec<T> eq(raw, modes);
        size_t n;
        auto cols = read_columns(argv[4], 8, n);
        const double dt = atof(argv[6]);
        const size_t num_steps = strtoull(argv[7], nullptr, 10), save_every = strtoull(argv[8], nullptr, 10);
        const int newton_var = atoi(argv[9]);
        ray_variables<T> v;
        double t0 = seconds();
        dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq);
        std::vector<T> residual(n, 0);
        std::vector<T *> in;
        for (auto &c : cols) in.push_back(c.data());
        const char *unknown_names[7] = {"w", "kx", "ky", "kz", "x", "y", "z"};
        const std::string prefix = "vmec" + std::to_string(modes) + "_";
        if (newton_var >= 0) {
            work_item<T> loss = make_loss_kernel(v, D.D, newton_var, static_cast<T> (1.0));
            fprintf(stderr, "loss_kernel ");
            loss.code.print_counts(stderr);
            if (argc > 10) loss.write_gfir("loss_kernel", (std::string(argv[10]) + "/" + prefix + "loss_kernel_" + unknown_names[newton_var] + "_f64.gfir").c_str());
            T last;
            const size_t it = converge(loss, n, in, residual.data(), static_cast<T> (1.0E-30), 1000, &last);
            fprintf(stderr, "{\"newton_iterations\": %zu, \"newton_last_max\": %.17g}\n", it, static_cast<double> (last));
        }
        work_item<T> solver = make_solver_kernel(v, eq, static_cast<T> (dt), D);
        fprintf(stderr, "solver_kernel (built in %.1f s) ", seconds() - t0);
        solver.code.print_counts(stderr);
        if (argc > 10) solver.write_gfir("solver_kernel", (std::string(argv[10]) + "/" + prefix + "solver_kernel_f64.gfir").c_str());
        std::vector<std::vector<double>> record;
        auto save = [&] () {
            for (size_t c = 0; c < 8; c++) record.push_back(cols[c]);
            record.emplace_back(residual.begin(), residual.end());
        };
        save();
        for (size_t step = 1; step <= num_steps; step++) {
            solver.run(n, in, {residual.data()});
            if (save_every && (step%save_every == 0 || step == num_steps)) save();
        }
        if (!save_every) save();
        write_columns(argv[5], record);
        return 0;
    }
    fprintf(stderr, "bad command\n");
    return 2;
}
