// ---------------------------------------------------------------------------
// ref_driver.cpp — REFERENCE-BACKED ORACLE (test infrastructure, never shipped
// in the product path).  Builds to oracle/_ref/gf_ref (git-ignored).
//
// What is real reference code here: the whole expression-graph layer, compiled
// from the headers where they lie under /root/reference/graph_framework
// (node.hpp, arithmetic.hpp, math.hpp, trigonometry.hpp, piecewise.hpp,
// vector.hpp, backend.hpp, register.hpp) — i.e. the node factories with their
// algebraic reduce(), the hash-consing caches and the symbolic df().  These
// headers need nothing outside the standard library.
//
// What cannot be built here (DESIGN.md "oracle"): equilibrium.hpp, dispersion.hpp,
// solver.hpp, workflow.hpp and jit.hpp include NetCDF-C and the in-process
// Clang/LLVM JIT (cpu_context.hpp:21-38), neither of which exists in this image.
// The graph CONSTRUCTION those headers perform for the hot path is therefore
// restated below against the reference's own node API, citing file:line, and
// the resulting DAGs are evaluated by a small tape interpreter that performs,
// for every node type, exactly the arithmetic the node's compile() method
// emits (one IEEE operation per node, real fma for fma_node, integer pow as
// repeated multiplication, gathers with the compile_index clamp).  Evaluation
// order does not change IEEE results of a DAG, so this is the reference kernel
// under strict (non-fast-math) floating point.
//
// Usage: gf_ref <tables.bin> <command> ...   (see main()).
// ---------------------------------------------------------------------------
#include "ref_builders.hpp"

// ---------------------------------------------------------------------------
// File helpers: SoA blocks of doubles.  in: uint64 n, then `cols` columns of n.
// ---------------------------------------------------------------------------
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

template<typename T>
static std::vector<std::vector<T>> convert(const std::vector<std::vector<double>> &in) {
    std::vector<std::vector<T>> out;
    for (auto &c : in) out.emplace_back(c.begin(), c.end());
    return out;
}
template<typename T>
static std::vector<std::vector<double>> to_double(const std::vector<std::vector<T>> &in) {
    std::vector<std::vector<double>> out;
    for (auto &c : in) out.emplace_back(c.begin(), c.end());
    return out;
}
template<typename T>
static std::vector<T *> pointers(std::vector<std::vector<T>> &cols, const size_t first, const size_t count) {
    std::vector<T *> p;
    for (size_t i = first; i < first + count; i++) p.push_back(cols[i].data());
    return p;
}

// ---------------------------------------------------------------------------
// Commands.
// ---------------------------------------------------------------------------
//  efit_test <in: x y z> <out: bx by bz ne te div>     graph_tests/efit_test.cpp:132-162
template<typename T>
static int cmd_efit_test(const raw_tables &raw, const char *in_path, const char *out_path) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 3, n));
    auto x = graph::variable<T> (1, "x");
    auto y = graph::variable<T> (1, "y");
    auto z = graph::variable<T> (1, "z");
    auto bvec = eq.get_magnetic_field(x, y, z);
    auto ne = eq.get_electron_density(x, y, z);
    auto te = eq.get_electron_temperature(x, y, z);
    auto div = bvec->get_x()->df(x) + bvec->get_y()->df(y) + bvec->get_z()->df(z);
    work_item<T> item({x, y, z}, {bvec->get_x(), bvec->get_y(), bvec->get_z(), ne, te, div}, {});
    item.code.print_counts(stderr);
    std::vector<std::vector<T>> outs(6, std::vector<T> (n));
    item.run(n, pointers(cols, 0, 3), pointers(outs, 0, 6));
    write_columns(out_path, to_double(outs));
    return 0;
}

//  dispersion <in: t w x y z kx ky kz> <out: D dDdw dDdkx dDdky dDdkz dDdx dDdy dDdz>
template<typename T>
static int cmd_dispersion(const raw_tables &raw, const char *in_path, const char *out_path) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 8, n));
    ray_variables<T> v;
    dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq);
    work_item<T> item(v.inputs(), {D.D, D.dDdw, D.dDdkx, D.dDdky, D.dDdkz, D.dDdx, D.dDdy, D.dDdz}, {});
    item.code.print_counts(stderr);
    std::vector<std::vector<T>> outs(8, std::vector<T> (n));
    item.run(n, pointers(cols, 0, 8), pointers(outs, 0, 8));
    write_columns(out_path, to_double(outs));
    return 0;
}

//  trace <in: t w x y z kx ky kz> <out> <dt> <num_steps> <save_every> <newton var or -1>
//  Newton init on the whole batch (per-shard global max, workflow.hpp:179-205),
//  then RK4 steps.  out: for each saved step 9 columns (8 state + residual).
//  First saved record = state after Newton (residual = D*D of the last Newton pass).
template<typename T>
static int cmd_trace(const raw_tables &raw, const char *in_path, const char *out_path,
                     const double dt, const size_t num_steps, const size_t save_every, const int newton_var,
                     const bool ordinary = false) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 8, n));
    ray_variables<T> v;
    dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq,
                              ordinary ? ordinary_wave_D<T> : cold_plasma_D<T>);
    std::vector<T> residual(n, 0);

    if (newton_var >= 0) {
        work_item<T> loss = make_loss_kernel(v, D.D, newton_var, static_cast<T> (1.0));
        fprintf(stderr, "loss_kernel ");
        loss.code.print_counts(stderr);
        T last;
        const size_t it = converge(loss, n, pointers(cols, 0, 8), residual.data(),
                                   static_cast<T> (1.0E-30), 1000, &last);
        fprintf(stderr, "{\"newton_iterations\": %zu, \"newton_last_max\": %.17g}\n", it, static_cast<double> (last));
    }

    work_item<T> solver = make_solver_kernel(v, eq, static_cast<T> (dt), D);
    fprintf(stderr, "solver_kernel ");
    solver.code.print_counts(stderr);

    std::vector<std::vector<double>> record;
    auto save = [&] () {
        for (size_t c = 0; c < 8; c++) record.emplace_back(cols[c].begin(), cols[c].end());
        record.emplace_back(residual.begin(), residual.end());
    };
    save();
    for (size_t s = 1; s <= num_steps; s++) {
        solver.run(n, pointers(cols, 0, 8), {residual.data()});
        if (save_every && (s%save_every == 0 || s == num_steps)) save();
    }
    if (!save_every) save();
    write_columns(out_path, record);
    return 0;
}

//  solver::adaptive_rk4 on the EFIT cold-plasma ray (graph_driver/xrays.cpp:353 `--solver=adaptive_rk4`):
//  Newton init of kx, then per step the converge item on (dt, lambda) and the RK4 step with that dt,
//  as workflow::manager::run executes the two items (workflow.hpp:359-363).  Columns: the 8 state
//  arrays, dt, lambda; records: those 10 + residual + the converge item's iteration count.
//  With `export_dir` the two items are also written as GFIR.
template<typename T>
static int cmd_trace_adaptive(const raw_tables &raw, const char *in_path, const char *out_path,
                              const size_t num_steps, const char *export_dir, const char *suffix) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 10, n));
    ray_variables<T> v;
    dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq, cold_plasma_D<T>);
    std::vector<T> residual(n, 0), loss_value(n, 0);
    work_item<T> newton_kx = make_loss_kernel(v, D.D, 1, static_cast<T> (1.0));
    T last;
    const size_t it = converge(newton_kx, n, pointers(cols, 0, 8), residual.data(), static_cast<T> (1.0E-30), 1000, &last);
    fprintf(stderr, "{\"newton_iterations\": %zu}\n", it);

    adaptive_rk4_items<T> items = make_adaptive_rk4<T> (v, eq, D);
    fprintf(stderr, "adaptive loss_kernel ");
    items.loss->code.print_counts(stderr);
    fprintf(stderr, "adaptive solver_kernel ");
    items.solver->code.print_counts(stderr);
    if (export_dir && export_dir[0]) {
        items.loss->write_gfir("adaptive_loss_kernel", std::string(export_dir) + "/adaptive_rk4_loss_kernel_" + suffix + ".gfir");
        items.solver->write_gfir("adaptive_solver_kernel", std::string(export_dir) + "/adaptive_rk4_solver_kernel_" + suffix + ".gfir");
    }

//  The first converge loop pass by pass, on a copy of the state: (dt, lambda, loss^2) after each
//  pass — what the device has to reproduce before everything turns NaN (<out>.passes).
    {
        auto copy = cols;
        std::vector<std::vector<double>> passes;
        for (int pass = 0; pass < 24; pass++) {
            items.loss->run(n, pointers(copy, 0, 10), {loss_value.data()});
            passes.emplace_back(copy[8].begin(), copy[8].end());
            passes.emplace_back(copy[9].begin(), copy[9].end());
            passes.emplace_back(loss_value.begin(), loss_value.end());
        }
        write_columns((std::string(out_path) + ".passes").c_str(), passes);
    }

    std::vector<std::vector<double>> record;
    std::vector<double> iterations(n, 0.0);
    auto save = [&] () {
        for (size_t c = 0; c < 10; c++) record.emplace_back(cols[c].begin(), cols[c].end());
        record.emplace_back(residual.begin(), residual.end());
        record.emplace_back(iterations);
    };
    save();
    for (size_t s = 1; s <= num_steps; s++) {
        const size_t used = converge(*items.loss, n, pointers(cols, 0, 10), loss_value.data(), static_cast<T> (1.0E-30), 1000, &last);
        std::fill(iterations.begin(), iterations.end(), static_cast<double> (used));
        items.solver->run(n, pointers(cols, 0, 9), {residual.data()});
        save();
    }
    write_columns(out_path, record);
    return 0;
}

//  graph_korc/xkorc.cpp:29-121 + efit::get_characteristic_field equilibrium.hpp:1585-1615.
template<typename T>
struct korc_graphs {
    efit<T> eq;
    T b0_value = 0, larmor_value = 0;
    size_t axis_iterations = 0;
    T axis_x = 0, axis_z = 0;
    std::unique_ptr<work_item<T>> axis_newton, bmod, init, step;

    explicit korc_graphs(const raw_tables &raw) : eq(raw) {
//  get_characteristic_field: two-unknown Newton (step 0.1) to the magnetic axis, then |B|.
        auto x_axis = graph::variable<T> (1, "x");
        auto y_axis = graph::variable<T> (1, "y");
        auto z_axis = graph::variable<T> (1, "z");
        auto b_axis = eq.get_magnetic_field(x_axis, y_axis, z_axis);
        auto b_mod = b_axis->length();
        auto func = (eq.psi_cache - eq.psimin)/eq.dpsi;
        const T newton_step = static_cast<T> (0.1);
        axis_newton.reset(new work_item<T> ({x_axis, y_axis, z_axis}, {func*func},
                                            {{x_axis - newton_step*func/func->df(x_axis), x_axis},
                                             {z_axis - newton_step*func/func->df(z_axis), z_axis}}));
        bmod.reset(new work_item<T> ({x_axis, y_axis, z_axis}, {b_mod}, {}));
        std::vector<std::vector<T>> axis = {{static_cast<T> (1.7)}, {static_cast<T> (0.0)}, {static_cast<T> (0.0)}};
        T res, last;
        axis_iterations = converge(*axis_newton, 1, pointers(axis, 0, 3), &res, static_cast<T> (1.0E-30), 1000, &last);
        bmod->run(1, pointers(axis, 0, 3), {&b0_value});
        axis_x = axis[0][0];
        axis_z = axis[2][0];

        auto b0 = graph::constant<T> (b0_value);
        const T q = 1.602176634E-19;
        const T me = 9.1093837139E-31;
        const T c = 299792458.0;
        auto gryo_period = me/(q*b0);
        auto larmor_radius = c*gryo_period;
        larmor_value = larmor_radius->evaluate().at(0);

        auto ux = graph::variable<T> (1, "u_{x}");
        auto uy = graph::variable<T> (1, "u_{y}");
        auto uz = graph::variable<T> (1, "u_{z}");
        auto x = graph::variable<T> (1, "x");
        auto y = graph::variable<T> (1, "y");
        auto z = graph::variable<T> (1, "z");
        auto pos = graph::vector(x, y, z);
        auto u_vec = graph::vector(ux, uy, uz);
        auto gamma = graph::variable<T> (1, "\\gamma");
        auto dt = graph::constant<T> (0.5);

        auto gamma_init = 1.0/graph::sqrt(1.0 - u_vec->dot(u_vec));
        auto u_init = gamma_init*u_vec;
        auto b_vec = eq.get_magnetic_field(pos->get_x(), pos->get_y(), pos->get_z())/b0;

        init.reset(new work_item<T> ({ux, uy, uz, gamma}, {},
                                     {{u_init->get_x(), ux}, {u_init->get_y(), uy}, {u_init->get_z(), uz}, {gamma_init, gamma}}));

        auto u_prime = u_vec - dt*u_vec->cross(b_vec)/(2.0*gamma);
        auto tau = -0.5*dt*b_vec;
        auto tau_sq = tau->dot(tau);
        auto speed_sq = u_prime->dot(u_prime);
        auto sigma = 1.0 + speed_sq - tau_sq;
        auto ustar = u_prime->dot(tau);
        auto gamma_next = graph::sqrt(0.5*(sigma + graph::sqrt(sigma*sigma + 4.0*(tau_sq + ustar*ustar))));
        auto t = tau/gamma_next;
        auto s = 1.0 + t->dot(t);
        auto u_prime_dot_t = u_prime->dot(t);
        auto u_next = (u_prime + u_prime_dot_t*t + u_prime->cross(t))/s;
        auto pos_next = pos + larmor_radius*dt*u_next/gamma_next;

        step.reset(new work_item<T> ({x, y, z, ux, uy, uz, gamma}, {},
                                     {{pos_next->get_x(), x}, {pos_next->get_y(), y}, {pos_next->get_z(), z},
                                      {u_next->get_x(), ux}, {u_next->get_y(), uy}, {u_next->get_z(), uz},
                                      {gamma_next, gamma}}));
    }

    void print_info() const {
        fprintf(stderr, "{\"axis_iterations\": %zu, \"axis_x\": %.17g, \"axis_z\": %.17g, \"b0\": %.17g, \"larmor_radius\": %.17g}\n",
                axis_iterations, static_cast<double> (axis_x), static_cast<double> (axis_z),
                static_cast<double> (b0_value), static_cast<double> (larmor_value));
        fprintf(stderr, "step ");
        step->code.print_counts(stderr);
    }
};

//  korc <in: x y z ux uy uz gamma> <out> <num_steps> <save_every>
//  Output: first record after initialize_gamma, then saved steps; b0 and larmor on stderr.
template<typename T>
static int cmd_korc(const raw_tables &raw, const char *in_path, const char *out_path,
                    const size_t num_steps, const size_t save_every) {
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 7, n));
    korc_graphs<T> g(raw);
    g.print_info();

    g.init->run(n, pointers(cols, 3, 4), {});
    std::vector<std::vector<double>> record;
    auto save = [&] () {
        for (size_t cidx = 0; cidx < 7; cidx++) record.emplace_back(cols[cidx].begin(), cols[cidx].end());
    };
    save();
    for (size_t k = 1; k <= num_steps; k++) {
        g.step->run(n, pointers(cols, 0, 7), {});
        if (save_every && (k%save_every == 0 || k == num_steps)) save();
    }
    if (!save_every) save();
    write_columns(out_path, record);
    return 0;
}

//  export_korc <dir> <suffix>: the four xkorc work items as GFIR.
template<typename T>
static int cmd_export_korc(const raw_tables &raw, const std::string dir, const std::string suffix) {
    korc_graphs<T> g(raw);
    g.print_info();
    g.axis_newton->write_gfir("axis_newton", dir + "/korc_axis_newton_" + suffix + ".gfir");
    g.bmod->write_gfir("bmod_at_axis", dir + "/korc_bmod_at_axis_" + suffix + ".gfir");
    g.init->write_gfir("initialize_gamma", dir + "/korc_initialize_gamma_" + suffix + ".gfir");
    g.step->write_gfir("step", dir + "/korc_step_" + suffix + ".gfir");
    return 0;
}

//  psi2 <in: r z> <out: psi psi_r psi_z psi_rr psi_rz psi_zr psi_zz>  (diagnostic)
template<typename T>
static int cmd_psi2(const raw_tables &raw, const char *in_path, const char *out_path) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 2, n));
    auto r = graph::variable<T> (1, "r");
    auto z = graph::variable<T> (1, "z");
    auto psi = eq.build_psi(r, eq.dr, eq.rmin, z, eq.dz, eq.zmin);
    auto psi_r = psi->df(r);
    auto psi_z = psi->df(z);
    work_item<T> item({r, z}, {psi, psi_r, psi_z, psi_r->df(r), psi_r->df(z), psi_z->df(r), psi_z->df(z)}, {});
    item.code.print_counts(stderr);
    std::vector<std::vector<T>> outs(7, std::vector<T> (n));
    item.run(n, pointers(cols, 0, 2), pointers(outs, 0, 7));
    write_columns(out_path, to_double(outs));
    return 0;
}

//  fields2 <in: x y z> <out: for f in (ne, ni, bx, by, bz): f, df/dx, df/dy, df/dz>  (diagnostic)
template<typename T>
static int cmd_fields2(const raw_tables &raw, const char *in_path, const char *out_path) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 3, n));
    auto x = graph::variable<T> (1, "x");
    auto y = graph::variable<T> (1, "y");
    auto z = graph::variable<T> (1, "z");
    auto b = eq.get_magnetic_field(x, y, z);
    std::vector<leaf<T>> f = {eq.get_electron_density(x, y, z), eq.get_ion_density(x, y, z),
                              b->get_x(), b->get_y(), b->get_z()};
    std::vector<leaf<T>> outs_nodes;
    for (auto &e : f) {
        outs_nodes.push_back(e);
        outs_nodes.push_back(e->df(x));
        outs_nodes.push_back(e->df(y));
        outs_nodes.push_back(e->df(z));
    }
    work_item<T> item({x, y, z}, outs_nodes, {});
    std::vector<std::vector<T>> outs(outs_nodes.size(), std::vector<T> (n));
    item.run(n, pointers(cols, 0, 3), pointers(outs, 0, outs.size()));
    write_columns(out_path, to_double(outs));
    return 0;
}

//  dparts <in: t w x y z kx ky kz> <out: for each intermediate of cold_plasma::D and D: f, df/dz, df/dx>  (diagnostic)
template<typename T>
static int cmd_dparts(const raw_tables &raw, const char *in_path, const char *out_path) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 8, n));
    ray_variables<T> v;
    std::vector<leaf<T>> parts;
    auto D = cold_plasma_D(v.w, graph::vector(v.kx, v.ky, v.kz), v.x, v.y, v.z, eq, &parts);
    parts.push_back(D);
    std::vector<leaf<T>> outs_nodes;
    for (auto &e : parts) {
        outs_nodes.push_back(e);
        outs_nodes.push_back(e->df(v.z));
        outs_nodes.push_back(e->df(v.x));
    }
    work_item<T> item(v.inputs(), outs_nodes, {});
    std::vector<std::vector<T>> outs(outs_nodes.size(), std::vector<T> (n));
    item.run(n, pointers(cols, 0, 8), pointers(outs, 0, outs.size()));
    write_columns(out_path, to_double(outs));
    return 0;
}

//  Children of a node through the public casts (diagnostic traversal).
template<typename T>
static std::vector<leaf<T>> children(leaf<T> n, std::string &kind) {
    if (graph::constant_cast(n).get()) { kind = "constant"; return {}; }
    if (graph::variable_cast(n).get()) { kind = "variable"; return {}; }
    if (auto p = graph::pseudo_variable_cast(n); p.get()) { kind = "pseudo"; return {p->get_arg()}; }
    if (auto x = graph::add_cast(n); x.get()) { kind = "add"; return {x->get_left(), x->get_right()}; }
    if (auto x = graph::subtract_cast(n); x.get()) { kind = "sub"; return {x->get_left(), x->get_right()}; }
    if (auto x = graph::multiply_cast(n); x.get()) { kind = "mul"; return {x->get_left(), x->get_right()}; }
    if (auto x = graph::divide_cast(n); x.get()) { kind = "div"; return {x->get_left(), x->get_right()}; }
    if (auto x = graph::fma_cast(n); x.get()) { kind = "fma"; return {x->get_left(), x->get_middle(), x->get_right()}; }
    if (auto x = graph::sqrt_cast(n); x.get()) { kind = "sqrt"; return {x->get_arg()}; }
    if (auto x = graph::pow_cast(n); x.get()) { kind = "pow"; return {x->get_left(), x->get_right()}; }
    if (auto x = graph::piecewise_1D_cast(n); x.get()) { kind = "pw1"; return {x->get_arg()}; }
    if (auto x = graph::piecewise_2D_cast(n); x.get()) { kind = "pw2"; return {x->get_left(), x->get_right()}; }
    if (auto x = graph::sin_cast(n); x.get()) { kind = "sin"; return {x->get_arg()}; }
    if (auto x = graph::cos_cast(n); x.get()) { kind = "cos"; return {x->get_arg()}; }
    if (auto x = graph::atan_cast(n); x.get()) { kind = "atan"; return {x->get_left(), x->get_right()}; }
    kind = "other";
    return {};
}

//  dfcheck <in: one ray t w x y z kx ky kz> <which: 0 cx2, 1 D> : walks the DAG and compares
//  node->df(z) against central differences of the node value.  (diagnostic)
template<typename T>
static int cmd_dfcheck(const raw_tables &raw, const char *in_path, const int which) {
    efit<T> eq(raw);
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 8, n));
    ray_variables<T> v;
    std::vector<leaf<T>> parts;
    auto D = cold_plasma_D(v.w, graph::vector(v.kx, v.ky, v.kz), v.x, v.y, v.z, eq, &parts);
    leaf<T> root = which == 0 ? parts[21] : D;

    std::vector<leaf<T>> order;
    std::map<graph::leaf_node<T> *, std::string> kinds;
    std::function<void(leaf<T>)> walk = [&] (leaf<T> node) {
        if (kinds.count(node.get())) return;
        std::string kind;
        auto ch = children(node, kind);
        kinds[node.get()] = kind;
        for (auto &c : ch) walk(c);
        order.push_back(node);
    };
    walk(root);
    fprintf(stderr, "%zu nodes\n", order.size());

    const T h = 1.0E-6;
    std::vector<T> in0(8), inp(8), inm(8);
    for (int c = 0; c < 8; c++) in0[c] = inp[c] = inm[c] = cols[c][0];
    inp[4] += h; inm[4] -= h;
    std::map<graph::leaf_node<T> *, bool> bad;
    for (auto &node : order) {
        work_item<T> item(v.inputs(), {node, node->df(v.z)}, {});
        std::vector<T> regs;
        item.code.run(in0.data(), regs);
        const T sym = regs[item.output_slots[1]];
        item.code.run(inp.data(), regs);
        const T fp = regs[item.output_slots[0]];
        item.code.run(inm.data(), regs);
        const T fm = regs[item.output_slots[0]];
        const T fd = (fp - fm)/(2*h);
        const bool is_bad = std::abs(sym - fd) > 1.0E-5*(std::abs(fd) + std::abs(sym)) + 1.0E-9;
        bad[node.get()] = is_bad;
        if (is_bad) {
            std::string kind;
            auto ch = children(node, kind);
            bool child_bad = false;
            for (auto &c : ch) child_bad = child_bad || bad[c.get()];
            printf("BAD %s node %p sym %.10e fd %.10e children_bad %d :", kind.c_str(), (void *)node.get(), (double)sym, (double)fd, (int)child_bad);
            for (auto &c : ch) printf(" %s@%p", kinds[c.get()].c_str(), (void *)c.get());
            printf("\n");
            if (!child_bad && kind == "div") {
                auto L = ch[0], R = ch[1];
                auto dL = L->df(v.z), dR = R->df(v.z);
                auto t1 = dL/R;
                auto t2 = L*dR;
                auto t3 = R*R;
                auto t4 = t2/t3;
                auto t5 = t1 - t4;
                auto t6 = L*dR/(R*R);
                work_item<T> chk(v.inputs(), {L, R, dL, dR, t1, t2, t3, t4, t5, t6}, {});
                std::vector<T> rr;
                chk.code.run(in0.data(), rr);
                auto val = [&] (int k) { return (double)rr[chk.output_slots[k]]; };
                printf("  L %.10e R %.10e dL %.10e dR %.10e\n", val(0), val(1), val(2), val(3));
                printf("  t1=dL/R sym %.10e num %.10e\n", val(4), val(2)/val(1));
                printf("  t2=L*dR sym %.10e num %.10e\n", val(5), val(0)*val(3));
                printf("  t3=R*R sym %.10e num %.10e\n", val(6), val(1)*val(1));
                printf("  t4=t2/t3 sym %.10e num %.10e\n", val(7), val(0)*val(3)/(val(1)*val(1)));
                printf("  t5=t1-t4 sym %.10e num %.10e\n", val(8), val(2)/val(1) - val(0)*val(3)/(val(1)*val(1)));
                printf("  t6=L*dR/(R*R) sym %.10e\n", val(9));
                std::function<void(leaf<T>, int, int)> show = [&] (leaf<T> e, int depth, int maxd) {
                    std::string kk;
                    auto cc = children(e, kk);
                    work_item<T> one(v.inputs(), {e}, {});
                    std::vector<T> r1;
                    one.code.run(in0.data(), r1);
                    printf("%*s%s %p = %.10e\n", 4 + 2*depth, "", kk.c_str(), (void *)e.get(), (double)r1[one.output_slots[0]]);
                    if (depth < maxd) for (auto &c2 : cc) show(c2, depth + 1, maxd);
                };
                printf("  tree t2 = L*dR\n"); show(t2, 0, 3);
                printf("  tree t3 = R*R\n"); show(t3, 0, 3);
                printf("  tree t4 = t2/t3\n"); show(t4, 0, 4);
                std::string k1, k2;
                children(t1, k1); children(t4, k2);
                printf("  kinds: L %s R %s dL %s dR %s t1 %s t4 %s\n", kinds[L.get()].c_str(), kinds[R.get()].c_str(), "-", "-", k1.c_str(), k2.c_str());
            }
        }
    }
    return 0;
}

//  emit <dt> <out.txt>: dump the solver_kernel tape as C-like SSA (feasibility probe)
template<typename T>
static int cmd_emit(const raw_tables &raw, const double dt, const char *out_path) {
    efit<T> eq(raw);
    ray_variables<T> v;
    dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq);
    work_item<T> solver = make_solver_kernel(v, eq, static_cast<T> (dt), D);
    FILE *f = fopen(out_path, "w");
    const auto &code = solver.code.code;
    std::map<const T *, int> table_ids;
    for (size_t i = 0; i < code.size(); i++) {
        const auto &c = code[i];
        switch (c.op) {
            case op_t::constant: fprintf(f, "const double r%zu = %.17g;\n", i, (double)c.value); break;
            case op_t::input: fprintf(f, "const double r%zu = in%d;\n", i, c.a); break;
            case op_t::add: fprintf(f, "const double r%zu = r%d + r%d;\n", i, c.a, c.b); break;
            case op_t::sub: fprintf(f, "const double r%zu = r%d - r%d;\n", i, c.a, c.b); break;
            case op_t::mul: fprintf(f, "const double r%zu = r%d*r%d;\n", i, c.a, c.b); break;
            case op_t::div: fprintf(f, "const double r%zu = r%d/r%d;\n", i, c.a, c.b); break;
            case op_t::fma: fprintf(f, "const double r%zu = fma(r%d, r%d, r%d);\n", i, c.a, c.b, c.c); break;
            case op_t::sqrt: fprintf(f, "const double r%zu = sqrt(r%d);\n", i, c.a); break;
            case op_t::powi: {
                fprintf(f, "const double r%zu = r%d", i, c.a);
                for (size_t k = 1; k < c.power; k++) fprintf(f, "*r%d", c.a);
                fprintf(f, ";\n");
                break;
            }
            case op_t::pow: fprintf(f, "const double r%zu = pow(r%d, r%d);\n", i, c.a, c.b); break;
            case op_t::gather1: {
                if (!table_ids.count(c.table)) { int id = table_ids.size(); table_ids[c.table] = id; }
                fprintf(f, "const double r%zu = T1_%d[IDX(r%d, %.17g, %.17g, %zu)];\n", i, table_ids[c.table], c.a, (double)c.offset, (double)c.scale, c.length - 1);
                break;
            }
            case op_t::gather2: {
                if (!table_ids.count(c.table)) { int id = table_ids.size(); table_ids[c.table] = id; }
                fprintf(f, "const double r%zu = T2_%d[IDX(r%d, %.17g, %.17g, %zu)*%zu + IDX(r%d, %.17g, %.17g, %zu)];\n", i, table_ids[c.table],
                        c.a, (double)c.offset, (double)c.scale, c.num_rows - 1, c.num_columns, c.b, (double)c.y_offset, (double)c.y_scale, c.num_columns - 1);
                break;
            }
            default: fprintf(f, "// unsupported op\n");
        }
    }
    for (auto &s : solver.setter_slots) fprintf(f, "out_set%d = r%d;\n", s.second, s.first);
    for (size_t o = 0; o < solver.output_slots.size(); o++) fprintf(f, "out%zu = r%d;\n", o, solver.output_slots[o]);
    fprintf(f, "// tables %zu\n", table_ids.size());
    fclose(f);
    return 0;
}

//  export <dir> <dt>: write the hot-path work items as GFIR files.
template<typename T>
static int cmd_export(const raw_tables &raw, const std::string dir, const double dt, const std::string suffix) {
    {
        efit<T> eq(raw);
        ray_variables<T> v;
        dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq);
        const char *names[] = {"w", "kx", "ky", "kz"};
        for (int var = 0; var < 4; var++) {
            work_item<T> loss = make_loss_kernel(v, D.D, var, static_cast<T> (1.0));
            loss.write_gfir("loss_kernel", dir + "/loss_kernel_" + names[var] + "_" + suffix + ".gfir");
        }
        work_item<T> solver = make_solver_kernel(v, eq, static_cast<T> (dt), D);
        solver.write_gfir("solver_kernel", dir + "/solver_kernel_" + suffix + ".gfir");
        work_item<T> disp(v.inputs(), {D.D, D.dDdw, D.dDdkx, D.dDdky, D.dDdkz, D.dDdx, D.dDdy, D.dDdz}, {});
        disp.write_gfir("dispersion_kernel", dir + "/dispersion_kernel_" + suffix + ".gfir");
    }
    {
        efit<T> eq(raw);
        auto x = graph::variable<T> (1, "x");
        auto y = graph::variable<T> (1, "y");
        auto z = graph::variable<T> (1, "z");
        auto bvec = eq.get_magnetic_field(x, y, z);
        auto ne = eq.get_electron_density(x, y, z);
        auto te = eq.get_electron_temperature(x, y, z);
        auto div = bvec->get_x()->df(x) + bvec->get_y()->df(y) + bvec->get_z()->df(z);
        work_item<T> item({x, y, z}, {bvec->get_x(), bvec->get_y(), bvec->get_z(), ne, te, div}, {});
        item.write_gfir("test_kernel", dir + "/efit_test_kernel_" + suffix + ".gfir");
    }
    return 0;
}

//  export_misc <dir> <suffix>: small items that exercise every remaining node type the way
//  graph_tests/jit_test.cpp, math_test.cpp, trigonometry_test.cpp and workflow_test.cpp:20-71 do
//  (transcendentals, general pow, setters that alias their own inputs).
template<typename T>
static int cmd_export_misc(const std::string dir, const std::string suffix) {
    auto a = graph::variable<T> (1, "a");
    auto b = graph::variable<T> (1, "b");
    auto c = graph::variable<T> (1, "c");
    work_item<T> math({a, b, c},
                      {graph::sin(a)*graph::cos(b), graph::atan(a, b), graph::exp(c/10.0),
                       graph::log(a*a + 1.0), graph::pow(a*a + 1.0, b), graph::sqrt(a*a + b*b)/(c*c + 2.0),
                       graph::fma(a, b, c)}, {});
    math.write_gfir("math_kernel", dir + "/misc_math_kernel_" + suffix + ".gfir");
//  workflow_test.cpp:20-71: x <- x + 1 style setters, an output of the OLD values, a swap.
    work_item<T> alias({a, b, c}, {a + b + c},
                       {{a + 1.0, a}, {a*b, b}, {graph::sqrt(c*c) + a, c}});
    alias.write_gfir("alias_kernel", dir + "/misc_alias_kernel_" + suffix + ".gfir");
    return 0;
}

//  weak_damping <in: kamp kx ky kz x y z t w (real columns)> <out: re(kamp) im(kamp)> <gfir path|->
//  The absorption item of xrays (absorption.hpp:346-432) on complex<double>, SAFE_MATH = true, evaluated
//  on the tape in the host's std::complex arithmetic (what a cpu_context kernel is written in) with the
//  reference's own special::erfi; the input columns become the real parts (output.hpp:412-470 reads the
//  real trajectory variables into complex buffers with stride 2).
static int cmd_weak_damping(const raw_tables &raw, const char *in_path, const char *out_path, const char *gfir_path) {
    typedef std::complex<double> T;
    size_t n;
    auto cols = read_columns(in_path, 9, n);
    efit<T, true> eq(raw);
    std::vector<leaf<T, true>> debug;
    if (getenv("GF_REF_DEBUG")) absorption_debug<T, true> = &debug;
    weak_damping_item<T, true> built(eq);
    std::vector<leaf<T, true>> in(built.inputs.begin(), built.inputs.end());
    std::vector<std::pair<leaf<T, true>, leaf<T, true>>> set;
    for (auto &s : built.setters) set.push_back({s.first, s.second});
    work_item<T, true> item(in, debug, set);
    item.code.print_counts(stderr);
    if (std::string(gfir_path) != "-") {
        item.write_gfir("weak_damping_kimg_kernel", gfir_path);
    }
    std::vector<std::vector<T>> columns;
    for (auto &c : cols) columns.emplace_back(c.begin(), c.end());
    std::vector<std::vector<T>> debug_values(debug.size(), std::vector<T> (n));
    item.run(n, pointers(columns, 0, 9), pointers(debug_values, 0, debug.size()));
    for (size_t d = 0; d < debug.size(); d++) {
        fprintf(stderr, "debug %zu: (%g, %g)\n", d, std::real(debug_values[d][0]), std::imag(debug_values[d][0]));
    }
    std::vector<std::vector<double>> out(2, std::vector<double> (n));
    for (size_t i = 0; i < n; i++) {
        out[0][i] = std::real(columns[0][i]);
        out[1][i] = std::imag(columns[0][i]);
    }
    write_columns(out_path, out);
    return 0;
}

//  root_finder <in: kamp kx ky kz x y z t w (real columns)> <out: re(kamp) im(kamp) iterations> <gfir dir|->
//  absorption::root_finder (absorption.hpp:146-290) on complex<double>, SAFE_MATH = true: init, the Newton
//  converge item on the whole batch as one shard (workflow.hpp:179-205 with T complex: the max is the element
//  of largest modulus, cpu_context.hpp:314-318; numeric_limits<std::complex>::max() is T()), final_kamp.
static int cmd_root_finder(const raw_tables &raw, const char *in_path, const char *out_path, const char *gfir_dir) {
    typedef std::complex<double> T;
    size_t n;
    auto cols = read_columns(in_path, 9, n);
    efit<T, true> eq(raw);
    root_finder_items<T, true> items(eq);
    items.loss->code.print_counts(stderr);
    if (std::string(gfir_dir) != "-") {
        const std::string dir(gfir_dir);
        items.init->write_gfir("root_find_init_kernel", dir + "/root_find_init_kernel_c64.gfir");
        items.loss->write_gfir("loss_kernel", dir + "/root_find_loss_kernel_c64.gfir");
        items.final_kamp->write_gfir("final_kamp", dir + "/root_find_final_kamp_c64.gfir");
    }
    std::vector<std::vector<T>> columns;
    for (auto &c : cols) columns.emplace_back(c.begin(), c.end());
    items.init->run(n, pointers(columns, 0, 7), {});
    std::vector<T> residual(n);
    auto max_kernel = [&] () -> T {
        items.loss->run(n, pointers(columns, 0, 9), {residual.data()});
        return *std::max_element(residual.begin(), residual.end(),
                                 [] (const T a, const T b) { return std::abs(a) < std::abs(b); });
    };
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
    items.final_kamp->run(n, pointers(columns, 0, 7), {});
    fprintf(stderr, "{\"iterations\": %zu, \"max_residual\": [%.17g, %.17g]}\n", iterations,
            std::real(max_residual), std::imag(max_residual));
    std::vector<std::vector<double>> out(3, std::vector<double> (n));
    for (size_t i = 0; i < n; i++) {
        out[0][i] = std::real(columns[0][i]);
        out[1][i] = std::imag(columns[0][i]);
        out[2][i] = static_cast<double> (iterations);
    }
    write_columns(out_path, out);
    return 0;
}

//  power <in: x y z x_last y_last z_last kamp power k_sum, then `records` blocks of (x y z kamp)> <out> <records> <gfir|->
//  bin_power's loop (graph_driver/xrays.cpp:745-775): per record copy x, y, z, kamp in, run the `power`
//  item once.  out: per record (power, d_power, k_sum).
template<typename T>
static int cmd_power(const char *in_path, const char *out_path, const size_t records, const char *gfir_path) {
    size_t n;
    auto cols = convert<T> (read_columns(in_path, 9 + 4*records, n));
    power_item<T> p;
    std::vector<leaf<T>> in(p.inputs.begin(), p.inputs.end());
    std::vector<std::pair<leaf<T>, leaf<T>>> set;
    for (auto &s : p.setters) set.push_back({s.first, s.second});
    work_item<T> item(in, {p.d_power}, set);
    if (std::string(gfir_path) != "-") {
        item.write_gfir("power", gfir_path);
    }
    std::vector<T> d_power(n);
    std::vector<std::vector<T>> out;
    for (size_t r = 0; r < records; r++) {
        for (size_t c = 0; c < 3; c++) cols[c] = cols[9 + 4*r + c];
        cols[6] = cols[9 + 4*r + 3];
        item.run(n, pointers(cols, 0, 9), {d_power.data()});
        out.push_back(cols[7]);
        out.push_back(d_power);
        out.push_back(cols[8]);
    }
    write_columns(out_path, to_double(out));
    return 0;
}

template<typename T>
static int dispatch(const raw_tables &raw, int argc, char **argv) {
    const std::string cmd = argv[3];
    if (cmd == "efit_test" && argc == 6) {
        return cmd_efit_test<T> (raw, argv[4], argv[5]);
    } else if (cmd == "psi2" && argc == 6) {
        return cmd_psi2<T> (raw, argv[4], argv[5]);
    } else if (cmd == "fields2" && argc == 6) {
        return cmd_fields2<T> (raw, argv[4], argv[5]);
    } else if (cmd == "dparts" && argc == 6) {
        return cmd_dparts<T> (raw, argv[4], argv[5]);
    } else if (cmd == "dfcheck" && argc == 6) {
        return cmd_dfcheck<T> (raw, argv[4], atoi(argv[5]));
    } else if (cmd == "emit" && argc == 6) {
        return cmd_emit<T> (raw, atof(argv[4]), argv[5]);
    } else if (cmd == "export" && argc == 7) {
        return cmd_export<T> (raw, argv[4], atof(argv[5]), argv[6]);
    } else if (cmd == "dispersion" && argc == 6) {
        return cmd_dispersion<T> (raw, argv[4], argv[5]);
    } else if (cmd == "trace_ordinary" && argc == 10) {
        return cmd_trace<T> (raw, argv[4], argv[5], atof(argv[6]), strtoull(argv[7], nullptr, 10),
                             strtoull(argv[8], nullptr, 10), atoi(argv[9]), true);
    } else if (cmd == "export_ordinary" && argc == 7) {
        efit<T> eq(raw);
        ray_variables<T> v;
        dispersion_interface<T> D(v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z, eq, ordinary_wave_D<T>);
        work_item<T> loss = make_loss_kernel(v, D.D, 1, static_cast<T> (1.0));
        loss.write_gfir("loss_kernel", std::string(argv[4]) + "/ordinary_wave_loss_kernel_kx_" + argv[6] + ".gfir");
        work_item<T> solver = make_solver_kernel(v, eq, static_cast<T> (atof(argv[5])), D);
        solver.write_gfir("solver_kernel", std::string(argv[4]) + "/ordinary_wave_solver_kernel_" + argv[6] + ".gfir");
        return 0;
    } else if (cmd == "trace" && argc == 10) {
        return cmd_trace<T> (raw, argv[4], argv[5], atof(argv[6]), strtoull(argv[7], nullptr, 10),
                             strtoull(argv[8], nullptr, 10), atoi(argv[9]));
    } else if (cmd == "trace_adaptive" && argc == 9) {
        return cmd_trace_adaptive<T> (raw, argv[4], argv[5], strtoull(argv[6], nullptr, 10), argv[7], argv[8]);
    } else if (cmd == "power" && argc == 8) {
        return cmd_power<T> (argv[4], argv[5], strtoull(argv[6], nullptr, 10), argv[7]);
    } else if (cmd == "export_misc" && argc == 6) {
        return cmd_export_misc<T> (argv[4], argv[5]);
    } else if (cmd == "export_korc" && argc == 6) {
        return cmd_export_korc<T> (raw, argv[4], argv[5]);
    } else if (cmd == "korc" && argc == 8) {
        return cmd_korc<T> (raw, argv[4], argv[5], strtoull(argv[6], nullptr, 10), strtoull(argv[7], nullptr, 10));
    }
    fprintf(stderr, "bad command\n");
    return 2;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr,
                "usage: gf_ref <tables.bin> <f64|f32> efit_test <in> <out>\n"
                "       gf_ref <tables.bin> <f64|f32> dispersion <in> <out>\n"
                "       gf_ref <tables.bin> <f64|f32> trace <in> <out> <dt> <num_steps> <save_every> <newton_var|-1>\n"
                "       gf_ref <tables.bin> <f64|f32> korc <in> <out> <num_steps> <save_every>\n");
        return 2;
    }
    const raw_tables raw(argv[1]);
    if (std::string(argv[2]) == "c64") {
        if (std::string(argv[3]) == "weak_damping" && argc == 7) {
            return cmd_weak_damping(raw, argv[4], argv[5], argv[6]);
        }
        if (std::string(argv[3]) == "root_finder" && argc == 7) {
            return cmd_root_finder(raw, argv[4], argv[5], argv[6]);
        }
        fprintf(stderr, "bad command\n");
        return 2;
    }
    if (std::string(argv[2]) == "f32") {
        return dispatch<float> (raw, argc, argv);
    }
    return dispatch<double> (raw, argc, argv);
}
