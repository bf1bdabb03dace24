// ---------------------------------------------------------------------------
// ref_scenarios.hpp — TEST INFRASTRUCTURE.  The scenarios of graph_tests/solver_test.cpp and
// graph_tests/physics_test.cpp, written once over a SOLVER type that offers the members of
// solver::solver_interface they use (set/get on the ray variables, init, compile, step):
//   ray_solver          (ref_physics.cpp)          the tape interpreter, host columns
//   hip_ray_solver      (hip_context_physics.cpp)  gpu::hip_context driven as jit::context does
// ---------------------------------------------------------------------------
#ifndef ref_scenarios_hpp
#define ref_scenarios_hpp

#include "ref_physics.hpp"

#include <memory>
#include <sstream>

typedef double T;

enum class method {rk2, rk4, split};

inline std::string g17(const double v) {
    char text[64];
    snprintf(text, sizeof(text), "%.17g", v);
    return text;
}

//  Printing goes to `out` so that two replays can be compared as text.
struct report {
    std::ostringstream out;
    bool first = true;
    bool all_passed = true;
    void begin(const std::string &name) {
        out << (first ? "" : ",") << "\n \"" << name << "\": {";
        first = false;
        first_field = true;
    }
    bool first_field = true;
    void field(const std::string &key, const std::string &value) {
        out << (first_field ? "" : ",") << "\n  \"" << key << "\": " << value;
        first_field = false;
    }
    void number(const std::string &key, const double value) { field(key, g17(value)); }
    void passed(const bool ok, const char *what) {
        field("reference_assertion", std::string("\"") + what + "\"");
        field("reference_assertion_holds", ok ? "true" : "false");
        if (!ok) {
            all_passed = false;
            fprintf(stderr, "ASSERTION OF THE REFERENCE TEST FAILS: %s\n", what);
        }
    }
    void end() { out << "\n }"; }
    static std::string list(const std::vector<std::string> &items) {
        std::string s = "[";
        for (size_t i = 0; i < items.size(); i++) s += (i ? ", " : "") + items[i];
        return s + "]";
    }
    static std::string counts(const std::vector<size_t> &items) {
        std::string s = "[";
        for (size_t i = 0; i < items.size(); i++) s += (i ? ", " : "") + std::to_string(items[i]);
        return s + "]";
    }
};

//  solver_test.cpp:28-60.
template<typename SOLVER>
void solver_test(report &r, const std::string &dir, const char *label, dispersion_function<T> f,
                 const method m, const T omega0, const T kx0, const T dt) {
    analytic_equilibrium<T> eq(analytic_kind::gaussian_density);
    const std::string name = std::string("solver_test_") + label + (m == method::rk2 ? "_rk2" : "_rk4");
    SOLVER solve(name, dir, eq, f);
    solve.set("w", omega0); solve.set("kx", kx0); solve.set("ky", 0.25); solve.set("kz", 0.15);
    const T tolerance = 1.0E-30;
    solve.init(1, tolerance);
    std::vector<std::string> states = {solve.state()};
    solve.compile(m, dt);
    bool ok = true;
    for (size_t i = 0; i < 5; i++) {
        solve.step();
        ok = ok && std::abs(solve.residual_at(0)) < std::abs(tolerance);
        states.push_back(solve.state());
    }
    r.begin(name);
    r.field("initial", "{\"w\": " + g17(omega0) + ", \"kx\": " + g17(kx0) + ", \"ky\": 0.25, \"kz\": 0.15}");
    r.number("dt", dt);
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list(states));
    r.passed(ok, "solver_test.cpp:54-58 residual < 1e-30 after each of 5 steps");
    r.end();
}

//  physics_test.cpp:24-73 with fixed values in place of the clock-seeded random ones.
template<typename SOLVER>
void test_constant(report &r, const std::string &dir) {
    analytic_equilibrium<T> eq(analytic_kind::slab);
    SOLVER solve("constant", dir, eq, simple_D<T>);
    solve.set("w", 0.7); solve.set("kx", 0.4); solve.set("x", 0.3); solve.set("y", 0.5); solve.set("z", 0.9);
    auto constant = [&] () {
        return solve.get("kx")*solve.get("x") + solve.get("ky")*solve.get("y") + solve.get("kz")*solve.get("z")
             - solve.get("w")*solve.get("t");
    };
    solve.init(1);
    const T c0 = constant();
    std::vector<std::string> states = {solve.state()};
    solve.compile(method::rk2, 1.0);
    for (size_t i = 0; i < 10; i++) solve.step();
    states.push_back(solve.state());
    r.begin("constant");
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list(states));
    r.number("constant_before", c0);
    r.number("constant_after", constant());
    r.passed(std::abs(c0 - constant()) < 5.0E-15, "physics_test.cpp:70-71 |c0 - c| < 5e-15");
    r.end();
}

//  physics_test.cpp:86-150 (bohm_gross) and :163-221 (light_wave).
template<typename SOLVER>
void test_wave_in_gradient(report &r, const std::string &dir, const bool bohm_gross, const method m, const T tolerance) {
    const T q = 1.602176634E-19, me = 9.1093837015E-31, mu0 = M_PI*4.0E-7, epsilon0 = 8.8541878138E-12;
    const T c = 1.0/sqrt(mu0*epsilon0);
    const T omega0 = 600.0, ne0 = 1.0E19, te = 1000.0;
    const T omega2 = (ne0*0.9*q*q)/(epsilon0*me*c*c);
    const T omega2p = (ne0*0.1*q*q)/(epsilon0*me*c*c);
    const T vth2 = 2*1.602176634E-19*te/(me*c*c);

    analytic_equilibrium<T> eq(analytic_kind::no_magnetic_field);
    const std::string name = std::string(bohm_gross ? "bohm_gross" : "light_wave") + (m == method::rk4 ? "_rk4" : "_split");
    SOLVER solve(name, dir, eq, bohm_gross ? bohm_gross_D<T> : light_wave_D<T>);
    solve.set("w", 600.0); solve.set("kx", bohm_gross ? 1000.0 : 100.0); solve.set("x", -1.0);
    solve.init(1);                        // CPU path of the reference: default tolerance for bohm_gross,
                                          // `tolerance` for light_wave; both stall long before either.
    std::vector<std::string> states = {solve.state()};
    solve.compile(m, 0.1);
    for (size_t i = 0; i < 20; i++) solve.step();
    states.push_back(solve.state());
    const T time = solve.get("t");
    T expected_x;
    if (bohm_gross) {
        const T k0 = std::sqrt(2.0/3.0*(omega0*omega0 - omega2)/vth2);
        expected_x = -3.0/8.0*vth2*omega2p/(omega0*omega0)*time*time + 3.0/2.0*vth2/omega0*k0*time - 1.0;
    } else {
        const T k0 = std::sqrt(omega0*omega0 - omega2);
        expected_x = -omega2p/(4.0*omega0*omega0)*time*time + k0/omega0*time - 1.0;
    }
    const T diff_x = solve.get("x") - expected_x;
    r.begin(name);
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list(states));
    r.number("expected_x", expected_x);
    r.number("tolerance", tolerance);
    r.passed(std::abs(diff_x*diff_x) < std::abs(tolerance), "physics_test.cpp:147-149/:218-220 (x - expected_x)^2 < tolerance");
    r.end();
}

//  physics_test.cpp:233-283.
template<typename SOLVER>
void test_acoustic_wave(report &r, const std::string &dir, const T tolerance) {
    const T q = 1.602176634E-19, mi = 3.34449469E-27, mu0 = M_PI*4.0E-7, epsilon0 = 8.8541878138E-12;
    const T c = 1.0/sqrt(mu0*epsilon0);
    const T te = 1000.0, ti = te, gamma = 3;
    const T vs = std::sqrt((q*te + gamma*q*ti)/mi)/c;
    analytic_equilibrium<T> eq(analytic_kind::no_magnetic_field);
    SOLVER solve("acoustic_wave_rk4", dir, eq, acoustic_wave_D<T>);
    solve.set("w", 1.0); solve.set("kx", 600.0);
    solve.init(1, tolerance);
    std::vector<std::string> states = {solve.state()};
    solve.compile(method::rk4, 0.0001);
    for (size_t i = 0; i < 20; i++) solve.step();
    states.push_back(solve.state());
    const T diff_x = solve.get("x")/solve.get("t") - vs;
    r.begin("acoustic_wave_rk4");
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list(states));
    r.number("vs", vs);
    r.number("tolerance", tolerance);
    r.passed(std::abs(diff_x*diff_x) < std::abs(tolerance), "physics_test.cpp:280-282 (x/t - vs)^2 < tolerance");
    r.end();
}

//  physics_test.cpp:341-383: Newton on x finds the O-mode cut-off of a linear density.
template<typename SOLVER>
void test_o_mode_wave(report &r, const std::string &dir) {
    const T q = 1.602176634E-19, me = 9.1093837015E-31, mu0 = M_PI*4.0E-7, epsilon0 = 8.8541878138E-12;
    const T c = 1.0/sqrt(mu0*epsilon0);
    const T ne0 = 1.0E19;
    const T omega2 = (ne0*q*q)/(epsilon0*me*c*c);
    const T omega0 = 1000.0;
    const T x_cut = (omega0*omega0 - 1.0 - omega2)/(omega2*0.1);
    analytic_equilibrium<T> eq(analytic_kind::slab_density);
    SOLVER solve("o_mode_wave", dir, eq, ordinary_wave_D<T>);
    solve.set("w", omega0);
    solve.init(4);
    const T diff = solve.get("x") - x_cut;
    r.begin("o_mode_wave");
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list({solve.state()}));
    r.number("x_cut", x_cut);
    r.passed(std::abs(diff*diff) < 8.0E-10, "physics_test.cpp:380-382 (x - x_cut)^2 < 8e-10");
    r.end();
}

//  physics_test.cpp:498-546 with (tolerance, n0, x0, kx0) = (2e-29, 0.7, 0.1, 22) of :629.
template<typename SOLVER>
void test_reflection(report &r, const std::string &dir, const T tolerance, const T n0, const T x0, const T kx0) {
    const T q = 1.602176634E-19, me = 9.1093837015E-31, mu0 = M_PI*4.0E-7, epsilon0 = 8.8541878138E-12;
    const T c = static_cast<T> (1.0)/sqrt(mu0*epsilon0);
    const T OmegaCE = -q/(me*c);
    analytic_equilibrium<T> eq(analytic_kind::slab);
    SOLVER solve("reflection", dir, eq, cold_plasma_D<T>);
    solve.set("w", OmegaCE); solve.set("kz", n0*OmegaCE); solve.set("x", x0);
    solve.init(4, tolerance);
    const T cutoff_location = solve.get("x");
    solve.set("x", cutoff_location - static_cast<T> (0.00001)*cutoff_location);
    solve.set("kx", kx0);
    solve.init(1, tolerance);
    std::vector<std::string> states = {solve.state()};
    solve.compile(method::rk4, 0.0001);
    T max_x = solve.get("x");
    T new_x = max_x;
    bool ok = true;
    size_t steps = 0;
    do {
        solve.step();
        steps++;
        new_x = solve.get("x");
        max_x = std::max(new_x, max_x);
        ok = ok && std::abs(max_x - cutoff_location) < 1.9E-6;
    } while (max_x == new_x && steps < 10000000);
    states.push_back(solve.state());
    r.begin("reflection");
    r.field("initial", "{\"w\": " + g17(OmegaCE) + ", \"kz\": " + g17(n0*OmegaCE) + ", \"x\": " + g17(x0) + ", \"kx_guess\": " + g17(kx0) + "}");
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list(states));
    r.number("cutoff_location", cutoff_location);
    r.number("max_x", max_x);
    r.number("tolerance", tolerance);
    r.field("steps", std::to_string(steps));
    r.passed(ok, "physics_test.cpp:538-539 |max_x - cutoff| < 1.9e-6 until the ray turns");
    r.end();
}

//  physics_test.cpp:395-485, two rays.
template<typename SOLVER>
void test_cold_plasma_cutoffs(report &r, const std::string &dir) {
    analytic_equilibrium<T> eq(analytic_kind::slab_density);
    SOLVER solve("cold_plasma_cutoffs", dir, eq, cold_plasma_D<T>, 2);
    solve.set("w", 1100.0);
    solve.set("x", 0, 25.0); solve.set("x", 1, 5.0);
    solve.init(4);
    T wpecut_pos = solve.get("x", 0);
    const T wrcut_pos = solve.get("x", 1);
    std::vector<std::string> states = {solve.state()};
    solve.set("x", 0.0);
    solve.set("kx", 0, 1000.0);     // O-mode
    solve.set("kx", 1, 500.0);      // X-mode
    solve.init(1);
    states.push_back(solve.state());
    solve.compile(method::rk4, 0.1);
    size_t steps_first = 0;
    while (std::abs(solve.get("t")) < 30.0) { solve.step(); steps_first++; }
    states.push_back(solve.state());
    const bool first = solve.get("x", 0) > wrcut_pos && solve.get("x", 0) < wpecut_pos && solve.get("x", 1) < wrcut_pos;

    solve.set("w", 800.0);
    solve.set("x", 0, 25.0); solve.set("x", 1, 5.0);
    solve.set("kx", 0.0);
    solve.set("t", 0.0);
    solve.init(4, 5.0E-30);
    wpecut_pos = solve.get("x", 1);
    states.push_back(solve.state());
    solve.set("x", 0.0);
    solve.set("kx", 0, 500.0);      // O-mode
    solve.set("kx", 1, 1500.0);     // X-mode
    solve.init(1);
    states.push_back(solve.state());
    size_t steps_second = 0;
    while (std::abs(solve.get("t")) < 60.0) { solve.step(); steps_second++; }
    states.push_back(solve.state());
    const bool second = solve.get("x", 0) < wpecut_pos && solve.get("x", 1) > wpecut_pos;

    r.begin("cold_plasma_cutoffs");
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list(states));
    r.number("wrcut_pos", wrcut_pos);
    r.number("wpecut_pos_second", wpecut_pos);
    r.field("steps", "[" + std::to_string(steps_first) + ", " + std::to_string(steps_second) + "]");
    r.passed(first && second, "physics_test.cpp:437-443,:479-484 O/X-mode rays stop at the expected cut-offs");
    r.end();
}

//  extra_ordinary_wave has no reference test of its own; pinned as a graph: D and its
//  Newton solve for kx on slab_field, then 10 rk4 steps.
template<typename SOLVER>
void extra_ordinary_wave(report &r, const std::string &dir) {
    analytic_equilibrium<T> eq(analytic_kind::slab_field);
    SOLVER solve("extra_ordinary_wave_rk4", dir, eq, extra_ordinary_wave_D<T>, 3);
    solve.set("w", 1500.0);
    solve.set("kx", 1200.0);
    solve.set("ky", 0, 0.0); solve.set("ky", 1, 40.0); solve.set("ky", 2, -25.0);
    solve.set("x", 0, 0.0); solve.set("x", 1, 1.5); solve.set("x", 2, -2.0);
    solve.init(1);
    std::vector<std::string> states = {solve.state()};
    solve.compile(method::rk4, 0.01);
    for (size_t i = 0; i < 10; i++) solve.step();
    states.push_back(solve.state());
    r.begin("extra_ordinary_wave_rk4");
    r.field("newton_iterations", report::counts(solve.newton_iterations));
    r.field("states", report::list(states));
    r.passed(true, "none (no reference test exercises extra_ordinary_wave); graph pinned only");
    r.end();
}


//  solver_test.cpp:85-87 (double; tolerance 1e-30, :113) and physics_test.cpp:627-637 with the
//  non-CUDA double tolerance 2e-29 (:649).
template<typename SOLVER>
void run_all(report &r, const std::string &dir) {
    for (const method m : {method::rk2, method::rk4}) {
        solver_test<SOLVER> (r, dir, "simple", simple_D<T>, m, 0.5, 0.25, 1.0);
        solver_test<SOLVER> (r, dir, "gaussian_well", gaussian_well_D<T>, m, 0.5, 0.25, 0.00001);
        solver_test<SOLVER> (r, dir, "cold_plasma", cold_plasma_D<T>, m, 900.0, 1000.0, 0.5/10000.0);
    }
    const T tolerance = 2.0E-29;
    test_constant<SOLVER> (r, dir);
    test_wave_in_gradient<SOLVER> (r, dir, true, method::rk4, tolerance);
    test_wave_in_gradient<SOLVER> (r, dir, true, method::split, tolerance);
    test_wave_in_gradient<SOLVER> (r, dir, false, method::rk4, tolerance);
    test_wave_in_gradient<SOLVER> (r, dir, false, method::split, tolerance);
    test_acoustic_wave<SOLVER> (r, dir, tolerance);
    test_o_mode_wave<SOLVER> (r, dir);
    test_reflection<SOLVER> (r, dir, tolerance, 0.7, 0.1, 22.0);
    test_cold_plasma_cutoffs<SOLVER> (r, dir);
    extra_ordinary_wave<SOLVER> (r, dir);
}

#endif /* ref_scenarios_hpp */
