"""The scenarios of graph_tests/solver_test.cpp and graph_tests/physics_test.cpp, written once
against a small solver interface so that the CPU oracle (tests/test_oracle.py) and the HIP
backend (tests/test_gpu_physics.py) replay exactly what oracle/ref_physics.cpp replayed on the
reference's own graph layer when it wrote tests/golden/physics_golden.json.

Interface of `solve` (mirrors solver::solver_interface, graph_framework/solver.hpp:123-430):
    set(key, value[, index])   variable->set(...)
    get(key[, index])          variable->evaluate().at(index) after sync_host
    init(variable, tolerance)  solver_interface::init
    compile()                  solver_interface::compile
    step()                     solver_interface::step
    state()                    [ray][t w x y z kx ky kz residual]
Each scenario returns {"states": [...], "newton_iterations": [...], "holds": bool, ...}: `holds`
is the assertion of the reference test itself.
"""
import math

import numpy as np

Q = 1.602176634E-19
ME = 9.1093837015E-31
MU0 = math.pi*4.0E-7
EPSILON0 = 8.8541878138E-12
C = 1.0/math.sqrt(MU0*EPSILON0)


def solver_test(make, label, method, omega0, kx0, dt):
    """solver_test.cpp:28-60: Newton init for kx, five steps, residual < tolerance after each."""
    del dt                                                  # a constant of the exported graph
    solve = make("solver_test_%s_%s" % (label, method))
    solve.set("w", omega0)
    solve.set("kx", kx0)
    solve.set("ky", 0.25)
    solve.set("kz", 0.15)
    tolerance = 1.0E-30
    solve.init("kx", tolerance)
    states = [solve.state()]
    solve.compile()
    holds = True
    for _ in range(5):
        solve.step()
        states.append(solve.state())
        holds = holds and abs(states[-1][0][8]) < abs(tolerance)
    return {"states": states, "newton_iterations": solve.newton_iterations, "holds": holds}


SOLVER_TESTS = [(label, method, omega0, kx0, dt)
                for method in ("rk2", "rk4")
                for label, omega0, kx0, dt in (("simple", 0.5, 0.25, 1.0),
                                               ("gaussian_well", 0.5, 0.25, 0.00001),
                                               ("cold_plasma", 900.0, 1000.0, 0.5/10000.0))]


def constant(make):
    """physics_test.cpp:24-73 (fixed values for the clock-seeded random ones)."""
    solve = make("constant")
    for key, value in (("w", 0.7), ("kx", 0.4), ("x", 0.3), ("y", 0.5), ("z", 0.9)):
        solve.set(key, value)

    def invariant():
        return (solve.get("kx")*solve.get("x") + solve.get("ky")*solve.get("y") + solve.get("kz")*solve.get("z")
                - solve.get("w")*solve.get("t"))

    solve.init("kx")
    c0 = invariant()
    states = [solve.state()]
    solve.compile()
    for _ in range(10):
        solve.step()
    states.append(solve.state())
    return {"states": states, "newton_iterations": solve.newton_iterations,
            "holds": abs(c0 - invariant()) < 5.0E-15, "constant_before": c0, "constant_after": invariant()}


def wave_in_gradient(make, relation, method, tolerance=2.0E-29):
    """physics_test.cpp:86-150 (bohm_gross) / :163-221 (light_wave), rk4 and split_simplextic."""
    omega0, ne0, te = 600.0, 1.0E19, 1000.0
    omega2 = (ne0*0.9*Q*Q)/(EPSILON0*ME*C*C)
    omega2p = (ne0*0.1*Q*Q)/(EPSILON0*ME*C*C)
    vth2 = 2*1.602176634E-19*te/(ME*C*C)
    solve = make("%s_%s" % (relation, method))
    solve.set("w", 600.0)
    solve.set("kx", 1000.0 if relation == "bohm_gross" else 100.0)
    solve.set("x", -1.0)
    solve.init("kx")
    states = [solve.state()]
    solve.compile()
    for _ in range(20):
        solve.step()
    states.append(solve.state())
    time = solve.get("t")
    if relation == "bohm_gross":
        k0 = math.sqrt(2.0/3.0*(omega0*omega0 - omega2)/vth2)
        expected_x = -3.0/8.0*vth2*omega2p/(omega0*omega0)*time*time + 3.0/2.0*vth2/omega0*k0*time - 1.0
    else:
        k0 = math.sqrt(omega0*omega0 - omega2)
        expected_x = -omega2p/(4.0*omega0*omega0)*time*time + k0/omega0*time - 1.0
    diff_x = solve.get("x") - expected_x
    return {"states": states, "newton_iterations": solve.newton_iterations,
            "holds": abs(diff_x*diff_x) < abs(tolerance), "expected_x": expected_x}


def acoustic_wave(make, tolerance=2.0E-29):
    """physics_test.cpp:233-283."""
    mi, te = 3.34449469E-27, 1000.0
    vs = math.sqrt((Q*te + 3*Q*te)/mi)/C
    solve = make("acoustic_wave_rk4")
    solve.set("w", 1.0)
    solve.set("kx", 600.0)
    solve.init("kx", tolerance)
    states = [solve.state()]
    solve.compile()
    for _ in range(20):
        solve.step()
    states.append(solve.state())
    diff_x = solve.get("x")/solve.get("t") - vs
    return {"states": states, "newton_iterations": solve.newton_iterations,
            "holds": abs(diff_x*diff_x) < abs(tolerance), "vs": vs}


def o_mode_wave(make):
    """physics_test.cpp:341-383: Newton on x lands on the O-mode cut-off."""
    omega2 = (1.0E19*Q*Q)/(EPSILON0*ME*C*C)
    omega0 = 1000.0
    x_cut = (omega0*omega0 - 1.0 - omega2)/(omega2*0.1)
    solve = make("o_mode_wave")
    solve.set("w", omega0)
    solve.init("x")
    diff = solve.get("x") - x_cut
    return {"states": [solve.state()], "newton_iterations": solve.newton_iterations,
            "holds": abs(diff*diff) < 8.0E-10, "x_cut": x_cut}


def reflection(make, tolerance=2.0E-29, n0=0.7, x0=0.1, kx0=22.0):
    """physics_test.cpp:498-546: a cold-plasma ray launched just below its cut-off turns back."""
    omega_ce = -Q/(ME*C)
    solve = make("reflection")
    solve.set("w", omega_ce)
    solve.set("kz", n0*omega_ce)
    solve.set("x", x0)
    solve.init("x", tolerance)
    cutoff_location = solve.get("x")
    solve.set("x", cutoff_location - 0.00001*cutoff_location)
    solve.set("kx", kx0)
    solve.init("kx", tolerance)
    states = [solve.state()]
    solve.compile()
    max_x = solve.get("x")
    holds = True
    steps = 0
    while True:
        solve.step()
        steps += 1
        new_x = solve.get("x")
        max_x = max(new_x, max_x)
        holds = holds and abs(max_x - cutoff_location) < 1.9E-6
        if max_x != new_x or steps >= 100000:
            break
    states.append(solve.state())
    return {"states": states, "newton_iterations": solve.newton_iterations, "holds": holds,
            "cutoff_location": cutoff_location, "max_x": max_x, "steps": steps}


def cold_plasma_cutoffs(make):
    """physics_test.cpp:395-485, two rays (O-mode, X-mode) against the slab_density cut-offs."""
    solve = make("cold_plasma_cutoffs", 2)
    solve.set("w", 1100.0)
    solve.set("x", 25.0, 0)
    solve.set("x", 5.0, 1)
    solve.init("x")
    wpecut_pos = solve.get("x", 0)
    wrcut_pos = solve.get("x", 1)
    states = [solve.state()]
    solve.set("x", 0.0)
    solve.set("kx", 1000.0, 0)
    solve.set("kx", 500.0, 1)
    solve.init("kx")
    states.append(solve.state())
    solve.compile()
    steps_first = 0
    while abs(solve.get("t")) < 30.0:
        solve.step()
        steps_first += 1
    states.append(solve.state())
    first = wrcut_pos < solve.get("x", 0) < wpecut_pos and solve.get("x", 1) < wrcut_pos

    solve.set("w", 800.0)
    solve.set("x", 25.0, 0)
    solve.set("x", 5.0, 1)
    solve.set("kx", 0.0)
    solve.set("t", 0.0)
    solve.init("x", 5.0E-30)
    wpecut_pos = solve.get("x", 1)
    states.append(solve.state())
    solve.set("x", 0.0)
    solve.set("kx", 500.0, 0)
    solve.set("kx", 1500.0, 1)
    solve.init("kx")
    states.append(solve.state())
    steps_second = 0
    while abs(solve.get("t")) < 60.0:
        solve.step()
        steps_second += 1
    states.append(solve.state())
    second = solve.get("x", 0) < wpecut_pos and solve.get("x", 1) > wpecut_pos
    return {"states": states, "newton_iterations": solve.newton_iterations, "holds": first and second,
            "wrcut_pos": wrcut_pos, "wpecut_pos_second": wpecut_pos, "steps": [steps_first, steps_second]}


def extra_ordinary_wave(make):
    """No reference test; the graph is pinned (oracle/ref_physics.cpp extra_ordinary_wave)."""
    solve = make("extra_ordinary_wave_rk4", 3)
    solve.set("w", 1500.0)
    solve.set("kx", 1200.0)
    for i, (ky, x) in enumerate(((0.0, 0.0), (40.0, 1.5), (-25.0, -2.0))):
        solve.set("ky", ky, i)
        solve.set("x", x, i)
    solve.init("kx")
    states = [solve.state()]
    solve.compile()
    for _ in range(10):
        solve.step()
    states.append(solve.state())
    return {"states": states, "newton_iterations": solve.newton_iterations, "holds": True}


def all_scenarios():
    """(golden key, callable(make))"""
    out = []
    for label, method, omega0, kx0, dt in SOLVER_TESTS:
        out.append(("solver_test_%s_%s" % (label, method),
                    lambda make, a=(label, method, omega0, kx0, dt): solver_test(make, *a)))
    out.append(("constant", constant))
    for relation in ("bohm_gross", "light_wave"):
        for method in ("rk4", "split"):
            out.append(("%s_%s" % (relation, method), lambda make, a=(relation, method): wave_in_gradient(make, *a)))
    out.append(("acoustic_wave_rk4", acoustic_wave))
    out.append(("o_mode_wave", o_mode_wave))
    out.append(("reflection", reflection))
    out.append(("cold_plasma_cutoffs", cold_plasma_cutoffs))
    out.append(("extra_ordinary_wave_rk4", extra_ordinary_wave))
    return out


def uses_exp(name):
    """Items whose graph contains an exp node (gaussian profiles): glibc and ocml may differ in
    the last bit there; everything else is +,-,*,/,fma only and must match bit for bit."""
    return "gaussian_well" in name or name.startswith("solver_test_cold_plasma")


def compare(result, golden, exact):
    """Assert a replay equals the golden record of the reference-backed oracle."""
    assert result["newton_iterations"] == golden["newton_iterations"]
    assert len(result["states"]) == len(golden["states"])
    for mine, theirs in zip(result["states"], golden["states"]):
        mine = np.asarray(mine, dtype=np.float64)
        theirs = np.asarray(theirs, dtype=np.float64)
        if exact:
            assert np.array_equal(mine, theirs), (mine, theirs)
        else:
            np.testing.assert_allclose(mine[:, :8], theirs[:, :8], rtol=1.0e-12, atol=1.0e-13)
            assert np.all(np.abs(mine[:, 8]) <= np.maximum(4.0*np.abs(theirs[:, 8]), 1.0e-28))
    if "steps" in golden:
        assert result["steps"] == golden["steps"]
    assert result["holds"] == golden["reference_assertion_holds"]
