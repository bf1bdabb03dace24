"""GPU parity: the HIP path (through the C ABI) against the CPU oracle that executes the
same reference DAG in strict IEEE arithmetic (oracle/gfir_interp.c), on identical inputs.

Tolerance: the north star asks 1e-6 relative fp64.  Every operation of the lowered kernel is
an IEEE add/sub/mul/div/fma/sqrt of the reference DAG, so these tests hold the GPU to
BIT-EXACT equality wherever no pow() is involved and to 1e-9 where it is.
"""
import os

import numpy as np
import pytest

from conftest import STATE, WORKLOADS, bench_state, random_plasma_state

pytestmark = pytest.mark.gpu


def _assert_close(a, b, rtol=1.0e-9):
    """The one non-IEEE-exact operation of these graphs is pow(x, 1.5) (device libm vs glibc,
    <= 1 ulp apart); a 1-ulp input change can surface at 1e-16*|largest term| in an output
    that is a cancelling sum, so the absolute floor is set from the output's scale."""
    scale = np.abs(b).max()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=1.0e-13*scale)


def _oracle(name):
    from oracle import gfir
    return gfir.Item(os.path.join(WORKLOADS, name))


def _run_item(name, columns, outputs, steps=1):
    """Run one workload item on the GPU on copies of `columns`; returns (columns, outputs)."""
    import graph_framework_amd as gfa
    n = columns[0].size
    ctx = gfa.Context(0)
    kernel = ctx.add_kernel(os.path.join(WORKLOADS, name), n)
    ctx.compile()
    in_keys = ["in%d" % i for i in range(len(columns))]
    out_keys = ["out%d" % i for i in range(outputs)]
    kernel.create_kernel_call(in_keys, out_keys, [c.copy() for c in columns])
    kernel.run(steps)
    ctx.wait()
    cols = [ctx.copy_to_host(k, np.empty_like(columns[0])) for k in in_keys]
    outs = [ctx.copy_to_host(k, np.empty_like(columns[0])) for k in out_keys]
    info = kernel.info()
    ctx.close()
    return cols, outs, info


def test_efit_test_kernel_matches_gold_and_oracle(efit_gold):
    """graph_tests/efit_test.cpp:132-187 on the GPU: same tolerances as the reference test."""
    g = efit_gold
    R, Z = np.meshgrid(g["r_grid"], g["z_grid"], indexing="ij")
    x, z = R.ravel().copy(), Z.ravel().copy()
    y = np.zeros_like(x)
    _, outs, _ = _run_item("efit_test_kernel_f64.gfir", [x, y, z], 6)

    def err2(test, expected):
        d = test - expected
        e = d/np.where(d == 0, 1.0, expected)
        return (e*e).max()

    assert err2(outs[0], g["bx_grid"].ravel()) <= 4.0e-12
    assert err2(outs[1], g["by_grid"].ravel()) <= 4.0e-23
    assert err2(outs[2], g["bz_grid"].ravel()) <= 1.0e-12
    assert err2(outs[3], g["ne_grid"].ravel()) <= 5.0e-13
    assert err2(outs[4], g["te_grid"].ravel()) <= 5.0e-13
    assert (outs[5]**2).max() <= 1.0e-20

    ref_outs, _ = _oracle("efit_test_kernel_f64.gfir").run([x.copy(), y.copy(), z.copy()])
    for a, b in zip(outs, ref_outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n", [1, 63, 1000])
def test_dispersion_kernel_bit_exact(n):
    s = random_plasma_state(n, seed=7 + n)
    cols = [s[k] for k in STATE]
    _, outs, _ = _run_item("dispersion_kernel_f64.gfir", cols, 8)
    ref_outs, _ = _oracle("dispersion_kernel_f64.gfir").run([c.copy() for c in cols])
    for a, b in zip(outs, ref_outs):
        _assert_close(a, b)


def test_loss_kernel_one_pass_bit_exact():
    s = random_plasma_state(513, seed=3)
    cols = [s[k] for k in STATE]
    new_cols, outs, _ = _run_item("loss_kernel_kx_f64.gfir", cols, 1)
    ref_cols = [c.copy() for c in cols]
    ref_outs, _ = _oracle("loss_kernel_kx_f64.gfir").run(ref_cols)
    np.testing.assert_array_equal(outs[0], ref_outs[0])
    for a, b in zip(new_cols, ref_cols):
        np.testing.assert_array_equal(a, b)


def test_bench_ray_newton_and_1000_steps_match_reference():
    """xrays_bench ICs: Newton init then 1000 RK4 steps; golden values are the reference's own
    output (reproduced bit-for-bit by oracle/_ref/gf_ref)."""
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    n = 300
    solve = Rk4ColdPlasmaEfit(bench_state(n))
    solve.init("kx")
    assert solve.newton_iterations == 24
    assert solve.host["kx"][0] == -500.00000357884727
    solve.compile()
    for _ in range(1000):
        solve.step()
    host = solve.sync_host()
    golden = dict(t=1.0000000000000007, x=2.6503724167948581, y=1.3098653092768473e-05,
                  z=5.7992231932273706e-04, kx=499.75806004711882, ky=2.4699010015446147e-03,
                  kz=2.7212694836099862)
#  Bit for bit: on this ray no pow(x, 1.5) rounding difference reaches the state in 1000 steps.
    for k, v in golden.items():
        assert np.all(host[k] == host[k][0])
        assert host[k][0] == v, (k, float(host[k][0]).hex(), float(v).hex())
    assert solve.check_residual(0) == 7.7779641626949096e-13


@pytest.mark.parametrize("body", ["assembly", "compiled"])
def test_solver_kernel_random_rays_vs_oracle(monkeypatch, body):
    """Both lowerings of the RK4 item: the assembly body (csrc/asm_body.hpp, the default) and the body hipcc compiles
    (GFHIP_ASM=0: shared reciprocals, window checks, the `_ieee` function inside the kernel)."""
    if body == "compiled":
        monkeypatch.setenv("GFHIP_ASM", "0")
    s = random_plasma_state(257, seed=11)
    cols = [s[k] for k in STATE]
    new_cols, outs, info = _run_item("solver_kernel_f64.gfir", cols, 1, steps=5)
    ref_cols = [c.copy() for c in cols]
    item = _oracle("solver_kernel_f64.gfir")
    ref_outs, _ = item.run(ref_cols, steps=5)
    for a, b in zip(new_cols, ref_cols):
        _assert_close(a, b)
    _assert_close(outs[0], ref_outs[0], rtol=1.0e-7)
    assert info.num_instructions == 3878
    assert info.segments == (1 if body == "assembly" else 0) and (info.vgprs == 256 if body == "assembly" else info.vgprs > 400)


def test_full_size_ensembles_through_size_independent_properties(golden_ref):
    """BASELINE.json's sizes (configs[1]: 1e6 identical rays; configs[2]: 1e7 incoherent rays),
    where the oracle cannot follow: the properties every ray-independent kernel must keep.
      * 1e6 identical rays, 100 steps: every lane holds the bits of lane 0, and lane 0 holds the
        reference's record of step 100 (tests/golden/ref_golden.npz);
      * incoherent rays: a ray's trajectory depends on nothing but its own state — the ensemble
        run in one piece, in reversed order and as two unequal shards gives every ray the same
        bits (tile, lane, workgroup and grid position do not matter) — and 64 rays picked across
        the 1e7 equal the oracle's trace of each of them alone, bit for bit."""
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, cli_distribution
    n = 1000000
    solve = Rk4ColdPlasmaEfit(bench_state(n))
    solve.init("kx")
    assert solve.newton_iterations == int(golden_ref["bench_newton_iterations"])
    solve.compile()
    steps = [int(s) for s in golden_ref["bench_steps"]]
    target = 100 if 100 in steps else steps[-1]
    for _ in range(target):
        solve.step()
    host = solve.sync_host()
    record = golden_ref["bench_records"][steps.index(target)]
    for k, expected in zip(STATE, record[:8]):
        assert np.all(host[k] == host[k][0]), k
        assert host[k][0] == expected, (k, host[k][0], expected)
    assert solve.work.context.flags() == 0

    def trace(state, count=20):
        run = Rk4ColdPlasmaEfit(state)
        run.init("kx", per_ray=True)            # per-ray Newton: no coupling through a global max
        run.compile()
        for _ in range(count):
            run.step()
        out = {k: v.copy() for k, v in run.sync_host().items()}
        out["residual"] = run.residual()
        assert run.work.context.flags() == 0
        return out

    rays = cli_distribution(10000000, seed=3)
    whole = trace(rays)
    backwards = trace({k: np.ascontiguousarray(v[::-1]) for k, v in rays.items()})
    cut = 3333337
    first = trace({k: np.ascontiguousarray(v[:cut]) for k, v in rays.items()})
    second = trace({k: np.ascontiguousarray(v[cut:]) for k, v in rays.items()})
    for k in list(STATE) + ["residual"]:
        assert np.array_equal(whole[k], backwards[k][::-1], equal_nan=True), k
        assert np.array_equal(whole[k][:cut], first[k], equal_nan=True), k
        assert np.array_equal(whole[k][cut:], second[k], equal_nan=True), k
    assert np.isfinite(whole["x"]).all() and np.ptp(whole["kz"]) > 1.0      # a genuinely incoherent beam
    picked = np.linspace(0, rays["x"].size - 1, 64).astype(np.int64)
    columns = [np.ascontiguousarray(rays[k][picked]) for k in STATE]
    loss, item = _oracle("loss_kernel_kx_f64.gfir"), _oracle("solver_kernel_f64.gfir")
    for i in range(picked.size):                          # the converge loop on each ray alone
        single = [c[i:i + 1].copy() for c in columns]
        loss.converge(single)
        for c, value in zip(columns, single):
            c[i] = value[0]
    outs, _ = item.run(columns, steps=20)
    for k, expected in zip(STATE, columns):
        assert np.array_equal(whole[k][picked], expected), k
    assert np.array_equal(whole["residual"][picked], outs[0])


def _same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def _every_single_step_agrees(item, start, steps, rtol):
    """For rays whose 100-step results differ in their last bits: step both sides ONE step at a
    time from the same state (the device is re-synchronised to the oracle after every step) and
    hold every component to `rtol`.  Returns the number of steps whose results were not bit-equal."""
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    state = [c.copy() for c in start]
    solve = Rk4ColdPlasmaEfit({k: c.copy() for k, c in zip(STATE, state)})
    solve.compile()
    events = 0
    for _ in range(steps):
        item.run(state, steps=1)
        solve.step(1)
        host = solve.sync_host()
        for k, expected in zip(STATE, state):
            assert np.array_equal(np.isfinite(host[k]), np.isfinite(expected)), k
            ok = np.isfinite(expected)
            assert np.allclose(host[k][ok], expected[ok], rtol=rtol, atol=0.0), k
        events += int(any((~_same(host[k], expected)).any() for k, expected in zip(STATE, state)))
        for k, expected in zip(STATE, state):
            host[k][:] = expected
        solve.sync_device()
    return events


def test_incoherent_beam_500_steps():
    """The CLI example's beam (graph_driver/xrays.cpp:392-399: the reference's own mt19937_64 samples)
    followed for 500 RK4 steps against the oracle's evaluation of the reference DAG.

    Tolerance, stated: every node but one is bit-exact.  `pow(x, 1.5)` (4 per step) is the host libm's
    on the reference side — glibc's here, accurate to ~0.5002 ulp, i.e. NOT always the correctly
    rounded value — and the correctly rounded value on the device (tests/test_gpu_pow.py).  The two
    differ in the last bit of about 0.4 % of the calls (profiles/r02_fission_experiment.md), but the
    power enters sums with much larger terms, so the difference reaches the stored state about once
    in 1e7 calls: of this beam's 4096 x 500 x 4 calls one does (ray 2472, step 442:
    x = 0x1.346fd7824e0a9p+2, glibc 0.50017 ulp off).  Hence: per 100-step chunk all but
    at most 2 rays are bit-identical; a ray that is not is stepped one step at a time on both sides
    from the same state and every step is held to 1e-9 relative (the north star's bound is 1e-6),
    with at most 2 steps of the 100 not bit-equal; the oracle then follows the device for that ray.

    Rays the reference graph itself drives to non-finite values (none among these 4096 with the
    reference's samples; tests/test_gpu_division.py and the 1e7-ray test cover them) must become
    non-finite in the same chunk on both sides and raise the status flag (bit 0)."""
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, cli_distribution
    n = 4096
    rays = {k: np.ascontiguousarray(v[:n]) for k, v in cli_distribution(200000, seed=0).items()}
    solve = Rk4ColdPlasmaEfit({k: v.copy() for k, v in rays.items()})
    solve.init("kx", per_ray=True)
    solve.compile()
    cols = [rays[k].copy() for k in STATE]
    loss = _oracle("loss_kernel_kx_f64.gfir")
    for i in range(n):                                   # the converge loop on each ray alone
        single = [c[i:i + 1].copy() for c in cols]
        loss.converge(single)
        for c, s in zip(cols, single):
            c[i] = s[0]
    assert np.array_equal(solve.sync_host()["kx"], cols[5])
    item = _oracle("solver_kernel_f64.gfir")
    flagged_at = None
    rounding_events = 0
    for chunk in range(5):
        start = [c.copy() for c in cols]
        item.run(cols, steps=100, threads=8)
        solve.step(100)                                  # fused launch: same bits as 100 launches
        host = {k: v.copy() for k, v in solve.sync_host().items()}
        differ = np.zeros(n, dtype=bool)
        for k, expected in zip(STATE, cols):
            differ |= ~_same(host[k], expected)
        assert differ.sum() <= 2, (chunk, np.flatnonzero(differ))
        if differ.any():
            events = _every_single_step_agrees(item, [c[differ] for c in start], 100, rtol=1.0e-9)
            assert 1 <= events <= 2, events
            rounding_events += events
            for k, c in zip(STATE, cols):
                c[differ] = host[k][differ]
        if flagged_at is None and solve.work.context.flags():
            flagged_at = chunk
    assert rounding_events <= 3
    lost = ~np.isfinite(cols[2])
    assert lost.sum() < n//100
    if lost.any():
        assert flagged_at is not None
    assert solve.residual().max() > 1.0                  # the beam does hold rays far off the dispersion surface


def test_both_lowerings_of_the_rk4_item_agree_at_full_size():
    """Where the oracle cannot follow (1e7 incoherent rays x 400 steps, 3e12 divisions): the assembly body with its redo
    launch and the body hipcc compiles with its IEEE function are two independent lowerings of the same DAG — every
    element of the state comes out with the same bits, the rays that blow up and the lanes that leave the division
    window (status flags set) included.  `profiles/diag/asm/compare_bodies.py` runs each in a process of its own."""
    import json
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "diag", "asm", "compare_bodies.py")
    out = subprocess.run([sys.executable, script, "10000000", "400"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    report = json.loads(out.stdout.strip().splitlines()[-1])
    assert report["bit_identical"] and not any(report["elements_that_differ"].values())
    assert report["assembly"]["segments"] == 1 and report["compiled"]["segments"] == 0
    assert report["assembly"]["flags"] == report["compiled"]["flags"] != 0          # lanes did take the IEEE path
    assert report["assembly"]["non_finite_rays"] == report["compiled"]["non_finite_rays"]
