"""The division contract of the lowering (graph_framework_amd/csrc/prelude.hpp) at its edges.

The reference divides with the compiler's IEEE `/` (arithmetic.hpp:3508); the lowering shares one
refined reciprocal per denominator and must still return the IEEE quotient for EVERY operand:
lanes whose operands leave the window in which the shared sequence is the IEEE one redo their
pass with the compiler's division.  Checked here bit for bit against the CPU oracle on operands
the benchmark never sees: denominators of 2^+-600 (fp64) / 2^+-110 (fp32), zero and infinite
denominators, zero numerators of either sign, infinite and overflowing numerators, and — in the
`checked` mode, which also tracks every numerator, and in fp32, whose quotients are rounded from an
fp64 product and need no condition on the numerator — subnormal and near-subnormal numerators.
"""
import numpy as np
import pytest

import gfir_random
from oracle import gfir

pytestmark = pytest.mark.gpu


def _bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def _same(got, want):
    """Bit equality (the sign of a zero included); NaNs must coincide (payloads may differ
    between the host's and the device's arithmetic)."""
    nan = np.isnan(want)
    return np.array_equal(np.isnan(got), nan) and np.array_equal(_bits(got)[~nan], _bits(want)[~nan])


def _operands(dtype, tiny_numerators):
    real = np.float64 if dtype == "f64" else np.float32
    big, small = (600, -600) if dtype == "f64" else (110, -110)
    info = np.finfo(real)
    denominators = [3.0, -3.0, 2.0**big, -2.0**big, 2.0**small, 2.0**small*1.5, 0.0, -0.0, np.inf, -np.inf,
                    float(info.max), float(info.tiny), float(info.tiny)/4, 1.0e-30, 7.0e10]
    numerators = [1.0, -1.0, 0.0, -0.0, 5.5, np.inf, -np.inf, float(info.max), float(info.max)/3, 1.0e-20]
    if tiny_numerators:
#  what the default mode does not track (prelude.hpp): non-zero numerators below 2^-969 / 2^-102
#  and quotients below the normal range
        numerators += [2.0**small, 1.0e-30, float(info.tiny), float(info.tiny)/8, -float(info.tiny)/1024,
                       float(np.nextafter(real(0), real(1))),
                       2.0**(-980 if dtype == "f64" else -105), 3.0*2.0**(-1000 if dtype == "f64" else -120)]
    with np.errstate(over="ignore", under="ignore"):
        n0, d0 = np.meshgrid(np.array(numerators, dtype=real), np.array(denominators, dtype=real), indexing="ij")
    n0, d0 = n0.ravel().copy(), d0.ravel().copy()
    rng = np.random.default_rng(17)
#  second numerator / denominator / free input: ordinary values with a sprinkling of the specials
    n1 = rng.uniform(-2.0, 2.0, n0.size).astype(real)
    d1 = rng.uniform(0.5, 4.0, n0.size).astype(real)
    x = rng.uniform(-1.0, 1.0, n0.size).astype(real)
    n1[::7] = 0.0
    n1[3::11] = -0.0
    d1[5::13] = np.array(denominators, dtype=real)[rng.integers(0, len(denominators), d1[5::13].size)]
    x[2::9] = 0.0
    return [n0, n1, d0, d1, x]


@pytest.mark.parametrize("mode", ["shared", "checked", "ieee"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_division_at_the_edges_is_the_ieee_quotient(monkeypatch, tmp_path, dtype, mode):
    from graph_framework_amd import Context
    monkeypatch.setenv("GFHIP_DIVISION", mode)
    monkeypatch.setenv("GFHIP_CACHE_DIR", str(tmp_path))
    blob = gfir_random.division_stress_item(dtype)
    oracle_item = gfir.Item(blob)
#  fp32 quotients go through fp64 (prelude.hpp) and carry no condition on the numerator: the tiny
#  numerators are part of every fp32 run
    columns = _operands(dtype, tiny_numerators=(mode != "shared" or dtype == "f32"))
    rays = columns[0].size

    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    in_keys, out_keys = ["n0", "n1", "d0", "d1", "x"], ["q0", "q1", "q2", "q3", "mix"]
    kernel.create_kernel_call(in_keys, out_keys, columns)
    expected = [c.copy() for c in columns]
    with np.errstate(all="ignore"):
        for launch_steps in (1, 2):
            expected_out, _ = oracle_item.run(expected, steps=launch_steps)
            kernel.run(launch_steps)
            context.wait()
            for key, want in zip(in_keys + out_keys, expected + expected_out):
                got = context.copy_to_host(key, np.empty(rays, dtype=oracle_item.np_dtype))
                assert _same(got, want), (key, launch_steps, np.flatnonzero(_bits(got) != _bits(want))[:8])
#  Lanes did leave the window (bit 0) and stored zeros that came from quotients (bit 1): the status
#  bits say that the IEEE function ran (never in `ieee` mode, which has no other path).
#  fp32 has no stored-zero rule (the fp64 product carries the IEEE sign of a zero): bit 0 only.
    assert context.flags() == (0 if mode == "ieee" else (3 if dtype == "f64" else 1))
    context.close()


def test_in_window_operands_never_take_the_second_body():
    """Ordinary operands (the fuzz items' range) must not raise the status bit."""
    from graph_framework_amd import Context
    blob = gfir_random.division_stress_item("f64")
    rng = np.random.default_rng(3)
    rays = 1000
    columns = [rng.uniform(0.5, 2.0, rays) for _ in range(5)]
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    kernel.create_kernel_call(["n0", "n1", "d0", "d1", "x"], ["q0", "q1", "q2", "q3", "mix"], columns)
    kernel.run(3)
    context.wait()
    assert context.flags() == 0
    expected = [c.copy() for c in columns]
    expected_out, _ = gfir.Item(blob).run(expected, steps=3)
    for key, want in zip(["x", "q0", "mix"], [expected[4], expected_out[0], expected_out[4]]):
        assert np.array_equal(context.copy_to_host(key, np.empty(rays)), want), key
    context.close()


def test_fp32_quotients_are_ieee_for_random_bit_patterns():
    """fp32 quotients are rounded from an fp64 product (prelude.hpp).  The separation argument there
    covers every finite numerator and every finite non-zero denominator; the lanes with zero, infinite
    or NaN denominators take the IEEE function.  Here: four million operand pairs drawn as random BIT
    PATTERNS (every exponent, subnormals, infinities and NaNs included) plus pairs built to land next to
    rounding boundaries, against the host's IEEE division, bit for bit."""
    from graph_framework_amd import Context
    rng = np.random.default_rng(23)
    b = gfir_random.Builder(rng, "f32", 2)
    n_node, d_node = b.inputs
    blob = gfir_random.serialize(b, [b.emit(gfir_random.DIV, n_node, d_node)], [], 2, "quotient")
    count = 4*1024*1024
    n = rng.integers(0, 2**32, count, dtype=np.uint64).astype(np.uint32).view(np.float32)
    d = rng.integers(0, 2**32, count, dtype=np.uint64).astype(np.uint32).view(np.float32)
#  quotients next to a boundary: n = fl(m*d) for m a float or a midpoint of two floats, then one ulp off
    m = rng.uniform(1.0, 2.0, 65536).astype(np.float32)
    half = (m.astype(np.float64) + np.spacing(m).astype(np.float64)/2)
    dd = rng.uniform(1.0, 2.0, 65536).astype(np.float32)
    near = (half*dd.astype(np.float64)).astype(np.float32)
    n[:65536], d[:65536] = near, dd
    n[65536:131072], d[65536:131072] = np.nextafter(near, np.float32(4.0)), dd
    n[131072:196608], d[131072:196608] = (m.astype(np.float64)*dd.astype(np.float64)).astype(np.float32), dd
    context = Context(0)
    kernel = context.add_kernel(blob, count)
    context.compile()
    kernel.create_kernel_call(["n", "d"], ["q"], [n, d])
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("q", np.empty(count, dtype=np.float32))
    context.close()
    with np.errstate(all="ignore"):
        want = n/d
    assert _same(got, want), np.flatnonzero(_bits(got) != _bits(want))[:8]
    assert np.isnan(want).sum() > 1000 and (np.abs(want) < np.finfo(np.float32).tiny).sum() > 1000   # the sample does reach the edges


@pytest.mark.parametrize("exponent", [-1.0, 3.0, -2.0, 1.5])
def test_pow_of_a_quotient_sees_the_ieee_sign_of_zero(exponent):
    """pow tells -0 from +0 when its exponent is an odd integer.  1 + exp(-pow(n/d, -1)) is 1 on a
    quotient of +0 and inf on -0 — a finite, non-zero stored value that no other check would catch — so a pow
    base that comes from a shared-reciprocal quotient joins the zero check (codegen.hpp, GFIR_POW) unless its
    exponent is a constant that is not an odd integer.  Device against oracle, zeros of both signs among the
    numerators."""
    from graph_framework_amd import Context
    from test_gpu_generic import Item, INPUT, DIV, POW, EXP, SUB, ADD
    it = Item("f64", False, ["n", "d"], name="pow_zero_sign")
    n, d = it.emit(INPUT, a=0), it.emit(INPUT, a=1)
    q = it.emit(DIV, n, d)
    p = it.emit(POW, q, it.constant(exponent))
    one, zero = it.constant(1.0), it.constant(0.0)
    observed = it.emit(ADD, one, it.emit(EXP, it.emit(SUB, zero, p)))
    blob = it.blob([observed], [])            # the power itself is NOT stored: its infinity would fail the finite check
    rng = np.random.default_rng(23)
    rays = 1024
    num = rng.uniform(-2.0, 2.0, rays)
    den = rng.uniform(0.5, 3.0, rays)*rng.choice([-1.0, 1.0], rays)
    num[::5] = 0.0
    num[2::7] = -0.0
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    kernel.create_kernel_call(["n", "d"], ["observed"], [num, den])
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("observed", np.empty(rays))
    context.close()
    with np.errstate(all="ignore"):
        want, _ = gfir.Item(blob).run([num.copy(), den.copy()])
#  pow and exp are the device libm's on one side and glibc's on the other: what must coincide exactly is WHICH lanes
#  overflow (the -0 quotients under an odd negative exponent) and which give exactly 1 or 2; the rest to 1e-13.
    want = want[0]
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(np.isnan(got), np.isnan(want))
    finite = np.isfinite(want)
    assert np.array_equal(got[finite] == 1.0, want[finite] == 1.0) and np.array_equal(got[finite] == 2.0, want[finite] == 2.0)
    np.testing.assert_allclose(got[finite], want[finite], rtol=1.0e-13, atol=0.0)
    if exponent == -1.0:
        negative_zero = np.signbit(num/den) & (num/den == 0.0)
        assert negative_zero.sum() > 50 and np.isinf(want[negative_zero]).all() and np.isinf(got[negative_zero]).all()


def test_fast_division_mode_meets_the_trajectory_tolerance(monkeypatch, golden_ref):
    """GFHIP_DIVISION=fast — q = n*r with the refined shared reciprocal and NO residual step, no
    checks, no second body (prelude.hpp, codegen.hpp) — is opt-in and NOT bit-exact (every quotient within ~1.5 ulp).
    north_star's bound on the trajectories is 1e-6 relative.  Gate: the benchmark ray — which starts on the
    double root of D at the plasma edge — over 1000 steps against the reference's record, every state component
    within 1e-6 of its scale: MET.  The CLI example's incoherent beam (4096 rays) over 500 steps against the
    oracle: NOT met ray by ray, and not meetable (see below); what is asserted instead is that the mode is no
    further from the reference than the reference's own one-ulp neighbour.  bench.py reports this mode's rate under
    its own key, never as `value`."""
    from conftest import STATE, bench_state
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, cli_distribution, workload
    monkeypatch.setenv("GFHIP_DIVISION", "fast")
    solve = Rk4ColdPlasmaEfit(bench_state(256))
    solve.init("kx")
    solve.compile()
    steps = [int(s) for s in golden_ref["bench_steps"]]
    done, worst = 0, 0.0
    for count, record in zip(steps, golden_ref["bench_records"]):
        solve.step(count - done)
        done = count
        host = solve.sync_host()
        for k, expected in zip(STATE, record[:8]):
            assert np.all(host[k] == host[k][0])
            scale = max(abs(expected), 1.0e-3 if k in ("y", "z", "ky", "kz") else 0.0)
            worst = max(worst, abs(host[k][0] - expected)/scale if scale else 0.0)
    assert done == 1000 and worst <= 1.0e-6, worst
    info = solve.solver.kernel.info()
    assert info.scratch_bytes == 0
    solve.work.context.close()

    n = 4096
    rays = {k: np.ascontiguousarray(v[:n]) for k, v in cli_distribution(200000, seed=0).items()}
    solve = Rk4ColdPlasmaEfit({k: v.copy() for k, v in rays.items()})
    solve.init("kx", per_ray=True)
    solve.compile()
    cols = [solve.sync_host()[k].copy() for k in STATE]
    nudged = [c.copy() for c in cols]
    nudged[5] = nudged[5]*(1.0 + 2.0**-52)                      # kx one unit in the last place off
    item = gfir.Item(workload("solver_kernel"))
    item.run(cols, steps=500, threads=8)
    item.run(nudged, steps=500, threads=8)
    solve.step(500)
    host = solve.sync_host()
    finite = np.isfinite(cols[2]) & np.isfinite(host["x"]) & np.isfinite(nudged[2])
    assert finite.sum() > 0.99*n

    def distance(other):
        return np.max(np.stack([np.abs(o[finite] - e[finite])/np.abs(e[finite]).max() for o, e in zip(other, cols)]), axis=0)

#  MEASURED, and the reason this mode stays opt-in and labelled: over 500 steps a tenth of this beam is chaotic — the
#  BIT-EXACT arithmetic started one unit in the last place of kx away ends more than 1e-6 off on 10 % of the rays
#  (median 1e-13, 95th percentile 1e-2).  No arithmetic that differs from the reference's in any bit can meet 1e-6 on
#  those rays (the reference's own fast-math build does not), so the gate on the beam is: the fast mode
#  is no further from the reference than that one-ulp neighbour of the reference is.
    fast, neighbour = distance([host[k] for k in STATE]), distance(nudged)
    assert np.median(fast) <= 1.0e-11 and np.median(neighbour) <= 1.0e-11
    assert (fast > 1.0e-6).mean() <= 1.5*(neighbour > 1.0e-6).mean() + 0.01, ((fast > 1.0e-6).mean(), (neighbour > 1.0e-6).mean())
    well_conditioned = neighbour <= 1.0e-9
    assert well_conditioned.mean() > 0.7 and np.quantile(fast[well_conditioned], 0.99) <= 1.0e-6
    solve.work.context.close()
