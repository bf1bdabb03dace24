"""The power-absorption pass of xrays on the device: the work item of
absorption::weak_damping (complex<double>, SAFE_MATH, erfi; absorption.hpp:346-432) and bin_power's
`power` item (graph_driver/xrays.cpp:674-790), through the C ABI, against the oracle and against the
records of the reference's own graph layer (tests/golden/absorption_golden.npz,
make_absorption_golden.py).

Tolerances, stated: the device's complex arithmetic is specified operation by operation like the
oracle's (textbook product, Smith's quotient, nothing fused), so the only differences are the
device libm's exp/sin/cos/log/hypot against glibc's: 1e-12 relative in EACH part of kamp (imaginary
parts run from 20 down to 1e-303).  Against the golden (std::complex arithmetic, the reference's
special::erfi) the same bound holds.  `power` is real arithmetic with one exp() per record: 1e-14
relative on power, 1e-15 absolute on d_power = |difference of two powers| on the same kamp.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, STATE, WORKLOADS
from test_oracle import absorption_columns

pytestmark = pytest.mark.gpu

INPUTS = ("kamp", "kx", "ky", "kz", "x", "y", "z", "t", "w")


def _close(got, want, rtol):
    assert np.array_equal(got.imag == 0.0, want.imag == 0.0)
    np.testing.assert_allclose(got.real, want.real, rtol=rtol, atol=0.0)
    np.testing.assert_allclose(got.imag, want.imag, rtol=rtol, atol=0.0)


def test_weak_damping_item_matches_the_oracle_and_the_reference_records():
    from graph_framework_amd import Context
    from oracle import gfir
    golden = np.load(os.path.join(GOLDEN, "absorption_golden.npz"))
    path = os.path.join(WORKLOADS, "weak_damping_kimg_kernel_c64.gfir")
    columns = absorption_columns(golden["records"])
    rays = columns[0].size
    context = Context(0)
    kernel = context.add_kernel(path, rays)
    context.compile()
    kernel.create_kernel_call(INPUTS, [], columns)
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("kamp", np.empty(rays, dtype=np.complex128))
    for name, column in zip(INPUTS[1:], columns[1:]):                      # no other input is written
        assert np.array_equal(context.copy_to_host(name, np.empty(rays, dtype=np.complex128)), column), name
    context.close()
    expected = [c.copy() for c in columns]
    gfir.Item(path).run(expected, steps=1)
    _close(got, expected[0], 1.0e-12)
    _close(got, golden["kamp"].reshape(-1), 1.0e-12)
    assert (got.imag > 1.0).any()


def test_absorption_pass_over_a_trajectory_file(tmp_path):
    """trace file -> WeakDamping (kamp added to the file) -> bin_power (power, d_power added): the
    three-stage pipeline of graph_driver/xrays.cpp:1100-1105 on the golden trajectories."""
    from graph_framework_amd.absorption import bin_power, run_absorption
    from graph_framework_amd.output import RAY_VARIABLES, ResultFile
    from oracle import gfir
    golden = np.load(os.path.join(GOLDEN, "absorption_golden.npz"))
    records = golden["records"]
    saved, _, n = records.shape
    path = str(tmp_path / "result0.nc")
    trace = ResultFile(path, n)
    for name, _ in RAY_VARIABLES:
        trace.create_variable(name)
    column = {k: i for i, k in enumerate(STATE + ("residual",))}
    for r in range(saved):
        trace.write({name: records[r, column[key]] for name, key in RAY_VARIABLES})
    trace.close()

    run_absorption(path, saved - 1)
    bin_power(path, saved - 1)

    result = ResultFile(path)
    assert result.records == saved
    kamp = np.stack([result.read("kamp", r) + 1j*result.read("kamp", r, part=1) for r in range(saved)])
    power = np.stack([result.read("power", r) for r in range(saved)])
    d_power = np.stack([result.read("d_power", r) for r in range(saved)])
    result.close()
    _close(kamp.reshape(-1), golden["kamp"].reshape(-1), 1.0e-12)
    np.testing.assert_array_equal(power[0], np.ones(n))
#  the `power` item on the device's own kamp, replayed on the oracle (device exp vs glibc's)
    item = gfir.Item(os.path.join(WORKLOADS, "power_f64.gfir"))
    first = [records[0, 2].copy(), records[0, 3].copy(), records[0, 4].copy()]
    columns = [c.copy() for c in first] + first + [np.zeros(n), np.ones(n), np.zeros(n)]
    for r in range(1, saved):
        for c in range(3):
            columns[c] = records[r, 2 + c].copy()
        columns[6] = kamp[r].imag.copy()
        outs, _ = item.run(columns, steps=1)
        np.testing.assert_allclose(power[r], columns[7], rtol=1.0e-14, atol=0.0)
        np.testing.assert_allclose(d_power[r], outs[0], rtol=1.0e-13, atol=1.0e-15)
    np.testing.assert_allclose(power[1:], golden["power"][:, 0], rtol=1.0e-10, atol=0.0)
    assert 0.0 < power[-1].min() and power[-1].max() < 0.5


def _trajectory_file(tmp_path, golden):
    from graph_framework_amd.output import RAY_VARIABLES, ResultFile
    records = golden["records"]
    saved, _, n = records.shape
    path = str(tmp_path / "result0.nc")
    trace = ResultFile(path, n)
    for name, _ in RAY_VARIABLES:
        trace.create_variable(name)
    column = {k: i for i, k in enumerate(STATE + ("residual",))}
    for r in range(saved):
        trace.write({name: records[r, column[key]] for name, key in RAY_VARIABLES})
    trace.close()
    return path


def test_root_finder_pass_over_a_trajectory_file(tmp_path):
    """`--absorption_model=root_find` (xrays.cpp:634-642): absorption::root_finder's three items per
    stored record — the complex Newton converge item included — against the oracle on the same
    records and against the reference graph layer's roots (tolerance and what is comparable:
    tests/test_oracle.py::root_finder_matches)."""
    from graph_framework_amd.absorption import run_absorption
    from graph_framework_amd.output import ResultFile
    from oracle import gfir
    from test_oracle import ABSORPTION_INPUTS, root_finder_matches
    golden = np.load(os.path.join(GOLDEN, "absorption_golden.npz"))
    records = golden["records"]
    saved, _, n = records.shape
    path = _trajectory_file(tmp_path, golden)
    model = run_absorption(path, saved - 1, model="root_find")
    result = ResultFile(path)
    kamp = np.stack([result.read("kamp", r) + 1j*result.read("kamp", r, part=1) for r in range(saved)])
    result.close()
    assert len(model.iterations) == saved
    assert root_finder_matches(kamp, golden["root_kamp"], model.iterations, golden["root_iterations"]) >= 15

    init = gfir.Item(os.path.join(WORKLOADS, "root_find_init_kernel_c64.gfir"))
    loss = gfir.Item(os.path.join(WORKLOADS, "root_find_loss_kernel_c64.gfir"))
    final = gfir.Item(os.path.join(WORKLOADS, "root_find_final_kamp_c64.gfir"))
    expected, iterations = [], []
    for r in range(saved):
        columns = [np.zeros(n, dtype=np.complex128)] + [records[r, STATE.index(k)].astype(np.complex128) for k in ABSORPTION_INPUTS]
        init.run(columns[:7])
        count, _, _ = loss.converge(columns)
        final.run(columns[:7])
        expected.append(columns[0].copy())
        iterations.append(count)
    assert root_finder_matches(kamp, np.stack(expected), model.iterations, iterations, reference=golden["root_kamp"]) >= 15
#  outside the plasma nothing but exact arithmetic is involved: same bits, same iteration counts
    assert np.array_equal(kamp[:3], np.stack(expected)[:3]) and list(model.iterations[:3]) == iterations[:3]


def test_example_runs_the_three_stages_of_xrays(tmp_path):
    """examples/trace_rays.py: trace (rk4 x ordinary_wave on EFIT, the CLI beam) -> kamp -> power on one
    trajectory file, as graph_driver/xrays.cpp:1100-1105 chains them."""
    import subprocess
    import sys
    from conftest import ROOT
    from graph_framework_amd.output import ResultFile
    prefix = str(tmp_path / "rays")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "trace_rays.py"), "--rays", "512", "--dispersion",
                          "ordinary_wave", "--steps", "20000", "--sub-steps", "2000", "--output", prefix,
                          "--absorption-model", "weak_damping"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "transmitted power" in out.stdout
    result = ResultFile(prefix + "0.nc")
    assert result.records == 11 and result.num_rays == 512
    power = result.read("power", 10)
    x_first, x_last = result.read("x", 0), result.read("x", 10)
    kamp_imag = np.stack([result.read("kamp", r, part=1) for r in range(11)])
    result.close()
    assert np.isfinite(x_last).all() and (x_last < x_first).mean() > 0.9      # the beam travels inward (a few rays reflect)
    assert np.isfinite(kamp_imag).all() and kamp_imag.max() > 1.0              # and crosses the resonance
    absorbed = np.isfinite(power) & (power < 0.9)
    assert absorbed.sum() > 256, absorbed.sum()


def test_weak_damping_on_a_perturbed_ensemble():
    """1e5 rays scattered around the golden trajectories (positions by up to 2 cm, wave vectors and
    frequency by 1 %), so that zeta = (1 - ec/w)/(n_par v_t) sweeps through the resonance on many
    distinct values: the device against the oracle, 1e-11 relative in each part of kamp (imaginary parts
    below 1e-280 are compared absolutely: they are products of numbers that underflow on both sides)."""
    from graph_framework_amd import Context
    from oracle import gfir
    golden = np.load(os.path.join(GOLDEN, "absorption_golden.npz"))
    base = absorption_columns(golden["records"])
    rng = np.random.default_rng(5)
    n = 100000
    pick = rng.integers(0, base[0].size, n)
    columns = [np.zeros(n, dtype=np.complex128)]
    for name, column in zip(INPUTS[1:], base[1:]):
        values = column.real[pick].copy()
        if name in ("x", "y", "z"):
            values += rng.uniform(-0.02, 0.02, n)
        elif name != "t":
            values *= 1.0 + rng.uniform(-0.01, 0.01, n)
        columns.append(values.astype(np.complex128))
    path = os.path.join(WORKLOADS, "weak_damping_kimg_kernel_c64.gfir")
    context = Context(0)
    kernel = context.add_kernel(path, n)
    context.compile()
    kernel.create_kernel_call(INPUTS, [], columns)
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("kamp", np.empty(n, dtype=np.complex128))
    context.close()
    expected = [c.copy() for c in columns]
    gfir.Item(path).run(expected, steps=1, threads=8)
    want = expected[0]
    assert np.array_equal(np.isfinite(got.real), np.isfinite(want.real))
    ok = np.isfinite(want.real) & np.isfinite(want.imag)
    assert ok.sum() > 0.99*n
    np.testing.assert_allclose(got.real[ok], want.real[ok], rtol=1.0e-11, atol=0.0)
    np.testing.assert_allclose(got.imag[ok], want.imag[ok], rtol=1.0e-11, atol=1.0e-280)
    assert (np.abs(want.imag[ok]) > 1.0).sum() > 1000                # the resonance is inside the sample
