"""Work items off the hot path — complex base types, SAFE_MATH guards, random draws, index nodes —
on the GPU against the CPU oracle (oracle/gfir_interp.c: gfi_run_generic).

The reference's own tests for these flavours run in test_gpu_workflows.py (workflow_test.cpp,
piecewise_test.cpp, c_binding_test.c over the real jit::context); here the two restatements are
held to each other on items that use every node type at once.  Complex arithmetic is "parity
unpinned" against the reference (no fixture of it covers complex kernels): sums,
products, quotients, fma and integer powers are compared bit for bit, elementary functions to a
few ulp of their real building blocks.
"""
import struct

import numpy as np
import pytest

from oracle import gfir

pytestmark = pytest.mark.gpu

N = 0xFFFFFFFF
CONST, INPUT, ADD, SUB, MUL, DIV, FMA, SQRT, POWI, POW, SIN, COS, ATAN2, EXP, LOG, GATHER1, GATHER2, INDEX1, INDEX2, RANDOM, ERFI = range(21)
DTYPES = {"f32": 0, "f64": 1, "c32": 2, "c64": 3}
NUMPY = {"f32": np.float32, "f64": np.float64, "c32": np.complex64, "c64": np.complex128}


class Item:
    def __init__(self, dtype, safe, inputs, name="generic"):
        self.dtype, self.safe, self.inputs, self.name = dtype, safe, inputs, name
        self.code, self.tables = [], []
        self.complex = dtype.startswith("c")

    def emit(self, op, a=N, b=N, c=N, aux=0, reserved=0, imm=(0.0, 0.0, 0.0, 0.0)):
        self.code.append((op, a, b, c, aux, reserved, tuple(imm)))
        return len(self.code) - 1

    def constant(self, value):
        value = complex(value)
        return self.emit(CONST, imm=(value.real, value.imag if self.complex else 0.0, 0.0, 0.0))

    def table(self, rows, cols, rng):
        base = np.float32 if self.dtype in ("f32", "c32") else np.float64
        data = rng.uniform(-1.0, 1.0, (rows, cols)).astype(base).astype(np.float64)
        if self.complex:
            data = data + 1j*rng.uniform(-1.0, 1.0, (rows, cols)).astype(base).astype(np.float64)
        self.tables.append(data)
        return len(self.tables) - 1

    def blob(self, outputs, setters):
        name = self.name.encode() + b"\0"*(4 - len(self.name) % 4)
        out = struct.pack("<8s8I", b"GFIR0001", DTYPES[self.dtype], len(self.inputs), len(outputs), len(setters),
                          len(self.tables), len(self.code), len(name), 1 if self.safe else 0) + name
        for symbol in self.inputs:
            text = symbol.encode() + b"\0"*(4 - len(symbol) % 4)
            out += struct.pack("<I", len(text)) + text
        for t in self.tables:
            out += struct.pack("<II", *t.shape)
            if self.complex:
                out += np.ascontiguousarray(np.stack([t.real, t.imag], axis=-1), dtype="<f8").tobytes()
            else:
                out += np.ascontiguousarray(t, dtype="<f8").tobytes()
        for op, a, b, c, aux, reserved, imm in self.code:
            out += struct.pack("<6I4d", op, a, b, c, aux, reserved, *imm)
        out += struct.pack("<%dI" % len(outputs), *outputs)
        for value, target in setters:
            out += struct.pack("<II", value, target)
        return out


def _every_node_type(dtype, safe):
    rng = np.random.default_rng(7)
    it = Item(dtype, safe, ["a", "b", "v"])
    a, b = it.emit(INPUT, a=0), it.emit(INPUT, a=1)
    two, small = it.constant(2.0 + 0.5j), it.constant(0.25 - 0.125j)
    s = it.emit(ADD, a, b)
    d = it.emit(SUB, a, b)
    p = it.emit(MUL, a, b)
    q = it.emit(DIV, a, it.emit(ADD, b, two))
    f = it.emit(FMA, a, b, small)
    cube = it.emit(POWI, a, aux=3)
    exact = [s, d, p, q, f, cube]
    root = it.emit(SQRT, it.emit(ADD, it.emit(MUL, a, a), two))
    e = it.emit(EXP, small)
    ex = it.emit(EXP, it.emit(MUL, a, small))
    lg = it.emit(LOG, it.emit(ADD, it.emit(MUL, b, b), two))
    pw = it.emit(POW, it.emit(ADD, it.emit(MUL, a, a), two), small)
    sn, cs = it.emit(SIN, a), it.emit(COS, b)
    at = it.emit(ATAN2, it.emit(ADD, a, two), it.emit(ADD, b, small))
    t1 = it.table(1, 9, rng)
    t2 = it.table(4, 5, rng)
    g1 = it.emit(GATHER1, a, aux=t1, imm=(0.25, -1.0, 0.0, 0.0))
    g2 = it.emit(GATHER2, a, b, aux=t2, imm=(0.5, -1.0, 0.4, -1.0))
    i1 = it.emit(INDEX1, b, c=2, aux=16, imm=(0.125, -1.0, 0.0, 0.0))           # v[idx(b)]: a buffer of 16 elements
    i2 = it.emit(INDEX2, a, b, c=2, aux=4, reserved=4, imm=(0.5, -1.0, 0.5, -1.0))
    mix = it.emit(FMA, g1, g2, it.emit(MUL, i1, i2))
    close = [root, e, ex, lg, pw, sn, cs, at]
    broken = it.emit(SUB, it.emit(DIV, a, it.emit(SUB, a, a)), it.emit(DIV, b, it.emit(SUB, b, b)))   # inf - inf or 0/0
    update = it.emit(DIV, p, it.emit(ADD, b, two))                                # a <- a*b/(b + 2)
    return it.blob(exact + [g1, g2, i1, i2, mix, broken] + close, [(update, 0)]), len(exact) + 6, len(close)


@pytest.mark.parametrize("safe", [False, True], ids=["plain", "safe_math"])
@pytest.mark.parametrize("dtype", ["f64", "f32", "c64", "c32"])
def test_every_node_type_against_the_oracle(dtype, safe):
    from graph_framework_amd import Context
    blob, exact, close = _every_node_type(dtype, safe)
    oracle_item = gfir.Item(blob)
    real = NUMPY[dtype]
    rng = np.random.default_rng(11)
    rays = 777

    def values(count):
        x = rng.uniform(-1.0, 1.0, count)
        if dtype.startswith("c"):
            x = x + 1j*rng.uniform(-1.0, 1.0, count)
        return x.astype(real)

    a, b, v = values(rays), values(rays), values(rays)
    a[::13] = 0                                  # what the SAFE_MATH guards look at
    b[5::17] = 0
    columns = [a, b, v]
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    out_keys = ["o%d" % i for i in range(exact + close)]
    kernel.create_kernel_call(["a", "b", "v"], out_keys, columns)
    expected = [c.copy() for c in columns]
    tight = 2.0e-15 if dtype in ("f64", "c64") else 2.0e-6
    loose = 1.0e-12 if dtype in ("f64", "c64") else 2.0e-5
    with np.errstate(all="ignore"):
        for launch in range(2):
            expected_out, _ = oracle_item.run(expected)
            kernel.run(1)
            context.wait()
            got_a = context.copy_to_host("a", np.empty(rays, dtype=real))
            assert np.array_equal(got_a, expected[0], equal_nan=True), launch
            for o, key in enumerate(out_keys):
                got = context.copy_to_host(key, np.empty(rays, dtype=real))
                want = expected_out[o]
                if o < exact:
                    assert np.array_equal(got, want, equal_nan=True), (key, launch)
                else:
                    assert np.array_equal(np.isnan(got), np.isnan(want)), (key, launch)
                    ok = ~np.isnan(want)
                    np.testing.assert_allclose(got[ok], want[ok], rtol=loose, atol=tight, err_msg=key)
            if safe:                             # NaN never reaches memory under SAFE_MATH (cpu_context.hpp:530-547)
                broken = context.copy_to_host(out_keys[exact - 1], np.empty(rays, dtype=real))
                assert not np.isnan(broken).any()
    context.close()


@pytest.mark.parametrize("dtype,safe", [("f64", False), ("f32", False), ("c64", False), ("f64", True)])
def test_random_draws_follow_the_reference_sequence(dtype, safe):
    """random_node (random.hpp:296-339): MT19937, one state per lane of 1024, advanced one word per
    draw; every USE of the node is a draw (the serializer emits one GFIR_RANDOM per use).  Element e
    draws from state e % 1024, launches continue the sequences."""
    from graph_framework_amd import Context
    it = Item(dtype, safe, [], name="draws")
    token = it.emit(CONST)
    first = it.emit(RANDOM, token)
    second = it.emit(RANDOM, first)
    scaled = it.emit(MUL, it.emit(RANDOM, second), it.constant(2.0**-32))
    blob = it.blob([first, second, scaled], [])
    oracle_item = gfir.Item(blob)
    states = oracle_item.seed(0)
    rays = 3000                                  # three elements per state for most lanes
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    kernel.create_kernel_call([], ["r1", "r2", "r3"])
    kernel.set_random_state("state", states)
    real = NUMPY[dtype]
    for launch in range(2):
        expected, _ = oracle_item.run_sized(rays, [])
        kernel.run(1)
        context.wait()
        for key, want in zip(("r1", "r2", "r3"), expected):
            got = context.copy_to_host(key, np.empty(rays, dtype=real))
            assert np.array_equal(got, want), (key, launch)
        if launch == 0:
            assert expected[0][0].real == real(2357136044).real      # MT19937, seed 0, first word (c_binding_test.c:272)
            assert expected[0][1].real == real(1791095845).real      # seed 1
    context.close()


def test_complex_converge_item_takes_the_element_of_largest_modulus():
    """create_max_call on a complex output (cpu_context.hpp:314-318): std::max_element on std::abs."""
    from graph_framework_amd import Context
    it = Item("c64", False, ["z"], name="modulus")
    z = it.emit(INPUT, a=0)
    blob = it.blob([it.emit(MUL, z, z)], [])
    rays = 5000
    rng = np.random.default_rng(3)
    values = (rng.uniform(-1, 1, rays) + 1j*rng.uniform(-1, 1, rays)).astype(np.complex128)
    values[1234] = 3.0 - 4.0j
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    kernel.create_kernel_call(["z"], ["zz"], [values])
    assert kernel.run_max_complex() == (3.0 - 4.0j)**2
    context.close()


@pytest.mark.parametrize("dtype,tolerance", [("c64", 2.0e-14), ("c32", 2.0e-5)])
def test_erfi_on_the_device_meets_the_reference_erfi_test(dtype, tolerance):
    """erfi_node (math.hpp:1440) on the device against the reference-held fixture of special::erfi
    (graph_tests/test_erfi.nc, tests/golden/test_erfi.npz) with the rule and the tolerances of
    graph_tests/erfi_test.cpp:50-83, :89-90 (2e-14 double, 2e-5 float)."""
    import os
    from conftest import GOLDEN
    from graph_framework_amd import Context
    fixture = np.load(os.path.join(GOLDEN, "test_erfi.npz"))
    it = Item(dtype, False, ["z"], name="erfi")
    blob = it.blob([it.emit(ERFI, it.emit(INPUT, a=0))], [])
    real = NUMPY[dtype]
    z = (fixture["x"] + 1j*fixture["y"]).astype(real)
    context = Context(0)
    kernel = context.add_kernel(blob, z.size)
    context.compile()
    kernel.create_kernel_call(["z"], ["erfi"], [z])
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("erfi", np.empty(z.size, dtype=real))
    gold = fixture["re"] + 1j*fixture["img"]
    if dtype == "c32":                                   # float: the arguments the device saw
        gold = np.array([gfir.erfi(complex(v)) for v in z])
    with np.errstate(all="ignore"):
        error = np.abs(1.0 - got[5:].astype(np.complex128)/gold[5:])
    finite = np.isfinite(gold[5:].real) & np.isfinite(gold[5:].imag) & np.isfinite(got[5:].real) & np.isfinite(got[5:].imag)
    assert finite.sum() > 200 and error[finite].max() <= tolerance, error[finite].max()
    oracle_values, _ = gfir.Item(blob).run([z.copy()])
    np.testing.assert_allclose(got, oracle_values[0], rtol=1.0e-13 if dtype == "c64" else 1.0e-5, atol=0.0)
    context.close()


def test_erfi_branches_on_the_device():
    """The branches erfi takes before its general formula (prelude.hpp, special_functions.hpp:1495-1517)
    agree with the oracle's: arguments on the real axis give exactly real values (to 1e-13: exp and the
    Weideman sum are the device libm's), on the imaginary axis i erf(y), Re(z^2) < -750 gives -+i."""
    from graph_framework_amd import Context
    it = Item("c64", False, ["z"], name="erfi_branches")
    blob = it.blob([it.emit(ERFI, it.emit(INPUT, a=0))], [])
    z = np.array([1.0e-3, 0.7, -3.0, 12.0, 26.0, -26.5, 27.0, -27.0, 0.3j, -2.0j, 7.5j, 1.0 + 28.0j, 1.0 - 28.0j,
                  0.0, 5.0 + 1.0e-300j], dtype=np.complex128)
    context = Context(0)
    kernel = context.add_kernel(blob, z.size)
    context.compile()
    kernel.create_kernel_call(["z"], ["erfi"], [z])
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("erfi", np.empty(z.size, dtype=np.complex128))
    context.close()
    want = np.array([gfir.erfi(complex(v)) for v in z])
    assert np.array_equal(got.imag[:8], np.zeros(8)) and np.array_equal(got.real[8:11], np.zeros(3))
    np.testing.assert_allclose(got.real[:8], want.real[:8], rtol=1.0e-13, atol=0.0)
    np.testing.assert_allclose(got.imag[8:11], want.imag[8:11], rtol=4.0e-16, atol=0.0)
    assert np.array_equal(got[11:14], want[11:14])
    assert abs(got[14].real - want[14].real) <= 1.0e-12*abs(want[14].real)        # off the axis: the general formula


def test_erfi_small_arguments_on_the_device():
    """off the axes and for small |z| the device leaves the cancelling general formula for the
    series special::erf_complex uses there (special_functions.hpp:1534-1553; prelude.hpp gf_erfi).  Each
    part of erfi within 1e-14 relative of the exactly summed power series and of the oracle, in both
    series regions (the device's exp and the Weideman sum are its own libm's)."""
    from graph_framework_amd import Context
    from test_oracle import _erfi_series_exact
    it = Item("c64", False, ["z"], name="erfi_small")
    blob = it.blob([it.emit(ERFI, it.emit(INPUT, a=0))], [])
    points = []
    for scale in (1.0e-8, 1.0e-5, 1.0e-3, 9.0e-3):
        points += [complex(scale, scale), complex(scale, -3.0*scale), complex(-scale, 7.9*scale), complex(-0.3*scale, -scale)]
    for re in (1.1e-2, 5.0e-2, 0.2, 0.45, 2.4):
        for im in (1.0e-9, -1.0e-6, 1.0e-4, -4.9e-3 if re < 1.0 else -1.0e-3):
            points += [complex(re, im), complex(-re, im)]
    z = np.array(points, dtype=np.complex128)
    context = Context(0)
    kernel = context.add_kernel(blob, z.size)
    context.compile()
    kernel.create_kernel_call(["z"], ["erfi"], [z])
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("erfi", np.empty(z.size, dtype=np.complex128))
    context.close()
    series = np.array([_erfi_series_exact(complex(v), 90) for v in z])
    oracle = np.array([gfir.erfi(complex(v)) for v in z])
    for want in (series, oracle):
        np.testing.assert_allclose(got.real, want.real, rtol=1.0e-14, atol=0.0)
        np.testing.assert_allclose(got.imag, want.imag, rtol=1.0e-14, atol=0.0)


def test_indexed_input_longer_than_the_ensemble():
    """an input that index_1D nodes read keeps its own length (piecewise.hpp:1530-1575).  With 20
    rays and a 64-element buffer the Python path used to upload only the first 20 elements; the elements
    past the ensemble size are the ones read here.  A buffer shorter than the indexed length is refused."""
    from graph_framework_amd import Context, GfHipError
    it = Item("f64", False, ["b", "v"], name="long_index")
    b = it.emit(INPUT, a=0)
    picked = it.emit(INDEX1, b, c=1, aux=64, imm=(0.125, -1.0, 0.0, 0.0))       # v[clamp((b + 1)/0.125)]
    blob = it.blob([it.emit(ADD, picked, b)], [])
    rays = 20
    position = np.linspace(1.6, 6.9, rays)                                      # indices 20 .. 63: all past the ensemble size
    v = np.arange(64, dtype=np.float64)*3.0 + 0.5
    context = Context(0)
    kernel = context.add_kernel(blob, rays)
    context.compile()
    kernel.create_kernel_call(["b", "v"], ["picked"], [position, v])
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("picked", np.empty(rays))
    index = np.clip(((position + 1.0)/0.125), 0, 63).astype(np.int64)
    assert index.min() >= rays and index.max() == 63
    assert np.array_equal(got, v[index] + position)
    oracle_out, _ = gfir.Item(blob).run([position.copy(), v.copy()])
    assert np.array_equal(got, oracle_out[0])
    assert context.buffer_info("v")[0] == 64
    other = Context(0)
    short = other.add_kernel(blob, rays)
    other.compile()
    with pytest.raises(GfHipError):
        short.create_kernel_call(["b", "v"], ["picked"], [position, v[:32]])
    context.close()
    other.close()


def test_separate_max_reduction_skips_a_nan_in_its_alignment_head():
    """reduce.hip folds the elements in front of the first 16-byte boundary with the same `v > m`
    as the rest, so a NaN at element 1 or 2 is skipped as std::max_element skips it (cpu_context.hpp:306-322);
    only a NaN at element 0 is the maximum.  An item of > 1500 nodes has no `<name>_max` entry and takes the
    separate reduction kernel; the fp32 output is a tensor that starts 4 bytes past a 16-byte boundary."""
    import torch
    from graph_framework_amd import Context
    it = Item("f32", False, ["a"], name="long_chain")
    node, zero = it.emit(INPUT, a=0), it.constant(0.0)
    for _ in range(1600):
        node = it.emit(ADD, node, zero)
    blob = it.blob([node], [])
    n = 5000
    rng = np.random.default_rng(3)
    for nan_at, expect_nan in ((1, False), (2, False), (0, True)):
        a = rng.uniform(-4.0, 3.0, n).astype(np.float32)
        a[nan_at] = np.nan
        out = torch.zeros(n + 1, dtype=torch.float32, device="cuda")
        context = Context(0, torch.cuda.current_stream().cuda_stream)
        context.set_buffer("o", out[1:])
        kernel = context.add_kernel(blob, n)
        context.compile()
        assert kernel.info().num_instructions > 1500
        kernel.create_kernel_call(["a"], ["o"], [a])
        value = kernel.run_max()
        torch.cuda.synchronize()
        if expect_nan:
            assert np.isnan(value)
        else:
            assert np.float32(value) == np.nanmax(a), (nan_at, value)
        context.close()
