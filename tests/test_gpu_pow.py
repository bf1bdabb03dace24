"""pow(x, 1.5) on the device is the correctly rounded value of x^(3/2).

The reference emits `pow(x, 1.5)` and leaves its value to the libm behind the backend (the host's on
cpu_context, CUDA's on cuda_context): glibc's is accurate to ~0.502 ulp, i.e. it misses the
correctly rounded value in a fraction of a percent of the calls.  The lowering computes x*sqrt(x) with the square
root's rounding error carried into the product (graph_framework_amd/csrc/prelude.hpp,
gf_pow_three_halves) and is held here to the exact value, rounded once — computed with integer
arithmetic below — on random arguments over the plasma terms' range and beyond, and on the one
argument of the CLI beam where glibc and the exact value part (tests/test_gpu_parity.py).
"""
import math
from fractions import Fraction

import numpy as np
import pytest

import gfir_random

pytestmark = pytest.mark.gpu

POW = 9


def _three_halves_correctly_rounded(x):
    """round-to-nearest-even of x^(3/2), exact: compare candidates' squares with x^3 in rationals."""
    fx = Fraction(x)
    cube = fx*fx*fx
    guess = math.sqrt(x)*x
    lo = guess
    while Fraction(lo)**2 > cube:
        lo = math.nextafter(lo, 0.0)
    while Fraction(math.nextafter(lo, math.inf))**2 <= cube:
        lo = math.nextafter(lo, math.inf)
    hi = math.nextafter(lo, math.inf)                    # lo <= x^1.5 < hi
    if Fraction(lo)**2 == cube:
        return lo
    mid = (Fraction(lo) + Fraction(hi))/2
    if mid*mid == cube:                                   # a tie cannot be the square root of a dyadic cube ... but be exact
        return lo if (np.float64(lo).view(np.uint64) & 1) == 0 else hi
    return lo if mid*mid > cube else hi


def _item():
    b = gfir_random.Builder(np.random.default_rng(0), "f64", 1)
    x, = b.inputs
    y = b.emit(POW, x, b.constant(1.5))
    return gfir_random.serialize(b, [y], [], 1, "pow_three_halves")


def test_pow_three_halves_is_correctly_rounded():
    from graph_framework_amd import Context
    rng = np.random.default_rng(11)
    x = np.concatenate([
        rng.uniform(0.5, 40.0, 6000),                                      # the plasma terms' range
        np.exp(rng.uniform(np.log(1.0e-190), np.log(1.0e190), 3000)),      # far beyond it
        np.array([float.fromhex("0x1.346fd7824e0a9p+2"),                   # glibc: 0.50017 ulp off here
                  1.0, 4.0, 2.25, 1.0e-190, 1.0e190, 0.0, np.inf]),
    ])
    context = Context(0)
    kernel = context.add_kernel(_item(), x.size)
    context.compile()
    kernel.create_kernel_call(["x"], ["y"], [x])
    kernel.run(1)
    context.wait()
    got = context.copy_to_host("y", np.empty(x.size))
    context.close()
    finite = np.isfinite(x) & (x > 0.0)
    want = np.array([_three_halves_correctly_rounded(float(v)) if ok else math.pow(v, 1.5) for v, ok in zip(x, finite)])
    mismatched = np.flatnonzero(got != want)
    assert mismatched.size == 0, [(float(x[i]).hex(), float(got[i]).hex(), float(want[i]).hex()) for i in mismatched[:5]]
    assert got[-2] == 0.0 and got[-1] == np.inf
    assert finite.sum() == x.size - 2
#  and the argument that started this test: the host libm is the side that is off by one
    hard = float.fromhex("0x1.346fd7824e0a9p+2")
    assert _three_halves_correctly_rounded(hard) == float.fromhex("0x1.528e20d580cebp+3")
