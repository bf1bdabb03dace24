"""The VMEC equilibrium on the device: the field item of equilibrium::vmec
(equilibrium.hpp:1868-2330) on the reference-held graph_tests/vmec.nc — 86 Fourier modes, 1036 spline
tables gathered per point, 172 sin/cos — through the C ABI against the oracle and the reference graph
layer's records (tests/golden/vmec_golden.npz).

Tolerance, stated: sin and cos are the device libm's here and glibc's on the reference side (each within
1 ulp of the exact value, not of each other), and R, Z, lambda are sums of 86 terms of either sign, so
the comparison is relative to the size of the vector a component belongs to: 1e-12 |B| for the field,
1e-13 of the major radius for the position; density and temperature involve pow(x, 1.5) only: 1e-14.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, WORKLOADS

pytestmark = pytest.mark.gpu


def test_vmec_field_item_matches_the_oracle_and_the_reference_records():
    from graph_framework_amd import Context
    golden = np.load(os.path.join(GOLDEN, "vmec_golden.npz"))
    inputs, want = golden["inputs"], golden["outputs"]
    n = inputs.shape[1]
    context = Context(0)
    kernel = context.add_kernel(os.path.join(WORKLOADS, "vmec_field_kernel_f64.gfir"), n)
    context.compile()
    out_keys = ["bx", "by", "bz", "x", "y", "z", "ne", "te"]
    kernel.create_kernel_call(["s", "u", "v"], out_keys, [c.copy() for c in inputs])
    kernel.run(1)
    context.wait()
    got = np.stack([context.copy_to_host(k, np.empty(n)) for k in out_keys])
    assert context.flags() & 1 == 0
    context.close()
    field = np.sqrt((want[:3]**2).sum(axis=0))
    assert (np.abs(got[:3] - want[:3]) <= 1.0e-12*field).all(), np.abs(got[:3] - want[:3]).max()/field.min()
    radius = np.hypot(want[3], want[4])
    assert (np.abs(got[3:6] - want[3:6]) <= 1.0e-13*radius).all()
    np.testing.assert_allclose(got[6:], want[6:], rtol=1.0e-14, atol=0.0)
