import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WORKLOADS = os.path.join(ROOT, "graph_framework_amd", "workloads")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def efit_tables():
    return np.load(os.path.join(GOLDEN, "efit_tables.npz"))


@pytest.fixture(scope="session")
def efit_gold():
    return np.load(os.path.join(GOLDEN, "efit_gold.npz"))


def bench_state(n, dtype=np.float64, **override):
    """The identical-ray initial conditions of graph_benchmark/xrays_bench.cpp:62-71."""
    values = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0)
    values.update(override)
    return {k: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=dtype), (n,)).copy())
            for k, v in values.items()}


def random_plasma_state(n, seed, dtype=np.float64):
    """Seeded rays INSIDE the plasma (non-degenerate dispersion surface)."""
    rng = np.random.default_rng(seed)
    phi = rng.uniform(-0.3, 0.3, n)
    r = rng.uniform(1.3, 2.2, n)
    state = dict(t=np.zeros(n), w=rng.uniform(450.0, 900.0, n),
                 x=r*np.cos(phi), y=r*np.sin(phi), z=rng.uniform(-0.5, 0.5, n),
                 kx=rng.uniform(-700.0, -200.0, n), ky=rng.uniform(-100.0, 100.0, n),
                 kz=rng.uniform(-50.0, 50.0, n))
    return {k: np.ascontiguousarray(v, dtype=dtype) for k, v in state.items()}


STATE = ("t", "w", "x", "y", "z", "kx", "ky", "kz")


@pytest.fixture(scope="session")
def golden_ref():
    """tests/golden/ref_golden.npz: records written by the reference's own graph layer."""
    return np.load(os.path.join(GOLDEN, "ref_golden.npz"))
