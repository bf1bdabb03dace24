"""Host-side mirror of solver::solver_interface / solver::rk4 for the xrays hot path.

Mirrors graph_framework/solver.hpp:123-430 (init, compile, step, sync_host,
sync_device, check_residual) for the (cold_plasma x EFIT x rk4) combination of
graph_benchmark/xrays_bench.cpp:53-102.  The two work items — `loss_kernel`
(Newton, newton.hpp:34-51) and `solver_kernel` (solver.hpp:303-349) — are the
DAGs the reference front end builds for that combination, exported as GFIR
(graph_framework_amd/workloads/); the time step dt is a constant of the graph
(xrays_bench.cpp:75-77), so a workload file is specific to its dt.
"""
import os

import numpy as np

from . import _lib
from .workflow import Manager

STATE = ("t", "w", "x", "y", "z", "kx", "ky", "kz")      # input order, solver.hpp:304-313
_NP = {"f64": np.float64, "f32": np.float32}


def workload(name, dtype="f64"):
    path = os.path.join(_lib.WORKLOAD_DIR, "%s_%s.gfir" % (name, dtype))
    if not os.path.exists(path):
        raise FileNotFoundError("no exported workload %s" % path)
    return path


def shard_bounds(total, shards, index):
    """Contiguous split of the reference drivers: batch = N/T, the first N%T shards get
    one more (graph_benchmark/xrays_bench.cpp:38-51, graph_driver/xrays.cpp:423-432)."""
    batch = total//shards
    extra = total % shards
    begin = index*batch + min(index, extra)
    return begin, begin + batch + (1 if extra > index else 0)


def cli_distribution(num_rays, seed=0, dtype="f64"):
    """The incoherent initial conditions of the xrays CLI example (graph_driver/xrays.cpp:392-399,
    :448-453; graph_driver/CMakeLists.txt:5-31): cylindrical r = 2.5, phi ~ N(0, 0.05),
    z ~ N(0, 0.05), ky ~ N(-100, 10), kz ~ N(0, 10), omega ~ N(700, 10), kx guess -700 (then
    Newton), seed = shard index — the reference's own samples: std::mt19937_64 and libstdc++'s
    std::normal_distribution restated in csrc/cli_distribution.cpp (pinned to samples of the real
    objects, tests/golden/cli_distribution_golden.npz)."""
    import ctypes
    lib = _lib.load()
    means = (ctypes.c_double*7)(700.0, -700.0, -100.0, 0.0, 0.0, 2.5, 0.0)           # omega, kx, ky, kz, z, radius, phi
    sigmas = (ctypes.c_double*7)(10.0, 0.0, 10.0, 10.0, 0.05, 0.0, 0.05)
    columns = [np.empty(num_rays, dtype=np.float64) for _ in STATE]
    pointers = (ctypes.c_void_p*8)(*[c.ctypes.data for c in columns])
    lib.gfhip_cli_distribution(int(seed), int(num_rays), means, sigmas, pointers)
    np_dtype = _NP[dtype]
    return {k: np.ascontiguousarray(c, dtype=np_dtype) for k, c in zip(STATE, columns)}


class RaySolver:
    """solver::solver_interface (solver.hpp:123-430) over exported work items: any
    (integrator x dispersion relation x equilibrium) whose `loss_kernel_<unknown>` and
    `solver_kernel` items exist as `<workload_prefix><item>_<dtype>.gfir`."""

    def __init__(self, state, dtype="f64", index=0, stream=None, prefix="", items=None, device_state=False,
                 dispersion="cold_plasma", workload_prefix=None):
        """state: dict of host arrays (or scalars) t,w,x,y,z,kx,ky,kz for this shard.
        items: optional {workload name: GFIR bytes} (e.g. received by broadcast from rank 0);
        by default the exported workload files are read.
        device_state: keep the state and the residual in torch CUDA tensors (self.device) adopted
        by the context, so that torch (streams, collectives, output.TrajectoryWriter) can use them."""
        self.dtype = dtype
        self.items = items or {}
#  Exported (dispersion x rk4 x EFIT) combinations: cold_plasma (dt = 1e-3, xrays_bench) and
#  ordinary_wave (dt = 1e-4, graph_tests/physics_test.cpp:583-618).
        if workload_prefix is not None:
            self.workload_prefix = workload_prefix
        else:
            self.workload_prefix = "" if dispersion == "cold_plasma" else dispersion + "_"
        self.np_dtype = _NP[dtype]
        sizes = [np.size(state[k]) for k in STATE if np.ndim(state[k]) > 0]
        self.num_rays = max(sizes) if sizes else 1
        self.host = {k: np.ascontiguousarray(np.broadcast_to(np.asarray(state[k], dtype=self.np_dtype),
                                                             (self.num_rays,)).copy()) for k in STATE}
        self.prefix = prefix
        self.keys = [prefix + k for k in STATE]
        self.residual_key = prefix + "residual"
        self.device = None
        self.torch_stream = None
        if device_state:
            import torch
            where = torch.device("cuda", index)
            torch_dtype = torch.float64 if dtype == "f64" else torch.float32
            if stream is None:
#  A dedicated torch stream (the default stream's handle is 0, which the C ABI reads as
#  "create a private stream"): kernels and torch work issued under it are ordered.
                self.torch_stream = torch.cuda.Stream(device=where)
                stream = self.torch_stream.cuda_stream
            self.device = {k: torch.from_numpy(self.host[k]).to(where) for k in STATE}
            self.device["residual"] = torch.zeros(self.num_rays, dtype=torch_dtype, device=where)
        self.work = Manager(index, stream)
        if self.device is not None:
            for k, tensor in self.device.items():
                self.work.context.set_buffer(prefix + k, tensor)
        self.newton = None
        self.solver = None
        self.newton_iterations = None
        self.newton_last_max = None

    def _item(self, name):
        return self.items[name] if name in self.items else workload(self.workload_prefix + name, self.dtype)

    def _initial(self):
        return {self.prefix + k: self.host[k] for k in STATE}

    def init(self, variable="kx", tolerance=1.0e-30, max_iterations=1000, per_ray=False):
        """solver_interface::init(x) (solver.hpp:254-274) -> dispersion_interface::solve
        (dispersion.hpp:1452-1475): its own manager in the reference, its own converge item
        here; the solved variable is copied back to the host array.
        per_ray=True runs the stall loop per ray inside ONE launch (wavefront ballot exit)
        instead of the reference's kernel + global max + host test per iteration."""
        work = self.work
        item = work.add_converge_item(self._item("loss_kernel_" + variable), self.keys,
                                      [self.prefix + "newton_residual"], self.num_rays, self._initial(),
                                      tolerance, max_iterations)
        work.context.compile()
        item.create_kernel_call()
        if per_ray:
            item.iterations, item.last_max = item.kernel.converge_per_ray(tolerance, max_iterations)
        else:
            item.run()
        self.newton = item
        self.newton_iterations, self.newton_last_max = item.iterations, item.last_max
        work.copy_to_host(self.prefix + variable, self.host[variable])
        return self.newton_last_max

    def compile(self):
        """solver_interface::compile (solver.hpp:303-349): the `solver_kernel` item."""
        work = self.work
        self.solver = work.add_item(self._item("solver_kernel"), self.keys, [self.residual_key],
                                    self.num_rays, self._initial())
        work.context.compile()
        self.solver.create_kernel_call()

    def step(self, steps=1):
        """solver_interface::step (solver.hpp:382): one launch; `steps` > 1 fuses that many
        RK4 steps into the launch."""
        self.solver.run(steps)

    def sync_host(self):
        """solver_interface::sync_host (solver.hpp:368-377): D2H of the 8 state arrays."""
        for k in STATE:
            self.work.copy_to_host(self.prefix + k, self.host[k])
        return self.host

    def sync_device(self):
        for k in STATE:
            self.work.copy_to_device(self.prefix + k, self.host[k])

    def check_residual(self, index):
        return self.work.check_value(index, self.residual_key)

    def residual(self):
        out = np.empty(self.num_rays, dtype=self.np_dtype)
        return self.work.copy_to_host(self.residual_key, out)


class Rk4ColdPlasmaEfit(RaySolver):
    """solver::rk4<dispersion::cold_plasma<T>> on an EFIT equilibrium (the xrays_bench
    combination; `dispersion="ordinary_wave"` selects physics_test.cpp:583-618's)."""


class AdaptiveRk4ColdPlasmaEfit(RaySolver):
    """solver::adaptive_rk4<dispersion::cold_plasma<T>> on the EFIT equilibrium (solver.hpp:877-1006,
    `--solver=adaptive_rk4` of graph_driver/xrays.cpp:353): dt is a per-ray variable, and every
    step first runs a converge item on the two unknowns (dt, lambda) of 1/dt + lambda*D(next)^2
    (newton.hpp:34-51), then the RK4 step with that dt — the two items of adaptive_rk4::compile
    (solver.hpp:951-1003) in the order workflow::manager::run executes them.

    On the reference's own graph layer that converge item drives dt and lambda to NaN inside the
    first step (tests/golden/make_adaptive_golden.py records it); this class reproduces the
    reference, it does not repair it."""

    def __init__(self, state, dt=1.0e-3, **kwargs):
        super().__init__(state, **kwargs)
        self.host["dt"] = np.full(self.num_rays, dt, dtype=self.np_dtype)
        self.host["lambda"] = np.full(self.num_rays, 1.0, dtype=self.np_dtype)       # solver.hpp:953
        self.adaptive = None
        self.adaptive_iterations = []

    def compile(self, tolerance=1.0e-30, max_iterations=1000):
        work = self.work
        initial = dict(self._initial(), **{self.prefix + "dt": self.host["dt"], self.prefix + "lambda": self.host["lambda"]})
        keys = self.keys + [self.prefix + "dt", self.prefix + "lambda"]
        self.adaptive = work.add_converge_item(self._item("adaptive_rk4_loss_kernel"), keys, [self.prefix + "adaptive_loss"],
                                               self.num_rays, initial, tolerance, max_iterations)
        self.solver = work.add_item(self._item("adaptive_rk4_solver_kernel"), keys[:-1], [self.residual_key],
                                    self.num_rays, initial)
        work.context.compile()
        self.adaptive.create_kernel_call()
        self.solver.create_kernel_call()

    def step(self, steps=1):
        for _ in range(steps):
            self.adaptive.run()
            self.adaptive_iterations.append(self.adaptive.iterations)
            self.solver.run(1)

    def sync_host(self):
        super().sync_host()
        for k in ("dt", "lambda"):
            self.work.copy_to_host(self.prefix + k, self.host[k])
        return self.host
