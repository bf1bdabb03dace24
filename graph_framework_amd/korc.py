"""Host-side mirror of the xkorc driver's workflow (graph_korc/xkorc.cpp:29-154).

Items (exported DAGs, graph_framework_amd/workloads/korc_*):
  axis_newton       efit::get_characteristic_field's two-unknown Newton to the magnetic axis
  bmod_at_axis      |B| there = b0                           (equilibrium.hpp:1585-1615)
  initialize_gamma  pre-item: gamma = 1/sqrt(1 - u.u), u <- gamma u      (xkorc.cpp:66-85)
  step              the relativistic push, dt = 0.5                       (xkorc.cpp:87-121)
b0 is a constant of the `step` graph (it is computed when the graph is built, xkorc.cpp:31),
so the exported `step` item carries the value the reference front end obtained.
"""
import numpy as np

from .workflow import Manager
from .xrays import workload

PARTICLE = ("x", "y", "z", "ux", "uy", "uz", "gamma")    # input order of `step`, xkorc.cpp:105-113
_NP = {"f64": np.float64, "f32": np.float32}


ITEMS = ("korc_axis_newton", "korc_bmod_at_axis", "korc_initialize_gamma", "korc_step")


def _item(items, name, dtype):
    """GFIR bytes received from rank 0 (`items`), else the exported workload file."""
    return items[name] if items and name in items else workload(name, dtype)


def characteristic_field(dtype="f64", index=0, stream=None, items=None):
    """efit::get_characteristic_field on the device: returns (b0, Newton iterations)."""
    np_dtype = _NP[dtype]
    work = Manager(index, stream)
    keys = ["axis_x", "axis_y", "axis_z"]
    initial = {"axis_x": np.array([1.7], np_dtype), "axis_y": np.array([0.0], np_dtype),
               "axis_z": np.array([0.0], np_dtype)}
    newton = work.add_converge_item(_item(items, "korc_axis_newton", dtype), keys, ["axis_residual"], 1, initial)
    bmod = work.add_item(_item(items, "korc_bmod_at_axis", dtype), keys, ["axis_bmod"], 1, initial)
    work.compile()
    newton.run()
    bmod.run()
    work.wait()
    b0 = work.check_value(0, "axis_bmod")
    iterations = newton.iterations
    work.context.close()
    return b0, iterations


class Korc:
    """workflow of ONE device thread of run_korc<T> (xkorc.cpp:23-150) on its shard of the
    particles: pre-item initialize_gamma, then `step` per run().  The reference splits the
    particles over its device threads as `batch + (extra > thread_number)` (xkorc.cpp:16-25);
    here one process per GPU holds one shard (xrays.shard_bounds is that split).

    items: optional {workload name: GFIR bytes} (received by broadcast from rank 0).
    device_state: keep the seven particle arrays in torch CUDA tensors (self.device) adopted by
    the context, so that the output-cadence all-gather reads them in place."""

    def __init__(self, particles, dtype="f64", index=0, stream=None, items=None, device_state=False):
        self.dtype = dtype
        self.np_dtype = _NP[dtype]
        sizes = [np.size(particles[k]) for k in PARTICLE if np.ndim(particles[k]) > 0]
        self.num_particles = max(sizes) if sizes else 1
        self.host = {k: np.ascontiguousarray(np.broadcast_to(np.asarray(particles[k], dtype=self.np_dtype),
                                                             (self.num_particles,)).copy()) for k in PARTICLE}
        self.device = None
        self.torch_stream = None
        if device_state:
            import torch
            where = torch.device("cuda", index)
            if stream is None:
                self.torch_stream = torch.cuda.Stream(device=where)      # see xrays.RaySolver
                stream = self.torch_stream.cuda_stream
            self.device = {k: torch.from_numpy(self.host[k]).to(where) for k in PARTICLE}
        self.work = Manager(index, stream)
        if self.device is not None:
            for k, tensor in self.device.items():
                self.work.context.set_buffer(k, tensor)
        self.init_item = self.work.add_preitem(_item(items, "korc_initialize_gamma", dtype),
                                               ["ux", "uy", "uz", "gamma"], [], self.num_particles, self.host)
        self.step_item = self.work.add_item(_item(items, "korc_step", dtype), list(PARTICLE), [],
                                            self.num_particles, self.host)

    def compile(self):
        self.work.compile()

    def pre_run(self):
        self.work.pre_run()

    def run(self, steps=1):
        self.work.run(steps)

    def wait(self):
        self.work.wait()

    def sync_host(self):
        for k in PARTICLE:
            self.work.copy_to_host(k, self.host[k])
        return self.host
