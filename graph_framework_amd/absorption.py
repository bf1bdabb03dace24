"""Host-side mirror of the power-absorption pass of xrays.

Reference: after `trace_ray` has written result<n>.nc, `calculate_power` (graph_driver/xrays.cpp:599-665)
constructs an `absorption::weak_damping<std::complex<double>, true>` per shard
(graph_framework/absorption.hpp:325-470), which adds the complex variable `kamp` to the trajectory
file and, per stored time index, reads the ray state, runs the one work item
`weak_damping_kimg_kernel` and writes kamp; `bin_power` (xrays.cpp:674-790) then integrates
Im(kamp) along the path with the item `power` and stores `power` and `d_power`.

Both items are the DAGs the reference front end builds (cold_plasma_expansion and
hot_plasma_expansion<z_erfi> on the EFIT equilibrium; complex base type with SAFE_MATH guards for the
first, double for the second), exported as GFIR; they run through the same C ABI as the hot path.
"""
import threading

import numpy as np

from .output import ResultFile
from .workflow import Manager
from .xrays import workload

#  input order of the two items: absorption.hpp:411-422, xrays.cpp:727-737
WEAK_DAMPING_INPUTS = ("kamp", "kx", "ky", "kz", "x", "y", "z", "t", "w")
POWER_INPUTS = ("x", "y", "z", "x_last", "y_last", "z_last", "kamp", "power", "k_sum")


class _Writer:
    """`sync.join(); work.wait(); sync = std::thread(dataset.write)` (absorption.hpp:462-467)."""

    def __init__(self):
        self.thread = None
        self.error = None

    def join(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        if self.error is not None:
            error, self.error = self.error, None
            raise error

    def start(self, work):
        def run():
            try:
                work()
            except Exception as error:          # surfaced by the next join()
                self.error = error
        self.thread = threading.Thread(target=run)
        self.thread.start()


class WeakDamping:
    """absorption::weak_damping<std::complex<double>, true> (absorption.hpp:325-470)."""

    def __init__(self, filename, index=0, stream=None, item=None):
        self.file = ResultFile(filename)                                    # result_file(filename): opened for update
        self.num_rays = self.file.num_rays
        self.work = Manager(index, stream)
        zeros = np.zeros(self.num_rays, dtype=np.complex128)
        self.host = {name: zeros.copy() for name in WEAK_DAMPING_INPUTS}
        self.item = self.work.add_item(item or workload("weak_damping_kimg_kernel", "c64"), WEAK_DAMPING_INPUTS, [],
                                       self.num_rays, self.host)
        self.sync = _Writer()

    def compile(self):
        """weak_damping::compile (absorption.hpp:438-454)."""
        self.work.compile()
        self.file.create_variable("kamp", parts=2)
        self.kamp = self.work.context.get_host_buffer("kamp")               # dataset.create_variable(..., work.get_context())

    def run(self, time_index):
        """weak_damping::run (absorption.hpp:456-468)."""
        for name, stored in (("w", "w"), ("kx", "kx"), ("ky", "ky"), ("kz", "kz"), ("x", "x"), ("y", "y"), ("z", "z"),
                             ("t", "time")):
#  the real trajectory variables land in the real parts (stride 2, output.hpp:305, :425-428)
            self.host[name].real[:] = self.file.read(stored, time_index)
            self.work.copy_to_device(name, self.host[name])
        self.work.run()
        self.sync.join()
        self.work.wait()
        record = {"kamp": self.kamp.copy()}
        self.sync.start(lambda: self.file.write(record, index=time_index))

    def close(self):
        self.sync.join()
        self.file.close()
        self.work.context.close()


class RootFinder(WeakDamping):
    """absorption::root_finder<std::complex<double>, true> (absorption.hpp:119-290): per stored record
    kamp <- 0 (`root_find_init_kernel`), Newton on the hot-plasma D(k + kamp k_hat) for the complex kamp
    (the converge item `loss_kernel` of solver::newton, newton.hpp:34-51: tolerance 1e-30, at most 1000
    iterations, the max over the shard is the element of largest modulus), kamp <- |k| + kamp
    (`final_kamp`)."""

    def __init__(self, filename, index=0, stream=None, items=None, tolerance=1.0e-30, max_iterations=1000):
        items = items or {}
        self.file = ResultFile(filename)
        self.num_rays = self.file.num_rays
        self.work = Manager(index, stream)
        zeros = np.zeros(self.num_rays, dtype=np.complex128)
        self.host = {name: zeros.copy() for name in WEAK_DAMPING_INPUTS}
        first = WEAK_DAMPING_INPUTS[:7]                                    # kamp kx ky kz x y z, absorption.hpp:170-178
        self.work.add_item(items.get("init") or workload("root_find_init_kernel", "c64"), first, [], self.num_rays, self.host)
        self.newton = self.work.add_converge_item(items.get("loss") or workload("root_find_loss_kernel", "c64"),
                                                  WEAK_DAMPING_INPUTS, ["root_find_residual"], self.num_rays, self.host,
                                                  tolerance, max_iterations)
        self.work.add_item(items.get("final") or workload("root_find_final_kamp", "c64"), first, [], self.num_rays, self.host)
        self.iterations = []
        self.sync = _Writer()

    def run(self, time_index):
        super().run(time_index)
        self.iterations.append(self.newton.iterations)


def run_absorption(filename, num_steps, index=0, model="weak_damping"):
    """run_absorption<ABSORPTION_MODEL> (xrays.cpp:551-585, model chosen as :634-651): records 0 .. num_steps."""
    power = (RootFinder if model == "root_find" else WeakDamping)(filename, index)
    power.compile()
    for j in range(num_steps + 1):
        power.run(j)
    power.close()
    return power


def bin_power(filename, num_steps, index=0, stream=None, item=None):
    """bin_power's per-shard body (xrays.cpp:694-786)."""
    file = ResultFile(filename)
    n = file.num_rays
    work = Manager(index, stream)
    host = {name: np.zeros(n) for name in POWER_INPUTS}
    host["power"][:] = 1.0                                                  # xrays.cpp:704-705
    item = work.add_item(item or workload("power", "f64"), POWER_INPUTS, ["d_power"], n, host)
    work.compile()
    file.create_variable("power")
    file.create_variable("d_power")
    power = work.context.get_host_buffer("power")
    d_power = work.context.get_host_buffer("d_power")

    for name in ("x", "y", "z"):                                            # dataset.read(file, 0) ... :767-771
        host[name][:] = file.read(name, 0)
        host[name + "_last"][:] = host[name]
    work.wait()                                                             # mirrors hold the initial values
    file.write({"power": power.copy(), "d_power": d_power.copy()}, index=0)
    sync = _Writer()
    for name in ("x_last", "y_last", "z_last"):
        work.copy_to_device(name, host[name])
    for j in range(1, num_steps + 1):
        for name in ("x", "y", "z"):
            host[name][:] = file.read(name, j)
            work.copy_to_device(name, host[name])
        host["kamp"][:] = file.read("kamp", j, part=1)                      # reference_imag_variable
        work.copy_to_device("kamp", host["kamp"])
        work.run()
        sync.join()
        work.wait()
        record = {"power": power.copy(), "d_power": d_power.copy()}
        sync.start(lambda record=record, j=j: file.write(record, index=j))
    sync.join()
    file.close()
    work.context.close()
    return item
