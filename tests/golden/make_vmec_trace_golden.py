#!/usr/bin/env python3
"""Work items and golden trajectory of ray tracing on the VMEC equilibrium — the (cold_plasma x rk4 x vmec)
combination `graph_driver/xrays.cpp:382` accepts — from the reference's own graph layer
(oracle/_ref/gf_ref_vmec trace), on ALL 86 Fourier modes of the reference-held graph_tests/vmec.nc:

    graph_framework_amd/workloads/vmec86_loss_kernel_kx_f64.gfir    Newton item for k_s (the first wave-vector
                                                                    component; newton.hpp:34-51): 4.7 k records
    graph_framework_amd/workloads/vmec86_solver_kernel_f64.gfir     one RK4 step (solver.hpp:303-349, :777-870):
                                                                    54 k records, 18 k gathers
    tests/golden/vmec_trace_golden.npz                              the rays below after the Newton solve and after
                                                                    steps 1, 2, 5, 10, 20 on the tape

Coordinates are the flux coordinates (x, y, z) = (s, u, v) and k = k_s e^s + k_u e^u + k_v e^v.  Rays: omega = 400
(the plasma frequency of this equilibrium's profile peaks near 190, so n^2 = 1 - wpe^2/w^2 ~ 0.8-0.95 along the
rays), s in [0.3, 0.8], random poloidal and toroidal angles, k_s guessed from |e^s| (tests/vmec_numpy.py) so
that |k| ~ 0.9 omega, pointing inwards, small random k_u, k_v; dt = 1e-3.

Building the two items on the reference graph layer takes about seven minutes (symbolic df of 86 modes).

    python tests/golden/make_vmec_trace_golden.py        (development container, needs /root/reference)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from make_vmec_golden import BINARY, write_vmec  # noqa: E402
from oracle import ref  # noqa: E402
from vmec_numpy import vmec_field  # noqa: E402

MODES = 86
STEPS = (1, 2, 5, 10, 20)


def initial_rays(n=16, seed=11):
    tables = np.load(os.path.join(HERE, "vmec_tables.npz"))
    rng = np.random.default_rng(seed)
    s = np.linspace(0.3, 0.8, n)
    u, v = rng.uniform(0.0, 2.0*np.pi, n), rng.uniform(0.0, 2.0*np.pi, n)
    w = np.full(n, 400.0)
    field = vmec_field(tables, s, u, v, MODES)
    ks = -0.9*w/np.linalg.norm(field["esups"], axis=0)
    ku, kv = rng.uniform(-2.0, 2.0, n), rng.uniform(-20.0, 20.0, n)
    return [np.zeros(n), w, s, u, v, ks, ku, kv]


def main():
    workloads = os.path.join(ROOT, "graph_framework_amd", "workloads")
    columns = initial_rays()
    with tempfile.TemporaryDirectory() as tmp:
        tables = os.path.join(tmp, "vmec.bin")
        write_vmec(tables)
        ref._write_columns(os.path.join(tmp, "in.bin"), columns)
        records = {}
#  one run per saved step count would rebuild the graph each time: save every step and pick
        proc = subprocess.run([BINARY, tables, "trace", str(MODES), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin"),
                               "1e-3", str(STEPS[-1]), "1", "1", workloads], check=True, stderr=subprocess.PIPE, text=True)
        print(proc.stderr.strip())
        out = ref._read_columns(os.path.join(tmp, "out.bin")).reshape(STEPS[-1] + 1, 9, -1)
        newton = [line for line in proc.stderr.splitlines() if "newton_iterations" in line][0]
    import json
    info = json.loads(newton)
    np.savez_compressed(os.path.join(HERE, "vmec_trace_golden.npz"), initial=np.stack(columns), steps=np.array((0,) + STEPS),
                        records=out[[0] + list(STEPS)], newton_iterations=info["newton_iterations"],
                        newton_last_max=info["newton_last_max"])
    print("after Newton: ks", out[0, 5, :3], "residual", out[0, 8, :3])
    print("after %d steps: s" % STEPS[-1], out[-1, 2, :3], "residual", out[-1, 8, :3])


if __name__ == "__main__":
    main()
