#!/usr/bin/env python3
"""Generate golden vectors with the reference-backed oracle oracle/_ref/gf_ref.

gf_ref is built from the reference's own expression-graph headers
(make -C oracle ref; needs /root/reference, i.e. runs in the development container only) and
evaluates the reference's DAGs in strict IEEE arithmetic.  The outputs committed here are
DATA (inputs + expected outputs):

    tests/golden/ref_golden.npz
        bench_*      xrays_bench ray (graph_benchmark/xrays_bench.cpp:62-71): state after the
                     Newton init and after 1, 10, 100, 500, 1000 RK4 steps, residual
        rays_*       64 seeded rays inside the plasma: inputs, state after 1 and 20 steps
        disp_*       256 seeded points: D and its 7 partials (dispersion.hpp:1390-1409)
        efit_*       graph_tests/efit_test.cpp outputs on the gold grid
        korc_*       xkorc push (graph_korc/xkorc.cpp): b0, larmor radius, particle after
                     initialize_gamma and 1, 10, 100, 1000 steps (fp64) / 100 steps (fp32)

It also re-exports the workload GFIR files under graph_framework_amd/workloads/.

Run:  python tests/golden/make_ref_golden.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref  # noqa: E402
from conftest import bench_state, random_plasma_state, STATE  # noqa: E402


def main():
    tables = np.load(os.path.join(HERE, "efit_tables.npz"))
    gold = np.load(os.path.join(HERE, "efit_gold.npz"))
    R = ref.Reference(tables)
    out = {}

#  Workload export (the DAGs the backend consumes).
    workloads = os.path.join(ROOT, "graph_framework_amd", "workloads")
    os.makedirs(workloads, exist_ok=True)
    subprocess.check_call([ref.BINARY, R.tables, "f64", "export", workloads, "0.001", "f64"])
    subprocess.check_call([ref.BINARY, R.tables, "f32", "export", workloads, "0.001", "f32"])
#  fp32 is only exercised on the Newton + RK4 items (and xkorc below); keep the tree small.
    for name in ("loss_kernel_w", "loss_kernel_ky", "loss_kernel_kz", "dispersion_kernel", "efit_test_kernel"):
        os.remove(os.path.join(workloads, name + "_f32.gfir"))

    for dtype in ("f64", "f32"):
        subprocess.check_call([ref.BINARY, R.tables, dtype, "export_misc", workloads, dtype])

#  xrays_bench ray.
    records, info = R.trace(bench_state(1), 1.0e-3, 1000, 1, 1)
    keep = [0, 1, 10, 100, 500, 1000]
    out["bench_steps"] = np.array(keep)
    out["bench_records"] = records[keep, :, 0]            # [saved, 9]
    out["bench_newton_iterations"] = np.array(info["newton_iterations"])
    out["bench_newton_last_max"] = np.array(info["newton_last_max"])

#  physics_test.cpp:583-618: ordinary_wave on EFIT, omega = 590, dt = 1e-4, 10000 RK4 steps.
    subprocess.check_call([ref.BINARY, R.tables, "f64", "export_ordinary", workloads, "0.0001", "f64"])
    records, info = R.trace(bench_state(1, w=590.0), 1.0e-4, 10000, 1000, 1, dispersion="ordinary_wave")
    out["ordinary_steps"] = np.arange(0, 10001, 1000)
    out["ordinary_records"] = records[:, :, 0]
    out["ordinary_newton_iterations"] = np.array(info["newton_iterations"])

#  Seeded rays inside the plasma.
    state = random_plasma_state(64, seed=2024)
    records, _ = R.trace(state, 1.0e-3, 20, 1, -1)
    out["rays_inputs"] = np.stack([state[k] for k in STATE])
    out["rays_step1"] = records[1]
    out["rays_step20"] = records[20]

#  Dispersion function and partials.
    state = random_plasma_state(256, seed=7)
    values, _ = R.dispersion(state)
    out["disp_inputs"] = np.stack([state[k] for k in STATE])
    out["disp_outputs"] = values

#  efit_test on the gold grid.
    Rg, Zg = np.meshgrid(gold["r_grid"], gold["z_grid"], indexing="ij")
    x, z = Rg.ravel(), Zg.ravel()
    values, _ = R.efit_test(x, np.zeros_like(x), z)
    out["efit_outputs"] = values

#  xkorc.
    for dtype, steps, keep in (("f64", 1000, [0, 1, 10, 100, 1000]), ("f32", 100, [0, 1, 10, 100])):
        particles = {k: np.full(1, v) for k, v in
                     dict(x=1.7, y=0.0, z=0.0, ux=0.0, uy=0.99, uz=0.1, gamma=0.0).items()}
        records, info = R.korc(particles, steps, 1, dtype)
        out["korc_%s_steps" % dtype] = np.array(keep)
        out["korc_%s_records" % dtype] = records[keep, :, 0]
        out["korc_%s_b0" % dtype] = np.array(info["b0"])
        out["korc_%s_larmor_radius" % dtype] = np.array(info["larmor_radius"])
        out["korc_%s_axis_iterations" % dtype] = np.array(info["axis_iterations"])
        subprocess.check_call([ref.BINARY, R.tables, dtype, "export_korc", workloads, dtype])

    np.savez_compressed(os.path.join(HERE, "ref_golden.npz"), **out)
    print("wrote ref_golden.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
