#!/usr/bin/env python3
"""Golden records of the absorption pass of xrays (graph_driver/xrays.cpp:599-790) from the
reference's own graph layer (oracle/_ref/gf_ref), and its two work items as GFIR:

    graph_framework_amd/workloads/weak_damping_kimg_kernel_c64.gfir   (absorption::weak_damping,
        absorption.hpp:346-432: complex<double>, SAFE_MATH; cold_plasma_expansion, hot_plasma_expansion
        with z_erfi, EFIT equilibrium)
    graph_framework_amd/workloads/power_f64.gfir                       (bin_power, xrays.cpp:674-790)
    graph_framework_amd/workloads/root_find_{init_kernel,loss_kernel,final_kamp}_c64.gfir
        (absorption::root_finder, absorption.hpp:146-290: Newton on the hot-plasma D for the complex kamp)
    tests/golden/absorption_golden.npz

Records: 8 rays of the CLI beam followed with rk4 x ordinary_wave (the documented example, xrays.cpp:
886); at every saved step kamp from the reference's HOST evaluation of the setter expression
(node evaluate(): backend::buffer arithmetic on std::complex, special::erfi), then the `power` item's
loop over those records with Im(kamp) on the reference tape.

    python tests/golden/make_absorption_golden.py        (development container, needs /root/reference)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402

STATE = ("t", "w", "x", "y", "z", "kx", "ky", "kz")


def main():
    tables = np.load(os.path.join(HERE, "efit_tables.npz"))
    R = ref.Reference(tables)
    from graph_framework_amd.xrays import cli_distribution     # the reference's samples (tests/test_cabi.py pins them)
    n = 8
    cli = cli_distribution(100000, seed=0)
    state = {k: np.ascontiguousarray(cli[k][:n], dtype=np.float64) for k in STATE}
    records, info = R.trace(state, 1.0e-3, 2000, save_every=100, newton_var=1, dispersion="ordinary_wave")
    saved = records.shape[0]
    flat = {k: records[:, i, :].reshape(-1) for i, k in enumerate(STATE)}
    workloads = os.path.join(ROOT, "graph_framework_amd", "workloads")
#  weak_damping inputs: kamp kx ky kz x y z t w
    columns = [np.zeros(saved*n)] + [flat[k] for k in ("kx", "ky", "kz", "x", "y", "z", "t", "w")]
    out, _ = R._run("c64", "weak_damping", columns, os.path.join(workloads, "weak_damping_kimg_kernel_c64.gfir"))
    kamp = (out[0] + 1j*out[1]).reshape(saved, n)
#  power inputs: x y z x_last y_last z_last kamp power k_sum, then per record (x y z Im kamp)
    first = [records[0, 2], records[0, 3], records[0, 4]]
    columns = first + first + [np.zeros(n), np.ones(n), np.zeros(n)]
    for r in range(1, saved):
        columns += [records[r, 2], records[r, 3], records[r, 4], kamp[r].imag]
    out, _ = R._run("f64", "power", columns, saved - 1, os.path.join(workloads, "power_f64.gfir"))
    power = out.reshape(saved - 1, 3, n)
#  root_finder: every record is one batch (the converge item's max runs over the 8 rays of the record)
    root, iterations = [], []
    for r in range(saved):
        columns = [np.zeros(n)] + [records[r, STATE.index(k)] for k in ("kx", "ky", "kz", "x", "y", "z", "t", "w")]
        out, info = R._run("c64", "root_finder", columns, workloads if r == 0 else "-")
        root.append(out[0] + 1j*out[1])
        iterations.append(int(out[2][0]))
    root = np.stack(root)
    np.savez_compressed(os.path.join(HERE, "absorption_golden.npz"), records=records, kamp=kamp, power=power,
                        root_kamp=root, root_iterations=np.array(iterations))
    print("root_finder iterations per record", iterations)
    print("root_finder kamp[6]", root[6, :3])
    print("records", records.shape, "kamp[1]", kamp[1, :3], "power[-1]", power[-1, 0, :3])
    print("finite kamp:", np.isfinite(kamp).all(), " max |Im kamp|", np.abs(kamp.imag).max())


if __name__ == "__main__":
    main()
