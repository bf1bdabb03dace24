#!/usr/bin/env python3
"""Golden records of solver::adaptive_rk4 (solver.hpp:877-1006) from the reference's own graph
layer (oracle/_ref/gf_ref, `trace_adaptive`), and the two work items of its compile() as GFIR:

    graph_framework_amd/workloads/adaptive_rk4_{loss,solver}_kernel_f64.gfir
    tests/golden/adaptive_rk4_golden.npz

What the records show: on the reference graph the converge item on (dt, lambda) of
loss = 1/dt + lambda*D(next)^2 drives both unknowns to NaN inside the FIRST step (d(loss)/d(lambda) =
D^2 ~ 1e-30 after the Newton init), so a ray traced with `--solver=adaptive_rk4` is NaN from step 1
on.  The backend reproduces exactly that: the per-pass values of the first loop, its iteration
count, and the NaN state afterwards.

    python tests/golden/make_adaptive_golden.py        (development container, needs /root/reference)
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
    n = 3
    state = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0)
    columns = [np.full(n, state[k]) for k in STATE] + [np.full(n, 1.0e-3), np.full(n, 1.0)]     # ..., dt, lambda
    columns[1][1], columns[2][1], columns[5][1] = 700.0, 2.0, -700.0                             # two rays inside the plasma
    columns[1][2], columns[2][2], columns[4][2], columns[5][2] = 800.0, 1.8, 0.1, -500.0
    workloads = os.path.join(ROOT, "graph_framework_amd", "workloads")
    steps = 3
    out, info = R._run("f64", "trace_adaptive", columns, steps, workloads, "f64")
    records = out.reshape(steps + 1, 12, n)
    passes = ref._read_columns(os.path.join(R.tmp.name, "out.bin.passes")).reshape(24, 3, n)
    np.savez_compressed(os.path.join(HERE, "adaptive_rk4_golden.npz"), inputs=np.stack(columns), records=records,
                        passes=passes, newton_iterations=np.array(info["newton_iterations"]))
    print("newton iterations", info["newton_iterations"], "first converge loop:", records[1, 11, 0], "iterations")
    print("dt after each of the first passes:", passes[:6, 0, 0])


if __name__ == "__main__":
    main()
