#!/usr/bin/env python3
"""Trace a beam through the EFIT equilibrium and write the trajectories — the flow of
graph_driver/xrays.cpp:419-461 (per device: initial distribution, Newton solve for kx, RK4 steps
with a record every `sub_steps`, result<rank>.nc) on the MI355X backend.

    python examples/trace_rays.py --rays 100000 --steps 1000 --sub-steps 100 [--output /tmp/rays]
    python examples/trace_rays.py --rays 10000 --dispersion ordinary_wave --steps 20000 --sub-steps 1000 --output /tmp/rays --absorption-model weak_damping
    python examples/trace_rays.py --equilibrium vmec --rays 20000 --steps 10 --sub-steps 2 --output /tmp/vmec_rays
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/trace_rays.py ...

One process per GPU; the ensemble is split as the reference splits it over device threads; every rank
writes its own file (as the reference does).  The exported work items fix dt = 1e-3.
With --absorption-model the two stages that follow the trace in graph_driver/xrays.cpp:1100-1105 run on
the same file: calculate_power (kamp per stored record) and bin_power (power, d_power).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def vmec_beam(count, seed, device):
    """A beam in the flux coordinates of the VMEC equilibrium: omega = 400, starting surfaces s in [0.3, 0.8], random
    poloidal and toroidal angles, k pointing inwards (k_s = -40: |k| ~ 0.9 omega on these surfaces, then the Newton solve),
    small random k_u, k_v — the ensemble of tests/golden/make_vmec_trace_golden.py at any size."""
    import numpy as np
    from graph_framework_amd.xrays import RaySolver
    rng = np.random.default_rng(seed)
    state = dict(t=np.zeros(count), w=np.full(count, 400.0), x=rng.uniform(0.3, 0.8, count), y=rng.uniform(0.0, 2.0*np.pi, count),
                 z=rng.uniform(0.0, 2.0*np.pi, count), kx=np.full(count, -40.0), ky=rng.uniform(-2.0, 2.0, count),
                 kz=rng.uniform(-20.0, 20.0, count))
    return RaySolver(state, index=device, device_state=True, workload_prefix="vmec86_")


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--rays", type=int, default=100000, help="ensemble size over all ranks")
    parser.add_argument("--steps", type=int, default=1000)
    parser.add_argument("--sub-steps", type=int, default=100, help="steps between trajectory records")
    parser.add_argument("--output", default=None, help="prefix of the result files (default: no output)")
    parser.add_argument("--dispersion", choices=["cold_plasma", "ordinary_wave"], default="cold_plasma",
                        help="exported combinations: cold_plasma (dt = 1e-3, xrays_bench) and ordinary_wave (dt = 1e-4, the CLI example's)")
    parser.add_argument("--equilibrium", choices=["efit", "vmec"], default="efit",
                        help="efit: graph_tests/efit.nc, Cartesian rays (the benchmark's); vmec: graph_tests/vmec.nc, all 86 Fourier modes, "
                             "rays in flux coordinates (x, y, z = s, u, v; graph_driver/xrays.cpp:382), cold_plasma, dt = 1e-3 — the RK4 step "
                             "is a 54 k-record item that runs as 10 segment kernels")
    parser.add_argument("--absorption-model", choices=["weak_damping", "root_find"], default=None,
                        help="after the trace: kamp and power into the result file (needs --output)")
    args = parser.parse_args()

    import torch
    from graph_framework_amd import distributed
    from graph_framework_amd.output import TrajectoryWriter
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, cli_distribution, shard_bounds

    rank, world, local_rank = distributed.init()
    torch.cuda.set_device(local_rank)
    begin, end = shard_bounds(args.rays, world, rank)
    if args.equilibrium == "vmec":
        solve = vmec_beam(end - begin, rank, local_rank)
    else:
        solve = Rk4ColdPlasmaEfit(cli_distribution(end - begin, seed=rank), index=local_rank, device_state=True,
                                  dispersion=args.dispersion)
    residual = solve.init("kx")
    solve.compile()
    writer = TrajectoryWriter(solve, "%s%d.nc" % (args.output, rank)) if args.output else None
    if writer:
        writer.write_step()

    start = time.perf_counter()
    for step in range(args.steps):
        solve.step()
        if writer and (step + 1) % args.sub_steps == 0:
            writer.write_step()
    host = solve.sync_host()
    elapsed = time.perf_counter() - start
    if writer:
        writer.close()
    import numpy as np
    lost = int((~np.isfinite(host["x"])).sum())          # rays the reference graph itself drives to NaN
    print("rank %d: %d rays, Newton %d iterations (max residual %.3e), %d steps in %.3f s = %.3e ray-steps/s; "
          "x in [%.4f, %.4f], %d rays non-finite, status flags %d"
          % (rank, end - begin, solve.newton_iterations, residual, args.steps, elapsed,
             (end - begin)*args.steps/elapsed, np.nanmin(host["x"]), np.nanmax(host["x"]), lost,
             solve.work.context.flags()))
    if writer and args.absorption_model:
        from graph_framework_amd.absorption import bin_power, run_absorption
        from graph_framework_amd.output import ResultFile
        solve.work.context.close()
        path = "%s%d.nc" % (args.output, rank)
        records = args.steps//args.sub_steps
        start = time.perf_counter()
        run_absorption(path, records, index=local_rank, model=args.absorption_model)
        bin_power(path, records, index=local_rank)
        elapsed = time.perf_counter() - start
        result = ResultFile(path)
        power = result.read("power", records)
        result.close()
        kept = power[np.isfinite(power) & (power <= 1.0)]
        print("rank %d: absorption (%s) + power over %d records in %.3f s; transmitted power %.4f (mean over %d of %d rays "
              "with a power in [0, 1])" % (rank, args.absorption_model, records + 1, elapsed,
                                          float(kept.mean()) if kept.size else float("nan"), kept.size, power.size))


if __name__ == "__main__":
    main()
