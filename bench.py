#!/usr/bin/env python3
"""Benchmark of the hot path: ray-steps/s of the xrays_bench cold-plasma RK4 `solver_kernel`.

Workload (BASELINE.json configs[1]): graph_benchmark/xrays_bench.cpp:53-102 — every ray
omega=500, x=2.5, kx=-600 -> Newton, dt=1e-3, cold_plasma on the EFIT equilibrium of
graph_tests/efit.nc, fp64 — with 1e6 rays PER GPU (weak scaling).  A "step" is one launch of
`solver_kernel` over the rank's rays (solver_interface::step, solver.hpp:382).  Setup, the
Newton init and kernel builds are outside the timed region, as in the reference
(xrays_bench.cpp:88-102); inputs are resident in HBM when the timed region starts.  The
reference's timed region also contains the final sync_host (8 D2H copies); that PCIe-inclusive
rate is reported separately as "value_with_sync_host" and is never `value`.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

BYTES_PER_RAY_STEP_F64 = 128      # read 8 + write 8 (7 setters + residual) x 8 B, SURVEY.md §8(d)
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def available_cores():
    """Host threads this process may actually use: the affinity mask, capped by the cgroup CPU
    quota when there is one (a container can see 256 CPUs and own 16)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota)/float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return cores


def dag_flops(path):
    """Floating-point operations of one pass of a work item, counted from its GFIR records:
    add/sub/mul/div/sqrt/pow 1, fma 2, integer power p -> p-1 multiplies (SURVEY.md §8(d) counts
    the same way: ~4.3 kflop per RK4 ray-step)."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    _magic, _dtype, ni, _no, _ns, nt, nins, nb, _r = struct.unpack_from("<8s8I", data, 0)
    pos = 40 + nb
    for _ in range(ni):
        (n,) = struct.unpack_from("<I", data, pos)
        pos += 4 + n
    for _ in range(nt):
        rows, cols = struct.unpack_from("<II", data, pos)
        pos += 8 + 8*rows*cols
    flops = 0
    for i in range(nins):
        op, _a, _b, _c, aux = struct.unpack_from("<5I", data, pos + 56*i)
        if op in (2, 3, 4, 5, 7, 9):
            flops += 1
        elif op == 6:
            flops += 2
        elif op == 8:
            flops += max(aux, 1) - 1
    return flops


def measured_traffic(kernel_name, rays):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE run on this same command), if they are for this launch size."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(kernel_name)
    except (OSError, ValueError):
        return None
    if entry and entry.get("rays_per_launch") == rays:
        return entry["traffic_bytes_per_launch"]
    return None


def cpu_baseline(target_seconds=12.0):
    """The CPU oracle timed on a bounded sample of the same workload: the solver_kernel DAG as
    the reference's cpu_context runs it (one compiled statement per node, serial loop per
    thread, contiguous shards; oracle/gfir_to_c.py, gcc -O2, strict IEEE), one thread per
    available core.  Falls back to the interpreter (oracle/gfir_interp.c) without gcc."""
    from oracle import gfir
    from graph_framework_amd.xrays import STATE, workload
    cores = available_cores()
    state = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0)
    rays = 256*cores
    columns = [np.full(rays, state[k]) for k in STATE]
    gfir.Item(workload("loss_kernel_kx")).converge(columns)
    how = "compiled by gcc from the DAG (oracle/gfir_to_c.py)"
    try:
        from oracle import gfir_to_c
        item = gfir_to_c.CompiledItem(workload("solver_kernel"))
    except Exception:
        item = gfir.Item(workload("solver_kernel"))
        how = "interpreted by oracle/gfir_interp.c"
    item.run(columns, steps=1, threads=cores)                       # warm: first touch, thread start
    steps, seconds, chunk = 0, 0.0, 4
    while seconds < target_seconds and steps < 20000:
        _, took = item.run(columns, steps=chunk, threads=cores)
        steps += chunk
        seconds += took
        chunk = min(chunk*2, 512)
    sample = ("%d rays x %d RK4 steps of the same solver_kernel DAG, strict IEEE, %s, %d threads (%.1f s)"
              % (rays, steps, how, cores, seconds))
    return rays*steps/seconds, cores, sample


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=10)
    parser.add_argument("--rays-per-gpu", type=int, default=1000000)
    parser.add_argument("--total-rays", type=int, default=0,
                        help="strong scaling (BASELINE configs[3]): a fixed ensemble split over the ranks as the "
                             "reference splits it over device threads (xrays_bench.cpp:38-51); overrides --rays-per-gpu")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                        help="torch.distributed backend (default nccl = RCCL); gloo + --share-gpu rehearses "
                             "the multi-rank path on a one-GPU box")
    parser.add_argument("--share-gpu", action="store_true",
                        help="rehearsal only: every rank uses device 0")
    parser.add_argument("--distribution", choices=["bench", "cli"], default="bench",
                        help="bench: identical rays of xrays_bench.cpp:62-71 (default, the metric's workload); "
                             "cli: the incoherent example distribution of graph_driver/xrays.cpp (BASELINE configs[2])")
    args = parser.parse_args()

    import torch
    from graph_framework_amd import distributed as gfd
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, STATE, shard_bounds, workload

    rank, world, local_rank = gfd.init(args.backend, device_index=0 if args.share_gpu else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    if args.total_rays > 0:
        begin, end = shard_bounds(args.total_rays, world, rank)
        n_local = end - begin
        total = args.total_rays
    else:
        n_local = args.rays_per_gpu
        total = n_local*world

#  Rank 0 reads the work items (they carry the equilibrium tables); RCCL broadcast to the rest.
    items = {}
    for name in ("loss_kernel_kx", "solver_kernel"):
        data = b""
        if rank == 0:
            with open(workload(name), "rb") as f:
                data = f.read()
        items[name] = gfd.broadcast_bytes(data, 0)

    if args.distribution == "bench":
        state = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0)
        initial = {k: np.full(n_local, v) for k, v in state.items()}
    else:
        from graph_framework_amd.xrays import cli_distribution
        initial = cli_distribution(n_local, seed=rank)
    solve = Rk4ColdPlasmaEfit(initial, index=local_rank, stream=torch.cuda.current_stream().cuda_stream,
                              items=items)
    solve.init("kx")
    solve.compile()

    for _ in range(args.warmup):
        solve.step()
#  HIP events on the launch stream around every 8th step of the timed region (an event pair
#  costs the stream 2-8 us; around every launch they would slow the loop they measure by 1-3 %).
    timing_period = 8 if args.steps >= 64 else 1
    solve.work.context.enable_timing(True, every=timing_period)

    gfd.barrier()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(args.steps):
        solve.step()
    torch.cuda.synchronize()
    gfd.barrier()
    elapsed = time.perf_counter() - start
    sync_start = time.perf_counter()
    host = solve.sync_host()
    sync_elapsed = time.perf_counter() - sync_start

    elapsed = gfd.max_over_ranks(elapsed)
    sync_elapsed = gfd.max_over_ranks(sync_elapsed)
    kernel_ms, launches = solve.solver.kernel.timing()
    kernel_ms = gfd.max_over_ranks(kernel_ms)

#  Output-cadence collective: all-gather of the trajectory state over xGMI (not in the step loop).
    gather_seconds = None
    if world > 1:
        on_device = torch.distributed.get_backend() == "nccl"
        shards = {k: (torch.from_numpy(host[k]).cuda() if on_device else torch.from_numpy(host[k])) for k in STATE}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in STATE:
            full = gfd.all_gather_shards(shards[k], total)
            assert full.numel() == total
        torch.cuda.synchronize()
        gather_seconds = gfd.max_over_ranks(time.perf_counter() - t0)

    if rank == 0:
        info = solve.solver.kernel.info()
        value = total*args.steps/elapsed
        flops = dag_flops(workload("solver_kernel"))
        achieved = n_local*BYTES_PER_RAY_STEP_F64/(kernel_ms*1.0e-3)/1.0e9 if kernel_ms > 0 else 0.0
        line = {
            "metric": "ray-steps/sec on xrays_bench cold-plasma; achieved HBM GB/s vs peak",
            "value": value,
            "unit": "ray-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1.0e3*elapsed/args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.total_rays > 0 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("xrays_bench cold-plasma RK4 solver_kernel on EFIT (efit.nc), identical rays "
                                    "omega=500 x=2.5 kx=Newton(-600), dt=1e-3, fp64") if args.distribution == "bench" else
                                   ("xrays cold-plasma RK4 solver_kernel on EFIT (efit.nc), incoherent CLI example "
                                    "distribution, dt=1e-3, fp64"),
                       "rays_per_gpu": n_local, "total_rays": total, "parallelism": "rays sharded x%d" % world,
                       "kernel_nodes": int(info.num_instructions), "vgprs": int(info.vgprs),
                       "lds_bytes": int(info.lds_bytes), "scratch_bytes": int(info.scratch_bytes),
                       "code_object_from_cache": bool(info.from_cache)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved/HBM_PEAK_GBPS, "traffic": measured_traffic(info.name.decode(), n_local),
                         "kernel": info.name.decode(), "kernel_ms": kernel_ms, "launches": int(launches),
                         "timed_every": timing_period,
                         "algorithmic_bytes_per_launch": n_local*BYTES_PER_RAY_STEP_F64,
                         "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r01_solver_kernel.md)",
                         "note": "the kernel is FP64-VALU issue bound (6.4k vector instructions per ray-step, "
                                 "VALU busy 86 % at one wave per SIMD), not HBM bound: see DESIGN.md section 3"},
            "fp64_vector": {"flops_per_ray_step": flops, "achieved_tflops": value/world*flops/1.0e12,
                            "peak_tflops": 78.6, "frac": value/world*flops/1.0e12/78.6,
                            "note": "per GPU; reference-DAG operation count, the roof that binds this kernel"},
            "value_with_sync_host": total*args.steps/(elapsed + sync_elapsed),
            "newton_iterations": solve.newton_iterations,
        }
        if gather_seconds is not None:
            line["all_gather_seconds"] = gather_seconds
        if world == 1 and not args.no_cpu_baseline:
            rate, cores, sample = cpu_baseline()
            line["cpu_baseline"] = {"value": rate, "unit": "ray-steps/s", "cores": cores, "kind": "port",
                                    "sample": sample}
        print(json.dumps(line))


if __name__ == "__main__":
    main()
