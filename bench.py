#!/usr/bin/env python3
"""Benchmark of the hot path: ray-steps/s of the xrays_bench cold-plasma RK4 `solver_kernel`.

Workload: graph_benchmark/xrays_bench.cpp:53-102 — every ray omega=500, x=2.5, kx=-600 -> Newton,
dt=1e-3, cold_plasma on the EFIT equilibrium of graph_tests/efit.nc, fp64.  A "step" is one launch
of `solver_kernel` over the rank's rays (solver_interface::step, solver.hpp:382).  Setup, the
Newton init and kernel builds are outside the timed region, as in the reference
(xrays_bench.cpp:88-102); the state is resident in HBM when the timed region starts.

Sizes:
    --gpus 1  (default)  1e7 rays, the size BASELINE.json's north_star quotes its target on
                         (--rays-per-gpu 1000000 is BASELINE configs[1]);
    --gpus N > 1         one ensemble of 1e8 rays (BASELINE configs[3]) split over the ranks as the
                         reference splits it over its device threads (xrays_bench.cpp:38-51):
                         strong scaling.  --rays-per-gpu R selects weak scaling instead.

Launch: one process per GPU.  Under torch.distributed.run (RANK/WORLD_SIZE in the environment) this
process is one rank.  WITHOUT that environment `--gpus N` starts its own N rank processes — before
anything here touches the GPU — collects rank 0's line and exits non-zero if any rank failed, like
the reference's own benchmark, which needs no launcher (one thread per device,
xrays_bench.cpp:34-108).

`--workload korc` runs the xkorc push instead (graph_korc/xkorc.cpp:29-154, fp32, BASELINE
configs[4]: 1e7 particles, split over the ranks like the reference's device threads) and prints a
line of the same shape with metric particle-steps/s.

`value` is the step-loop rate (K launches between barrier + synchronize on both sides).  The
reference's timed region also holds the final sync_host (8 D2H copies, xrays_bench.cpp:95-102);
that PCIe-inclusive rate of the same run is `value_with_sync_host`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_RAY_STEP_F64 = 128      # read 8 + write 8 (7 setters + residual) x 8 B, SURVEY.md §8(d)
BYTES_PER_RAY_ITERATION_F64 = 72  # loss_kernel, one pass: read 7 (t is not read) + write 2 (the setter target and the residual); the max is
                                  # reduced inside the launch.  A launch of `<name>_batch` runs several passes on state kept in registers
                                  # and also saves the setter target for the undo: 7 reads + 3 writes = 80 B per ray per LAUNCH.
BYTES_PER_RAY_BATCH_F64 = 80
BYTES_PER_PARTICLE_STEP_F32 = 56  # xkorc step: (7 reads + 7 writes) x 4 B, SURVEY.md §8(d)
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
DEFAULT_RAYS_ONE_GPU = 10000000   # north_star: "1e7 cold-plasma rays at 1 MI355X"
DEFAULT_TOTAL_RAYS = 100000000    # BASELINE configs[3]
BENCH_RAY = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0)


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=10)
    parser.add_argument("--rays-per-gpu", type=int, default=0,
                        help="weak scaling: this many rays on every rank (default at --gpus 1: 1e7)")
    parser.add_argument("--total-rays", type=int, default=0,
                        help="strong scaling: one ensemble split over the ranks, xrays_bench.cpp:38-51 "
                             "(default at --gpus N > 1: 1e8, BASELINE configs[3])")
    parser.add_argument("--warmup-seconds", type=float, default=0.3,
                        help="keep launching untimed steps after the --warmup steps until this much time has passed "
                             "(the chip settles its clock under load over ~0.2 s)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-extra", action="store_true",
                        help="skip the secondary roofline objects (loss_kernel, xkorc step) and the weak-scaling leg")
    parser.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                        help="torch.distributed backend (default nccl = RCCL); gloo + --share-gpu rehearses "
                             "the multi-rank path on a one-GPU box")
    parser.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses device 0")
    parser.add_argument("--rehearse-cpu", action="store_true",
                        help="no GPU: run only the multi-rank plumbing (rendezvous over gloo, item broadcast, shard "
                             "split, all-gather, max-over-ranks) and print a rehearsal line; covered by tests/")
    parser.add_argument("--fail-rank", type=int, default=-1, help="(rehearsal) this rank exits with an error")
    parser.add_argument("--workload", choices=["rays", "korc"], default="rays",
                        help="rays: the metric's workload (xrays_bench solver_kernel, fp64); korc: the xkorc particle "
                             "push of BASELINE configs[4] (fp32; --gpus N > 1 splits 1e7 particles over the ranks as "
                             "graph_korc/xkorc.cpp:16-25 splits them over its device threads) -- a second bench "
                             "line with its own metric, never the headline")
    parser.add_argument("--force-collectives", action="store_true",
                        help="initialise the process group for ONE rank as well and run every collective through the "
                             "backend (--gpus 1 --backend nccl --force-collectives executes the RCCL code path on a "
                             "one-GPU box)")
    parser.add_argument("--distribution", choices=["bench", "cli"], default="bench",
                        help="bench: identical rays of xrays_bench.cpp:62-71 (the metric's workload); "
                             "cli: the incoherent example distribution of graph_driver/xrays.cpp (BASELINE configs[2])")
    return parser.parse_args(argv)


# ----------------------------------------------------------------------------------------------
#  Self-launch: N rank processes, started before any GPU call of this process.
# ----------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(world, argv, timeout=None):
    """Start `world` copies of this script as ranks 0..world-1 (fresh processes, rendezvous on
    127.0.0.1), wait for all of them, print rank 0's stdout.  Returns the exit code: non-zero
    if any rank failed (the others are then stopped, by PID)."""
    port = free_port()
    procs, files = [], []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = tempfile.TemporaryFile(mode="w+")
        files.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    deadline = None if timeout is None else time.time() + timeout
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for rank, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = rank
        if deadline is not None and time.time() > deadline:
            failed = -1
        time.sleep(0.1)
    if failed is None:
        for rank, p in enumerate(procs):
            if p.returncode != 0:
                failed = rank
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: rank %s failed (exit codes %s)\n"
                         % ("timeout" if failed < 0 else failed, [p.returncode for p in procs]))
#  Rank 0's result line goes to stdout; anything else a library wrote there (gloo announces its
#  connections on stdout) goes to stderr, so that stdout stays ONE JSON line.
    files[0].seek(0)
    for text in files[0].read().splitlines():
        (sys.stdout if text.startswith("{") else sys.stderr).write(text + "\n")
    sys.stdout.flush()
    for f in files:
        f.close()
    return 0 if failed is None else 1


# ----------------------------------------------------------------------------------------------
#  Helpers of one rank.
# ----------------------------------------------------------------------------------------------
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


def vector_issue(gfir_path, rays, kernel_ms, num_cus=256):
    """The roof that binds the RK4 kernel: vector instructions of one pass of its assembly body (counted in the kernel
    text the lowering writes) and the time a SIMD spends per instruction of a wave, next to what a microbenchmark of
    nothing but fp64 vector instructions gets from a SIMD of this chip (profiles/diag/fp64_issue/)."""
    try:
        from graph_framework_amd.backend import generate_piece_sources
        text = generate_piece_sources(gfir_path)[0][0]
        start = text.index("asm volatile(\n", text.index("float dmax"))
        statement = text[start:text.index("                : [", start)]
    except (ValueError, IndexError):
        return None                                  # the compiled body (GFHIP_ASM=0): hipcc's count is not in the text
    instructions = sum(1 for line in statement.split("\n") if line.lstrip().startswith('"v_'))
    tiles_per_simd = rays/64.0/(4.0*num_cus)
    return {"vector_instructions_per_pass": instructions,
            "ns_per_instruction": 1.0e6*kernel_ms/(tiles_per_simd*instructions) if kernel_ms > 0 and instructions else None,
            "microbenchmark_ns_per_instruction": [1.83, 2.43],
            "note": "ns a SIMD spends per vector instruction of a pass (two waves resident, every SIMD of 256 CUs busy); the "
                    "microbenchmark range is dependent mul/add chains at four waves to one dependent fma chain at one wave"}


def measured_traffic(kernel_name, source_hash, rays):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json, written by
    profiles/summarize.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command) —
    only if they were taken for THIS code object (source hash) and launch size; else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            entries = json.load(f)
    except (OSError, ValueError):
        return None
    for entry in entries if isinstance(entries, list) else []:
        if (entry.get("kernel") == kernel_name and entry.get("source_hash") == "%016x" % source_hash
                and entry.get("rays_per_launch") == rays):
            return entry.get("traffic_bytes_per_launch")
    return None


def sample_statistics(samples):
    """Spread of the per-launch HIP-event durations (ms) over the timed region, first quarter
    against last quarter included: a clock that is still settling shows up as a trend."""
    if not samples:
        return None
    ordered = sorted(samples)
    quarter = max(1, len(samples)//4)
    return {"min": ordered[0], "median": ordered[len(ordered)//2], "max": ordered[-1],
            "first_quarter_mean": sum(samples[:quarter])/quarter, "last_quarter_mean": sum(samples[-quarter:])/quarter}


def cpu_baseline(target_seconds=12.0):
    """The CPU oracle timed on a bounded sample of the same workload: the solver_kernel DAG as
    the reference's cpu_context runs it (one compiled statement per node, serial loop per
    thread, contiguous shards; oracle/gfir_to_c.py, gcc -O2, strict IEEE), one thread per
    available core.  Falls back to the interpreter (oracle/gfir_interp.c) without gcc."""
    import numpy as np
    from oracle import gfir
    from graph_framework_amd.xrays import STATE, workload
    cores = available_cores()
    rays = 256*cores
    columns = [np.full(rays, BENCH_RAY[k]) for k in STATE]
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


def cpu_baseline_korc(target_seconds=8.0):
    """The CPU oracle timed on a bounded sample of the xkorc push: the `step` DAG in fp32 as the reference's
    cpu_context runs it (one compiled statement per node, serial loop per thread, contiguous shards;
    oracle/gfir_to_c.py, gcc -O2, strict IEEE), one thread per available core."""
    import numpy as np
    from oracle import gfir, gfir_to_c
    from graph_framework_amd import korc as gk
    from graph_framework_amd.xrays import workload
    cores = available_cores()
    particles = 4096*cores
    start = dict(x=1.7, y=0.0, z=0.0, ux=0.0, uy=0.99, uz=0.1, gamma=0.0)
    columns = {k: np.full(particles, start[k], dtype=np.float32) for k in gk.PARTICLE}
    gfir.Item(workload("korc_initialize_gamma", "f32")).run([columns[k] for k in ("ux", "uy", "uz", "gamma")])
    how = "compiled by gcc from the DAG (oracle/gfir_to_c.py)"
    try:
        item = gfir_to_c.CompiledItem(workload("korc_step", "f32"))
    except Exception:
        item = gfir.Item(workload("korc_step", "f32"))
        how = "interpreted by oracle/gfir_interp.c"
    ordered = [columns[k] for k in gk.PARTICLE]
    item.run(ordered, steps=1, threads=cores)
    steps, seconds, chunk = 0, 0.0, 8
    while seconds < target_seconds and steps < 100000:
        _, took = item.run(ordered, steps=chunk, threads=cores)
        steps += chunk
        seconds += took
        chunk = min(chunk*2, 2048)
    sample = ("%d particles x %d steps of the same korc `step` DAG, fp32, strict IEEE, %s, %d threads (%.1f s)"
              % (particles, steps, how, cores, seconds))
    return particles*steps/seconds, cores, sample


def hbm_roofline_from(kernel, samples, units, bytes_per_unit):
    """roofline object of one kernel from a list of its HIP-event launch durations (ms)."""
    ms = sum(samples)/len(samples) if samples else 0.0
    info = kernel.info()
    achieved = units*bytes_per_unit/(ms*1.0e-3)/1.0e9 if ms > 0 else 0.0
    name = info.name.decode()
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved/HBM_PEAK_GBPS,
            "traffic": measured_traffic(name, info.source_hash, units),
            "kernel": name, "kernel_ms": ms, "launches": len(samples), "units_per_launch": units,
            "algorithmic_bytes_per_launch": units*bytes_per_unit, "source_hash": "%016x" % info.source_hash}


def hbm_roofline(kernel, units, bytes_per_unit, note=None):
    """roofline object of one kernel from its HIP-event launch durations (gfhip_kernel_timing)."""
    ms, launches = kernel.timing()
    info = kernel.info()
    achieved = units*bytes_per_unit/(ms*1.0e-3)/1.0e9 if ms > 0 else 0.0
    name = info.name.decode()
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved/HBM_PEAK_GBPS,
           "traffic": measured_traffic(name, info.source_hash, units),
           "kernel": name, "kernel_ms": ms, "launches": int(launches), "units_per_launch": units,
           "algorithmic_bytes_per_launch": units*bytes_per_unit, "source_hash": "%016x" % info.source_hash}
    if note:
        out["note"] = note
    return out


def extra_rooflines(n_loss, newton_roofline):
    """Secondary kernels of the path at BASELINE.json's sizes, outside the timed region: the
    Newton `loss_kernel` (measured during this run's init) and the xkorc fp32 push, 1e7 particles."""
    import numpy as np
    from graph_framework_amd import korc as gk
    extras = {"loss_kernel": newton_roofline}
    n = 10000000
    push = gk.Korc(dict(x=1.7, y=0.0, z=0.0, ux=0.0, uy=0.99, uz=0.1, gamma=np.zeros(n)), "f32")
    push.compile()
    push.pre_run()
    for _ in range(20):
        push.run()
    push.wait()
    push.work.context.enable_timing(True, every=4)
    start = time.perf_counter()
    steps = 200
    for _ in range(steps):
        push.run()
    push.wait()
    elapsed = time.perf_counter() - start
    roof = hbm_roofline(push.step_item.kernel, n, BYTES_PER_PARTICLE_STEP_F32)
    roof["value"] = n*steps/elapsed
    roof["value_unit"] = "particle-steps/s"
    roof["workload"] = "xkorc step fp32, 1e7 particles on one GPU (BASELINE configs[4] per-GPU kernel)"
    extras["korc_step_f32"] = roof
    push.work.context.close()
    return extras


def fast_division_rate(rays, steps=100):
    """The opt-in tolerance mode GFHIP_DIVISION=fast (q = n*r, no residual step, no checks; NOT bit-exact): the same
    step loop on the same rays, reported under its own key and never as `value`.  Its gate is
    tests/test_gpu_division.py::test_fast_division_mode_meets_the_trajectory_tolerance."""
    import numpy as np
    import torch
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    os.environ["GFHIP_DIVISION"] = "fast"
    try:
        solve = Rk4ColdPlasmaEfit({k: np.full(rays, v) for k, v in BENCH_RAY.items()})
        solve.init("kx")
        solve.compile()
    finally:
        del os.environ["GFHIP_DIVISION"]
    start = time.perf_counter()
    while time.perf_counter() - start < 0.3:
        for _ in range(10):
            solve.step()
        solve.work.wait()
    solve.work.context.enable_timing(True, every=4)
    start = time.perf_counter()
    for _ in range(steps):
        solve.step()
    solve.work.wait()
    elapsed = time.perf_counter() - start
    ms, _ = solve.solver.kernel.timing()
    info = solve.solver.kernel.info()
    solve.work.context.close()
    torch.cuda.synchronize()
    return {"value": rays*steps/elapsed, "unit": "ray-steps/s", "kernel_ms": ms, "steps": steps, "vgprs": int(info.vgprs),
            "scratch_bytes": int(info.scratch_bytes), "source_hash": "%016x" % info.source_hash,
            "note": "GFHIP_DIVISION=fast, opt-in: every quotient within ~1.5 ulp, NOT the reference's bits; the benchmark ray stays "
                    "within 1e-6 of the reference record over 1000 steps, an incoherent beam is as far from the reference as the "
                    "reference's own one-ulp neighbour (tests/test_gpu_division.py).  Never the headline."}


def rehearse_cpu(args):
    """The rank-side plumbing of run_rank without a device: what tests/test_distributed.py drives
    through the self-launcher (and through torch.distributed.run) on a CPU-only host."""
    import torch
    from graph_framework_amd import distributed as gfd
    from graph_framework_amd import korc as gk
    from graph_framework_amd.xrays import STATE, shard_bounds, workload
    rank, world, _ = gfd.init("gloo", force_collectives=args.force_collectives)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if rank == args.fail_rank:
        raise SystemExit("rank %d fails on request" % rank)
    total = args.total_rays or (args.rays_per_gpu*world if args.rays_per_gpu else 1003)
    begin, end = shard_bounds(total, world, rank)
    if args.workload == "korc":
        names, dtype, columns, torch_dtype = gk.ITEMS, "f32", gk.PARTICLE, torch.float32
    else:
        names, dtype, columns, torch_dtype = ("loss_kernel_kx",), "f64", STATE, torch.float64
    item_bytes = 0
    for name in names:
        data = b""
        if rank == 0:
            with open(workload(name, dtype), "rb") as f:
                data = f.read()
        item = gfd.broadcast_bytes(data, 0)
        with open(workload(name, dtype), "rb") as f:
            assert item == f.read()
        item_bytes += len(item)
    gathered = 0
    for k in range(len(columns)):
        full = gfd.all_gather_shards(torch.arange(begin, end, dtype=torch_dtype) + k, total)
        assert torch.equal(full, torch.arange(total, dtype=torch_dtype) + k)
        gathered += full.numel()
    slowest = gfd.max_over_ranks(float(rank))
    gfd.barrier()
    if rank == 0:
        print(json.dumps({"rehearsal": "cpu", "workload": args.workload, "n_gpus": world, "total_rays": total,
                          "item_bytes": item_bytes, "items": len(names),
                          "gathered_elements": gathered, "slowest_rank": slowest,
                          "launcher": os.environ.get("GF_BENCH_LAUNCHER", "torchrun")}))
        sys.stdout.flush()


def collective_info(args):
    """Which backend ran the collectives of this line (and that they were forced for one rank)."""
    import torch
    info = {"backend": torch.distributed.get_backend(), "forced_for_one_rank": bool(args.force_collectives)}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        info["rccl_version"] = None
    return info


def run_rank_korc(args):
    """--workload korc: the xkorc push (graph_korc/xkorc.cpp:29-154) in fp32, one shard per rank.

    The reference starts one host thread per device and gives thread i
    `batch + (extra > i)` particles (xkorc.cpp:16-25), every thread building its own equilibrium and
    characteristic field (:29-31) and its own workflow::manager(thread_number) (:74).  Here: one
    process per GPU with the same split (shard_bounds), rank 0 reads the four work items and
    broadcasts them (the `step` item carries the folded EFIT tables and b0), every rank solves the
    characteristic field itself, runs the pre-item and steps independently; the seven particle
    arrays are all-gathered from the device tensors at output cadence.  No per-step collective."""
    import numpy as np
    import torch
    from graph_framework_amd import distributed as gfd
    from graph_framework_amd import korc as gk
    from graph_framework_amd.xrays import shard_bounds, workload

    rank, world, local_rank = gfd.init(args.backend, device_index=0 if args.share_gpu else None,
                                       force_collectives=args.force_collectives)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    collectives = world > 1 or args.force_collectives

    per_gpu, total = args.rays_per_gpu, args.total_rays
    if not per_gpu and not total:
        total = DEFAULT_RAYS_ONE_GPU                        # BASELINE configs[4]: 1e7 particles in all
    strong = total > 0
    if strong:
        begin, end = shard_bounds(total, world, rank)
        n_local = end - begin
    else:
        n_local, total = per_gpu, per_gpu*world
        begin = rank*per_gpu

    items = {}
    for name in gk.ITEMS:
        data = b""
        if rank == 0:
            with open(workload(name, "f32"), "rb") as f:
                data = f.read()
        items[name] = gfd.broadcast_bytes(data, 0)

    b0, axis_iterations = gk.characteristic_field("f32", index=local_rank, items=items)
#  xkorc.cpp:47-64: every particle x = (1.7, 0, 0), u = (0, 0.99, 0.1).
    push = gk.Korc(dict(x=1.7, y=0.0, z=0.0, ux=0.0, uy=0.99, uz=0.1, gamma=np.zeros(n_local)), "f32",
                   index=local_rank, items=items, device_state=True)
    push.compile()
    push.pre_run()
    warm_start = time.perf_counter()
    warm_steps = 0
    for _ in range(args.warmup):
        push.run()
        warm_steps += 1
    torch.cuda.synchronize()
    while time.perf_counter() - warm_start < args.warmup_seconds:
        for _ in range(10):
            push.run()
        warm_steps += 10
        torch.cuda.synchronize()
    timing_period = 8 if args.steps >= 64 else 1
    push.work.context.enable_timing(True, every=timing_period)

    gfd.barrier()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(args.steps):
        push.run()
    torch.cuda.synchronize()
    gfd.barrier()
    elapsed = gfd.max_over_ranks(time.perf_counter() - start)
    samples = push.step_item.kernel.timing_samples()
    kernel_ms = gfd.max_over_ranks(sum(samples)/len(samples) if samples else 0.0)
    push.work.context.enable_timing(False)

    gather_seconds = None
    checksum = None
    if collectives:
        on_device = torch.distributed.get_backend() == "nccl"
        torch.cuda.synchronize()
        gfd.barrier()
        t0 = time.perf_counter()
        checksum = 0.0
        for k in gk.PARTICLE:
            shard = push.device[k] if on_device else push.device[k].cpu()
            full = gfd.all_gather_shards(shard, total)
            assert full.numel() == total
            checksum += float(full.double().sum().item())
            del full
        torch.cuda.synchronize()
        gather_seconds = gfd.max_over_ranks(time.perf_counter() - t0)

    if rank != 0:
        return
    info = push.step_item.kernel.info()
    name = info.name.decode()
    achieved = n_local*BYTES_PER_PARTICLE_STEP_F32/(kernel_ms*1.0e-3)/1.0e9 if kernel_ms > 0 else 0.0
    distributed_info = {"world_size": world, "launcher": os.environ.get("GF_BENCH_LAUNCHER", "torchrun" if world > 1 else "none")}
    if collectives:
        distributed_info.update(collective_info(args))
    line = {
        "metric": "particle-steps/sec on the xkorc push (BASELINE configs[4]); achieved HBM GB/s vs peak",
        "value": total*args.steps/elapsed,
        "unit": "particle-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1.0e3*elapsed/args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "xkorc relativistic push `step` on EFIT (efit.nc), x=(1.7,0,0) u=(0,0.99,0.1) dt=0.5, fp32 "
                               "(graph_korc/xkorc.cpp:47-121)",
                   "particles_per_gpu": n_local, "total_particles": total,
                   "parallelism": "particles sharded x%d (xkorc.cpp:16-25)" % world,
                   "kernel_nodes": int(info.num_instructions), "vgprs": int(info.vgprs),
                   "code_object_from_cache": bool(info.from_cache), "warmup_steps_run": warm_steps,
                   "b0": b0, "axis_newton_iterations": axis_iterations},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved/HBM_PEAK_GBPS, "traffic": measured_traffic(name, info.source_hash, n_local),
                     "kernel": name, "kernel_ms": kernel_ms, "launches": len(samples), "timed_every": timing_period,
                     "algorithmic_bytes_per_launch": n_local*BYTES_PER_PARTICLE_STEP_F32,
                     "source_hash": "%016x" % info.source_hash},
        "distributed": distributed_info,
    }
    if gather_seconds is not None:
        line["all_gather_seconds"] = gather_seconds
        line["all_gather_bytes"] = total*4*len(gk.PARTICLE)
        line["all_gather_checksum"] = checksum
    if world == 1 and not args.no_cpu_baseline:
        rate, cores, sample = cpu_baseline_korc()
        line["cpu_baseline"] = {"value": rate, "unit": "particle-steps/s", "cores": cores, "kind": "port", "sample": sample}
    print(json.dumps(line))
    sys.stdout.flush()


def run_rank(args):
    import numpy as np
    import torch
    from graph_framework_amd import distributed as gfd
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, STATE, shard_bounds, workload

    rank, world, local_rank = gfd.init(args.backend, device_index=0 if args.share_gpu else None,
                                       force_collectives=args.force_collectives)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    collectives = world > 1 or args.force_collectives

    rays_per_gpu, total_rays = args.rays_per_gpu, args.total_rays
    if not rays_per_gpu and not total_rays:
        if world == 1:
            rays_per_gpu = DEFAULT_RAYS_ONE_GPU
        else:
            total_rays = DEFAULT_TOTAL_RAYS
    strong = total_rays > 0
    if strong:
        begin, end = shard_bounds(total_rays, world, rank)
        n_local, total = end - begin, total_rays
    else:
        n_local, total = rays_per_gpu, rays_per_gpu*world

#  Rank 0 reads the work items (they carry the equilibrium tables); RCCL broadcast to the rest.
    items = {}
    for name in ("loss_kernel_kx", "solver_kernel"):
        data = b""
        if rank == 0:
            with open(workload(name), "rb") as f:
                data = f.read()
        items[name] = gfd.broadcast_bytes(data, 0)

    def make_solver(count, seed):
        if args.distribution == "bench":
            initial = {k: np.full(count, v) for k, v in BENCH_RAY.items()}
        else:
            from graph_framework_amd.xrays import cli_distribution
            initial = cli_distribution(count, seed=seed)
#  The state lives in torch CUDA tensors adopted by the context (device_state): the output-cadence
#  all-gather below reads them in place.
        solver = Rk4ColdPlasmaEfit(initial, index=local_rank, items=items, device_state=True)
        solver.work.context.enable_timing(True, every=1)
        solver.init("kx")
        solver.compile()
        return solver

    newton_start = time.perf_counter()
    solve = make_solver(n_local, rank)
    newton_wall = time.perf_counter() - newton_start
    newton_samples = solve.newton.kernel.timing_samples()
    newton_batch = max(int(solve.newton.kernel.info().converge_batch), 1)
    newton_roofline = hbm_roofline_from(solve.newton.kernel, newton_samples, n_local,
                                        BYTES_PER_RAY_BATCH_F64 if newton_batch > 1 else BYTES_PER_RAY_ITERATION_F64)
    newton_roofline.update({
        "passes": solve.newton_iterations + 1, "passes_per_launch": newton_batch,
        "init_kernel_ms_total": sum(newton_samples), "ms_per_pass": sum(newton_samples)/(solve.newton_iterations + 1),
        "note": "Newton init of this run (workflow.hpp:179-205 on the global max, %d passes): launches of `<name>_batch` run up to %d "
                "passes on state kept in registers, each pass with its own max, the loop's test on the device; the last launch redoes "
                "the final batch with the exact pass count when the loop ended inside it.  setup_seconds_incl_build_and_upload = %.3f"
                % (solve.newton_iterations + 1, newton_batch, newton_wall)})
    solve.work.context.enable_timing(False)

    warm_start = time.perf_counter()
    warm_steps = 0
    for _ in range(args.warmup):
        solve.step()
        warm_steps += 1
    torch.cuda.synchronize()
    while time.perf_counter() - warm_start < args.warmup_seconds:
        for _ in range(10):
            solve.step()
        warm_steps += 10
        torch.cuda.synchronize()
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
    solve.sync_host()
    sync_elapsed = time.perf_counter() - sync_start

    elapsed = gfd.max_over_ranks(elapsed)
    sync_elapsed = gfd.max_over_ranks(sync_elapsed)
    samples = solve.solver.kernel.timing_samples()
    kernel_ms = sum(samples)/len(samples) if samples else 0.0
    launches = len(samples)
    kernel_ms = gfd.max_over_ranks(kernel_ms)
    solve.work.context.enable_timing(False)

#  Output-cadence collective: all-gather of the trajectory state over xGMI, from the device
#  tensors the kernels write (not in the step loop; gloo rehearsals gather host copies).
    gather_seconds = None
    if collectives:
        on_device = torch.distributed.get_backend() == "nccl"
        torch.cuda.synchronize()
        gfd.barrier()
        t0 = time.perf_counter()
        for k in STATE:
            shard = solve.device[k] if on_device else solve.device[k].cpu()
            full = gfd.all_gather_shards(shard, total)
            assert full.numel() == total
            del full
        torch.cuda.synchronize()
        gather_seconds = gfd.max_over_ranks(time.perf_counter() - t0)

#  Weak-scaling leg beside a strong-scaling run: the same loop at 1e7 rays on every rank.
    weak = None
    if strong and world > 1 and not args.no_extra:
        weak_rays, weak_steps = DEFAULT_RAYS_ONE_GPU, 50
        other = make_solver(weak_rays, rank)
        other.work.context.enable_timing(False)
        for _ in range(10):
            other.step()
        gfd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(weak_steps):
            other.step()
        torch.cuda.synchronize()
        gfd.barrier()
        weak_elapsed = gfd.max_over_ranks(time.perf_counter() - t0)
        weak = {"scaling": "weak", "rays_per_gpu": weak_rays, "steps": weak_steps,
                "value": weak_rays*world*weak_steps/weak_elapsed, "unit": "ray-steps/s"}
        other.work.context.close()

    if rank != 0:
        return
    info = solve.solver.kernel.info()
    value = total*args.steps/elapsed
    flops = dag_flops(workload("solver_kernel"))
    name = info.name.decode()
    achieved = n_local*BYTES_PER_RAY_STEP_F64/(kernel_ms*1.0e-3)/1.0e9 if kernel_ms > 0 else 0.0
    distributed_info = {"world_size": world, "launcher": os.environ.get("GF_BENCH_LAUNCHER", "torchrun" if world > 1 else "none")}
    if collectives:
        distributed_info.update(collective_info(args))
    line = {
        "metric": "ray-steps/sec on xrays_bench cold-plasma; achieved HBM GB/s vs peak",
        "value": value,
        "unit": "ray-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1.0e3*elapsed/args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
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
                   "code_object_from_cache": bool(info.from_cache), "warmup_steps_run": warm_steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved/HBM_PEAK_GBPS, "traffic": measured_traffic(name, info.source_hash, n_local),
                     "kernel": name, "kernel_ms": kernel_ms, "launches": int(launches),
                     "timed_every": timing_period, "kernel_ms_samples": sample_statistics(samples),
                     "algorithmic_bytes_per_launch": n_local*BYTES_PER_RAY_STEP_F64,
                     "source_hash": "%016x" % info.source_hash,
                     "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE of profiles/traffic.json), null "
                                     "unless measured for this source hash and launch size",
                     "note": "the kernel is FP64-VALU issue bound, not HBM bound: ~5.9 k vector instructions per ray-step "
                             "at ~4.2 cycles each, the vector unit busy 98 % of the time its waves are resident (two waves "
                             "per SIMD, the pass as gfx950 assembly with a register assignment of its own: "
                             "csrc/asm_body.hpp, DESIGN.md section 3)"},
        "vector_issue": vector_issue(workload("solver_kernel"), n_local, kernel_ms),
        "fp64_vector": {"flops_per_ray_step": flops, "achieved_tflops": value/world*flops/1.0e12,
                        "peak_tflops": 78.6, "frac": value/world*flops/1.0e12/78.6,
                        "note": "per GPU; reference-DAG operation count, the roof that binds this kernel"},
        "value_with_sync_host": total*args.steps/(elapsed + sync_elapsed),
        "sync_host_seconds": sync_elapsed,
        "newton_iterations": solve.newton_iterations,
        "distributed": distributed_info,
    }
    if gather_seconds is not None:
        line["all_gather_seconds"] = gather_seconds
        line["all_gather_bytes"] = total*8*len(STATE)
    if weak is not None:
        line["weak_scaling"] = weak
    if world == 1 and not args.no_extra:
        line["roofline_extra"] = extra_rooflines(n_local, newton_roofline)
        if args.distribution == "bench":
            solve.work.context.close()
            line["fast_division"] = fast_division_rate(n_local)
    if world == 1 and not args.no_cpu_baseline:
        rate, cores, sample = cpu_baseline()
        line["cpu_baseline"] = {"value": rate, "unit": "ray-steps/s", "cores": cores, "kind": "port",
                                "sample": sample}
    print(json.dumps(line))
    sys.stdout.flush()


def main():
    args = parse_arguments()
    if args.force_collectives and args.gpus == 1:
#  A one-rank group still needs a rendezvous address.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
#  No launcher around us: become the launcher.  Nothing above has imported torch or touched HIP.
        os.environ["GF_BENCH_LAUNCHER"] = "bench.py"
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
#  stdout is ONE JSON line: libraries announce themselves on file descriptor 1 (RCCL prints its
#  version banner there, gloo its connections), so descriptor 1 is pointed at stderr for the life
#  of the rank and Python's sys.stdout keeps the real one.
    sys.stdout.flush()
    sys.stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.rehearse_cpu:
        rehearse_cpu(args)
    elif args.workload == "korc":
        run_rank_korc(args)
    else:
        run_rank(args)
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass


if __name__ == "__main__":
    main()
