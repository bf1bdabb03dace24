#!/usr/bin/env python3
"""Secondary measurements (not the driver's contract; bench.py is): the other items of the hot
path at BASELINE.json's sizes, each with HIP-event kernel time and its HBM-roofline fraction.

    python bench_extra.py korc_f32 | korc_f64 | loss | cli | fused
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def korc(dtype, n=10000000, steps=200):
    """BASELINE configs[4] on one GPU: xkorc push, 1e7 particles.  Algorithmic bytes per
    particle-step: (7 reads + 7 writes) x 4 B = 56 (fp32) / 112 (fp64), SURVEY §8(d)."""
    from graph_framework_amd import korc as gk
    push = gk.Korc(dict(x=1.7, y=0.0, z=0.0, ux=0.0, uy=0.99, uz=0.1, gamma=np.zeros(n)), dtype)
    push.compile()
    push.pre_run()
    for _ in range(10):
        push.run()
    push.work.context.enable_timing(True)
    push.wait()
    start = time.perf_counter()
    for _ in range(steps):
        push.run()
    push.wait()
    elapsed = time.perf_counter() - start
    ms, launches = push.step_item.kernel.timing()
    bytes_per = 56 if dtype == "f32" else 112
    achieved = n*bytes_per/(ms*1.0e-3)/1.0e9
    return {"workload": "xkorc step, %d particles, %s" % (n, dtype), "value": n*steps/elapsed,
            "unit": "particle-steps/s", "kernel_ms": ms, "launches": int(launches),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved/8000.0}}


def loss(n=10000000, per_ray=False):
    """Newton init of the benchmark at 1e7 rays: `loss_kernel_max` (the pass with the max of D^2 reduced
    inside the launch: 80 B per ray-iteration, no re-read) under the device-side converge loop,
    or (per_ray) the whole loop inside one launch."""
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    solve = Rk4ColdPlasmaEfit({k: np.full(n, v) for k, v in
                               dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0).items()})
    solve.work.context.enable_timing(True)
    solve.work.context.wait()
    start = time.perf_counter()
    solve.init("kx", per_ray=per_ray)
    elapsed = time.perf_counter() - start
    if per_ray:
        return {"workload": "Newton init, per-ray loop in one launch, %d rays fp64" % n,
                "iterations": solve.newton_iterations, "init_seconds_including_build": elapsed}
    ms, launches = solve.newton.kernel.timing()
    achieved = n*80/(ms*1.0e-3)/1.0e9
    return {"workload": "loss_kernel (Newton) %d rays fp64" % n, "iterations": solve.newton_iterations,
            "init_seconds_including_build": elapsed, "kernel_ms": ms, "launches": int(launches),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved/8000.0,
                         "note": "80 B per ray-iteration (8 reads + 2 writes); the max is reduced inside the launch"}}


def solver_f32(n=1000000, steps=200):
    """The same RK4 item in fp32 (the reference benchmarks float as well, xrays_bench.cpp:124-135;
    its published A100 figure is fp32).  64 B per ray-step."""
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    solve = Rk4ColdPlasmaEfit({k: np.full(n, v) for k, v in
                               dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0).items()},
                              dtype="f32")
    solve.init("kx")
    solve.compile()
    for _ in range(10):
        solve.step()
    solve.work.context.enable_timing(True)
    solve.work.wait()
    start = time.perf_counter()
    for _ in range(steps):
        solve.step()
    solve.work.wait()
    elapsed = time.perf_counter() - start
    ms, launches = solve.solver.kernel.timing()
    info = solve.solver.kernel.info()
    achieved = n*64/(ms*1.0e-3)/1.0e9
    return {"workload": "solver_kernel fp32, %d rays" % n, "value": n*steps/elapsed, "unit": "ray-steps/s",
            "kernel_ms": ms, "vgprs": int(info.vgprs), "scratch_bytes": int(info.scratch_bytes),
            "newton_iterations": solve.newton_iterations, "flags": solve.work.context.flags(),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved/8000.0}}


def stream(dtype, n=10000000, steps=50):
    """Calibration: a trivial item (3 arrays read+written, 1 written; workflow_test-style setters)
    through the same lowering, i.e. the HBM rate the launch geometry itself reaches."""
    import graph_framework_amd as gfa
    np_dtype = np.float32 if dtype == "f32" else np.float64
    ctx = gfa.Context(0)
    kernel = ctx.add_kernel(os.path.join(ROOT, "graph_framework_amd", "workloads", "misc_alias_kernel_%s.gfir" % dtype), n)
    ctx.compile()
    kernel.create_kernel_call(["a", "b", "c"], ["s"], [np.full(n, v, np_dtype) for v in (0.5, 0.25, 0.125)])
    for _ in range(5):
        kernel.run(1)
    ctx.enable_timing(True)
    ctx.wait()
    for _ in range(steps):
        kernel.run(1)
    ms, launches = kernel.timing()
    achieved = n*np_dtype().itemsize*7/(ms*1.0e-3)/1.0e9
    return {"workload": "stream calibration (3 in/out + 1 out), %d elements %s" % (n, dtype), "kernel_ms": ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved/8000.0}}


def stream7(dtype, n=10000000, steps=50):
    """Calibration with the xkorc push's own traffic pattern: 7 arrays read and written in place
    by a trivial item (v <- v*1.0000001 + 0.5), built here as GFIR."""
    import struct
    import graph_framework_amd as gfa
    np_dtype = np.float32 if dtype == "f32" else np.float64
    code = []
    for i in range(7):
        code.append((1, i, 0xFFFFFFFF, 0xFFFFFFFF, 0, (0.0,)*4))                 # INPUT i
    code.append((0, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0, (float(np_dtype(1.0000001)), 0.0, 0.0, 0.0)))
    code.append((0, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0, (0.5, 0.0, 0.0, 0.0)))
    for i in range(7):
        code.append((6, i, 7, 8, 0, (0.0,)*4))                                   # FMA(v, a, b)
    name = b"stream7\0"
    blob = struct.pack("<8s8I", b"GFIR0001", 1 if dtype == "f64" else 0, 7, 0, 7, 0, len(code), len(name), 0) + name
    for i in range(7):
        blob += struct.pack("<I", 4) + ("v%d" % i).encode() + b"\0\0"
    for op, a, b, c, aux, imm in code:
        blob += struct.pack("<6I4d", op, a, b, c, aux, 0, *imm)
    for i in range(7):
        blob += struct.pack("<II", 9 + i, i)
    ctx = gfa.Context(0)
    kernel = ctx.add_kernel(blob, n)
    ctx.compile()
    kernel.create_kernel_call(["v%d" % i for i in range(7)], [], [np.full(n, 0.25, np_dtype) for _ in range(7)])
    for _ in range(5):
        kernel.run(1)
    ctx.enable_timing(True)
    ctx.wait()
    for _ in range(steps):
        kernel.run(1)
    ms, launches = kernel.timing()
    achieved = n*np_dtype().itemsize*14/(ms*1.0e-3)/1.0e9
    return {"workload": "stream calibration (7 arrays in/out, the push's pattern), %d elements %s" % (n, dtype),
            "kernel_ms": ms, "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                                          "frac": achieved/8000.0}}


def trajectory(n=1000000, steps=400, every=100):
    """the step loop with solver_interface::write_step every `every` steps
    (snapshot D2D on the compute stream, D2H on a second stream, HDF5 append on a writer thread)."""
    import tempfile
    import torch
    from graph_framework_amd.output import TrajectoryWriter
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    state = {k: np.full(n, v) for k, v in dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0).items()}
    solve = Rk4ColdPlasmaEfit(state, device_state=True)
    solve.init("kx")
    solve.compile()
    for _ in range(10):
        solve.step()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        solve.step()
    torch.cuda.synchronize()
    plain = time.perf_counter() - start
    with tempfile.TemporaryDirectory() as directory:
        path = os.path.join(directory, "result0.nc")
        writer = TrajectoryWriter(solve, path)
        start = time.perf_counter()
        for step in range(steps):
            solve.step()
            if (step + 1) % every == 0:
                writer.write_step()
        torch.cuda.synchronize()
        loop = time.perf_counter() - start
        writer.close()
        total = time.perf_counter() - start
        size = os.path.getsize(path)
    return {"workload": "solver_kernel 1e6 rays with write_step every %d steps (%d records)" % (every, steps//every),
            "value": n*steps/loop, "unit": "ray-steps/s", "without_output": n*steps/plain,
            "seconds_until_file_closed": total, "file_bytes": size,
            "note": "write_step joins the previous writer thread first (solver.hpp:419), so the loop is throttled "
                    "only if a record takes longer to write than `every` steps take to compute"}


def absorption(n=1000000, steps=20):
    """The absorption pass's kernel (absorption::weak_damping, complex<double> + SAFE_MATH + erfi) on n rays
    spread over the golden trajectories' records (inside, at and outside the resonance).  Algorithmic bytes
    per ray: 9 complex inputs read + kamp written = 160 B."""
    from graph_framework_amd import Context
    from graph_framework_amd.xrays import workload
    golden = np.load(os.path.join(ROOT, "tests", "golden", "absorption_golden.npz"))
    records = golden["records"]
    flat = {k: records[:, i, :].reshape(-1) for i, k in enumerate(("t", "w", "x", "y", "z", "kx", "ky", "kz"))}
    pick = np.random.default_rng(0).integers(0, flat["t"].size, n)
    keys = ("kamp", "kx", "ky", "kz", "x", "y", "z", "t", "w")
    columns = [np.zeros(n, dtype=np.complex128)] + [flat[k][pick].astype(np.complex128) for k in keys[1:]]
    context = Context(0)
    kernel = context.add_kernel(workload("weak_damping_kimg_kernel", "c64"), n)
    context.compile()
    kernel.create_kernel_call(keys, [], columns)
    kernel.run(1)
    context.enable_timing(True)
    context.wait()
    start = time.perf_counter()
    for _ in range(steps):
        kernel.run(1)
    context.wait()
    elapsed = time.perf_counter() - start
    ms, launches = kernel.timing()
    info = kernel.info()
    achieved = n*160/(ms*1.0e-3)/1.0e9
    return {"workload": "weak_damping_kimg_kernel, %d rays, complex<double> SAFE_MATH" % n, "value": n*steps/elapsed,
            "unit": "rays/s", "kernel_ms": ms, "launches": int(launches), "vgprs": info.vgprs, "nodes": info.num_instructions,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved/8000.0,
                         "note": "160 B per ray; 423 complex nodes with one erfi: arithmetic bound"}}


def fused(n=1000000, steps=200, per_launch=10):
    """RK4 with `per_launch` steps fused into one launch (xrays_bench's SUB_STEPS = 10)."""
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    solve = Rk4ColdPlasmaEfit({k: np.full(n, v) for k, v in
                               dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0).items()})
    solve.init("kx")
    solve.compile()
    solve.step(per_launch)
    solve.work.wait()
    start = time.perf_counter()
    for _ in range(steps//per_launch):
        solve.step(per_launch)
    solve.work.wait()
    elapsed = time.perf_counter() - start
    return {"workload": "solver_kernel, %d steps per launch" % per_launch, "value": n*steps/elapsed, "unit": "ray-steps/s"}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "korc_f32"
    if what.startswith("korc"):
        out = korc(what.split("_")[1])
    elif what == "loss":
        out = loss()
    elif what == "loss_per_ray":
        out = loss(per_ray=True)
    elif what.startswith("stream7"):
        out = stream7(what.split("_")[1])
    elif what.startswith("stream"):
        out = stream(what.split("_")[1])
    elif what == "solver_f32":
        out = solver_f32()
    elif what == "trajectory":
        out = trajectory()
    elif what == "fused":
        out = fused()
    elif what == "absorption":
        out = absorption()
    else:
        raise SystemExit("unknown workload")
    print(json.dumps(out))
