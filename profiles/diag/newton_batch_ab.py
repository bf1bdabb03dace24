"""the Newton init (converge item `loss_kernel`, workflow.hpp:179-205) at 1e7 rays with 1 ... 6 passes per
launch (`<name>_batch`, GFHIP_CONVERGE_BATCH): kernel time of the whole loop by HIP events, launches, iterations.

    python profiles/diag/newton_batch_ab.py prebuild      (CPU container: the kernels of every batch size -> kernel_cache)
    python profiles/diag/newton_batch_ab.py run [rays]     (GPU box: one subprocess per batch size, JSON lines)
"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
BATCHES = ("1", "2", "3", "4", "6")


def environment(batch):
    env = dict(os.environ)
    env["GFHIP_CONVERGE_BATCH"] = batch
    return env


def prebuild():
    from concurrent.futures import ThreadPoolExecutor
    from graph_framework_amd import build
    texts = []
    for batch in BATCHES:
        out = subprocess.run([sys.executable, "-c",
                              "import sys, json; sys.path.insert(0, %r)\n"
                              "from graph_framework_amd.backend import generate_piece_sources\n"
                              "from graph_framework_amd.xrays import workload\n"
                              "print(json.dumps(generate_piece_sources(workload('loss_kernel_kx'))))" % ROOT],
                             env=environment(batch), capture_output=True, text=True, check=True)
        texts += [tuple(piece) for piece in json.loads(out.stdout)]
    with ThreadPoolExecutor(max_workers=8) as pool:
        print("%d kernels in the cache" % len(list(pool.map(lambda piece: build.compile_source(piece[0], piece[1]), texts))))


def one():
    import numpy as np
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit
    rays = int(sys.argv[2])
    bench = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0)
    best = None
    for _ in range(3):
        solve = Rk4ColdPlasmaEfit({k: np.full(rays, v) for k, v in bench.items()})
        solve.work.context.enable_timing(True, every=1)
        start = time.perf_counter()
        solve.init("kx")
        wall = time.perf_counter() - start
        samples = solve.newton.kernel.timing_samples()
        info = solve.newton.kernel.info()
        line = {"passes_per_launch": max(int(info.converge_batch), 1), "iterations": solve.newton_iterations, "launches": len(samples),
                "kernel_ms_total": sum(samples), "kernel_ms": samples, "vgprs": int(info.vgprs), "init_wall_ms_incl_build": 1.0e3*wall,
                "kx": float(solve.host["kx"][0])}
        solve.work.context.close()
        if best is None or line["kernel_ms_total"] < best["kernel_ms_total"]:
            best = line
    print(json.dumps(best))


def run():
    rays = sys.argv[2] if len(sys.argv) > 2 else "10000000"
    for batch in BATCHES:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", rays], env=environment(batch), capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.returncode == 0 else json.dumps({"batch": batch, "error": out.stderr[-400:]}))
        sys.stdout.flush()


if __name__ == "__main__":
    {"prebuild": prebuild, "one": one, "run": run}[sys.argv[1]]()
