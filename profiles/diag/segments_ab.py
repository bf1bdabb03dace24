"""the RK4 item cut into segments whose out-of-window lanes are redone by a launch of their own
(no IEEE function inside the kernels), with and without two waves per SIMD, against the one-kernel form.

    python profiles/diag/segments_ab.py prebuild          (CPU container: hipcc the kernels of every configuration
                                                           into graph_framework_amd/kernel_cache, which travels)
    python profiles/diag/segments_ab.py run [rays] [steps] (GPU box: one subprocess per configuration, JSON lines)
    python profiles/diag/segments_ab.py one               (internal: time the configuration of the environment)

Each configuration is a set of environment variables read by the lowering (options.hpp).  `run` checks every
configuration bit for bit against the one-kernel form on 20 000 incoherent rays x 5 steps before timing it.
"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CONFIGURATIONS = [
    ("one kernel", {}),
    ("3 segments", {"GFHIP_SEGMENTS": "3"}),
    ("3 segments, 2 waves/SIMD", {"GFHIP_SEGMENTS": "3", "GFHIP_WAVES_PER_SIMD": "2"}),
    ("4 segments", {"GFHIP_SEGMENTS": "4"}),
    ("4 segments, 2 waves/SIMD", {"GFHIP_SEGMENTS": "4", "GFHIP_WAVES_PER_SIMD": "2"}),
    ("6 segments, 2 waves/SIMD", {"GFHIP_SEGMENTS": "6", "GFHIP_WAVES_PER_SIMD": "2"}),
    ("8 segments, 2 waves/SIMD", {"GFHIP_SEGMENTS": "8", "GFHIP_WAVES_PER_SIMD": "2"}),
]
KNOBS = ("GFHIP_SEGMENTS", "GFHIP_WAVES_PER_SIMD", "GFHIP_HANDOVER_BYTES")


def environment(overrides):
    env = {k: v for k, v in os.environ.items() if k not in KNOBS}
    env.update(overrides)
    return env


def prebuild():
    from concurrent.futures import ThreadPoolExecutor
    texts = []
    for _, overrides in CONFIGURATIONS:
        out = subprocess.run([sys.executable, "-c",
                              "import sys, json; sys.path.insert(0, %r)\n"
                              "from graph_framework_amd.backend import generate_piece_sources\n"
                              "from graph_framework_amd.xrays import workload\n"
                              "print(json.dumps(generate_piece_sources(workload('solver_kernel'))))" % ROOT],
                             env=environment(overrides), capture_output=True, text=True, check=True)
        texts += [tuple(piece) for piece in json.loads(out.stdout)]
    from graph_framework_amd import build
    with ThreadPoolExecutor(max_workers=8) as pool:
        built = list(pool.map(lambda piece: build.compile_source(piece[0], piece[1]), texts))
    print("%d kernels in the cache" % len(built))


def one():
    import numpy as np
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, STATE, cli_distribution
    rays, steps = int(sys.argv[2]), int(sys.argv[3])
#  parity first: 20 000 incoherent rays, 5 steps, against the records of the one-kernel form
    check = cli_distribution(20000, seed=3)
    solve = Rk4ColdPlasmaEfit(check)
    solve.init("kx")
    solve.compile()
    for _ in range(5):
        solve.step()
    host = solve.sync_host()
    flags = solve.work.context.flags()
    digest = {k: host[k].tobytes().hex()[:0] for k in STATE}
    np.savez(sys.argv[4], **{k: host[k] for k in STATE})
    info = solve.solver.kernel.info()
    solve.work.context.close()

    bench = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-600.0, ky=0.0, kz=0.0)
    solve = Rk4ColdPlasmaEfit({k: np.full(rays, v) for k, v in bench.items()})
    solve.init("kx")
    solve.compile()
    start = time.perf_counter()
    while time.perf_counter() - start < 0.4:
        for _ in range(10):
            solve.step()
        solve.work.wait()
    solve.work.context.enable_timing(True, every=4)
    t0 = time.perf_counter()
    for _ in range(steps):
        solve.step()
    solve.work.wait()
    elapsed = time.perf_counter() - t0
    ms, launches = solve.solver.kernel.timing()
    print(json.dumps({"segments": int(info.segments), "vgprs": int(info.vgprs), "scratch": int(info.scratch_bytes),
                      "ms_per_step": 1.0e3*elapsed/steps, "event_ms": ms, "ray_steps_per_s": rays*steps/elapsed, "flags_on_cli_beam": flags}))
    del digest


def run():
    import numpy as np
    import tempfile
    rays = sys.argv[2] if len(sys.argv) > 2 else "10000000"
    steps = sys.argv[3] if len(sys.argv) > 3 else "100"
    reference = None
    with tempfile.TemporaryDirectory() as tmp:
        for name, overrides in CONFIGURATIONS:
            state = os.path.join(tmp, "state.npz")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", rays, steps, state], env=environment(overrides),
                                 capture_output=True, text=True)
            if out.returncode:
                print(json.dumps({"configuration": name, "error": out.stderr[-400:]}))
                continue
            line = json.loads(out.stdout.strip().splitlines()[-1])
            got = dict(np.load(state))
            if reference is None:
                reference = got
            line["bit_identical_to_one_kernel"] = all(np.array_equal(got[k], reference[k], equal_nan=True) for k in reference)
            line["configuration"] = name
            print(json.dumps(line))
            sys.stdout.flush()


if __name__ == "__main__":
    {"prebuild": prebuild, "one": one, "run": run}[sys.argv[1]]()
