"""The two lowerings of the RK4 item against each other where the oracle cannot follow: an incoherent CLI beam of
`rays` rays through `steps` steps with the assembly body (default) and with the body hipcc compiles (GFHIP_ASM=0) —
every state array must come out bit for bit the same, NaNs of blown-up rays and lanes redone by the IEEE path included.

    python profiles/diag/asm/compare_bodies.py [rays=1000000] [steps=300] [dispersion=cold_plasma|ordinary_wave]          (GPU box)
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, STATE, cli_distribution
    rays, steps = int(sys.argv[2]), int(sys.argv[3])
    solve = Rk4ColdPlasmaEfit(cli_distribution(rays, seed=5), dispersion=sys.argv[5])
    residual = solve.init("kx")
    solve.compile()
    for _ in range(steps):
        solve.step()
    host = solve.sync_host()
    np.savez(sys.argv[4], **{k: host[k] for k in STATE})
    info = solve.solver.kernel.info()
    print(json.dumps({"segments": int(info.segments), "vgprs": int(info.vgprs), "flags": int(solve.work.context.flags()),
                      "newton_residual": float(residual), "non_finite_rays": int((~np.isfinite(host["x"])).sum())}))


def main():
    import numpy as np
    rays = sys.argv[1] if len(sys.argv) > 1 else "1000000"
    steps = sys.argv[2] if len(sys.argv) > 2 else "300"
    dispersion = sys.argv[3] if len(sys.argv) > 3 else "cold_plasma"
    results = {}
    with tempfile.TemporaryDirectory() as directory:
        for body, value in (("assembly", "1"), ("compiled", "0")):
            path = os.path.join(directory, body + ".npz")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", rays, steps, path, dispersion],
                                 env=dict(os.environ, GFHIP_ASM=value), capture_output=True, text=True, check=True)
            results[body] = (json.loads(out.stdout.strip().splitlines()[-1]), dict(np.load(path)))
    (first, a), (second, b) = results["assembly"], results["compiled"]
    different = {k: int((a[k].view(np.uint64) != b[k].view(np.uint64)).sum()) for k in a}
    print(json.dumps({"rays": int(rays), "steps": int(steps), "dispersion": dispersion, "assembly": first, "compiled": second,
                      "elements_that_differ": different, "bit_identical": not any(different.values())}))
    sys.exit(0 if not any(different.values()) else 1)


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
