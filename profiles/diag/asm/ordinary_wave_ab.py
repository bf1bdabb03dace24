import sys, time, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, cli_distribution
rays = 10000000
solve = Rk4ColdPlasmaEfit(cli_distribution(rays, seed=1), dispersion="ordinary_wave")
solve.init("kx"); solve.compile()
for _ in range(20): solve.step()
solve.work.wait()
t = time.perf_counter()
for _ in range(100): solve.step()
solve.work.wait()
e = time.perf_counter() - t
info = solve.solver.kernel.info()
print(json.dumps({"asm": os.environ.get("GFHIP_ASM", "1"), "ms_per_step": 1e3*e/100, "segments": int(info.segments), "vgprs": int(info.vgprs)}))
