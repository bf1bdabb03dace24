import os, sys, subprocess
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "..", ".."))
STATE = ("t", "w", "x", "y", "z", "kx", "ky", "kz")
if len(sys.argv) > 1:
    from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, workload
    from oracle import gfir
    before = np.load(os.path.join(here, "parity_before.npy"))
    sub = [before[i].copy() for i in range(8)]
    item = gfir.Item(workload("solver_kernel"))
    item.run(sub, steps=1)
    solve = Rk4ColdPlasmaEfit({k: before[i].copy() for i, k in enumerate(STATE)})
    solve.compile()
    solve.step(1)
    h = solve.sync_host()
    print(sys.argv[1], {k: int(h[k].view(np.int64)[0] - e.view(np.int64)[0]) for k, e in zip(STATE, sub)}, "flags", solve.work.context.flags(), flush=True)
    sys.exit(0)
for env in [{}, {"GFHIP_DIVISION": "ieee"}, {"GFHIP_DIVISION": "checked"}, {"GFHIP_POW": "libm"}, {"GFHIP_COMPACT_TABLES": "0"},
            {"GFHIP_PARK": "0"}, {"GFHIP_SCHEDULE": "source"}]:
    e = dict(os.environ); e.update(env); e["GFHIP_CACHE_DIR"] = "/tmp/gfcache_" + "_".join(env.values() or ["default"])
    subprocess.run([sys.executable, __file__, str(env)], env=e)
