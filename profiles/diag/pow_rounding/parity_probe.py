import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests"))
from graph_framework_amd.xrays import Rk4ColdPlasmaEfit, cli_distribution, workload
from oracle import gfir
STATE = ("t", "w", "x", "y", "z", "kx", "ky", "kz")
n = 4096
rays = {k: np.ascontiguousarray(v[:n]) for k, v in cli_distribution(200000, seed=0).items()}
solve = Rk4ColdPlasmaEfit({k: v.copy() for k, v in rays.items()})
solve.init("kx", per_ray=True)
solve.compile()
start = solve.sync_host()
cols = [start[k].copy() for k in STATE]
item = gfir.Item(workload("solver_kernel"))
step = 0
bad = None
for chunk in range(50):
    prev = [c.copy() for c in cols]
    item.run(cols, steps=10, threads=8)
    solve.step(10)
    host = solve.sync_host()
    diff = np.zeros(n, bool)
    for k, e in zip(STATE, cols):
        diff |= ~((host[k] == e) | (np.isnan(host[k]) & np.isnan(e)))
    if diff.any():
        bad = np.flatnonzero(diff)
        print("chunk", chunk, "steps", chunk*10, "differing rays", bad)
        break
if bad is None:
    print("no difference"); sys.exit(0)
# replay the ten steps one by one for the differing rays on both sides
sub = [p[bad].copy() for p in prev]
solve2 = Rk4ColdPlasmaEfit({k: p[bad].copy() for k, p in zip(STATE, prev)})
solve2.compile()
for s in range(10):
    before = [c.copy() for c in sub]
    outs, _ = item.run(sub, steps=1)
    solve2.step(1)
    h = solve2.sync_host()
    res = solve2.residual()
    for k, e, b in zip(STATE, sub, before):
        same = (h[k] == e) | (np.isnan(h[k]) & np.isnan(e))
        if not same.all():
            print("step", s, k, "before", [float.hex(float(v)) for v in b], "oracle", [float.hex(float(v)) for v in e], "device", [float.hex(float(v)) for v in h[k]])
    print("step", s, "residual oracle", outs[0], "device", res, "flags", solve2.work.context.flags())
    if any(((h[k] != e) & ~(np.isnan(h[k]) & np.isnan(e))).any() for k, e in zip(STATE, sub)):
        print("state before:", {k: [float.hex(float(v)) for v in b] for k, b in zip(STATE, before)})
        np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "parity_before.npy"), np.stack(before))
        break
