"""Which segment of a split item differs from the oracle (GPU)?  python profiles/diag/fission_check.py <max_nodes>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import fission_prototype as fp  # noqa: E402
from graph_framework_amd import Context  # noqa: E402
from graph_framework_amd.xrays import workload  # noqa: E402
from oracle import gfir  # noqa: E402

blob = open(workload("solver_kernel"), "rb").read()
item, cuts = fp.bisect(fp.Item(blob), int(sys.argv[1]))
pieces, scratch = fp.split(item, cuts)
n = 4096
keys = ["t", "w", "x", "y", "z", "kx", "ky", "kz"]
state = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-500.00000357884727, ky=0.0, kz=0.0)
rng = np.random.default_rng(1)
columns = [np.full(n, state[k]) for k in keys]
columns[3] += rng.normal(0, 0.01, n)
columns[4] += rng.normal(0, 0.01, n)
columns[6] += rng.normal(0, 5.0, n)
columns[7] += rng.normal(0, 5.0, n)
bufs = {k: c.copy() for k, c in zip(keys, columns)}
for j, p in enumerate(pieces):
    os.makedirs(os.path.join(ROOT, "gpurun_out", "fission"), exist_ok=True)
    in_keys = keys + ["scratch%d" % c for c in p["cut_in"]]
    out_keys = ["scratch%d" % i if kind == "scratch" else "residual" for kind, i in p["out_ids"]]
    ins = [bufs[k].copy() for k in in_keys]
    expected = [c.copy() for c in ins]
    outs, _ = gfir.Item(p["blob"]).run(expected, steps=1)
    context = Context(0)
    kernel = context.add_kernel(p["blob"], n)
    context.compile()
    kernel.create_kernel_call(in_keys, out_keys, ins)
    kernel.run(1)
    context.wait()
    flags = context.flags()
    bad = []
    for key, want in list(zip(out_keys, outs)) + list(zip(in_keys, expected)):
        got = context.copy_to_host(key, np.empty(n))
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        if not same.all():
            lanes = np.flatnonzero(~same)
            bad.append((key, lanes.size, int(lanes[0]), float(got[lanes[0]]).hex(), float(want[lanes[0]]).hex()))
    info = kernel.info()
    context.close()
    print("segment", j, "nodes", p["nodes"], "regs", info.vgprs, "flags", flags, "mismatches", bad[:4], flush=True)
    if bad:
        with open(os.path.join(ROOT, "gpurun_out", "fission", "bad_segment_%d.gfir" % j), "wb") as f:
            f.write(p["blob"])
        np.save(os.path.join(ROOT, "gpurun_out", "fission", "bad_segment_%d_inputs.npy" % j), np.stack(ins))
    for key, want in zip(out_keys, outs):
        bufs[key] = want
    for key, want in zip(in_keys, expected):
        bufs[key] = want
