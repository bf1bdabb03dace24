"""One random item through the assembly body against the oracle: python try_item.py seed inputs nodes  (environment selects the pool etc.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gfir_random
from oracle import gfir
from graph_framework_amd import Context
seed, inputs, nodes = (int(x) for x in sys.argv[1:4])
rays = 1777
blob, _ = gfir_random.random_item(seed, "f64", inputs, nodes, 3, 3)
item = gfir.Item(blob)
rng = np.random.default_rng(2000 + seed)
initial = [rng.uniform(-1.0, 1.0, rays) for _ in range(inputs)]
context = Context(0)
kernel = context.add_kernel(blob, rays)
context.compile()
in_keys = ["in%d" % i for i in range(inputs)]; out_keys = ["out%d" % i for i in range(3)]
kernel.create_kernel_call(in_keys, out_keys, initial)
expected = [c.copy() for c in initial]
expected_out, _ = item.run(expected, steps=1)
kernel.run(1); context.wait()
bad = []
for key, want in zip(in_keys + out_keys, expected + expected_out):
    got = context.copy_to_host(key, np.empty(rays))
    wrong = np.flatnonzero(got != want)
    if wrong.size: bad.append((key, int(wrong.size), float(np.max(np.abs(got - want)))))
print(" ".join("%s=%s" % (k, os.environ.get(k)) for k in ("GFHIP_ASM_POOL_LO", "GFHIP_LDS_BUDGET", "GFHIP_ASM_LOAD_AHEAD", "GFHIP_ASM_RELOAD_AHEAD")), "flags", context.flags(), "mismatches", bad)
