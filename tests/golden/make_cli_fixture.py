#!/usr/bin/env python3
"""tests/golden/cli_distribution_golden.npz: samples of libstdc++'s std::mt19937_64 +
std::normal_distribution in the configuration of the reference's CLI example (make_cli_fixture.cpp,
compiled here with g++), for seeds 0, 1, 7 and a shard of 1000 rays: the first 16 and the last
sample of every drawn variable."""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

with tempfile.TemporaryDirectory() as tmp:
    binary = os.path.join(tmp, "make_cli_fixture")
    subprocess.check_call(["g++", "-O1", "-o", binary, os.path.join(HERE, "make_cli_fixture.cpp")])
    text = subprocess.run([binary, "1000", "16"], capture_output=True, text=True, check=True).stdout
rows = [line.split() for line in text.splitlines()]
out = {"seeds": np.array([0, 1, 7]), "shard": np.array(1000)}
for seed in (0, 1, 7):
    for v, name in enumerate(("w", "ky", "kz", "z", "phi")):
        picked = [(int(r[2]), float.fromhex(r[3])) for r in rows if int(r[0]) == seed and int(r[1]) == v]
        out["seed%d_%s_index" % (seed, name)] = np.array([p[0] for p in picked])
        out["seed%d_%s" % (seed, name)] = np.array([p[1] for p in picked])
np.savez_compressed(os.path.join(HERE, "cli_distribution_golden.npz"), **out)
print("wrote cli_distribution_golden.npz", len(rows), "samples")
