#!/usr/bin/env python3
"""Generates tests/golden/physics_golden.json and the physics_* work items.

Runs oracle/_ref/gf_ref_physics (the reference's expression-graph layer compiled from
/root/reference, `make -C oracle ref`): it replays graph_tests/solver_test.cpp and
graph_tests/physics_test.cpp on the reference graphs, checks each test's own assertion,
prints the golden states and exports every work item as GFIR.  Only runs in the
development container (the reference does not travel); the outputs are committed.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BINARY = os.path.join(ROOT, "oracle", "_ref", "gf_ref_physics")
WORKLOADS = os.path.join(ROOT, "graph_framework_amd", "workloads")


def main():
    if not os.path.exists(BINARY):
        sys.exit("build it first: make -C oracle ref (needs /root/reference)")
    proc = subprocess.run([BINARY, WORKLOADS], check=True, stdout=subprocess.PIPE, text=True)
    golden = json.loads(proc.stdout)
    failed = [k for k, v in golden.items() if not v["reference_assertion_holds"]]
    if failed:
        sys.exit("reference assertions fail on the reference graphs: %s" % failed)
    with open(os.path.join(HERE, "physics_golden.json"), "w") as f:
        json.dump(golden, f, indent=1)
        f.write("\n")
    print("wrote physics_golden.json: %d scenarios" % len(golden))


if __name__ == "__main__":
    main()
