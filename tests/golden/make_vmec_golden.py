#!/usr/bin/env python3
"""Golden values of the VMEC equilibrium (equilibrium.hpp:1868-2650; the reference-held spline file
graph_tests/vmec.nc: 86 Fourier modes, 198 flux surfaces) from the reference's own graph layer
(oracle/_ref/gf_ref_vmec), and the field item as GFIR:

    graph_framework_amd/workloads/vmec_field_kernel_f64.gfir   inputs (s, u, v); outputs B (3), x, y, z, ne, te
    tests/golden/vmec_golden.npz
    tests/golden/vmec_tables.npz       the spline tables of graph_tests/vmec.nc as plain arrays (a data fixture, like
                                       efit_tables.npz): what the independent numpy evaluation of tests/test_oracle.py reads

Only the field evaluation: the ray equations on this equilibrium are not part of the fixtures, because
the reference's reducer does not get through d(cold_plasma D)/ds on the VMEC graph.

    python tests/golden/make_vmec_golden.py        (development container, needs /root/reference)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_fixtures import H5File  # noqa: E402
from oracle import ref  # noqa: E402

BINARY = os.path.join(ROOT, "oracle", "_ref", "gf_ref_vmec")


def write_vmec(path, source="/root/reference/graph_tests/vmec.nc"):
    f = H5File(source)
    scalars = [float(f.read(k)) for k in ("sminh", "sminf", "ds", "dphi", "signj")]
    numsf = f.read("chi_c0").size
    nummn, numsh = f.read("lmns_c0").shape
    with open(path, "wb") as out:
        out.write(np.array(scalars).tobytes())
        out.write(np.array([numsf, numsh, nummn], dtype=np.uint64).tobytes())
        for k in range(4):
            out.write(f.read("chi_c%d" % k).astype("<f8").tobytes())
        for quantity in ("rmnc", "zmns", "lmns"):
            for k in range(4):
                out.write(np.ascontiguousarray(f.read("%s_c%d" % (quantity, k)), dtype="<f8").tobytes())
        out.write(f.read("xm").astype("<f8").tobytes())
        out.write(f.read("xn").astype("<f8").tobytes())
    f.close()
    return nummn


def write_tables_npz(path, source="/root/reference/graph_tests/vmec.nc"):
    f = H5File(source)
    out = {k: float(f.read(k)) for k in ("sminh", "sminf", "ds", "dphi", "signj")}
    for k in range(4):
        out["chi_c%d" % k] = f.read("chi_c%d" % k)
        for quantity in ("rmnc", "zmns", "lmns"):
            out["%s_c%d" % (quantity, k)] = f.read("%s_c%d" % (quantity, k))
    out["xm"], out["xn"] = f.read("xm"), f.read("xn")
    f.close()
    np.savez_compressed(path, **out)


def main():
    write_tables_npz(os.path.join(HERE, "vmec_tables.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        tables = os.path.join(tmp, "vmec.bin")
        modes = write_vmec(tables)
        rng = np.random.default_rng(7)
        n = 64
        s, u, v = rng.uniform(0.02, 0.98, n), rng.uniform(0.0, 2.0*np.pi, n), rng.uniform(0.0, 2.0*np.pi, n)
        s[:4] = [0.5, 0.0100001, 0.9899, 0.25]                  # near both ends of the spline range
        u[0] = v[0] = 0.0
        ref._write_columns(os.path.join(tmp, "in.bin"), [s, u, v])
        gfir = os.path.join(ROOT, "graph_framework_amd", "workloads", "vmec_field_kernel_f64.gfir")
        proc = subprocess.run([BINARY, tables, "field", str(modes), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin"), gfir],
                              check=True, stderr=subprocess.PIPE, text=True)
        print(proc.stderr.strip().splitlines()[-2])
        out = ref._read_columns(os.path.join(tmp, "out.bin"))
    np.savez_compressed(os.path.join(HERE, "vmec_golden.npz"), inputs=np.stack([s, u, v]), outputs=out)
    print("B at the first point", out[:3, 0], "position", out[3:6, 0], "ne, te", out[6:, 0])


if __name__ == "__main__":
    main()
