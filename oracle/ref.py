"""Driver for the reference-backed oracle binary oracle/_ref/gf_ref.

TEST INFRASTRUCTURE ONLY.  gf_ref is built in the development container from
the reference's expression-graph headers (oracle/Makefile target `ref`); it is
used to generate the committed fixtures under tests/golden/ and, where the
binary is present, to cross-check the CPU restatement.  Product code never
imports this module.
"""
import json
import os
import subprocess
import tempfile

import numpy as np

from . import oracle as _oracle

HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(HERE, "_ref", "gf_ref")


def available():
    return os.path.exists(BINARY)


def write_tables(tables, path):
    scalars, numr, numz, numpsi, psi, te, ne, pres, fpol = _oracle.pack_tables(tables)
    with open(path, "wb") as f:
        f.write(scalars.tobytes())
        f.write(np.array([numr, numz, numpsi], dtype=np.uint64).tobytes())
        for block in (psi, te, ne, pres, fpol):
            f.write(np.ascontiguousarray(block, dtype=np.float64).tobytes())


def _write_columns(path, cols):
    cols = [np.ascontiguousarray(c, dtype=np.float64) for c in cols]
    with open(path, "wb") as f:
        f.write(np.array([cols[0].size], dtype=np.uint64).tobytes())
        for c in cols:
            f.write(c.tobytes())


def _read_columns(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(raw[:8].view(np.uint64)[0])
    data = raw[8:].view(np.float64)
    return data.reshape(-1, n)


def _parse_stderr(text):
    info = {}
    for line in text.splitlines():
        brace = line.find("{")
        if brace < 0:
            continue
        try:
            obj = json.loads(line[brace:])
        except ValueError:
            continue
        key = line[:brace].strip()
        if key:
            info[key] = obj
        else:
            info.update(obj)
    return info


class Reference:
    def __init__(self, tables):
        if not available():
            raise RuntimeError("oracle/_ref/gf_ref not built (make -C oracle ref, needs /root/reference)")
        self.tmp = tempfile.TemporaryDirectory()
        self.tables = os.path.join(self.tmp.name, "tables.bin")
        write_tables(tables, self.tables)

    def _run(self, dtype, command, in_cols, *args):
        inp = os.path.join(self.tmp.name, "in.bin")
        out = os.path.join(self.tmp.name, "out.bin")
        _write_columns(inp, in_cols)
        proc = subprocess.run([BINARY, self.tables, dtype, command, inp, out] + [str(a) for a in args],
                              check=True, stderr=subprocess.PIPE, text=True)
        return _read_columns(out), _parse_stderr(proc.stderr)

    def efit_test(self, x, y, z, dtype="f64"):
        return self._run(dtype, "efit_test", [x, y, z])

    def dispersion(self, state, dtype="f64"):
        return self._run(dtype, "dispersion", [state[k] for k in ("t", "w", "x", "y", "z", "kx", "ky", "kz")])

    def trace(self, state, dt, num_steps, save_every=0, newton_var=1, dtype="f64", dispersion="cold_plasma"):
        """Returns records[(num_saved), 9, n] (t,w,x,y,z,kx,ky,kz,residual) and info."""
        command = "trace" if dispersion == "cold_plasma" else "trace_ordinary"
        out, info = self._run(dtype, command, [state[k] for k in ("t", "w", "x", "y", "z", "kx", "ky", "kz")],
                              repr(float(dt)), num_steps, save_every, newton_var)
        return out.reshape(-1, 9, out.shape[1]), info

    def korc(self, particles, num_steps, save_every=0, dtype="f64"):
        out, info = self._run(dtype, "korc", [particles[k] for k in ("x", "y", "z", "ux", "uy", "uz", "gamma")],
                              num_steps, save_every)
        return out.reshape(-1, 7, out.shape[1]), info
