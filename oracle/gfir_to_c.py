"""GFIR -> plain C translation of the CPU ORACLE, for timing (test infrastructure only).

oracle/gfir_interp.c interprets a work item record by record, which costs ~10x what the
reference's cpu_context pays for the same arithmetic: the reference JIT-compiles one C++
statement per node (cpu_context.hpp:428-501) and runs a serial `for` over its shard
(cpu_context.hpp:487).  This module emits exactly that kernel as C — one statement per GFIR
record, same operations, `-ffp-contract=off`, no fast-math — compiles it with gcc and loads
it, so that bench.py's cpu_baseline times compiled code like the reference's, not an
interpreter.  tests/test_oracle.py checks the compiled kernel bit-for-bit against the
interpreter.
"""
import ctypes
import hashlib
import os
import struct
import subprocess
import tempfile

import numpy as np

OPS = dict(CONST=0, INPUT=1, ADD=2, SUB=3, MUL=4, DIV=5, FMA=6, SQRT=7, POWI=8, POW=9,
           SIN=10, COS=11, ATAN2=12, EXP=13, LOG=14, GATHER1=15, GATHER2=16)


def parse(data):
    magic, dtype, ni, no, ns, nt, nins, nb, _ = struct.unpack_from("<8s8I", data, 0)
    assert magic == b"GFIR0001"
    pos = 40
    name = data[pos:pos + nb].split(b"\0")[0].decode()
    pos += nb
    for _ in range(ni):
        (n,) = struct.unpack_from("<I", data, pos)
        pos += 4 + n
    tables = []
    for _ in range(nt):
        r, c = struct.unpack_from("<II", data, pos)
        pos += 8
        tables.append((r, c, np.frombuffer(data, dtype="<f8", count=r*c, offset=pos).copy()))
        pos += 8*r*c
    ins_dtype = np.dtype([("op", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("aux", "<u4"),
                          ("res", "<u4"), ("imm", "<f8", 4)])
    ins = np.frombuffer(data, dtype=ins_dtype, count=nins, offset=pos)
    pos += nins*ins_dtype.itemsize
    outputs = np.frombuffer(data, dtype="<u4", count=no, offset=pos)
    pos += 4*no
    setters = np.frombuffer(data, dtype="<u4", count=2*ns, offset=pos).reshape(-1, 2)
    return dict(name=name, f64=(dtype == 1), num_inputs=ni, tables=tables, ins=ins, outputs=outputs, setters=setters)


def to_c(item):
    f64 = item["f64"]
    real = "double" if f64 else "float"
    sfx = "" if f64 else "f"

    def lit(v):
        if f64:
            return float(v).hex()
        return float(np.float32(v)).hex() + "f"

    lines = ["#include <math.h>", "#include <stddef.h>", "typedef %s real;" % real]
    for t, (rows, cols, values) in enumerate(item["tables"]):
        body = ",".join(lit(v) for v in values)
        lines.append("static const real T%d[%d] = {%s};" % (t, rows*cols, body))
    lines.append("static inline size_t idx(const real x, const real scale, const real offset, const real last) {")
    lines.append("    return (size_t)fmin%s(fmax%s((x - offset)/scale, (real)0), last);" % (sfx, sfx))
    lines.append("}")
    lines.append("void kernel(real **columns, real **outs, const size_t begin, const size_t end) {")
    lines.append("    for (size_t e = begin; e < end; e++) {")
    for i, c in enumerate(item["ins"]):
        op, a, b, cc, aux, imm = int(c["op"]), int(c["a"]), int(c["b"]), int(c["c"]), int(c["aux"]), c["imm"]
        if op == OPS["CONST"]:
            e = lit(imm[0])
        elif op == OPS["INPUT"]:
            e = "columns[%d][e]" % a
        elif op == OPS["ADD"]:
            e = "r%d + r%d" % (a, b)
        elif op == OPS["SUB"]:
            e = "r%d - r%d" % (a, b)
        elif op == OPS["MUL"]:
            e = "r%d*r%d" % (a, b)
        elif op == OPS["DIV"]:
            e = "r%d/r%d" % (a, b)
        elif op == OPS["FMA"]:
            e = "fma%s(r%d, r%d, r%d)" % (sfx, a, b, cc)
        elif op == OPS["SQRT"]:
            e = "sqrt%s(r%d)" % (sfx, a)
        elif op == OPS["POWI"]:
            e = "*".join(["r%d" % a]*aux)
        elif op == OPS["POW"]:
            e = "pow%s(r%d, r%d)" % (sfx, a, b)
        elif op == OPS["SIN"]:
            e = "sin%s(r%d)" % (sfx, a)
        elif op == OPS["COS"]:
            e = "cos%s(r%d)" % (sfx, a)
        elif op == OPS["ATAN2"]:
            e = "atan2%s(r%d, r%d)" % (sfx, b, a)
        elif op == OPS["EXP"]:
            e = "exp%s(r%d)" % (sfx, a)
        elif op == OPS["LOG"]:
            e = "log%s(r%d)" % (sfx, a)
        elif op == OPS["GATHER1"]:
            rows, cols, _ = item["tables"][aux]
            e = "T%d[idx(r%d, %s, %s, %s)]" % (aux, a, lit(imm[0]), lit(imm[1]), lit(cols - 1))
        elif op == OPS["GATHER2"]:
            rows, cols, _ = item["tables"][aux]
            e = "T%d[idx(r%d, %s, %s, %s)*%d + idx(r%d, %s, %s, %s)]" % (
                aux, a, lit(imm[0]), lit(imm[1]), lit(rows - 1), cols, b, lit(imm[2]), lit(imm[3]), lit(cols - 1))
        else:
            raise ValueError("unsupported op %d" % op)
        lines.append("        const real r%d = %s;" % (i, e))
    for value, target in item["setters"]:
        lines.append("        columns[%d][e] = r%d;" % (target, value))
    for o, value in enumerate(item["outputs"]):
        lines.append("        outs[%d][e] = r%d;" % (o, value))
    lines.append("    }")
    lines.append("}")
    return "\n".join(lines) + "\n"


class CompiledItem:
    """The work item as a gcc-compiled shared object; same call shape as gfir.Item.run."""

    def __init__(self, source, cache_dir=None):
        if not isinstance(source, (bytes, bytearray)):
            with open(source, "rb") as f:
                source = f.read()
        self.item = parse(bytes(source))
        self.np_dtype = np.float64 if self.item["f64"] else np.float32
        text = to_c(self.item)
        digest = hashlib.sha1(text.encode()).hexdigest()[:16]
        cache_dir = cache_dir or os.path.join(tempfile.gettempdir(), "gfir_to_c")
        os.makedirs(cache_dir, exist_ok=True)
        lib_path = os.path.join(cache_dir, "kernel_%s.so" % digest)
        if not os.path.exists(lib_path):
            c_path = os.path.join(cache_dir, "kernel_%s.c" % digest)
            with open(c_path, "w") as f:
                f.write(text)
            subprocess.check_call(["gcc", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared",
                                   "-o", lib_path, c_path, "-lm"])
        self.lib = ctypes.CDLL(lib_path)
        self.lib.kernel.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]

    def run(self, columns, steps=1, threads=1):
        """In-place passes over SoA columns, contiguous shards per thread (xrays_bench.cpp:38-51).
        Returns (outputs, wall seconds)."""
        import threading
        import time
        n = columns[0].size
        outs = [np.empty(n, dtype=self.np_dtype) for _ in range(len(self.item["outputs"]))]
        col_ptrs = (ctypes.c_void_p*max(len(columns), 1))(*[c.ctypes.data for c in columns])
        out_ptrs = (ctypes.c_void_p*max(len(outs), 1))(*[o.ctypes.data for o in outs])
        threads = max(1, min(threads, n))
        batch, extra = n//threads, n % threads

        def work(index):
            begin = index*batch + min(index, extra)
            end = begin + batch + (1 if extra > index else 0)
            for _ in range(steps):
                self.lib.kernel(col_ptrs, out_ptrs, begin, end)      # ctypes releases the GIL

        start = time.perf_counter()
        pool = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in pool:
            t.start()
        for t in pool:
            t.join()
        return outs, time.perf_counter() - start
