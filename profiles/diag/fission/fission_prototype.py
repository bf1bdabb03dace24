"""Kernel fission of a large work item at low-liveness cut points (prototype, GPU).

The RK4 item is one 3.9 k-node DAG that needs ~500 registers (one wave per SIMD, ~620 AGPR copies per
ray-step).  Any topological order computes the same IEEE values, and so does any split of the DAG into
consecutive segments that hand their live values over through memory.  This script splits a GFIR item
into `segments` sub-items at the positions of the emission order where the fewest values are live,
runs them one after the other through the C ABI, checks the result bit for bit against the unsplit
item, and prints registers and time per segment.

    python profiles/diag/fission_prototype.py [segments ...]
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)

CONST, INPUT, FMA = 0, 1, 6
ONE_OPERAND = {7, 8, 10, 11, 13, 14, 15, 17, 19, 20}
NONE = 0xFFFFFFFF


class Item:
    def __init__(self, blob):
        magic, self.dtype, ni, no, ns, nt, nc, name_bytes, self.flags = struct.unpack_from("<8s8I", blob, 0)
        pos = 40
        self.name = blob[pos:pos + name_bytes].split(b"\0")[0].decode()
        pos += name_bytes
        self.symbols = []
        for _ in range(ni):
            n, = struct.unpack_from("<I", blob, pos)
            self.symbols.append(blob[pos + 4:pos + 4 + n])
            pos += 4 + n
        self.tables = []
        for _ in range(nt):
            rows, cols = struct.unpack_from("<II", blob, pos)
            pos += 8
            self.tables.append((rows, cols, blob[pos:pos + rows*cols*8]))
            pos += rows*cols*8
        self.code = []
        for _ in range(nc):
            self.code.append(list(struct.unpack_from("<6I4d", blob, pos)))
            pos += 56
        self.outputs = list(struct.unpack_from("<%dI" % no, blob, pos))
        pos += 4*no
        self.setters = [struct.unpack_from("<II", blob, pos + 8*k) for k in range(ns)]

    def operands(self, i):
        op, a, b, c = self.code[i][:4]
        if op in (CONST, INPUT):
            return []
        if op == FMA:
            return [a, b, c]
        if op in ONE_OPERAND:
            return [a]
        return [a, b]


def serialize(dtype, flags, name, symbols, tables, code, outputs, setters):
    name_bytes = name.encode() + b"\0"*(4 - len(name) % 4)
    blob = struct.pack("<8s8I", b"GFIR0001", dtype, len(symbols), len(outputs), len(setters), len(tables), len(code),
                       len(name_bytes), flags) + name_bytes
    for s in symbols:
        blob += struct.pack("<I", len(s)) + s
    for rows, cols, data in tables:
        blob += struct.pack("<II", rows, cols) + data
    for rec in code:
        blob += struct.pack("<6I4d", *rec)
    blob += struct.pack("<%dI" % len(outputs), *outputs)
    for value, target in setters:
        blob += struct.pack("<II", value, target)
    return blob


def renumber(item, order):
    new_index = {old: p for p, old in enumerate(order)}
    out = Item.__new__(Item)
    out.__dict__.update(item.__dict__)
    out.code = []
    for old in order:
        rec = list(item.code[old])
        ops = item.operands(old)
        if rec[0] == FMA:
            rec[1], rec[2], rec[3] = (new_index[o] for o in ops)
        elif len(ops) == 1:
            rec[1] = new_index[ops[0]]
        elif len(ops) == 2:
            rec[1], rec[2] = new_index[ops[0]], new_index[ops[1]]
        out.code.append(rec)
    out.outputs = [new_index[o] for o in item.outputs]
    out.setters = [(new_index[v], t) for v, t in item.setters]
    return out


def schedule_range(item, lo, hi, seed=None):
    """csrc/schedule.hpp restricted to the records [lo, hi) of an item already in a valid order: greedy list
    schedule (emit the ready node that frees the most operands, ties to the node that became ready last).
    A value handed over from before `lo` is scheduled like an input of the segment (it is loaded when the
    schedule asks for it); values used at or after `hi` (or stored) are roots."""
    n = len(item.code)
    later = set()
    for i in range(hi, n):
        later.update(item.operands(i))
    later.update(v for v, _ in item.setters)
    later.update(item.outputs)
    nodes = list(range(lo, hi))
    handed = sorted({o for i in nodes for o in item.operands(i) if o < lo and item.code[o][0] != CONST})
    load = {o: -(k + 1) for k, o in enumerate(handed)}           # virtual load node of a handed-over value
    every = {}
    for i in nodes:
        every[i] = sorted(set(load.get(o, o) for o in item.operands(i) if o >= lo or o in load))
    for o in handed:
        every[load[o]] = []
    users = {v: [] for v in every}
    consumers_left = {v: 0 for v in every}
    pending = {}
    for v, ops in every.items():
        pending[v] = len(ops)
        for o in ops:
            users[o].append(v)
            consumers_left[o] += 1
    is_root = {v: (v >= 0 and v in later) for v in every}
    is_const = {v: (v >= 0 and item.code[v][0] == CONST) for v in every}
    ready = {v for v in every if pending[v] == 0}
    stamp = {v: 0 for v in every}
    order = []
    emitted = 0
    rng = np.random.default_rng(seed) if seed is not None else None
    while ready:
        best, best_score = None, 1 << 30
        candidates = sorted(ready)
        if rng is not None:
            rng.shuffle(candidates)
        for v in candidates:
            freed = sum(1 for o in every[v] if not is_const[o] and consumers_left[o] == 1 and not is_root[o])
            score = (0 if is_const[v] else 1) - freed
            if score < best_score or (score == best_score and stamp[v] > stamp[best]):
                best, best_score = v, score
        ready.discard(best)
        emitted += 1
        if best >= 0:
            order.append(best)
        for o in every[best]:
            consumers_left[o] -= 1
        for u in users[best]:
            pending[u] -= 1
            if pending[u] == 0:
                ready.add(u)
                stamp[u] = emitted
    return order


def schedule_for_pressure(item):
    return renumber(item, schedule_range(item, 0, len(item.code)))


def bisect(item, max_nodes, min_nodes=300, tries=12):
    """Recursive bisection: schedule a segment on its own, cut it where the fewest values are live (middle
    60 % of its records), repeat while a segment holds more than `max_nodes` records.  Returns the item in
    the refined order and the cut positions."""
    item = schedule_for_pressure(item)
    cuts = []
    work = [(0, len(item.code))]
    while work:
        lo, hi = work.pop()
        if hi - lo <= max_nodes:
            continue
        best = None
        for seed in [None] + list(range(tries)):
            order = list(range(lo)) + schedule_range(item, lo, hi, seed) + list(range(hi, len(item.code)))
            trial = renumber(item, order)
#  cost of a cut at p: the values defined in [lo, p) and used in [p, hi) (loaded by the second half;
#  values handed in from before `lo`, or needed only after `hi`, pass by in memory at no cost)
            last = {}
            for i in range(lo, hi):
                for o in trial.operands(i):
                    last[o] = i
            cross = np.zeros(hi - lo + 1, dtype=int)
            for i in range(lo, hi):
                if trial.code[i][0] not in (CONST, INPUT) and last.get(i, -1) > i:
                    cross[i + 1 - lo:last[i] + 1 - lo] += 1
            a, b = lo + max(min_nodes, int(0.25*(hi - lo))), hi - max(min_nodes, int(0.25*(hi - lo)))
            if a >= b:
                break
            cut = a + int(np.argmin(cross[a - lo:b - lo]))
            if best is None or cross[cut - lo] < best[0]:
                best = (int(cross[cut - lo]), cut, trial)
        if best is None:
            continue
        _, cut, item = best
        cuts.append(cut)
        work.append((lo, cut))
        work.append((cut, hi))
    return item, sorted(cuts)


def liveness(item):
    n = len(item.code)
    last = [-1]*n
    for i in range(n):
        for o in item.operands(i):
            last[o] = i
    for o in item.outputs:
        last[o] = n
    for v, _ in item.setters:
        last[v] = n
    live = np.zeros(n + 1, dtype=int)           # live[p]: values defined before p and used at or after p
    for i in range(n):
        if item.code[i][0] in (CONST, INPUT) or last[i] < 0:
            continue
        live[i + 1:last[i] + 1] += 1
    return live, last


def choose_cuts(item, segments, window=0.12):
    live, _ = liveness(item)
    n = len(item.code)
    cuts = []
    for k in range(1, segments):
        centre = k*n//segments
        lo, hi = max(1, int(centre - window*n/segments*4)), min(n - 1, int(centre + window*n/segments*4))
        lo = max(lo, (cuts[-1] + 50) if cuts else 1)
        cuts.append(lo + int(np.argmin(live[lo:max(hi, lo + 1)])))
    return cuts, live


def split(item, cuts):
    n = len(item.code)
    bounds = [0] + list(cuts) + [n]
    _, last = liveness(item)
    pieces = []
    carried = {}                                # node -> scratch id
    for j in range(len(bounds) - 1):
        lo, hi = bounds[j], bounds[j + 1]
        final = j == len(bounds) - 2
        symbols = list(item.symbols)
        code, where = [], {}
        cut_in = []

        def local(node):
            if node in where:
                return where[node]
            rec = item.code[node]
            if rec[0] in (CONST, INPUT):
                code.append(list(rec))
            else:
                assert node < lo, (node, lo)
                cut_in.append(node)
                symbols.append(("s%d" % carried[node]).encode().ljust(4, b"\0"))
                code.append([INPUT, len(symbols) - 1, NONE, NONE, 0, 0, 0.0, 0.0, 0.0, 0.0])
            where[node] = len(code) - 1
            return where[node]

        used_tables = {}
        for i in range(lo, hi):
            rec = list(item.code[i])
            if rec[0] in (CONST, INPUT):
                continue                         # copied where used
            ops = item.operands(i)
            mapped = [local(o) for o in ops]
            for slot, m in zip((1, 2, 3), mapped):
                rec[slot] = m
            if rec[0] in (15, 16):
                rec[4] = used_tables.setdefault(rec[4], len(used_tables))
            code.append(rec)
            where[i] = len(code) - 1
        outputs, out_ids = [], []
        for i in range(lo, hi):
            if item.code[i][0] in (CONST, INPUT):
                continue
            if last[i] >= hi and not (final and last[i] == n):
                pass
            if last[i] >= hi and not final:
                carried[i] = len(carried)
                outputs.append(where[i])
                out_ids.append(("scratch", carried[i]))
        for k, o in enumerate(item.outputs):
            if lo <= o < hi and item.code[o][0] not in (CONST, INPUT):
                outputs.append(where[o])
                out_ids.append(("output", k))
            elif final and (o < lo or item.code[o][0] in (CONST, INPUT)) and not any(x == ("output", k) for p in pieces for x in p["out_ids"]):
                outputs.append(local(o))
                out_ids.append(("output", k))
        setters = []
        if final:
            for v, target in item.setters:
                setters.append((local(v), target))
        tables = [None]*len(used_tables)
        for old, new in used_tables.items():
            tables[new] = item.tables[old]
        blob = serialize(item.dtype, item.flags, "%s_s%d" % (item.name, j), symbols, tables, code, outputs, setters)
        pieces.append({"blob": blob, "cut_in": [carried[c] for c in cut_in], "out_ids": out_ids, "nodes": len(code),
                       "num_inputs": len(symbols)})
    return pieces, len(carried)


def main():
    from graph_framework_amd import Context
    from graph_framework_amd.xrays import workload
    path = workload("solver_kernel")
    blob = open(path, "rb").read()
    n = int(os.environ.get("RAYS", 1000000))
    state = dict(t=0.0, w=500.0, x=2.5, y=0.0, z=0.0, kx=-500.00000357884727, ky=0.0, kz=0.0)
    keys = ["t", "w", "x", "y", "z", "kx", "ky", "kz"]
    rng = np.random.default_rng(1)
    columns = [np.full(n, state[k]) for k in keys]
    columns[3] += rng.normal(0, 0.01, n)
    columns[4] += rng.normal(0, 0.01, n)
    columns[6] += rng.normal(0, 5.0, n)
    columns[7] += rng.normal(0, 5.0, n)
    steps = 50

    def run(pieces):
        context = Context(0)
        kernels = []
        for p in pieces:
            kernel = context.add_kernel(p["blob"], n)
            kernels.append(kernel)
        context.compile()
        for p, kernel in zip(pieces, kernels):
            in_keys = keys + ["scratch%d" % c for c in p["cut_in"]]
            out_keys = ["scratch%d" % i if kind == "scratch" else "residual" for kind, i in p["out_ids"]]
            init = [c.copy() for c in columns] + [np.zeros(n) for _ in p["cut_in"]]
            kernel.create_kernel_call(in_keys, out_keys, init)
        for kernel in kernels:                   # warm up
            kernel.run(1)
        context.wait()
        context.enable_timing(True)
        for _ in range(steps):
            for kernel in kernels:
                kernel.run(1)
        context.wait()
        times = [kernel.timing()[0] for kernel in kernels]
        infos = [kernel.info() for kernel in kernels]
        result = [context.copy_to_host(k, np.empty(n)) for k in keys + ["residual"]]
        flags = context.flags()
        context.close()
        return result, times, infos, flags

    whole, times, infos, flags = run([{"blob": blob, "cut_in": [], "out_ids": [("output", 0)]}])
    print("unsplit: %.4f ms  regs %s  flags %d" % (sum(times), [(i.vgprs, i.agprs, i.scratch_bytes) for i in infos], flags), flush=True)
    for max_nodes in [int(a) for a in sys.argv[1:]] or [2000, 1200, 600]:
        item, cuts = bisect(Item(blob), max_nodes)
        live, _ = liveness(item)
        segments = len(cuts) + 1
        pieces, scratch = split(item, cuts)
        result, times, infos, flags = run(pieces)
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(whole, result))
        traffic = sum(len(p["cut_in"]) + len(p["out_ids"]) + 8 for p in pieces) + 7
        print("segments %d cuts %s live %s scratch values %d: %.4f ms %s  regs %s  bit-identical %s flags %d  doubles moved/ray %d"
              % (segments, cuts, [int(live[c]) for c in cuts], scratch, sum(times), ["%.4f" % t for t in times],
                 [(i.vgprs, i.agprs, i.scratch_bytes) for i in infos], same, flags, traffic), flush=True)


if __name__ == "__main__":
    main()
