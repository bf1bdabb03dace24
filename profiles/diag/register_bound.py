"""How many register copies does the RK4 item need at best?  (analysis on the CPU.)

The emitted order of the lowered item (the text hipcc compiles) is simulated with 256 architectural VGPRs per lane:
every fp64 value takes two, an index one; constants are literals.  When a definition finds no free register the live
value whose next use is farthest away is evicted (Belady) to the AGPR file (written once: values are immutable) and
read back before its next use.  Reported: copies (v_accvgpr_write + v_accvgpr_read, two per fp64 value each way) for
several amounts of headroom left to the temporaries of the division, square-root and index sequences.

    python profiles/diag/register_bound.py [workload.gfir]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_framework_amd import generate_source  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "graph_framework_amd", "workloads", "solver_kernel_f64.gfir")
    source, _ = generate_source(open(path, "rb").read())
    lines = source.splitlines()
    start = next(i for i, l in enumerate(lines) if "__global__" in l)
    body = lines[start:]
    b0 = next(i for i, l in enumerate(body) if "float dmax" in l)
    b1 = next(i for i, l in enumerate(body) if re.match(r"\s*sv0 = ", l))
    define = re.compile(r"^\s*const (real|unsigned int|float|double) (\w+) = (.*);")
    order = []
    for l in body[b0 + 1:b1]:
        m = define.match(l)
        if m and not re.fullmatch(r"-?0x[0-9a-f.]+p[+-]\d+", m.group(3)):
            order.append((m.group(2), 1 if m.group(1) == "unsigned int" else 2, set(re.findall(r"\b[A-Za-z_]\w*\b", m.group(3)))))
    names = {n for n, _, _ in order}
    inputs = {"v%d" % i: 2 for i in range(8)}                      # the state, live from the start
    uses = {}
    for position, (_, _, used) in enumerate(order):
        for x in used:
            if x in names or x in inputs:
                uses.setdefault(x, []).append(position)
    tail = "\n".join(body[b1:b1 + 40])
    for x in set(re.findall(r"\b[A-Za-z_]\w*\b", tail)):
        if x in names or x in inputs:
            uses.setdefault(x, []).append(len(order))
    size = dict(inputs)
    size.update({n: s for n, s, _ in order})

    for headroom in (16, 32, 48):
        budget = 256 - headroom
        resident, spilled = dict(inputs), set()
        used_registers = sum(resident.values())
        writes = reads = 0
        cursor = {x: 0 for x in uses}

        def next_use(x, position):
            u = uses.get(x, [])
            while cursor[x] < len(u) and u[cursor[x]] < position:
                cursor[x] += 1
            return u[cursor[x]] if cursor[x] < len(u) else 1 << 30

        def make_room(need, position, keep):
            nonlocal used_registers, writes
            while used_registers + need > budget:
                victim = max((x for x in resident if x not in keep), key=lambda x: next_use(x, position))
                if next_use(victim, position) < (1 << 30) and victim not in spilled:
                    spilled.add(victim)
                    writes += resident[victim]
                used_registers -= resident.pop(victim)

        peak_spilled = 0
        for position, (name, registers, used) in enumerate(order):
            alive_spilled = sum(1 for x in spilled if next_use(x, position) < (1 << 30))
            peak_spilled = max(peak_spilled, alive_spilled)
            operands = [x for x in used if x in size and x in uses]
            for x in operands:
                if x not in resident:                                   # reload
                    make_room(size[x], position, set(operands))
                    resident[x] = size[x]
                    used_registers += size[x]
                    reads += size[x]
            for x in list(resident):                                    # free what dies here
                if next_use(x, position + 1) == 1 << 30 and x != name:
                    used_registers -= resident.pop(x)
            if name in uses:
                make_room(registers, position, set())
                resident[name] = registers
                used_registers += registers
        print("256 VGPRs, %2d kept for temporaries: %4d v_accvgpr_write + %4d v_accvgpr_read = %4d copies per pass "
              "(hipcc: 256 + 428 = 684); at most %d spilled values alive at once (LDS slots, were they parked there: "
              "%d ds_write + %d ds_read)" % (headroom, writes, reads, writes + reads, peak_spilled, writes//2, reads//2))


if __name__ == "__main__":
    main()
