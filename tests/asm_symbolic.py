"""Symbolic replay of the assembly body (graph_framework_amd/csrc/asm_body.hpp) on the CPU.

The statement the lowering writes for GFHIP_ASM=1 is executed instruction by instruction on SYMBOLIC values: every
register holds the hash-consed expression it was computed from (or half of one), LDS slots hold what was written to
them, loads are in flight until an s_waitcnt whose count covers them.  The writer annotates what each sequence defines
(`; def r12`, `; def q7` for a shared reciprocal, `; def c1_4` for a table value, `; spill`, `; fill`, `; out`); every
definition is held against the expression the item's DAG gives for that node, built here from the same machine
sequences (prelude.hpp: gf_rcp, gf_div, gf_sqrt_window, gf_pow_three_halves_window).  What this proves: the register
assignment, the LDS round trips, the re-loading of table values, the constants' pool and the wait counts never hand an
instruction a wrong or unfinished operand — for any item, without a GPU.  What it does not: the bits of the sequences
themselves (the GPU tests hold those against the oracle) and the layout of the table packs.
"""
import re
import struct

from oracle.gfir_to_c import OPS, parse

INLINE = {"0": 0.0, "0.5": 0.5, "-0.5": -0.5, "1.0": 1.0, "-1.0": -1.0, "2.0": 2.0, "-2.0": -2.0, "4.0": 4.0, "-4.0": -4.0}


def bits(value):
    return struct.unpack("<Q", struct.pack("<d", float(value)))[0]


class Expressions:
    """Hash-consed expressions: equal structure <=> equal id."""

    def __init__(self):
        self.ids = {}
        self.keys = []

    def mk(self, *key):
        found = self.ids.get(key)
        if found is None:
            found = len(self.keys)
            self.ids[key] = found
            self.keys.append(key)
        return found

    def const(self, value_bits):
        return self.mk("const", int(value_bits))

    def neg(self, x):
        key = self.keys[x]
        if key[0] == "const":
            return self.const(key[1] ^ (1 << 63))
        if key[0] == "neg":
            return key[1]
        return self.mk("neg", x)

    def show(self, x, depth=3):
        key = self.keys[x]
        if depth == 0 or key[0] in ("const", "in", "cell"):
            return "%s%s" % (key[0], list(key[1:]))
        return "%s(%s)" % (key[0], ", ".join(self.show(k, depth - 1) if isinstance(k, int) and key[0] not in ("const", "in", "cell", "mul24", "mad24") else str(k) for k in key[1:]))


class Expected:
    """The expression of every node of the item, through the sequences the writer emits."""

    def __init__(self, item, ex):
        self.item, self.ex = item, ex
        self.ins = item["ins"]
        self.memo = {}
        self.reciprocals = {}
        self.cells = {}             # name -> expression (from the replay's `def c..` lines)
        self.alias = {}             # gather node -> cell name
        self.one, self.half, self.zero = ex.const(bits(1.0)), ex.const(bits(0.5)), ex.const(bits(0.0))

    def reciprocal_of(self, d):
        ex = self.ex
        r = ex.mk("rcp", d)
        e = ex.mk("fma", ex.neg(d), r, self.one)
        r = ex.mk("fma", r, e, r)
        e = ex.mk("fma", ex.neg(d), r, self.one)
        return ex.mk("fma", r, e, r)

    def quotient(self, n, d, r):
        ex = self.ex
        q = ex.mk("mul", n, r)
        e = ex.mk("fma", ex.neg(d), q, n)
        return ex.mk("fma", e, r, q)

    def square_root(self, x):
        ex = self.ex
        y = ex.mk("rsq", x)
        g = ex.mk("mul", x, y)
        h = ex.mk("mul", y, self.half)
        r = ex.mk("fma", ex.neg(h), g, self.half)
        g = ex.mk("fma", g, r, g)
        h = ex.mk("fma", h, r, h)
        d = ex.mk("fma", ex.neg(g), g, x)
        g = ex.mk("fma", d, h, g)
        d = ex.mk("fma", ex.neg(g), g, x)
        return ex.mk("fma", d, h, g)

    def three_halves(self, x):
        ex = self.ex
        s = self.square_root(x)
        t = ex.mk("mul", ex.mk("rcp", s), self.half)
        u = ex.mk("fma", ex.neg(s), s, x)
        t = ex.mk("mul", u, t)
        p = ex.mk("mul", x, s)
        u = ex.mk("fma", x, s, ex.neg(p))
        t = ex.mk("mul", x, t)
        u = ex.mk("add", u, t)
        return ex.mk("add", p, u)

    def index_quotient(self, arg, scale, offset):
        ex = self.ex
        n = ex.mk("add", self.node(arg), ex.const(bits(-offset)))
        q = ex.mk("mul", n, ex.const(bits(1.0/scale)))
        e = ex.mk("fma", ex.const(bits(-scale)), q, n)
        return ex.mk("fma", e, ex.const(bits(1.0/scale)), q)

    def cell_offset(self, i, stride, lds_pack):
        """Byte offset of the cell of gather node i's group (plus the LDS base of a staged pack)."""
        ex, c = self.ex, self.ins[i]
        rows, cols, _ = self.item["tables"][int(c["aux"])]
        two = int(c["op"]) == OPS["GATHER2"]
        quotients = [self.index_quotient(int(c["a"]), float(c["imm"][0]), float(c["imm"][1]))]
        lengths = [rows if two else cols]
        if two:
            quotients.append(self.index_quotient(int(c["b"]), float(c["imm"][2]), float(c["imm"][3])))
            lengths.append(cols)
        index = [ex.mk("cvt_u32", ex.mk("min", ex.mk("max", q, self.zero), ex.const(bits(float(length - 1)))))
                 for q, length in zip(quotients, lengths)]
        cell = ex.mk("mad24", index[0], cols, index[1]) if two else index[0]
        offset = ex.mk("mul24", stride, cell)
        if lds_pack >= 0:
            offset = ex.mk("add_u32", "lds%d" % lds_pack, offset)
        return offset, quotients

    def node(self, i):
        if i in self.memo:
            return self.memo[i]
        ex, c = self.ex, self.ins[i]
        op, a, b, cc, aux = int(c["op"]), int(c["a"]), int(c["b"]), int(c["c"]), int(c["aux"])
        if op == OPS["CONST"]:
            out = ex.const(bits(c["imm"][0]))
        elif op == OPS["INPUT"]:
            out = ex.mk("in", a)
        elif op == OPS["ADD"]:
            out = ex.mk("add", self.node(a), self.node(b))
        elif op == OPS["SUB"]:
            out = ex.mk("add", self.node(a), ex.neg(self.node(b)))
        elif op == OPS["MUL"]:
            out = ex.mk("mul", self.node(a), self.node(b))
        elif op == OPS["FMA"]:
            out = ex.mk("fma", self.node(a), self.node(b), self.node(cc))
        elif op == OPS["POWI"]:
            out = ex.mk("mul", self.node(a), self.node(a))
            for _ in range(2, aux):
                out = ex.mk("mul", out, self.node(a))
        elif op == OPS["DIV"]:
            out = self.quotient(self.node(a), self.node(b), self.reciprocal(b))
        elif op == OPS["SQRT"]:
            out = self.square_root(self.node(a))
        elif op == OPS["POW"]:
            out = self.three_halves(self.node(a))
        elif op in (OPS["GATHER1"], OPS["GATHER2"]):
            out = self.cells[self.alias[i]]
        else:
            raise ValueError("no sequence for op %d" % op)
        self.memo[i] = out
        return out

    def reciprocal(self, d):
        if d not in self.reciprocals:
            self.reciprocals[d] = self.reciprocal_of(self.node(d))
        return self.reciprocals[d]


class ReplayError(AssertionError):
    pass


def statement_of(source):
    """The lines of the assembly statement of a generated kernel text."""
    start = source.index("asm volatile(\n", source.index("float dmax"))
    end = source.index("                : [", start)
    return [m.group(1) for m in re.finditer(r'^\s*"(.*)\\n"$', source[start:end], re.M)]


def replay(blob, source):
    """Replays the statement of `source` (generated for the item `blob`); returns statistics, raises ReplayError."""
    item = parse(blob)
    ex = Expressions()
    expected = Expected(item, ex)
    lines = statement_of(source)

    vgpr = {}                       # register number -> ("lo" | "hi", expression) | ("u32", expression)
    sgpr = {}                       # register number -> 32-bit integer
    pending = {}                    # register number -> (counter, sequence)
    issued = {"vm": 0, "lgkm": 0}
    done = {"vm": -1, "lgkm": -1}
    slots = {}                      # (base, offset) -> expression
    named = {}                      # statement operand -> expression (outputs written)
    defined = {}                    # annotation name -> expression
    groups = {}                     # group number -> (offset expression, quotient expressions)
    tracked = {"dmax": set(), "dmin": set(), "vmax": set()}
    outputs = []
    stats = {"instructions": 0, "spills": 0, "fills": 0, "loads": 0, "definitions": 0}
    load_columns = {}               # group -> table value -> byte offset within the cell it is loaded from

    def fail(message, line):
        raise ReplayError("%s\n    in: %s" % (message, line))

    def check_ready(register, line):
        if register in pending:
            counter, sequence = pending[register]
            if sequence > done[counter]:
                fail("v%d is read while its load (%s #%d) is in flight" % (register, counter, sequence), line)
            del pending[register]

    def write32(register, value, line):
        if register in pending and pending[register][1] > done[pending[register][0]]:
            fail("v%d is overwritten while a load into it is in flight" % register, line)
        pending.pop(register, None)
        vgpr[register] = value

    def read_pair(text, line):
        negative = text.startswith("-")
        if negative:
            text = text[1:]
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", text)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            if hi != lo + 1 or lo % 2:
                fail("misaligned pair", line)
            check_ready(lo, line)
            check_ready(hi, line)
            a, b = vgpr.get(lo), vgpr.get(hi)
            if not a or not b or a[0] != "lo" or b[0] != "hi" or a[1] != b[1]:
                fail("v[%d:%d] does not hold one value: %r %r" % (lo, hi, a, b), line)
            value = a[1]
        elif re.fullmatch(r"s\[(\d+):(\d+)\]", text):
            lo = int(text[2:text.index(":")])
            if lo not in sgpr or lo + 1 not in sgpr:
                fail("constant read before it was loaded", line)
            value = ex.const(sgpr[lo] | (sgpr[lo + 1] << 32))
        elif text in INLINE:
            value = ex.const(bits(INLINE[text]))
        elif re.fullmatch(r"%\[v(\d+)\]", text):
            value = ex.mk("in", int(text[3:-1]))
        else:
            fail("operand %r not understood" % text, line)
        return ex.neg(value) if negative else value

    def write_pair(text, value, line):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", text)
        if m:
            lo = int(m.group(1))
            write32(lo, ("lo", value), line)
            write32(lo + 1, ("hi", value), line)
        elif re.fullmatch(r"%\[\w+\]", text):
            named[text[2:-1]] = value
        else:
            fail("destination %r not understood" % text, line)

    def read32(text, line):
        m = re.fullmatch(r"v(\d+)", text)
        if m:
            check_ready(int(m.group(1)), line)
            value = vgpr.get(int(m.group(1)))
            if not value:
                fail("v%s read before written" % m.group(1), line)
            return value
        m = re.fullmatch(r"%\[(lds\d+|park\d+)\]", text)
        if m:
            return ("u32", (m.group(1),))
        fail("32-bit operand %r not understood" % text, line)

    def define(name, value, line):
        stats["definitions"] += 1
        kind = name[0]
        if kind == "r":
            want = expected.node(int(name[1:]))
        elif kind == "q":
            want = expected.reciprocal(int(name[1:]))
        else:
            want = expected.cells.get(name)
            if want is None:
                fail("table value %s defined before its kind is known" % name, line)
        if want != value:
            fail("%s is defined as %s, the item says %s" % (name, ex.show(value), ex.show(want)), line)
        defined[name] = value

    for line in lines:
        text, _, note = line.partition(";")
        text, note = text.strip(), note.strip()
        if not text:
            m = re.fullmatch(r"alias r(\d+) = (c\d+_\d+)", note)
            if m:
                expected.alias[int(m.group(1))] = m.group(2)
                group, table = (int(x) for x in m.group(2)[1:].split("_"))
                if m.group(2) not in expected.cells:
#  A stored table's value: whatever is loaded at its column of the group's cell; a derived one is declared by its `def`.
                    expected.cells.setdefault(m.group(2), ex.mk("cell", group, table))
            continue
        stats["instructions"] += 1
        opcode, _, rest = text.partition(" ")
        operands = [o.strip() for o in re.split(r",\s*(?![^()]*\))", rest)] if rest else []
        if opcode == "s_waitcnt":
            for counter, count in re.findall(r"(vm|lgkm)cnt\((\d+)\)", rest):
                done[counter] = max(done[counter], issued[counter] - 1 - int(count))
        elif opcode == "s_nop":
            pass
        elif opcode == "s_mov_b32":
            sgpr[int(operands[0][1:])] = int(operands[1], 16)
        elif opcode in ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_min_f64"):
            values = [read_pair(o, line) for o in operands[1:]]
            kind = opcode[2:-4]
            write_pair(operands[0], ex.mk(kind, *values), line)
        elif opcode in ("v_rcp_f64_e32", "v_rsq_f64_e32"):
            write_pair(operands[0], ex.mk(opcode[2:5], read_pair(operands[1], line)), line)
        elif opcode == "v_mov_b64":
            write_pair(operands[0], read_pair(operands[1], line), line)
        elif opcode == "v_mov_b32_e32":
            source_text = operands[1]
            value = sgpr[int(source_text[1:])] if source_text.startswith("s") else int(source_text, 16)
            register = int(operands[0][1:])
            write32(register, ("raw", value), line)
            lo = register & ~1
            a, b = vgpr.get(lo), vgpr.get(lo + 1)
            if a and b and a[0] == "raw" and b[0] == "raw":
                constant = ex.const(a[1] | (b[1] << 32))
                vgpr[lo], vgpr[lo + 1] = ("lo", constant), ("hi", constant)
        elif opcode == "v_cvt_u32_f64_e32":
            write32(int(operands[0][1:]), ("u32", ex.mk("cvt_u32", read_pair(operands[1], line))), line)
        elif opcode == "v_mad_u32_u24":
            a, b = read32(operands[1], line), read32(operands[3], line)
            write32(int(operands[0][1:]), ("u32", ex.mk("mad24", a[1], int(operands[2]), b[1])), line)
        elif opcode == "v_mul_u32_u24_e32":
            a = read32(operands[2], line)
            write32(int(operands[0][1:]), ("u32", ex.mk("mul24", int(operands[1], 16), a[1])), line)
        elif opcode == "v_add_u32_e32":
            a, b = read32(operands[1], line), read32(operands[2], line)
            write32(int(operands[0][1:]), ("u32", ex.mk("add_u32", a[1][0] if isinstance(a[1], tuple) else a[1], b[1])), line)
        elif opcode in ("v_maximum3_f32", "v_minimum3_f32"):
            tracker = operands[0][2:-1]
            if operands[1][2:-1] != tracker:
                fail("tracker mixed up", line)
            for o in operands[2:]:
                m = re.fullmatch(r"abs\(v(\d+)\)", o)
                register = int(m.group(1))
                check_ready(register, line)
                value = vgpr.get(register)
                if not value or value[0] != "hi":
                    fail("tracked register v%d is not the high half of a value" % register, line)
                tracked[tracker].add(value[1])
        elif opcode == "global_load_dwordx2":
            address = read32(operands[1], line)
            m = re.fullmatch(r"%\[pack(\d+)\] offset:(\d+)", operands[2])
            lo = int(re.fullmatch(r"v\[(\d+):(\d+)\]", operands[0]).group(1))
            name = re.fullmatch(r"def (c\d+_\d+)", note).group(1)
            group = int(name[1:].split("_")[0])
            if group not in groups or groups[group][0] != address[1]:
                fail("%s is loaded through a register that does not hold the cell offset of group %d" % (name, group), line)
            value = expected.cells.setdefault(name, ex.mk("cell", *[int(x) for x in name[1:].split("_")]))
            write_pair(operands[0], value, line)
            for register in (lo, lo + 1):
                pending[register] = ("vm", issued["vm"])
            issued["vm"] += 1
            stats["loads"] += 1
            defined[name] = value
            column = load_columns.setdefault(group, {}).setdefault(name, int(m.group(2)))
            if column != int(m.group(2)):
                fail("%s is loaded from two different columns" % name, line)
        elif opcode == "global_load_dwordx4":
#  two neighbouring columns of the cell in one load
            address = read32(operands[1], line)
            offset = int(re.fullmatch(r"%\[pack(\d+)\] offset:(\d+)", operands[2]).group(2))
            lo = int(re.fullmatch(r"v\[(\d+):(\d+)\]", operands[0]).group(1))
            names = re.fullmatch(r"def (c\d+_\d+) (c\d+_\d+)", note).groups()
            for k, name in enumerate(names):
                group = int(name[1:].split("_")[0])
                if group not in groups or groups[group][0] != address[1]:
                    fail("%s is loaded through a register that does not hold the cell offset of group %d" % (name, group), line)
                value = expected.cells.setdefault(name, ex.mk("cell", *[int(x) for x in name[1:].split("_")]))
                write_pair("v[%d:%d]" % (lo + 2*k, lo + 2*k + 1), value, line)
                defined[name] = value
                if load_columns.setdefault(group, {}).setdefault(name, offset + 8*k) != offset + 8*k:
                    fail("%s is loaded from two different columns" % name, line)
            for register in range(lo, lo + 4):
                pending[register] = ("vm", issued["vm"])
            issued["vm"] += 1
            stats["loads"] += 1
        elif opcode == "ds_read_b64":
            lo = int(re.fullmatch(r"v\[(\d+):(\d+)\]", operands[0]).group(1))
            address_text, _, offset = operands[1].partition(" offset:")
            if note.startswith("fill "):
                name = note[5:]
                key = (address_text, int(offset))
                if key not in slots:
                    fail("fill from a slot nothing was written to", line)
                value = slots[key]
                if defined.get(name) != value:
                    fail("slot holds another value than %s" % name, line)
                stats["fills"] += 1
            else:
                name = re.fullmatch(r"def (c\d+_\d+)", note).group(1)
                group = int(name[1:].split("_")[0])
                address = read32(address_text, line)
                if group not in groups or groups[group][0] != address[1]:
                    fail("%s is loaded through a register that does not hold the cell offset of group %d" % (name, group), line)
                value = expected.cells.setdefault(name, ex.mk("cell", *[int(x) for x in name[1:].split("_")]))
                defined[name] = value
                stats["loads"] += 1
            write_pair(operands[0], value, line)
            for register in (lo, lo + 1):
                pending[register] = ("lgkm", issued["lgkm"])
            issued["lgkm"] += 1
        elif opcode == "ds_write_b64":
            address_text = operands[0]
            data_text, _, offset = operands[1].partition(" offset:")
            value = read_pair(data_text, line)
            name = note[6:]
            if defined.get(name) != value:
                fail("spill of %s writes another value" % name, line)
            slots[(address_text, int(offset))] = value
            issued["lgkm"] += 1
            stats["spills"] += 1
        else:
            fail("opcode %s not understood" % opcode, line)

        if note.startswith("def ") and opcode not in ("global_load_dwordx2", "global_load_dwordx4", "ds_read_b64"):
            words = note.split()
            name = words[1]
            if name[0] == "g":
                node_index, stride, lds = int(words[3]), int(words[5]), int(words[9])
                pack = int(words[7])
                want, quotients = expected.cell_offset(node_index, stride, pack if lds else -1)
                got = vgpr[int(operands[0][1:])][1]
                if got != want:
                    fail("cell offset of group %s is %s, the item says %s" % (name, ex.show(got, 6), ex.show(want, 6)), line)
                groups[int(name[1:])] = (got, quotients)
            elif name[0] == "c":
#  a derived table value: factor times its parent's
                parent_name, factor_bits = words[3], int(words[5])
                expected.cells[name] = ex.mk("mul", expected.cells[parent_name], ex.const(factor_bits))
                value = read_pair(operands[0], line)
                define(name, value, line)
            else:
                define(name, read_pair(operands[0], line), line)
        elif note.startswith("out "):
            words = note.split()
            want = ex.const(int(words[2])) if words[1] == "constant" else expected.node(int(words[1][1:]))
            got = named[operands[0][2:-1]]
            if got != want:
                fail("output %s is %s, the item says %s" % (operands[0], ex.show(got), ex.show(want)), line)
            outputs.append(operands[0][2:-1])

#  Every stored value written, every denominator / root argument / index quotient tracked.
    wanted_outputs = ["sv%d" % k for k in range(len(item["setters"]))] + ["so%d" % o for o in range(len(item["outputs"]))]
    if sorted(outputs) != sorted(wanted_outputs):
        raise ReplayError("outputs written: %s, wanted: %s" % (outputs, wanted_outputs))
    wanted = set()
    for i, c in enumerate(item["ins"]):
        op = int(c["op"])
        if op == OPS["DIV"] and ("q%d" % int(c["b"])) in defined:
            wanted.add(expected.node(int(c["b"])))
        if op in (OPS["SQRT"], OPS["POW"]) and ("r%d" % i) in defined:
            wanted.add(expected.node(int(c["a"])))
    if tracked["dmax"] != wanted or tracked["dmin"] != wanted:
        raise ReplayError("window check tracks %d values, the item has %d denominators and root arguments" % (len(tracked["dmax"]), len(wanted)))
    wanted_quotients = set()
    for _, quotients in groups.values():
        wanted_quotients.update(quotients)
    if tracked["vmax"] != wanted_quotients:
        raise ReplayError("finite check of the index quotients is incomplete")
    stats["slots"] = len(slots)
    return stats
