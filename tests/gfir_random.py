"""Random work items for differential testing (oracle interpreter vs HIP lowering).

Builds GFIR items (include/gfir.h) out of the node types whose device arithmetic is IEEE-exact
— constants, inputs, +, -, *, /, fma, sqrt, integer pow, 1-D and 2-D gathers — so that the two
sides must agree bit for bit.  Values are kept O(1) by construction (denominators u*u + c,
squashing v/(1 + v*v) after sums) so that neither side leaves the range in which the
shared-reciprocal division is the IEEE one; sub-expressions, denominators and gather arguments
are reused on purpose (hash-consed DAGs share them), and tables include exact multiples of one
another (the reducer folds constants into tables, which the lowering compacts).
"""
import struct

import numpy as np

CONST, INPUT, ADD, SUB, MUL, DIV, FMA, SQRT, POWI, GATHER1, GATHER2 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 16
NONE = 0xFFFFFFFF


class Builder:
    def __init__(self, rng, dtype, num_inputs):
        self.rng = rng
        self.dtype = dtype
        self.real = np.float64 if dtype == "f64" else np.float32
        self.code = []
        self.bound = []                 # |value| <= bound[i]
        self.tables = []
        self.cse = {}
        self.inputs = [self.emit(INPUT, a=i, bound=1.0) for i in range(num_inputs)]
        self.values = list(self.inputs)
        self.denominators = []

    def emit(self, op, a=NONE, b=NONE, c=NONE, aux=0, imm=(0.0, 0.0, 0.0, 0.0), bound=1.0):
        key = (op, a, b, c, aux, tuple(imm))
        if key in self.cse:             # hash-consing: one record per distinct node
            return self.cse[key]
        self.code.append((op, a, b, c, aux, tuple(imm)))
        self.bound.append(float(bound))
        self.cse[key] = len(self.code) - 1
        return len(self.code) - 1

    def constant(self, value):
        value = float(self.real(value))
        return self.emit(CONST, imm=(value, 0.0, 0.0, 0.0), bound=abs(value))

    def pick(self):
        values = self.values
        if self.rng.random() < 0.6:     # mostly recent values: deep chains; sometimes old ones: long live ranges
            return values[int(self.rng.integers(max(0, len(values) - 24), len(values)))]
        return values[int(self.rng.integers(0, len(values)))]

    def denominator(self):
        if self.denominators and self.rng.random() < 0.7:
            return self.denominators[int(self.rng.integers(0, len(self.denominators)))]
        u = self.pick()
        c = self.constant(self.rng.uniform(0.5, 2.0))
        d = self.emit(FMA, u, u, c, bound=self.bound[u]**2 + 2.0) if self.rng.random() < 0.5 else \
            self.emit(ADD, self.emit(MUL, u, u, bound=self.bound[u]**2), c, bound=self.bound[u]**2 + 2.0)
        self.denominators.append(d)
        return d

    def squash(self, v):
        """v/(1 + v*v): into [-0.5, 0.5]."""
        one = self.constant(1.0)
        d = self.emit(FMA, v, v, one, bound=self.bound[v]**2 + 1.0)
        return self.emit(DIV, v, d, bound=0.5)

    def table(self, rows, cols):
        """A new table, or an exact power-of-two multiple of an earlier one of the same shape."""
        same = [t for t in self.tables if t.shape == (rows, cols)]
        if same and self.rng.random() < 0.4:
            data = same[int(self.rng.integers(0, len(same)))]*float(self.rng.choice([0.5, 2.0, -4.0, 3.0]))
            data = data.astype(self.real).astype(np.float64)
        else:
            data = self.rng.uniform(-1.0, 1.0, (rows, cols)).astype(self.real).astype(np.float64)
        self.tables.append(data)
        return len(self.tables) - 1

    def gather_argument(self, length):
        """(node, scale, offset) with the node's range reaching past both ends of the table."""
        v = self.pick()
        bound = max(self.bound[v], 1.0e-3)
        scale = float(self.real(2.2*bound/length*self.rng.uniform(0.6, 1.0)))
        offset = float(self.real(-bound*self.rng.uniform(0.7, 1.0)))
        return v, scale, offset

    def grow(self):
        r = self.rng.random()
        a, b = self.pick(), self.pick()
        if r < 0.18:
            v = self.emit(ADD, a, b, bound=self.bound[a] + self.bound[b])
        elif r < 0.30:
            v = self.emit(SUB, a, b, bound=self.bound[a] + self.bound[b])
        elif r < 0.50:
            v = self.emit(MUL, a, b, bound=self.bound[a]*self.bound[b])
        elif r < 0.66:
            c = self.pick()
            v = self.emit(FMA, a, b, c, bound=self.bound[a]*self.bound[b] + self.bound[c])
        elif r < 0.84:
            d = self.denominator()
            v = self.emit(DIV, a, d, bound=self.bound[a]/0.5)
        elif r < 0.87:
            c = self.constant(self.rng.uniform(0.25, 2.0))
            s = self.emit(FMA, a, a, c, bound=self.bound[a]**2 + 2.0)
            v = self.emit(SQRT, s, bound=np.sqrt(self.bound[a]**2 + 2.0))
        elif r < 0.91:
            k = int(self.rng.integers(2, 5))
            v = self.emit(POWI, a, aux=k, bound=self.bound[a]**k)
        elif r < 0.96:
            cols = int(self.rng.choice([7, 16, 33]))
            shared = getattr(self, "argument1", None)
            if shared is None or shared[3] != cols or self.rng.random() < 0.3:
                self.argument1 = self.gather_argument(cols) + (cols,)
            arg, scale, offset, _ = self.argument1
            table = self.table(1, cols)
            v = self.emit(GATHER1, arg, aux=table, imm=(scale, offset, 0.0, 0.0),
                          bound=float(np.abs(self.tables[table]).max()))
        else:
            rows, cols = int(self.rng.choice([5, 12])), int(self.rng.choice([6, 9]))
            shared = getattr(self, "argument2", None)
            if shared is None or shared[6:] != (rows, cols) or self.rng.random() < 0.3:
                self.argument2 = self.gather_argument(rows) + self.gather_argument(cols) + (rows, cols)
            x, xs, xo, y, ys, yo, _, _ = self.argument2
            table = self.table(rows, cols)
            v = self.emit(GATHER2, x, y, aux=table, imm=(xs, xo, ys, yo),
                          bound=float(np.abs(self.tables[table]).max()))
        if self.bound[v] > 4.0:
            v = self.squash(v)
        if v not in self.values:
            self.values.append(v)
        return v


def random_item(seed, dtype="f64", num_inputs=6, num_nodes=400, num_outputs=3, num_setters=3, name="fuzz"):
    """Returns (GFIR bytes, number of instruction records)."""
    rng = np.random.default_rng(seed)
    b = Builder(rng, dtype, num_inputs)
    while len(b.code) < num_nodes:
        b.grow()
    tail = b.values[-max(32, num_outputs + num_setters):]
    outputs = [tail[int(rng.integers(0, len(tail)))] for _ in range(num_outputs)]
    setters = []
    for target in rng.permutation(num_inputs)[:num_setters]:
        v = tail[int(rng.integers(0, len(tail)))]
        if b.bound[v] > 1.0:            # setter targets feed the next pass: keep them in [-1, 1]
            v = b.squash(v)
        setters.append((v, int(target)))

    return serialize(b, outputs, setters, num_inputs, name), len(b.code)


def serialize(b, outputs, setters, num_inputs, name):
    """GFIR bytes (include/gfir.h) of a Builder's records with the given output nodes and
    (value node, input index) setters."""
    dtype = b.dtype
    name_bytes = name.encode() + b"\0"*(4 - len(name) % 4)
    blob = struct.pack("<8s8I", b"GFIR0001", 1 if dtype == "f64" else 0, num_inputs, len(outputs), len(setters),
                       len(b.tables), len(b.code), len(name_bytes), 0)
    blob += name_bytes
    for i in range(num_inputs):
        symbol = ("v%d" % i).encode()
        symbol += b"\0"*(4 - len(symbol) % 4)
        blob += struct.pack("<I", len(symbol)) + symbol
    for t in b.tables:
        blob += struct.pack("<II", t.shape[0], t.shape[1]) + np.ascontiguousarray(t, dtype="<f8").tobytes()
    for op, a, bb, c, aux, imm in b.code:
        blob += struct.pack("<6I4d", op, a, bb, c, aux, 0, *imm)
    blob += struct.pack("<%dI" % len(outputs), *outputs)
    for value, target in setters:
        blob += struct.pack("<II", value, target)
    return blob


def division_stress_item(dtype="f64", name="division_stress"):
    """A hand-made item around the corner cases of division: inputs n0, n1, d0, d1, x;
    quotients that share a denominator, a quotient used as a denominator, as a gather argument
    and under a square root; outputs and a setter that store quotients (and so may store a zero)."""
    rng = np.random.default_rng(5)
    b = Builder(rng, dtype, 5)
    n0, n1, d0, d1, x = b.inputs
    q0 = b.emit(DIV, n0, d0)
    q1 = b.emit(DIV, n1, d0)                               # shares d0's reciprocal
    q2 = b.emit(DIV, b.emit(MUL, n0, n1), d1)
    q3 = b.emit(DIV, x, b.emit(ADD, q0, b.constant(2.0)))  # a quotient inside a denominator
    table = b.table(1, 16)
    g = b.emit(GATHER1, q1, aux=table, imm=(float(b.real(0.25)), float(b.real(-2.0)), 0.0, 0.0))   # a quotient as gather argument
    table2 = b.table(5, 6)
    g2 = b.emit(GATHER2, q0, q2, aux=table2, imm=(float(b.real(0.5)), float(b.real(-1.0)), float(b.real(0.5)), float(b.real(-1.0))))
    s = b.emit(SQRT, b.emit(MUL, q0, q0))
    mix = b.emit(FMA, g, q3, b.emit(MUL, g2, s))
    outputs = [q0, q1, q2, q3, mix]
    setters = [(b.emit(DIV, x, d1), 4)]                    # x <- x/d1
    return serialize(b, outputs, setters, 5, name)
