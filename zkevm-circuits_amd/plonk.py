"""Host-side mirror of the constraint-system types that feed `Evaluator::evaluate_h`.

Mirrors the `halo2_proofs 0.3.0` interface the reference's circuits are written against
(`plonk::{Expression, ConstraintSystem, Column, Rotation}` — e.g.
/root/reference/zkevm-circuits/src/core_circuit.rs:119-289 builds gates and `lookup_any`s with exactly
these constructors) and `plonk::evaluation::{Evaluator::new, GraphEvaluator::add_expression}`, which
lowers gate / lookup expressions into the straight-line program the accelerator consumes.

Expressions are nested tuples so they can be handed to any checker unchanged:
    ("const", v) ("fixed", col, rot) ("advice", col, rot) ("instance", col, rot) ("challenge", i)
    ("neg", a) ("sum", a, b) ("prod", a, b) ("scaled", a, f)
Field values are canonical integers mod r.
"""
from __future__ import annotations

from dataclasses import dataclass, field

FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

# ValueSource kinds / Calculation opcodes: numeric values are the wire format of include/zkhip.h
VS_CONSTANT, VS_INTERMEDIATE, VS_FIXED, VS_ADVICE, VS_INSTANCE, VS_CHALLENGE, VS_BETA, VS_GAMMA, \
    VS_THETA, VS_Y, VS_PREVIOUS = range(11)
OP_ADD, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_HORNER, OP_STORE = range(8)

ANY_ADVICE, ANY_FIXED, ANY_INSTANCE = VS_ADVICE, VS_FIXED, VS_INSTANCE


# ----------------------------------------------------------------------------- Expression helpers
def Constant(v):
    return ("const", v % FR_MODULUS)


def Fixed(col, rot=0):
    return ("fixed", col, rot)


def Advice(col, rot=0):
    return ("advice", col, rot)


def Instance(col, rot=0):
    return ("instance", col, rot)


def Challenge(i):
    return ("challenge", i)


def Neg(a):
    return ("neg", a)


def Sum(a, b):
    return ("sum", a, b)


def Sub(a, b):
    return ("sum", a, ("neg", b))


def Product(a, b):
    return ("prod", a, b)


def Scaled(a, f):
    return ("scaled", a, f % FR_MODULUS)


def expr_degree(e) -> int:
    t = e[0]
    if t in ("const", "challenge"):
        return 0
    if t in ("fixed", "advice", "instance"):
        return 1
    if t in ("neg", "scaled"):
        return expr_degree(e[1])
    if t == "sum":
        return max(expr_degree(e[1]), expr_degree(e[2]))
    if t == "prod":
        return expr_degree(e[1]) + expr_degree(e[2])
    raise ValueError(t)


def _walk_queries(e, out):
    t = e[0]
    if t in ("fixed", "advice", "instance"):
        out.add((t, e[1], e[2]))
    elif t in ("neg", "scaled"):
        _walk_queries(e[1], out)
    elif t in ("sum", "prod"):
        _walk_queries(e[1], out)
        _walk_queries(e[2], out)


# ----------------------------------------------------------------------------- ConstraintSystem
@dataclass
class Lookup:
    input_expressions: list
    table_expressions: list

    def required_degree(self) -> int:
        ind = max([1] + [expr_degree(e) for e in self.input_expressions])
        tbl = max([1] + [expr_degree(e) for e in self.table_expressions])
        return max(4, 2 + ind + tbl)


@dataclass
class ConstraintSystem:
    num_fixed_columns: int = 0
    num_advice_columns: int = 0
    num_instance_columns: int = 0
    num_challenges: int = 0
    gates: list = field(default_factory=list)          # list of gates; a gate = list of polynomials
    lookups: list = field(default_factory=list)        # list[Lookup]
    permutation_columns: list = field(default_factory=list)  # [(ANY_*, index)]
    minimum_degree: int | None = None
    advice_column_phase: list | None = None            # phase of each advice column (None: all in the first phase)
    challenge_phase: list | None = None                # phase after which each challenge is squeezed (None: the first)
    # Upstream's ConstraintSystem records every (column, rotation) query in the order of the circuit's first `query_*` /
    # `enable_equality` call for it; that order decides the order of the evaluations in the proof.  A caller that mirrors a
    # real circuit sets these lists ([(column index, rotation)]); left at None they are derived from the expressions in
    # order of first appearance (gates, lookups, permutation columns), which is what a circuit gets that configures in
    # that order.
    advice_queries: list | None = None
    fixed_queries: list | None = None
    instance_queries: list | None = None
    num_selectors: int = 0                             # selectors the circuit declared (they are fixed columns here already)

    def phases(self):
        """[(phase, its advice columns, its challenges)] in proving order"""
        adv = self.advice_column_phase or [0] * self.num_advice_columns
        chl = self.challenge_phase or [0] * self.num_challenges
        if len(adv) != self.num_advice_columns or len(chl) != self.num_challenges:
            raise ValueError("phase lists do not match the column / challenge counts")
        return [(ph, [c for c, p in enumerate(adv) if p == ph], [c for c, p in enumerate(chl) if p == ph])
                for ph in sorted(set(adv) | set(chl))]

    def _memo(self, name, compute):
        """degree() and blinding_factors() walk every expression; callers ask for them per device call, so the values
        are kept until the lists of gates / lookups change length"""
        key = (len(self.gates), sum(len(g) for g in self.gates), len(self.lookups), self.minimum_degree,
               len(self.permutation_columns))
        cache = self.__dict__.setdefault("_memo_cache", {})
        if cache.get("key") != key:
            cache.clear()
            cache["key"] = key
        if name not in cache:
            cache[name] = compute()
        return cache[name]

    def degree(self) -> int:
        return self._memo("degree", self._degree)

    def blinding_factors(self) -> int:
        return self._memo("blinding_factors", self._blinding_factors)

    def _degree(self) -> int:
        d = 3  # permutation argument's required degree
        for lk in self.lookups:
            d = max(d, lk.required_degree())
        for g in self.gates:
            for poly in g:
                d = max(d, expr_degree(poly))
        return max(d, self.minimum_degree or 1)

    def _blinding_factors(self) -> int:
        per_col = {}
        q = set()
        for g in self.gates:
            for poly in g:
                _walk_queries(poly, q)
        for lk in self.lookups:
            for e in lk.input_expressions + lk.table_expressions:
                _walk_queries(e, q)
        # `enable_equality` queries an advice column of the permutation argument at the current rotation, and upstream's
        # num_advice_queries counts that query like any other
        for kind, idx in self.permutation_columns:
            if kind == ANY_ADVICE:
                q.add(("advice", idx, 0))
        for (t, col, _rot) in q:
            if t == "advice":
                per_col[col] = per_col.get(col, 0) + 1
        factors = max([1] + list(per_col.values()))
        return max(3, factors) + 2

    def minimum_rows(self) -> int:
        return self.blinding_factors() + 3

    def permutation_chunk_len(self) -> int:
        return self.degree() - 2

    def num_permutation_sets(self) -> int:
        n = len(self.permutation_columns)
        c = self.permutation_chunk_len()
        return (n + c - 1) // c


# ----------------------------------------------------------------------------- GraphEvaluator
class GraphEvaluator:
    """`plonk::evaluation::GraphEvaluator`: CSE'd straight-line program over ValueSources."""

    def __init__(self):
        self.constants = [0, 1, 2]
        self.rotations = []
        self.calculations = []   # calculation i has target i
        self._index = {}

    @property
    def num_intermediates(self):
        return len(self.calculations)

    def add_rotation(self, rot: int) -> int:
        if rot in self.rotations:
            return self.rotations.index(rot)
        self.rotations.append(rot)
        return len(self.rotations) - 1

    def add_constant(self, c: int):
        c %= FR_MODULUS
        if c in self.constants:
            return (VS_CONSTANT, self.constants.index(c), 0)
        self.constants.append(c)
        return (VS_CONSTANT, len(self.constants) - 1, 0)

    def add_calculation(self, calc):
        key = repr(calc)
        if key in self._index:
            return (VS_INTERMEDIATE, self._index[key], 0)
        self.calculations.append(calc)
        self._index[key] = len(self.calculations) - 1
        return (VS_INTERMEDIATE, len(self.calculations) - 1, 0)

    def add_expression(self, e):
        zero, one, two = (VS_CONSTANT, 0, 0), (VS_CONSTANT, 1, 0), (VS_CONSTANT, 2, 0)
        t = e[0]
        if t == "const":
            return self.add_constant(e[1])
        if t == "fixed":
            return self.add_calculation((OP_STORE, (VS_FIXED, e[1], self.add_rotation(e[2]))))
        if t == "advice":
            return self.add_calculation((OP_STORE, (VS_ADVICE, e[1], self.add_rotation(e[2]))))
        if t == "instance":
            return self.add_calculation((OP_STORE, (VS_INSTANCE, e[1], self.add_rotation(e[2]))))
        if t == "challenge":
            return self.add_calculation((OP_STORE, (VS_CHALLENGE, e[1], 0)))
        if t == "neg":
            if e[1][0] == "const":
                return self.add_constant(-e[1][1])
            ra = self.add_expression(e[1])
            return ra if ra == zero else self.add_calculation((OP_NEGATE, ra))
        if t == "sum":
            a, b = e[1], e[2]
            if b[0] == "neg":
                ra, rb = self.add_expression(a), self.add_expression(b[1])
                if ra == zero:
                    return self.add_calculation((OP_NEGATE, rb))
                if rb == zero:
                    return ra
                return self.add_calculation((OP_SUB, ra, rb))
            ra, rb = self.add_expression(a), self.add_expression(b)
            if ra == zero:
                return rb
            if rb == zero:
                return ra
            return self.add_calculation((OP_ADD, ra, rb) if ra <= rb else (OP_ADD, rb, ra))
        if t == "prod":
            ra, rb = self.add_expression(e[1]), self.add_expression(e[2])
            if ra == zero or rb == zero:
                return zero
            if ra == one:
                return rb
            if rb == one:
                return ra
            if ra == two:
                return self.add_calculation((OP_DOUBLE, rb))
            if rb == two:
                return self.add_calculation((OP_DOUBLE, ra))
            if ra == rb:
                return self.add_calculation((OP_SQUARE, ra))
            return self.add_calculation((OP_MUL, ra, rb) if ra <= rb else (OP_MUL, rb, ra))
        if t == "scaled":
            f = e[2] % FR_MODULUS
            if f == 0:
                return zero
            if f == 1:
                return self.add_expression(e[1])
            cst = self.add_constant(f)
            ra = self.add_expression(e[1])
            return self.add_calculation((OP_MUL, ra, cst))
        raise ValueError(t)

    # -- forms handed to checkers / the accelerator -------------------------------------------
    def to_model(self):
        """dict form (constants, rotations, calculations, num_intermediates)."""
        return {"constants": list(self.constants), "rotations": list(self.rotations),
                "calculations": list(self.calculations), "num_intermediates": self.num_intermediates}

    def wire_words(self):
        """u32 calculation stream of include/zkhip.h (`zkhip_graph.calc_words`)."""
        w = []
        for c in self.calculations:
            op = c[0]
            w.append(op)
            if op == OP_HORNER:
                start, parts, factor = c[1], c[2], c[3]
                w += list(start) + list(factor) + [len(parts)]
                for p in parts:
                    w += list(p)
            else:
                for vs in c[1:]:
                    w += list(vs)
        return w


class Evaluator:
    """`plonk::evaluation::Evaluator::new(cs)`: one graph for all custom gates, one per lookup."""

    def __init__(self, cs: ConstraintSystem):
        self.custom_gates = GraphEvaluator()
        parts = []
        for gate in cs.gates:
            for poly in gate:
                parts.append(self.custom_gates.add_expression(poly))
        self.custom_gates.add_calculation((OP_HORNER, (VS_PREVIOUS, 0, 0), tuple(parts), (VS_Y, 0, 0)))
        self.lookups = []
        for lk in cs.lookups:
            g = GraphEvaluator()

            def lc(exprs):
                ps = tuple(g.add_expression(x) for x in exprs)
                return g.add_calculation((OP_HORNER, (VS_CONSTANT, 0, 0), ps, (VS_THETA, 0, 0)))

            cin = lc(lk.input_expressions)
            ctb = lc(lk.table_expressions)
            right_gamma = g.add_calculation((OP_ADD, ctb, (VS_GAMMA, 0, 0)))
            lcb = g.add_calculation((OP_ADD, cin, (VS_BETA, 0, 0)))
            g.add_calculation((OP_MUL, lcb, right_gamma))
            self.lookups.append(g)
