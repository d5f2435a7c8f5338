"""Synthetic constraint systems shaped like the reference's circuits, for parity tests and the benchmark.

The reference's SuperCircuit (/root/reference/zkevm-circuits/src/super_circuit.rs:59-206) is authored in Rust
against halo2 and cannot be instantiated here, so the evaluate_h workload is parameterised by the same
quantities its ConstraintSystem exposes: advice / fixed / instance column counts, gate polynomials with their
degree and rotation set (core gates query rotations -5..+1, /root/reference/zkevm-circuits/src/execution.rs:2157-2159),
multi-column `lookup_any` arguments (/root/reference/zkevm-circuits/src/execution.rs:2305-2334) and the
permutation columns.
"""
from __future__ import annotations

import random

import numpy as np

from .plonk import (ANY_ADVICE, ANY_FIXED, ANY_INSTANCE, Advice, Challenge, Constant, ConstraintSystem, Fixed,
                    Instance, Lookup, Product, Scaled, Sub, Sum)


def synthetic_cs(num_advice=8, num_fixed=4, num_instance=1, num_gate_polys=16, max_degree=5,
                 rotations=(-2, -1, 0, 1), num_lookups=2, lookup_width=3, num_perm_cols=4, num_challenges=1,
                 seed=20250523) -> ConstraintSystem:
    rnd = random.Random(seed)
    cs = ConstraintSystem(num_fixed_columns=num_fixed, num_advice_columns=num_advice,
                          num_instance_columns=num_instance, num_challenges=num_challenges)

    def adv():
        return Advice(rnd.randrange(num_advice), rnd.choice(rotations))

    def term(deg):
        e = adv()
        for _ in range(deg - 1):
            e = Product(e, adv())
        return e

    polys = []
    for i in range(num_gate_polys):
        deg = 2 + (i % (max_degree - 1))          # 2 .. max_degree
        sel = Fixed(i % num_fixed, 0)
        body = term(deg - 1)
        extra = Scaled(adv(), rnd.randrange(1, 1 << 64))
        if i % 3 == 0:
            body = Sub(body, extra)
        elif i % 3 == 1:
            body = Sum(body, Product(Constant(rnd.randrange(1, 1 << 128)), adv())) if deg >= 3 else Sum(body, extra)
        else:
            body = Sum(body, Sub(Instance(0, 0) if num_instance else adv(),
                                 Challenge(0) if num_challenges else Constant(7)))
        polys.append(Product(sel, body))
    # group polynomials into gates of 4 (the grouping does not change evaluate_h)
    for s in range(0, len(polys), 4):
        cs.gates.append(polys[s:s + 4])
    for i in range(num_lookups):
        inputs = [adv() for _ in range(lookup_width)]
        table = [Fixed((i + j) % num_fixed, 0) if j % 2 == 0 else Advice((i + j) % num_advice, 0)
                 for j in range(lookup_width)]
        cs.lookups.append(Lookup(inputs, table))
    perm = []
    for i in range(num_perm_cols):
        r = i % 4
        if r == 3 and num_fixed:
            perm.append((ANY_FIXED, i % num_fixed))
        elif r == 2 and num_instance:
            perm.append((ANY_INSTANCE, 0))
        else:
            perm.append((ANY_ADVICE, i % num_advice))
    cs.permutation_columns = perm
    return cs


def satisfied_cs(units, degree9=False, extra_gates=0):
    """A constraint system that `satisfied_witness` satisfies, for proofs that a verifier can check: `units` copies side
    by side of five advice columns a, b, c, s, d with the gates q (a b - c), a challenge multiple of it,
    q_acc (s(+1) - s - c) and q_prev (d - a(-1) - b), one lookup of (q a, q b) in a fixed table of squares per unit, three
    permutation columns per unit plus the shared instance column; with degree9 one gate polynomial of degree nine
    (extended domain 8n)."""
    q, q_acc, q_prev = Fixed(0), Fixed(1), Fixed(4)
    cs = ConstraintSystem(num_fixed_columns=5, num_advice_columns=5 * units, num_instance_columns=1, num_challenges=1)
    for u in range(units):
        a, b, c, s, d = (Advice(5 * u + j) for j in range(5))
        mul = Sub(Product(a, b), c)
        polys = [Product(q, mul), Product(q, Product(Challenge(0), mul))]
        if degree9 and u == 0:
            high = Product(q, mul)
            for term in (a, b, c, s, d, a):
                high = Product(high, Sum(term, Constant(u + 1)))
            polys.append(high)
        # extra_gates more polynomials per unit, each the satisfied q (a b - c) times up to six seeded factors
        # (column at a rotation + constant): they vanish on the circuit's rows, and their values on the extended domain,
        # which the quotient is made of, depend on every factor
        g_rng = np.random.default_rng(1000 + u)
        for g in range(extra_gates):
            poly = Product(q, mul)
            for _ in range(1 + g % 6):
                col, rot = 5 * u + int(g_rng.integers(0, 5)), int(g_rng.integers(-1, 2))
                poly = Product(poly, Sum(Advice(col, rot), Constant(int(g_rng.integers(1, 1 << 30)))))
            polys.append(poly)
        cs.gates.append(polys)
        cs.gates.append([Product(q_acc, Sub(Sub(Advice(5 * u + 3, 1), s), c)),
                         Product(q_prev, Sub(Sub(d, Advice(5 * u, -1)), b))])
        cs.lookups.append(Lookup([Product(q, a), Product(q, b)], [Fixed(2), Fixed(3)]))
        cs.permutation_columns += [(ANY_ADVICE, 5 * u + 3), (ANY_ADVICE, 5 * u), (ANY_ADVICE, 5 * u + 2)]
    cs.permutation_columns.append((ANY_INSTANCE, 0))
    return cs


def satisfied_witness(cs, n, seed, units, active_rows=500, as_arrays=False):
    """(advice, instance, fixed, copy-constraint assembly) for `satisfied_cs(units, ..)` on n rows: `active_rows` rows per
    unit carry b = a^2, c = a b and the running sum s, tied to the public inputs and to each other by copy constraints.
    as_arrays: advice columns as (usable, 4) Montgomery limb arrays (zero below the active rows) instead of lists."""
    usable = n - (cs.blinding_factors() + 1)
    m = min(usable - 2, active_rows)
    rng = np.random.default_rng(seed)
    advice, public = [], []
    from .keygen import Assembly
    asm = Assembly(n, len(cs.permutation_columns))       # argument columns: (s, a, c) per unit, then the instance
    inst_col = 3 * units
    for u in range(units):
        a = [int(v) for v in rng.integers(0, 11, size=m)] + [0] * (usable - m)
        a[0], a[3], a[7] = 2, 8, 3
        a[4] = a[1]
        b = [v * v for v in a]
        c = [v * w for v, w in zip(a, b)]
        s = [5 + u]
        for i in range(m):
            s.append(s[-1] + c[i])
        s += [0] * (usable - len(s))
        d = [0] + [a[i - 1] + b[i] for i in range(1, m)] + [0] * (usable - m)
        advice += [a, b, c, s, d]
        public += [5 + u, s[m]]
        asm.copy(3 * u, 0, inst_col, 2 * u)
        asm.copy(3 * u, m, inst_col, 2 * u + 1)
        asm.copy(3 * u + 1, 1, 3 * u + 1, 4)
        asm.copy(3 * u + 2, 0, 3 * u + 1, 3)
        if u:
            asm.copy(3 * u + 1, 7, 3 * (u - 1) + 1, 7)        # a cell shared between neighbouring units
    fixed = [[1] * m + [0] * (n - m), [1] * m + [0] * (n - m),
             [i if i < usable else 0 for i in range(n)], [i * i if i < usable else 0 for i in range(n)],
             [0] + [1] * (m - 1) + [0] * (n - m)]
    if as_arrays:
        from .prover import fr_limbs
        arrays = []
        for col in advice:
            arr = np.zeros((usable, 4), dtype=np.uint64)
            arr[:m + 1] = fr_limbs(col[:m + 1])
            assert not any(col[m + 1:])
            arrays.append(arr)
        advice = arrays
    return advice, [public], fixed, asm
