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
