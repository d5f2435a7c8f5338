"""Synthetic constraint systems shaped like the reference's circuits, for parity tests and the benchmark.

The reference's SuperCircuit (/root/reference/zkevm-circuits/src/super_circuit.rs:59-206) is authored in Rust
against halo2 and cannot be instantiated here, so the evaluate_h workload is parameterised by the same
quantities its ConstraintSystem exposes: advice / fixed / instance column counts, gate polynomials with their
degree and rotation set (core gates query rotations -5..+1, /root/reference/zkevm-circuits/src/execution.rs:2157-2159),
multi-column `lookup_any` arguments (/root/reference/zkevm-circuits/src/execution.rs:2305-2334) and the
permutation columns.
"""
from __future__ import annotations

import json
import os
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


def satisfied_cs(units, degree9=False, extra_gates=0, lookup_copies=1, perm_units=None, plain_lookups=False):
    """A constraint system that `satisfied_witness` satisfies, for proofs that a verifier can check: `units` copies side
    by side of five advice columns a, b, c, s, d with the gates q (a b - c), a challenge multiple of it,
    q_acc (s(+1) - s - c) and q_prev (d - a(-1) - b), one lookup of (q a, q b) in a fixed table of squares per unit, three
    permutation columns per unit plus the shared instance column; with degree9 one gate polynomial of degree nine
    (extended domain 8n).
    lookup_copies: how often a unit declares its lookup (an int, or one count per unit): halo2 does not merge equal lookup
    arguments, so every copy costs the prover a compression, a sort, a grand product and three commitments -- the way to
    give a stand-in the lookup count of a real circuit without more columns.  perm_units: only the first that many units
    put their columns into the permutation argument (None: all), for circuits with few equality-enabled columns.
    plain_lookups: the lookups read (a, b) on every row, without the selector (the witness has (0, 0), a table row, where
    q = 0): inputs of degree one, as cell-manager lookups have them, so the lookup's product term has degree four."""
    copies = [lookup_copies] * units if isinstance(lookup_copies, int) else list(lookup_copies)
    assert len(copies) == units
    perm_units = units if perm_units is None else min(perm_units, units)
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
        for _ in range(copies[u]):
            cs.lookups.append(Lookup([a, b] if plain_lookups else [Product(q, a), Product(q, b)], [Fixed(2), Fixed(3)]))
        if u < perm_units:
            cs.permutation_columns += [(ANY_ADVICE, 5 * u + 3), (ANY_ADVICE, 5 * u), (ANY_ADVICE, 5 * u + 2)]
    cs.permutation_columns.append((ANY_INSTANCE, 0))
    return cs


def satisfied_witness(cs, n, seed, units, active_rows=500, as_arrays=False, zeros=None):
    """(advice, instance, fixed, copy-constraint assembly) for `satisfied_cs(units, ..)` on n rows: `active_rows` rows per
    unit carry b = a^2, c = a b and the running sum s, tied to the public inputs and to each other by copy constraints.
    as_arrays: advice columns as (usable, 4) Montgomery limb arrays (zero below the active rows) instead of lists."""
    usable = n - (cs.blinding_factors() + 1)
    m = min(usable - 2, active_rows)
    rng = np.random.default_rng(seed)
    advice, public = [], []
    from .keygen import Assembly
    asm = Assembly(n, len(cs.permutation_columns))       # argument columns: (s, a, c) per unit, then the instance
    perm_units = sum(1 for kind, _ in cs.permutation_columns[:3 * units] if kind == ANY_ADVICE) // 3
    perm_units = min(perm_units, (len(cs.permutation_columns) - 1) // 3)
    ic = next(i for i, (kind, _) in enumerate(cs.permutation_columns) if kind == ANY_INSTANCE)
    perm_units = min(perm_units, ic // 3)
    inst_col = 3 * perm_units
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
        if u >= perm_units:
            continue
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
            arr = zeros((usable, 4)) if zeros else np.zeros((usable, 4), dtype=np.uint64)
            arr[:m + 1] = fr_limbs(col[:m + 1])
            assert not any(col[m + 1:])
            arrays.append(arr)
        advice = arrays
    return advice, [public], fixed, asm


# ----------------------------------------------------------------------------- SuperCircuit stand-in
# Rotation set of the reference's keccak cell-manager columns: 0..11 inside a round, (i + 1) * 12 for i < 17 up to +204,
# and one round back (/root/reference/zkevm-circuits/src/keccak_circuit.rs:148,158-162,
# /root/reference/zkevm-circuits/src/keccak_circuit/param.rs:1-2).
KECCAK_ROTATIONS = tuple(range(12)) + tuple(12 * (i + 1) for i in range(17)) + (-12,)


def standin_cs(units=52, lanes=8, wide_lookups=8, lookup_width=10, extra_gates=0, lookup_copies=1, perm_units=None):
    """A satisfiable stand-in for the reference's SuperCircuit that carries its hard features together, on top of
    `satisfied_cs(units, degree9=True, extra_gates)`:
      * `lanes` keccak-like advice columns w, each queried at the 30 rotations of KECCAK_ROTATIONS (so that
        blinding_factors = 32 and the permutation argument's last rotation is -33), with a linear gate over all of them,
        a product across the +204 rotation and a degree-nine gate;
      * `wide_lookups` lookups of `lookup_width` input columns into `lookup_width` advice table columns, the shape of the
        merged `lookup_any`s of /root/reference/zkevm-circuits/src/execution.rs:2305-2334;
      * two second-phase advice columns and three challenges usable after the first phase
        (/root/reference/zkevm-circuits/src/core_circuit.rs:153-154, /root/reference/zkevm-circuits/src/util.rs:63-67).
    Column order: the units' 5 * units advice columns, per lane (w, t, u), the table columns, per wide lookup its input
    columns, the two second-phase columns.  Fixed: the 5 of satisfied_cs, then q_k, q_w, q_2."""
    cs = satisfied_cs(units, degree9=True, extra_gates=extra_gates, lookup_copies=lookup_copies, perm_units=perm_units)
    base_adv = cs.num_advice_columns
    q_k, q_w, q_2 = Fixed(5), Fixed(6), Fixed(7)
    cs.num_fixed_columns = 8
    cs.num_challenges = 3
    col = base_adv
    lane_cols = []
    for j in range(lanes):
        w, t, u = col, col + 1, col + 2
        col += 3
        lane_cols.append((w, t, u))
        lin = Advice(t)
        for i, r in enumerate(KECCAK_ROTATIONS):
            lin = Sub(lin, Scaled(Advice(w, r), i + 2 + j))
        far = Sub(Product(Advice(w, 0), Advice(w, 204)), Advice(u))
        high = Product(q_k, far)
        for i, r in enumerate((1, 5, 11, 12, 96, -12)):
            high = Product(high, Sum(Advice(w, r), Constant(7 * j + i + 1)))
        cs.gates.append([Product(q_k, lin), Product(q_k, far), high])
    table_cols = list(range(col, col + lookup_width))
    col += lookup_width
    wide_inputs = []
    for _ in range(wide_lookups):
        ins = list(range(col, col + lookup_width))
        col += lookup_width
        wide_inputs.append(ins)
        cs.lookups.append(Lookup([Product(q_w, Advice(c)) for c in ins], [Advice(c) for c in table_cols]))
    r1, r2 = col, col + 1
    col += 2
    cs.num_advice_columns = col
    x = wide_inputs[0] if wide_inputs else [0, 1, 2]
    rlc = Sub(Sub(Sub(Advice(r1), Advice(x[0])), Product(Challenge(0), Advice(x[1]))), Product(Challenge(1), Advice(x[2])))
    cs.gates.append([Product(q_2, rlc), Product(q_2, Sub(Advice(r2), Product(Challenge(2), Advice(0))))])
    cs.advice_column_phase = [0] * (col - 2) + [1, 1]
    cs.challenge_phase = [0, 0, 0]
    cs.permutation_columns += [(ANY_ADVICE, lane_cols[0][0])] if lanes else []
    cs.permutation_columns.append((ANY_ADVICE, r1))
    cs._standin = dict(units=units, lanes=lane_cols, table=table_cols, wide=wide_inputs, second=(r1, r2), x=x,
                       perm_units=units if perm_units is None else min(perm_units, units))
    return cs


CENSUS_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "supercircuit_census.json")


def supercircuit_shape(feature="standard"):
    """(A, F, L, P, ..) of the reference's SuperCircuit from the committed static census of its sources
    (tests/golden/supercircuit_census.json, written by tools/supercircuit_census.py with a file:line per counted call):
    advice / lookup counts are the middle of the census bracket (the keccak circuit allocates its cell columns and their
    lookups at run time), the permutation columns its equality-enabled sites plus the instance column and the Poseidon
    chip's state columns, rotations from the execution gadgets' NUM_ROW and the keccak round layout."""
    with open(CENSUS_PATH) as f:
        c = json.load(f)
    sh, ex = c["shape"][feature], c["execution_gadgets"]
    mid = lambda pair: (pair[0] + pair[1]) // 2                                     # noqa: E731
    return {"feature": feature, "advice": mid(sh["advice_with_keccak_cell_columns"]), "advice_second_phase": sh["advice_second_phase"],
            "fixed": sh["fixed_counted"], "instance": sh["instance"], "lookups": mid(sh["lookups_with_keccak_cell_lookups"]),
            "create_gate_calls": sh["gates_counted"], "permutation_columns": sh["enable_equality_sites"] + 1 + 10,
            "degree": 9, "challenges": 3, "max_rotation": 204 if feature == "standard" else 1,
            "min_rotation": -ex["max_NUM_ROW"], "widest_lookup": 10,
            "source": "tests/golden/supercircuit_census.json (tools/supercircuit_census.py over the reference's Rust sources)"}


def supercircuit_standin(feature="standard", extra_gates=None):
    """`standin_cs` sized from `supercircuit_shape(feature)`: units, lookup copies and permutation units are solved so
    that the advice, lookup and permutation column counts land on the census figures (within one unit's five columns).
    Returns (constraint system, shape dictionary with both the census target and what the stand-in has)."""
    shape = supercircuit_shape(feature)
    standard = feature == "standard"
    lanes = 8 if standard else 0
    wide, width = (8, 10) if standard else (3, 8)
    other = 3 * lanes + width + wide * width + 2
    units = max(1, round((shape["advice"] - other) / 5))
    unit_lookups = max(units, shape["lookups"] - wide)
    copies = [unit_lookups // units + (1 if u < unit_lookups % units else 0) for u in range(units)]
    perm_units = max(1, (shape["permutation_columns"] - 1 - (2 if lanes else 1)) // 3)
    if extra_gates is None:
        extra_gates = 38 if standard else 4
    cs = standin_cs(units=units, lanes=lanes, wide_lookups=wide, lookup_width=width, extra_gates=extra_gates,
                    lookup_copies=copies, perm_units=perm_units)
    built = {"advice": cs.num_advice_columns, "fixed": cs.num_fixed_columns, "lookups": len(cs.lookups),
             "permutation_columns": len(cs.permutation_columns), "gate_polynomials": sum(len(g) for g in cs.gates),
             "degree": cs.degree(), "blinding_factors": cs.blinding_factors(), "challenges": cs.num_challenges,
             "units": units, "lanes": lanes, "wide_lookups": wide, "lookup_width": width, "perm_units": perm_units}
    return cs, {"census": shape, "standin": built}


def standin_witness(cs, n, seed, active_rows=3000, zeros=None):
    """(advice callable for `prover.create_proof`, instance, fixed, assembly) satisfying `standin_cs`.  Advice columns are
    (usable, 4) Montgomery limb arrays, zero below the active rows; the callable fills the second-phase columns from the
    challenges it is given.  zeros: allocator of zeroed uint64 arrays for the columns (`Context.host_zeros` puts the
    witness into page-locked memory, which the device reads by DMA)."""
    zeros = zeros or (lambda shape: np.zeros(shape, dtype=np.uint64))
    from .keygen import Assembly
    from .prover import fr_limbs
    from .plonk import FR_MODULUS as P
    info = cs._standin
    units = info["units"]
    usable = n - (cs.blinding_factors() + 1)
    m = min(usable - 210, active_rows)
    assert m > 300, "not enough rows for the +204 rotation"
    perm_units = info.get("perm_units", units)
    base_cs = satisfied_cs(units, degree9=True, perm_units=perm_units)
    base_cs_bf = cs.blinding_factors()
    # the units' columns: satisfied_witness sizes its rows from the constraint system's blinding factors, so hand it ours
    shim = ConstraintSystem()
    shim.blinding_factors = lambda: base_cs_bf
    shim.permutation_columns = base_cs.permutation_columns
    advice, instance, fixed, _ = satisfied_witness(shim, n, seed, units, active_rows=m, as_arrays=True, zeros=zeros)
    rng = np.random.default_rng(seed + 1)
    act = m + 205                                         # rows that may be non-zero

    def arr(vals):
        a = zeros((usable, 4))
        a[:len(vals)] = fr_limbs(vals)
        return a

    first, last = 12, m                                   # q_k rows: first <= i < last (i - 12 >= 0, i + 204 < act)
    for j, (w, t, u) in enumerate(info["lanes"]):
        wv = [int(v) for v in rng.integers(0, 1 << 30, size=act)]
        tv, uv = [0] * act, [0] * act
        for i in range(first, last):
            tv[i] = sum((k + 2 + j) * wv[i + r] for k, r in enumerate(KECCAK_ROTATIONS)) % P
            uv[i] = wv[i] * wv[i + 204] % P
        advice += [arr(wv), arr(tv), arr(uv)]
    width = len(info["table"])
    table_rows = 64
    for j in range(width):
        advice.append(arr([i * (j + 1) for i in range(table_rows)]))       # row 0 is all zero: what inactive rows look up
    for ins in info["wide"]:
        pick = [int(v) for v in rng.integers(0, table_rows, size=m)]
        for j in range(width):
            advice.append(arr([p * (j + 1) for p in pick]))
    xs = [[int(v) for v in fr_from_limbs_rows(advice[c][:m])] for c in info["x"][:3]]
    a0 = [int(v) for v in fr_from_limbs_rows(advice[0][:m])]
    zero = zeros((usable, 4))
    fixed = [list(c) for c in fixed]
    fixed.append([1 if first <= i < last else 0 for i in range(n)])        # q_k
    fixed.append([1] * m + [0] * (n - m))                                   # q_w
    fixed.append([1] * m + [0] * (n - m))                                   # q_2
    asm = Assembly(n, len(cs.permutation_columns))
    inst_col = 3 * perm_units
    public = instance[0]
    for u in range(perm_units):                                             # the copy constraints of satisfied_witness
        asm.copy(3 * u, 0, inst_col, 2 * u)
        asm.copy(3 * u, m, inst_col, 2 * u + 1)
        asm.copy(3 * u + 1, 1, 3 * u + 1, 4)
        asm.copy(3 * u + 2, 0, 3 * u + 1, 3)
        if u:
            asm.copy(3 * u + 1, 7, 3 * (u - 1) + 1, 7)

    def advice_for(phase, challenges):
        if phase == 0:
            return advice + [zero, zero]
        c0, c1, c2 = challenges[:3]
        r1 = [(xs[0][i] + c0 * xs[1][i] + c1 * xs[2][i]) % P for i in range(m)]
        r2 = [c2 * a0[i] % P for i in range(m)]
        return advice + [arr(r1), arr(r2)]
    return advice_for, [public], fixed, asm


def fr_from_limbs_rows(rows):
    """(len, 4) Montgomery limbs -> canonical integers"""
    from .plonk import FR_MODULUS as P
    rinv = pow(1 << 256, P - 2, P)
    return [sum(int(w) << (64 * j) for j, w in enumerate(r)) * rinv % P for r in rows]
