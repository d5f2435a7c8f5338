"""The test circuit of the end-to-end prover tests: constraint system, a satisfying witness, a KZG test SRS with a
known trapdoor.  Shared by the GPU suite (which proves), the CPU suite (which verifies the committed proof fixture) and
tests/golden/gen_proof_golden.py."""
import numpy as np

import bn254 as bn
from util import affine_arr, affine_pt, fr_arr

from zkevm_circuits_amd import keygen, plonk

P = bn.FR_MODULUS
TAU = 0x1F3A9C0D5E6B7788990A1B2C3D4E5F60718293A4B5C6D7E8F9  # trapdoor of the test SRS


def circuit():
    from zkevm_circuits_amd.plonk import Advice, Challenge, Fixed, Product, Sub
    a, b, c = Advice(0), Advice(1), Advice(2)
    q, q_acc = Fixed(0), Fixed(1)
    mul = Sub(Product(a, b), c)
    cs = plonk.ConstraintSystem(num_fixed_columns=4, num_advice_columns=4, num_instance_columns=1, num_challenges=1)
    cs.gates = [[Product(q, mul), Product(q, Product(Challenge(0), mul))],
                [Product(q_acc, Sub(Sub(Advice(3, 1), Advice(3)), c))]]
    cs.lookups = [plonk.Lookup([Product(q, a), Product(q, b)], [Fixed(2), Fixed(3)])]
    cs.permutation_columns = [(plonk.ANY_ADVICE, 3), (plonk.ANY_INSTANCE, 0), (plonk.ANY_ADVICE, 0), (plonk.ANY_ADVICE, 2)]
    return cs


def witness(cs, n, seed):
    """rows < m are active: b = a^2 (lookup into the table of squares), c = a b, s accumulates c; the copy constraints
    tie s[0] and s[m] to the instance column and a few cells of a and c to each other."""
    usable = n - (cs.blinding_factors() + 1)
    m = min(usable - 2, 3000)
    rng = np.random.default_rng(seed)
    a = [int(v) for v in rng.integers(0, 11, size=m)] + [0] * (usable - m)
    a[0], a[3] = 2, 8
    a[4] = a[1]
    b = [v * v for v in a]
    c = [v * w for v, w in zip(a, b)]
    s = [5]
    for i in range(m):
        s.append(s[-1] + c[i])
    s += [0] * (usable - len(s))
    fixed = [[1] * m + [0] * (n - m), [1] * m + [0] * (n - m),
             [i if i < usable else 0 for i in range(n)], [i * i if i < usable else 0 for i in range(n)]]
    asm = keygen.Assembly(n, len(cs.permutation_columns))       # argument columns: s, instance, a, c
    asm.copy(0, 0, 1, 0)
    asm.copy(0, m, 1, 1)
    asm.copy(2, 1, 2, 4)
    asm.copy(3, 0, 2, 3)
    return [a, b, c, s], [[5, s[m]]], fixed, asm


def srs(co, n):
    """g[i] = tau^i G by the C oracle's fixed-base multiplication, spot-checked against the big-int model"""
    g = co.g1_powers_of_tau(fr_arr([TAU])[0], n, threads=min(16, co.max_threads()))
    for i in (0, 3, n - 1):
        assert affine_pt(g[i]) == bn.g1_mul(bn.G1_GENERATOR, pow(TAU, i, P))
    return g


def wide_circuit(units, degree9=False, extra_gates=0):
    """`units` copies of the circuit above side by side, each with a fifth advice column d = a(-1) + b (a gate with a
    negative rotation), one lookup per unit, three permutation columns per unit plus the shared instance column;
    with degree9 one gate polynomial of degree nine (extended domain 8n)."""
    from zkevm_circuits_amd.plonk import Advice, Challenge, Constant, Fixed, Product, Sub, Sum
    q, q_acc, q_prev = Fixed(0), Fixed(1), Fixed(4)
    cs = plonk.ConstraintSystem(num_fixed_columns=5, num_advice_columns=5 * units, num_instance_columns=1, num_challenges=1)
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
        cs.lookups.append(plonk.Lookup([Product(q, a), Product(q, b)], [Fixed(2), Fixed(3)]))
        cs.permutation_columns += [(plonk.ANY_ADVICE, 5 * u + 3), (plonk.ANY_ADVICE, 5 * u), (plonk.ANY_ADVICE, 5 * u + 2)]
    cs.permutation_columns.append((plonk.ANY_INSTANCE, 0))
    return cs


def wide_witness(cs, n, seed, units, active_rows=500, as_arrays=False):
    """as_arrays: advice columns as (usable, 4) Montgomery limb arrays (zero below the active rows) instead of lists"""
    usable = n - (cs.blinding_factors() + 1)
    m = min(usable - 2, active_rows)
    rng = np.random.default_rng(seed)
    advice, public = [], []
    asm = keygen.Assembly(n, len(cs.permutation_columns))       # argument columns: (s, a, c) per unit, then the instance
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
        arrays = []
        for col in advice:
            arr = np.zeros((usable, 4), dtype=np.uint64)
            arr[:m + 1] = fr_arr(col[:m + 1])
            assert not any(col[m + 1:])
            arrays.append(arr)
        advice = arrays
    return advice, [public], fixed, asm
