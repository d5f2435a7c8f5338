"""CPU suite, part 2: host logic — Expression -> GraphEvaluator lowering, ConstraintSystem quantities, the wire
format, and the multi-GPU partition helpers."""
import os
import random
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import bn254 as bn
import halo2_model as hm
from zkevm_circuits_amd import plonk, sharding, synthetic

P = bn.FR_MODULUS


def _direct(e, cols, ch, idx, rs, size):
    t = e[0]
    if t == "const":
        return e[1]
    if t in ("fixed", "advice", "instance"):
        return cols[t][e[1]][(idx + e[2] * rs) % size]
    if t == "challenge":
        return ch[e[1]]
    if t == "neg":
        return (-_direct(e[1], cols, ch, idx, rs, size)) % P
    if t == "sum":
        return (_direct(e[1], cols, ch, idx, rs, size) + _direct(e[2], cols, ch, idx, rs, size)) % P
    if t == "prod":
        return _direct(e[1], cols, ch, idx, rs, size) * _direct(e[2], cols, ch, idx, rs, size) % P
    return _direct(e[1], cols, ch, idx, rs, size) * e[2] % P


def test_add_expression_matches_direct_evaluation():
    rnd = random.Random(3)
    cs = synthetic.synthetic_cs(num_advice=5, num_fixed=3, num_gate_polys=24, seed=9)
    size, rs = 64, 4
    cols = {k: [[rnd.randrange(P) for _ in range(size)] for _ in range(n)]
            for k, n in (("fixed", 3), ("advice", 5), ("instance", 1))}
    ch = [rnd.randrange(P)]
    inp = hm.EvalInputs(cols["fixed"], cols["advice"], cols["instance"], ch, 11, 13, 17, 19)
    for gate in cs.gates:
        for poly in gate:
            g = plonk.GraphEvaluator()
            g.add_expression(poly)
            prog = g.to_model()
            for idx in (0, 1, 17, 63):
                assert hm.graph_evaluate(prog, inp, 0, idx, rs, size) == _direct(poly, cols, ch, idx, rs, size)


def test_constant_folding_rules():
    g = plonk.GraphEvaluator()
    A = plonk.Advice
    assert g.add_expression(plonk.Product(plonk.Constant(0), A(0))) == (plonk.VS_CONSTANT, 0, 0)
    one_times = g.add_expression(plonk.Product(plonk.Constant(1), A(0)))
    assert one_times == g.add_expression(A(0))                      # 1 * a = a, and Store is de-duplicated
    dbl = g.add_expression(plonk.Product(plonk.Constant(2), A(0)))
    assert g.calculations[dbl[1]][0] == plonk.OP_DOUBLE
    sq = g.add_expression(plonk.Product(A(0), A(0)))
    assert g.calculations[sq[1]][0] == plonk.OP_SQUARE
    sub = g.add_expression(plonk.Sub(A(0), A(1)))
    assert g.calculations[sub[1]][0] == plonk.OP_SUB
    n = len(g.calculations)
    g.add_expression(plonk.Sub(A(0), A(1)))                          # common sub-expression: nothing new
    assert len(g.calculations) == n
    assert g.add_expression(plonk.Scaled(A(0), 0)) == (plonk.VS_CONSTANT, 0, 0)
    assert g.add_expression(plonk.Neg(plonk.Constant(5))) == (plonk.VS_CONSTANT, g.constants.index(P - 5), 0)


def test_constraint_system_quantities():
    cs = synthetic.synthetic_cs()
    assert cs.degree() == 5                    # max gate degree 5; lookups need max(4, 2 + 1 + 1) = 4
    assert cs.permutation_chunk_len() == 3
    assert cs.num_permutation_sets() == 2
    assert cs.blinding_factors() >= 5 and cs.minimum_rows() == cs.blinding_factors() + 3
    lk = plonk.Lookup([plonk.Product(plonk.Advice(0), plonk.Product(plonk.Advice(1), plonk.Advice(2)))], [plonk.Fixed(0)])
    assert lk.required_degree() == 2 + 3 + 1
    # the reference's core lookups: degree-5 condition * degree-1 entry against a degree-1 table -> degree 9 -> e = 3
    cond = plonk.Advice(0)
    for i in range(4):
        cond = plonk.Product(cond, plonk.Advice(i + 1))
    assert plonk.Lookup([plonk.Product(cond, plonk.Advice(6))], [plonk.Advice(7)]).required_degree() == 9
    assert hm.EvaluationDomain(9, 19).extended_k == 22


def test_blinding_factors_count_the_equality_query():
    """a permutation column queried by gates only at rotations 1, 2, 3 is also queried at rotation 0 by enable_equality:
    four distinct queries, blinding_factors = 4 + 2"""
    cs = plonk.ConstraintSystem(num_fixed_columns=1, num_advice_columns=2)
    e = plonk.Advice(0, 1)
    for r in (2, 3):
        e = plonk.Sum(e, plonk.Advice(0, r))
    cs.gates = [[plonk.Product(plonk.Fixed(0), e)]]
    assert cs.blinding_factors() == 5
    cs.permutation_columns = [(plonk.ANY_ADVICE, 0)]
    assert cs.blinding_factors() == 6
    cs.permutation_columns = [(plonk.ANY_ADVICE, 1), (plonk.ANY_FIXED, 0)]
    assert cs.blinding_factors() == 5


def test_wire_format_roundtrip():
    cs = synthetic.synthetic_cs(num_gate_polys=8)
    ev = plonk.Evaluator(cs)
    words = ev.custom_gates.wire_words()
    # re-parse the stream and compare with the calculation list
    pos, parsed = 0, []
    while pos < len(words):
        op = words[pos]
        pos += 1
        if op == plonk.OP_HORNER:
            start, factor, npart = tuple(words[pos:pos + 3]), tuple(words[pos + 3:pos + 6]), words[pos + 6]
            pos += 7
            parts = tuple(tuple(words[pos + 3 * i:pos + 3 * i + 3]) for i in range(npart))
            pos += 3 * npart
            parsed.append((op, start, parts, factor))
        else:
            nsrc = 2 if op in (plonk.OP_ADD, plonk.OP_SUB, plonk.OP_MUL) else 1
            parsed.append((op,) + tuple(tuple(words[pos + 3 * i:pos + 3 * i + 3]) for i in range(nsrc)))
            pos += 3 * nsrc
    assert parsed == [tuple(c) for c in ev.custom_gates.calculations]
    last = ev.custom_gates.calculations[-1]
    assert last[0] == plonk.OP_HORNER and last[1] == (plonk.VS_PREVIOUS, 0, 0) and last[3] == (plonk.VS_Y, 0, 0)
    assert len(ev.lookups) == len(cs.lookups)
    assert ev.lookups[0].calculations[-1][0] == plonk.OP_MUL


def test_partition_helpers():
    for n, world in ((10, 1), (10, 3), (7, 8), (524288, 8), (5, 2)):
        cover = []
        for r in range(world):
            first, count = sharding.base_range_for_rank(n, r, world)
            cover += list(range(first, first + count))
        assert cover == list(range(n))
        cols = sorted(c for r in range(world) for c in sharding.columns_for_rank(n if n < 100 else 97, r, world))
        assert cols == list(range(n if n < 100 else 97))
    assert sharding.cosets_for_rank(21, 19, 1, 2) == [1, 3]
    assert sorted(sum((sharding.cosets_for_rank(22, 19, r, 8) for r in range(8)), [])) == list(range(8))
    # coset partition is rotation-closed: (idx + rot * 2^e) mod 2^ext_k stays in idx's coset
    e, ek = 2, 8
    for idx in range(1 << ek):
        for rot in (-5, -1, 1, 3):
            assert ((idx + rot * (1 << e)) % (1 << ek)) % (1 << e) == idx % (1 << e)


def test_coset_partition():
    """how ranks share the cosets of the extended domain: whole cosets up to 2^e ranks, row ranges of a coset beyond,
    ranks past the largest power of two own nothing; the pieces tile the domain exactly once"""
    for ek, k, world in ((22, 19, 1), (22, 19, 2), (22, 19, 8), (21, 19, 8), (21, 19, 3), (21, 19, 5), (20, 19, 8), (12, 10, 64)):
        parts = sharding.coset_partition(ek, k, world)
        assert len(parts) == world
        e, n = ek - k, 1 << k
        owners = [p for p in parts if p is not None]
        assert len(owners) == min(1 << (world.bit_length() - 1), (1 << e) * max(1, n // 256))
        assert all(p is None for p in parts[len(owners):])
        seen = set()
        for cosets, first, count in owners:
            assert count == (1 << ek) // len(owners)
            for q in range(first, first + count):             # compact row q of the cosets' layout
                seen.add((q // len(cosets)) * (1 << e) + cosets[q % len(cosets)]) if ek <= 12 else None
        if ek <= 12:
            assert seen == set(range(1 << ek))
    assert sharding.coset_partition(21, 19, 8)[5] == ([1], 1 << 18, 1 << 18)
    assert sharding.coset_partition(22, 19, 2)[1] == ([1, 3, 5, 7], 0, 4 << 19)
    assert sharding.coset_partition(21, 19, 3) == [([0, 2], 0, 2 << 19), ([1, 3], 0, 2 << 19), None]


def test_shplonk_intermediate_sets_match_model():
    import halo2_model as hm
    from zkevm_circuits_amd import shplonk
    rng = np.random.default_rng(3)
    pts = [int(x) for x in rng.integers(1, 2 ** 62, size=4)]
    queries = [(0, pts[0], 11), (1, pts[0], 12), (1, pts[1], 13), (2, pts[1], 14), (2, pts[0], 15), (3, pts[2], 16),
               (0, pts[0], 11), (4, pts[3], 17), (4, pts[2], 18), (5, pts[2], 19)]
    a = hm.construct_intermediate_sets(queries)
    b = shplonk.construct_intermediate_sets([shplonk.Query(*q) for q in queries])
    assert a == b
    sets, sup = b
    assert sup == sorted(set(pts))
    assert [[key for key, _ in cs] for _, cs in sets] == [[0], [1, 2], [3, 5], [4]]
    assert sets[1][0] == sorted([pts[0], pts[1]]) and dict(sets[1][1])[2] == [15, 14] if pts[0] < pts[1] else [14, 15]
    for points in ([5], [5, 9], [2, 3, 11, 17]):
        ev = [int(x) for x in rng.integers(0, 2 ** 62, size=len(points))]
        assert shplonk.lagrange_interpolate(points, ev) == hm.lagrange_interpolate(points, ev)


def test_constraint_system_phases_and_memoised_degree():
    from zkevm_circuits_amd import plonk
    from zkevm_circuits_amd.plonk import Advice, Fixed, Product
    cs = plonk.ConstraintSystem(num_fixed_columns=1, num_advice_columns=3, num_challenges=3,
                                advice_column_phase=[0, 1, 0], challenge_phase=[0, 0, 1])
    assert cs.phases() == [(0, [0, 2], [0, 1]), (1, [1], [2])]
    assert plonk.ConstraintSystem(num_advice_columns=2, num_challenges=1).phases() == [(0, [0, 1], [0])]
    with pytest.raises(ValueError):
        plonk.ConstraintSystem(num_advice_columns=2, advice_column_phase=[0]).phases()
    # degree() / blinding_factors() are cached and follow the gate list when it grows
    assert cs.degree() == 3 and cs.blinding_factors() == 5
    poly = Fixed(0)
    for rot in range(-3, 3):
        poly = Product(poly, Advice(0, rot))
    cs.gates.append([poly])
    assert cs.degree() == 7 and cs.blinding_factors() == 6 + 2
    cs.lookups.append(plonk.Lookup([poly], [Fixed(0)]))
    assert cs.degree() == 2 + 7 + 1


def test_collect_queries_order():
    from zkevm_circuits_amd import plonk, prover
    from zkevm_circuits_amd.plonk import Advice, Fixed, Instance, Product, Sum
    cs = plonk.ConstraintSystem(num_fixed_columns=2, num_advice_columns=3, num_instance_columns=1)
    cs.gates = [[Product(Fixed(1), Sum(Advice(2, 1), Advice(0))), Product(Advice(2, 1), Instance(0, -1))]]
    cs.lookups = [plonk.Lookup([Advice(0)], [Fixed(0)])]
    cs.permutation_columns = [(plonk.ANY_ADVICE, 1), (plonk.ANY_INSTANCE, 0), (plonk.ANY_ADVICE, 0)]
    adv, fix, inst = prover.collect_queries(cs)
    assert adv == [(2, 1), (0, 0), (1, 0)] and fix == [(1, 0), (0, 0)] and inst == [(0, -1), (0, 0)]


def _eval_on_row(e, cols, challenges, i, n):
    t = e[0]
    if t == "const":
        return e[1]
    if t in ("fixed", "advice", "instance"):
        return cols[t][e[1]][(i + e[2]) % n]
    if t == "challenge":
        return challenges[e[1]]
    if t == "neg":
        return -_eval_on_row(e[1], cols, challenges, i, n) % P
    if t == "scaled":
        return _eval_on_row(e[1], cols, challenges, i, n) * e[2] % P
    a, b = _eval_on_row(e[1], cols, challenges, i, n), _eval_on_row(e[2], cols, challenges, i, n)
    return (a + b) % P if t == "sum" else a * b % P


def test_standin_witness_satisfies_its_constraint_system():
    """what `MockProver::assert_satisfied` checks, for the SuperCircuit stand-in of the benchmark and the GPU suite: every
    gate polynomial vanishes on every usable row (the unusable rows hold random values, as the prover's blinding rows
    do), every lookup input row is a row of its table, the copy constraints hold; the shape carries the keccak rotation
    set, 32 blinding factors, wide lookups into advice tables and second-phase columns"""
    from zkevm_circuits_amd.synthetic import fr_from_limbs_rows
    k = 10
    n = 1 << k
    # lookup copies and a permutation argument over the first two units only, as `supercircuit_standin` sizes them
    cs = synthetic.standin_cs(units=3, lanes=2, wide_lookups=2, lookup_width=8, extra_gates=3, lookup_copies=[3, 2, 2], perm_units=2)
    assert cs.degree() == 9 and cs.blinding_factors() == 32 and cs.num_challenges == 3
    assert len(cs.lookups) == 7 + 2 and len(cs.permutation_columns) == 3 * 2 + 1 + 2
    assert [len(c) for _, c, _ in cs.phases()] == [cs.num_advice_columns - 2, 2]
    rots = {r for g in cs.gates for p in g for (t, _, r) in _queries(p) if t == "advice"}
    assert {204, -12, 11} <= rots
    assert max(len(lk.input_expressions) for lk in cs.lookups) == 8
    advice_for, instance, fixed, asm = synthetic.standin_witness(cs, n, 5, active_rows=400)
    usable = n - (cs.blinding_factors() + 1)
    challenges = [random.Random(9).randrange(P) for _ in range(3)]
    first, second = advice_for(0, [0, 0, 0]), advice_for(1, challenges)
    assert all(np.array_equal(a, b) for a, b in zip(first[:-2], second[:-2])) and not first[-1].any()
    rnd = random.Random(1)
    cols = {"advice": [fr_from_limbs_rows(a) + [rnd.randrange(P) for _ in range(n - usable)] for a in second],
            "fixed": [[v % P for v in c] for c in fixed],
            "instance": [list(c) + [0] * (n - len(c)) for c in instance]}
    for gate in cs.gates:
        for poly in gate:
            assert all(_eval_on_row(poly, cols, challenges, i, n) == 0 for i in range(usable))
    for lk in cs.lookups:
        table = {tuple(_eval_on_row(e, cols, challenges, i, n) for e in lk.table_expressions) for i in range(usable)}
        assert all(tuple(_eval_on_row(e, cols, challenges, i, n) for e in lk.input_expressions) in table for i in range(usable))
    name = {plonk.ANY_ADVICE: "advice", plonk.ANY_FIXED: "fixed", plonk.ANY_INSTANCE: "instance"}
    for c, (kind, idx) in enumerate(cs.permutation_columns):
        for r in range(usable):
            c2, r2 = int(asm.map_col[c][r]), int(asm.map_row[c][r])
            k2, i2 = cs.permutation_columns[c2]
            assert cols[name[kind]][idx][r] == cols[name[k2]][i2][r2]


def _queries(e):
    out = set()
    plonk._walk_queries(e, out)
    return out


def test_rng_batched_draws_equal_single_draws():
    """`FieldRng.fr_many` (what the prover uses for the blinding rows) continues the stream exactly like repeated `fr`"""
    from zkevm_circuits_amd import prover
    a, b = prover.FieldRng(11), prover.FieldRng(11)
    seq = [a.fr() for _ in range(40)]
    assert b.fr_many(7) + [b.fr()] + b.fr_many(32) == seq
    assert a.fr() == b.fr()
    assert len(prover.OsFieldRng().fr_many(5)) == 5 and all(0 <= v < prover.P for v in prover.OsFieldRng().fr_many(5))


def test_keygen_polys_of_a_rank_without_cosets_makes_no_empty_device_call():
    """world sizes that are no power of two leave ranks that own no coset (`sharding.coset_partition`): their key keeps
    the 2^k-sized forms and no extended vector, and never asks the device for a zero-sized block or an empty coset list
    (which the library rejects -- on that rank alone, while the owners wait in the next collective)."""
    import numpy as np
    from zkevm_circuits_amd import keygen, sharding

    parts = sharding.coset_partition(12, 10, 3)
    assert parts[2] is None and parts[0] is not None and parts[1] is not None

    class FakeCtx:
        def __init__(self):
            self.next, self.live = 0x1000, {}

        def alloc(self, nbytes):
            assert nbytes > 0, "zero-sized device block requested"
            self.next += 0x100000
            self.live[self.next] = nbytes
            return self.next

        def upload(self, arr, dptr=None):
            return dptr if dptr is not None else self.alloc(np.asarray(arr).nbytes)

        def free(self, p):
            assert p in self.live, "freed a block that was never allocated"
            del self.live[p]

        def sync(self):
            pass

        def poly_lincomb_dev(self, *a):
            pass

    class FakeDom:
        n, extended_len = 1 << 10, 1 << 12

        def lagrange_to_coeff_dev(self, *a):
            pass

        def permutation_sigma_dev(self, *a):
            pass

        def coeff_to_extended_dev(self, *a):
            raise AssertionError("a sharded key never forms a whole extended vector")

        def coeff_to_cosets_dev(self, d_poly, cosets, d_out):
            assert len(cosets) > 0, "empty coset list sent to the device"

    n = FakeDom.n
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(2)]
    for cosets in ([], [1, 3]):
        ctx = FakeCtx()
        asm = keygen.Assembly(n, 2)
        pk = keygen.keygen_polys(ctx, FakeDom(), fixed, asm, 5, cosets)
        assert pk.cosets == cosets
        if not cosets:
            assert pk.fixed_cosets == [0, 0] and pk.permutation_cosets == [0, 0] and (pk.l0, pk.l_last, pk.l_active_row) == (0, 0, 0)
        assert 0 not in pk.pointers() and sorted(pk.pointers()) == sorted(ctx.live)
        for p in pk.pointers():
            ctx.free(p)
        assert not ctx.live


def test_supercircuit_census_is_current_and_sizes_the_standins():
    """tests/golden/supercircuit_census.json is what tools/supercircuit_census.py writes from the reference's sources (read
    as text; skipped where the reference tree is absent), and the stand-ins built from it land on its counts"""
    import json
    import subprocess
    import sys
    committed = json.load(open(os.path.join(ROOT, "tests", "golden", "supercircuit_census.json")))
    if os.path.isdir("/root/reference/zkevm-circuits/src"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "supercircuit_census.py"), "/root/reference"],
                             capture_output=True, text=True, check=True)
        assert json.loads(out.stdout) == committed
    assert committed["execution_gadgets"]["gadgets"] == 78
    # every counted site carries its file:line
    for sub in committed["subcircuits"].values():
        assert all(re.fullmatch(r"[\w/.-]+\.rs:\d+", e["at"]) for e in sub["sites"])
    for feature in ("standard", "fast_test"):
        cs, shape = synthetic.supercircuit_standin(feature)
        census = shape["census"]
        assert abs(cs.num_advice_columns - census["advice"]) <= 5 and len(cs.lookups) == census["lookups"]
        assert cs.degree() == census["degree"] and cs.num_challenges == census["challenges"]
        assert len(cs.permutation_columns) <= census["permutation_columns"]
    std = synthetic.supercircuit_shape("standard")
    assert 340 <= std["advice"] <= 380 and 170 <= std["lookups"] <= 230 and std["max_rotation"] == 204


def test_chacha20_keystream_and_per_thread_random_polynomial():
    """the ChaCha20 block function against the all-zero-key known answer (block 0 of the original 64-bit-counter variant,
    which `rand_chacha::ChaCha20Rng::from_seed([0; 32])` starts with), and upstream's chunking of the random polynomial:
    threads (+ 1) streams seeded in order from the main RNG, so the polynomial depends on the thread count"""
    from zkevm_circuits_amd import prover
    w = prover.chacha20_blocks(bytes(32), 0, 3)
    assert w[0].astype("<u4").tobytes().hex() == (
        "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    # consecutive blocks are one stream: block 1 computed alone equals row 1 of a batch, and differs from block 0
    assert np.array_equal(prover.chacha20_blocks(bytes(32), 1, 1)[0], w[1]) and not np.array_equal(w[0], w[1])
    assert np.array_equal(prover.chacha20_blocks(bytes(32), (1 << 32) - 1, 2)[1], prover.chacha20_blocks(bytes(32), 1 << 32, 1)[0])
    n = 64
    polys = {t: prover.random_poly_per_thread_streams(prover.FieldRng(5), n, t) for t in (1, 3, 4)}
    assert all(p.shape == (n, 4) for p in polys.values())
    assert not np.array_equal(polys[1], polys[3]) and not np.array_equal(polys[3], polys[4])
    # one thread: one stream seeded with the RNG's first 32 bytes, coefficient i = block i of it reduced mod r
    seed = prover.FieldRng(5).fill_bytes(32)
    blk = prover.chacha20_blocks(seed, 0, n)
    want = [int.from_bytes(blk[i].astype("<u4").tobytes(), "little") % P for i in range(n)]
    assert [prover.fr_from_mont_limbs(r) for r in polys[1]] == want
    # three threads over 64 coefficients: chunks of 21, a fourth stream for the last coefficient; the main RNG has handed
    # out four seeds afterwards
    rng = prover.FieldRng(5)
    prover.random_poly_per_thread_streams(rng, n, 3)
    ref = prover.FieldRng(5)
    seeds = [ref.fill_bytes(32) for _ in range(4)]
    assert rng.fill_bytes(8) == ref.fill_bytes(8)
    last = prover.chacha20_blocks(seeds[3], 0, 1)[0]
    assert prover.fr_from_mont_limbs(polys[3][63]) == int.from_bytes(last.astype("<u4").tobytes(), "little") % P


def test_pinned_verifying_key_text_and_query_lists():
    """`transcript_repr` hashes the `{:?}` text of the pinned verifying key, which spells out the whole constraint system: the
    text of a two-column circuit, its sensitivity to a gate and to the query order, and the explicit query lists"""
    from zkevm_circuits_amd import prover
    from zkevm_circuits_amd.plonk import Advice, ConstraintSystem, Fixed, Lookup, Product, Sum
    cs = ConstraintSystem(num_fixed_columns=1, num_advice_columns=2, num_instance_columns=0)
    cs.gates = [[Product(Fixed(0), Sum(Advice(0), Advice(1, 1)))]]
    cs.lookups = [Lookup([Advice(1)], [Fixed(0)])]
    cs.permutation_columns = [(plonk.ANY_ADVICE, 0)]
    text = prover.pinned_vk_debug(cs, 3, 5, 7, [(1, 2)], [None])
    q = "0x" + "0" * 63
    assert text == (
        'PinnedVerificationKey { base_modulus: "0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47", '
        'scalar_modulus: "0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", '
        f'domain: PinnedEvaluationDomain {{ k: 3, extended_k: 5, omega: {q}7 }}, cs: PinnedConstraintSystem {{ '
        'num_fixed_columns: 1, num_advice_columns: 2, num_instance_columns: 0, num_selectors: 0, num_challenges: 0, '
        'advice_column_phase: [0, 0], challenge_phase: [], gates: [Product(Fixed { query_index: 0, column_index: 0, rotation: '
        'Rotation(0) }, Sum(Advice { query_index: 0, column_index: 0, rotation: Rotation(0), phase: Phase(0) }, Advice { '
        'query_index: 1, column_index: 1, rotation: Rotation(1), phase: Phase(0) }))], advice_queries: [(Column { index: 0, '
        'column_type: Advice }, Rotation(0)), (Column { index: 1, column_type: Advice }, Rotation(1)), (Column { index: 1, '
        'column_type: Advice }, Rotation(0))], instance_queries: [], fixed_queries: [(Column { index: 0, column_type: Fixed }, '
        'Rotation(0))], permutation: Argument { columns: [Column { index: 0, column_type: Advice }] }, lookups: [Argument { '
        'input_expressions: [Advice { query_index: 2, column_index: 1, rotation: Rotation(0), phase: Phase(0) }], '
        'table_expressions: [Fixed { query_index: 0, column_index: 0, rotation: Rotation(0) }] }], constants: [], '
        f'minimum_degree: None }}, fixed_commitments: [({q}1, {q}2)], permutation: VerifyingKey {{ commitments: [Infinity] }} }}')
    base = prover._vk_digest(cs, 3, 5, 7, [(1, 2)], [None])
    other = ConstraintSystem(num_fixed_columns=1, num_advice_columns=2)
    other.gates = [[Product(Fixed(0), Sum(Advice(0), Advice(1, -1)))]]
    other.lookups, other.permutation_columns = cs.lookups, cs.permutation_columns
    assert prover._vk_digest(other, 3, 5, 7, [(1, 2)], [None]) != base          # the digest binds the gates
    assert prover._vk_digest(cs, 3, 5, 7, [(1, 3)], [None]) != base
    # the circuit's own query order: (1, 0) first, as if it had enabled equality on column 1 before building its gate
    cs2 = ConstraintSystem(num_fixed_columns=1, num_advice_columns=2)
    cs2.gates, cs2.lookups, cs2.permutation_columns = cs.gates, cs.lookups, cs.permutation_columns
    cs2.advice_queries = [(1, 0), (0, 0), (1, 1)]
    assert prover.collect_queries(cs2)[0] == [(1, 0), (0, 0), (1, 1)]
    assert prover.collect_queries(cs)[0] == [(0, 0), (1, 1), (1, 0)]
    assert prover._vk_digest(cs2, 3, 5, 7, [(1, 2)], [None]) != base
    import verifier_model as vm
    assert vm.collect_queries(cs2)[0] == [(1, 0), (0, 0), (1, 1)] and vm.collect_queries(cs)[0] == [(0, 0), (1, 1), (1, 0)]
    cs2.advice_queries = [(1, 0), (0, 0)]
    cs2.__dict__.pop("_memo_cache", None)
    with pytest.raises(ValueError):
        prover.collect_queries(cs2)


def test_split_extension_runs_and_column_flags_without_a_device():
    """`evalh_host.extend_for_split` sends maximal runs of equally flagged polynomials to the whole / the half-domain
    extension with the right addresses, and `columns_used` cuts the library's flat flag list into the constraint
    system's column groups (a recording stand-in takes the place of the device domain)"""
    from zkevm_circuits_amd import evalh_host

    class Dom:
        n, extended_len = 8, 32
        calls = []

        def coeff_to_extended_split_dev(self, src, dst, batch):
            self.calls.append(("whole", src, dst, batch))

        def coeff_to_extended_half_dev(self, src, dst, stride, batch):
            self.calls.append(("half", src, dst, stride, batch))

    dom = Dom()
    evalh_host.extend_for_split(dom, [True, True, False, True, False, False], 1000, 50000)
    n32, en32 = 8 * 32, 32 * 32
    assert dom.calls == [("whole", 1000, 50000, 2), ("half", 1000 + 2 * n32, 50000 + 2 * en32, 32, 1),
                         ("whole", 1000 + 3 * n32, 50000 + 3 * en32, 1), ("half", 1000 + 4 * n32, 50000 + 4 * en32, 32, 2)]
    dom.calls.clear()
    evalh_host.extend_for_split(dom, [], 0, 0)
    assert dom.calls == []

    cs = synthetic.satisfied_cs(2, degree9=True, plain_lookups=True)
    sets = cs.num_permutation_sets()
    total = (cs.num_fixed_columns + cs.num_advice_columns + cs.num_instance_columns + 3 + 1 + sets
             + len(cs.permutation_columns) + 3 * len(cs.lookups))

    class FlagDom:
        def evaluate_h_columns_used(self, desc):
            assert desc.term_select == 2 and desc.term_degree_max == 4 and desc.num_lookups == len(cs.lookups)
            return np.arange(total) % 2

    used = evalh_host.columns_used(FlagDom(), cs, plonk.Evaluator(cs), (2, 4))
    flat = (used["fixed"] + used["advice"] + used["instance"] + used["l"] + ["x"] + used["perm_z"] + used["perm_sigma"]
            + [f for t in used["lookups"] for f in t])
    assert len(flat) == total and len(used["lookups"]) == len(cs.lookups) and len(used["perm_z"]) == sets
    assert [f for f in flat if f != "x"] == [bool(i % 2) for i in range(total) if flat[i] != "x"]
