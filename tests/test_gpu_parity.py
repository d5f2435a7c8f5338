"""GPU suite: the HIP path, called through the C ABI, against the golden vectors, the C oracle on seeded inputs, and
size-independent properties at the benchmark's full sizes.  Bit-exact everywhere (integer arithmetic)."""
import os
import numpy as np
import pytest

import bn254 as bn
from util import (affine_arr, affine_pt, fq_arr, fq_ints, fr_arr, fr_ints, golden, hex_fr, hex_pt, rand_fr,
                  witness_like)

from zkevm_circuits_amd import evalh_host, ffi, plonk, synthetic
import zkevm_circuits_amd as zk

pytestmark = pytest.mark.gpu
P = bn.FR_MODULUS


# ------------------------------------------------------------------------------------------------ field / curve
@pytest.mark.parametrize("fname,fid", [("fr", 0), ("fq", 1)])
def test_field_golden_and_random(ctx, co, fname, fid):
    g = golden("field.json")["fields"][fname]
    to_arr = fr_arr if fid == 0 else fq_arr
    to_int = fr_ints if fid == 0 else fq_ints
    a = to_arr([int(c["a"], 16) for c in g["cases"]])
    b = to_arr([int(c["b"], 16) for c in g["cases"]])
    for op, key in ((0, "add"), (1, "sub"), (2, "mul")):
        assert to_int(ctx.field_binop(fid, op, a, b)) == [int(c[key], 16) for c in g["cases"]], key
    assert to_int(ctx.field_batch_invert(fid, a)) == [int(c["inv"], 16) for c in g["cases"]]
    rng = np.random.default_rng(1)
    x, y = rand_fr(rng, 20000), rand_fr(rng, 20000)
    for op in (0, 1, 2):
        assert np.array_equal(ctx.field_binop(fid, op, x, y), co.field_binop(fid, op, x, y))
    assert np.array_equal(ctx.field_binop(fid, 3, x, y), co.field_binop(fid, 2, co.field_binop(fid, 2, x, x), y))
    assert np.array_equal(ctx.field_binop(fid, 2, np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64)),
                          np.zeros((0, 4), np.uint64))


def test_lazy_arithmetic_matches_packed_on_the_device(ctx):
    """The signed-digit multiplier of the hot kernels in every form (interleaved and chained columns, products, squares,
    a b +- c d, lazy / negated / repeated / extreme operands) against the packed Montgomery arithmetic, which the golden
    vectors and the C oracle pin above; Fr and Fq, two seeds."""
    for seed in (1, 0xC0FFEE):
        assert ctx.selftest_arithmetic(seed, 1 << 16) == [0, 0, 0, 0]
    with pytest.raises(ffi.ZkhipError):
        ctx.selftest_arithmetic(1, 0)


def test_poseidon_kat_through_device_fr(ctx):
    """The reference's Poseidon known answers recomputed with every field operation running on the device: the hash of
    (1, 2) (/root/reference/poseidon/src/hash.rs:903-918, lane 0 of the permutation) and the whole state after one
    permutation of (0, 1, 2) (/root/reference/poseidon/src/poseidon/septidon/tests.rs:58-61, all three lanes)."""
    k = golden("reference_kats.json")
    rounds = [[int(c, 16) for c in r] for r in k["poseidon_params"]["round_constants"]]
    mds = [[fr_arr([int(c, 16)]) for c in r] for r in k["poseidon_params"]["mds"]]
    mul = lambda x, y: ctx.field_binop(0, 2, x, y)   # noqa: E731
    add = lambda x, y: ctx.field_binop(0, 0, x, y)   # noqa: E731

    def permute(state):
        st = fr_arr(state)
        for rnd, rcs in enumerate(rounds):
            st = add(st, fr_arr(rcs))
            ns = 3 if (rnd < 4 or rnd >= 61) else 1
            x = st[:ns]
            x2 = mul(x, x)
            st = np.concatenate([mul(mul(x2, x2), x), st[ns:]])
            st = np.array([add(add(mul(mds[i][0], st[0:1]), mul(mds[i][1], st[1:2])), mul(mds[i][2], st[2:3]))[0]
                           for i in range(3)], dtype=np.uint64)
        return fr_ints(st)

    kat = k["poseidon_hash_with_domain"][0]
    assert permute([int(kat["domain"], 16)] + [int(x, 16) for x in kat["inputs"]])[0] == int(kat["expected"], 16)
    for kat in k["poseidon_permutation_full_state"]:
        assert permute([int(x, 16) for x in kat["state"]]) == [int(x, 16) for x in kat["expected"]]


def test_timing_levels_bracket_the_named_stages_only(ctx, co):
    """zkhip_timing_enable: level 1 brackets every stage of an MSM with HIP events, level 2 only the accumulate kernel (and
    the NTT), level 0 nothing; the result does not depend on the level."""
    rng = np.random.default_rng(5)
    n = 4096
    bases = co.g1_bases(n)
    bid = ctx.register_bases(bases)
    sc = rand_fr(rng, n)
    points = []
    for level, expect in ((1, {"msm_sort": 1, "msm_accumulate": 1, "msm_merge": 1, "msm_reduce": 1}),
                          (2, {"msm_sort": 0, "msm_accumulate": 1, "msm_merge": 0, "msm_reduce": 0}),
                          (False, {"msm_sort": 0, "msm_accumulate": 0, "msm_merge": 0, "msm_reduce": 0})):
        ctx.timing_enable(level)
        ctx.timing_reset()
        points.append(ctx.g1_to_affine(ctx.msm(bid, sc)))
        for family, launches in expect.items():
            us, cnt = ctx.timing_get(family)
            assert cnt == launches, (level, family)
            assert (us > 0) == (launches > 0), (level, family)
    ctx.timing_enable(False)
    ctx.release_bases(bid)
    assert np.array_equal(points[0], points[1]) and np.array_equal(points[0], points[2])
    assert np.array_equal(points[0], co.g1_to_affine(co.best_multiexp(sc, bases, threads=4)))


def test_curve_add_golden(ctx, co):
    g = golden("curve.json")
    p = affine_arr([hex_pt(c["p"]) for c in g["add"]])
    q = affine_arr([hex_pt(c["q"]) for c in g["add"]])
    assert [affine_pt(r) for r in ctx.g1_add_affine(p, q)] == [hex_pt(c["sum"]) for c in g["add"]]
    bases = co.g1_bases(3000)
    assert np.array_equal(ctx.g1_add_affine(bases[:1500], bases[1500:]), co.g1_add_affine(bases[:1500], bases[1500:]))


# ------------------------------------------------------------------------------------------------ MSM
def test_msm_golden(ctx, co):
    g = golden("msm.json")
    bids = {}
    for case in g["cases"]:
        n = case["n"]
        if n not in bids:
            bids[n] = ctx.register_bases(co.g1_bases(n))
        jac = ctx.msm(bids[n], hex_fr(case["scalars"]))
        assert affine_pt(ctx.g1_to_affine(jac)) == hex_pt(case["result"]), (n, case["dist"])
    for b in bids.values():
        ctx.release_bases(b)


@pytest.mark.parametrize("n", [1, 3, 1000, 4097, 1 << 14, (1 << 16) + 11])
def test_msm_vs_oracle(ctx, co, n):
    rng = np.random.default_rng(n)
    bases = co.g1_bases(n)
    bid = ctx.register_bases(bases)
    dists = {"uniform": rand_fr(rng, n), "witness": witness_like(rng, n, lambda c: co.from_canonical(0, c)),
             "ones": fr_arr([1] * min(n, 4097)) if n <= 4097 else None,
             "zeros": np.zeros((n, 4), dtype=np.uint64)}
    for name, sc in dists.items():
        if sc is None:
            continue
        m = len(sc)
        got = ctx.g1_to_affine(ctx.msm(bid, sc))
        exp = co.g1_to_affine(co.best_multiexp(sc, bases[:m], threads=8))
        assert np.array_equal(got, exp), (n, name)
    # empty input and a prefix of the registered bases
    assert not ctx.msm(bid, np.zeros((0, 4), np.uint64)).any()
    if n > 3:
        sc = rand_fr(rng, n // 2)
        assert np.array_equal(ctx.g1_to_affine(ctx.msm(bid, sc)),
                              co.g1_to_affine(co.best_multiexp(sc, bases[:n // 2], threads=8)))
    ctx.release_bases(bid)


def test_msm_base_range_split_and_fold(ctx, co):
    n = 5000
    rng = np.random.default_rng(2)
    bases, sc = co.g1_bases(n), rand_fr(rng, n)
    bid = ctx.register_bases(bases)
    d_sc = ctx.upload(sc)
    whole = ctx.g1_to_affine(ctx.msm(bid, sc))
    from zkevm_circuits_amd import sharding
    parts = []
    for r in range(3):
        first, cnt = sharding.base_range_for_rank(n, r, 3)
        parts.append(ctx.msm_range_dev(bid, first, d_sc + first * 32, cnt))
    assert np.array_equal(ctx.g1_to_affine(ctx.g1_sum_jacobian(np.stack(parts))), whole)
    batch = ctx.msm_batch_dev(bid, [d_sc, d_sc], n)
    assert np.array_equal(ctx.g1_to_affine(batch[0]), whole) and np.array_equal(ctx.g1_to_affine(batch[1]), whole)
    ctx.free(d_sc)
    ctx.release_bases(bid)


def test_msm_full_size_known_answer(ctx, co):
    """n = 2^19 with bases (i+1)G: sum_i s_i (i+1) G = (sum_i s_i (i+1) mod r) G, checked with one scalar
    multiplication instead of a CPU MSM; uniform and witness-like scalars."""
    n = 1 << 19
    rng = np.random.default_rng(20250523)
    bases = co.g1_bases(n)
    bid = ctx.register_bases(bases)
    for name in ("uniform", "witness"):
        sc = rand_fr(rng, n) if name == "uniform" else witness_like(rng, n, lambda c: co.from_canonical(0, c))
        canon = co.to_canonical(0, sc)
        k = 0
        vals = [sum(int(x) << (64 * j) for j, x in enumerate(row)) for row in canon]
        k = sum(v * (i + 1) for i, v in enumerate(vals)) % P
        exp = bn.g1_mul(bn.G1_GENERATOR, k)
        got = affine_pt(ctx.g1_to_affine(ctx.msm(bid, sc)))
        assert got == exp, name
    ctx.release_bases(bid)


def test_msm_batch_with_skewed_columns_matches_single_calls(ctx):
    """the pipelined batch over columns of different scalar distributions (uniform, witness-like, 16-bit, all ones:
    heavy-bucket pre-levels of either kind in neighbouring workspaces) equals the single calls"""
    n = 1 << 17
    rng = np.random.default_rng(17)
    bid = ctx.register_bases(ctx.g1_multiples(n))
    to_mont = lambda c: ctx.field_binop(0, 2, c, np.repeat(fr_arr([(1 << 256) % P]), len(c), axis=0))
    small = np.zeros((n, 4), dtype=np.uint64)
    small[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
    ones = np.zeros((n, 4), dtype=np.uint64)
    ones[:, 0] = 1
    cols = [rand_fr(rng, n), witness_like(rng, n, to_mont), to_mont(small), to_mont(ones),
            rand_fr(rng, n), np.zeros((n, 4), dtype=np.uint64)]
    d_cols = [ctx.upload(c) for c in cols]
    batch = ctx.msm_batch_dev(bid, d_cols, n)
    for i, d in enumerate(d_cols):
        assert np.array_equal(ctx.g1_to_affine(batch[i]), ctx.g1_to_affine(ctx.msm_range_dev(bid, 0, d, n))), i
    # all ones: sum of the bases = (n (n + 1) / 2) G
    assert affine_pt(ctx.g1_to_affine(batch[3])) == bn.g1_mul(bn.G1_GENERATOR, n * (n + 1) // 2)
    # a batch longer than the ring of workspaces (ten, with two sort and two accumulate streams and nine tails in
    # flight): every workspace is reused, with a column of another distribution than it held before
    singles = [ctx.g1_to_affine(ctx.msm_range_dev(bid, 0, d, n)) for d in d_cols]
    order = [(5 * i + 3) % len(d_cols) for i in range(27)]
    long_batch = ctx.msm_batch_dev(bid, [d_cols[k] for k in order], n)
    for i, k in enumerate(order):
        assert np.array_equal(ctx.g1_to_affine(long_batch[i]), singles[k]), (i, k)
    for d in d_cols:
        ctx.free(d)
    ctx.release_bases(bid)


@pytest.mark.parametrize("n", [140001, 200000, 1 << 20])
def test_msm_other_window_widths_known_answer(ctx, co, n):
    """window widths 15 (2^17 < n <= 2^18) and 16 with more than 2^19 points, against the closed form for bases (i+1)G"""
    rng = np.random.default_rng(n)
    bid = ctx.register_bases(ctx.g1_multiples(n))
    sc = rand_fr(rng, n)
    vals = fr_ints(sc)
    k = sum(v * (i + 1) for i, v in enumerate(vals)) % P
    assert affine_pt(ctx.g1_to_affine(ctx.msm(bid, sc))) == bn.g1_mul(bn.G1_GENERATOR, k)
    ctx.release_bases(bid)


# ------------------------------------------------------------------------------------------------ NTT / domain
def test_ntt_golden(ctx):
    for case in golden("ntt.json")["cases"]:
        out = ctx.ntt(hex_fr(case["input"]), case["log_n"], fr_arr([int(case["omega"], 16)])[0])
        assert np.array_equal(out, hex_fr(case["output"])), case["log_n"]


@pytest.mark.parametrize("log_n", [1, 4, 8, 9, 11, 13, 16, 17, 18])
def test_ntt_vs_oracle(ctx, co, log_n):
    rng = np.random.default_rng(log_n)
    omega = fr_arr([pow(bn.FR_ROOT_OF_UNITY, 1 << (28 - log_n), P)])[0]
    a = rand_fr(rng, 1 << log_n)
    assert np.array_equal(ctx.ntt(a, log_n, omega), co.best_fft(a, omega, log_n, threads=8))


def test_ntt_rejects_bad_omega(ctx):
    with pytest.raises(zk.ZkhipError):
        ctx.ntt(np.zeros((8, 4), np.uint64), 3, fr_arr([5])[0])


def test_domain_golden(ctx):
    for case in golden("domain.json")["cases"]:
        d = zk.Domain(ctx, case["j"], case["k"])
        assert d.extended_k == case["extended_k"]
        assert fr_ints(d.constant(0)) == [int(case["omega"], 16)]
        assert fr_ints(d.constant(2)) == [int(case["extended_omega"], 16)]
        coeff = hex_fr(case["coeff"])
        assert np.array_equal(d.coeff_to_lagrange(coeff), hex_fr(case["lagrange"]))
        assert np.array_equal(d.lagrange_to_coeff(hex_fr(case["lagrange"])), coeff)
        assert np.array_equal(d.coeff_to_extended(coeff), hex_fr(case["extended"]))
        assert np.array_equal(d.extended_to_coeff(hex_fr(case["big_extended"])), hex_fr(case["big_coeff"]))
        assert np.array_equal(d.divide_by_vanishing_poly(hex_fr(case["big_extended"])), hex_fr(case["big_divided"]))
        d.close()


@pytest.mark.parametrize("j,k", [(5, 12), (9, 11), (3, 13), (4, 10), (12, 9), (17, 10)])
def test_domain_vs_oracle_batched(ctx, co, j, k):
    rng = np.random.default_rng(j * 100 + k)
    d, od = zk.Domain(ctx, j, k), co.Domain(j, k, threads=8)
    n, en = d.n, d.extended_len
    batch = 3
    cols = [rand_fr(rng, n) for _ in range(batch)]
    d_in = ctx.upload(np.concatenate(cols))
    d_ext = ctx.alloc(batch * en * 32)
    d.coeff_to_extended_dev(d_in, d_ext, batch)
    ext = ctx.download(d_ext, (batch, en, 4))
    for b in range(batch):
        assert np.array_equal(ext[b], od.coeff_to_extended(cols[b]))
    d.divide_by_vanishing_poly_dev(d_ext, batch)
    div = ctx.download(d_ext, (batch, en, 4))
    for b in range(batch):
        assert np.array_equal(div[b], od.divide_by_vanishing_poly(ext[b]))
    d_co = ctx.alloc(batch * n * d.quotient_poly_degree * 32)
    d.extended_to_coeff_dev(d_ext, d_co, batch)
    back = ctx.download(d_co, (batch, n * d.quotient_poly_degree, 4))
    for b in range(batch):
        assert np.array_equal(back[b], od.extended_to_coeff(div[b]))
    d.lagrange_to_coeff_dev(d_in, batch)
    l2c = ctx.download(d_in, (batch, n, 4))
    for b in range(batch):
        assert np.array_equal(l2c[b], od.lagrange_to_coeff(cols[b]))
    d.coeff_to_lagrange_dev(d_in, batch)
    assert np.array_equal(ctx.download(d_in, (batch, n, 4)), np.stack(cols))      # round trip
    for p in (d_in, d_ext, d_co):
        ctx.free(p)


@pytest.mark.parametrize("ext_k", [21, 22])
def test_coset_extension_full_size_properties(ctx, co, ext_k):
    """k = 19 -> 2^21 / 2^22: (a) a few outputs against direct evaluation p(zeta * w^i) by Horner on big ints,
    (b) extended_to_coeff inverts coeff_to_extended, (c) linearity."""
    check_coset_extension_properties(ctx, co, 19, 5 if ext_k == 21 else 9, ext_k)


def check_coset_extension_properties(ctx, co, k, j, ext_k):
    d = zk.Domain(ctx, j, k)
    assert d.extended_k == ext_k
    rng = np.random.default_rng(ext_k)
    n, en = d.n, d.extended_len
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    d_a, d_b = ctx.upload(a), ctx.upload(b)
    d_ea, d_eb = ctx.alloc(en * 32), ctx.alloc(en * 32)
    d.coeff_to_extended_dev(d_a, d_ea)
    d.coeff_to_extended_dev(d_b, d_eb)
    ea, eb = ctx.download(d_ea, (en, 4)), ctx.download(d_eb, (en, 4))
    coeffs = fr_ints(a)
    w = fr_ints(d.constant(2))[0]
    for i in (0, 1, en // 3 + 5, en - 1):
        x = bn.FR_ZETA * pow(w, i, P) % P
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + c) % P
        assert fr_ints(ea[i])[0] == acc, i
    d_s = ctx.upload(co.field_binop(0, 0, a, b))
    d_es = ctx.alloc(en * 32)
    d.coeff_to_extended_dev(d_s, d_es)
    assert np.array_equal(ctx.download(d_es, (en, 4)), ctx.field_binop(0, 0, ea, eb))
    d_back = ctx.alloc(n * d.quotient_poly_degree * 32)
    d.extended_to_coeff_dev(d_ea, d_back)
    back = ctx.download(d_back, (n * d.quotient_poly_degree, 4))
    assert np.array_equal(back[:n], a) and not back[n:].any()
    for p in (d_a, d_b, d_ea, d_eb, d_s, d_es, d_back):
        ctx.free(p)
    d.close()


# ------------------------------------------------------------------------------------------------ evaluate_h
def test_evaluate_h_golden(ctx):
    from evalh_cases import build_call
    for case in golden("evalh.json")["cases"]:
        cs, call = build_call(case)
        d = zk.Domain(ctx, cs.degree(), case["k"])
        out = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
        assert np.array_equal(out, hex_fr(case["h"])), case["cs"]
        d.close()


def _random_call(cs, rng, n, en, od):
    call = evalh_host.EvalhCall(cs)
    call.fixed = [rand_fr(rng, en) for _ in range(cs.num_fixed_columns)]
    coeff = {"advice": [rand_fr(rng, n) for _ in range(cs.num_advice_columns)],
             "instance": [rand_fr(rng, n) for _ in range(cs.num_instance_columns)],
             "lookups": [tuple(rand_fr(rng, n) for _ in range(3)) for _ in cs.lookups]}
    call.challenges = rand_fr(rng, cs.num_challenges)
    call.y, call.beta, call.gamma, call.theta = (rand_fr(rng, 1)[0] for _ in range(4))
    call.l0, call.l_last, call.l_active = rand_fr(rng, en), rand_fr(rng, en), rand_fr(rng, en)
    call.perm_sigma = [rand_fr(rng, en) for _ in cs.permutation_columns]
    call.perm_z = [rand_fr(rng, en) for _ in range(cs.num_permutation_sets())]
    ext = {"advice": [od.coeff_to_extended(a) for a in coeff["advice"]],
           "instance": [od.coeff_to_extended(a) for a in coeff["instance"]],
           "lookups": [tuple(od.coeff_to_extended(p) for p in t) for t in coeff["lookups"]]}
    return call, coeff, ext


@pytest.mark.parametrize("k,kw", [
    (7, dict()),
    (9, dict(num_advice=12, num_gate_polys=40, num_lookups=3, num_perm_cols=7)),
    (11, dict(num_advice=24, num_fixed=6, num_gate_polys=160, num_lookups=4, num_perm_cols=9,
              rotations=(-5, -2, -1, 0, 1, 3))),
    (10, dict(num_advice=40, num_fixed=8, num_gate_polys=300, max_degree=9, num_lookups=2, num_perm_cols=11)),
    (8, dict(num_lookups=0, num_perm_cols=0)),
])
def test_evaluate_h_vs_oracle(ctx, co, k, kw):
    cs = synthetic.synthetic_cs(**kw)
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    rng = np.random.default_rng(k)
    call, coeff, ext = _random_call(cs, rng, d.n, d.extended_len, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
    exp = od.evaluate_h(desc)
    assert np.array_equal(d.evaluate_h(desc), exp)
    # upstream's calling convention: advice / instance / lookup polynomials in coefficient form
    call.advice, call.instance, call.lookups = coeff["advice"], coeff["instance"], coeff["lookups"]
    assert np.array_equal(d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_COEFF)), exp)
    # quotient tail: divide by the vanishing polynomial, back to coefficients
    assert np.array_equal(d.extended_to_coeff(d.divide_by_vanishing_poly(exp)),
                          od.extended_to_coeff(od.divide_by_vanishing_poly(exp)))
    d.close()


_SPLIT_SHAPES = [
    (7, dict()),
    (8, dict(num_advice=12, num_gate_polys=40, num_lookups=3, num_perm_cols=7)),
    (8, dict(num_advice=24, num_fixed=6, num_gate_polys=120, max_degree=9, num_lookups=4, lookup_width=3, num_perm_cols=9,
             rotations=(-5, -2, -1, 0, 1, 3))),
    (7, dict(num_advice=6, num_gate_polys=12, max_degree=3, num_lookups=1, num_perm_cols=2)),      # extended domain 2n
    (6, dict(num_lookups=0, num_perm_cols=0)),
]


@pytest.mark.parametrize("k,kw", _SPLIT_SHAPES)
def test_evaluate_h_low_and_high_degree_terms_add_up(ctx, co, k, kw):
    """desc.term_select: the terms of degree <= bound and the others, each with its own power of y, add up to h as the
    oracle computes it, for every bound (a bound of 0 or of the constraint degree leaves one side empty)"""
    cs = synthetic.synthetic_cs(**kw)
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    rng = np.random.default_rng(500 + k)
    call, coeff, ext = _random_call(cs, rng, d.n, d.extended_len, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    exp = od.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
    sizes = {}
    for bound in (0, 1, 2, 3, 4, 6, cs.degree()):
        parts = []
        for sel in (ffi.TERMS_LOW, ffi.TERMS_HIGH):
            call.terms = (sel, bound)
            parts.append(d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)))
            sizes[(bound, sel)] = ctx.evaluate_h_last_stats()["multiplications"]
        assert np.array_equal(co.field_binop(0, 0, parts[0], parts[1]), exp), bound
        if bound == 0:
            assert not parts[0].any() and np.array_equal(parts[1], exp)
        if bound == cs.degree():
            assert not parts[1].any() and np.array_equal(parts[0], exp)
    call.terms = None
    assert np.array_equal(d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)), exp)
    assert sizes[(0, ffi.TERMS_LOW)] == 0 and sizes[(cs.degree(), ffi.TERMS_HIGH)] == 0
    d.close()


def test_split_extension_layout_and_half_domain_extension(ctx, co):
    """coeff_to_extended_split_dev = coeff_to_extended with the even-indexed rows first, then the odd-indexed ones;
    extend_half_dev carries the values of a polynomial of degree < 2^(extended_k - 1) from the even to the odd rows"""
    for k, j in ((5, 3), (6, 4), (7, 5), (8, 9), (11, 9)):
        d, od = zk.Domain(ctx, j, k), co.Domain(j, k, threads=8)
        en, half = d.extended_len, d.extended_len // 2
        rng = np.random.default_rng(k * 10 + j)
        a, b = rand_fr(rng, d.n), rand_fr(rng, d.n)
        ea, eb = od.coeff_to_extended(a), od.coeff_to_extended(b)
        d_in = ctx.upload(np.concatenate([a, b]))
        d_out = ctx.alloc(2 * en * 32)
        d.coeff_to_extended_split_dev(d_in, d_out, 2)
        got = ctx.download(d_out, (2, en, 4))
        for g, e in ((got[0], ea), (got[1], eb)):
            assert np.array_equal(g[:half], e[0::2]) and np.array_equal(g[half:], e[1::2])
        # the even-indexed rows alone, by one transform of half the size: back to back, and into the even halves of
        # vectors in the split layout (whose odd halves stay as they were)
        d_half = ctx.alloc(2 * half * 32)
        d.coeff_to_extended_half_dev(d_in, d_half, None, 2)
        assert np.array_equal(ctx.download(d_half, (2, half, 4)), np.stack([ea[0::2], eb[0::2]]))
        ctx.upload(np.concatenate([eb, ea]), d_out)
        d.coeff_to_extended_half_dev(d_in, d_out, en, 2)
        got = ctx.download(d_out, (2, en, 4))
        assert np.array_equal(got[0][:half], ea[0::2]) and np.array_equal(got[1][:half], eb[0::2])
        assert np.array_equal(got[0][half:], eb[half:]) and np.array_equal(got[1][half:], ea[half:])
        ctx.free(d_half)
        with pytest.raises(zk.ZkhipError):
            d.coeff_to_extended_half_dev(d_in, d_out, half - 1, 2)
        # degree < n <= en / 2: one polynomial; and, from four cosets on, a product of two (degree < 2n <= en / 2)
        d_odd = ctx.alloc(2 * half * 32)
        d_even = ctx.upload(np.concatenate([ea[0::2], eb[0::2]]))
        d.extend_half_dev(d_even, d_odd, 2)
        odd = ctx.download(d_odd, (2, half, 4))
        assert np.array_equal(odd[0], ea[1::2]) and np.array_equal(odd[1], eb[1::2])
        if en >= 4 * d.n:
            prod = co.field_binop(0, 2, ea, eb)
            d_pe = ctx.upload(prod[0::2])
            d.extend_half_dev(d_pe, d_odd)
            assert np.array_equal(ctx.download(d_odd, (half, 4)), prod[1::2])
            ctx.free(d_pe)
        for p in (d_in, d_out, d_odd, d_even):
            ctx.free(p)
        d.close()


@pytest.mark.parametrize("k,kw", _SPLIT_SHAPES)
def test_evaluate_h_with_low_degree_terms_on_half_the_domain(ctx, co, k, kw):
    """`evalh_host.evaluate_h_split`: every column a genuine polynomial of degree < n in the parity-split layout, the terms of
    degree <= m / 2 evaluated on the even-indexed rows only and extended; h equals the oracle's whole-domain evaluation bit for
    bit, and so do the quotient's coefficients"""
    cs = synthetic.synthetic_cs(**kw)
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    n, en = d.n, d.extended_len
    rng = np.random.default_rng(900 + k)
    names = (["fixed"] * cs.num_fixed_columns + ["advice"] * cs.num_advice_columns + ["instance"] * cs.num_instance_columns
             + ["l"] * 3 + ["lookup"] * (3 * len(cs.lookups)) + ["sigma"] * len(cs.permutation_columns)
             + ["z"] * cs.num_permutation_sets())
    coeffs = [rand_fr(rng, n) for _ in names]
    d_coeff = ctx.upload(np.concatenate(coeffs))
    d_nat, d_split = ctx.alloc(len(names) * en * 32), ctx.alloc(len(names) * en * 32)
    d.coeff_to_extended_dev(d_coeff, d_nat, len(names))
    d.coeff_to_extended_split_dev(d_coeff, d_split, len(names))

    def fill(call, base):
        it = iter(base + i * en * 32 for i in range(len(names)))
        take = lambda cnt: [next(it) for _ in range(cnt)]                               # noqa: E731
        call.fixed, call.advice, call.instance = take(cs.num_fixed_columns), take(cs.num_advice_columns), take(cs.num_instance_columns)
        call.l0, call.l_last, call.l_active = take(3)
        call.lookups = [tuple(take(3)) for _ in cs.lookups]
        call.perm_sigma, call.perm_z = take(len(cs.permutation_columns)), take(cs.num_permutation_sets())
    call = evalh_host.EvalhCall(cs)
    call.challenges = rand_fr(rng, cs.num_challenges)
    call.y, call.beta, call.gamma, call.theta = (rand_fr(rng, 1)[0] for _ in range(4))
    # the oracle on host copies of the natural layout
    nat = ctx.download(d_nat, (len(names), en, 4))
    host = evalh_host.EvalhCall(cs)
    host.challenges, host.y, host.beta, host.gamma, host.theta = call.challenges, call.y, call.beta, call.gamma, call.theta
    it = iter(nat)
    take = lambda cnt: [next(it) for _ in range(cnt)]                                   # noqa: E731
    host.fixed, host.advice, host.instance = take(cs.num_fixed_columns), take(cs.num_advice_columns), take(cs.num_instance_columns)
    host.l0, host.l_last, host.l_active = take(3)
    host.lookups = [tuple(take(3)) for _ in cs.lookups]
    host.perm_sigma, host.perm_z = take(len(cs.permutation_columns)), take(cs.num_permutation_sets())
    exp = od.evaluate_h(host.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
    d_h = ctx.alloc(en * 32)
    fill(call, d_nat)
    d.evaluate_h(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_h)
    assert np.array_equal(ctx.download(d_h, (en, 4)), exp)
    ctx.upload(np.zeros((en, 4), dtype=np.uint64), d_h)
    fill(call, d_split)
    evalh_host.evaluate_h_split(ctx, d, call, d_h)
    assert np.array_equal(ctx.download(d_h, (en, 4)), exp)
    gain = evalh_host.split_gain(d, cs, call.ev)
    assert gain[0] + gain[1] >= gain[2] * 0.9 and gain[0] <= gain[2] and gain[1] <= gain[2]
    d.divide_by_vanishing_poly_dev(d_h)
    d_q = ctx.alloc(n * d.quotient_poly_degree * 32)
    d.extended_to_coeff_dev(d_h, d_q)
    assert np.array_equal(ctx.download(d_q, (n * d.quotient_poly_degree, 4)), od.extended_to_coeff(od.divide_by_vanishing_poly(exp)))
    for p in (d_coeff, d_nat, d_split, d_h, d_q):
        ctx.free(p)
    d.close()


def test_evaluate_h_program_is_built_once_per_shape(ctx, co):
    """The fused program is kept per constraint-system shape (upstream: `ProvingKey::ev`): a second call with other
    columns, challenges and another extended size reuses it, a different shape builds its own, and every result still
    equals the oracle's."""
    cs = synthetic.synthetic_cs(num_advice=10, num_gate_polys=24, num_lookups=2, num_perm_cols=5, seed=77)
    ev = plonk.Evaluator(cs)
    before = ctx.evaluate_h_cache_stats()
    for rnd, k in ((1, 7), (2, 7), (3, 8)):
        d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
        call, _, ext = _random_call(cs, np.random.default_rng(100 + rnd), d.n, d.extended_len, od)
        call.ev = ev
        call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
        desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
        assert np.array_equal(d.evaluate_h(desc), od.evaluate_h(desc))
        d.close()
    after = ctx.evaluate_h_cache_stats()
    assert after["builds"] - before["builds"] == 1 and after["hits"] - before["hits"] == 2
    cs2 = synthetic.synthetic_cs(num_advice=10, num_gate_polys=24, num_lookups=2, num_perm_cols=5, seed=78)
    d, od = zk.Domain(ctx, cs2.degree(), 7), co.Domain(cs2.degree(), 7, threads=8)
    call, _, ext = _random_call(cs2, np.random.default_rng(5), d.n, d.extended_len, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
    assert np.array_equal(d.evaluate_h(desc), od.evaluate_h(desc))
    assert ctx.evaluate_h_cache_stats()["builds"] - after["builds"] == 1
    d.close()


def _factored_terms_cs(lookups):
    """Gate polynomials that take every path of the term grouping in the program builder: shared factors on the left and
    on the right, a factor whose terms are interleaved with terms of another factor and with terms that have none, a
    square, a product that two polynomials share (read twice, so it stays a term of its own), bare columns and sums as
    polynomials, a selector times a condition in front of several constraints, a lone product at the end."""
    from zkevm_circuits_amd.plonk import (ANY_ADVICE, ANY_FIXED, ANY_INSTANCE, Advice, Challenge, Constant, ConstraintSystem,
                                         Fixed, Instance, Lookup, Product, Scaled, Sub, Sum)
    cs = ConstraintSystem(num_fixed_columns=3, num_advice_columns=6, num_instance_columns=1, num_challenges=1)
    a, b, c, d, e, f = (Advice(i) for i in range(6))
    q0, q1, q2 = Fixed(0), Fixed(1), Fixed(2)
    cond = Sub(Constant(1), Advice(5, 1))
    shared = Product(a, Advice(1, -1))
    cs.gates = [
        [Product(q0, Sub(Product(a, b), c)), Sum(a, Advice(2, 1)), Product(Sub(d, e), q0), Product(q1, Sum(e, f))],
        [Product(c, c), Product(q0, Scaled(Advice(3, -1), 12345)), shared, Product(q1, Sub(b, Challenge(0)))],
        [Product(Product(q2, cond), Sub(a, b)), Product(Product(q2, cond), Sub(c, d)), d,
         Product(Product(q2, cond), Product(e, Advice(0, 2)))],
        [shared, Product(Sub(f, Constant(7)), q1), Product(q0, Instance(0)), Product(q0, Sum(a, Constant(3)))],
        [Product(Sum(a, b), Sum(c, d))],
    ]
    for i in range(lookups):
        cs.lookups.append(Lookup([Product(q0, Advice(i)), Sum(b, Advice(i + 1, 1))], [Fixed(1), Advice(4)]))
    cs.permutation_columns = [(ANY_ADVICE, 0), (ANY_ADVICE, 2), (ANY_FIXED, 1), (ANY_INSTANCE, 0)]
    return cs


@pytest.mark.parametrize("lookups", [0, 2])
def test_evaluate_h_factored_terms_vs_oracle(ctx, co, lookups):
    """h = sum over factors of F * (weighted sum of the co-factors) + the chain of the other terms: the same values as
    upstream's term-by-term chain, which the oracle evaluates"""
    cs = _factored_terms_cs(lookups)
    for k in (6, 9):
        d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
        call, coeff, ext = _random_call(cs, np.random.default_rng(40 + k + lookups), d.n, d.extended_len, od)
        call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
        desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
        exp = od.evaluate_h(desc)
        assert np.array_equal(d.evaluate_h(desc), exp)
        # a window of rows of two cosets in compact layout runs the same program
        e = d.extended_k - k
        if e >= 1:
            cosets = [0, (1 << e) - 1]
            call.cosets = cosets
            compact = lambda a: np.ascontiguousarray(np.stack([a[j::1 << e] for j in cosets], axis=1).reshape(-1, 4))
            call.fixed = [compact(x) for x in call.fixed]
            call.advice = [compact(x) for x in ext["advice"]]
            call.instance = [compact(x) for x in ext["instance"]]
            call.lookups = [tuple(compact(x) for x in t) for t in ext["lookups"]]
            call.perm_sigma = [compact(x) for x in call.perm_sigma]
            call.perm_z = [compact(x) for x in call.perm_z]
            call.l0, call.l_last, call.l_active = compact(call.l0), compact(call.l_last), compact(call.l_active)
            got = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
            assert np.array_equal(got, compact(exp))
        d.close()


def _additive_cs(degree, rotations, width):
    """Gates that are long sums of columns and sums of sums times sums: exercises the magnitude bookkeeping of the
    interpreter's lazy arithmetic (sums of many 32 p column values) and extended domains of 2^1 .. 2^4 times n."""
    from zkevm_circuits_amd.plonk import Advice, Constant, ConstraintSystem, Fixed, Lookup, Neg, Product, Sub, Sum
    import random
    rnd = random.Random(width * 100 + degree)
    A, F = 12, 3
    cs = ConstraintSystem(num_fixed_columns=F, num_advice_columns=A, num_instance_columns=1, num_challenges=1)

    def adv():
        return Advice(rnd.randrange(A), rnd.choice(rotations))

    def long_sum(w):
        e = adv()
        for i in range(w - 1):
            e = Sum(e, adv()) if i % 3 else Sub(e, adv())
        return e

    polys = []
    for i in range(6):
        body = long_sum(width)
        for _ in range(degree - 2):
            body = Product(body, long_sum(2 + i % 3))
        polys.append(Product(Fixed(i % F, 0), Sum(body, Neg(Product(Constant(rnd.randrange(1, 1 << 200)), long_sum(width))))))
    cs.gates.append(polys)
    if degree >= 4:   # a lookup argument alone needs degree 4
        cs.lookups.append(Lookup([long_sum(3), adv()], [Fixed(0, 0), Advice(1, 0)]))
    cs.permutation_columns = [(3, 0), (3, 1), (2, 0)]   # advice 0, advice 1, fixed 0
    return cs


@pytest.mark.parametrize("k,degree,rotations,width", [
    (8, 3, (-1, 0, 1), 6),          # extended domain 2n
    (8, 5, (-3, 0, 2), 40),         # sums of 40 columns
    (9, 9, (-12, 0, 204), 9),       # extended domain 8n, keccak-like far rotation
    (6, 12, (-1, 0, 1), 5),         # extended domain 16n
])
def test_evaluate_h_additive_gates_and_domains(ctx, co, k, degree, rotations, width):
    cs = _additive_cs(degree, rotations, width)
    assert cs.degree() == degree
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    rng = np.random.default_rng(k * 31 + degree)
    call, coeff, ext = _random_call(cs, rng, d.n, d.extended_len, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
    assert d.extended_k - k == {3: 1, 5: 2, 9: 3, 12: 4}[degree]
    assert np.array_equal(d.evaluate_h(desc), od.evaluate_h(desc))
    d.close()


def _sum_of_products_cs(width):
    """Gates built from sums and differences of products, the shapes the program builder fuses: x + p q, x - p q,
    p q - x, a b + p q and a b - p q with columns, constants, intermediate values and long (un-normalised) sums as
    operands."""
    from zkevm_circuits_amd.plonk import Advice, Constant, ConstraintSystem, Fixed, Product, Sub, Sum
    import random
    rnd = random.Random(width)
    A, F = 10, 3
    cs = ConstraintSystem(num_fixed_columns=F, num_advice_columns=A, num_instance_columns=1, num_challenges=1)

    def adv():
        return Advice(rnd.randrange(A), rnd.choice((-1, 0, 1)))

    def long_sum(w):
        e = adv()
        for i in range(w - 1):
            e = Sum(e, adv()) if i % 2 else Sub(e, adv())
        return e

    const = Constant(rnd.randrange(1, 1 << 250))
    polys = [
        Sum(Product(adv(), adv()), Product(adv(), adv())),
        Sub(Product(long_sum(width), adv()), Product(adv(), long_sum(width))),
        Sub(adv(), Product(adv(), const)),
        Sub(Product(adv(), adv()), long_sum(width)),
        Sum(long_sum(3), Product(long_sum(width), long_sum(width))),
        Sum(Product(Sum(adv(), adv()), Sum(adv(), adv())), Product(Sub(adv(), adv()), Sum(adv(), const))),
        Product(Sum(Product(adv(), adv()), Product(adv(), adv())),
                Sub(Product(adv(), long_sum(2)), Product(long_sum(width), long_sum(width)))),
        Sub(Product(Product(adv(), adv()), long_sum(width)), Product(Product(adv(), adv()), Product(adv(), adv()))),
    ]
    cs.gates.append([Product(Fixed(i % F, 0), p) for i, p in enumerate(polys)] + polys[:3])
    return cs


@pytest.mark.parametrize("width", [2, 7, 30])
def test_evaluate_h_fused_products(ctx, co, width, monkeypatch):
    """the two-product and multiply-add instructions against the oracle, and against the same program left unfused"""
    k = 7
    cs = _sum_of_products_cs(width)
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    rng = np.random.default_rng(width)
    call, coeff, ext = _random_call(cs, rng, d.n, d.extended_len, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
    exp = od.evaluate_h(desc)
    assert np.array_equal(d.evaluate_h(desc), exp)
    fused = ctx.evaluate_h_last_stats()
    monkeypatch.setenv("ZKHIP_EVALH_FUSE", "0")
    assert np.array_equal(d.evaluate_h(desc), exp)
    plain = ctx.evaluate_h_last_stats()
    assert fused["multiplications"] == plain["multiplications"]       # the same products ...
    assert fused["instructions"] <= plain["instructions"] - 10         # ... in fewer instructions
    d.close()


def _factors_of_sums_cs():
    """Gates whose products have sums and differences of memory operands as factors -- the shapes the program builder turns
    into a (b +- c) instructions: column + constant, column - column, constant - column (the reversed difference), a sum on
    either side or on both sides of a product, and a chain of such factors behind a selector."""
    from zkevm_circuits_amd.plonk import Advice, Constant, ConstraintSystem, Fixed, Product, Sub, Sum
    import random
    rnd = random.Random(11)
    A, F = 8, 3
    cs = ConstraintSystem(num_fixed_columns=F, num_advice_columns=A, num_instance_columns=1, num_challenges=1)
    adv = lambda: Advice(rnd.randrange(A), rnd.choice((-1, 0, 1)))          # noqa: E731
    const = lambda: Constant(rnd.randrange(1, 1 << 250))                    # noqa: E731
    chain = Product(Fixed(0, 0), adv())
    for _ in range(6):
        chain = Product(chain, Sum(adv(), const()))
    polys = [
        Product(Sum(adv(), const()), adv()),
        Product(adv(), Sub(adv(), adv())),
        Product(Sub(const(), adv()), Fixed(1, 0)),
        Product(Sum(adv(), adv()), Sum(adv(), const())),
        Product(Product(Fixed(2, 0), Sub(adv(), const())), Sub(Fixed(1, 1), adv())),
        chain,
    ]
    cs.gates.append(polys)
    return cs


def test_evaluate_h_sums_moved_into_products(ctx, co, monkeypatch):
    """the a (b +- c) instruction against the oracle, and against the same program with every sum left an instruction of
    its own"""
    k = 7
    cs = _factors_of_sums_cs()
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    rng = np.random.default_rng(5)
    call, coeff, ext = _random_call(cs, rng, d.n, d.extended_len, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
    exp = od.evaluate_h(desc)
    assert np.array_equal(d.evaluate_h(desc), exp)
    moved = ctx.evaluate_h_last_stats()
    monkeypatch.setenv("ZKHIP_EVALH_PREADD", "0")
    assert np.array_equal(d.evaluate_h(desc), exp)
    plain = ctx.evaluate_h_last_stats()
    assert moved["multiplications"] == plain["multiplications"]
    assert moved["instructions"] <= plain["instructions"] - 5
    d.close()


def test_evaluate_h_rejects_malformed_graph(ctx):
    cs = synthetic.synthetic_cs(num_lookups=0, num_perm_cols=0)
    d = zk.Domain(ctx, cs.degree(), 6)
    rng = np.random.default_rng(0)
    call, coeff, ext = _random_call(cs, rng, d.n, d.extended_len, type("O", (), {"coeff_to_extended": lambda s, a: np.zeros((d.extended_len, 4), np.uint64)})())
    call.advice, call.instance = ext["advice"][:-1], ext["instance"]      # one advice column short
    with pytest.raises(zk.ZkhipError):
        d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
    d.close()


# ------------------------------------------------------------------------------------------------ grand products, openings
def _with_zeros(rng, n, every=97):
    a = rand_fr(rng, n)
    a[::every] = 0
    return a


@pytest.mark.parametrize("n", [1, 2, 255, 4096, 70001])
def test_batch_invert_and_prefix_product(ctx, n):
    import halo2_model as hm
    rng = np.random.default_rng(n)
    a = _with_zeros(rng, n) if n > 2 else rand_fr(rng, n)
    d = ctx.upload(a)
    ctx.batch_invert_dev(0, d, n)
    got = ctx.download(d, (n, 4))
    exp = hm.batch_invert(fr_ints(a))
    assert fr_ints(got) == exp
    # product with the input is one wherever the input is non-zero
    prod = ctx.field_binop(0, 2, got, a)
    one = fr_arr([1])[0]
    nz = a.any(axis=1)
    assert (prod[nz] == one).all() and not prod[~nz].any()
    # running product, in place and out of place
    m = rand_fr(rng, n)
    init = rand_fr(rng, 1)[0]
    d_m, d_o = ctx.upload(m), ctx.alloc(n * 32)
    ctx.prefix_product_dev(d_m, d_o, n, init)
    exp = hm.prefix_product(fr_ints(m), fr_ints(init[None])[0])
    assert fr_ints(ctx.download(d_o, (n, 4))) == exp
    ctx.prefix_product_dev(d_m, d_m, n, init)
    assert fr_ints(ctx.download(d_m, (n, 4))) == exp
    for p in (d, d_m, d_o):
        ctx.free(p)


def test_batch_invert_fq_and_empty(ctx):
    rng = np.random.default_rng(5)
    a = rand_fr(rng, 1000)          # < 2^252, valid Fq residues as well
    d = ctx.upload(a)
    ctx.batch_invert_dev(1, d, 1000)
    got = ctx.download(d, (1000, 4))
    assert (ctx.field_binop(1, 2, got, a) == fq_arr([1])[0]).all()
    ctx.batch_invert_dev(0, 0, 0)
    ctx.free(d)


@pytest.mark.parametrize("k,cols", [(4, 1), (9, 3), (12, 7)])
def test_permutation_and_lookup_products(ctx, k, cols):
    import halo2_model as hm
    n = 1 << k
    rng = np.random.default_rng(k * 10 + cols)
    d = zk.Domain(ctx, 5, k)
    omega = fr_ints(d.constant(0)[None])[0]
    values = [rand_fr(rng, n) for _ in range(cols)]
    sigmas = [rand_fr(rng, n) for _ in range(cols)]
    beta, gamma, z0 = (rand_fr(rng, 1)[0] for _ in range(3))
    j0 = 3
    delta_start = fr_arr([pow(bn.FR_DELTA, j0, P)])[0]
    d_vals = [ctx.upload(v) for v in values]
    d_sigs = [ctx.upload(s) for s in sigmas]
    d_z = ctx.alloc(n * 32)
    d.permutation_product_dev(d_vals, d_sigs, beta, gamma, delta_start, z0, d_z)
    exp = hm.permutation_product(omega, [fr_ints(v) for v in values], [fr_ints(s) for s in sigmas], fr_ints(beta[None])[0],
                                 fr_ints(gamma[None])[0], pow(bn.FR_DELTA, j0, P), fr_ints(z0[None])[0])
    assert fr_ints(ctx.download(d_z, (n, 4))) == exp
    # lookup product; a' / s' with some (a' + beta) = 0 rows would be a broken witness, so plain random data
    a, s, ap, sp = (rand_fr(rng, n) for _ in range(4))
    ptrs = [ctx.upload(x) for x in (a, s, ap, sp)]
    d.lookup_product_dev(ptrs[0], ptrs[1], ptrs[2], ptrs[3], beta, gamma, d_z)
    exp = hm.lookup_product(fr_ints(a), fr_ints(s), fr_ints(ap), fr_ints(sp), fr_ints(beta[None])[0], fr_ints(gamma[None])[0])
    assert fr_ints(ctx.download(d_z, (n, 4))) == exp
    # several lookups in one call (more than one group of 32 at the small size; a zero denominator stays zero as in the
    # single call): the same vectors as lookup after lookup
    count = 35 if k <= 9 else 3
    sets = []
    for i in range(count):
        a, s, ap, sp = (rand_fr(rng, n) for _ in range(4))
        if i == 1:
            ap[5] = fr_arr([(-fr_ints(beta[None])[0]) % P])[0]           # a' + beta = 0 on one row
        sets.append([ctx.upload(x) for x in (a, s, ap, sp)])
    singles = []
    for q in sets:
        d.lookup_product_dev(q[0], q[1], q[2], q[3], beta, gamma, d_z)
        singles.append(ctx.download(d_z, (n, 4)))
    d_zs = [ctx.alloc(n * 32) for _ in range(count)]
    d.lookup_product_batch_dev([q[0] for q in sets], [q[1] for q in sets], [q[2] for q in sets], [q[3] for q in sets],
                               beta, gamma, d_zs)
    for i in range(count):
        assert np.array_equal(ctx.download(d_zs[i], (n, 4)), singles[i]), i
    d.lookup_product_batch_dev([], [], [], [], beta, gamma, [])
    with pytest.raises(zk.ZkhipError):
        d.lookup_product_batch_dev([sets[0][0]], [sets[0][1]], [sets[0][2]], [0], beta, gamma, d_zs[:1])
    for p in d_vals + d_sigs + ptrs + [d_z] + d_zs + [x for q in sets for x in q]:
        ctx.free(p)
    d.close()


def test_grand_product_full_size_telescopes(ctx):
    """n = 2^19: with sigma_j[i] = delta^j omega^i (the identity permutation) every fraction is one, so z is constant;
    and for a lookup with a' = a, s' = s likewise.  Checks the scan and the inversion at full size without a CPU pass."""
    k = 19
    n = 1 << k
    rng = np.random.default_rng(19)
    d = zk.Domain(ctx, 5, k)
    omega = fr_ints(d.constant(0)[None])[0]
    beta, gamma, z0 = (rand_fr(rng, 1)[0] for _ in range(3))
    # identity permutation columns: delta^j * omega^i, built with the device NTT-free way: powers by prefix product
    om = np.repeat(fr_arr([omega]), n, axis=0)
    d_om = ctx.upload(om)
    sig = []
    for j in range(2):
        dj = ctx.alloc(n * 32)
        ctx.prefix_product_dev(d_om, dj, n, fr_arr([pow(bn.FR_DELTA, j, P)])[0])
        sig.append(dj)
    vals = [ctx.upload(rand_fr(rng, n)) for _ in range(2)]
    d_z = ctx.alloc(n * 32)
    d.permutation_product_dev(vals, sig, beta, gamma, fr_arr([1])[0], z0, d_z)
    z = ctx.download(d_z, (n, 4))
    assert (z == z0).all()
    d.lookup_product_dev(vals[0], vals[1], vals[0], vals[1], beta, gamma, d_z)
    assert (ctx.download(d_z, (n, 4)) == fr_arr([1])[0]).all()
    for p in [d_om, d_z] + sig + vals:
        ctx.free(p)
    d.close()


@pytest.mark.parametrize("n", [1, 17, 4096, 100003, 1 << 19])
def test_eval_polynomial(ctx, n):
    import halo2_model as hm
    rng = np.random.default_rng(n)
    c = rand_fr(rng, n)
    x = rand_fr(rng, 1)[0]
    d = ctx.upload(c)
    got = fr_ints(ctx.eval_polynomial_dev(d, n, x)[None])[0]
    assert got == hm.eval_poly(fr_ints(c), fr_ints(x[None])[0])
    assert not ctx.eval_polynomial_dev(0, 0, x).any()
    ctx.free(d)


def test_eval_polynomials_batch(ctx):
    """many (polynomial, point) pairs in one call equal the single calls"""
    rng = np.random.default_rng(8)
    n = 5000
    polys = [ctx.upload(rand_fr(rng, n)) for _ in range(5)]
    queries = [(polys[i % 5], rand_fr(rng, 1)[0]) for i in range(23)]
    got = ctx.eval_polynomials_dev([q[0] for q in queries], np.stack([q[1] for q in queries]), n)
    for (ptr, x), g_ in zip(queries, got):
        assert np.array_equal(g_, ctx.eval_polynomial_dev(ptr, n, x))
    assert ctx.eval_polynomials_dev([], np.zeros((0, 4), np.uint64), n).shape == (0, 4)
    for p_ in polys:
        ctx.free(p_)


def test_grand_products_golden(ctx):
    g = golden("products.json")
    for c in g["permutation"]:
        n = 1 << c["k"]
        d = zk.Domain(ctx, 3, c["k"])
        d_vals = [ctx.upload(hex_fr(v)) for v in c["values"]]
        d_sigs = [ctx.upload(hex_fr(v)) for v in c["sigmas"]]
        d_z = ctx.alloc(n * 32)
        d.permutation_product_dev(d_vals, d_sigs, hex_fr([c["beta"]])[0], hex_fr([c["gamma"]])[0],
                                  fr_arr([pow(bn.FR_DELTA, c["first_column"], P)])[0], hex_fr([c["z0"]])[0], d_z)
        assert np.array_equal(ctx.download(d_z, (n, 4)), hex_fr(c["z"]))
        for p in d_vals + d_sigs + [d_z]:
            ctx.free(p)
        d.close()
    for c in g["lookup"]:
        n = 1 << c["k"]
        d = zk.Domain(ctx, 4, c["k"])
        ptrs = [ctx.upload(hex_fr(c[key])) for key in ("input", "table", "permuted_input", "permuted_table")]
        d_z = ctx.alloc(n * 32)
        d.lookup_product_dev(ptrs[0], ptrs[1], ptrs[2], ptrs[3], hex_fr([c["beta"]])[0], hex_fr([c["gamma"]])[0], d_z)
        assert np.array_equal(ctx.download(d_z, (n, 4)), hex_fr(c["z"]))
        for p in ptrs + [d_z]:
            ctx.free(p)
        d.close()
    for c in g["batch_invert"]:
        a = hex_fr(c["input"])
        d_a = ctx.upload(a)
        ctx.batch_invert_dev(0, d_a, len(a))
        assert np.array_equal(ctx.download(d_a, (len(a), 4)), hex_fr(c["output"]))
        ctx.free(d_a)
    for c in g["eval_polynomial"]:
        co_ = hex_fr(c["coeff"])
        d_c = ctx.upload(co_)
        assert np.array_equal(ctx.eval_polynomial_dev(d_c, len(co_), hex_fr([c["x"]])[0]), hex_fr([c["value"]])[0])
        ctx.free(d_c)


def test_replay_script_small_shape(ctx, capsys, monkeypatch, tmp_path):
    """replay.py (the create_proof operator multiset) runs end to end on a small shape and reports every phase"""
    import json
    import runpy
    import sys
    log = tmp_path / "replay.log"
    monkeypatch.setattr(sys, "argv", ["replay.py", "--k", "10", "--advice", "9", "--fixed", "4", "--lookups", "3",
                                      "--gate-polys", "20", "--degree", "5", "--perm-cols", "5",
                                      "--reference-log", str(log)])
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "replay.py"), run_name="__main__")
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["msm_count"] == 9 + 3 * 3 + line["shape"]["permutation_sets"] + 1 + 4
    assert {"evaluate_h", "coeff_to_extended", "advice_commit_msm", "eval_polynomial", "shplonk_multiopen"} <= set(line["phases_s"])
    assert line["shplonk"]["queries"] == line["eval_polynomial_count"] and line["shplonk"]["kate_division_calls"] >= 4
    text = log.read_text().splitlines()
    assert text[0].startswith("[Circuit summary]/id:replay_k10, max_num_row:1024, degree:10,")
    assert any(l.startswith("End:") and "[Create_proof]/replay_k10/" in l for l in text)


# ------------------------------------------------------------------------------------------------ SHPLONK opening
@pytest.mark.parametrize("n,num", [(1, 1), (5, 2), (256, 3), (2049, 7), (70001, 66), (1 << 16, 131)])
def test_poly_lincomb(ctx, n, num):
    """sum_k c_k p_k with ragged lengths, odd and even term counts, more terms than one folding interval, aliasing."""
    rng = np.random.default_rng(n + num)
    lens = [n if k % 3 != 1 else max(1, n // (k + 1)) for k in range(num)]
    polys = [rand_fr(rng, l) for l in lens]
    coeffs = rand_fr(rng, num)
    if num > 2:
        coeffs[1] = 0
        coeffs[2] = fr_arr([P - 1])[0]
    d_polys = [ctx.upload(p) for p in polys]
    d_out = ctx.alloc(n * 32)
    ctx.poly_lincomb_dev(d_polys, lens, coeffs, n, d_out)
    got = fr_ints(ctx.download(d_out, (n, 4)))
    cs = fr_ints(coeffs)
    ps = [fr_ints(p) for p in polys]
    check = range(n) if n <= 4096 else [0, 1, 2, n // 7, n // 3, n // 2, n - 2, n - 1]
    for i in check:
        assert got[i] == sum(c * p[i] for c, p in zip(cs, ps) if i < len(p)) % P, i
    # in place on the first polynomial (full length)
    ctx.poly_lincomb_dev(d_polys, lens, coeffs, n, d_polys[0])
    assert np.array_equal(ctx.download(d_polys[0], (n, 4)), ctx.download(d_out, (n, 4)))
    for p in d_polys + [d_out]:
        ctx.free(p)


def test_poly_lincomb_empty_and_errors(ctx):
    d_out = ctx.upload(rand_fr(np.random.default_rng(0), 8))
    ctx.poly_lincomb_dev([], [], np.zeros((0, 4), np.uint64), 8, d_out)     # empty sum = zero polynomial
    assert not ctx.download(d_out, (8, 4)).any()
    with pytest.raises(ffi.ZkhipError):
        ctx.poly_lincomb_dev([d_out], [9], fr_arr([1]), 8, d_out)           # longer than the output
    ctx.free(d_out)


@pytest.mark.parametrize("n", [2, 3, 8, 9, 2047, 2048, 2049, 70001, 1 << 19, (1 << 20) + 5])
def test_kate_division(ctx, n):
    import halo2_model as hm
    rng = np.random.default_rng(n)
    a = rand_fr(rng, n)
    z = rand_fr(rng, 1)[0]
    d_a, d_q = ctx.upload(a), ctx.alloc(n * 32)
    ctx.kate_division_dev(d_a, n, z, d_q)
    q = ctx.download(d_q, (n - 1, 4))
    zi = fr_ints(z[None])[0]
    if n <= 70001:
        assert fr_ints(q) == hm.kate_division(fr_ints(a), zi)
    else:
        # a(X) - a(z) = q(X) (X - z) at a random x, and the top / bottom coefficients from the recurrence
        x = rand_fr(rng, 1)[0]
        xi = fr_ints(x[None])[0]
        ax, az = (fr_ints(ctx.eval_polynomial_dev(d_a, n, w)[None])[0] for w in (x, z))
        qx = fr_ints(ctx.eval_polynomial_dev(d_q, n - 1, x)[None])[0]
        assert (ax - az) % P == qx * (xi - zi) % P
        ai, qi = fr_ints(a[-3:]), fr_ints(q[-2:])
        assert qi[1] == ai[2] and qi[0] == (ai[1] + zi * ai[2]) % P
        assert (fr_ints(a[:1])[0] + zi * fr_ints(q[:1])[0]) % P == az
    ctx.free(d_a)
    ctx.free(d_q)


class _ReplayTranscript:
    """Stands in for the Blake2b transcript: hands out preset challenges, records the points written."""

    def __init__(self, challenges):
        self.challenges, self.points = list(challenges), []

    def squeeze_challenge(self):
        return self.challenges.pop(0)

    def write_point(self, jac):
        self.points.append(np.array(jac, dtype=np.uint64))


def _shplonk_case(rng, k, shape):
    """shape: list of rotation tuples, one per polynomial, as halo2 queries them (advice at 0 / +1 / -1, fixed at 0, a
    permutation product at 0, +1 and last row ...).  Returns polys (canonical coefficients) and consistent queries."""
    import halo2_model as hm
    n = 1 << k
    dom = hm.EvaluationDomain(3, k)
    x = int(rng.integers(1, 2 ** 62))
    polys, queries = {}, []
    for key, rots in enumerate(shape):
        polys[key] = fr_ints(rand_fr(rng, n))
    # halo2 orders queries by argument, not by polynomial: interleave the rotations across polynomials
    for r_idx in range(max(len(r) for r in shape)):
        for key, rots in enumerate(shape):
            if r_idx < len(rots):
                pt = dom.rotate_omega(x, rots[r_idx])
                queries.append((key, pt, hm.eval_poly(polys[key], pt)))
    return polys, queries


@pytest.mark.parametrize("k,shape", [
    (5, [(0,)]),
    (6, [(0,), (0, 1), (0, 1), (0, -1, 1), (0,), (1,)]),
    (9, [(0, 1, -1)] * 3 + [(0,)] * 5 + [(0, 1)] * 2 + [(0, 1, -5)] + [(-1, 0)]),
    (12, [(0,), (0, 1), (0, 1, -3), (0,)]),
])
def test_shplonk_create_proof_vs_model(ctx, co, k, shape):
    """The device SHPLONK prover against the big-int model: both quotient polynomials coefficient by coefficient, and
    both commitments against the closed form for bases (i+1)G."""
    import halo2_model as hm
    from zkevm_circuits_amd import shplonk
    n = 1 << k
    rng = np.random.default_rng(k)
    polys, queries = _shplonk_case(rng, k, shape)
    y, v, u = (int(c) for c in rng.integers(2, 2 ** 62, size=3))
    commit = lambda coeffs: bn.g1_mul(bn.G1_GENERATOR, sum(c * (i + 1) for i, c in enumerate(coeffs)) % P)
    h1, h2, h_x, final = hm.shplonk_create_proof(polys, queries, y, v, u, commit)

    bid = ctx.register_bases(co.g1_bases(n))
    d_polys = {key: ctx.upload(fr_arr(c)) for key, c in polys.items()}
    tr = _ReplayTranscript([y, v, u])
    prover = shplonk.ProverSHPLONK(ctx, bid, n)
    d_h, d_f = prover.create_proof(d_polys, [shplonk.Query(*q) for q in queries], tr)
    assert fr_ints(ctx.download(d_h, (n, 4))) == h_x
    assert fr_ints(ctx.download(d_f, (n, 4))) == final + [0]
    assert [affine_pt(ctx.g1_to_affine(p)) for p in tr.points] == [h1, h2]
    assert not tr.challenges
    # inconsistent evaluation: the linearisation polynomial cannot vanish at u
    bad = [shplonk.Query(*q) for q in queries]
    bad[0].eval = (bad[0].eval + 1) % P
    with pytest.raises(ValueError):
        prover.create_proof(d_polys, bad, _ReplayTranscript([y, v, u]))
    for p in list(d_polys.values()) + [d_h, d_f]:
        ctx.free(p)
    ctx.release_bases(bid)


def test_shplonk_golden(ctx, co):
    from zkevm_circuits_amd import shplonk
    g = golden("shplonk.json")
    for c in g["cases"]:
        n = 1 << c["k"]
        bid = ctx.register_bases(co.g1_bases(n))
        d_polys = {i: ctx.upload(hex_fr(p)) for i, p in enumerate(c["polys"])}
        tr = _ReplayTranscript([int(c[name], 16) for name in ("y", "v", "u")])
        queries = [shplonk.Query(key, int(pt, 16), int(ev, 16)) for key, pt, ev in c["queries"]]
        d_h, d_f = shplonk.ProverSHPLONK(ctx, bid, n).create_proof(d_polys, queries, tr)
        assert np.array_equal(ctx.download(d_h, (n, 4)), hex_fr(c["h_x"]))
        assert np.array_equal(ctx.download(d_f, (n - 1, 4)), hex_fr(c["final"]))
        assert [affine_pt(ctx.g1_to_affine(p)) for p in tr.points] == [hex_pt(c["h1"]), hex_pt(c["h2"])]
        for p in list(d_polys.values()) + [d_h, d_f]:
            ctx.free(p)
        ctx.release_bases(bid)
    for c in g["kate_division"]:
        a = hex_fr(c["a"])
        d_a, d_q = ctx.upload(a), ctx.alloc(len(a) * 32)
        ctx.kate_division_dev(d_a, len(a), hex_fr([c["point"]])[0], d_q)
        assert np.array_equal(ctx.download(d_q, (len(a) - 1, 4)), hex_fr(c["q"]))
        ctx.free(d_a)
        ctx.free(d_q)
    for c in g["lincomb"]:
        d_ps = [ctx.upload(hex_fr(p)) for p in c["polys"]]
        d_o = ctx.alloc(c["n"] * 32)
        ctx.poly_lincomb_dev(d_ps, [len(p) for p in c["polys"]], hex_fr(c["coeffs"]), c["n"], d_o)
        assert np.array_equal(ctx.download(d_o, (c["n"], 4)), hex_fr(c["out"]))
        for p in d_ps + [d_o]:
            ctx.free(p)


# ------------------------------------------------------------------------------------------------ coset split (multi-GPU)
def _compact(a, cosets, e):
    """extended vector -> compact layout of the listed cosets (row q * m + s = extended row q * 2^e + cosets[s])"""
    a = np.asarray(a)
    out = np.empty((a.shape[0] // (1 << e) * len(cosets), 4), dtype=np.uint64)
    for s, j in enumerate(cosets):
        out[s::len(cosets)] = a[j::1 << e]
    return out


@pytest.mark.parametrize("j,k", [(3, 6), (5, 9), (9, 11), (17, 8)])
def test_coset_subsets_of_the_extended_domain(ctx, co, j, k):
    """coeff_to_cosets = the listed rows of coeff_to_extended; gather / scatter between the two layouts"""
    d, od = zk.Domain(ctx, j, k), co.Domain(j, k, threads=4)
    e, n, en = d.extended_k - k, d.n, d.extended_len
    rng = np.random.default_rng(j * 100 + k)
    batch = 3
    coeff = rand_fr(rng, batch * n)
    full = [od.coeff_to_extended(coeff[b * n:(b + 1) * n]) for b in range(batch)]
    d_c = ctx.upload(coeff)
    lists = [list(range(1 << e)), [(1 << e) - 1], list(range(1, 1 << e, 2)) or [0], [0]]
    if e >= 2:
        lists.append([3, 0])
    for cosets in lists:
        m = len(cosets)
        d_o = ctx.alloc(batch * n * m * 32)
        d.coeff_to_cosets_dev(d_c, cosets, d_o, batch)
        got = ctx.download(d_o, (batch * n * m, 4))
        for b in range(batch):
            assert np.array_equal(got[b * n * m:(b + 1) * n * m], _compact(full[b], cosets, e)), (cosets, b)
        ctx.free(d_o)
        d_e, d_m, d_back = ctx.upload(full[0]), ctx.alloc(n * m * 32), ctx.upload(np.zeros((en, 4), np.uint64))
        d.extended_to_cosets_dev(d_e, cosets, d_m)
        assert np.array_equal(ctx.download(d_m, (n * m, 4)), _compact(full[0], cosets, e))
        d.cosets_to_extended_dev(d_m, cosets, d_back)
        back = ctx.download(d_back, (en, 4))
        keep = np.zeros(en, bool)
        for jj in cosets:
            keep[jj::1 << e] = True
        assert np.array_equal(back[keep], full[0][keep]) and not back[~keep].any()
        for p in (d_e, d_m, d_back):
            ctx.free(p)
    with pytest.raises(ffi.ZkhipError):
        d.coeff_to_cosets_dev(d_c, [0, 0], d_c, 1)               # repeated coset
    with pytest.raises(ffi.ZkhipError):
        d.coeff_to_cosets_dev(d_c, [1 << e], d_c, 1)             # out of range
    ctx.free(d_c)
    d.close()


@pytest.mark.parametrize("k,kw", [
    (7, dict()),
    (9, dict(num_advice=12, num_gate_polys=40, num_lookups=3, num_perm_cols=7, rotations=(-5, -1, 0, 1, 3))),
    (8, dict(num_advice=20, num_fixed=8, num_gate_polys=100, max_degree=9, num_lookups=2, num_perm_cols=11)),
])
def test_evaluate_h_split_by_coset_matches_whole(ctx, co, k, kw):
    """what each GPU of a 1 / 2 / 4 / 8-way split computes (its cosets only, compact columns) put back together equals
    the single-GPU evaluate_h, in extended and in coefficient form"""
    from zkevm_circuits_amd import sharding
    cs = synthetic.synthetic_cs(**kw)
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    e = d.extended_k - k
    rng = np.random.default_rng(k + 50)
    call, coeff, ext = _random_call(cs, rng, d.n, d.extended_len, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    whole = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
    pk = dict(fixed=call.fixed, l0=call.l0, l_last=call.l_last, l_active=call.l_active, perm_sigma=call.perm_sigma,
              perm_z=call.perm_z)
    for world in [w for w in (1, 2, 4, 8) if w <= (1 << e)]:
        for form in (ffi.FORM_EXTENDED, ffi.FORM_COEFF):
            back = np.zeros_like(whole)
            for rank in range(world):
                cosets = sharding.cosets_for_rank(d.extended_k, k, rank, world)
                c = lambda a: _compact(a, cosets, e)
                call.cosets = cosets
                call.fixed = [c(a) for a in pk["fixed"]]
                call.l0, call.l_last, call.l_active = c(pk["l0"]), c(pk["l_last"]), c(pk["l_active"])
                call.perm_sigma, call.perm_z = [c(a) for a in pk["perm_sigma"]], [c(a) for a in pk["perm_z"]]
                if form == ffi.FORM_EXTENDED:
                    call.advice, call.instance = [c(a) for a in ext["advice"]], [c(a) for a in ext["instance"]]
                    call.lookups = [tuple(c(p) for p in t) for t in ext["lookups"]]
                else:
                    call.advice, call.instance, call.lookups = coeff["advice"], coeff["instance"], coeff["lookups"]
                part = d.evaluate_h(call.build(ffi.MEM_HOST, form))
                assert part.shape[0] == d.n * len(cosets)
                for s, j in enumerate(cosets):
                    back[j::1 << e] = part[s::len(cosets)]
            assert np.array_equal(back, whole), (world, form)
    d.close()


@pytest.mark.parametrize("k,kw", [
    (9, dict(num_advice=12, num_gate_polys=40, num_lookups=3, num_perm_cols=7)),
    (10, dict(num_advice=40, num_fixed=8, num_gate_polys=300, max_degree=9, num_lookups=2, num_perm_cols=11)),
])
def test_evaluate_h_row_windows_match_whole(ctx, co, k, kw):
    """a row window of the extended domain, and of a coset's compact layout, gives exactly those rows of the whole
    evaluation: what ranks that share a coset compute (more GPUs than cosets), with the columns whole and no exchange of
    neighbouring rows; windows that are no multiple of a workgroup and windows at both ends included"""
    from zkevm_circuits_amd import sharding
    cs = synthetic.synthetic_cs(**kw)
    d, od = zk.Domain(ctx, cs.degree(), k), co.Domain(cs.degree(), k, threads=8)
    e, n, en = d.extended_k - k, d.n, d.extended_len
    rng = np.random.default_rng(k + 70)
    call, coeff, ext = _random_call(cs, rng, n, en, od)
    call.advice, call.instance, call.lookups = ext["advice"], ext["instance"], ext["lookups"]
    whole = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
    for first, count in ((0, 256), (en - 512, 512), (en // 2 - 100, 777), (3, 1), (0, en)):
        call.rows = (first, count)
        part = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
        assert part.shape[0] == count and np.array_equal(part, whole[first:first + count]), (first, count)
    call.rows = (en - 5, 6)
    with pytest.raises(ffi.ZkhipError):
        d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
    # sixteen ranks over 2^e cosets: every rank a row range of one coset, from the coefficient form
    pk = dict(fixed=call.fixed, l0=call.l0, l_last=call.l_last, l_active=call.l_active, perm_sigma=call.perm_sigma,
              perm_z=call.perm_z)
    world = 2 << e
    back = np.zeros_like(whole)
    for cosets, first, count in sharding.coset_partition(d.extended_k, k, world):
        c = lambda a: _compact(a, cosets, e)
        call.cosets, call.rows = cosets, (first, count)
        call.fixed = [c(a) for a in pk["fixed"]]
        call.l0, call.l_last, call.l_active = c(pk["l0"]), c(pk["l_last"]), c(pk["l_active"])
        call.perm_sigma, call.perm_z = [c(a) for a in pk["perm_sigma"]], [c(a) for a in pk["perm_z"]]
        call.advice, call.instance, call.lookups = coeff["advice"], coeff["instance"], coeff["lookups"]
        part = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_COEFF))
        assert len(cosets) == 1 and part.shape[0] == count == n // 2
        back[cosets[0]::1 << e][first:first + count] = part
    assert np.array_equal(back, whole)
    d.close()


# ------------------------------------------------------------------------------------------------ lookup permutation
def _lookup_pair(rng, n, usable, table_size, distinct_inputs):
    """table column: `table_size` distinct values padded with repeats of the first one (as halo2 tables are padded);
    input column: draws from a subset of the table"""
    table_vals = [int(v) for v in rng.integers(1, 2 ** 62, size=table_size)] + [0, P - 1]
    table = [table_vals[i] if i < len(table_vals) else table_vals[0] for i in range(n)]
    order = rng.permutation(usable)
    table[:usable] = [table[i] for i in order]
    pool = [table_vals[int(i)] for i in rng.integers(0, len(table_vals), size=distinct_inputs)]
    inputs = [pool[int(i)] for i in rng.integers(0, len(pool), size=n)]
    return inputs, table


@pytest.mark.parametrize("k,blinding,table_size,distinct", [(4, 5, 6, 3), (7, 5, 100, 100), (10, 6, 300, 17),
                                                            (12, 5, 4000, 4000), (16, 7, 1 << 15, 5000)])
def test_lookup_permute_vs_model(ctx, k, blinding, table_size, distinct):
    import halo2_model as hm
    n = 1 << k
    usable = n - (blinding + 1)
    table_size = min(table_size, usable - 2)
    rng = np.random.default_rng(k)
    inputs, table = _lookup_pair(rng, n, usable, table_size, distinct)
    exp_a, exp_s = hm.permute_expression_pair(inputs, table, usable)
    d = zk.Domain(ctx, 3, k)
    d_a, d_s = ctx.upload(fr_arr(inputs)), ctx.upload(fr_arr(table))
    marker = rand_fr(rng, n)
    d_pa, d_ps = ctx.upload(marker), ctx.upload(marker)
    d.lookup_permute_dev(d_a, d_s, usable, d_pa, d_ps)
    got_a, got_s = ctx.download(d_pa, (n, 4)), ctx.download(d_ps, (n, 4))
    assert fr_ints(got_a[:usable]) == exp_a
    assert fr_ints(got_s[:usable]) == exp_s
    # the blinding rows belong to the caller
    assert np.array_equal(got_a[usable:], marker[usable:]) and np.array_equal(got_s[usable:], marker[usable:])
    # the lookup argument's row conditions hold: A'[i] = S'[i] or A'[i] = A'[i-1], and A'[0] = S'[0]
    assert exp_a[0] == exp_s[0] and all(a == s or a == p_ for a, s, p_ in zip(exp_a[1:], exp_s[1:], exp_a[:-1]))
    # an input value outside the table is the constraint-system failure upstream reports
    bad = list(inputs)
    bad[3] = 12345678901234567890123
    ctx.upload(fr_arr(bad), d_a)
    with pytest.raises(ffi.ZkhipError):
        d.lookup_permute_dev(d_a, d_s, usable, d_pa, d_ps)
    with pytest.raises(ValueError):
        hm.permute_expression_pair(bad, table, usable)
    for p_ in (d_a, d_s, d_pa, d_ps):
        ctx.free(p_)
    d.close()


def test_lookup_permute_full_size_feeds_the_grand_product(ctx):
    """k = 19: permuted columns from the device sort, then the lookup grand product over them telescopes to one on the
    last usable row (the property the verifier's l_last check relies on)"""
    k, blinding = 19, 5
    n = 1 << k
    usable = n - (blinding + 1)
    rng = np.random.default_rng(19)
    table = rand_fr(rng, n)
    pick = rng.integers(0, usable, size=n)
    inputs = table[pick]
    d = zk.Domain(ctx, 3, k)
    d_a, d_s, d_pa, d_ps, d_z = (ctx.upload(x) for x in (inputs, table, inputs, table, table))
    d.lookup_permute_dev(d_a, d_s, usable, d_pa, d_ps)
    pa, ps = ctx.download(d_pa, (n, 4)), ctx.download(d_ps, (n, 4))
    # sortedness of A' (canonical order) at sampled positions and the row conditions everywhere, on limbs
    same_as = (pa[1:usable] == ps[1:usable]).all(axis=1)
    same_prev = (pa[1:usable] == pa[:usable - 1]).all(axis=1)
    assert (same_as | same_prev).all() and (pa[0] == ps[0]).all()
    idx = rng.integers(0, usable - 1, size=2000)
    vals = fr_ints(pa[idx]), fr_ints(pa[idx + 1])
    assert all(a <= b for a, b in zip(*vals))
    beta, gamma = rand_fr(rng, 2)
    d.lookup_product_dev(d_a, d_s, d_pa, d_ps, beta, gamma, d_z)
    z = ctx.download(d_z, (n, 4))
    one = fr_arr([1])[0]
    assert np.array_equal(z[0], one) and np.array_equal(z[usable], one)
    for p_ in (d_a, d_s, d_pa, d_ps, d_z):
        ctx.free(p_)
    d.close()


@pytest.mark.parametrize("exact", [False, True])
def test_lookup_permute_batch_of_mixed_columns(ctx, exact, monkeypatch):
    """five lookups permuted in one call, each stressing another side of the radix sort: table indices below 2^10 (two
    sort passes), full-width values, values whose varying bits straddle two key windows so that different values agree on
    the whole 48-bit window (the batch then takes the exact re-sort), a column of one repeated value, and 62-bit values;
    each against the model of upstream's loop.  exact: the re-sort on all 256 bits forced for the whole batch."""
    import halo2_model as hm
    if exact:
        monkeypatch.setenv("ZKHIP_LOOKUP_EXACT_SORT", "1")
    k, blinding = 11, 5
    n = 1 << k
    usable = n - (blinding + 1)
    rng = np.random.default_rng(31 + exact)
    cases = []
    small_table = [i % 1000 for i in range(n)]
    cases.append(([int(v) for v in rng.integers(0, 1000, size=n)], small_table))
    wide = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(300)]
    cases.append(([wide[int(i)] for i in rng.integers(0, 300, size=n)], [wide[i % 300] for i in range(n)]))
    # 2^200 + i and i: the window ends at bit 200, where every small i looks alike
    split = [(1 << 200) + i for i in range(200)] + list(range(1, 200))
    cases.append(([split[int(i)] for i in rng.integers(0, len(split), size=n)], [split[i % len(split)] for i in range(n)]))
    cases.append(([7] * n, [7] * n))
    cases.append(_lookup_pair(rng, n, usable, 500, 40))
    d = zk.Domain(ctx, 3, k)
    d_in = [ctx.upload(fr_arr(a)) for a, _ in cases]
    d_tb = [ctx.upload(fr_arr(t)) for _, t in cases]
    d_pa = [ctx.alloc(n * 32) for _ in cases]
    d_ps = [ctx.alloc(n * 32) for _ in cases]
    d.lookup_permute_batch_dev(d_in, d_tb, usable, d_pa, d_ps)
    for (a, t), pa, ps in zip(cases, d_pa, d_ps):
        exp_a, exp_s = hm.permute_expression_pair(a, t, usable)
        assert fr_ints(ctx.download(pa, (n, 4))[:usable]) == exp_a
        assert fr_ints(ctx.download(ps, (n, 4))[:usable]) == exp_s
    # the error names the lookup whose input left its table
    bad = list(cases[3][0])
    bad[5] = 8
    ctx.upload(fr_arr(bad), d_in[3])
    with pytest.raises(ffi.ZkhipError, match="lookup 3"):
        d.lookup_permute_batch_dev(d_in, d_tb, usable, d_pa, d_ps)
    d.lookup_permute_batch_dev([], [], usable, [], [])
    for p_ in d_in + d_tb + d_pa + d_ps:
        ctx.free(p_)
    d.close()


def test_lookup_permute_golden(ctx):
    for c in golden("lookup_permute.json")["cases"]:
        n, usable = 1 << c["k"], c["usable_rows"]
        d = zk.Domain(ctx, 3, c["k"])
        d_a, d_s = ctx.upload(hex_fr(c["input"])), ctx.upload(hex_fr(c["table"]))
        d_pa, d_ps = ctx.alloc(n * 32), ctx.alloc(n * 32)
        d.lookup_permute_dev(d_a, d_s, usable, d_pa, d_ps)
        assert np.array_equal(ctx.download(d_pa, (usable, 4)), hex_fr(c["permuted_input"]))
        assert np.array_equal(ctx.download(d_ps, (usable, 4)), hex_fr(c["permuted_table"]))
        for p_ in (d_a, d_s, d_pa, d_ps):
            ctx.free(p_)
        d.close()


# ------------------------------------------------------------------------------------------------ lookup compression
def _direct_expr(e, cols, ch, idx, size):
    t = e[0]
    if t == "const":
        return e[1]
    if t in ("fixed", "advice", "instance"):
        return cols[t][e[1]][(idx + e[2]) % size]
    if t == "challenge":
        return ch[e[1]]
    if t == "neg":
        return (-_direct_expr(e[1], cols, ch, idx, size)) % P
    if t == "sum":
        return (_direct_expr(e[1], cols, ch, idx, size) + _direct_expr(e[2], cols, ch, idx, size)) % P
    if t == "prod":
        return _direct_expr(e[1], cols, ch, idx, size) * _direct_expr(e[2], cols, ch, idx, size) % P
    return _direct_expr(e[1], cols, ch, idx, size) * e[2] % P


@pytest.mark.parametrize("k,kw", [(6, dict()), (9, dict(num_advice=12, num_fixed=5, num_lookups=4, lookup_width=4,
                                                        rotations=(-3, -1, 0, 1, 2)))])
def test_lookup_compress_expressions_on_lagrange_basis(ctx, k, kw):
    """compress_expressions: each lookup side folded in theta over the 2^k rows, against direct evaluation of the
    expression trees; host and device memory"""
    cs = synthetic.synthetic_cs(**kw)
    d = zk.Domain(ctx, cs.degree(), k)
    n = d.n
    rng = np.random.default_rng(k)
    cols = {"fixed": [rand_fr(rng, n) for _ in range(cs.num_fixed_columns)],
            "advice": [rand_fr(rng, n) for _ in range(cs.num_advice_columns)],
            "instance": [rand_fr(rng, n) for _ in range(cs.num_instance_columns)]}
    ints = {t: [fr_ints(c) for c in v] for t, v in cols.items()}
    challenges = rand_fr(rng, cs.num_challenges)
    ch = fr_ints(challenges)
    theta = rand_fr(rng, 1)[0]
    th = fr_ints(theta[None])[0]
    for lk in cs.lookups:
        for exprs in (lk.input_expressions, lk.table_expressions):
            exp = []
            for row in range(n):
                acc = 0
                for e in exprs:
                    acc = (acc * th + _direct_expr(e, ints, ch, row, n)) % P
                exp.append(acc)
            call = evalh_host.lookup_compress_call(cs, exprs, cols["fixed"], cols["advice"], cols["instance"], challenges, theta)
            got = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
            assert got.shape[0] == n and fr_ints(got) == exp
    # device-resident columns
    dev = {t: [ctx.upload(c) for c in v] for t, v in cols.items()}
    d_out = ctx.alloc(n * 32)
    call = evalh_host.lookup_compress_call(cs, cs.lookups[0].input_expressions, dev["fixed"], dev["advice"], dev["instance"],
                                           challenges, theta)
    d.evaluate_h(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_out)
    call_h = evalh_host.lookup_compress_call(cs, cs.lookups[0].input_expressions, cols["fixed"], cols["advice"],
                                             cols["instance"], challenges, theta)
    assert np.array_equal(ctx.download(d_out, (n, 4)), d.evaluate_h(call_h.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)))
    # every lookup's two sides in one call (the kernels back to back, one wait): the same values as graph by graph; twice,
    # so that the second call finds its programs cached and rewrites the staged tables of the first
    expected = []
    for lk in cs.lookups:
        for exprs in (lk.input_expressions, lk.table_expressions):
            c = evalh_host.lookup_compress_call(cs, exprs, cols["fixed"], cols["advice"], cols["instance"], challenges, theta)
            expected.append(d.evaluate_h(c.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)))
    d_ins = [ctx.alloc(n * 32) for _ in cs.lookups]
    d_tbs = [ctx.alloc(n * 32) for _ in cs.lookups]
    for subset in (cs.lookups, cs.lookups[:1]):
        evalh_host.compress_lookups_dev(d, cs, subset, dev["fixed"], dev["advice"], dev["instance"], challenges, theta,
                                        d_ins[:len(subset)], d_tbs[:len(subset)])
        for i in range(len(subset)):
            assert np.array_equal(ctx.download(d_ins[i], (n, 4)), expected[2 * i])
            assert np.array_equal(ctx.download(d_tbs[i], (n, 4)), expected[2 * i + 1])
    evalh_host.compress_lookups_dev(d, cs, [], dev["fixed"], dev["advice"], dev["instance"], challenges, theta, [], [])
    with pytest.raises(ffi.ZkhipError):       # a null output
        evalh_host.compress_lookups_dev(d, cs, cs.lookups[:1], dev["fixed"], dev["advice"], dev["instance"], challenges, theta,
                                        [0], d_tbs[:1])
    for p_ in d_ins + d_tbs:
        ctx.free(p_)
    # the Lagrange basis takes neither lookups nor coefficient form
    with pytest.raises(ffi.ZkhipError):
        d.evaluate_h(call_h.build(ffi.MEM_HOST, ffi.FORM_COEFF))
    for v in dev.values():
        for p_ in v:
            ctx.free(p_)
    ctx.free(d_out)
    d.close()


# ------------------------------------------------------------------------------------------------ key generation
@pytest.mark.parametrize("k,j,blinding,ncols", [(4, 3, 3, 2), (6, 5, 5, 3), (9, 9, 6, 5)])
def test_keygen_polys_vs_model(ctx, k, j, blinding, ncols):
    """fixed cosets, permutation polynomials from copy-constraint cycles, and l0 / l_last / l_active_row on the extended
    domain against the big-int model"""
    import halo2_model as hm
    from zkevm_circuits_amd import keygen
    d, md = zk.Domain(ctx, j, k), hm.EvaluationDomain(j, k)
    n, en = d.n, d.extended_len
    rng = np.random.default_rng(k)
    fixed = [rand_fr(rng, n) for _ in range(2)]
    asm = keygen.Assembly(n, ncols)
    for _ in range(3 * n):
        a = [int(x) for x in rng.integers(0, ncols, size=2)]
        r = [int(x) for x in rng.integers(0, n - blinding - 1, size=2)]
        asm.copy(a[0], r[0], a[1], r[1])
    asm.copy(0, 0, 0, 0)
    with pytest.raises(IndexError):
        asm.copy(ncols, 0, 0, 0)
    # every cycle closes, and the mapping is a permutation of the cells
    cells = {(int(c), int(r)) for col in range(ncols) for c, r in zip(asm.map_col[col], asm.map_row[col])}
    assert len(cells) == ncols * n
    pk = keygen.keygen_polys(ctx, d, fixed, asm, blinding)
    for col, dp, dc in zip(fixed, pk.fixed_polys, pk.fixed_cosets):
        coeff = md.lagrange_to_coeff(fr_ints(col))
        assert fr_ints(ctx.download(dp, (n, 4))) == coeff
        assert fr_ints(ctx.download(dc, (en, 4))) == md.coeff_to_extended(coeff)
    mapping = [list(zip((int(x) for x in asm.map_col[c]), (int(x) for x in asm.map_row[c]))) for c in range(ncols)]
    sig = hm.permutation_sigmas(md.omega, mapping)
    for c in range(ncols):
        assert fr_ints(ctx.download(pk.permutations[c], (n, 4))) == sig[c]
        coeff = md.lagrange_to_coeff(sig[c])
        assert fr_ints(ctx.download(pk.permutation_polys[c], (n, 4))) == coeff
        assert fr_ints(ctx.download(pk.permutation_cosets[c], (en, 4))) == md.coeff_to_extended(coeff)
    # a mapping entry outside the argument is an error, not an out-of-bounds read
    bad_col = asm.map_col[0].copy()
    bad_col[1] = ncols
    d_bc, d_br = keygen._upload_words(ctx, bad_col), keygen._upload_words(ctx, asm.map_row[0])
    with pytest.raises(ffi.ZkhipError):
        d.permutation_sigma_dev(d_bc, d_br, ncols, pk.permutations[0])
    ctx.free(d_bc)
    ctx.free(d_br)
    l0, l_last, l_active = hm.lagrange_selectors(md, blinding)
    assert fr_ints(ctx.download(pk.l0, (en, 4))) == l0
    assert fr_ints(ctx.download(pk.l_last, (en, 4))) == l_last
    assert fr_ints(ctx.download(pk.l_active_row, (en, 4))) == l_active
    for p_ in pk.pointers():
        ctx.free(p_)
    d.close()


def test_keygen_sigmas_feed_the_permutation_product(ctx):
    """a witness that satisfies its copy constraints makes the permutation grand product close: z returns to one after
    the last usable row"""
    from zkevm_circuits_amd import keygen
    k, ncols, blinding = 10, 3, 5
    d = zk.Domain(ctx, 5, k)
    n = d.n
    usable = n - (blinding + 1)
    rng = np.random.default_rng(77)
    asm = keygen.Assembly(n, ncols)
    vals = [rand_fr(rng, n) for _ in range(ncols)]
    for _ in range(2 * n):
        c = [int(x) for x in rng.integers(0, ncols, size=2)]
        r = [int(x) for x in rng.integers(0, usable, size=2)]
        asm.copy(c[0], r[0], c[1], r[1])
    # equal values along every cycle: give each cell the value of its cycle's representative
    for c in range(ncols):
        for r in range(n):
            vals[c][r] = vals[int(asm.aux_col[c][r])][int(asm.aux_row[c][r])] if (asm.aux_col[c][r], asm.aux_row[c][r]) != (c, r) else vals[c][r]
    for c in range(ncols):      # a second pass settles cells whose representative was overwritten above
        for r in range(n):
            vals[c][r] = vals[int(asm.aux_col[c][r])][int(asm.aux_row[c][r])]
    pk = keygen.keygen_polys(ctx, d, [], asm, blinding)
    d_vals = [ctx.upload(v) for v in vals]
    d_z = ctx.alloc(n * 32)
    beta, gamma = rand_fr(rng, 2)
    one = fr_arr([1])[0]
    d.permutation_product_dev(d_vals, pk.permutations, beta, gamma, one, one, d_z)
    z = ctx.download(d_z, (n, 4))
    assert np.array_equal(z[0], one) and np.array_equal(z[usable], one)
    for p_ in pk.pointers() + d_vals + [d_z]:
        ctx.free(p_)
    d.close()


@pytest.mark.parametrize("k", [1, 3, 5])
def test_g1_fft_vs_definition(ctx, co, k):
    """the G1-valued transform behind g_lagrange against its O(n^2) definition, with and without the n^-1 scaling, and
    with an identity point among the inputs"""
    import halo2_model as hm
    n = 1 << k
    omega_inv = pow(bn.FR_ROOT_OF_UNITY, (1 << (bn.FR_S - k)) * ((1 << k) - 1), P)
    pts = [bn.g1_mul(bn.G1_GENERATOR, 3 * i + 1) for i in range(n)]
    if n >= 4:
        pts[2] = None
    n_inv = pow(n, P - 2, P)
    exp = hm.g1_fft(pts, omega_inv, k)
    got = ctx.g1_fft(affine_arr(pts), k, fr_arr([omega_inv])[0])
    assert [affine_pt(r) for r in got] == exp
    got = ctx.g1_fft(affine_arr(pts), k, fr_arr([omega_inv])[0], fr_arr([n_inv])[0])
    assert [affine_pt(r) for r in got] == [bn.g1_mul(e, n_inv) if e is not None else None for e in exp]
    with pytest.raises(ffi.ZkhipError):
        ctx.g1_fft(affine_arr(pts), k, fr_arr([5])[0])          # not a 2^k-th root of unity


@pytest.mark.parametrize("k", [8, 13])
def test_g_to_lagrange_commits_like_the_monomial_basis(ctx, co, k):
    """commit_lagrange(values) with the device-built g_lagrange equals commit(lagrange_to_coeff(values)) with g: the
    identity that makes the two bases interchangeable in create_proof"""
    from zkevm_circuits_amd import keygen
    n = 1 << k
    d = zk.Domain(ctx, 3, k)
    g = co.g1_bases(n)
    g_lagrange = keygen.g_to_lagrange(ctx, d, g)
    rng = np.random.default_rng(k)
    values = rand_fr(rng, n)
    coeff = d.lagrange_to_coeff(values)
    b_g, b_l = ctx.register_bases(g), ctx.register_bases(g_lagrange)
    assert np.array_equal(ctx.g1_to_affine(ctx.msm(b_l, values)), ctx.g1_to_affine(ctx.msm(b_g, coeff)))
    ctx.release_bases(b_g)
    ctx.release_bases(b_l)
    d.close()


def test_params_file_tables_commit_unchanged(ctx, co, tmp_path):
    """g and g_lagrange written as a raw-bytes parameter file and read back memory-mapped are the bases as they were:
    commitments from the file's tables equal those from the arrays in memory (no conversion pass between file and device)"""
    from zkevm_circuits_amd import formats, keygen
    k = 10
    n = 1 << k
    d = zk.Domain(ctx, 3, k)
    g = co.g1_bases(n)
    g_lagrange = keygen.g_to_lagrange(ctx, d, g)
    path = str(tmp_path / "k10.params")
    formats.write_params_raw(path, formats.ParamsRaw(k, g, g_lagrange, bytes(128), bytes(128)))
    p = formats.read_params_raw(path)
    assert p.k == k and np.array_equal(p.g, g) and np.array_equal(p.g_lagrange, g_lagrange)
    rng = np.random.default_rng(4)
    values = rand_fr(rng, n)
    b_file, b_mem = ctx.register_bases(p.g_lagrange), ctx.register_bases(g_lagrange)
    got = ctx.g1_to_affine(ctx.msm(b_file, values))
    assert np.array_equal(got, ctx.g1_to_affine(ctx.msm(b_mem, values)))
    assert np.array_equal(got, co.g1_to_affine(co.best_multiexp(values, g_lagrange, threads=8)))
    ctx.release_bases(b_file)
    ctx.release_bases(b_mem)
    d.close()


# ------------------------------------------------------------------------------------------------ evaluate_h at full size
class _DeviceColumn:
    """Lazy view of a device-resident column for the big-int model: rows are fetched one by one on first use."""
    R_INV = pow(1 << 256, P - 2, P)

    def __init__(self, tensor):
        self.t, self.cache = tensor, {}

    def __getitem__(self, i):
        v = self.cache.get(i)
        if v is None:
            limbs = self.t[i].cpu().numpy().view(np.uint64)
            v = sum(int(w) << (64 * j) for j, w in enumerate(limbs)) * self.R_INV % P
            self.cache[i] = v
        return v


@pytest.mark.parametrize("kw", [
    dict(num_advice=64, num_fixed=16, num_instance=1, num_gate_polys=256, max_degree=5, rotations=(-2, -1, 0, 1),
         num_lookups=16, lookup_width=3, num_perm_cols=8),
    dict(num_advice=40, num_fixed=8, num_instance=1, num_gate_polys=300, max_degree=9, rotations=(-5, -1, 0, 1, 3),
         num_lookups=4, lookup_width=4, num_perm_cols=11),
], ids=["bench-config-2^21", "degree-9-2^22"])
def test_evaluate_h_full_size_sampled_rows(ctx, kw):
    """k = 19 with all columns resident on the device -- the benchmark's constraint system (extended domain 2^21, 143
    columns) and a degree-9 one (extended domain 2^22, as the SuperCircuit): rows sampled over the whole domain (first,
    last, wrap-around neighbours of the rotations, random) against the big-int model evaluated on those rows only"""
    check_evaluate_h_sampled_rows(ctx, 19, synthetic.synthetic_cs(**kw))


def check_evaluate_h_sampled_rows(ctx, k, cs):
    import torch
    import halo2_model as hm
    d = zk.Domain(ctx, cs.degree(), k)
    en = d.extended_len
    gen = torch.Generator(device="cuda")
    gen.manual_seed(19)

    def col():
        t = torch.randint(-(1 << 63), (1 << 63) - 1, (en, 4), dtype=torch.int64, device="cuda", generator=gen)
        t[:, 3] &= 0x0FFFFFFFFFFFFFFF
        return t
    T = {"fixed": [col() for _ in range(cs.num_fixed_columns)], "advice": [col() for _ in range(cs.num_advice_columns)],
         "instance": [col() for _ in range(cs.num_instance_columns)], "l": [col() for _ in range(3)],
         "lookups": [tuple(col() for _ in range(3)) for _ in cs.lookups],
         "sigma": [col() for _ in cs.permutation_columns], "perm_z": [col() for _ in range(cs.num_permutation_sets())]}
    torch.cuda.synchronize()
    rng = np.random.default_rng(19)
    call = evalh_host.EvalhCall(cs)
    ptr = lambda t: t.data_ptr()
    call.fixed, call.advice, call.instance = [ptr(t) for t in T["fixed"]], [ptr(t) for t in T["advice"]], [ptr(t) for t in T["instance"]]
    call.challenges = rand_fr(rng, cs.num_challenges)
    call.y, call.beta, call.gamma, call.theta = (rand_fr(rng, 1)[0] for _ in range(4))
    call.l0, call.l_last, call.l_active = (ptr(t) for t in T["l"])
    call.lookups = [tuple(ptr(t) for t in tri) for tri in T["lookups"]]
    call.perm_sigma, call.perm_z = [ptr(t) for t in T["sigma"]], [ptr(t) for t in T["perm_z"]]
    out = torch.empty((en, 4), dtype=torch.int64, device="cuda")
    d.evaluate_h(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), out.data_ptr())
    torch.cuda.synchronize()

    D = _DeviceColumn
    one = lambda limbs: fr_ints(np.asarray(limbs)[None])[0]
    inp = hm.EvalInputs([D(t) for t in T["fixed"]], [D(t) for t in T["advice"]], [D(t) for t in T["instance"]],
                        fr_ints(call.challenges), one(call.beta), one(call.gamma), one(call.theta), one(call.y))
    ev = call.ev
    desc = {"gates": ev.custom_gates.to_model(), "l0": D(T["l"][0]), "l_last": D(T["l"][1]), "l_active": D(T["l"][2]),
            "lookups": [{"program": g.to_model(), "product": D(z), "permuted_input": D(a), "permuted_table": D(s)}
                        for g, (z, a, s) in zip(ev.lookups, T["lookups"])],
            "perm": {"columns": cs.permutation_columns, "sigma": [D(t) for t in T["sigma"]], "z": [D(t) for t in T["perm_z"]],
                     "chunk_len": cs.permutation_chunk_len(), "blinding_factors": cs.blinding_factors()}}
    md = hm.EvaluationDomain(cs.degree(), k)
    rows = [0, 1, 3, 4, 7, 8, 39, en - 1, en - 2, en - 5, en - 8, en - 24, en // 2, en // 2 + 1]
    rows += [int(r) for r in rng.integers(0, en, size=8)]
    got = _DeviceColumn(out)
    for r in rows:
        assert got[r] == hm.evaluate_h_row(md, inp, desc, r), r
    d.close()


def test_shplonk_full_size_identity(ctx, co):
    """k = 19: after create_proof, h(X) and the final quotient q(X) satisfy, at a random r,
         q(r) (r - u) z_0 = sum_i v^i z_i sum_j y^j (P_ij(r) - R_ij(u)) - Z_T(u) h(r)
    with every polynomial evaluated on the device and the few low-degree pieces recomputed here; h itself is pinned by
    the prover's own check that the right-hand side vanishes at u"""
    from zkevm_circuits_amd import shplonk
    k = 19
    n = 1 << k
    rng = np.random.default_rng(1919)
    md_omega = pow(bn.FR_ROOT_OF_UNITY, 1 << (bn.FR_S - k), P)
    x = int(rng.integers(2, 2 ** 62))
    shape = [(0,), (0, 1), (0, 1), (0, -1), (0, 1, -1)]
    d_polys = {key: ctx.upload(rand_fr(rng, n)) for key in range(len(shape))}
    pts = {r: x * pow(md_omega, r % n, P) % P for r in (-1, 0, 1)}
    flat = [(key, r) for r_i in range(3) for key, rots in enumerate(shape) if r_i < len(rots) for r in [rots[r_i]]]
    vals = ctx.eval_polynomials_dev([d_polys[key] for key, _ in flat], fr_arr([pts[r] for _, r in flat]), n)
    queries = [shplonk.Query(key, pts[r], v) for (key, r), v in zip(flat, fr_ints(vals))]
    y, v, u, r = (int(c) for c in rng.integers(2, 2 ** 62, size=4))
    bid = ctx.register_bases(ctx.g1_multiples(n))
    tr = _ReplayTranscript([y, v, u])
    d_h, d_q = shplonk.ProverSHPLONK(ctx, bid, n).create_proof(d_polys, queries, tr)
    at = lambda d_p: fr_ints(ctx.eval_polynomial_dev(d_p, n, fr_arr([r])[0])[None])[0]
    sets, sup = shplonk.construct_intermediate_sets(queries)
    rhs, z0 = 0, None
    for i, (points, cs_) in enumerate(sets):
        z_i = shplonk.evaluate_vanishing_polynomial([p_ for p_ in sup if p_ not in points], u)
        z0 = z_i if z0 is None else z0
        inner = 0
        for j, (key, evals) in enumerate(cs_):
            low = shplonk.lagrange_interpolate(points, evals)
            inner = (inner + pow(y, j, P) * (at(d_polys[key]) - shplonk.eval_small(low, u))) % P
        rhs = (rhs + pow(v, i, P) * z_i % P * inner) % P
    rhs = (rhs - shplonk.evaluate_vanishing_polynomial(sup, u) * at(d_h)) % P
    assert at(d_q) * (r - u) % P * z0 % P == rhs
    assert len(tr.points) == 2 and len(sets) == 4
    for p_ in list(d_polys.values()) + [d_h, d_q]:
        ctx.free(p_)
    ctx.release_bases(bid)
