"""CPU suite, part 1: the C restatement (oracle/oracle.c) against the golden vectors of the Python big-int model,
and both against the known answers the reference's own tests hold (Poseidon KATs, Fr byte conventions)."""
import numpy as np
import pytest

import bn254 as bn
import halo2_model as hm
from util import (affine_arr, affine_pt, fq_arr, fq_ints, fr_arr, fr_ints, golden, hex_fr, hex_pt)

P = bn.FR_MODULUS


def test_constants_rederived():
    bn.self_check()
    g = golden("field.json")
    assert int(g["fields"]["fr"]["modulus"], 16) == bn.FR_MODULUS
    assert int(g["fields"]["fq"]["modulus"], 16) == bn.FQ_MODULUS
    assert int(g["fr_constants"]["root_of_unity"], 16) == bn.FR_ROOT_OF_UNITY
    assert int(g["fr_constants"]["zeta"], 16) == bn.FR_ZETA
    assert int(g["fr_constants"]["delta"], 16) == bn.FR_DELTA


@pytest.mark.parametrize("fname,fid", [("fr", 0), ("fq", 1)])
def test_field_ops(co, fname, fid):
    g = golden("field.json")["fields"][fname]
    F = bn.Fr if fid == 0 else bn.Fq
    to_arr = fr_arr if fid == 0 else fq_arr
    to_int = fr_ints if fid == 0 else fq_ints
    a = to_arr([int(c["a"], 16) for c in g["cases"]])
    b = to_arr([int(c["b"], 16) for c in g["cases"]])
    for op, key in ((0, "add"), (1, "sub"), (2, "mul")):
        got = to_int(co.field_binop(fid, op, a, b))
        assert got == [int(c[key], 16) for c in g["cases"]], key
    assert to_int(co.field_binop(fid, 2, a, a)) == [int(c["sqr"], 16) for c in g["cases"]]
    assert to_int(co.field_invert(fid, a)) == [int(c["inv"], 16) for c in g["cases"]]
    # Montgomery <-> canonical round trip (`to_repr` little-endian)
    canon = co.to_canonical(fid, a)
    assert [sum(int(x) << (64 * i) for i, x in enumerate(r)) for r in canon] == [int(c["a"], 16) for c in g["cases"]]
    assert np.array_equal(co.from_canonical(fid, canon), a)
    assert int(g["R"], 16) == F.R


def _poseidon_c(co, state, rounds, mds):
    """Poseidon permutation driven through the C oracle's Fr add/mul (pins its arithmetic on the reference KAT)."""
    def mul(x, y):
        return co.field_binop(0, 2, x, y)

    def add(x, y):
        return co.field_binop(0, 0, x, y)

    st = fr_arr(state)
    mds_a = [[fr_arr([c]) for c in row] for row in mds]
    for rnd, rcs in enumerate(rounds):
        st = add(st, fr_arr(rcs))
        full = rnd < 4 or rnd >= 4 + 57
        n_sbox = 3 if full else 1
        x = st[:n_sbox]
        x2 = mul(x, x)
        st = np.concatenate([mul(mul(x2, x2), x), st[n_sbox:]])
        new = []
        for i in range(3):
            acc = mul(mds_a[i][0], st[0:1])
            acc = add(acc, mul(mds_a[i][1], st[1:2]))
            acc = add(acc, mul(mds_a[i][2], st[2:3]))
            new.append(acc[0])
        st = np.array(new, dtype=np.uint64)
    return fr_ints(st)


def test_poseidon_kats_from_reference(co):
    k = golden("reference_kats.json")
    rounds = [[int(c, 16) for c in r] for r in k["poseidon_params"]["round_constants"]]
    mds = [[int(c, 16) for c in r] for r in k["poseidon_params"]["mds"]]
    assert (rounds, mds) == bn.poseidon_params()          # fixture == regenerated constants
    for kat in k["poseidon_hash_with_domain"]:
        inp = [int(x, 16) for x in kat["inputs"]]
        dom = int(kat["domain"], 16)
        exp = int(kat["expected"], 16)
        assert bn.poseidon_hash_with_domain(inp, dom) == exp
        assert _poseidon_c(co, [dom, inp[0], inp[1]], rounds, mds)[0] == exp
    for kat in k["poseidon_hash_msg"]:
        assert bn.poseidon_hash_msg([int(x, 16) for x in kat["msg"]], int(kat["cap"], 16)) == int(kat["expected"], 16)
    # all three lanes of one permutation (/root/reference/poseidon/src/poseidon/septidon/tests.rs:58-61)
    for kat in k["poseidon_permutation_full_state"]:
        st, exp = [int(x, 16) for x in kat["state"]], [int(x, 16) for x in kat["expected"]]
        assert bn.poseidon_permute(st) == exp
        assert _poseidon_c(co, st, rounds, mds) == exp


def test_fr_byte_conventions(co):
    k = golden("reference_kats.json")["fr_bytes"]
    one = bn.Fr.from_uniform_bytes(bytes.fromhex(k["from_uniform_bytes_one_le64"]))
    assert one == 1
    assert fr_ints(fr_arr([one])) == [1]
    mx = int.from_bytes(bytes.fromhex(k["max_value_le32"]), "little")
    canon = co.to_canonical(0, fr_arr([mx]))[0]
    assert b"".join(int(x).to_bytes(8, "little") for x in canon) == bytes.fromhex(k["max_value_le32"])


def test_curve_add(co):
    g = golden("curve.json")
    p = affine_arr([hex_pt(c["p"]) for c in g["add"]])
    q = affine_arr([hex_pt(c["q"]) for c in g["add"]])
    got = co.g1_add_affine(p, q)
    assert [affine_pt(r) for r in got] == [hex_pt(c["sum"]) for c in g["add"]]
    bases = co.g1_bases(1000)
    for kk, v in g["multiples_of_g"].items():
        if int(kk) <= 1000:
            assert affine_pt(bases[int(kk) - 1]) == hex_pt(v)


def test_multiexp_golden(co):
    g = golden("msm.json")
    for case in g["cases"]:
        n = case["n"]
        bases = co.g1_bases(n)
        sc = hex_fr(case["scalars"])
        for threads in (1, 3):
            jac = co.best_multiexp(sc, bases, threads=threads)
            assert affine_pt(co.g1_to_affine(jac)) == hex_pt(case["result"]), (n, case["dist"], threads)


def test_best_fft_golden(co):
    for case in golden("ntt.json")["cases"]:
        out = co.best_fft(hex_fr(case["input"]), fr_arr([int(case["omega"], 16)])[0], case["log_n"], threads=2)
        assert np.array_equal(out, hex_fr(case["output"])), case["log_n"]


def test_domain_golden(co):
    for case in golden("domain.json")["cases"]:
        d = co.Domain(case["j"], case["k"], threads=2)
        assert d.extended_k == case["extended_k"]
        assert fr_ints(d.constant(0)) == [int(case["omega"], 16)]
        assert fr_ints(d.constant(2)) == [int(case["extended_omega"], 16)]
        assert fr_ints(d.constant(6)) == [int(case["ifft_divisor"], 16)]
        assert fr_ints(d.constant(7)) == [int(case["extended_ifft_divisor"], 16)]
        coeff = hex_fr(case["coeff"])
        assert np.array_equal(d.coeff_to_lagrange(coeff), hex_fr(case["lagrange"]))
        assert np.array_equal(d.lagrange_to_coeff(hex_fr(case["lagrange"])), coeff)
        assert np.array_equal(d.coeff_to_extended(coeff), hex_fr(case["extended"]))
        assert np.array_equal(d.extended_to_coeff(hex_fr(case["big_extended"])), hex_fr(case["big_coeff"]))
        assert np.array_equal(d.divide_by_vanishing_poly(hex_fr(case["big_extended"])), hex_fr(case["big_divided"]))


def test_evaluate_h_golden(co):
    from evalh_cases import build_call
    import zkevm_circuits_amd.ffi as ffi
    for case in golden("evalh.json")["cases"]:
        cs, call = build_call(case)
        d = co.Domain(cs.degree(), case["k"], threads=2)
        assert d.extended_k == case["extended_k"]
        assert cs.blinding_factors() == case["blinding_factors"]
        out = d.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED))
        assert np.array_equal(out, hex_fr(case["h"])), case["cs"]


def test_grand_products_golden():
    """the Python model's batch-inverse + scan formulation against the per-row definitions stored in products.json"""
    g = golden("products.json")
    ints = lambda hs: [int(h, 16) for h in hs]
    for c in g["permutation"]:
        dom = hm.EvaluationDomain(3, c["k"])
        z = hm.permutation_product(dom.omega, [ints(v) for v in c["values"]], [ints(v) for v in c["sigmas"]],
                                   int(c["beta"], 16), int(c["gamma"], 16), pow(bn.FR_DELTA, c["first_column"], bn.FR_MODULUS),
                                   int(c["z0"], 16))
        assert z == ints(c["z"])
    for c in g["lookup"]:
        z = hm.lookup_product(ints(c["input"]), ints(c["table"]), ints(c["permuted_input"]), ints(c["permuted_table"]),
                              int(c["beta"], 16), int(c["gamma"], 16))
        assert z == ints(c["z"])
    for c in g["batch_invert"]:
        assert hm.batch_invert(ints(c["input"])) == ints(c["output"])
    for c in g["eval_polynomial"]:
        assert hm.eval_poly(ints(c["coeff"]), int(c["x"], 16)) == int(c["value"], 16)


def test_shplonk_golden():
    """the model's Kate-division / interpolation formulation of the SHPLONK prover against the long-division derivation
    stored in shplonk.json"""
    g = golden("shplonk.json")
    ints = lambda hs: [int(h, 16) for h in hs]
    Pm = bn.FR_MODULUS
    commit = lambda cs: bn.g1_mul(bn.G1_GENERATOR, sum(c * (i + 1) for i, c in enumerate(cs)) % Pm)
    for c in g["cases"]:
        polys = {i: ints(p) for i, p in enumerate(c["polys"])}
        queries = [(key, int(pt, 16), int(ev, 16)) for key, pt, ev in c["queries"]]
        h1, h2, h_x, fin = hm.shplonk_create_proof(polys, queries, int(c["y"], 16), int(c["v"], 16), int(c["u"], 16), commit)
        assert h_x == ints(c["h_x"]) and fin == ints(c["final"])
        assert [h1, h2] == [(int(p[0], 16), int(p[1], 16)) for p in (c["h1"], c["h2"])]
    for c in g["kate_division"]:
        assert hm.kate_division(ints(c["a"]), int(c["point"], 16)) == ints(c["q"])


def test_lookup_permute_golden():
    g = golden("lookup_permute.json")
    ints = lambda hs: [int(h, 16) for h in hs]
    for c in g["cases"]:
        a, s_ = hm.permute_expression_pair(ints(c["input"]), ints(c["table"]), c["usable_rows"])
        assert a == ints(c["permuted_input"]) and s_ == ints(c["permuted_table"])


def test_model_multiexp_threads_agree():
    rnd = np.random.default_rng(5)
    n = 40
    bases = bn.g1_bases(n)
    sc = [int(x) for x in rnd.integers(0, 1 << 62, size=n)]
    assert hm.best_multiexp(sc, bases, threads=1) == hm.best_multiexp(sc, bases, threads=7) == bn.g1_msm_naive(sc, bases)
