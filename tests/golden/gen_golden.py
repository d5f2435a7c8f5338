#!/usr/bin/env python3
"""Generates tests/golden/*.json from the Python big-int model (oracle/bn254.py, oracle/halo2_model.py).

The vectors are independent of the C restatement and of the HIP code: NTTs are checked against an O(n^2)
DFT, coset extension against direct polynomial evaluation, evaluate_h against direct evaluation of the
expression trees (no GraphEvaluator), MSM against double-and-add.  Known answers that come from the
reference's own tests are cited next to them.  Run from the repository root:  python tests/golden/gen_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bn254 as bn  # noqa: E402
import halo2_model as hm  # noqa: E402
import zkhip_pkg  # noqa: E402

zk = zkhip_pkg.load()
from zkevm_circuits_amd import plonk, synthetic  # noqa: E402

P = bn.FR_MODULUS
SEED = 20250523


def hx(v):
    return format(v, "x")


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(name, os.path.getsize(path), "bytes")


# --------------------------------------------------------------------------------------------- field
def gen_field():
    rnd = random.Random(SEED)
    out = {"comment": "canonical integers (hex); a op b mod p; inv(0)=0", "fields": {}}
    for name, F in (("fr", bn.Fr), ("fq", bn.Fq)):
        p = F.p
        edge = [0, 1, 2, p - 1, p - 2, F.R, F.R2, (p + 1) // 2, (1 << 253), (1 << 29) - 1, 1 << 29, (1 << 232) + 5]
        vals = edge + [rnd.randrange(p) for _ in range(40)]
        cases = []
        for i, a in enumerate(vals):
            b = vals[(i * 7 + 3) % len(vals)]
            cases.append({"a": hx(a), "b": hx(b), "add": hx((a + b) % p), "sub": hx((a - b) % p),
                          "mul": hx(a * b % p), "sqr": hx(a * a % p), "inv": hx(F.inv(a))})
        out["fields"][name] = {"modulus": hx(p), "R": hx(F.R), "R2": hx(F.R2), "inv64": hx(F.inv64), "cases": cases}
    out["fr_constants"] = {"S": bn.FR_S, "root_of_unity": hx(bn.FR_ROOT_OF_UNITY), "zeta": hx(bn.FR_ZETA),
                           "delta": hx(bn.FR_DELTA)}
    dump("field.json", out)


# --------------------------------------------------------------------------------------------- poseidon / bytes
def gen_reference_kats():
    out = {
        "comment": "known answers held by the reference's own tests",
        "poseidon_hash_with_domain": [
            {"source": "/root/reference/poseidon/src/hash.rs:903-918", "inputs": ["1", "2"], "domain": "0",
             "expected": "115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a"},
            {"source": "/root/reference/poseidon/src/hash.rs:903-918", "inputs": ["1", "2"], "domain": "100",
             "expected": "05390df727dcce2ddb8faa3acb4798ad4e95b74de05e5cc7e40496658913ae85"},
        ],
        "poseidon_hash_msg": [
            {"source": "/root/reference/poseidon/src/hash.rs:921-950", "msg": ["1", "2", "3000000"], "cap": "2d",
             "expected": "212b546f9c67c4fdcba131035644aa1d8baa8943f84b0a27e8f65b5bd532213e"},
            {"source": "/root/reference/poseidon/src/hash.rs:921-950", "msg": ["1", "2", "3", "0"], "cap": "32",
             "expected": "066397f309d55f6caf6419cbb4120f5ada8e54254061b4b448359de388ab5526"},
        ],
        # the whole state after one permutation of (0, 1, 2): all three lanes, where the hash KATs above pin lane 0 only
        "poseidon_permutation_full_state": [
            {"source": "/root/reference/poseidon/src/poseidon/septidon/tests.rs:58-61", "state": ["0", "1", "2"],
             "expected": ["115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a",
                          "0fca49b798923ab0239de1c9e7a4a9a2210312b6a2f616d18b5a87f9b628ae29",
                          "0e7ae82e40091e63cbd4f16a6d16310b3729d4b6e138fcf54110e2867045a30c"]},
        ],
        "fr_bytes": {
            "source": "/root/reference/zkevm-circuits/src/util.rs:832-882",
            "from_uniform_bytes_one_le64": "01" + "00" * 63,
            "expects_one": True,
            "max_value_le32": "ff" * 31 + "2f",
            "max_value_roundtrips": True,
        },
    }
    # regenerate the permutation trace of the first KAT so the device can be checked round by round
    rounds, mds = bn.poseidon_params()
    out["poseidon_params"] = {"t": 3, "r_f": 8, "r_p": 57, "round_constants": [[hx(c) for c in r] for r in rounds],
                              "mds": [[hx(c) for c in r] for r in mds],
                              "derivation": "Grain LFSR of the Poseidon reference parameter script (field=1, sbox=0, n=254)"}
    for kat in out["poseidon_hash_with_domain"]:
        got = bn.poseidon_hash_with_domain([int(x, 16) for x in kat["inputs"]], int(kat["domain"], 16))
        assert got == int(kat["expected"], 16), "poseidon KAT mismatch"
    for kat in out["poseidon_hash_msg"]:
        got = bn.poseidon_hash_msg([int(x, 16) for x in kat["msg"]], int(kat["cap"], 16))
        assert got == int(kat["expected"], 16), "poseidon msg KAT mismatch"
    for kat in out["poseidon_permutation_full_state"]:
        got = bn.poseidon_permute([int(x, 16) for x in kat["state"]])
        assert got == [int(x, 16) for x in kat["expected"]], "poseidon full-state KAT mismatch"
    assert bn.Fr.from_uniform_bytes(bytes.fromhex(out["fr_bytes"]["from_uniform_bytes_one_le64"])) == 1
    mx = int.from_bytes(bytes.fromhex(out["fr_bytes"]["max_value_le32"]), "little")
    assert mx < P and bn.Fr.to_repr(mx) == bytes.fromhex(out["fr_bytes"]["max_value_le32"])
    dump("reference_kats.json", out)


# --------------------------------------------------------------------------------------------- curve / msm
def pt(P_):
    return None if P_ is None else [hx(P_[0]), hx(P_[1])]


def gen_curve():
    G = bn.G1_GENERATOR
    g2, g3, g5 = bn.g1_mul(G, 2), bn.g1_mul(G, 3), bn.g1_mul(G, 5)
    cases = [(G, g2), (g2, G), (G, G), (G, bn.g1_neg(G)), (None, g3), (g3, None), (None, None), (g3, g5),
             (g5, g5), (bn.g1_mul(G, P - 1), G), (bn.g1_mul(G, 123456789), bn.g1_mul(G, P - 123456789))]
    out = {"comment": "affine (x, y) canonical hex; null = identity", "add": [
        {"p": pt(a), "q": pt(b), "sum": pt(bn.g1_add(a, b))} for a, b in cases]}
    out["multiples_of_g"] = {str(k): pt(bn.g1_mul(G, k)) for k in (1, 2, 3, 4, 5, 16, 31, 32, 1000, 65537)}
    dump("curve.json", out)


def gen_msm():
    rnd = random.Random(SEED)
    out = {"comment": "bases[i] = (i+1)G; scalars canonical hex; result affine canonical hex (null = identity). "
                      "Result checked two ways: double-and-add and the multiexp_serial model.", "cases": []}
    for n in (1, 2, 31, 32, 256):
        bases = bn.g1_bases(n)
        dists = {
            "zeros": [0] * n,
            "ones": [1] * n,
            "bits": [rnd.randrange(2) for _ in range(n)],
            "bytes": [rnd.randrange(256) for _ in range(n)],
            "uniform": [rnd.randrange(P) for _ in range(n)],
            "top": [P - 1 - rnd.randrange(3) for _ in range(n)],
            "witness": [rnd.choice([0, 0, 0, 1, 1, rnd.randrange(256), rnd.randrange(1 << 16), rnd.randrange(P)])
                        for _ in range(n)],
        }
        for name, sc in dists.items():
            # sum_i s_i (i+1) G  ==  (sum_i s_i (i+1) mod r) G
            k = sum(s * (i + 1) for i, s in enumerate(sc)) % P
            direct = bn.g1_mul(bn.G1_GENERATOR, k)
            if n <= 32:
                assert bn.g1_msm_naive(sc, bases) == direct
                assert hm.best_multiexp(sc, bases, threads=3) == direct
            elif name in ("uniform", "witness"):
                assert hm.multiexp_serial(sc, bases) == direct
            out["cases"].append({"n": n, "dist": name, "scalars": [hx(s) for s in sc], "result": pt(direct)})
    dump("msm.json", out)


# --------------------------------------------------------------------------------------------- ntt / domain
def gen_ntt():
    rnd = random.Random(SEED)
    out = {"comment": "best_fft(a, omega, log_n); omega = ROOT_OF_UNITY^(2^(28-log_n)); canonical hex", "cases": []}
    for log_n in range(1, 11):
        n = 1 << log_n
        omega = pow(bn.FR_ROOT_OF_UNITY, 1 << (bn.FR_S - log_n), P)
        a = [rnd.randrange(P) for _ in range(n)]
        if log_n == 3:
            a = [0, 1, P - 1, 2, 0, 0, P - 2, 5]
        res = hm.best_fft(list(a), omega, log_n)
        if log_n <= 7:
            assert res == hm.dft_naive(a, omega)
        else:   # spot-check 8 outputs against the definition
            for i in rnd.sample(range(n), 8):
                assert res[i] == sum(a[j] * pow(omega, i * j, P) for j in range(n)) % P
        inv = hm.best_fft(list(res), pow(omega, -1, P), log_n)
        assert [v * pow(n, -1, P) % P for v in inv] == a
        out["cases"].append({"log_n": log_n, "omega": hx(omega), "input": [hx(v) for v in a], "output": [hx(v) for v in res]})
    dump("ntt.json", out)


def gen_domain():
    rnd = random.Random(SEED)
    out = {"comment": "EvaluationDomain::new(j, k); coeff_to_extended[i] = p(zeta * extended_omega^i)", "cases": []}
    for (j, k) in ((2, 4), (3, 4), (4, 5), (5, 4), (5, 6), (9, 4), (6, 5)):
        dom = hm.EvaluationDomain(j, k)
        n = dom.n
        coeff = [rnd.randrange(P) for _ in range(n)]
        ext = dom.coeff_to_extended(coeff)
        for i in range(dom.extended_len()):
            x = dom.g_coset * pow(dom.extended_omega, i, P) % P
            assert ext[i] == hm.eval_poly(coeff, x)
        lag = dom.coeff_to_lagrange(coeff)
        assert all(lag[i] == hm.eval_poly(coeff, pow(dom.omega, i, P)) for i in range(n))
        assert dom.lagrange_to_coeff(lag) == coeff
        # a polynomial of degree < n*(j-1) survives extended -> coeff
        big = [rnd.randrange(P) for _ in range(n * dom.quotient_poly_degree)]
        big_ext = [hm.eval_poly(big, dom.g_coset * pow(dom.extended_omega, i, P) % P) for i in range(dom.extended_len())]
        assert dom.extended_to_coeff(big_ext) == big
        div = dom.divide_by_vanishing_poly(big_ext)
        for i in (0, 1, dom.extended_len() - 1):
            x = dom.g_coset * pow(dom.extended_omega, i, P) % P
            assert div[i] == big_ext[i] * pow((pow(x, n, P) - 1) % P, -1, P) % P
        out["cases"].append({
            "j": j, "k": k, "extended_k": dom.extended_k,
            "omega": hx(dom.omega), "extended_omega": hx(dom.extended_omega),
            "ifft_divisor": hx(dom.ifft_divisor), "extended_ifft_divisor": hx(dom.extended_ifft_divisor),
            "t_evaluations": [hx(v) for v in dom.t_evaluations],
            "coeff": [hx(v) for v in coeff], "lagrange": [hx(v) for v in lag], "extended": [hx(v) for v in ext],
            "big_coeff": [hx(v) for v in big], "big_extended": [hx(v) for v in big_ext],
            "big_divided": [hx(v) for v in div],
        })
    dump("domain.json", out)


# --------------------------------------------------------------------------------------------- evaluate_h
def eval_expr(e, cols, challenges, idx, rot_scale, size):
    t = e[0]
    if t == "const":
        return e[1]
    if t in ("fixed", "advice", "instance"):
        return cols[t][e[1]][(idx + e[2] * rot_scale) % size]
    if t == "challenge":
        return challenges[e[1]]
    if t == "neg":
        return (-eval_expr(e[1], cols, challenges, idx, rot_scale, size)) % P
    if t == "sum":
        return (eval_expr(e[1], cols, challenges, idx, rot_scale, size)
                + eval_expr(e[2], cols, challenges, idx, rot_scale, size)) % P
    if t == "prod":
        return (eval_expr(e[1], cols, challenges, idx, rot_scale, size)
                * eval_expr(e[2], cols, challenges, idx, rot_scale, size)) % P
    if t == "scaled":
        return eval_expr(e[1], cols, challenges, idx, rot_scale, size) * e[2] % P
    raise ValueError(t)


def direct_evaluate_h(cs, dom, data):
    """evaluate_h from the definitions: expression trees, no GraphEvaluator."""
    size, rs = dom.extended_len(), 1 << (dom.extended_k - dom.k)
    cols = {"fixed": data["fixed"], "advice": data["advice"], "instance": data["instance"]}
    ch, y, beta, gamma, theta = data["challenges"], data["y"], data["beta"], data["gamma"], data["theta"]
    l0, ll, la = data["l0"], data["l_last"], data["l_active"]
    out = []
    kind_name = {plonk.ANY_ADVICE: "advice", plonk.ANY_FIXED: "fixed", plonk.ANY_INSTANCE: "instance"}
    for idx in range(size):
        v = 0
        for gate in cs.gates:
            for poly in gate:
                v = (v * y + eval_expr(poly, cols, ch, idx, rs, size)) % P
        z = data["perm_z"]
        if z:
            bf = cs.blinding_factors()
            r_next, r_last = (idx + rs) % size, (idx - (bf + 1) * rs) % size
            v = (v * y + (1 - z[0][idx]) * l0[idx]) % P
            v = (v * y + (z[-1][idx] ** 2 - z[-1][idx]) * ll[idx]) % P
            for s in range(1, len(z)):
                v = (v * y + (z[s][idx] - z[s - 1][r_last]) * l0[idx]) % P
            x = bn.FR_ZETA * pow(dom.extended_omega, idx, P) % P     # the row's point zeta * w^idx
            chunk = cs.permutation_chunk_len()
            for s in range(len(z)):
                left, right = z[s][r_next], z[s][idx]
                for c in range(s * chunk, min((s + 1) * chunk, len(cs.permutation_columns))):
                    kind, ci = cs.permutation_columns[c]
                    val = cols[kind_name[kind]][ci][idx]
                    left = left * (val + beta * data["perm_sigma"][c][idx] + gamma) % P
                    right = right * (val + beta * x * pow(bn.FR_DELTA, c, P) + gamma) % P
                v = (v * y + (left - right) * la[idx]) % P
        for lk, (zc, ac, sc) in zip(cs.lookups, data["lookups"]):
            cin = 0
            for e in lk.input_expressions:
                cin = (cin * theta + eval_expr(e, cols, ch, idx, rs, size)) % P
            ctb = 0
            for e in lk.table_expressions:
                ctb = (ctb * theta + eval_expr(e, cols, ch, idx, rs, size)) % P
            r_next, r_prev = (idx + rs) % size, (idx - rs) % size
            v = (v * y + (1 - zc[idx]) * l0[idx]) % P
            v = (v * y + (zc[idx] ** 2 - zc[idx]) * ll[idx]) % P
            v = (v * y + (zc[r_next] * (ac[idx] + beta) * (sc[idx] + gamma) - zc[idx] * (cin + beta) * (ctb + gamma))
                 * la[idx]) % P
            v = (v * y + (ac[idx] - sc[idx]) * l0[idx]) % P
            v = (v * y + (ac[idx] - sc[idx]) * (ac[idx] - ac[r_prev]) * la[idx]) % P
        out.append(v)
    return out


def random_evalh_data(cs, dom, rnd):
    size = dom.extended_len()

    def col():
        return [rnd.randrange(P) for _ in range(size)]

    return {
        "fixed": [col() for _ in range(cs.num_fixed_columns)], "advice": [col() for _ in range(cs.num_advice_columns)],
        "instance": [col() for _ in range(cs.num_instance_columns)],
        "challenges": [rnd.randrange(P) for _ in range(cs.num_challenges)],
        "y": rnd.randrange(P), "beta": rnd.randrange(P), "gamma": rnd.randrange(P), "theta": rnd.randrange(P),
        "l0": col(), "l_last": col(), "l_active": col(),
        "lookups": [(col(), col(), col()) for _ in cs.lookups],
        "perm_sigma": [col() for _ in cs.permutation_columns],
        "perm_z": [col() for _ in range(cs.num_permutation_sets())],
    }


def model_evaluate_h(cs, dom, data):
    """Same data through the GraphEvaluator path (plonk.Evaluator lowering + halo2_model interpreter)."""
    ev = plonk.Evaluator(cs)
    inp = hm.EvalInputs(data["fixed"], data["advice"], data["instance"], data["challenges"], data["beta"],
                        data["gamma"], data["theta"], data["y"])
    desc = {"gates": ev.custom_gates.to_model(), "l0": data["l0"], "l_last": data["l_last"], "l_active": data["l_active"],
            "lookups": [{"program": g.to_model(), "product": z, "permuted_input": a, "permuted_table": s}
                        for g, (z, a, s) in zip(ev.lookups, data["lookups"])],
            "perm": {"columns": cs.permutation_columns, "sigma": data["perm_sigma"], "z": data["perm_z"],
                     "chunk_len": cs.permutation_chunk_len(), "blinding_factors": cs.blinding_factors()}}
    return hm.evaluate_h(dom, inp, desc)


def toy_cs():
    """One gate with rotations +-1, one 2-column lookup, a 2-column permutation (the minimal shape of every
    argument evaluate_h handles)."""
    A, F = plonk.Advice, plonk.Fixed
    cs = plonk.ConstraintSystem(num_fixed_columns=2, num_advice_columns=3, num_instance_columns=1, num_challenges=1)
    cs.gates.append([plonk.Product(F(0, 0), plonk.Sub(plonk.Product(A(0, -1), A(1, 1)), A(2, 0))),
                     plonk.Product(F(1, 0), plonk.Sum(plonk.Product(A(0, 0), A(0, 0)),
                                                      plonk.Sub(plonk.Scaled(A(1, 0), 5), plonk.Challenge(0))))])
    cs.lookups.append(plonk.Lookup([A(0, 0), plonk.Product(F(0, 0), A(1, 1))], [F(1, 0), A(2, 0)]))
    cs.permutation_columns = [(plonk.ANY_ADVICE, 0), (plonk.ANY_INSTANCE, 0)]
    return cs


def gen_evalh():
    rnd = random.Random(SEED)
    out = {"comment": "evaluate_h on extended cosets; canonical hex; `cs` names the constructor in "
                      "tests/golden/gen_golden.py (toy) or synthetic.synthetic_cs kwargs", "cases": []}
    specs = [("toy", None, 4), ("toy", None, 6),
             ("synthetic", dict(num_advice=4, num_fixed=2, num_gate_polys=6, num_lookups=1, num_perm_cols=5,
                                rotations=[-2, -1, 0, 1], seed=1), 6),
             ("synthetic", dict(num_advice=6, num_fixed=3, num_gate_polys=10, max_degree=4, num_lookups=2,
                                lookup_width=2, num_perm_cols=3, rotations=[-5, 0, 1], seed=2), 6)]
    for name, kw, k in specs:
        cs = toy_cs() if name == "toy" else synthetic.synthetic_cs(**kw)
        dom = hm.EvaluationDomain(cs.degree(), k)
        data = random_evalh_data(cs, dom, rnd)
        direct = direct_evaluate_h(cs, dom, data)
        via_graph = model_evaluate_h(cs, dom, data)
        assert direct == via_graph, "GraphEvaluator path disagrees with the direct formula"
        enc = {key: ([[hx(v) for v in c] for c in val] if key in ("fixed", "advice", "instance", "perm_sigma", "perm_z")
                     else [[[hx(v) for v in c] for c in t] for t in val] if key == "lookups"
                     else [hx(v) for v in val] if isinstance(val, list) else hx(val))
               for key, val in data.items()}
        out["cases"].append({"cs": name, "kwargs": kw, "k": k, "degree": cs.degree(), "extended_k": dom.extended_k,
                             "blinding_factors": cs.blinding_factors(), "data": enc, "h": [hx(v) for v in direct]})
    dump("evalh.json", out)


# --------------------------------------------------------------------------------------------- grand products
def gen_products():
    """Permutation / lookup grand products, batch inversion and polynomial evaluation from their definitions: every
    fraction is formed with its own modular inverse (pow(d, P - 2, P)), no batch trick, no scan."""
    rnd = random.Random(SEED + 7)
    out = {"comment": "grand products before blinding, canonical hex; z has n rows, z[0] = z0", "permutation": [],
           "lookup": [], "batch_invert": [], "eval_polynomial": []}
    inv = lambda v: pow(v, P - 2, P)
    for k, cols, j0 in [(3, 1, 0), (4, 3, 2), (5, 4, 5)]:
        n = 1 << k
        dom = hm.EvaluationDomain(3, k)
        values = [[rnd.randrange(P) for _ in range(n)] for _ in range(cols)]
        sigmas = [[rnd.randrange(P) for _ in range(n)] for _ in range(cols)]
        beta, gamma, z0 = rnd.randrange(P), rnd.randrange(P), rnd.randrange(1, P)
        z = [z0]
        for i in range(n - 1):
            num = den = 1
            for j in range(cols):
                num = num * (values[j][i] + pow(bn.FR_DELTA, j0 + j, P) * beta % P * pow(dom.omega, i, P) + gamma) % P
                den = den * (values[j][i] + beta * sigmas[j][i] + gamma) % P
            z.append(z[-1] * num % P * inv(den) % P)
        assert hm.permutation_product(dom.omega, values, sigmas, beta, gamma, pow(bn.FR_DELTA, j0, P), z0) == z
        out["permutation"].append({"k": k, "first_column": j0, "values": [[hx(v) for v in c] for c in values],
                                   "sigmas": [[hx(v) for v in c] for c in sigmas], "beta": hx(beta), "gamma": hx(gamma),
                                   "z0": hx(z0), "z": [hx(v) for v in z]})
    for k in (3, 5):
        n = 1 << k
        a, s, ap, sp = ([rnd.randrange(P) for _ in range(n)] for _ in range(4))
        beta, gamma = rnd.randrange(P), rnd.randrange(P)
        z = [1]
        for i in range(n - 1):
            z.append(z[-1] * (a[i] + beta) % P * (s[i] + gamma) % P * inv((ap[i] + beta) * (sp[i] + gamma) % P) % P)
        assert hm.lookup_product(a, s, ap, sp, beta, gamma) == z
        out["lookup"].append({"k": k, "input": [hx(v) for v in a], "table": [hx(v) for v in s],
                              "permuted_input": [hx(v) for v in ap], "permuted_table": [hx(v) for v in sp],
                              "beta": hx(beta), "gamma": hx(gamma), "z": [hx(v) for v in z]})
    vals = [0, 1, P - 1, 2, 0, 0] + [rnd.randrange(P) for _ in range(27)]
    expect = [inv(v) if v else 0 for v in vals]
    assert hm.batch_invert(vals) == expect
    out["batch_invert"].append({"input": [hx(v) for v in vals], "output": [hx(v) for v in expect]})
    for n in (1, 5, 40):
        c = [rnd.randrange(P) for _ in range(n)]
        x = rnd.randrange(P)
        direct = sum(ci * pow(x, i, P) for i, ci in enumerate(c)) % P
        assert hm.eval_poly(c, x) == direct
        out["eval_polynomial"].append({"coeff": [hx(v) for v in c], "x": hx(x), "value": hx(direct)})
    dump("products.json", out)


def poly_divmod(a, b):
    """schoolbook long division of coefficient lists (lowest first): a = q b + r"""
    a = list(a)
    q = [0] * max(1, len(a) - len(b) + 1)
    lead_inv = pow(b[-1], P - 2, P)
    for i in range(len(a) - len(b), -1, -1):
        f = a[i + len(b) - 1] * lead_inv % P
        q[i] = f
        for j, bj in enumerate(b):
            a[i + j] = (a[i + j] - f * bj) % P
    return q, a[:len(b) - 1]


def gen_shplonk():
    """SHPLONK opening from its definition: h(X) = sum_i v^i (sum_j y^j (P_ij - R_ij)) / Z_i by schoolbook long
    division with the expanded vanishing polynomial (remainder checked to be zero), R_ij by solving the interpolation
    conditions, the final quotient L(X) / (X - u) / z_0 again by long division.  Commitments are taken against the bases
    (i + 1) G, where sum_i c_i (i + 1) G needs one scalar multiplication."""
    rnd = random.Random(SEED + 11)
    out = {"comment": "canonical hex; queries = [poly index, point, eval]; bases are (i+1)G", "cases": [],
           "kate_division": [], "lincomb": []}
    for k, shape in [(3, [(0,), (0, 1)]), (5, [(0, 1, -1), (0,), (0, 1), (0,), (0, 1, -1), (-2,)])]:
        n = 1 << k
        dom = hm.EvaluationDomain(3, k)
        x = rnd.randrange(1, P)
        polys = [[rnd.randrange(P) for _ in range(n)] for _ in shape]
        queries = []
        for r_idx in range(3):
            for key, rots in enumerate(shape):
                if r_idx < len(rots):
                    pnt = dom.rotate_omega(x, rots[r_idx])
                    queries.append((key, pnt, sum(c * pow(pnt, i, P) for i, c in enumerate(polys[key])) % P))
        y, v, u = rnd.randrange(P), rnd.randrange(P), rnd.randrange(P)
        # group by point set, in order of first appearance of each polynomial, then of each set
        point_sets = {}
        for key, pnt, _ in queries:
            point_sets.setdefault(key, set()).add(pnt)
        groups = []
        for key in sorted(point_sets, key=lambda kk: [q[0] for q in queries].index(kk)):
            fs = frozenset(point_sets[key])
            for gset, keys in groups:
                if gset == fs:
                    keys.append(key)
                    break
            else:
                groups.append((fs, [key]))
        all_points = sorted({pnt for _, pnt, _ in queries})

        def expand(roots):
            z = [1]
            for r in roots:
                z = [((z[i - 1] if i else 0) - r * (z[i] if i < len(z) else 0)) % P for i in range(len(z) + 1)]
            return z

        def add_scaled(acc, poly, c):
            return [((acc[i] if i < len(acc) else 0) + c * (poly[i] if i < len(poly) else 0)) % P
                    for i in range(max(len(acc), len(poly)))]

        h_x, l_x, z_diffs = [0] * n, [0] * n, []
        for i, (fs, keys) in enumerate(groups):
            pts = sorted(fs)
            z_i_poly = expand(pts)
            z_diff = 1
            for p_ in all_points:
                if p_ not in fs:
                    z_diff = z_diff * (u - p_) % P
            z_diffs.append(z_diff)
            num, lin = [0] * n, [0] * n
            for j, key in enumerate(keys):
                # R = P mod Z_i is the unique polynomial of degree < |points| agreeing with P on the points
                _, r = poly_divmod(polys[key], z_i_poly)
                for pnt in pts:
                    ev = [e for kk, pp, e in queries if kk == key and pp == pnt][0]
                    assert sum(c * pow(pnt, t, P) for t, c in enumerate(r)) % P == ev
                num = add_scaled(num, add_scaled(polys[key], r, P - 1), pow(y, j, P))
                r_u = sum(c * pow(u, t, P) for t, c in enumerate(r)) % P
                lin = add_scaled(lin, add_scaled(polys[key], [r_u], P - 1), pow(y, j, P))
            q, rem = poly_divmod(num, z_i_poly)
            assert not any(rem)
            h_x = add_scaled(h_x, q, pow(v, i, P))
            l_x = add_scaled(l_x, lin, pow(v, i, P) * z_diff % P)
        zt = 1
        for p_ in all_points:
            zt = zt * (u - p_) % P
        l_x = add_scaled(l_x, h_x, (P - zt) % P)
        fin, rem = poly_divmod(l_x, [(P - u) % P, 1])
        assert not any(rem)
        z0_inv = pow(z_diffs[0], P - 2, P)
        fin = [c * z0_inv % P for c in fin]
        commit = lambda cs: bn.g1_mul(bn.G1_GENERATOR, sum(c * (i + 1) for i, c in enumerate(cs)) % P)
        h1, h2 = commit(h_x), commit(fin)
        model = hm.shplonk_create_proof(dict(enumerate(polys)), queries, y, v, u, commit)
        assert model == (h1, h2, h_x, fin)
        out["cases"].append({"k": k, "polys": [[hx(c) for c in p_] for p_ in polys],
                             "queries": [[key, hx(pnt), hx(ev)] for key, pnt, ev in queries], "y": hx(y), "v": hx(v),
                             "u": hx(u), "h_x": [hx(c) for c in h_x], "final": [hx(c) for c in fin], "h1": pt(h1),
                             "h2": pt(h2)})
    for n in (2, 7, 33):
        a = [rnd.randrange(P) for _ in range(n)]
        z = rnd.randrange(P)
        q, _ = poly_divmod(a, [(P - z) % P, 1])
        assert hm.kate_division(a, z) == q
        out["kate_division"].append({"a": [hx(c) for c in a], "point": hx(z), "q": [hx(c) for c in q]})
    lens = [9, 4, 9, 1, 6]
    ps = [[rnd.randrange(P) for _ in range(l)] for l in lens]
    cs = [rnd.randrange(P), 0, P - 1, 1, rnd.randrange(P)]
    res = [sum(c * p_[i] for c, p_ in zip(cs, ps) if i < len(p_)) % P for i in range(9)]
    out["lincomb"].append({"polys": [[hx(c) for c in p_] for p_ in ps], "coeffs": [hx(c) for c in cs],
                           "n": 9, "out": [hx(c) for c in res]})
    dump("shplonk.json", out)


def gen_lookup_permute():
    """permute_expression_pair from its specification rather than upstream's loop: A' is A sorted; among all S' that are
    permutations of S with S'[i] = A'[i] on first occurrences, upstream's is the one whose remaining rows, read from the
    last repeated row backwards, are ascending.  Built here by multiset subtraction with collections.Counter."""
    import collections
    rnd = random.Random(SEED + 13)
    out = {"comment": "canonical hex; rows [0, usable_rows) only", "cases": []}
    for k, usable, tsize in [(3, 5, 3), (5, 26, 9), (6, 58, 58)]:
        n = 1 << k
        tvals = [rnd.randrange(P) for _ in range(tsize - 1)] + [0]
        table = [tvals[i % tsize] if i >= tsize else tvals[i] for i in range(n)]
        rnd.shuffle(table)
        present = sorted(set(table[:usable]))
        inputs = [rnd.choice(present[:max(1, len(present) // 2)]) for _ in range(n)]
        a_sorted = sorted(inputs[:usable])
        firsts = [i for i in range(usable) if i == 0 or a_sorted[i] != a_sorted[i - 1]]
        rest = collections.Counter(table[:usable]) - collections.Counter(a_sorted[i] for i in firsts)
        left = sorted(rest.elements())
        repeated = [i for i in range(usable) if i not in set(firsts)]
        assert len(left) == len(repeated)
        s_perm = [None] * usable
        for i in firsts:
            s_perm[i] = a_sorted[i]
        for v, row in zip(left, reversed(repeated)):
            s_perm[row] = v
        assert hm.permute_expression_pair(inputs, table, usable) == (a_sorted, s_perm)
        out["cases"].append({"k": k, "usable_rows": usable, "input": [hx(v) for v in inputs], "table": [hx(v) for v in table],
                             "permuted_input": [hx(v) for v in a_sorted], "permuted_table": [hx(v) for v in s_perm]})
    dump("lookup_permute.json", out)


if __name__ == "__main__":
    bn.self_check()
    gen_field()
    gen_reference_kats()
    gen_curve()
    gen_msm()
    gen_ntt()
    gen_domain()
    gen_evalh()
    gen_products()
    gen_shplonk()
    gen_lookup_permute()
