"""Big-int model of `halo2_proofs::plonk::verifier::verify_proof` for KZG + SHPLONK (TEST INFRASTRUCTURE).

The reference checks every proof it makes with `verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<_>,
Challenge255<_>, Blake2bRead<_,_,_>, SingleStrategy<_>>` (/root/reference/prover/src/test/proof_test.rs:317-330,
/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:423-436).  This file restates that verifier from the
published halo2 v0.3.0 algorithm on Python integers: it replays the transcript, evaluates the gate, permutation and
lookup identities from the opened evaluations (an independent route to the quotient identity: expression trees on
single field elements, not the straight-line program on cosets that the prover ran), and checks the SHPLONK opening.

The final pairing check e(h2, [tau]_2 - [u]_2) = e(L, [1]_2) is done in G1 with the known trapdoor of a test SRS:
(tau - u) * z_0(u) * h2 = L.  That is the same equation, usable only where tau is known (upstream's `unsafe_setup`
situation); no pairing code is needed.

Only tests/ may import this module.  Parity unpinned: the reference holds no proof bytes to replay; what this model
pins is that a proof produced on the device satisfies the verifier's equations.
"""
from __future__ import annotations

import hashlib

import bn254 as bn
import halo2_model as hm

P = bn.FR_MODULUS
Q = bn.FQ_MODULUS


class VerifyError(Exception):
    pass


def decompress_point(b: bytes):
    b = bytearray(b)
    if b[31] & 0x80:
        return None
    sign = (b[31] >> 6) & 1
    b[31] &= 0x3F
    x = int.from_bytes(b, "little")
    if x >= Q:
        raise VerifyError("point: x out of range")
    y2 = (x * x * x + bn.G1_B) % Q
    y = pow(y2, (Q + 1) // 4, Q)
    if y * y % Q != y2:
        raise VerifyError("point: not on the curve")
    if (y & 1) != sign:
        y = Q - y
    return (x, y)


class Blake2bRead:
    def __init__(self, proof: bytes):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.buf, self.pos = bytes(proof), 0

    def _take(self, k):
        if self.pos + k > len(self.buf):
            raise VerifyError("proof too short")
        out = self.buf[self.pos:self.pos + k]
        self.pos += k
        return out

    def squeeze_challenge(self) -> int:
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % P

    def common_point(self, xy):
        if xy is None:
            raise VerifyError("point at infinity in the transcript")
        self.state.update(b"\x01" + xy[0].to_bytes(32, "little") + xy[1].to_bytes(32, "little"))

    def common_scalar(self, v: int):
        self.state.update(b"\x02" + (v % P).to_bytes(32, "little"))

    def read_point(self):
        xy = decompress_point(self._take(32))
        self.common_point(xy)
        return xy

    def read_scalar(self) -> int:
        v = int.from_bytes(self._take(32), "little")
        if v >= P:
            raise VerifyError("scalar out of range")
        self.common_scalar(v)
        return v


def collect_queries(cs):
    """[(column, rotation)] per column kind in order of first appearance: gates, lookups, permutation columns."""
    out = {"advice": [], "fixed": [], "instance": []}

    def add(kind, q):
        if q not in out[kind]:
            out[kind].append(q)

    def walk(e):
        if e[0] in out:
            add(e[0], (e[1], e[2]))
        elif e[0] in ("neg", "scaled"):
            walk(e[1])
        elif e[0] in ("sum", "prod"):
            walk(e[1])
            walk(e[2])

    for gate in cs.gates:
        for poly in gate:
            walk(poly)
    for lk in cs.lookups:
        for e in lk.input_expressions + lk.table_expressions:
            walk(e)
    names = {3: "advice", 2: "fixed", 4: "instance"}     # Any::{Advice, Fixed, Instance} as the constraint system numbers them
    for kind, idx in cs.permutation_columns:
        add(names[kind], (idx, 0))
    # a constraint system that carries its own query lists (the order of the circuit's first query_* calls, as upstream's
    # ConstraintSystem records it) is read in that order, like upstream's verifier reads `cs.advice_queries`
    res = []
    for kind in ("advice", "fixed", "instance"):
        given = getattr(cs, kind + "_queries", None)
        if given is not None:
            given = [tuple(q) for q in given]
            if len(set(given)) != len(given) or not set(out[kind]) <= set(given):
                raise VerifyError(kind + "_queries does not list every query once")
            res.append(given)
        else:
            res.append(out[kind])
    return tuple(res)


def eval_expression(e, ev, challenges):
    t = e[0]
    if t == "const":
        return e[1] % P
    if t in ("fixed", "advice", "instance"):
        return ev[t][(e[1], e[2])]
    if t == "challenge":
        return challenges[e[1]]
    if t == "neg":
        return (-eval_expression(e[1], ev, challenges)) % P
    if t == "sum":
        return (eval_expression(e[1], ev, challenges) + eval_expression(e[2], ev, challenges)) % P
    if t == "prod":
        return eval_expression(e[1], ev, challenges) * eval_expression(e[2], ev, challenges) % P
    if t == "scaled":
        return eval_expression(e[1], ev, challenges) * e[2] % P
    raise ValueError(t)


def verify_proof(cs, k: int, vk: dict, instance, proof: bytes, tau: int) -> None:
    """vk: {"fixed_commitments", "permutation_commitments": [(x, y)], "transcript_repr": int}; instance: per column the
    public values; tau: the trapdoor of the test SRS g[i] = tau^i * G.  Raises VerifyError when a check fails."""
    n = 1 << k
    dom = hm.EvaluationDomain(cs.degree(), k)
    omega = dom.omega
    bf = cs.blinding_factors()
    usable = n - (bf + 1)
    chunk = cs.permutation_chunk_len()
    pcols = cs.permutation_columns
    sets = cs.num_permutation_sets()
    pieces = cs.degree() - 1
    adv_q, fix_q, inst_q = collect_queries(cs)

    t = Blake2bRead(proof)
    t.common_scalar(vk["transcript_repr"])
    if len(instance) != cs.num_instance_columns:
        raise VerifyError("InvalidInstances")
    for col in instance:
        if len(col) > usable:
            raise VerifyError("InstanceTooLarge")
        for v in col:
            t.common_scalar(v)
    advice_c = [None] * cs.num_advice_columns
    challenges = [0] * cs.num_challenges
    adv_phase = cs.advice_column_phase or [0] * cs.num_advice_columns
    chl_phase = cs.challenge_phase or [0] * cs.num_challenges
    for phase in sorted(set(adv_phase) | set(chl_phase)):       # per phase: its commitments, then its challenges
        for c in range(cs.num_advice_columns):
            if adv_phase[c] == phase:
                advice_c[c] = t.read_point()
        for c in range(cs.num_challenges):
            if chl_phase[c] == phase:
                challenges[c] = t.squeeze_challenge()
    theta = t.squeeze_challenge()
    lk_permuted = [(t.read_point(), t.read_point()) for _ in cs.lookups]
    beta = t.squeeze_challenge()
    gamma = t.squeeze_challenge()
    perm_c = [t.read_point() for _ in range(sets)]
    lk_product = [t.read_point() for _ in cs.lookups]
    random_c = t.read_point()
    y = t.squeeze_challenge()
    h_c = [t.read_point() for _ in range(pieces)]
    x = t.squeeze_challenge()

    advice_evals = {q: t.read_scalar() for q in adv_q}
    fixed_evals = {q: t.read_scalar() for q in fix_q}
    random_eval = t.read_scalar()
    sigma_evals = [t.read_scalar() for _ in pcols]
    perm_evals = []
    for s in range(sets):
        z, z_next = t.read_scalar(), t.read_scalar()
        perm_evals.append((z, z_next, t.read_scalar() if s + 1 < sets else None))
    lk_evals = [tuple(t.read_scalar() for _ in range(5)) for _ in cs.lookups]     # z, z_next, a', a'_prev, s'

    # Lagrange basis values at a point: L_i(p) = omega^i (p^n - 1) / (n (p - omega^i))
    n_inv = pow(n, P - 2, P)

    def lagrange_at(pt, rows):
        num = (pow(pt, n, P) - 1) * n_inv % P
        return {i: num * pow(omega, i, P) % P * pow((pt - pow(omega, i, P)) % P, P - 2, P) % P for i in rows}

    def point(rot):
        return x * pow(omega, rot % n, P) % P

    instance_evals = {}
    for col, rot in inst_q:
        li = lagrange_at(point(rot), range(len(instance[col])))
        instance_evals[(col, rot)] = sum(v * li[i] for i, v in enumerate(instance[col])) % P
    ev = {"advice": advice_evals, "fixed": fixed_evals, "instance": instance_evals}

    xn = pow(x, n, P)
    lx = lagrange_at(x, [0] + list(range(usable, n)))
    l_0, l_last = lx[0], lx[usable]
    l_blind = sum(lx[i] for i in range(usable + 1, n)) % P
    active = (1 - l_last - l_blind) % P

    expressions = []
    for gate in cs.gates:
        for poly in gate:
            expressions.append(eval_expression(poly, ev, challenges))
    # permutation argument
    if sets:
        kind_name = {3: "advice", 2: "fixed", 4: "instance"}
        expressions.append(l_0 * (1 - perm_evals[0][0]) % P)
        z_l = perm_evals[-1][0]
        expressions.append(l_last * (z_l * z_l - z_l) % P)
        for s in range(1, sets):
            expressions.append(l_0 * (perm_evals[s][0] - perm_evals[s - 1][2]) % P)
        for s in range(sets):
            z, z_next, _ = perm_evals[s]
            left, right = z_next, z
            cur = beta * x % P * pow(bn.FR_DELTA, s * chunk, P) % P
            for j, (kind, idx) in enumerate(pcols[s * chunk:(s + 1) * chunk]):
                col_eval = ev[kind_name[kind]][(idx, 0)]
                left = left * ((col_eval + beta * sigma_evals[s * chunk + j] + gamma) % P) % P
                right = right * ((col_eval + cur + gamma) % P) % P
                cur = cur * bn.FR_DELTA % P
            expressions.append((left - right) * active % P)
    # lookups
    for lk, (z, z_next, pa, pa_prev, ps) in zip(cs.lookups, lk_evals):
        def compress(exprs):
            acc = 0
            for e in exprs:
                acc = (acc * theta + eval_expression(e, ev, challenges)) % P
            return acc
        expressions.append(l_0 * (1 - z) % P)
        expressions.append(l_last * (z * z - z) % P)
        left = z_next * (pa + beta) % P * (ps + gamma) % P
        right = z * (compress(lk.input_expressions) + beta) % P * (compress(lk.table_expressions) + gamma) % P
        expressions.append((left - right) * active % P)
        expressions.append(l_0 * (pa - ps) % P)
        expressions.append(active * (pa - ps) % P * (pa - pa_prev) % P)

    h_fold = 0
    for v in expressions:
        h_fold = (h_fold * y + v) % P
    expected_h_eval = h_fold * pow(xn - 1, P - 2, P) % P
    h_commitment = None
    for c in reversed(h_c):
        h_commitment = bn.g1_add(bn.g1_mul(h_commitment, xn) if h_commitment else None, c)

    # queries in the prover's order: (commitment, point, eval)
    last_rot = -(bf + 1)
    queries = [(("advice", c), point(r), advice_evals[(c, r)]) for c, r in adv_q]
    for s in range(sets):
        z, z_next, z_last = perm_evals[s]
        queries += [(("perm", s), point(0), z), (("perm", s), point(1), z_next)]
        if s + 1 < sets:
            queries.append((("perm", s), point(last_rot), z_last))
    for i, (z, z_next, pa, pa_prev, ps) in enumerate(lk_evals):
        queries += [(("lk_z", i), point(0), z), (("lk_a", i), point(0), pa), (("lk_s", i), point(0), ps),
                    (("lk_a", i), point(-1), pa_prev), (("lk_z", i), point(1), z_next)]
    queries += [(("fixed", c), point(r), fixed_evals[(c, r)]) for c, r in fix_q]
    queries += [(("sigma", j), point(0), sigma_evals[j]) for j in range(len(pcols))]
    queries += [(("h",), point(0), expected_h_eval), (("random",), point(0), random_eval)]
    commitment = {("h",): h_commitment, ("random",): random_c}
    commitment.update({("advice", c): p for c, p in enumerate(advice_c)})
    commitment.update({("perm", s): p for s, p in enumerate(perm_c)})
    for i in range(len(cs.lookups)):
        commitment[("lk_z", i)] = lk_product[i]
        commitment[("lk_a", i)], commitment[("lk_s", i)] = lk_permuted[i]
    commitment.update({("fixed", c): p for c, p in enumerate(vk["fixed_commitments"])})
    commitment.update({("sigma", j): p for j, p in enumerate(vk["permutation_commitments"])})

    # SHPLONK
    y_s = t.squeeze_challenge()
    v_s = t.squeeze_challenge()
    h1 = t.read_point()
    u = t.squeeze_challenge()
    h2 = t.read_point()
    if t.pos != len(t.buf):
        raise VerifyError("trailing bytes in the proof")
    rotation_sets, super_points = hm.construct_intermediate_sets(queries)
    acc_point, acc_const = None, 0
    z_0 = None
    power_v = 1
    for pts, commitments in rotation_sets:
        z_i = hm.evaluate_vanishing_polynomial([p for p in super_points if p not in pts], u)
        if z_0 is None:
            z_0 = z_i
        power_y = 1
        for key, evals in commitments:
            c = power_v * z_i % P * power_y % P
            acc_point = bn.g1_add(acc_point, bn.g1_mul(commitment[key], c))
            acc_const = (acc_const + c * hm.eval_poly(hm.lagrange_interpolate(pts, evals), u)) % P
            power_y = power_y * y_s % P
        power_v = power_v * v_s % P
    zt = hm.evaluate_vanishing_polynomial(super_points, u)
    lhs = bn.g1_add(acc_point, bn.g1_neg(bn.g1_mul(bn.G1_GENERATOR, acc_const)))
    lhs = bn.g1_add(lhs, bn.g1_neg(bn.g1_mul(h1, zt)))
    rhs = bn.g1_mul(h2, (tau - u) * z_0 % P)
    if lhs != rhs:
        raise VerifyError("opening proof is invalid (the quotient identity or an evaluation does not hold)")
