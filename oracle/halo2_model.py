"""Python big-int model of the halo2 polynomial pipeline (TEST INFRASTRUCTURE, not a product path).

Restates, on plain Python integers, the algorithms of the out-of-tree `halo2_proofs 0.3.0` fork the
reference depends on (/root/reference/zkevm-circuits/Cargo.toml:15, /root/reference/Cargo.lock:2189-2211):
`arithmetic::{best_multiexp,best_fft}`, `poly::EvaluationDomain`, `plonk::evaluation::{GraphEvaluator,
Evaluator::evaluate_h}`.  The fork's source is not in the reference tree; the reference reaches these
functions through `create_proof` (/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:265,315;
/root/reference/prover/src/chunk/prover.rs:85).  PARITY UNPINNED at this boundary: the reference holds
no golden commitments/cosets/quotients, so this model is checked against direct mathematical
definitions (O(n^2) DFT, direct polynomial evaluation, expression-tree evaluation) instead.

Values here are canonical integers mod r; conversion to the in-memory Montgomery limbs is in bn254.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

from bn254 import (FR_DELTA, FR_MODULUS, FR_ROOT_OF_UNITY, FR_S, FR_ZETA, g1_add, g1_neg)

P = FR_MODULUS


# ---------------------------------------------------------------------------------------------
# arithmetic::best_fft  (in place, natural order in and out:  out[i] = sum_j a[j] * omega^(i*j))
# ---------------------------------------------------------------------------------------------
def bitreverse(n: int, l: int) -> int:
    r = 0
    for _ in range(l):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def best_fft(a, omega: int, log_n: int):
    n = 1 << log_n
    assert len(a) == n
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [1] * max(n // 2, 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % P
    m = 1
    while m < n:
        step = n // (2 * m)
        for k in range(0, n, 2 * m):
            for j in range(m):
                t = a[k + j + m] * tw[j * step] % P
                a[k + j + m] = (a[k + j] - t) % P
                a[k + j] = (a[k + j] + t) % P
        m *= 2
    return a


def dft_naive(a, omega: int):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, P) for j in range(n)) % P for i in range(n)]


# ---------------------------------------------------------------------------------------------
# poly::EvaluationDomain
# ---------------------------------------------------------------------------------------------
class EvaluationDomain:
    """`EvaluationDomain::new(j, k)`: j = constraint-system degree, n = 2^k rows."""

    def __init__(self, j: int, k: int):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ext_k = k
        while (1 << ext_k) < self.n * self.quotient_poly_degree:
            ext_k += 1
        self.extended_k = ext_k
        assert ext_k <= FR_S
        w = FR_ROOT_OF_UNITY
        for _ in range(ext_k, FR_S):
            w = w * w % P
        self.extended_omega = w
        self.extended_omega_inv = pow(w, -1, P)
        for _ in range(k, ext_k):
            w = w * w % P
        self.omega = w
        self.omega_inv = pow(w, -1, P)
        self.g_coset = FR_ZETA
        self.g_coset_inv = FR_ZETA * FR_ZETA % P
        self.ifft_divisor = pow(self.n, -1, P)
        self.extended_ifft_divisor = pow(1 << ext_k, -1, P)
        # t_evaluations stores 1 / ((zeta * extended_omega^i)^n - 1) for i < 2^(ext_k - k)
        self.t_evaluations = [
            pow((pow(self.g_coset * pow(self.extended_omega, i, P) % P, self.n, P) - 1) % P, -1, P)
            for i in range(1 << (ext_k - k))
        ]

    def extended_len(self):
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a):
        a = list(a)
        best_fft(a, self.omega_inv, self.k)
        return [v * self.ifft_divisor % P for v in a]

    def coeff_to_lagrange(self, a):
        return best_fft(list(a), self.omega, self.k)

    def _distribute_powers_zeta(self, a, into_coset: bool):
        cz = [1, self.g_coset, self.g_coset_inv] if into_coset else [1, self.g_coset_inv, self.g_coset]
        return [v * cz[i % 3] % P for i, v in enumerate(a)]

    def coeff_to_extended(self, a):
        assert len(a) == self.n
        a = self._distribute_powers_zeta(a, True)
        a += [0] * (self.extended_len() - self.n)
        return best_fft(a, self.extended_omega, self.extended_k)

    def extended_to_coeff(self, a):
        assert len(a) == self.extended_len()
        a = best_fft(list(a), self.extended_omega_inv, self.extended_k)
        a = [v * self.extended_ifft_divisor % P for v in a]
        a = self._distribute_powers_zeta(a, False)
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        m = len(self.t_evaluations)
        return [v * self.t_evaluations[i % m] % P for i, v in enumerate(a)]

    def rotate_omega(self, value: int, rotation: int) -> int:
        if rotation >= 0:
            return value * pow(self.omega, rotation, P) % P
        return value * pow(self.omega_inv, -rotation, P) % P


def eval_poly(coeffs, x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


# ---------------------------------------------------------------------------------------------
# arithmetic::multiexp_serial / best_multiexp (unsigned c-bit windows, c = ceil(ln n))
# ---------------------------------------------------------------------------------------------
def multiexp_serial(coeffs, bases):
    n = len(coeffs)
    if n < 4:
        c = 1
    elif n < 32:
        c = 3
    else:
        c = math.ceil(math.log(n))
    segments = 256 // c + 1
    acc = None
    for seg in reversed(range(segments)):
        for _ in range(c):
            acc = g1_add(acc, acc)
        buckets = [None] * ((1 << c) - 1)
        for s, b in zip(coeffs, bases):
            w = (s >> (seg * c)) & ((1 << c) - 1)
            if w:
                buckets[w - 1] = g1_add(buckets[w - 1], b)
        running = None
        for b in reversed(buckets):
            running = g1_add(running, b)
            acc = g1_add(acc, running)
    return acc


def best_multiexp(coeffs, bases, threads: int = 1):
    assert len(coeffs) == len(bases)
    n = len(coeffs)
    if n > threads:
        chunk = n // threads
        acc = None
        for s in range(0, n, chunk):
            acc = g1_add(acc, multiexp_serial(coeffs[s:s + chunk], bases[s:s + chunk]))
        return acc
    return multiexp_serial(coeffs, bases)


# ---------------------------------------------------------------------------------------------
# plonk::evaluation::GraphEvaluator (serialised form) and its interpreter
#
# ValueSource = (kind, a, b);  Calculation = (op, args...) ; program = dict with
#   constants: [int], rotations: [int], calculations: [(op, ...)], num_intermediates
# Each calculation i writes intermediate i (target == position, as upstream assigns them).
# ---------------------------------------------------------------------------------------------
VS_CONSTANT, VS_INTERMEDIATE, VS_FIXED, VS_ADVICE, VS_INSTANCE, VS_CHALLENGE, VS_BETA, VS_GAMMA, \
    VS_THETA, VS_Y, VS_PREVIOUS = range(11)
OP_ADD, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_HORNER, OP_STORE = range(8)


def rotation_idx(idx: int, rot: int, rot_scale: int, isize: int) -> int:
    return (idx + rot * rot_scale) % isize


class EvalInputs:
    """Column cosets and per-proof scalars handed to the evaluator (all canonical ints)."""

    def __init__(self, fixed, advice, instance, challenges, beta, gamma, theta, y):
        self.fixed, self.advice, self.instance = fixed, advice, instance
        self.challenges = challenges
        self.beta, self.gamma, self.theta, self.y = beta, gamma, theta, y


def graph_evaluate(prog, inp: EvalInputs, previous_value: int, idx: int, rot_scale: int, isize: int) -> int:
    rots = [rotation_idx(idx, r, rot_scale, isize) for r in prog["rotations"]]
    inter = [0] * prog["num_intermediates"]
    consts = prog["constants"]

    def get(vs):
        kind, a, b = vs
        if kind == VS_CONSTANT:
            return consts[a]
        if kind == VS_INTERMEDIATE:
            return inter[a]
        if kind == VS_FIXED:
            return inp.fixed[a][rots[b]]
        if kind == VS_ADVICE:
            return inp.advice[a][rots[b]]
        if kind == VS_INSTANCE:
            return inp.instance[a][rots[b]]
        if kind == VS_CHALLENGE:
            return inp.challenges[a]
        if kind == VS_BETA:
            return inp.beta
        if kind == VS_GAMMA:
            return inp.gamma
        if kind == VS_THETA:
            return inp.theta
        if kind == VS_Y:
            return inp.y
        if kind == VS_PREVIOUS:
            return previous_value
        raise ValueError(kind)

    last = 0
    for target, calc in enumerate(prog["calculations"]):
        op = calc[0]
        if op == OP_ADD:
            v = (get(calc[1]) + get(calc[2])) % P
        elif op == OP_SUB:
            v = (get(calc[1]) - get(calc[2])) % P
        elif op == OP_MUL:
            v = get(calc[1]) * get(calc[2]) % P
        elif op == OP_SQUARE:
            v = get(calc[1]) ** 2 % P
        elif op == OP_DOUBLE:
            v = 2 * get(calc[1]) % P
        elif op == OP_NEGATE:
            v = (-get(calc[1])) % P
        elif op == OP_HORNER:
            f = get(calc[3])
            v = get(calc[1])
            for part in calc[2]:
                v = (v * f + get(part)) % P
        elif op == OP_STORE:
            v = get(calc[1])
        else:
            raise ValueError(op)
        inter[target] = v
        last = v
    return last


# ---------------------------------------------------------------------------------------------
# Evaluator::evaluate_h  (single circuit instance)
#
# desc keys:
#   gates: program | None
#   lookups: [ {program, product, permuted_input, permuted_table} ]   (extended cosets)
#   perm: None | {columns: [(kind, index)], sigma: [coset], z: [coset], chunk_len, blinding_factors}
#   l0, l_last, l_active: extended cosets
# ---------------------------------------------------------------------------------------------
def evaluate_h_row(dom: EvaluationDomain, inp: EvalInputs, desc, idx: int) -> int:
    """One row of evaluate_h.  Upstream sweeps the gates over all rows, then the permutation argument, then each lookup;
    the value of a row depends on that row's running value only, so the sweeps commute with the row loop.  Columns are
    only ever indexed, so they may be any object with __getitem__ (e.g. a lazy view of device memory)."""
    size = dom.extended_len()
    rot_scale = 1 << (dom.extended_k - dom.k)
    y, beta, gamma = inp.y, inp.beta, inp.gamma
    l0, l_last, l_active = desc["l0"], desc["l_last"], desc["l_active"]
    v = 0
    if desc.get("gates") is not None:
        v = graph_evaluate(desc["gates"], inp, v, idx, rot_scale, size)

    perm = desc.get("perm")
    if perm and perm["z"]:
        z = perm["z"]
        last_rot = -(perm["blinding_factors"] + 1)
        chunk_len = perm["chunk_len"]
        cols = perm["columns"]
        delta_start = beta * FR_ZETA % P
        src = {VS_ADVICE: inp.advice, VS_FIXED: inp.fixed, VS_INSTANCE: inp.instance}
        r_next = rotation_idx(idx, 1, rot_scale, size)
        r_last = rotation_idx(idx, last_rot, rot_scale, size)
        v = (v * y + (1 - z[0][idx]) * l0[idx]) % P
        zl = z[-1][idx]
        v = (v * y + (zl * zl - zl) * l_last[idx]) % P
        for s in range(1, len(z)):
            v = (v * y + (z[s][idx] - z[s - 1][r_last]) * l0[idx]) % P
        current_delta = delta_start * pow(dom.extended_omega, idx, P) % P
        for s in range(len(z)):
            chunk = cols[s * chunk_len:(s + 1) * chunk_len]
            sig = perm["sigma"][s * chunk_len:(s + 1) * chunk_len]
            left = z[s][r_next]
            for (kind, ci), sg in zip(chunk, sig):
                left = left * (src[kind][ci][idx] + beta * sg[idx] + gamma) % P
            right = z[s][idx]
            for (kind, ci) in chunk:
                right = right * (src[kind][ci][idx] + current_delta + gamma) % P
                current_delta = current_delta * FR_DELTA % P
            v = (v * y + (left - right) * l_active[idx]) % P

    for lk in desc.get("lookups", []):
        zc, ac, sc = lk["product"], lk["permuted_input"], lk["permuted_table"]
        table_value = graph_evaluate(lk["program"], inp, 0, idx, rot_scale, size)
        r_next = rotation_idx(idx, 1, rot_scale, size)
        r_prev = rotation_idx(idx, -1, rot_scale, size)
        a_minus_s = (ac[idx] - sc[idx]) % P
        v = (v * y + (1 - zc[idx]) * l0[idx]) % P
        v = (v * y + (zc[idx] * zc[idx] - zc[idx]) * l_last[idx]) % P
        v = (v * y + (zc[r_next] * (ac[idx] + beta) * (sc[idx] + gamma) - zc[idx] * table_value)
             * l_active[idx]) % P
        v = (v * y + a_minus_s * l0[idx]) % P
        v = (v * y + a_minus_s * (ac[idx] - ac[r_prev]) * l_active[idx]) % P
    return v


def evaluate_h(dom: EvaluationDomain, inp: EvalInputs, desc) -> list:
    return [evaluate_h_row(dom, inp, desc, idx) for idx in range(dom.extended_len())]


# ---------------------------------------------------------------------------------------------
# Grand products and openings: the prover steps right after the hot path inside create_proof
# (`ff::BatchInvert`, `plonk::permutation::prover::Argument::commit`,
#  `plonk::lookup::prover::Permuted::commit_product`, `arithmetic::eval_polynomial`; the reference reaches them
#  through /root/reference/zkevm-circuits/benches/benchmark_list/main.rs:265).  Blinding rows are not modelled:
#  upstream fills the last `blinding_factors` rows of z from the prover's RNG after these computations.
# ---------------------------------------------------------------------------------------------
def batch_invert(values):
    """Every non-zero element replaced by its inverse; zeros stay (ff::BatchInvert)."""
    acc = 1
    tmp = []
    for v in values:
        tmp.append(acc)
        if v:
            acc = acc * v % P
    acc = pow(acc, P - 2, P)
    out = list(values)
    for i in range(len(values) - 1, -1, -1):
        v = values[i]
        if v:
            out[i] = tmp[i] * acc % P
            acc = acc * v % P
    return out


def prefix_product(m, init: int):
    """out[0] = init, out[i] = init * m[0] * ... * m[i-1]; len(m) outputs."""
    out = []
    acc = init % P
    for v in m:
        out.append(acc)
        acc = acc * v % P
    return out


def permutation_product(omega: int, values, sigmas, beta: int, gamma: int, delta_start: int, z0: int):
    """One column set of permutation::prover::Argument::commit, all n rows of z (before blinding).
    values[j], sigmas[j]: Lagrange-basis columns; delta_start = delta^(index of the set's first column)."""
    n = len(values[0])
    modified = [1] * n
    for v, s in zip(values, sigmas):
        for i in range(n):
            modified[i] = modified[i] * ((beta * s[i] + gamma + v[i]) % P) % P
    modified = batch_invert(modified)
    deltaomega = delta_start
    for v in values:
        cur = deltaomega
        for i in range(n):
            modified[i] = modified[i] * ((cur * beta + gamma + v[i]) % P) % P
            cur = cur * omega % P
        deltaomega = deltaomega * FR_DELTA % P
    return prefix_product(modified, z0)


def lookup_product(compressed_input, compressed_table, permuted_input, permuted_table, beta: int, gamma: int):
    """lookup::prover::Permuted::commit_product, all n rows of z (before blinding)."""
    n = len(compressed_input)
    prod = [((beta + permuted_input[i]) * (gamma + permuted_table[i])) % P for i in range(n)]
    prod = batch_invert(prod)
    for i in range(n):
        prod[i] = prod[i] * ((compressed_input[i] + beta) % P) % P * ((compressed_table[i] + gamma) % P) % P
    return prefix_product(prod, 1)


# ---------------------------------------------------------------------------------------------
# arithmetic::{kate_division, lagrange_interpolate} and the SHPLONK opening prover
# (`poly::kzg::multiopen::shplonk::{construct_intermediate_sets, prover::ProverSHPLONK::create_proof}`;
# selected by the reference at /root/reference/zkevm-circuits/benches/benchmark_list/main.rs:26,267 and
# /root/reference/prover/src/test/proof_test.rs:297).  Restated from the published upstream algorithm.
# ---------------------------------------------------------------------------------------------
def kate_division(a, point: int):
    """Coefficients of a(X) / (X - point), remainder dropped: upstream walks from the leading coefficient down,
    q[i-1] = a[i] + point * q[i]."""
    q = [0] * (len(a) - 1)
    tmp = 0
    for i in range(len(a) - 1, 0, -1):
        lead = (a[i] + tmp) % P
        q[i - 1] = lead
        tmp = lead * point % P
    return q


def lagrange_interpolate(points, evals):
    """Coefficients (lowest first) of the polynomial of degree < len(points) through (points[i], evals[i])."""
    assert len(points) == len(evals)
    n = len(points)
    if n == 1:
        return [evals[0] % P]
    out = [0] * n
    for j in range(n):
        # numerator prod_{k != j} (X - x_k), denominator prod_{k != j} (x_j - x_k)
        num = [1]
        den = 1
        for k in range(n):
            if k == j:
                continue
            num = [((num[i - 1] if i > 0 else 0) - points[k] * (num[i] if i < len(num) else 0)) % P
                   for i in range(len(num) + 1)]
            den = den * (points[j] - points[k]) % P
        scale = evals[j] * pow(den, P - 2, P) % P
        for i in range(n):
            out[i] = (out[i] + num[i] * scale) % P
    return out


def evaluate_vanishing_polynomial(roots, z: int) -> int:
    acc = 1
    for r in roots:
        acc = acc * (z - r) % P
    return acc


def construct_intermediate_sets(queries):
    """queries: list of (commitment_key, point, eval); commitment_key is any hashable identity of the polynomial.
    Returns (rotation_sets, super_point_set): rotation_sets = [(points_sorted, [(commitment_key, evals)])] in upstream's
    order (first appearance of the commitment, then first appearance of its point set); point sets are ordered as a
    BTreeSet<Fr>, i.e. by canonical integer value."""
    def get_eval(key, point):
        for k, p, e in queries:
            if k == key and p == point:
                return e
        raise KeyError
    super_points = set()
    commitment_map = []      # [(key, set(points))]
    for key, point, _ in queries:
        super_points.add(point)
        for entry in commitment_map:
            if entry[0] == key:
                entry[1].add(point)
                break
        else:
            commitment_map.append((key, {point}))
    set_map = []             # [(frozenset(points), [keys])]
    for key, pts in commitment_map:
        fs = frozenset(pts)
        for entry in set_map:
            if entry[0] == fs:
                entry[1].append(key)
                break
        else:
            set_map.append((fs, [key]))
    rotation_sets = []
    for fs, keys in set_map:
        pts = sorted(fs)
        rotation_sets.append((pts, [(key, [get_eval(key, p) for p in pts]) for key in keys]))
    return rotation_sets, sorted(super_points)


def shplonk_create_proof(polys, queries, y: int, v: int, u: int, commit):
    """polys: {key: coefficient list of length n}; queries as in construct_intermediate_sets; y, v, u: the three
    transcript challenges in the order upstream squeezes them (u after the first commitment is written);
    commit(coeffs) -> point.  Returns (h1_commitment, h2_commitment, h_x, final_poly)."""
    n = len(next(iter(polys.values())))
    rotation_sets, super_points = construct_intermediate_sets(queries)

    def lincomb(items, base):
        acc, power = None, 1
        for poly in items:
            scaled = [c * power % P for c in poly]
            acc = scaled if acc is None else [(a + b) % P for a, b in zip(acc, scaled)]
            power = power * base % P
        return acc

    ext = []   # per rotation set: (points, [(poly, low_degree_equivalent)])
    for pts, commitments in rotation_sets:
        ext.append((pts, [(polys[key], lagrange_interpolate(pts, evals)) for key, evals in commitments]))

    quotients = []
    for pts, cs in ext:
        numerators = []
        for poly, low in cs:
            num = list(poly)
            for i, r in enumerate(low):
                num[i] = (num[i] - r) % P
            numerators.append(num)
        n_x = lincomb(numerators, y)
        for p in pts:
            n_x = kate_division(n_x, p)
        quotients.append(n_x + [0] * (n - len(n_x)))
    h_x = lincomb(quotients, v)
    h1 = commit(h_x)

    contributions, z_diffs = [], []
    for pts, cs in ext:
        diffs = [p for p in super_points if p not in pts]
        z_i = evaluate_vanishing_polynomial(diffs, u)
        inner = []
        for poly, low in cs:
            r_eval = eval_poly(low, u)
            lp = list(poly)
            lp[0] = (lp[0] - r_eval) % P
            inner.append(lp)
        l_i = lincomb(inner, y)
        contributions.append([c * z_i % P for c in l_i])
        z_diffs.append(z_i)
    l_x = lincomb(contributions, v)
    zt_eval = evaluate_vanishing_polynomial(super_points, u)
    l_x = [(a - b * zt_eval) % P for a, b in zip(l_x, h_x)]
    assert eval_poly(l_x, u) == 0, "linearisation polynomial does not vanish at u"
    final = kate_division(l_x, u)
    z0_inv = pow(z_diffs[0], P - 2, P)
    final = [c * z0_inv % P for c in final]
    h2 = commit(final)
    return h1, h2, h_x, final


# ---------------------------------------------------------------------------------------------
# plonk::lookup::prover::permute_expression_pair (the sort behind the permuted lookup columns), restated step by
# step from the published upstream algorithm; the blinding rows (prover RNG) are left out.
# ---------------------------------------------------------------------------------------------
def permute_expression_pair(input_expression, table_expression, usable_rows: int):
    """Returns (permuted_input, permuted_table) over the first usable_rows rows, or raises ValueError where upstream
    returns Error::ConstraintSystemFailure."""
    permuted_input = sorted(v % P for v in input_expression[:usable_rows])
    leftover = {}
    for v in table_expression[:usable_rows]:
        leftover[v % P] = leftover.get(v % P, 0) + 1
    permuted_table = [0] * usable_rows
    repeated_rows = []
    for row, value in enumerate(permuted_input):
        if row == 0 or value != permuted_input[row - 1]:
            permuted_table[row] = value
            if leftover.get(value, 0) > 0:
                leftover[value] -= 1
            else:
                raise ValueError("input value not in table")
        else:
            repeated_rows.append(row)
    for value in sorted(leftover):            # BTreeMap iteration: ascending
        for _ in range(leftover[value]):
            permuted_table[repeated_rows.pop()] = value
    assert not repeated_rows
    return permuted_input, permuted_table


# ---------------------------------------------------------------------------------------------
# plonk::keygen::keygen_pk pieces: permutation polynomials from the cycle mapping, Lagrange selectors
# (restated from the published upstream algorithm; reached from the reference at
# /root/reference/zkevm-circuits/benches/benchmark_list/main.rs:121-138)
# ---------------------------------------------------------------------------------------------
def permutation_sigmas(omega: int, mapping):
    """mapping[c][r] = (column, row) the cycles send cell (c, r) to; returns sigma[c][r] = delta^column * omega^row."""
    return [[pow(FR_DELTA, mc, P) * pow(omega, mr, P) % P for mc, mr in col] for col in mapping]


def lagrange_selectors(dom: "EvaluationDomain", blinding_factors: int):
    """(l0, l_last, l_active_row) on the extended domain, as keygen_pk builds them."""
    n = dom.n
    l0 = [0] * n
    l0[0] = 1
    l_blind = [0] * n
    for i in range(n - blinding_factors, n):
        l_blind[i] = 1
    l_last = [0] * n
    l_last[n - blinding_factors - 1] = 1
    ext = [dom.coeff_to_extended(dom.lagrange_to_coeff(v)) for v in (l0, l_blind, l_last)]
    l_active = [(1 - (a + b)) % P for a, b in zip(ext[2], ext[1])]
    return ext[0], ext[2], l_active


def g1_fft(points, omega: int, log_n: int):
    """arithmetic::best_fft over G1 points (None = identity): out[i] = sum_j omega^(i j) points[j], by definition."""
    from bn254 import g1_mul
    n = 1 << log_n
    out = []
    for i in range(n):
        acc = None
        for j, pt in enumerate(points):
            acc = g1_add(acc, g1_mul(pt, pow(omega, i * j, P)) if pt is not None else None)
        out.append(acc)
    return out
