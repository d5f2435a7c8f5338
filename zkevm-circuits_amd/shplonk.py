"""Host mirror of the SHPLONK opening prover over device-resident polynomials.

Follows `halo2_proofs::poly::kzg::multiopen::shplonk::{construct_intermediate_sets, prover::ProverSHPLONK::create_proof}`
(the multi-open scheme the reference proves with: /root/reference/zkevm-circuits/benches/benchmark_list/main.rs:26,267,
/root/reference/prover/src/test/proof_test.rs:297).  What stays on the host is what upstream does per rotation set on a
handful of field elements (set construction, Lagrange interpolation of the evaluations, vanishing-polynomial values) and
the transcript; every operation over 2^k coefficients runs on the device through the C ABI:

    sums of `poly * power` (quotient numerators, h(X), linearisation polynomial)  -> zkhip_poly_lincomb_dev
    div_by_vanishing / kate_division                                              -> zkhip_kate_division_dev
    params.commit                                                                 -> zkhip_msm_range_dev

The results are the same field elements upstream computes: every step is exact arithmetic in Fr, only the grouping of
additions differs.  Values on this interface are canonical integers; polynomials are device pointers to 2^k Montgomery
coefficients.
"""
from __future__ import annotations

import functools

import numpy as np

from .evalh_host import fr_to_mont_limbs
from .plonk import FR_MODULUS

P = FR_MODULUS


class Query:
    """One `ProverQuery`: polynomial `key` (any hashable identity, e.g. its index), opened at `point` with value `eval`."""

    def __init__(self, key, point: int, eval: int):
        self.key, self.point, self.eval = key, point % P, eval % P


def _inv(x: int) -> int:
    return pow(x, P - 2, P)


@functools.lru_cache(maxsize=256)
def _lagrange_basis(points):
    """Coefficients of the Lagrange basis polynomials of a point tuple, each already divided by its denominator: every
    polynomial opened at the same rotation set shares them."""
    n = len(points)
    basis = []
    for j in range(n):
        num, den = [1], 1
        for k in range(n):
            if k == j:
                continue
            nxt = [0] * (len(num) + 1)
            for i, c in enumerate(num):          # num * (X - x_k)
                nxt[i + 1] = (nxt[i + 1] + c) % P
                nxt[i] = (nxt[i] - c * points[k]) % P
            num = nxt
            den = den * (points[j] - points[k]) % P
        inv = _inv(den)
        basis.append([c * inv % P for c in num])
    return basis


def lagrange_interpolate(points, evals):
    """`arithmetic::lagrange_interpolate`: coefficients, lowest first, of the polynomial of degree < len(points)."""
    n = len(points)
    if n == 1:
        return [evals[0] % P]
    basis = _lagrange_basis(tuple(points))
    return [sum(evals[j] * basis[j][i] for j in range(n)) % P for i in range(n)]


def evaluate_vanishing_polynomial(roots, z: int) -> int:
    acc = 1
    for r in roots:
        acc = acc * (z - r) % P
    return acc


def eval_small(coeffs, x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def construct_intermediate_sets(queries):
    """Groups queries as upstream does: commitments in order of first appearance, each with the ordered set of its
    points (BTreeSet<Fr>: ascending canonical value); point sets in order of first appearance, each with its commitments.
    Returns ([(points, [(key, evals)])], super_point_set)."""
    evals = {}
    for q in queries:
        evals.setdefault((q.key, q.point), q.eval)   # upstream's get_eval takes the first match
    commitment_map, index = [], {}
    super_points = set()
    for q in queries:
        super_points.add(q.point)
        if q.key in index:
            commitment_map[index[q.key]][1].add(q.point)
        else:
            index[q.key] = len(commitment_map)
            commitment_map.append((q.key, {q.point}))
    set_map, set_index = [], {}
    for key, pts in commitment_map:
        fs = frozenset(pts)
        if fs in set_index:
            set_map[set_index[fs]][1].append(key)
        else:
            set_index[fs] = len(set_map)
            set_map.append((fs, [key]))
    rotation_sets = []
    for fs, keys in set_map:
        pts = sorted(fs)
        rotation_sets.append((pts, [(key, [evals[(key, p)] for p in pts]) for key in keys]))
    return rotation_sets, sorted(super_points)


class ProverSHPLONK:
    """`ProverSHPLONK::create_proof` for polynomials resident on one device.

    ctx: ffi.Context; bases_id: the registered SRS `g` (monomial basis); n = 2^k.
    """

    def __init__(self, ctx, bases_id: int, n: int):
        self.ctx, self.bases_id, self.n = ctx, bases_id, n

    def _limbs(self, vals):
        return np.array([fr_to_mont_limbs(v) for v in vals], dtype=np.uint64).reshape(len(vals), 4)

    def _commit(self, d_poly: int):
        return self.ctx.msm_range_dev(self.bases_id, 0, d_poly, self.n)

    def create_proof(self, polys: dict, queries, transcript):
        """polys: {key: device pointer}; queries: iterable of Query; transcript: object with squeeze_challenge() -> int
        and write_point(jacobian limbs).  Returns the device pointers (h_x, final quotient), which the caller frees."""
        ctx, n = self.ctx, self.n
        queries = list(queries)
        y = transcript.squeeze_challenge() % P
        v = transcript.squeeze_challenge() % P
        rotation_sets, super_points = construct_intermediate_sets(queries)
        # low-degree equivalents r_ij(X) of every commitment over its set's points
        ext = [(pts, [(key, lagrange_interpolate(pts, evals)) for key, evals in cs]) for pts, cs in rotation_sets]

        scratch = [ctx.alloc(n * 32), ctx.alloc(n * 32)]
        d_quot = []
        for pts, cs in ext:
            # N_i(X) = sum_j y^j (P_ij(X) - R_ij(X)): the P part on the device, the low-degree part folded on the host
            ptrs, lens, coeffs = [], [], []
            low = [0] * len(pts)
            power = 1
            for key, r in cs:
                ptrs.append(polys[key])
                lens.append(n)
                coeffs.append(power)
                for i, c in enumerate(r):
                    low[i] = (low[i] - power * c) % P
                power = power * y % P
            d_low = ctx.upload(self._limbs(low))
            ptrs.append(d_low)
            lens.append(len(low))
            coeffs.append(1)
            cur, other = scratch
            ctx.poly_lincomb_dev(ptrs, lens, self._limbs(coeffs), n, cur)
            # Q_i(X) = N_i(X) / prod (X - point): one division per point, the length shrinks by one each time
            length = n
            for p in pts:
                ctx.kate_division_dev(cur, length, fr_to_mont_limbs(p), other)
                cur, other = other, cur
                length -= 1
            d_q = ctx.alloc(n * 32)
            # resize(n, 0): copy `length` coefficients, zero the rest (a one-term combination with coefficient one)
            ctx.poly_lincomb_dev([cur], [length], self._limbs([1]), n, d_q)
            d_quot.append(d_q)
            ctx.sync()
            ctx.free(d_low)
        # h(X) = sum_i v^i Q_i(X)
        d_h = ctx.alloc(n * 32)
        powers_v = [pow(v, i, P) for i in range(len(d_quot))]
        ctx.poly_lincomb_dev(d_quot, [n] * len(d_quot), self._limbs(powers_v), n, d_h)
        ctx.sync()
        for d in d_quot:
            ctx.free(d)
        transcript.write_point(self._commit(d_h))
        u = transcript.squeeze_challenge() % P

        # L(X) = sum_i v^i z_i sum_j y^j (P_ij(X) - r_ij(u)) - Z_T(u) h(X), one combination over all polynomials
        ptrs, lens, coeffs = [], [], []
        constant = 0
        z_diffs = []
        for i, (pts, cs) in enumerate(ext):
            diffs = [p for p in super_points if p not in pts]
            z_i = evaluate_vanishing_polynomial(diffs, u)
            z_diffs.append(z_i)
            outer = powers_v[i] * z_i % P
            power = 1
            for key, r in cs:
                c = outer * power % P
                ptrs.append(polys[key])
                lens.append(n)
                coeffs.append(c)
                constant = (constant - c * eval_small(r, u)) % P
                power = power * y % P
        zt_eval = evaluate_vanishing_polynomial(super_points, u)
        ptrs.append(d_h)
        lens.append(n)
        coeffs.append((-zt_eval) % P)
        d_const = ctx.upload(self._limbs([constant]))
        ptrs.append(d_const)
        lens.append(1)
        coeffs.append(1)
        d_l, d_final = scratch
        ctx.poly_lincomb_dev(ptrs, lens, self._limbs(coeffs), n, d_l)
        # upstream asserts L(u) = 0 in debug builds; here always, the division below drops the remainder silently
        rem = ctx.eval_polynomial_dev(d_l, n, fr_to_mont_limbs(u))
        if rem.any():
            raise ValueError("shplonk: the linearisation polynomial does not vanish at u (inconsistent evaluations)")
        ctx.kate_division_dev(d_l, n, fr_to_mont_limbs(u), d_final)
        # normalise by the first set's difference value; resize to n coefficients for the commitment
        d_out = ctx.alloc(n * 32)
        ctx.poly_lincomb_dev([d_final], [n - 1], self._limbs([_inv(z_diffs[0])]), n, d_out)
        transcript.write_point(self._commit(d_out))
        ctx.sync()
        for d in scratch + [d_const]:
            ctx.free(d)
        return d_h, d_out
