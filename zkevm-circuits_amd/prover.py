"""Host mirror of `halo2_proofs::plonk::{keygen::{keygen_vk, keygen_pk}, prover::create_proof}` over the C ABI.

The reference proves with
`create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<Bn256>, Challenge255<_>, _, Blake2bWrite<_,_,_>, C>(&params,
&pk, &[circuit], &[&[&instance]], rng, &mut transcript, ..)` (/root/reference/prover/src/test/proof_test.rs:297-306,
/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:265-281).  This module runs the same phases in the same
order for one circuit whose witness is already assigned, with every operation over 2^k or 2^extended_k field elements
on the device: commitments (`zkhip_msm_batch_dev`), lookup compression and the permuted columns, the grand products,
`coeff_to_extended`, `evaluate_h`, the division by the vanishing polynomial, evaluations at x and the SHPLONK multi-open.
What stays on the host is what upstream does on single field elements: the transcript, the RNG and the bookkeeping.

Where bytes, not field elements, are decided, upstream's rules are restated here; none of them can be pinned (the
reference holds no proof bytes, no key file and no halo2 source), and none touches the device work:
  * the verifying key's `transcript_repr` is upstream's: Blake2b-512, personalised "Halo2-Verify-Key", over the length and
    the text of the `{:?}` rendering of the pinned verifying key -- moduli, domain, the whole constraint system (columns,
    phases, gates, queries, permutation columns, lookups), the fixed and permutation commitments (`pinned_vk_debug`);
    the rendering follows the derive(Debug) layout of halo2_proofs 0.3.0 as published, and a fork that adds or renames a
    field changes every byte of it;
  * the random polynomial of the vanishing argument: by default one draw of 2^k residues from the RNG (`rng.limbs`);
    `create_proof(..., random_poly_threads=T)` draws it as upstream does, chunk-parallel from T (+1) ChaCha20 streams
    each seeded with 32 bytes of the main RNG, element i of a chunk = `Fr::random` of that chunk's stream, so that the
    polynomial depends on the thread count exactly as upstream's does;
  * the order of the column queries: the lists `ConstraintSystem.{advice,fixed,instance}_queries` when the caller sets
    them (a circuit's order of first `query_*` / `enable_equality` calls), else first appearance in gates, lookups and
    permutation columns;
  * compressed points follow halo2curves 0.6.1 as published (little-endian x, sign of y in bit 6 and identity in bit 7
    of the last byte).
What cannot be restated: the fork's own additions to these formats (its `create_proof` takes a seventh argument and a
`tracing::Trace`; whether its key rendering differs from 0.3.0 is unknown).
"""
from __future__ import annotations

import hashlib

import numpy as np

from . import evalh_host, ffi, keygen as kg, sharding, shplonk
from .evalh_host import fr_to_mont_limbs
from .plonk import ANY_ADVICE, ANY_FIXED, ANY_INSTANCE, FR_MODULUS, ConstraintSystem, Evaluator

P = FR_MODULUS
FQ_MODULUS = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
FR_DELTA = pow(7, 1 << 28, P)
_FR_RINV = pow(1 << 256, P - 2, P)
_FQ_RINV = pow(1 << 256, FQ_MODULUS - 2, FQ_MODULUS)


def fr_from_mont_limbs(limbs) -> int:
    return sum(int(w) << (64 * j) for j, w in enumerate(limbs)) * _FR_RINV % P


def fq_from_mont_limbs(limbs) -> int:
    return sum(int(w) << (64 * j) for j, w in enumerate(limbs)) * _FQ_RINV % FQ_MODULUS


def fr_limbs(vals) -> np.ndarray:
    """canonical integers -> (len, 4) Montgomery limbs"""
    r = evalh_host.R256
    buf = b"".join((v % P * r % P).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def compress_point(xy) -> bytes:
    """`G1Affine::to_bytes`: xy = (x, y) canonical integers, None for the identity."""
    if xy is None:
        b = bytearray(32)
        b[31] |= 0x80
        return bytes(b)
    b = bytearray(xy[0].to_bytes(32, "little"))
    b[31] |= (xy[1] & 1) << 6
    return bytes(b)


class Blake2bWrite:
    """`transcript::Blake2bWrite<Vec<u8>, G1Affine, Challenge255<_>>`."""

    PREFIX_CHALLENGE, PREFIX_POINT, PREFIX_SCALAR = b"\x00", b"\x01", b"\x02"

    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.out = bytearray()

    def squeeze_challenge(self) -> int:
        self.state.update(self.PREFIX_CHALLENGE)
        return int.from_bytes(self.state.copy().digest(), "little") % P     # Challenge255: from_uniform_bytes

    def common_point(self, xy):
        if xy is None:
            raise ValueError("cannot write points at infinity to the transcript")
        self.state.update(self.PREFIX_POINT)
        self.state.update(xy[0].to_bytes(32, "little"))
        self.state.update(xy[1].to_bytes(32, "little"))

    def common_scalar(self, v: int):
        self.state.update(self.PREFIX_SCALAR)
        self.state.update((v % P).to_bytes(32, "little"))

    def write_point(self, point):
        """point: Jacobian (12) or affine (8) Montgomery limbs as the C ABI returns them."""
        xy = affine_ints(point)
        self.common_point(xy)
        self.out += compress_point(xy)

    def write_scalar(self, v: int):
        self.common_scalar(v)
        self.out += (v % P).to_bytes(32, "little")

    def finalize(self) -> bytes:
        return bytes(self.out)


def affine_ints(point):
    """Jacobian (12 limbs) or affine (8 limbs) point -> (x, y) canonical integers, None for the identity."""
    point = np.asarray(point, dtype=np.uint64).reshape(-1)
    if point.size == 12:
        out = np.zeros(8, dtype=np.uint64)
        rc = ffi.lib().zkhip_g1_to_affine(ffi._p(np.ascontiguousarray(point)), ffi._p(out))
        if rc != 0:
            raise ffi.ZkhipError(f"zkhip_g1_to_affine: status {rc}")
        point = out
    if not point.any():
        return None
    return fq_from_mont_limbs(point[:4]), fq_from_mont_limbs(point[4:])


class FieldRng:
    """`Fr::random(rng)` stand-in: 64 uniform bytes reduced mod r, drawn from a seeded numpy generator."""

    def __init__(self, seed: int):
        self.g = np.random.default_rng(seed)

    def fr(self) -> int:
        return int.from_bytes(self.g.bytes(64), "little") % P

    def fr_many(self, count: int) -> list:
        """the next `count` draws of `fr` (the generator hands out the same bytes in one call as in `count` calls)"""
        raw = self.g.bytes(64 * count)
        return [int.from_bytes(raw[64 * i:64 * i + 64], "little") % P for i in range(count)]

    def limbs(self, n: int) -> np.ndarray:
        """n field elements at once as (n, 4) Montgomery limbs: residues uniform below 2^252 (< r), which is as good a
        stand-in for a polynomial with random coefficients and keeps 2^k draws out of the interpreter"""
        a = self.g.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
        return a

    def fill_bytes(self, count: int) -> bytes:
        """`RngCore::fill_bytes`: the next `count` bytes of the stream"""
        return self.g.bytes(count)


class OsFieldRng:
    """The prover's RNG for real proofs: `Fr::random(OsRng)` as the reference draws it
    (/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:277): 64 bytes of operating-system entropy reduced mod r
    per element (`from_uniform_bytes`), so every blinding value is uniform to within 2^-258.  `FieldRng` is for
    reproducible tests and benchmarks only: its generator is not cryptographic and its `limbs` are not uniform below r."""

    def fr(self) -> int:
        import secrets
        return int.from_bytes(secrets.token_bytes(64), "little") % P

    def fr_many(self, count: int) -> list:
        import secrets
        raw = secrets.token_bytes(64 * count)
        return [int.from_bytes(raw[64 * i:64 * i + 64], "little") % P for i in range(count)]

    def fill_bytes(self, count: int) -> bytes:
        import secrets
        return secrets.token_bytes(count)

    def limbs(self, n: int) -> np.ndarray:
        import secrets
        raw = np.frombuffer(secrets.token_bytes(64 * n), dtype=np.uint8).reshape(n, 64)
        r = evalh_host.R256
        # wide reduction of every 64-byte draw, then Montgomery form
        out = np.empty((n, 4), dtype=np.uint64)
        for i in range(n):
            v = int.from_bytes(raw[i].tobytes(), "little") % P * r % P
            out[i] = [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
        return out


def collect_queries(cs: ConstraintSystem):
    """(advice_queries, fixed_queries, instance_queries) as [(column, rotation)], each in order of first appearance.
    Upstream's ConstraintSystem carries these lists as fields filled while the circuit is configured; here they are
    derived once per constraint system and kept on it."""
    return cs._memo("queries", lambda: _collect_queries(cs))


def _collect_queries(cs: ConstraintSystem):
    derived = _derive_queries(cs)
    given = (cs.advice_queries, cs.fixed_queries, cs.instance_queries)
    out = []
    for name, have, want in zip(("advice", "fixed", "instance"), derived, given):
        if want is None:
            out.append(have)
            continue
        want = [tuple(q) for q in want]
        if len(set(want)) != len(want) or not set(have) <= set(want):
            raise ValueError(f"{name}_queries must list every (column, rotation) the expressions query, once")
        out.append(want)
    return tuple(out)


def _derive_queries(cs: ConstraintSystem):
    out = {"advice": {}, "fixed": {}, "instance": {}}     # insertion-ordered sets

    def add(kind, q):
        out[kind].setdefault(q, None)

    def walk(e):
        t = e[0]
        if t in out:
            add(t, (e[1], e[2]))
        elif t in ("neg", "scaled"):
            walk(e[1])
        elif t in ("sum", "prod"):
            walk(e[1])
            walk(e[2])

    for gate in cs.gates:
        for poly in gate:
            walk(poly)
    for lk in cs.lookups:
        for e in lk.input_expressions + lk.table_expressions:
            walk(e)
    names = {ANY_ADVICE: "advice", ANY_FIXED: "fixed", ANY_INSTANCE: "instance"}
    for kind, idx in cs.permutation_columns:
        add(names[kind], (idx, 0))
    return list(out["advice"]), list(out["fixed"]), list(out["instance"])


class ParamsKZG:
    """`ParamsKZG<Bn256>` on one device: the monomial-basis SRS `g` and its Lagrange form, both registered for MSM."""

    def __init__(self, ctx, dom, g_affine):
        self.ctx, self.k, self.n = ctx, dom.k, dom.n
        g_affine = np.ascontiguousarray(g_affine, dtype=np.uint64).reshape(self.n, 8)
        self.g = ctx.register_bases(g_affine)
        self.g_lagrange = ctx.register_bases(kg.g_to_lagrange(ctx, dom, g_affine))

    @classmethod
    def from_registered(cls, ctx, dom, g: int, g_lagrange: int):
        """over tables that are registered already (the caller releases them; `close` must not be called)"""
        self = cls.__new__(cls)
        self.ctx, self.k, self.n, self.g, self.g_lagrange = ctx, dom.k, dom.n, g, g_lagrange
        return self

    def _commit(self, bid, d_polys, dist, owners=None):
        d_polys = list(d_polys)
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return self.ctx.msm_batch_dev(bid, d_polys, self.n)
        # columns are independent: every rank commits the columns it owns (contiguous blocks, `sharding.column_blocks`,
        # unless the caller names the owner of every column: then only the owner needs to hold the column at all); the
        # points are gathered in column order
        if owners is None:
            owners = sharding.block_owner(len(d_polys), dist.get_world_size())
        mine = [c for c, r in enumerate(owners) if r == dist.get_rank()]
        local = self.ctx.msm_batch_dev(bid, [d_polys[c] for c in mine], self.n) if mine else []
        return sharding.gather_commitments(dict(zip(mine, local)), len(d_polys), dist)

    def commit(self, d_polys, dist=None, owners=None):
        return self._commit(self.g, d_polys, dist, owners)

    def commit_lagrange(self, d_polys, dist=None, owners=None):
        return self._commit(self.g_lagrange, d_polys, dist, owners)

    def close(self):
        self.ctx.release_bases(self.g)
        self.ctx.release_bases(self.g_lagrange)


class ProvingKey:
    """`ProvingKey<G1Affine>`: the verifying key's commitments plus the device-resident key polynomials."""

    def __init__(self):
        self.cs = None
        self.polys = None                     # keygen.ProvingKeyPolys
        self.fixed_values = []                # device pointers, Lagrange basis
        self.fixed_commitments, self.permutation_commitments = [], []      # (x, y) integers
        self.transcript_repr = 0
        self.ev = None                        # plonk.Evaluator, built once per key as upstream's `ProvingKey::ev`
        self.split_programs = None            # multiplications per row of the low / high / whole evaluate_h programs
        self.high_degree_columns = None       # split evaluation: `evalh_host.columns_used` of the terms above the bound

    def pointers(self):
        return self.polys.pointers() + self.fixed_values


SPLIT_MIN_LOW_SHARE = 0.15      # the half-domain evaluation is used when the low-degree terms hold this share of the products
                                # (measured: a share of 0.13 at four cosets loses 2 %, 0.18 at eight gains 6.5 %, 0.35 gains 17 %)


def keygen(ctx, dom, params: ParamsKZG, cs: ConstraintSystem, fixed, assembly, dist=None, low_degree_split=None) -> ProvingKey:
    """`keygen_vk` + `keygen_pk` for assigned fixed columns (lists of n canonical integers) and copy constraints.
    dist: an initialised torch.distributed module with one rank per GPU; every rank then keeps the key's extended vectors
    only on the cosets it evaluates in `create_proof(..., dist=dist)` (1 / world of them up to 2^e ranks), and such a key
    proves only with the same process group.
    low_degree_split (single GPU): evaluate the terms of the quotient numerator whose degree is at most half the number of
    cosets on the even-indexed half of the extended domain only (`evalh_host.evaluate_h_split`); the key's extended vectors
    are then stored split by row parity.  None decides from the constraint system: on when those terms hold at least
    SPLIT_MIN_LOW_SHARE of the program's multiplications.  The proof bytes do not depend on it."""
    n = dom.n
    if n < cs.minimum_rows():
        raise ValueError("not enough rows available")
    if len(fixed) != cs.num_fixed_columns or any(len(c) != n for c in fixed):
        raise ValueError("fixed columns do not match the constraint system")
    pk = ProvingKey()
    pk.cs = cs
    pk.ev = Evaluator(cs)
    evalh_host.prepare(dom, cs, pk.ev)       # the device program of evaluate_h, built once per key like `pk.ev`
    fixed_limbs = [fr_limbs(c) for c in fixed]
    own = None
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        part = sharding.coset_partition(dom.extended_k, dom.k, dist.get_world_size())[dist.get_rank()]
        own = part[0] if part is not None else []
    split = False
    if own is None and dom.extended_k > dom.k and low_degree_split is not False:
        low, high, whole = evalh_host.split_gain(dom, cs, pk.ev)
        pk.split_programs = {"low": low, "high": high, "whole": whole}
        # the split evaluation holds every extended coset at once (no coset groups): it is chosen by itself only where
        # the key's and the proof's extended vectors and the transform scratch of a batch fit the free memory
        vectors = (cs.num_fixed_columns + len(cs.permutation_columns) + 3 + cs.num_advice_columns + cs.num_instance_columns
                   + 3 * len(cs.lookups) + cs.num_permutation_sets() + 3 * 64 + 4)
        fits = vectors * dom.extended_len * 32 <= 0.8 * ctx.mem_info()[0]
        split = True if low_degree_split else (fits and whole > 0 and low >= SPLIT_MIN_LOW_SHARE * whole)
    pk.polys = kg.keygen_polys(ctx, dom, fixed_limbs, assembly if cs.permutation_columns else None, cs.blinding_factors(), own,
                               split=split)
    if split:
        evalh_host.warm_up_split(ctx, dom)
        pk.high_degree_columns = evalh_host.columns_used(dom, cs, pk.ev, (ffi.TERMS_HIGH, (dom.extended_len // dom.n) // 2))
    pk.fixed_values = [ctx.upload(a) for a in fixed_limbs]
    if pk.fixed_values:
        pk.fixed_commitments = [affine_ints(j) for j in params.commit_lagrange(pk.fixed_values)]
    if pk.polys.permutations:
        pk.permutation_commitments = [affine_ints(j) for j in params.commit_lagrange(pk.polys.permutations)]
    pk.transcript_repr = _vk_digest(cs, dom.k, dom.extended_k, fr_from_mont_limbs(dom.constant(0)), pk.fixed_commitments,
                                    pk.permutation_commitments)
    return pk


def _point_limbs(xy) -> np.ndarray:
    """(x, y) canonical integers (None: identity) -> 8 Montgomery limbs, the RawBytes form of a G1Affine"""
    if xy is None:
        return np.zeros(8, dtype=np.uint64)
    r = 1 << 256
    words = []
    for c in xy:
        m = c * r % FQ_MODULUS
        words += [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    return np.array(words, dtype=np.uint64)


def _hex_fe(v: int) -> str:
    """`{:?}` of a halo2curves field element: 0x and the canonical value, 64 hex digits, most significant first"""
    return "0x" + format(v, "064x")


def _debug_expr(e, cs: ConstraintSystem, index) -> str:
    """`{:?}` of a `plonk::Expression` (derive(Debug) over the enum's tuple and struct variants)"""
    t = e[0]
    if t == "const":
        return f"Constant({_hex_fe(e[1])})"
    if t in ("fixed", "advice", "instance"):
        q = index[t][(e[1], e[2])]
        name = {"fixed": "Fixed", "advice": "Advice", "instance": "Instance"}[t]
        phase = ""
        if t == "advice":
            ph = (cs.advice_column_phase or [0] * cs.num_advice_columns)[e[1]]
            phase = f", phase: Phase({ph})"
        return f"{name} {{ query_index: {q}, column_index: {e[1]}, rotation: Rotation({e[2]}){phase} }}"
    if t == "challenge":
        ph = (cs.challenge_phase or [0] * cs.num_challenges)[e[1]]
        return f"Challenge(Challenge {{ index: {e[1]}, phase: Phase({ph}) }})"
    if t == "neg":
        return f"Negated({_debug_expr(e[1], cs, index)})"
    if t == "sum":
        return f"Sum({_debug_expr(e[1], cs, index)}, {_debug_expr(e[2], cs, index)})"
    if t == "prod":
        return f"Product({_debug_expr(e[1], cs, index)}, {_debug_expr(e[2], cs, index)})"
    if t == "scaled":
        return f"Scaled({_debug_expr(e[1], cs, index)}, {_hex_fe(e[2])})"
    raise ValueError(t)


def pinned_vk_debug(cs: ConstraintSystem, k: int, extended_k: int, omega: int, fixed_commitments, permutation_commitments) -> str:
    """The text upstream hashes into `transcript_repr`: `format!("{:?}", vk.pinned())` of halo2_proofs 0.3.0 --
    `PinnedVerificationKey { base_modulus, scalar_modulus, domain: PinnedEvaluationDomain { k, extended_k, omega }, cs:
    PinnedConstraintSystem { .. }, fixed_commitments, permutation: VerifyingKey { commitments } }`.  Restated from the
    published derive(Debug) layout; unpinned (no key or proof bytes of the reference exist to check it against)."""
    adv_q, fix_q, inst_q = collect_queries(cs)
    index = {"advice": {q: i for i, q in enumerate(adv_q)}, "fixed": {q: i for i, q in enumerate(fix_q)},
             "instance": {q: i for i, q in enumerate(inst_q)}}
    col = lambda kind, i: f"Column {{ index: {i}, column_type: {kind} }}"                      # noqa: E731
    queries = lambda kind, qs: "[" + ", ".join(f"({col(kind, c)}, Rotation({r}))" for c, r in qs) + "]"   # noqa: E731
    kind_name = {ANY_ADVICE: "Advice", ANY_FIXED: "Fixed", ANY_INSTANCE: "Instance"}
    point = lambda p: "Infinity" if p is None else f"({_hex_fe(p[0])}, {_hex_fe(p[1])})"       # noqa: E731
    phases = cs.advice_column_phase or [0] * cs.num_advice_columns
    gates = ", ".join(_debug_expr(poly, cs, index) for gate in cs.gates for poly in gate)
    lookups = ", ".join(
        "Argument { input_expressions: [" + ", ".join(_debug_expr(e, cs, index) for e in lk.input_expressions)
        + "], table_expressions: [" + ", ".join(_debug_expr(e, cs, index) for e in lk.table_expressions) + "] }"
        for lk in cs.lookups)
    pinned_cs = (
        f"PinnedConstraintSystem {{ num_fixed_columns: {cs.num_fixed_columns}, num_advice_columns: {cs.num_advice_columns}, "
        f"num_instance_columns: {cs.num_instance_columns}, num_selectors: {cs.num_selectors}, "
        f"num_challenges: {cs.num_challenges}, advice_column_phase: [{', '.join(str(p) for p in phases)}], "
        f"challenge_phase: [{', '.join(str(p) for p in (cs.challenge_phase or [0] * cs.num_challenges))}], "
        f"gates: [{gates}], advice_queries: {queries('Advice', adv_q)}, instance_queries: {queries('Instance', inst_q)}, "
        f"fixed_queries: {queries('Fixed', fix_q)}, permutation: Argument {{ columns: ["
        + ", ".join(col(kind_name[kind], i) for kind, i in cs.permutation_columns)
        + f"] }}, lookups: [{lookups}], constants: [], minimum_degree: "
        + ("None" if cs.minimum_degree is None else f"Some({cs.minimum_degree})") + " }")
    return (f"PinnedVerificationKey {{ base_modulus: \"{_hex_fe(FQ_MODULUS)}\", scalar_modulus: \"{_hex_fe(P)}\", "
            f"domain: PinnedEvaluationDomain {{ k: {k}, extended_k: {extended_k}, omega: {_hex_fe(omega)} }}, cs: {pinned_cs}, "
            f"fixed_commitments: [{', '.join(point(c) for c in fixed_commitments)}], "
            f"permutation: VerifyingKey {{ commitments: [{', '.join(point(c) for c in permutation_commitments)}] }} }}")


def _vk_digest(cs: ConstraintSystem, k: int, extended_k: int, omega: int, fixed_commitments, permutation_commitments) -> int:
    """`VerifyingKey::from_parts`: transcript_repr = from_uniform_bytes(Blake2b-512("Halo2-Verify-Key", len as u64 LE || text))
    over the `{:?}` text of the pinned key -- the digest binds the whole constraint system, not only the commitments."""
    text = pinned_vk_debug(cs, k, extended_k, omega, fixed_commitments, permutation_commitments).encode()
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(len(text).to_bytes(8, "little"))
    h.update(text)
    return int.from_bytes(h.digest(), "little") % P


# ---- ChaCha20 (the `rand_chacha::ChaCha20Rng` keystream: 64-bit block counter from 0, zero stream id) ------------------
def chacha20_blocks(seed: bytes, first_block: int, count: int) -> np.ndarray:
    """(count, 16) uint32 keystream words of blocks first_block .. first_block + count - 1"""
    assert len(seed) == 32
    st = np.zeros((count, 16), dtype=np.uint32)
    st[:, 0:4] = np.frombuffer(b"expand 32-byte k", dtype="<u4")
    st[:, 4:12] = np.frombuffer(seed, dtype="<u4")
    ctr = np.arange(first_block, first_block + count, dtype=np.uint64)
    st[:, 12] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    st[:, 13] = (ctr >> np.uint64(32)).astype(np.uint32)
    x = st.copy()

    def rotl(v, c):
        return (v << np.uint32(c)) | (v >> np.uint32(32 - c))

    def quarter(a, b, c, d):
        x[:, a] += x[:, b]; x[:, d] = rotl(x[:, d] ^ x[:, a], 16)      # noqa: E702
        x[:, c] += x[:, d]; x[:, b] = rotl(x[:, b] ^ x[:, c], 12)      # noqa: E702
        x[:, a] += x[:, b]; x[:, d] = rotl(x[:, d] ^ x[:, a], 8)       # noqa: E702
        x[:, c] += x[:, d]; x[:, b] = rotl(x[:, b] ^ x[:, c], 7)       # noqa: E702
    for _ in range(10):
        quarter(0, 4, 8, 12); quarter(1, 5, 9, 13); quarter(2, 6, 10, 14); quarter(3, 7, 11, 15)      # noqa: E702
        quarter(0, 5, 10, 15); quarter(1, 6, 11, 12); quarter(2, 7, 8, 13); quarter(3, 4, 9, 14)      # noqa: E702
    return x + st


def random_poly_per_thread_streams(rng, n: int, threads: int) -> np.ndarray:
    """The random polynomial of `vanishing::Argument::commit` as upstream draws it: n_chunks = threads (+ 1 when threads does
    not divide n) ChaCha20 generators, each seeded with 32 bytes from the main RNG in order; chunk j (n / threads
    coefficients) is filled from generator j, one `Fr::random` per coefficient = the next eight 64-bit words of the stream
    (one 64-byte block) read as a little-endian 512-bit integer and reduced mod r.  Returns (n, 4) Montgomery limbs."""
    if threads < 1:
        raise ValueError("threads must be positive")
    chunk = max(1, n // threads)
    n_chunks = threads + (1 if n % threads else 0)
    seeds = [rng.fill_bytes(32) for _ in range(n_chunks)]
    out = []
    for j in range(n_chunks):
        count = min(chunk, n - j * chunk)
        if count <= 0:
            break
        words = chacha20_blocks(seeds[j], 0, count)
        raw = np.ascontiguousarray(words).view(np.uint8).reshape(count, 64)
        out.append(fr_limbs([int.from_bytes(raw[i].tobytes(), "little") % P for i in range(count)]))
    return np.concatenate(out)


def save_proving_key(ctx, dom, pk: ProvingKey, path: str, layout: str = "v3") -> None:
    """`ProvingKey::write(_, SerdeFormat::RawBytes)` (/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:456-461):
    downloads the key's polynomials and writes them behind the verifying key (`formats.write_proving_key_raw`)."""
    from . import formats
    if pk.polys.cosets is not None:
        raise ValueError("a key generated for a multi-GPU partition holds only its rank's cosets")
    n, en = dom.n, dom.extended_len
    get = lambda ptrs, rows: [ctx.download(p, (rows, 4)) for p in ptrs]
    if pk.polys.split:
        # the file holds extended vectors in extended order: undo the split by row parity on the way out
        m = en // n
        d_tmp = ctx.alloc(en * 32)

        def natural(p):
            dom.cosets_to_extended_dev(p, list(range(0, m, 2)), d_tmp)
            dom.cosets_to_extended_dev(p + en * 16, list(range(1, m, 2)), d_tmp)
            return ctx.download(d_tmp, (en, 4))
        ext = lambda ptrs: [natural(p) for p in ptrs]
    else:
        natural = lambda p: ctx.download(p, (en, 4))
        ext = lambda ptrs: get(ptrs, en)
    vk = formats.VerifyingKeyRaw(dom.k, [_point_limbs(c) for c in pk.fixed_commitments] or np.zeros((0, 8), np.uint64),
                                 [_point_limbs(c) for c in pk.permutation_commitments] or np.zeros((0, 8), np.uint64))
    raw = formats.ProvingKeyRaw(
        vk, l0=natural(pk.polys.l0), l_last=natural(pk.polys.l_last),
        l_active_row=natural(pk.polys.l_active_row), fixed_values=get(pk.fixed_values, n),
        fixed_polys=get(pk.polys.fixed_polys, n), fixed_cosets=ext(pk.polys.fixed_cosets),
        permutations=get(pk.polys.permutations, n), permutation_polys=get(pk.polys.permutation_polys, n),
        permutation_cosets=ext(pk.polys.permutation_cosets))
    if pk.polys.split:
        ctx.free(d_tmp)
    formats.write_proving_key_raw(path, raw, layout)


def load_proving_key(ctx, dom, cs: ConstraintSystem, path: str, num_selectors: int = 0) -> ProvingKey:
    """`ProvingKey::read::<_, Circuit>(_, SerdeFormat::RawBytes)` (/root/reference/zkevm-circuits/benches/benchmark_list/main.rs:
    486-501): as upstream, the constraint system comes from the circuit (`cs`), the file supplies commitments and
    polynomials, which are uploaded as they are.  The result proves like a key from `keygen`.
    num_selectors: the circuit's selector count.  A key written upstream carries that many packed selector bit vectors
    after the permutation commitments (the constraint system here has its selectors as fixed columns already, so it cannot
    say how many the circuit declared); `save_proving_key` writes none.  Reading keys the reference wrote is unpinned:
    the container layouts restate published halo2 and no such file exists to check them against."""
    from . import formats
    raw = formats.read_proving_key_raw(path, len(cs.permutation_columns), num_selectors)
    if raw.vk.k != dom.k:
        raise ValueError(f"the key is for k = {raw.vk.k}, the domain has k = {dom.k}")
    raw.check(dom.extended_k)
    if len(raw.fixed_values) != cs.num_fixed_columns:
        raise ValueError("the key's fixed columns do not match the constraint system")
    pk = ProvingKey()
    pk.cs = cs
    pk.ev = Evaluator(cs)
    evalh_host.prepare(dom, cs, pk.ev)
    up = lambda arrs: [ctx.upload(a) for a in arrs]
    pk.polys = kg.ProvingKeyPolys()
    pk.polys.l0, pk.polys.l_last, pk.polys.l_active_row = (ctx.upload(a) for a in (raw.l0, raw.l_last, raw.l_active_row))
    pk.fixed_values = up(raw.fixed_values)
    pk.polys.fixed_polys, pk.polys.fixed_cosets = up(raw.fixed_polys), up(raw.fixed_cosets)
    pk.polys.permutations, pk.polys.permutation_polys = up(raw.permutations), up(raw.permutation_polys)
    pk.polys.permutation_cosets = up(raw.permutation_cosets)
    pk.fixed_commitments = [affine_ints(c) for c in raw.vk.fixed_commitments]
    pk.permutation_commitments = [affine_ints(c) for c in raw.vk.permutation_commitments]
    pk.transcript_repr = _vk_digest(cs, dom.k, dom.extended_k, fr_from_mont_limbs(dom.constant(0)), pk.fixed_commitments,
                                    pk.permutation_commitments)
    return pk


class DeviceColumn:
    """An advice column that already lives in device memory: `rows` Montgomery-limb elements at `ptr` (the usable rows; a
    witness generator that uploads columns while it produces them, or a prover that serves proofs back to back and sends
    the next witness over the host link while this proof runs, hands `create_proof` these instead of host arrays).  The
    memory stays the caller's: the prover copies it into its polynomial block."""

    def __init__(self, ptr: int, rows: int):
        self.ptr, self.rows = int(ptr), int(rows)

    def __len__(self):
        return self.rows


class ProofWorkspace:
    """The large device blocks of `create_proof` -- the polynomial block, the block of extended cosets (123 GB at the
    SuperCircuit's width), h -- kept between proofs of one proving key.  A prover that serves proof after proof passes the
    same workspace to every call: releasing a block of that size and asking for it again costs seconds in the driver
    (measured: 4.1 s of `hipMalloc` in the second of two consecutive stand-in proofs, against 1.7 s for the proof itself),
    which a first proof on fresh memory does not pay.  `close()` returns the blocks."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.blocks = {}          # (name, bytes) -> device pointer

    def take(self, name: str, nbytes: int) -> int:
        p = self.blocks.get((name, nbytes))
        if p is None:
            for key in [k for k in self.blocks if k[0] == name]:      # another circuit's size: make room first
                self.ctx.free(self.blocks.pop(key))
            p = self.blocks[(name, nbytes)] = self.ctx.alloc(nbytes)
        return p

    def close(self):
        for p in self.blocks.values():
            self.ctx.free(p)
        self.blocks = {}


def create_proof(ctx, dom, params: ParamsKZG, pk: ProvingKey, advice, instance, rng: FieldRng, transcript: Blake2bWrite,
                 timings: dict | None = None, dist=None, coset_groups: int | None = None,
                 random_poly_threads: int | None = None, workspace: ProofWorkspace | None = None):
    """One circuit.  advice: per column the values of the usable rows (a list of canonical integers, a
    (usable rows, 4) uint64 array of Montgomery limbs, or a `DeviceColumn`; the remaining rows are blinding), or a callable
    (phase, challenges so far) -> such a list, called once per phase of `cs.phases()` as upstream re-runs witness
    synthesis per phase (only the phase's own columns are read from it); instance: per column at most that many
    canonical integers.  Writes the proof into `transcript` and returns its bytes.  timings: filled with the wall
    seconds of each phase (the device is synchronised at every phase boundary, only when given).
    dist: an initialised torch.distributed module with one rank per GPU: every rank calls with the same arguments.  Sharded
    by column (contiguous blocks of columns per rank): the upload of the witness, every commitment, and -- by lookup -- the
    compression, the permuted columns, the grand product and the coefficient forms of a lookup's three polynomials;
    sharded by coset: the quotient numerator (zkevm_circuits_amd/sharding.py).  Exchanged: the uploaded advice columns and
    the lookups' polynomials (one all-gather per phase, 2^k-sized vectors), the commitments (96 bytes each), the
    pieces of h.  Replicated: the transcript and the RNG (every rank draws every value, so the streams stay in step), the
    permutation argument, the evaluations and the multi-open.  Every rank returns the same bytes as a single-rank call.
    A lookup input outside its table raises on the lookup's owner; the other ranks then fail in the next collective.
    coset_groups (single rank): evaluate the quotient numerator in that many passes over disjoint coset groups, which
    bounds the resident extended data to 1 / coset_groups of the whole; None picks the smallest count that fits the free
    device memory (1 on a whole MI355X at the SuperCircuit's width).  The proof bytes do not depend on it.
    random_poly_threads: draw the vanishing argument's random polynomial as upstream does for that many rayon threads
    (`random_poly_per_thread_streams`); None draws 2^k residues from `rng.limbs` in one go (the fast path of benchmarks).
    workspace: a `ProofWorkspace` kept by a caller that proves repeatedly: the large blocks then outlive the call."""
    cs = pk.cs
    n, en = dom.n, dom.extended_len
    bf = cs.blinding_factors()
    usable = n - (bf + 1)
    if (not callable(advice) and len(advice) != cs.num_advice_columns) or len(instance) != cs.num_instance_columns:
        raise ValueError("InvalidInstances / advice columns do not match the constraint system")
    if any(len(c) > usable for c in instance):
        raise ValueError("InstanceTooLarge")
    if not callable(advice) and any(len(c) != usable for c in advice):
        raise ValueError("an advice column holds the usable rows (integers, or (usable rows, 4) uint64 limbs)")
    omega = fr_from_mont_limbs(dom.constant(0))
    owned = []
    pinned_stages = []            # page-locked staging buffers, released when the proof is done

    def alloc(rows, keep_as=None):
        """keep_as: the name under which a caller's workspace keeps this block for the next proof"""
        if keep_as is not None and workspace is not None:
            return workspace.take(keep_as, rows * 32)
        p = ctx.alloc(rows * 32)
        owned.append(p)
        return p

    def limb(v):
        return np.array(fr_to_mont_limbs(v), dtype=np.uint64)

    import time
    clock = [time.perf_counter(), None]

    def mark(phase):
        """closes the running phase and opens `phase`"""
        if timings is None:
            return
        ctx.sync()
        now = time.perf_counter()
        if clock[1] is not None:
            timings[clock[1]] = timings.get(clock[1], 0.0) + now - clock[0]
        clock[0], clock[1] = now, phase

    def draws(count):
        """the next `count` values of the RNG, in the order single draws would come"""
        many = getattr(rng, "fr_many", None)
        return many(count) if many else [rng.fr() for _ in range(count)]

    def blind_rows(d_col, first_row, own=True):
        """the blinding rows of a column from the RNG; a rank that does not hold the column still draws them"""
        vals = draws(n - first_row)
        if own:
            ctx.upload(fr_limbs(vals), d_col + first_row * 32)

    try:
        # ---- instances and advice ---------------------------------------------------------------------------
        mark("advice")
        transcript.common_scalar(pk.transcript_repr)
        # every polynomial that goes to coefficient form and to the extended domain later lives in one block, so that
        # those transforms run as batches: advice | instance | per lookup (z, a', s') | permutation z
        n_adv, n_inst, n_lk, sets = cs.num_advice_columns, len(instance), len(cs.lookups), cs.num_permutation_sets()
        total = n_adv + n_inst + 3 * n_lk + sets
        sharded = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
        rank, world = (dist.get_rank(), dist.get_world_size()) if sharded else (0, 1)
        if sharded:
            # the block is a tensor of the collective library's framework, so that ranges of it can be broadcast
            import torch
            t_block = torch.empty((n * max(total, 1), 4), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            block = t_block.data_ptr()
        else:
            block = alloc(n * max(total, 1), "polynomials")
        slot = [block + i * n * 32 for i in range(total)]

        def exchange(slots_by_owner):
            """slots_by_owner[r]: the slots of the block that rank r has filled; afterwards every rank holds them all:
            one all-gather when every rank's slots are one run of neighbours (the contiguous blocks of `block_owner` are),
            otherwise maximal runs of neighbouring slots travel as one broadcast each"""
            ctx.sync()
            sorted_slots = [sorted(o) for o in slots_by_owner]
            if all(not o or o[-1] - o[0] + 1 == len(o) for o in sorted_slots):
                sharding.exchange_runs(t_block, [((o[0] if o else 0) * n, len(o) * n) for o in sorted_slots], dist)
                torch.cuda.synchronize()
                return
            for r, owned_slots in enumerate(slots_by_owner):
                owned_slots = sorted(owned_slots)
                i = 0
                while i < len(owned_slots):
                    j = i
                    while j + 1 < len(owned_slots) and owned_slots[j + 1] == owned_slots[j] + 1:
                        j += 1
                    sharding.broadcast_(t_block[owned_slots[i] * n:(owned_slots[j] + 1) * n], r, dist)
                    i = j + 1
            torch.cuda.synchronize()
        adv, inst = slot[:n_adv], slot[n_adv:n_adv + n_inst]
        lk_slots = [slot[n_adv + n_inst + 3 * i:n_adv + n_inst + 3 * i + 3] for i in range(n_lk)]
        perm_z = slot[total - sets:]
        # When every extended coset fits the device at once (the single-GPU path of the quotient phase below), the
        # extended form of an advice column is computed as soon as the column has arrived and been committed: it needs no
        # challenge, and while the witness crosses the host link the multiplier has little else to do.  The column then
        # exists in three forms: Lagrange (where it was uploaded: lookups and the permutation argument read it),
        # coefficient (`early_coeff`, moved over the Lagrange form when the quotient phase starts) and extended.
        BATCH = 64
        split = bool(pk.polys.split)       # extended vectors split by row parity: low-degree terms on half the domain
        if split and (sharded or coset_groups not in (None, 1)):
            raise ValueError("the proving key stores its extended vectors split by row parity (keygen(..., low_degree_split)): "
                             "it proves on one GPU with every coset resident")
        to_extended = dom.coeff_to_extended_split_dev if split else dom.coeff_to_extended_dev
        # split evaluation: the odd-indexed rows of a column are read by the terms above the degree bound only, so a
        # column that none of them names is extended onto the even-indexed half alone (a transform of half the size)
        whole_rows = [True] * max(total, 1)
        if split and pk.high_degree_columns is not None:
            used = pk.high_degree_columns
            whole_rows = (list(used["advice"]) + list(used["instance"]) + [f for t in used["lookups"] for f in t]
                          + list(used["perm_z"]))

        def extend_slots(s0, cnt, src, dst):
            """coefficient form of the polynomial slots s0 .. s0 + cnt - 1 (back to back at src) -> extended form at dst"""
            if split:
                evalh_host.extend_for_split(dom, whole_rows[s0:s0 + cnt], src, dst)
            else:
                to_extended(src, dst, cnt)
        early_ext = early_coeff = None
        early_done = set()
        held = workspace is not None and ("extended", en * max(total, 1) * 32) in workspace.blocks
        if (not sharded and coset_groups in (None, 1) and pk.polys.cosets is None and n_adv
                and (held or en * 32 * (total + 3 * BATCH) + n * 32 * n_adv <= 0.85 * ctx.mem_info()[0])):
            early_ext = alloc(en * max(total, 1), "extended")
            early_coeff = alloc(n * n_adv, "advice coefficients")

        def extend_early(cols):
            """coefficient and extended form of freshly committed advice columns (maximal runs of neighbouring columns
            go out as batches)"""
            if early_ext is None:
                return
            cols = sorted(cols)
            i = 0
            while i < len(cols):
                j = i
                while j + 1 < len(cols) and cols[j + 1] == cols[j] + 1 and j + 1 - i < BATCH:
                    j += 1
                c0, cnt = cols[i], j - i + 1
                ctx.copy_dev(early_coeff + c0 * n * 32, adv[c0], cnt * n * 32)
                dom.lagrange_to_coeff_dev(early_coeff + c0 * n * 32, cnt)
                extend_slots(c0, cnt, early_coeff + c0 * n * 32, early_ext + c0 * en * 32)
                early_done.update(cols[i:j + 1])
                i = j + 1

        for col, d in zip(instance, inst):
            for v in col:
                transcript.common_scalar(v)
            padded = np.zeros((n, 4), dtype=np.uint64)           # the rows beyond the instance values are zero
            if len(col):
                padded[:len(col)] = fr_limbs(col)
            ctx.upload(padded, d)
        # per phase: the phase's columns are assigned (from the challenges of the phases before), blinded, committed
        # and written, then the phase's challenges are squeezed
        challenges = [0] * cs.num_challenges
        for phase, columns, phase_challenges in cs.phases():
            assigned = advice(phase, list(challenges)) if callable(advice) else advice
            for c in columns:
                col = assigned[c]
                if isinstance(col, DeviceColumn) and (col.rows != usable or not col.ptr):
                    raise ValueError("a device-resident advice column holds the usable rows")
                if len(col) != usable or (isinstance(col, np.ndarray) and (col.dtype != np.uint64 or col.shape != (usable, 4))):
                    raise ValueError("an advice column holds the usable rows (integers, or (usable rows, 4) uint64 limbs)")
            points = []
            if sharded and columns:
                # every rank uploads and commits its block of the phase's columns, then the blocks are exchanged (the
                # lookups and the permutation argument read every column); the blinding rows are drawn by every rank
                owner = sharding.block_owner(len(columns), world)
                for c, r in zip(columns, owner):
                    col, d = assigned[c], adv[c]
                    blinding = draws(n - usable)
                    if r != rank:
                        continue
                    if isinstance(col, DeviceColumn):
                        ctx.copy_dev(d, col.ptr, usable * 32)
                        ctx.upload(fr_limbs(blinding), d + usable * 32)
                    elif isinstance(col, np.ndarray):
                        ctx.upload(col, d)
                        ctx.upload(fr_limbs(blinding), d + usable * 32)
                    else:
                        ctx.upload(np.concatenate([fr_limbs(col), fr_limbs(blinding)]), d)
                points.extend(params.commit_lagrange([adv[c] for c in columns], dist, owner))
                exchange([[c for c, o in zip(columns, owner) if o == r] for r in range(world)])
            elif columns and all(isinstance(assigned[c], DeviceColumn) for c in columns):
                # device-resident columns: copied into the block group by group, each group committed and extended while
                # the next is copied
                GROUP = 32
                for g0 in range(0, len(columns), GROUP):
                    group = columns[g0:g0 + GROUP]
                    for c in group:
                        ctx.copy_dev(adv[c], assigned[c].ptr, usable * 32)
                        blind_rows(adv[c], usable)
                    points.extend(params.commit_lagrange([adv[c] for c in group], dist))
                    extend_early(group)
            elif columns and all(isinstance(assigned[c], np.ndarray) and assigned[c].flags.c_contiguous
                                 and ctx.is_pinned(assigned[c]) for c in columns):
                # page-locked columns: the copies of one group of columns run on the transfer stream while the group
                # before it is committed, so that the link and the multiplier work side by side
                # (a short first group: the multiplier has nothing to do until it has arrived).  The extension of a group
                # is queued two commitments later, and what is still owed after the last commitment runs while the host
                # writes the points to the transcript and prepares the lookups; the staging buffer is released at the end
                # of the proof for the same reason (releasing page-locked memory waits for the device).
                GROUP, FIRST, LAG = 32, 8, 2
                stage = ctx.host_zeros((len(columns), n - usable, 4))      # the blinding rows of every column
                pinned_stages.append(stage)
                starts = [0] + list(range(min(FIRST, len(columns)), len(columns), GROUP))
                groups = [columns[a:b] for a, b in zip(starts, starts[1:] + [len(columns)])]

                def draw(gi):
                    """the blinding rows of a group, from the RNG (groups in order: the stream is that of column after column)"""
                    if gi < len(groups):
                        for i, _ in enumerate(groups[gi], starts[gi]):
                            stage[i] = fr_limbs(draws(n - usable))

                def issue(gi):
                    for i, c in enumerate(groups[gi], starts[gi]):
                        ctx.upload_overlapped(assigned[c], adv[c])
                        ctx.upload_overlapped(stage[i], adv[c] + usable * 32)

                draw(0)
                issue(0)
                draw(1)
                for gi, group in enumerate(groups):
                    ctx.upload_fence()                      # the compute stream waits for this group's copies
                    if gi + 1 < len(groups):
                        # queued before any kernel of this round: the first copy after a fence starts behind what the
                        # compute stream holds at that moment, which is nothing but the wait above
                        issue(gi + 1)
                    if gi >= LAG:
                        extend_early(groups[gi - LAG])
                    draw(gi + 2)                            # host work under the transforms just queued
                    points.extend(params.commit_lagrange([adv[c] for c in group], dist))
                for group in groups[max(0, len(groups) - LAG):]:
                    extend_early(group)
            else:
                for c in columns:                               # the unusable rows of every column are blinding
                    col, d = assigned[c], adv[c]
                    blinding = fr_limbs(draws(n - usable))
                    if isinstance(col, DeviceColumn):
                        ctx.copy_dev(d, col.ptr, usable * 32)
                        ctx.upload(blinding, d + usable * 32)
                    elif isinstance(col, np.ndarray):
                        ctx.upload(col, d)
                        ctx.upload(blinding, d + usable * 32)
                    else:
                        ctx.upload(np.concatenate([fr_limbs(col), blinding]), d)
                if columns:
                    points.extend(params.commit_lagrange([adv[c] for c in columns], dist))
            draws(len(columns))                                 # one Blind per column: drawn, unused by KZG
            for jac in points:
                transcript.write_point(jac)
            for c in phase_challenges:
                challenges[c] = transcript.squeeze_challenge()
        ch_limbs = fr_limbs(challenges) if challenges else np.zeros((0, 4), dtype=np.uint64)
        theta = transcript.squeeze_challenge()

        # ---- lookups: compressed expressions, permuted columns ----------------------------------------------
        mark("lookup_permuted")
        # With several ranks a lookup belongs to one of them (contiguous blocks of lookups): its owner compresses, sorts,
        # blinds, multiplies up and commits; the other ranks only advance the RNG and receive the three polynomials in
        # coefficient form before the quotient phase.
        lk_owner = sharding.block_owner(n_lk, world) if sharded else [0] * n_lk
        lookups = []
        for lk, (_, d_pa, d_ps), o in zip(cs.lookups, lk_slots, lk_owner):
            if o != rank:
                lookups.append([0, 0, d_pa, d_ps, 0])
                continue
            lookups.append([alloc(n), alloc(n), d_pa, d_ps, 0])
        mine = [lk for lk, o in zip(lookups, lk_owner) if o == rank]
        # the compressed input and table expressions of all (own) lookups: one call, the kernels back to back
        evalh_host.compress_lookups_dev(dom, cs, [lk for lk, o in zip(cs.lookups, lk_owner) if o == rank], pk.fixed_values, adv,
                                        inst, ch_limbs, limb(theta), [lk[0] for lk in mine], [lk[1] for lk in mine])
        # the permuted columns of all lookups in one sequence of launches; blinding rows and blinds in upstream's order
        if mine:
            dom.lookup_permute_batch_dev([lk[0] for lk in mine], [lk[1] for lk in mine], usable,
                                         [lk[2] for lk in mine], [lk[3] for lk in mine])
        for (_, _, d_pa, d_ps, _), o in zip(lookups, lk_owner):
            blind_rows(d_pa, usable, o == rank)
            blind_rows(d_ps, usable, o == rank)
            rng.fr(), rng.fr()
        # no challenge is squeezed between the lookups, so their commitments go out as one pipelined batch, written in
        # upstream's order (a' then s' of each lookup)
        if lookups:
            for jac in params.commit_lagrange([p for lk in lookups for p in (lk[2], lk[3])], dist,
                                              [o for o in lk_owner for _ in range(2)]):
                transcript.write_point(jac)
        beta = transcript.squeeze_challenge()
        gamma = transcript.squeeze_challenge()

        # ---- permutation argument ---------------------------------------------------------------------------
        mark("permutation_product")
        chunk = cs.permutation_chunk_len()
        pcols = cs.permutation_columns
        lagrange_of = {ANY_ADVICE: adv, ANY_FIXED: pk.fixed_values, ANY_INSTANCE: inst}
        last_z = 1
        for s, d_z in enumerate(perm_z):
            cols = pcols[s * chunk:(s + 1) * chunk]
            dom.permutation_product_dev([lagrange_of[kind][i] for kind, i in cols],
                                        pk.polys.permutations[s * chunk:s * chunk + len(cols)], limb(beta), limb(gamma),
                                        limb(pow(FR_DELTA, s * chunk, P)), limb(last_z), d_z)
            blind_rows(d_z, usable + 1)
            last_z = fr_from_mont_limbs(ctx.download(d_z + usable * 32, (1, 4))[0])
            rng.fr()
        if perm_z:
            for jac in params.commit_lagrange(perm_z, dist):
                transcript.write_point(jac)

        # ---- lookup grand products --------------------------------------------------------------------------
        mark("lookup_product")
        own = [(lk, d_z) for lk, (d_z, _, _), o in zip(lookups, lk_slots, lk_owner) if o == rank]
        if own:
            dom.lookup_product_batch_dev([lk[0] for lk, _ in own], [lk[1] for lk, _ in own], [lk[2] for lk, _ in own],
                                         [lk[3] for lk, _ in own], limb(beta), limb(gamma), [d_z for _, d_z in own])
        for lk, (d_z, _, _), o in zip(lookups, lk_slots, lk_owner):
            blind_rows(d_z, usable + 1, o == rank)
            rng.fr()
            lk[4] = d_z
        if lookups:
            for jac in params.commit_lagrange([lk[4] for lk in lookups], dist, lk_owner):
                transcript.write_point(jac)

        # ---- vanishing argument: random polynomial ----------------------------------------------------------
        mark("random_poly")
        d_random = ctx.upload(rng.limbs(n) if random_poly_threads is None
                              else random_poly_per_thread_streams(rng, n, random_poly_threads))
        owned.append(d_random)
        rng.fr()
        transcript.write_point(params.commit([d_random])[0])
        y = transcript.squeeze_challenge()

        # ---- coefficient form and extended cosets -----------------------------------------------------------
        mark("quotient")
        per_lookup = [(lk[4], lk[2], lk[3]) for lk in lookups]
        assert [p for t in per_lookup for p in t] == slot[n_adv + n_inst:total - sets]
        todo = [i for i in range(total) if i not in early_done]      # slots still in Lagrange form only (advice slots come first)

        def runs(idx):
            """maximal runs of neighbouring slots, at most BATCH long: (first, count)"""
            out, i = [], 0
            while i < len(idx):
                j = i
                while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1 and j + 1 - i < BATCH:
                    j += 1
                out.append((idx[i], j - i + 1))
                i = j + 1
            return out

        if sharded and n_lk:
            # a lookup's polynomials go to coefficient form on its owner and reach the other ranks in that form
            lk0 = n_adv + n_inst
            mine_slots = [lk0 + 3 * i + j for i, o in enumerate(lk_owner) if o == rank for j in range(3)]
            for s0, cnt in runs(mine_slots):
                dom.lagrange_to_coeff_dev(slot[s0], cnt)
            exchange([[lk0 + 3 * i + j for i, o in enumerate(lk_owner) if o == r for j in range(3)] for r in range(world)])
            todo = [i for i in todo if not lk0 <= i < lk0 + 3 * n_lk]
        for s0, cnt in runs(todo):
            dom.lagrange_to_coeff_dev(slot[s0], cnt)
        for c0, cnt in runs(sorted(early_done)):                     # their coefficient form replaces the Lagrange form
            ctx.copy_dev(slot[c0], early_coeff + c0 * n * 32, cnt * n * 32)
        call = evalh_host.EvalhCall(cs, pk.ev)
        call.challenges = ch_limbs
        call.y, call.beta, call.gamma, call.theta = limb(y), limb(beta), limb(gamma), limb(theta)
        d_h = alloc(en, "h")

        def fill(of_poly, of_key):
            """of_poly: coefficient polynomial -> its evaluations; of_key: extended key vector -> the same rows of it"""
            call.fixed = [of_key(p) for p in pk.polys.fixed_cosets]
            call.advice = [of_poly(p) for p in adv]
            call.instance = [of_poly(p) for p in inst]
            call.l0, call.l_last, call.l_active = (of_key(p) for p in (pk.polys.l0, pk.polys.l_last, pk.polys.l_active_row))
            call.lookups = [tuple(of_poly(p) for p in t) for t in per_lookup]
            call.perm_sigma = [of_key(p) for p in pk.polys.permutation_cosets]
            call.perm_z = [of_poly(p) for p in perm_z]

        if not sharded:
            if pk.polys.cosets is not None:
                raise ValueError("the proving key was generated for a multi-GPU partition (keygen(..., dist=...))")
            # Footprint of the quotient phase: every extended coset at once is total * 2^extended_k * 32 bytes plus the
            # transform scratch of a batch (83 GB + 34 GB at the SuperCircuit's width).  When that does not fit the free
            # memory -- a smaller GPU or a partition of this one -- the numerator is evaluated coset group by coset group
            # (what the ranks of the multi-GPU split do side by side), which holds 1 / groups of the cosets at a time.
            free_bytes = ctx.mem_info()[0]
            m = en // n
            if early_ext is not None or split:
                coset_groups = 1                   # the block of all extended cosets is already there (or has to be)
            if coset_groups is None:
                coset_groups = 1
                while coset_groups < m and (en // coset_groups) * 32 * (total + 3 * BATCH) > 0.85 * free_bytes:
                    coset_groups *= 2
            if coset_groups < 1 or coset_groups > m or coset_groups & (coset_groups - 1):
                raise ValueError("coset_groups must be a power of two between 1 and the number of cosets")
            BATCH = max(1, min(BATCH, int(0.1 * free_bytes) // (3 * en * 32)))
        if not sharded and coset_groups == 1:
            ext_block = early_ext if early_ext is not None else alloc(en * max(total, 1), "extended")
            for s0, cnt in runs(todo):
                extend_slots(s0, cnt, slot[s0], ext_block + s0 * en * 32)
            fill(lambda p: ext_block + (p - block) // (n * 32) * en * 32, lambda p: p)
            mark("quotient_evaluate_h")
            if split:
                evalh_host.evaluate_h_split(ctx, dom, call, d_h)
            else:
                dom.evaluate_h(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_h)
        elif not sharded:
            mark("quotient_evaluate_h")
            for g in range(coset_groups):
                cosets = list(range(g, m, coset_groups))
                rows = n * len(cosets)
                group = []

                def of_poly(p):
                    d = ctx.alloc(rows * 32)
                    group.append(d)
                    dom.coeff_to_cosets_dev(p, cosets, d)
                    return d

                def of_key(p):
                    d = ctx.alloc(rows * 32)
                    group.append(d)
                    dom.extended_to_cosets_dev(p, cosets, d)
                    return d
                fill(of_poly, of_key)
                call.cosets = cosets
                d_piece = ctx.alloc(rows * 32)
                group.append(d_piece)
                dom.evaluate_h(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_piece)
                dom.cosets_to_extended_dev(d_piece, cosets, d_h)
                ctx.sync()
                for d in group:
                    ctx.free(d)
            call.cosets = None
        else:
            import torch

            def evaluate(cosets, row_first, row_count):
                """this rank's rows of the numerator: every column in the compact layout of its cosets, of which the rows
                [row_first, row_first + row_count) are evaluated"""
                if not cosets:
                    return torch.empty((0, 4), dtype=torch.int64, device="cuda")
                rows = n * len(cosets)

                def of_poly(p):
                    d = alloc(rows)
                    dom.coeff_to_cosets_dev(p, cosets, d)
                    return d

                def of_key(p):
                    if pk.polys.cosets is not None:          # the key was generated for this partition: compact already
                        if pk.polys.cosets != list(cosets):
                            raise ValueError("the proving key holds other cosets than this rank evaluates")
                        return p
                    d = alloc(rows)
                    dom.extended_to_cosets_dev(p, cosets, d)
                    return d
                fill(of_poly, of_key)
                call.cosets = cosets
                call.rows = (row_first, row_count) if row_count != rows else None
                out = torch.empty((row_count, 4), dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                dom.evaluate_h(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), out.data_ptr())
                ctx.sync()
                return out
            pieces = sharding.evaluate_h_by_cosets(dom.extended_k, dom.k, evaluate, dist)
            assembled = sharding.assemble_cosets(pieces, dom.k)    # row ranges of a coset are joined on torch's stream
            torch.cuda.synchronize()                               # ... which the context's stream does not follow
            for cosets, rows in assembled:
                dom.cosets_to_extended_dev(rows.data_ptr(), cosets, d_h)
            ctx.sync()
        mark("quotient_pieces")
        dom.divide_by_vanishing_poly_dev(d_h)
        pieces = dom.quotient_poly_degree
        d_hc = alloc(n * pieces, "quotient pieces")
        dom.extended_to_coeff_dev(d_h, d_hc)
        h_pieces = [d_hc + i * n * 32 for i in range(pieces)]
        for _ in h_pieces:
            rng.fr()
        for jac in params.commit(h_pieces, dist):
            transcript.write_point(jac)
        x = transcript.squeeze_challenge()

        # ---- evaluations at x -------------------------------------------------------------------------------
        mark("evaluations")
        adv_q, fix_q, _ = collect_queries(cs)
        last_rot = -(bf + 1)
        xn = pow(x, n, P)
        d_hpoly = alloc(n)
        ctx.poly_lincomb_dev(h_pieces, [n] * pieces, fr_limbs([pow(xn, i, P) for i in range(pieces)]), n, d_hpoly)

        def point(rot):
            return x * pow(omega, rot % n, P) % P

        groups = {
            "advice": [(adv[c], r) for c, r in adv_q],
            "fixed": [(pk.polys.fixed_polys[c], r) for c, r in fix_q],
            "random": [(d_random, 0)],
            "sigma": [(p, 0) for p in pk.polys.permutation_polys],
            "perm": [q for s, z in enumerate(perm_z)
                     for q in [(z, 0), (z, 1)] + ([(z, last_rot)] if s + 1 < sets else [])],
            "lookup": [q for z, pa, ps in per_lookup for q in [(z, 0), (z, 1), (pa, 0), (pa, -1), (ps, 0)]],
            "h": [(d_hpoly, 0)],
        }
        flat = [q for name in ("advice", "fixed", "random", "sigma", "perm", "lookup", "h") for q in groups[name]]
        vals = ctx.eval_polynomials_dev([p for p, _ in flat], fr_limbs([point(r) for _, r in flat]), n)
        evals = {}
        for (p, r), v in zip(flat, vals):
            evals[(p, r)] = fr_from_mont_limbs(v)
        for name in ("advice", "fixed", "random", "sigma", "perm", "lookup"):
            for q in groups[name]:
                transcript.write_scalar(evals[q])

        # ---- multi-open: queries in upstream's order ---------------------------------------------------------
        mark("multiopen")
        order = list(groups["advice"]) + list(groups["perm"])
        for z, pa, ps in per_lookup:
            order += [(z, 0), (pa, 0), (ps, 0), (pa, -1), (z, 1)]
        order += groups["fixed"] + groups["sigma"] + groups["h"] + groups["random"]
        queries = [shplonk.Query(p, point(r), evals[(p, r)]) for p, r in order]
        polys = {p: p for p, _ in order}
        outs = shplonk.ProverSHPLONK(ctx, params.g, n).create_proof(polys, queries, transcript)
        owned.extend(outs)
        ctx.sync()
        mark(None)
    finally:
        for p in owned:
            ctx.free(p)
        for a in pinned_stages:
            ctx.host_free(a)
    return transcript.finalize()
