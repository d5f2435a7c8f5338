"""Host-side assembly of the `evaluate_h` call: ConstraintSystem + Evaluator + column data -> zkhip_evalh_desc.

Mirrors what `plonk::prover::create_proof` hands to `Evaluator::evaluate_h` (pk cosets, advice / instance
polynomials, challenges, y, beta, gamma, theta, lookup and permutation commitments-in-progress).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import ffi
from .plonk import ANY_ADVICE, ANY_FIXED, ANY_INSTANCE, FR_MODULUS, ConstraintSystem, Evaluator

R256 = (1 << 256) % FR_MODULUS


def fr_to_mont_limbs(v: int):
    m = (v % FR_MODULUS) * R256 % FR_MODULUS
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


class EvalhCall:
    """Column data for one evaluate_h call.

    All arrays are numpy uint64 (len, 4) Montgomery limbs, or device pointers (ints) when mem == MEM_DEVICE.
    fixed / l0 / l_last / l_active / perm_sigma / perm_z are extended cosets; advice / instance / lookup
    polynomials are in `form` (coefficients of length 2^k, or extended cosets).
    """

    def __init__(self, cs: ConstraintSystem, ev: Evaluator | None = None):
        self.cs = cs
        self.ev = ev or Evaluator(cs)
        self.fixed, self.advice, self.instance = [], [], []
        self.challenges = np.zeros((0, 4), dtype=np.uint64)
        self.y = self.beta = self.gamma = self.theta = None
        self.l0 = self.l_last = self.l_active = None
        self.lookups = []        # [(product, permuted_input, permuted_table)]
        self.perm_sigma, self.perm_z = [], []
        self.cosets = None       # list of coset indices: evaluate only these, every extended vector in compact layout
        self.rows = None         # (first, count): evaluate only this window of the rows (columns stay whole)
        self.graph = None        # with basis = BASIS_LAGRANGE: the graph evaluated over the 2^k rows of the circuit
        self.basis = ffi.BASIS_EXTENDED
        self.terms = None        # (TERMS_LOW | TERMS_HIGH, degree bound): evaluate only the terms of h on one side of it

    def build(self, mem=ffi.MEM_HOST, form=ffi.FORM_EXTENDED):
        cs, ev = self.cs, self.ev
        b = ffi.DescBuilder()
        d = ffi.EvalhDesc()
        d.mem, d.poly_form = mem, form
        d.num_fixed, d.num_advice, d.num_instance = len(self.fixed), len(self.advice), len(self.instance)
        d.num_challenges = len(self.challenges)

        def ptrs(arrs):
            return b.host_ptrs(arrs) if mem == ffi.MEM_HOST else b.ptr_array(arrs)

        def ptr(a):
            if mem == ffi.MEM_HOST:
                a = ffi.u64(a)
                b.keep.append(a)
                return C.c_void_p(a.ctypes.data)
            return C.c_void_p(int(a))

        d.fixed_cosets, d.advice, d.instance = ptrs(self.fixed), ptrs(self.advice), ptrs(self.instance)
        ch = ffi.u64(self.challenges)
        b.keep.append(ch)
        d.challenges = C.c_void_p(ch.ctypes.data) if len(ch) else None
        d.y, d.beta, d.gamma, d.theta = (b.limbs(self.y), b.limbs(self.beta), b.limbs(self.gamma),
                                         b.limbs(self.theta))
        if self.basis == ffi.BASIS_EXTENDED:
            d.l0, d.l_last, d.l_active_row = ptr(self.l0), ptr(self.l_last), ptr(self.l_active)
        d.basis = self.basis
        g = self.graph if self.graph is not None else ev.custom_gates
        if g is not None:                      # None: a descriptor for `Domain.evaluate_graphs_dev`, which brings its graphs
            gates = b.graph(g, fr_to_mont_limbs)
            b.keep.append(gates)
            d.custom_gates = C.pointer(gates)
        lk_arr = (ffi.Lookup * max(len(self.lookups), 1))()
        for i, (g, (prod, pin, ptab)) in enumerate(zip(ev.lookups, self.lookups)):
            lk_arr[i].graph = b.graph(g, fr_to_mont_limbs)
            lk_arr[i].product, lk_arr[i].permuted_input, lk_arr[i].permuted_table = ptr(prod), ptr(pin), ptr(ptab)
        b.keep.append(lk_arr)
        d.num_lookups = len(self.lookups)
        d.lookups = C.cast(lk_arr, C.POINTER(ffi.Lookup))
        kinds = np.array([k for k, _ in cs.permutation_columns], dtype=np.uint32)
        idxs = np.array([i for _, i in cs.permutation_columns], dtype=np.uint32)
        b.keep += [kinds, idxs]
        d.num_perm_columns = len(kinds) if self.perm_z else 0
        d.perm_column_kind = C.c_void_p(kinds.ctypes.data) if len(kinds) else None
        d.perm_column_index = C.c_void_p(idxs.ctypes.data) if len(idxs) else None
        d.perm_sigma_cosets = ptrs(self.perm_sigma)
        d.num_perm_sets = len(self.perm_z)
        d.perm_product_cosets = ptrs(self.perm_z)
        d.perm_chunk_len = cs.permutation_chunk_len()
        d.blinding_factors = cs.blinding_factors()
        if self.cosets is not None:
            cl = np.ascontiguousarray(np.asarray(self.cosets, dtype=np.uint32))
            b.keep.append(cl)
            d.num_cosets, d.cosets = len(cl), C.c_void_p(cl.ctypes.data)
        if self.rows is not None:
            d.row_first, d.row_count = self.rows
        if self.terms is not None:
            d.term_select, d.term_degree_max = self.terms
        self._keep = b
        return d


def prepare(dom, cs: ConstraintSystem, ev: Evaluator, terms=None):
    """Key-generation step: has the device library build (and keep) the evaluate_h program of this constraint system, as
    upstream builds `ProvingKey::ev`.  No column data is needed: the descriptor carries counts and graphs only.
    terms: the (selection, degree bound) of `EvalhCall.terms`; returns the program's size (instructions,
    multiplications, column reads, value slots per row)."""
    dom.evaluate_h_prepare(_shape_only(cs, ev, terms).build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED))
    return dom.ctx.evaluate_h_last_stats()


def _shape_only(cs: ConstraintSystem, ev: Evaluator, terms=None) -> EvalhCall:
    """a call that carries the counts and graphs of the constraint system and no column data"""
    call = EvalhCall(cs, ev)
    call.terms = terms
    call.fixed, call.advice, call.instance = [0] * cs.num_fixed_columns, [0] * cs.num_advice_columns, [0] * cs.num_instance_columns
    call.challenges = np.zeros((cs.num_challenges, 4), dtype=np.uint64)
    call.y = call.beta = call.gamma = call.theta = np.zeros(4, dtype=np.uint64)
    call.l0 = call.l_last = call.l_active = 0
    call.lookups = [(0, 0, 0)] * len(cs.lookups)
    call.perm_sigma = [0] * len(cs.permutation_columns)
    call.perm_z = [0] * cs.num_permutation_sets()
    return call


def columns_used(dom, cs: ConstraintSystem, ev: Evaluator, terms=None):
    """Which columns the program of this constraint system (restricted to `terms` as in `prepare`) reads: a dict of
    boolean lists  fixed / advice / instance / l (l0, l_last, l_active) / perm_z / perm_sigma  and  lookups  (one
    (product, permuted input, permuted table) triple per lookup)."""
    flags = [bool(v) for v in dom.evaluate_h_columns_used(_shape_only(cs, ev, terms).build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED))]
    out, at = {}, 0

    def take(count):
        nonlocal at
        at += count
        return flags[at - count:at]

    out["fixed"], out["advice"], out["instance"] = take(cs.num_fixed_columns), take(cs.num_advice_columns), take(cs.num_instance_columns)
    out["l"] = take(3)
    sets = cs.num_permutation_sets()
    if sets:
        take(1)                                                    # the column of X values, which the library keeps itself
    out["perm_z"] = take(sets)
    out["perm_sigma"] = take(len(cs.permutation_columns) if sets else 0)
    out["lookups"] = [tuple(take(3)) for _ in cs.lookups]
    assert at == len(flags), (at, len(flags))
    return out


# ---- low-degree terms on half the extended domain ---------------------------------------------------------------------
# h = sum_i y^(N-1-i) term_i.  A term of degree d in the columns (each a polynomial of degree < n = 2^k) has degree at most
# d (n - 1); the sum L of the terms with d <= m / 2 (m = 2^e cosets make up the extended domain) is therefore determined by
# its values on m n / 2 points.  The even-indexed extended rows are such a set -- the coset zeta <extended_omega^2> -- so L
# is evaluated there only and carried to the odd-indexed rows by one inverse and one forward transform of half the
# extended size (`Domain.extend_half_dev`); the other terms run on every row as before.  Every column must be in the
# parity-split layout (`Domain.coeff_to_extended_split_dev`: even-indexed rows, then odd-indexed rows): a half is then the
# compact layout of the even / the odd cosets, which `zkhip_evaluate_h` reads through its coset list.
def split_gain(dom, cs: ConstraintSystem, ev: Evaluator):
    """(low, high, all) multiplications per row of the three programs, built and cached as `prepare` does.  The split
    evaluates low + 2 * high per pair of rows where the whole program costs 2 * all."""
    m = 1 << (dom.extended_k - dom.k)
    if m < 2:
        return None
    whole = prepare(dom, cs, ev)
    low = prepare(dom, cs, ev, (ffi.TERMS_LOW, m // 2))
    high = prepare(dom, cs, ev, (ffi.TERMS_HIGH, m // 2))
    return low["multiplications"], high["multiplications"], whole["multiplications"]


def warm_up_split(ctx, dom):
    """Key-generation step of the split evaluation: the transforms of half the extended size get their twiddle tables now
    (built once per context and size), not inside the first proof."""
    half = dom.extended_len // 2
    d = ctx.alloc(2 * half * 32)
    ctx.upload(np.zeros((1, 4), dtype=np.uint64), d)              # contents are irrelevant; any residues transform
    dom.extend_half_dev(d, d + half * 32)
    ctx.sync()
    ctx.free(d)


def extend_for_split(dom, whole_rows, src: int, dst: int):
    """Coefficient polynomials back to back at src -> extended vectors in the parity-split layout back to back at dst, one
    per flag of `whole_rows`: a flagged polynomial gets every row (`coeff_to_extended_split_dev`), another -- one that no
    term above the degree bound reads, `columns_used(.., (TERMS_HIGH, bound))` -- the even-indexed half only, by a
    transform of half the size; its odd-indexed half is left as it is and never read."""
    n, en = dom.n, dom.extended_len
    i, cnt = 0, len(whole_rows)
    while i < cnt:
        j = i
        while j + 1 < cnt and bool(whole_rows[j + 1]) == bool(whole_rows[i]):
            j += 1
        if whole_rows[i]:
            dom.coeff_to_extended_split_dev(src + i * n * 32, dst + i * en * 32, j - i + 1)
        else:
            dom.coeff_to_extended_half_dev(src + i * n * 32, dst + i * en * 32, en, j - i + 1)
        i = j + 1


def shifted(call: EvalhCall, offset: int) -> EvalhCall:
    """a copy of a device-memory call with `offset` bytes added to every extended-vector pointer"""
    import copy
    c = copy.copy(call)
    mv = lambda ps: [p + offset for p in ps]                                        # noqa: E731
    c.fixed, c.advice, c.instance = mv(call.fixed), mv(call.advice), mv(call.instance)
    c.l0, c.l_last, c.l_active = call.l0 + offset, call.l_last + offset, call.l_active + offset
    c.lookups = [tuple(p + offset for p in t) for t in call.lookups]
    c.perm_sigma, c.perm_z = mv(call.perm_sigma), mv(call.perm_z)
    return c


def evaluate_h_split(ctx, dom, call: EvalhCall, d_h: int):
    """`Evaluator::evaluate_h` over device columns in the parity-split layout (extended form), with the terms of degree
    <= m / 2 evaluated on the even-indexed rows only.  Writes the 2^extended_k values of h, in extended order, to d_h."""
    m = 1 << (dom.extended_k - dom.k)
    half = dom.extended_len // 2
    even, odd = list(range(0, m, 2)), list(range(1, m, 2))
    bound = m // 2
    one = np.array([fr_to_mont_limbs(1)] * 2, dtype=np.uint64)
    bufs = [ctx.alloc(half * 32) for _ in range(4)]
    d_low_even, d_low_odd, d_high_even, d_high_odd = bufs
    try:
        lo = shifted(call, 0)
        lo.cosets, lo.terms = even, (ffi.TERMS_LOW, bound)
        dom.evaluate_h(lo.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_low_even)
        dom.extend_half_dev(d_low_even, d_low_odd)
        hi = shifted(call, 0)
        hi.cosets, hi.terms = even, (ffi.TERMS_HIGH, bound)
        dom.evaluate_h(hi.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_high_even)
        hi = shifted(call, half * 32)
        hi.cosets, hi.terms = odd, (ffi.TERMS_HIGH, bound)
        dom.evaluate_h(hi.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED), d_high_odd)
        ctx.poly_lincomb_dev([d_low_even, d_high_even], [half, half], one, half, d_high_even)
        ctx.poly_lincomb_dev([d_low_odd, d_high_odd], [half, half], one, half, d_high_odd)
        dom.cosets_to_extended_dev(d_high_even, even, d_h)
        dom.cosets_to_extended_dev(d_high_odd, odd, d_h)
        ctx.sync()
    finally:
        for p in bufs:
            ctx.free(p)


def compress_graph(expressions):
    """The graph of `compress_expressions` in `plonk::lookup::prover::Argument::commit_permuted`: the expressions of one
    side of a lookup folded by Horner in theta (first expression = highest power), as upstream's Evaluator builds it for
    the extended domain."""
    from .plonk import GraphEvaluator, OP_HORNER, VS_CONSTANT, VS_THETA
    g = GraphEvaluator()
    parts = tuple(g.add_expression(x) for x in expressions)
    g.add_calculation((OP_HORNER, (VS_CONSTANT, 0, 0), parts, (VS_THETA, 0, 0)))
    return g


def lookup_compress_call(cs: ConstraintSystem, expressions, fixed, advice, instance, challenges, theta):
    """EvalhCall that evaluates theta-compressed `expressions` over the Lagrange-basis columns (2^k rows each)."""
    call = EvalhCall(cs, ev=_NoGraphs())
    call.basis = ffi.BASIS_LAGRANGE
    call.graph = compress_graph(expressions)
    call.fixed, call.advice, call.instance = fixed, advice, instance
    call.challenges = challenges
    call.y = call.beta = call.gamma = np.zeros(4, dtype=np.uint64)
    call.theta = theta
    return call


def compress_lookups_dev(dom, cs: ConstraintSystem, lookups, fixed, advice, instance, challenges, theta, d_inputs, d_tables):
    """`compress_expressions` of `Argument::commit_permuted` for every lookup of `lookups` in one library call: the
    theta-compressed input expressions of lookup i go to d_inputs[i], its table expressions to d_tables[i] (device vectors
    of 2^k elements) -- what `lookup_compress_call` + `Domain.evaluate_h` do pair by pair, without a wait per graph.  The
    graphs are built once per lookup and kept on it."""
    if not lookups:
        return
    b = ffi.DescBuilder()
    graphs = (ffi.Graph * (2 * len(lookups)))()
    outs = []
    for i, lk in enumerate(lookups):
        pair = lk.__dict__.get("_compress_graphs")
        if pair is None:
            pair = lk.__dict__["_compress_graphs"] = (compress_graph(lk.input_expressions), compress_graph(lk.table_expressions))
        graphs[2 * i], graphs[2 * i + 1] = b.graph(pair[0], fr_to_mont_limbs), b.graph(pair[1], fr_to_mont_limbs)
        outs += [d_inputs[i], d_tables[i]]
    call = EvalhCall(cs, ev=_NoGraphs())
    call.basis = ffi.BASIS_LAGRANGE
    call.fixed, call.advice, call.instance = fixed, advice, instance
    call.challenges = challenges
    call.y = call.beta = call.gamma = np.zeros(4, dtype=np.uint64)
    call.theta = theta
    desc = call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED)
    dom.evaluate_graphs_dev(desc, graphs, outs)
    del b


class _NoGraphs:
    custom_gates = None
    lookups = ()
