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
        gates = b.graph(self.graph if self.graph is not None else ev.custom_gates, fr_to_mont_limbs)
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
        self._keep = b
        return d


def prepare(dom, cs: ConstraintSystem, ev: Evaluator):
    """Key-generation step: has the device library build (and keep) the evaluate_h program of this constraint system, as
    upstream builds `ProvingKey::ev`.  No column data is needed: the descriptor carries counts and graphs only."""
    call = EvalhCall(cs, ev)
    call.fixed, call.advice, call.instance = [0] * cs.num_fixed_columns, [0] * cs.num_advice_columns, [0] * cs.num_instance_columns
    call.challenges = np.zeros((cs.num_challenges, 4), dtype=np.uint64)
    call.y = call.beta = call.gamma = call.theta = np.zeros(4, dtype=np.uint64)
    call.l0 = call.l_last = call.l_active = 0
    call.lookups = [(0, 0, 0)] * len(cs.lookups)
    call.perm_sigma = [0] * len(cs.permutation_columns)
    call.perm_z = [0] * cs.num_permutation_sets()
    dom.evaluate_h_prepare(call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED))


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


class _NoGraphs:
    custom_gates = None
    lookups = ()
