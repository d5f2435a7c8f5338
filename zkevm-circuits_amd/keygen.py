"""Host mirror of the proving-key pieces the hot path reads, built on the device.

Follows `halo2_proofs::plonk::keygen::keygen_pk` and `plonk::permutation::keygen::Assembly` (the reference calls them
at /root/reference/zkevm-circuits/benches/benchmark_list/main.rs:121-138 and
/root/reference/prover/src/test/create_all_file.rs:143-153): `fixed_polys` / `fixed_cosets`, the permutation
polynomials and their cosets, and the three Lagrange selectors l0, l_last, l_active_row on the extended domain.  The
copy-constraint bookkeeping (`Assembly::copy`, a union of cycles over (column, row) cells) is host logic and stays
here; everything of length 2^k or 2^extended_k runs through the C ABI.
"""
from __future__ import annotations

import numpy as np

from .evalh_host import fr_to_mont_limbs


class Assembly:
    """`permutation::keygen::Assembly`: the cycles of the copy constraints over `num_columns` columns of n rows.

    mapping[c][r] = next cell of the cycle through (c, r); aux[c][r] = representative of the cycle; sizes = cycle sizes,
    kept at the representative, so that the smaller cycle is merged into the larger one as upstream does."""

    def __init__(self, n: int, num_columns: int):
        self.n, self.num_columns = n, num_columns
        rows = np.arange(n, dtype=np.uint32)
        self.map_col = [np.full(n, c, dtype=np.uint32) for c in range(num_columns)]
        self.map_row = [rows.copy() for _ in range(num_columns)]
        self.aux_col = [np.full(n, c, dtype=np.uint32) for c in range(num_columns)]
        self.aux_row = [rows.copy() for _ in range(num_columns)]
        self.sizes = [np.ones(n, dtype=np.uint64) for _ in range(num_columns)]

    def copy(self, left_column: int, left_row: int, right_column: int, right_row: int):
        if not (0 <= left_column < self.num_columns and 0 <= right_column < self.num_columns):
            raise IndexError("copy: column is not part of the permutation argument")
        if not (0 <= left_row < self.n and 0 <= right_row < self.n):
            raise IndexError("copy: row out of bounds")
        left = (int(self.aux_col[left_column][left_row]), int(self.aux_row[left_column][left_row]))
        right = (int(self.aux_col[right_column][right_row]), int(self.aux_row[right_column][right_row]))
        if left == right:
            return
        if self.sizes[left[0]][left[1]] < self.sizes[right[0]][right[1]]:
            left, right = right, left
        self.sizes[left[0]][left[1]] += self.sizes[right[0]][right[1]]
        i = right
        while True:
            self.aux_col[i[0]][i[1]], self.aux_row[i[0]][i[1]] = left
            i = (int(self.map_col[i[0]][i[1]]), int(self.map_row[i[0]][i[1]]))
            if i == right:
                break
        lc, lr = self.map_col[left_column][left_row], self.map_row[left_column][left_row]
        self.map_col[left_column][left_row] = self.map_col[right_column][right_row]
        self.map_row[left_column][left_row] = self.map_row[right_column][right_row]
        self.map_col[right_column][right_row], self.map_row[right_column][right_row] = lc, lr


def _upload_words(ctx, a):
    return ctx.upload(np.ascontiguousarray(a, dtype=np.uint32).view(np.uint64))


class ProvingKeyPolys:
    """Device pointers of the key material `Evaluator::evaluate_h` and the permutation argument read."""

    def __init__(self):
        self.fixed_polys, self.fixed_cosets = [], []
        self.permutations, self.permutation_polys, self.permutation_cosets = [], [], []
        self.l0 = self.l_last = self.l_active_row = 0
        self.cosets = None     # None: the "coset" vectors are whole extended vectors; a list: compact layout of these cosets
        self.split = False     # whole extended vectors stored split by row parity (even-indexed rows, then odd-indexed)

    def pointers(self):
        """every device block the key owns (a rank that evaluates no coset holds no extended vectors: those entries are 0)"""
        every = (self.fixed_polys + self.fixed_cosets + self.permutations + self.permutation_polys + self.permutation_cosets
                 + [self.l0, self.l_last, self.l_active_row])
        return [p for p in every if p]


def keygen_polys(ctx, dom, fixed_values, assembly: Assembly | None, blinding_factors: int, cosets=None,
                 split: bool = False) -> ProvingKeyPolys:
    """fixed_values: list of (n, 4) Montgomery arrays (Lagrange basis).  Returns device-resident key polynomials; the
    caller frees `pointers()`.  cosets: a rank of a multi-GPU prover that evaluates only these cosets of the extended
    domain keeps the key's extended vectors (fixed_cosets, permutation_cosets, l0, l_last, l_active_row) in the compact
    layout of those cosets -- n * len(cosets) rows, straight from the coefficients, never forming the whole vector; an
    empty list (a rank that owns no coset) leaves those entries 0.  split (whole vectors only): the extended vectors are
    stored split by row parity, the layout `evalh_host.evaluate_h_split` reads."""
    n, en = dom.n, dom.extended_len
    pk = ProvingKeyPolys()
    pk.cosets = list(cosets) if cosets is not None else None
    pk.split = bool(split) and cosets is None

    def extend(d_poly):
        if pk.cosets is not None and not pk.cosets:
            # a rank beyond the owners of the partition (world sizes that are no power of two) evaluates no row of the
            # quotient: it keeps the 2^k-sized forms, which the replicated phases read, and no extended vector at all
            return 0
        if pk.cosets is None:
            d_coset = ctx.alloc(en * 32)
            if pk.split:
                dom.coeff_to_extended_split_dev(d_poly, d_coset)
            else:
                dom.coeff_to_extended_dev(d_poly, d_coset)
        else:
            d_coset = ctx.alloc(n * len(pk.cosets) * 32)
            dom.coeff_to_cosets_dev(d_poly, pk.cosets, d_coset)
        return d_coset

    def to_coeff_and_coset(d_lagrange):
        d_poly = ctx.alloc(n * 32)
        ctx.poly_lincomb_dev([d_lagrange], [n], np.array([fr_to_mont_limbs(1)], dtype=np.uint64), n, d_poly)   # copy
        dom.lagrange_to_coeff_dev(d_poly)
        return d_poly, extend(d_poly)

    # fixed columns: fixed_polys = lagrange_to_coeff(fixed_values), fixed_cosets = coeff_to_extended(fixed_polys)
    for col in fixed_values:
        d_l = ctx.upload(col)
        d_p, d_c = to_coeff_and_coset(d_l)
        ctx.sync()
        ctx.free(d_l)
        pk.fixed_polys.append(d_p)
        pk.fixed_cosets.append(d_c)

    # permutation polynomials: sigma_c[row] = delta^(mapping column) * omega^(mapping row)
    if assembly is not None:
        for c in range(assembly.num_columns):
            d_mc, d_mr = _upload_words(ctx, assembly.map_col[c]), _upload_words(ctx, assembly.map_row[c])
            d_sigma = ctx.alloc(n * 32)
            dom.permutation_sigma_dev(d_mc, d_mr, assembly.num_columns, d_sigma)
            d_p, d_c = to_coeff_and_coset(d_sigma)
            ctx.sync()
            ctx.free(d_mc)
            ctx.free(d_mr)
            pk.permutations.append(d_sigma)
            pk.permutation_polys.append(d_p)
            pk.permutation_cosets.append(d_c)

    # l0 = indicator of row 0, l_blind = indicator of the last `blinding_factors` rows, l_last = indicator of the row
    # before them; l_active_row = 1 - (l_last + l_blind).  Upstream forms l_active_row point by point on the extended
    # domain; the coefficients of 1 - l_last - l_blind extended give the same values.
    one = fr_to_mont_limbs(1)
    ind = np.zeros((3, n, 4), dtype=np.uint64)
    ind[0, 0] = one
    ind[1, n - blinding_factors - 1] = one
    if blinding_factors:
        ind[2, n - blinding_factors:] = one
    d_ind = [ctx.upload(a) for a in ind]
    for d in d_ind:
        dom.lagrange_to_coeff_dev(d)
    d_one = ctx.upload(np.array([one], dtype=np.uint64))
    d_act = ctx.alloc(n * 32)
    minus_one = fr_to_mont_limbs(-1)
    ctx.poly_lincomb_dev([d_one, d_ind[1], d_ind[2]], [1, n, n], np.array([one, minus_one, minus_one], dtype=np.uint64), n, d_act)
    outs = []
    for d in (d_ind[0], d_ind[1], d_act):
        outs.append(extend(d))
    ctx.sync()
    for d in d_ind + [d_one, d_act]:
        ctx.free(d)
    pk.l0, pk.l_last, pk.l_active_row = outs
    return pk


def g_to_lagrange(ctx, dom, g_affine):
    """`poly::kzg::commitment::g_to_lagrange`: the SRS in the Lagrange basis, g_lagrange[i] = n^-1 sum_j omega^(-i j) g[j]
    (a G1-valued inverse FFT).  g_affine: (2^k, 8) affine limbs; returns the same shape."""
    return ctx.g1_fft(g_affine, dom.k, dom.constant(1), dom.constant(6))   # omega_inv, ifft_divisor = n^-1
