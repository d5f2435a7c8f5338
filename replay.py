#!/usr/bin/env python3
"""Replay of the polynomial pipeline of one `create_proof` call on one MI355X.

The reference's SuperCircuit cannot be instantiated here (its halo2 fork and a Rust toolchain are absent), so this
script runs the *operator multiset* that `create_proof` issues for a constraint system of a given shape, on synthetic
columns, in upstream's phase order:

  1. advice:        commit_lagrange (MSM) and lagrange_to_coeff (iNTT) for every advice column
  2. lookups:       per lookup the theta-compression of its input and table expressions over the Lagrange-basis columns,
                    the sort that builds the permuted columns (permute_expression_pair), two commitments
                    of the permuted columns, the grand product z and its commitment, three iNTTs
  3. permutation:   per column set the grand product z, its commitment and iNTT
  4. quotient:      coeff_to_extended of every advice / instance / lookup polynomial, evaluate_h,
                    divide_by_vanishing_poly, extended_to_coeff, commitments of the d-1 pieces of h
  5. openings:      eval_polynomial for every (polynomial, rotation) query, then the SHPLONK multi-open over those
                    queries: rotation sets, quotient and linearisation polynomials, Kate divisions, two commitments

Witness generation, transcript hashing and the RNG stay on the host in the reference and are not part of this number.
The default shape is the k = 19 SuperCircuit's (standard feature set) as the committed static census of the reference's
sources gives it (tests/golden/supercircuit_census.json via `synthetic.supercircuit_shape`: advice, fixed, lookup and
permutation column counts, each with the file:line of the calls counted), constraint degree 9 (extended domain 8n); the
number of gate polynomials (the census counts `create_gate` calls, not the polynomials each returns) stays an estimate.
Prints one JSON line with per-phase seconds.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import zkhip_pkg  # noqa: E402

zk = zkhip_pkg.load()
from zkevm_circuits_amd import evalh_host, ffi, shplonk, synthetic  # noqa: E402


def rand_fr_device(n, gen):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=gen)
    t[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return t


class Phases:
    def __init__(self, ctx):
        self.ctx, self.t = ctx, {}

    def run(self, name, fn):
        torch.cuda.synchronize()
        self.ctx.sync()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.ctx.sync()
        self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=19)
    census = synthetic.supercircuit_shape("standard")
    ap.add_argument("--advice", type=int, default=census["advice"])
    ap.add_argument("--fixed", type=int, default=census["fixed"])
    ap.add_argument("--lookups", type=int, default=census["lookups"])
    ap.add_argument("--lookup-width", type=int, default=4)
    ap.add_argument("--perm-cols", type=int, default=census["permutation_columns"])
    ap.add_argument("--degree", type=int, default=census["degree"])
    ap.add_argument("--gate-polys", type=int, default=3000)
    ap.add_argument("--no-split", action="store_true", help="evaluate every term of the quotient numerator on every row")
    ap.add_argument("--reference-log", metavar="PATH",
                    help="also write the summary / Start / End lines the reference's benchmark prints "
                         "(zkevm-circuits/benches/benchmark_list/main.rs:44-52,68-71,236-245), with the replay's total as "
                         "the [Create_proof] phase")
    args = ap.parse_args()

    torch.cuda.set_device(0)
    ctx = zk.Context(0)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250523)
    k, n = args.k, 1 << args.k

    t_host0 = time.perf_counter()
    cs = synthetic.synthetic_cs(num_advice=args.advice, num_fixed=args.fixed, num_instance=1,
                                num_gate_polys=args.gate_polys, max_degree=args.degree, rotations=(-5, -2, -1, 0, 1),
                                num_lookups=args.lookups, lookup_width=args.lookup_width, num_perm_cols=args.perm_cols)
    d = cs.degree()
    dom = zk.Domain(ctx, d, k)
    en, ek = dom.extended_len, dom.extended_k
    call = evalh_host.EvalhCall(cs)
    host_setup_s = time.perf_counter() - t_host0

    A, F, I, L = cs.num_advice_columns, cs.num_fixed_columns, cs.num_instance_columns, len(cs.lookups)
    sets = cs.num_permutation_sets()
    bases = ctx.g1_multiples(n)
    bid = ctx.register_bases(bases)            # stands for both `g` and `g_lagrange`: same size, same kernels

    with torch.cuda.stream(stream):
        # Lagrange-basis witness columns (n rows each), one tensor per group
        advice = rand_fr_device(A * n, gen)
        instance = rand_fr_device(I * n, gen)
        lk_cols = rand_fr_device(L * 4 * n, gen)        # per lookup: compressed input, compressed table, a', s'
        lk_z = torch.empty((L * n, 4), dtype=torch.int64, device="cuda")
        perm_z = torch.empty((sets * n, 4), dtype=torch.int64, device="cuda")
        sigma_lagrange = rand_fr_device(len(cs.permutation_columns) * n, gen)
        # proving-key cosets
        fixed_ext = rand_fr_device(F * en, gen)
        sigma_ext = rand_fr_device(len(cs.permutation_columns) * en, gen)
        l_ext = rand_fr_device(3 * en, gen)
        # extended cosets produced in phase 4
        n_ext_cols = A + I + 3 * L + sets
        ext = torch.empty((n_ext_cols * en, 4), dtype=torch.int64, device="cuda")
        h = torch.empty((en, 4), dtype=torch.int64, device="cuda")
        h_coeff = torch.empty((n * dom.quotient_poly_degree, 4), dtype=torch.int64, device="cuda")
    stream.synchronize()
    sc = rand_fr_device(12, gen).cpu().numpy().view(np.uint64)
    beta, gamma, theta, y, x = sc[0], sc[1], sc[2], sc[3], sc[4]

    def colptr(t, i, rows):
        return t.data_ptr() + i * rows * 32

    ph = Phases(ctx)
    CH = 64   # columns per batched call

    def batched(fn, total):
        for s in range(0, total, CH):
            fn(s, min(CH, total - s))

    # ---- 1. advice --------------------------------------------------------------------------------------
    ph.run("advice_commit_msm", lambda: batched(
        lambda s, c: ctx.msm_batch_dev(bid, [colptr(advice, s + i, n) for i in range(c)], n), A))
    # lookup expressions are compressed over the Lagrange-basis columns, i.e. before the columns go to coefficient form
    # (upstream keeps both forms); the results are stand-ins only: a satisfied lookup is constructed below
    t0 = time.perf_counter()
    lag_fixed = [colptr(fixed_ext, i, n) for i in range(F)]
    lag_advice = [colptr(advice, i, n) for i in range(A)]
    lag_instance = [colptr(instance, i, n) for i in range(I)]
    ch_limbs = rand_fr_device(cs.num_challenges, gen).cpu().numpy().view(np.uint64)
    compress_descs = []
    for lk in cs.lookups:
        for exprs in (lk.input_expressions, lk.table_expressions):
            c = evalh_host.lookup_compress_call(cs, exprs, lag_fixed, lag_advice, lag_instance, ch_limbs, theta)
            compress_descs.append((c, c.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED)))
    host_setup_s += time.perf_counter() - t0

    def lookup_compress():
        for j, (_, desc_) in enumerate(compress_descs):
            dom.evaluate_h(desc_, colptr(lk_cols, 4 * (j // 2) + (j % 2), n))
    ph.run("lookup_compress", lookup_compress)
    ph.run("advice_lagrange_to_coeff", lambda: batched(lambda s, c: dom.lagrange_to_coeff_dev(colptr(advice, s, n), c), A))
    dom.lagrange_to_coeff_dev(instance.data_ptr(), I)

    # ---- 2. lookups -------------------------------------------------------------------------------------
    # permuted columns a', s' from (compressed input, compressed table): the table column is random and the input column
    # draws its values from the table, as a satisfied lookup does
    usable = n - (cs.blinding_factors() + 1)

    def lookup_permute():   # all lookups of the proof in one batched call, as prover.create_proof issues them
        dom.lookup_permute_batch_dev([colptr(lk_cols, 4 * i, n) for i in range(L)], [colptr(lk_cols, 4 * i + 1, n) for i in range(L)],
                                     usable, [colptr(lk_cols, 4 * i + 2, n) for i in range(L)],
                                     [colptr(lk_cols, 4 * i + 3, n) for i in range(L)])
    with torch.cuda.stream(stream):
        view = lk_cols.view(L, 4, n, 4)
        pick = torch.randint(0, usable, (n,), device="cuda", generator=gen)
        for i in range(L):
            view[i, 0] = view[i, 1][pick]
    stream.synchronize()
    ph.run("lookup_permute", lookup_permute)
    # commitments of a' and s' (2L MSMs), then z from (input, table, a', s') and its commitment, then 3L iNTTs
    ph.run("lookup_commit_msm", lambda: batched(
        lambda s, c: ctx.msm_batch_dev(bid, [colptr(lk_cols, 4 * (s + i) + 2 + j, n) for i in range(c) for j in range(2)], n), L))

    def lookup_products():
        for i in range(L):
            dom.lookup_product_dev(colptr(lk_cols, 4 * i, n), colptr(lk_cols, 4 * i + 1, n), colptr(lk_cols, 4 * i + 2, n),
                                   colptr(lk_cols, 4 * i + 3, n), beta, gamma, colptr(lk_z, i, n))
    ph.run("lookup_grand_product", lookup_products)
    ph.run("lookup_commit_msm", lambda: batched(
        lambda s, c: ctx.msm_batch_dev(bid, [colptr(lk_z, s + i, n) for i in range(c)], n), L))

    def lookup_intt():
        batched(lambda s, c: dom.lagrange_to_coeff_dev(colptr(lk_z, s, n), c), L)
        for i in range(L):   # a' and s' sit at stride 4 in lk_cols: two single-column calls per lookup
            dom.lagrange_to_coeff_dev(colptr(lk_cols, 4 * i + 2, n), 2)
    ph.run("lookup_lagrange_to_coeff", lookup_intt)

    # ---- 3. permutation ---------------------------------------------------------------------------------
    chunk = cs.permutation_chunk_len()
    pcols = cs.permutation_columns
    one = np.array([0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f], dtype=np.uint64)  # R mod r

    def perm_products():
        for s in range(sets):
            c0, c1 = s * chunk, min((s + 1) * chunk, len(pcols))
            vals = [colptr(advice, (c % A), n) for c in range(c0, c1)]     # stand-in column values
            sigs = [colptr(sigma_lagrange, c, n) for c in range(c0, c1)]
            dom.permutation_product_dev(vals, sigs, beta, gamma, one, one, colptr(perm_z, s, n))
    ph.run("permutation_grand_product", perm_products)
    ph.run("permutation_commit_msm", lambda: ctx.msm_batch_dev(bid, [colptr(perm_z, s, n) for s in range(sets)], n))
    ph.run("permutation_lagrange_to_coeff", lambda: dom.lagrange_to_coeff_dev(perm_z.data_ptr(), sets))
    ph.run("random_poly_commit_msm", lambda: ctx.msm_range_dev(bid, 0, colptr(advice, 0, n), n))

    # ---- 4. quotient ------------------------------------------------------------------------------------
    # Low-degree terms on half the extended domain (zkevm_circuits_amd/evalh_host.py: evaluate_h_split), under the rule
    # `prover.keygen` applies: on when the terms of degree <= (number of cosets) / 2 hold enough of the program's products.
    # The extended vectors are then stored split by row parity (the key's would be, from key generation on).
    from zkevm_circuits_amd import prover as _prover
    split_programs = None
    split = False
    if not args.no_split and ek > k:
        low, high, whole = evalh_host.split_gain(dom, cs, call.ev)
        split_programs = {"low": low, "high": high, "whole": whole}
        split = whole > 0 and low >= _prover.SPLIT_MIN_LOW_SHARE * whole
    half_extended = 0
    if split:
        evalh_host.warm_up_split(ctx, dom)          # as key generation does: twiddle tables of the half-size transforms
        # columns that no term above the degree bound reads are extended onto the even-indexed rows only
        used = evalh_host.columns_used(dom, cs, call.ev, (ffi.TERMS_HIGH, (en // n) // 2))
        half_extended = sum(not f for f in list(used["advice"]) + list(used["instance"]) + list(used["perm_z"])
                            + [g for t in used["lookups"] for g in t])

        def to_ext(src, dst, cnt, flags):
            evalh_host.extend_for_split(dom, flags, src, dst)
    else:
        def to_ext(src, dst, cnt, flags):
            dom.coeff_to_extended_dev(src, dst, cnt)

    # coefficient polynomials -> extended cosets (advice | instance | per lookup z, a', s' | perm z)
    def extend_all():
        o = 0
        f = used if split else None
        batched(lambda s, c: to_ext(colptr(advice, s, n), colptr(ext, o + s, en), c, f and f["advice"][s:s + c]), A)
        o += A
        to_ext(instance.data_ptr(), colptr(ext, o, en), I, f and f["instance"])
        o += I
        for i in range(L):
            to_ext(colptr(lk_z, i, n), colptr(ext, o + 3 * i, en), 1, f and f["lookups"][i][:1])
            to_ext(colptr(lk_cols, 4 * i + 2, n), colptr(ext, o + 3 * i + 1, en), 2, f and f["lookups"][i][1:])
        o += 3 * L
        to_ext(perm_z.data_ptr(), colptr(ext, o, en), sets, f and f["perm_z"])
    ph.run("coeff_to_extended", extend_all)

    call.fixed = [colptr(fixed_ext, i, en) for i in range(F)]
    call.advice = [colptr(ext, i, en) for i in range(A)]
    call.instance = [colptr(ext, A + i, en) for i in range(I)]
    call.challenges = rand_fr_device(cs.num_challenges, gen).cpu().numpy().view(np.uint64)
    call.y, call.beta, call.gamma, call.theta = y, beta, gamma, theta
    call.l0, call.l_last, call.l_active = (colptr(l_ext, i, en) for i in range(3))
    o = A + I
    call.lookups = [(colptr(ext, o + 3 * i, en), colptr(ext, o + 3 * i + 1, en), colptr(ext, o + 3 * i + 2, en)) for i in range(L)]
    call.perm_sigma = [colptr(sigma_ext, c, en) for c in range(len(pcols))]
    call.perm_z = [colptr(ext, o + 3 * L + s, en) for s in range(sets)]
    t0 = time.perf_counter()
    desc = call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED)
    host_setup_s += time.perf_counter() - t0
    ctx.timing_enable(True)
    ctx.timing_reset()
    if split:
        ph.run("evaluate_h", lambda: evalh_host.evaluate_h_split(ctx, dom, call, h.data_ptr()))
        k_us, k_n = ctx.timing_get("evalh")
        evalh_kernel_ms = k_us * k_n / 1e3                       # the three launches together
        evalh_program = dict(ctx.evaluate_h_last_stats(), note="the last of three programs (high-degree terms); see split_programs")
    else:
        ph.run("evaluate_h", lambda: dom.evaluate_h(desc, h.data_ptr()))
        evalh_kernel_ms = ctx.timing_get("evalh")[0] / 1e3
        evalh_program = ctx.evaluate_h_last_stats()
    ctx.timing_enable(False)
    ph.run("divide_by_vanishing_poly", lambda: dom.divide_by_vanishing_poly_dev(h.data_ptr()))
    ph.run("extended_to_coeff", lambda: dom.extended_to_coeff_dev(h.data_ptr(), h_coeff.data_ptr()))
    pieces = dom.quotient_poly_degree
    ph.run("h_commit_msm", lambda: ctx.msm_batch_dev(bid, [colptr(h_coeff, i, n) for i in range(pieces)], n))

    # ---- 5. openings ------------------------------------------------------------------------------------
    # one query per (polynomial, rotation): advice columns at the rotation patterns gates use (0 | 0,+1 | 0,-1 | 0,+1,-1),
    # fixed and sigma polynomials at 0, per lookup z at (0,+1), a' at (0,-1), s' at 0, per permutation set z at
    # (0, +1, last row), the random polynomial and the combined h polynomial at 0
    FRP = shplonk.P
    r_inv = pow(1 << 256, FRP - 2, FRP)
    to_int = lambda limbs: sum(int(w) << (64 * j) for j, w in enumerate(limbs)) * r_inv % FRP
    omega, x_int = to_int(dom.constant(0)), to_int(x)
    last_rot = -(cs.blinding_factors() + 1)
    shapes = []
    adv_pat = [(0,), (0, 1), (0, -1), (0, 1, -1)]
    shapes += [(colptr(advice, i, n), adv_pat[i % 4]) for i in range(A)]
    shapes += [(colptr(fixed_ext, i, n), (0,)) for i in range(F)]              # stand-ins of the right size
    shapes += [(colptr(sigma_lagrange, c, n), (0,)) for c in range(len(pcols))]
    for i in range(L):
        shapes += [(colptr(lk_z, i, n), (0, 1)), (colptr(lk_cols, 4 * i + 2, n), (0, -1)), (colptr(lk_cols, 4 * i + 3, n), (0,))]
    shapes += [(colptr(perm_z, s_, n), (0, 1, last_rot)) for s_ in range(sets)]
    shapes += [(colptr(h_coeff, 0, n), (0,)), (colptr(h_coeff, 1, n), (0,))]
    points = {r: x_int * pow(omega, r % n, FRP) % FRP for r in {r for _, rots in shapes for r in rots}}
    point_limbs = {r: np.array(evalh_host.fr_to_mont_limbs(p_), dtype=np.uint64) for r, p_ in points.items()}
    n_queries = sum(len(rots) for _, rots in shapes)
    queries = []

    def openings():
        flat = [(key, ptr, r) for key, (ptr, rots) in enumerate(shapes) for r in rots]
        vals = ctx.eval_polynomials_dev([f[1] for f in flat], np.stack([point_limbs[f[2]] for f in flat]), n)
        for (key, _, r), v in zip(flat, vals):
            queries.append(shplonk.Query(key, points[r], to_int(v)))
    ph.run("eval_polynomial", openings)

    class Transcript:   # the Blake2b transcript stays on the host in the reference; challenges are stand-ins
        def __init__(self):
            self.i = 5
        def squeeze_challenge(self):
            self.i += 1
            return to_int(sc[self.i])
        def write_point(self, jac):
            pass
    polys = {key: ptr for key, (ptr, _) in enumerate(shapes)}
    prover = shplonk.ProverSHPLONK(ctx, bid, n)
    ctx.timing_enable(True)
    ctx.timing_reset()
    outs = ph.run("shplonk_multiopen", lambda: prover.create_proof(polys, queries, Transcript()))
    lincomb_us, lincomb_n = ctx.timing_get("poly_lincomb")
    kate_us, kate_n = ctx.timing_get("kate_division")
    ctx.timing_enable(False)
    for p_ in outs:
        ctx.free(p_)

    total = sum(ph.t.values())
    msm_s = sum(v for kk, v in ph.t.items() if kk.endswith("_msm"))
    n_msm = A + 3 * L + sets + 1 + pieces
    line = {
        "what": "replay of the create_proof polynomial pipeline on synthetic columns (one MI355X); host-side witness "
                "generation and transcript hashing are not included",
        "shape": {"k": k, "extended_k": ek, "advice": A, "fixed": F, "instance": I, "lookups": L,
                  "lookup_width": args.lookup_width, "permutation_columns": len(pcols), "permutation_sets": sets,
                  "degree": d, "gate_polynomials": args.gate_polys,
                  "shape_source": "tests/golden/supercircuit_census.json (static census of the reference's sources, "
                                  "tools/supercircuit_census.py) for every count left at its default; gate_polynomials is an "
                                  "estimate (the census counts create_gate calls: "
                                  f"{census['create_gate_calls']})"},
        "total_s": total, "phases_s": ph.t, "msm_count": n_msm, "msm_total_s": msm_s, "msm_ms_each": msm_s / n_msm * 1e3,
        "coset_ntt_count": A + I + 3 * L + sets, "evalh_kernel_ms": evalh_kernel_ms, "evalh_program": evalh_program,
        "evaluate_h_low_degree_split": {"used": split, "multiplications_per_row": split_programs,
                                        "polynomials_extended_to_the_even_rows_only": half_extended,
                                        "what": "terms of degree <= cosets / 2 evaluated on the even-indexed rows only and "
                                                "extended by two half-size transforms"}, "eval_polynomial_count": n_queries,
        "shplonk": {"polynomials": len(shapes), "queries": n_queries, "poly_lincomb_calls": lincomb_n,
                    "poly_lincomb_avg_us": lincomb_us, "kate_division_calls": kate_n, "kate_division_avg_us": kate_us},
        "host_program_build_s": host_setup_s,
        "device_memory_GB": torch.cuda.max_memory_allocated() / 1e9,
    }
    print(json.dumps(line))
    if args.reference_log:
        from zkevm_circuits_amd import formats
        with open(args.reference_log, "w") as f:
            f.write("\n".join(formats.reference_log_lines(f"replay_k{args.k}", (1 << args.k), args.k,
                                                           {"create_proof": total})) + "\n")
    ctx.release_bases(bid)


if __name__ == "__main__":
    main()
