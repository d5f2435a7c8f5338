"""GPU suite, multi-rank: two processes share cuda:0 (gloo carries the few exchanges) and run the sharded pipeline on the
HIP path; rank 0 compares every result with the unsharded computation.  The partition logic is the one `bench.py --gpus N`
uses over RCCL with one GPU per rank."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dump_stacks_if_stuck(seconds=100):
    """a rank that is still running after `seconds` prints every thread's stack and exits: a hang becomes a failure that
    names its place"""
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(seconds, exit=True, file=sys.stderr)


def _worker(rank, world, port, q):
    import sys
    _dump_stacks_if_stuck()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch
    import zkhip_pkg
    zk = zkhip_pkg.load()
    from zkevm_circuits_amd import evalh_host, ffi, sharding, synthetic
    from util import rand_fr
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = zk.Context(0)
        k, ncols = 10, 5
        n = 1 << k
        rng = np.random.default_rng(4242)                     # the same inputs on every rank
        bases = ctx.g1_multiples(n)
        bid = ctx.register_bases(bases)
        cols = [rand_fr(rng, n) for _ in range(ncols)]
        d_cols = [ctx.upload(c) for c in cols]

        # (1) column-sharded commitments, gathered in column order
        mine = {c: ctx.msm_range_dev(bid, 0, d_cols[c], n) for c in sharding.columns_for_rank(ncols, rank, world)}
        allc = sharding.gather_commitments(mine, ncols, dist)
        # (2) one MSM split by base range
        split = sharding.msm_base_split(n, lambda first, cnt: ctx.msm_range_dev(bid, first, d_cols[0] + 32 * first, cnt),
                                        ctx.g1_sum_jacobian, dist)
        # (3) evaluate_h split by coset, every rank starting from the coefficient form of the columns
        cs = synthetic.synthetic_cs(num_advice=6, num_fixed=3, num_gate_polys=30, num_lookups=2, num_perm_cols=4,
                                    rotations=(-2, -1, 0, 1))
        dom = zk.Domain(ctx, cs.degree(), k)
        e, en = dom.extended_k - k, dom.extended_len
        call = evalh_host.EvalhCall(cs)
        ext_cols = {name: [rand_fr(rng, en) for _ in range(cnt)] for name, cnt in
                    (("fixed", cs.num_fixed_columns), ("sigma", len(cs.permutation_columns)),
                     ("perm_z", cs.num_permutation_sets()), ("l", 3))}
        coeff = {"advice": [rand_fr(rng, n) for _ in range(cs.num_advice_columns)],
                 "instance": [rand_fr(rng, n) for _ in range(cs.num_instance_columns)],
                 "lookups": [tuple(rand_fr(rng, n) for _ in range(3)) for _ in cs.lookups]}
        call.challenges = rand_fr(rng, cs.num_challenges)
        call.y, call.beta, call.gamma, call.theta = (rand_fr(rng, 1)[0] for _ in range(4))
        call.advice, call.instance, call.lookups = coeff["advice"], coeff["instance"], coeff["lookups"]

        def compact(a, cosets):
            out = np.empty((n * len(cosets), 4), dtype=np.uint64)
            for s, j in enumerate(cosets):
                out[s::len(cosets)] = a[j::1 << e]
            return out

        def evaluate(cosets, row_first, row_count):
            if not cosets:
                return torch.empty((0, 4), dtype=torch.int64)
            call.cosets = cosets
            call.rows = (row_first, row_count) if row_count != n * len(cosets) else None
            call.fixed = [compact(a, cosets) for a in ext_cols["fixed"]]
            call.perm_sigma = [compact(a, cosets) for a in ext_cols["sigma"]]
            call.perm_z = [compact(a, cosets) for a in ext_cols["perm_z"]]
            call.l0, call.l_last, call.l_active = (compact(a, cosets) for a in ext_cols["l"])
            part = dom.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_COEFF))
            return torch.from_numpy(part.view(np.int64).copy())
        pieces = sharding.evaluate_h_by_cosets(dom.extended_k, k, evaluate, dist)
        h = np.zeros((en, 4), dtype=np.uint64)
        for cosets, rows in sharding.assemble_cosets(pieces, k):
            arr = rows.numpy().view(np.uint64)
            for s, j in enumerate(cosets):
                h[j::1 << e] = arr[s::len(cosets)]

        if rank == 0:   # the unsharded computation, on this rank alone
            ref_c = np.stack([ctx.g1_to_affine(ctx.msm_range_dev(bid, 0, d_cols[c], n)) for c in range(ncols)])
            got_c = np.stack([ctx.g1_to_affine(j) for j in allc])
            ok_cols = bool(np.array_equal(ref_c, got_c))
            ok_split = bool(np.array_equal(ctx.g1_to_affine(split), ref_c[0]))
            call.cosets = None
            call.fixed, call.perm_sigma, call.perm_z = ext_cols["fixed"], ext_cols["sigma"], ext_cols["perm_z"]
            call.l0, call.l_last, call.l_active = ext_cols["l"]
            whole = dom.evaluate_h(call.build(ffi.MEM_HOST, ffi.FORM_COEFF))
            ok_h = bool(np.array_equal(whole, h))
            q.put((ok_cols, ok_split, ok_h, len(pieces)))
        dist.barrier()
        dom.close()
        ctx.close()
    except Exception:
        import traceback
        q.put(f"rank {rank} failed:\n{traceback.format_exc()}")     # the test fails at once instead of waiting
        raise
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_the_unsharded_pipeline():
    world = 2
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        result = q.get(timeout=240)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert not isinstance(result, str), result
    ok_cols, ok_split, ok_h, owners = result
    assert all(p.exitcode == 0 for p in procs)
    assert ok_cols, "column-sharded commitments differ from the unsharded ones"
    assert ok_split, "base-range split MSM differs from the whole MSM"
    assert ok_h, "coset-split evaluate_h differs from the whole-domain evaluate_h"
    assert owners == 2


def _prover_worker(rank, world, port, q):
    import sys
    _dump_stacks_if_stuck()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch  # noqa: F401
    import zkhip_pkg
    zk = zkhip_pkg.load()
    import coracle
    import prover_cases as pc
    import verifier_model as vm
    from zkevm_circuits_amd import keygen, plonk, prover
    from zkevm_circuits_amd.plonk import Advice, Fixed, Product, Sub
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def small_circuit(n):
        """degree three, i.e. an extended domain of two cosets: four ranks share each coset by row ranges"""
        cs = plonk.ConstraintSystem(num_fixed_columns=1, num_advice_columns=2)
        cs.gates = [[Product(Fixed(0), Sub(Product(Advice(0), Advice(0)), Advice(1, 1)))]]
        cs.permutation_columns = [(plonk.ANY_ADVICE, 0), (plonk.ANY_ADVICE, 1)]
        usable = n - (cs.blinding_factors() + 1)
        a = [(3 * i) % 7 for i in range(usable)]
        a[9] = a[5] * a[5]
        b = [0] + [v * v for v in a[:-1]]
        asm = keygen.Assembly(n, 2)
        asm.copy(1, 6, 0, 9)
        return cs, [a, b], [], [[1] * (usable - 1) + [0] * (n - usable + 1)], asm
    try:
        ctx = zk.Context(0)
        results = []
        cases = [(8, 1, False), (9, 3, True)] if world == 2 else [(9, 1, False), (10, None, None)]
        for k, units, degree9 in cases:
            if units is None:
                cs, advice, instance, fixed, asm = small_circuit(1 << k)
            else:
                cs = pc.wide_circuit(units, degree9, 3)
                advice, instance, fixed, asm = pc.wide_witness(cs, 1 << k, k, units)
            dom = zk.Domain(ctx, cs.degree(), k)
            params = prover.ParamsKZG(ctx, dom, pc.srs(coracle, dom.n))
            # the key of the sharded prover: every rank keeps the key's extended vectors on its own cosets only
            ctx.reset_peak()
            base = ctx.allocated_bytes
            pk_s = prover.keygen(ctx, dom, params, cs, fixed, asm, dist=dist)
            sharded = prover.create_proof(ctx, dom, params, pk_s, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(),
                                          dist=dist)
            peak_sharded = ctx.peak_allocated_bytes - base
            for p in pk_s.pointers():
                ctx.free(p)
            if rank == 0:
                ctx.reset_peak()
                base = ctx.allocated_bytes
                pk = prover.keygen(ctx, dom, params, cs, fixed, asm)
                whole = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite())
                peak_whole = ctx.peak_allocated_bytes - base
                vk = {"fixed_commitments": pk.fixed_commitments, "permutation_commitments": pk.permutation_commitments,
                      "transcript_repr": pk.transcript_repr}
                same_vk = pk.fixed_commitments == pk_s.fixed_commitments and pk.transcript_repr == pk_s.transcript_repr
                # the first 32-byte item of the proof that differs, if any (tells the phase that went wrong)
                differ = [i for i in range(0, max(len(whole), len(sharded)), 32) if whole[i:i + 32] != sharded[i:i + 32]]
                note = f"k={k}: first differing proof item {differ[0] // 32} of {len(whole) // 32}" if differ else ""
                try:
                    vm.verify_proof(cs, k, vk, instance, whole, pc.TAU)
                    vm.verify_proof(cs, k, vk, instance, sharded, pc.TAU)
                except vm.VerifyError as err:
                    note += f"; verifier: {err}"
                results.append((not differ and same_vk and not note, peak_sharded, peak_whole, dom.extended_k - k, note))
                for p in pk.pointers():
                    ctx.free(p)
            dist.barrier()
            params.close()
            dom.close()
        if rank == 0:
            q.put(results)
        dist.barrier()
        ctx.close()
    except Exception:
        import traceback
        q.put(f"rank {rank} failed:\n{traceback.format_exc()}")     # the test fails at once instead of waiting
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_create_the_same_proof_as_one(world):
    """create_proof with the commitments column-sharded and the quotient numerator coset-sharded: the bytes equal the
    single-rank proof and pass the verifier model.  Two ranks: extended domains 4n and 8n, whole cosets per rank.  Four
    ranks: 4n (one coset each) and 2n (two ranks share a coset by row ranges).  Three ranks: the third owns no coset, keeps
    no extended vector (key generation included) and only joins the collectives.  Each rank's peak of resident column data
    (key generation + proof) shrinks with the share of the extended domain it holds."""
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_prover_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = q.get(timeout=300)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert not isinstance(results, str), results
    assert all(p.exitcode == 0 for p in procs)
    assert [r[0] for r in results] == [True, True], [r[4] for r in results]
    for ok, peak_sharded, peak_whole, e, _ in results:
        owners = 1 << (world.bit_length() - 1)             # the largest power of two within the world size
        share = max(1.0 / owners, 1.0 / (1 << e))         # the part of the extended domain rank 0 holds
        # the 2^k-sized polynomials (coefficient and Lagrange forms) stay whole on every rank: they come on top of the
        # rank's share of the extended vectors, and weigh most where the extended domain is only 2n
        assert peak_sharded < peak_whole, (peak_sharded, peak_whole, share)
        if e >= 2:
            assert peak_sharded < (share + 0.3) * peak_whole, (peak_sharded, peak_whole, share)
        print(f"world {world}, extended domain 2^{e} n: peak resident {peak_sharded / 1e6:.1f} MB per rank sharded, "
              f"{peak_whole / 1e6:.1f} MB single rank")


def _rccl_worker(port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch
    import zkhip_pkg
    zk = zkhip_pkg.load()
    from zkevm_circuits_amd import sharding
    from util import rand_fr
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        ctx = zk.Context(0)
        n, ncols = 1 << 10, 3
        rng = np.random.default_rng(7)
        bid = ctx.register_bases(ctx.g1_multiples(n))
        d_cols = [ctx.upload(rand_fr(rng, n)) for _ in range(ncols)]
        single = [ctx.msm_range_dev(bid, 0, d, n) for d in d_cols]
        # the three exchanges of the prover over RCCL: device tensors in, device tensors out
        allc = sharding.gather_commitments({c: single[c] for c in range(ncols)}, ncols, dist)
        split = sharding.msm_base_split(n, lambda first, cnt: ctx.msm_range_dev(bid, first, d_cols[0] + 32 * first, cnt),
                                        ctx.g1_sum_jacobian, dist, device=torch.device("cuda", 0))
        piece = torch.arange(4 * 4 * n, dtype=torch.int64, device="cuda").reshape(-1, 4)
        pieces = sharding.evaluate_h_by_cosets(12, 10, lambda cosets, first, count: piece, dist)
        t = torch.ones(8, dtype=torch.int64, device="cuda")
        dist.broadcast(t, src=0)
        dist.barrier()
        ok = (np.array_equal(allc, np.stack(single)) and np.array_equal(ctx.g1_to_affine(split), ctx.g1_to_affine(single[0]))
              and len(pieces) == 1 and pieces[0][0] == [0, 1, 2, 3] and torch.equal(pieces[0][2], piece)
              and pieces[0][2].is_cuda and dist.get_backend() == "nccl")
        q.put(bool(ok))
        ctx.close()
    except Exception:
        import traceback
        q.put(f"failed:\n{traceback.format_exc()}")
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_backend_carries_the_exchanges_on_one_rank():
    """The prover's exchanges (commitment table, partial sums of a split MSM, pieces of h, a broadcast) through the RCCL
    backend itself, which needs one GPU per rank: a world of one rank on this GPU.  The library is initialised, the
    collectives run on device tensors, and the results equal the local values."""
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    p = mpctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    try:
        result = q.get(timeout=240)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert result is True, result
    assert p.exitcode == 0
