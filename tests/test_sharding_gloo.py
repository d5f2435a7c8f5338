"""CPU suite, part 4: the N > 1 orchestration on world_size 2 over gloo.  The compute callables are the C oracle
here (the checker stands in for the device so the exchange logic can run without a GPU)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ncols, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import zkhip_pkg
    zkhip_pkg.load()
    import coracle as co
    from zkevm_circuits_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)                      # same inputs on every rank
    bases = co.g1_bases(n)
    scal = rng.integers(0, 2 ** 64, size=(ncols, n, 4), dtype=np.uint64)
    scal[..., 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    # (1) one MSM split by base range, all-gather of partials + local fold
    full = sharding.msm_base_split(
        n, lambda first, cnt: co.best_multiexp(scal[0][first:first + cnt], bases[first:first + cnt]),
        co.g1_sum_jacobian, dist)
    # (2) column-sharded commitments gathered in column order
    mine = {c: co.best_multiexp(scal[c], bases) for c in sharding.columns_for_rank(ncols, rank, world)}
    allc = sharding.gather_commitments(mine, ncols, dist)
    # (3) evaluate_h split by coset: every rank evaluates its cosets (here: reads them out of a stand-in h that all
    # ranks derive from the same seed), one all-gather, and the pieces are put back in extended order.  Two shapes:
    # four cosets (whole cosets per rank up to four ranks) and two cosets of 2^9 rows (ranks beyond two share a coset by
    # row ranges); ranks past the largest power of two own nothing and still take part in the collective.
    import torch
    for k, ek in ((4, 6), (9, 10)):
        h_full = np.random.default_rng(7).integers(0, 2 ** 63, size=(1 << ek, 4), dtype=np.int64)

        def evaluate(cosets, row_first, row_count):
            m = len(cosets)
            out = np.zeros(((1 << k) * m, 4), dtype=np.int64)
            for s_, j in enumerate(cosets):
                out[s_::m] = h_full[j::1 << (ek - k)]
            return torch.from_numpy(out[row_first:row_first + row_count].copy())
        pieces = sharding.evaluate_h_by_cosets(ek, k, evaluate, dist)
        owners = sum(p is not None for p in sharding.coset_partition(ek, k, world))
        assert len(pieces) == owners and all(rows.shape[0] == (1 << ek) // owners for _, _, rows in pieces)
        h_back = np.zeros_like(h_full)
        whole = sharding.assemble_cosets(pieces, k)
        for cosets, rows in whole:
            assert rows.shape[0] == (1 << k) * len(cosets)
            for s_, j in enumerate(cosets):
                h_back[j::1 << (ek - k)] = rows.numpy()[s_::len(cosets)]
        assert np.array_equal(h_back, h_full)
        assert sorted(j for cosets, _ in whole for j in cosets) == list(range(1 << (ek - k)))
    # (4) a block of polynomials of which every rank has filled its contiguous share (uneven shares, one rank with
    # nothing when there are more ranks than columns): one padded all-gather, afterwards every rank holds the whole block
    for cols_ in (ncols, world + 1, world - 1):
        rows_per = 6
        whole_block = torch.from_numpy(np.random.default_rng(5).integers(0, 2 ** 63, size=(cols_ * rows_per, 4), dtype=np.int64))
        runs = [(f * rows_per, c * rows_per) for f, c in sharding.column_blocks(cols_, world)]
        blk = torch.zeros_like(whole_block)
        f, c = runs[rank]
        blk[f:f + c] = whole_block[f:f + c]
        sharding.exchange_runs(blk, runs, dist)
        assert torch.equal(blk, whole_block), (cols_, rank)
    q.put((rank, co.g1_to_affine(full), np.stack([co.g1_to_affine(j) for j in allc])))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3, 4])
def test_world_matches_single_process(co, world):
    """world 2: whole cosets per rank; world 3: the third rank owns no coset and the job neither fails nor hangs; world 4:
    two ranks share each of the two cosets of the second shape by row ranges"""
    n, ncols = 200, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, ncols, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(99)
    bases = co.g1_bases(n)
    scal = rng.integers(0, 2 ** 64, size=(ncols, n, 4), dtype=np.uint64)
    scal[..., 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    ref_full = co.g1_to_affine(co.best_multiexp(scal[0], bases))
    ref_cols = np.stack([co.g1_to_affine(co.best_multiexp(scal[c], bases)) for c in range(ncols)])
    for rank, full, cols in results:
        assert np.array_equal(full, ref_full), rank
        assert np.array_equal(cols, ref_cols), rank
