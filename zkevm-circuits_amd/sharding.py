"""Multi-GPU partitioning of the proving pipeline: one process per GPU, torch.distributed (RCCL) for the few
exchanges the path needs.

* Columns (advice / lookup / permutation polynomials) are independent units of iNTT, coset NTT and commit:
  column c belongs to rank c mod world.  No data-path collective; commitments are gathered in column order.
* One large MSM splits its base range: rank g owns bases [g*n/G, (g+1)*n/G) and produces a Jacobian partial
  sum.  G1 addition is not a reduction operator of the collective library, so the "reduce" is an all-gather of
  G x 96 bytes followed by a local fold.
* evaluate_h splits by coset: extended rows congruent to j mod 2^e form the coset zeta*w_ext^j*<omega>, and a
  rotation by r rows moves by r*2^e extended rows, i.e. stays inside the coset, so a rank that owns whole
  cosets needs no halo exchange.

The compute callables are injected so the same orchestration runs on the HIP path and, in CPU tests, on a
checker.
"""
from __future__ import annotations

import numpy as np


def columns_for_rank(num_columns: int, rank: int, world: int):
    """Round-robin column ownership (keeps per-rank work balanced when column counts are not multiples)."""
    return list(range(rank, num_columns, world))


def base_range_for_rank(n: int, rank: int, world: int):
    """Contiguous base range [first, first + count) of rank `rank`; the remainder goes to the low ranks."""
    q, r = divmod(n, world)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def cosets_for_rank(extended_k: int, k: int, rank: int, world: int):
    """Coset indices j in [0, 2^(extended_k-k)) owned by `rank` (row idx belongs to coset idx mod 2^e)."""
    m = 1 << (extended_k - k)
    return list(range(rank, m, world))


def gather_commitments(local: dict, num_columns: int, dist=None):
    """local: {column index: 12-limb Jacobian}.  Returns the (num_columns, 12) array in column order on every
    rank (transcript order must not depend on the partition)."""
    out = np.zeros((num_columns, 12), dtype=np.uint64)
    for c, v in local.items():
        out[c] = v
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return out
    import torch
    world = dist.get_world_size()
    mine = torch.from_numpy(out.view(np.int64).copy())
    if dist.get_backend() == "nccl":            # RCCL moves device memory only
        mine = mine.cuda()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    res = np.zeros_like(out)
    for r, g in enumerate(gathered):
        arr = g.cpu().numpy().view(np.uint64)
        for c in columns_for_rank(num_columns, r, world):
            res[c] = arr[c]
    return res


def msm_base_split(n: int, msm_range, fold, dist=None, device=None):
    """One MSM over n bases split by base range.

    msm_range(first, count) -> 12-limb Jacobian partial of this rank's range;
    fold(array of partials (G, 12)) -> 12-limb Jacobian sum.  Every rank returns the full result."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return fold(np.asarray(msm_range(0, n), dtype=np.uint64).reshape(1, 12))
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    first, count = base_range_for_rank(n, rank, world)
    part = np.asarray(msm_range(first, count), dtype=np.uint64).reshape(12)
    t = torch.from_numpy(part.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    parts = np.stack([g.cpu().numpy().view(np.uint64) for g in gathered])
    return fold(parts)


def evaluate_h_by_cosets(extended_k: int, k: int, evaluate, dist=None):
    """evaluate_h split by coset.  evaluate(cosets) -> torch tensor (2^k * len(cosets), 4) int64: the compact rows of
    h on the listed cosets (compact row q * len(cosets) + s = extended row q * 2^e + cosets[s]), on the device the
    process group communicates on.  Returns [(cosets of rank r, its rows)] for every rank that owns cosets, on every
    rank: one all-gather of the pieces of h (2^extended_k x 32 bytes in total), the only exchange this phase needs.
    With more ranks than cosets the surplus ranks own nothing (splitting a coset further by row ranges would need the
    rotation halos this partition exists to avoid)."""
    e = extended_k - k
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        cosets = list(range(1 << e))
        return [(cosets, evaluate(cosets))]
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    owners = min(world, 1 << e)
    mine = cosets_for_rank(extended_k, k, rank, owners) if rank < owners else []
    per = (1 << e) // owners
    rows = (1 << k) * per
    if mine:
        piece = evaluate(mine)
        assert tuple(piece.shape) == (rows, 4), (tuple(piece.shape), rows)
    else:
        ref = evaluate([])            # an empty tensor on the right device, for the collective's shape
        piece = torch.zeros((rows, 4), dtype=torch.int64, device=ref.device)
    gathered = [torch.empty_like(piece) for _ in range(world)]
    dist.all_gather(gathered, piece.contiguous())
    return [(cosets_for_rank(extended_k, k, r, owners), gathered[r]) for r in range(owners)]
