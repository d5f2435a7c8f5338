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


def column_blocks(num_columns: int, world: int):
    """Contiguous column ownership: [(first, count)] per rank, sizes differing by at most one.  The prover uses blocks
    rather than the round-robin of `columns_for_rank` where the owners' results are polynomials that every rank needs
    afterwards: a rank's columns are then one contiguous range of the polynomial block, i.e. one broadcast."""
    return [base_range_for_rank(num_columns, r, world) for r in range(world)]


def block_owner(num_columns: int, world: int):
    """owner rank of every column under `column_blocks`"""
    out = []
    for r, (_, count) in enumerate(column_blocks(num_columns, world)):
        out += [r] * count
    return out


def _on_group_device(t, dist):
    """RCCL moves device memory only: the tensor goes to this rank's GPU under the nccl backend"""
    return t.cuda() if dist.get_backend() == "nccl" else t


# Device tensors travel as they are over RCCL.  Under any other backend (gloo: the rehearsal of the multi-rank path on
# fewer GPUs than ranks, and the CPU tests) they are staged through host memory here, explicitly, so that the backend only
# ever sees host tensors: gloo's own handling of device tensors was seen to hang in an all-gather when a third process
# held the same GPU.
def broadcast_(t, src: int, dist):
    """in-place broadcast of a (device or host) tensor from rank `src`"""
    if dist.get_backend() == "nccl" or not t.is_cuda:
        dist.broadcast(t, src=src)
        return
    import torch
    host = t.cpu() if dist.get_rank() == src else torch.empty(t.shape, dtype=t.dtype)
    dist.broadcast(host, src=src)
    if dist.get_rank() != src:
        t.copy_(host)


def all_gather_rows(piece, dist):
    """[tensor of rank 0, tensor of rank 1, ..] of equally shaped pieces, each on the device `piece` lives on"""
    import torch
    world = dist.get_world_size()
    if dist.get_backend() == "nccl" or not piece.is_cuda:
        gathered = [torch.empty_like(piece) for _ in range(world)]
        dist.all_gather(gathered, piece.contiguous())
        return gathered
    host = piece.contiguous().cpu()
    gathered = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(gathered, host)
    return [g.to(piece.device) for g in gathered]


def exchange_runs(block, runs, dist):
    """block: a (rows, 4) tensor that every rank holds; runs[r] = (first row, row count) of the part that rank r has filled
    (count 0: nothing).  Afterwards every rank holds every part: ONE all-gather of the parts, padded to the longest (a ring
    all-gather keeps every xGMI link of every GPU busy at once, where a broadcast per owner is bound by the links of one
    source at a time), then the received parts are copied into place."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    longest = max(count for _, count in runs)
    if longest == 0:
        return
    first, count = runs[rank]
    if count == longest:
        mine = block[first:first + count]
    else:
        mine = torch.zeros((longest,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        if count:
            mine[:count].copy_(block[first:first + count])
    if dist.get_backend() == "nccl":
        out = torch.empty((world * longest,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out, mine.contiguous())
        parts = [out[r * longest:(r + 1) * longest] for r in range(world)]
    else:
        parts = all_gather_rows(mine, dist)
    for r, (f, c) in enumerate(runs):
        if r != rank and c:
            block[f:f + c].copy_(parts[r][:c])


def gather_commitments(local: dict, num_columns: int, dist=None):
    """local: {column index: 12-limb Jacobian} of the columns this rank committed, under any partition in which every
    column has exactly one owner.  Returns the (num_columns, 12) array in column order on every rank (transcript order
    must not depend on the partition): one all-reduce of the table in which every rank has filled in its own rows and
    left the others zero (a sum of one value and zeros; the limbs travel as 32-bit halves in 64-bit lanes, so the sum
    cannot wrap whatever the backend's integer type)."""
    out = np.zeros((num_columns, 12), dtype=np.uint64)
    for c, v in local.items():
        out[c] = v
    if dist is None or not dist.is_initialized():
        return out
    import torch
    halves = out.view(np.uint32).astype(np.int64)                  # (num_columns, 24), every entry below 2^32
    t = _on_group_device(torch.from_numpy(halves), dist)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return np.ascontiguousarray(t.cpu().numpy().astype(np.uint32)).view(np.uint64).reshape(num_columns, 12)


def msm_base_split(n: int, msm_range, fold, dist=None, device=None):
    """One MSM over n bases split by base range.

    msm_range(first, count) -> 12-limb Jacobian partial of this rank's range;
    fold(array of partials (G, 12)) -> 12-limb Jacobian sum.  Every rank returns the full result."""
    if dist is None or not dist.is_initialized():
        return fold(np.asarray(msm_range(0, n), dtype=np.uint64).reshape(1, 12))
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    first, count = base_range_for_rank(n, rank, world)
    part = np.asarray(msm_range(first, count), dtype=np.uint64).reshape(12)
    t = torch.from_numpy(part.view(np.int64).copy())
    if dist.get_backend() == "nccl":
        t = t.to(device) if device is not None else t.cuda()
    gathered = all_gather_rows(t, dist)
    return fold(np.stack([g.cpu().numpy().view(np.uint64) for g in gathered]))


def coset_partition(extended_k: int, k: int, world: int):
    """How `world` ranks share the 2^e cosets of the extended domain (e = extended_k - k), per rank
    (cosets, row_first, row_count) or None for a rank that owns nothing:
      * the owners are the first 2^floor(log2 world) ranks (whole cosets only split evenly over a power of two);
      * up to 2^e owners: owner r holds the cosets r, r + owners, ... whole (row_first 0, all rows of its compact layout);
      * more owners than cosets: owner r holds coset r mod 2^e and evaluates the rows [part * 2^k / s, (part + 1) * 2^k / s)
        of it, part = r div 2^e, s = owners / 2^e.  The owner still holds the whole coset of every column (a rotation by r
        rows of the circuit moves r rows inside the coset), so no neighbouring rows are exchanged.
    Every owner's piece has 2^extended_k / owners rows."""
    if world < 1:
        raise ValueError("world size must be positive")
    e = extended_k - k
    m, n = 1 << e, 1 << k
    owners = 1 << (world.bit_length() - 1)
    owners = min(owners, m * max(1, n // 256))              # pieces of at least one workgroup of rows
    out = []
    for r in range(world):
        if r >= owners:
            out.append(None)
        elif owners <= m:
            cosets = list(range(r, m, owners))
            out.append((cosets, 0, n * len(cosets)))
        else:
            s = owners // m
            out.append(([r % m], (r // m) * (n // s), n // s))
    return out


def evaluate_h_by_cosets(extended_k: int, k: int, evaluate, dist=None):
    """evaluate_h split by coset (and, with more ranks than cosets, by row ranges inside a coset: `coset_partition`).
    evaluate(cosets, row_first, row_count) -> torch tensor (row_count, 4) int64: the rows [row_first, row_first +
    row_count) of h in the compact layout of the listed cosets (compact row q * len(cosets) + s = extended row
    q * 2^e + cosets[s]), on the device the process group communicates on.  Returns, on every rank,
    [(cosets, row_first, rows tensor)] for every owner: one all-gather of the pieces of h (2^extended_k x 32 bytes in
    total), the only exchange this phase needs.  Ranks beyond the largest power of two own nothing and only take part in
    the collective."""
    e = extended_k - k
    if dist is None or not dist.is_initialized():
        cosets = list(range(1 << e))
        return [(cosets, 0, evaluate(cosets, 0, (1 << k) * len(cosets)))]
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    parts = coset_partition(extended_k, k, world)
    rows = next(p[2] for p in parts if p is not None)
    if any(p is not None and p[2] != rows for p in parts):
        raise ValueError("coset partition with unequal pieces")      # same on every rank, before any collective
    mine = parts[rank]
    if mine is not None:
        piece = evaluate(*mine)
        if tuple(piece.shape) != (rows, 4):
            raise ValueError(f"evaluate returned {tuple(piece.shape)}, expected {(rows, 4)}")
    else:
        ref = evaluate([], 0, 0)          # an empty tensor on the right device, for the collective's shape
        piece = torch.zeros((rows, 4), dtype=torch.int64, device=ref.device)
    gathered = all_gather_rows(piece, dist)
    return [(p[0], p[1], gathered[r]) for r, p in enumerate(parts) if p is not None]


def assemble_cosets(pieces, k: int):
    """[(cosets, row_first, rows)] of evaluate_h_by_cosets -> [(cosets, rows of all 2^k * len(cosets) compact rows)]:
    the row ranges of a shared coset concatenated in order"""
    import torch
    n = 1 << k
    by_cosets = {}
    for cosets, first, rows in pieces:
        by_cosets.setdefault(tuple(cosets), []).append((first, rows))
    out = []
    for cosets, parts in by_cosets.items():
        parts.sort(key=lambda t: t[0])
        if len(parts) == 1 and parts[0][1].shape[0] == n * len(cosets):
            out.append((list(cosets), parts[0][1]))
            continue
        at = 0
        for first, rows in parts:
            if first != at:
                raise ValueError("row ranges of a coset do not tile it")
            at += rows.shape[0]
        if at != n * len(cosets):
            raise ValueError("row ranges of a coset do not cover it")
        out.append((list(cosets), torch.cat([rows for _, rows in parts])))
    return out
