"""ttcr_amd.dist -- source sharding across the GPUs of one node (one process per GPU).

The reference's only parallelism is "independent sources over threads"
(ttcr/Grid3D.h:810-853, block sizes from get_blk_size :451-465).  Here the same block
distribution shards the unique sources over the ranks of a torch.distributed group
(backend "nccl" == RCCL over xGMI on MI355X; "gloo" in the CPU tests):

  * the slowness model is broadcast once from rank 0 (`broadcast_slowness`),
  * every rank solves its block of sources locally -- no collective during a solve,
  * receiver traveltimes are gathered to rank 0 (`raytrace_sharded`) -- a few KB.

`solve_fn(src_rows, rcv_rows) -> tt` is the local solver; by default it is the grid's own
`raytrace` (HIP path).  The CPU tests inject the oracle there to exercise the sharding /
gather logic without a GPU; the product path never does.
"""
import numpy as np


def blk_sizes(n_src, n_workers):
    """get_blk_size (ttcr/Grid3D.h:451-465): sizes of the contiguous blocks, the first
    blocks get the remainder."""
    n_blk = min(n_workers, n_src)
    if n_blk <= 0:
        return []
    base, rem = divmod(n_src, n_blk)
    return [base + (1 if b < rem else 0) for b in range(n_blk)]


def shard_bounds(n_src, world, rank):
    """[start, end) of the sources owned by `rank`."""
    sizes = blk_sizes(n_src, world)
    sizes += [0] * (world - len(sizes))
    start = int(np.sum(sizes[:rank]))
    return start, start + sizes[rank]


def unique_sources(source):
    """Unique source rows in first-occurrence order (rgrid.pyx:926-938) and, per input row,
    the index of its unique source."""
    source = np.asarray(source)
    _, first, inv = np.unique(source, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)  # unique ids sorted by first occurrence
    rank_of = np.empty_like(order)
    rank_of[order] = np.arange(order.size)
    return source[np.sort(first)], rank_of[np.asarray(inv).ravel()]


def broadcast_slowness(tensor, group=None, src=0, always=False):
    """Broadcast the slowness tensor (already allocated on every rank) from `src`.  always: issue the collective in a group of
    one rank as well (a single-GPU box can then load and run the backend -- RCCL -- end to end: tests, bench.py --force-dist)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and (always or dist.get_world_size(group) > 1):
        dist.broadcast(tensor, src=src, group=group)
    return tensor


def raytrace_sharded(source, rcv, solve_fn, group=None, device=None, dtype=np.float64, always_gather=False):
    """Data-parallel `raytrace(source, rcv)` over the ranks of `group`.

    source/rcv follow the ttcrpy pair convention (one row per datum, equal row counts).
    Each rank solves the rows whose unique source falls in its block and the result is
    gathered on rank 0, which returns tt in the input row order (other ranks return None).
    always_gather: go through the collective in a group of one rank as well (see broadcast_slowness).
    """
    import torch
    import torch.distributed as dist

    source = np.asarray(source)
    rcv = np.asarray(rcv)
    if source.shape[0] != rcv.shape[0]:
        raise ValueError('src and rcv should be of equal size')
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    uniq, which = unique_sources(source)
    lo, hi = shard_bounds(uniq.shape[0], world, rank)
    mine = np.nonzero((which >= lo) & (which < hi))[0]
    tt_local = np.zeros(0, dtype=dtype)
    if mine.size:
        tt_local = np.asarray(solve_fn(source[mine], rcv[mine]), dtype=dtype)
    if world == 1 and not (always_gather and dist.is_initialized()):
        out = np.zeros(source.shape[0], dtype=dtype)
        out[mine] = tt_local
        return out
    # fixed-size gather: every rank contributes a buffer of the full row count (KB-sized)
    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    buf = torch.zeros(source.shape[0], dtype=tdt, device=device)
    if mine.size:
        buf[torch.as_tensor(mine, device=device)] = torch.as_tensor(tt_local, dtype=tdt, device=device)
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)   # KB-sized; the collective every backend implements natively
    if rank != 0:
        return None
    out = np.zeros(source.shape[0], dtype=dtype)
    for r in range(world):
        l2, h2 = shard_bounds(uniq.shape[0], world, r)
        rows = np.nonzero((which >= l2) & (which < h2))[0]
        out[rows] = gathered[r].cpu().numpy()[rows]
    return out
