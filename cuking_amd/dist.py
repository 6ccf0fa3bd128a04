"""Pair-space sharding over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" for CPU tests).

Replaces the reference's multi-VM fan-out (cloud_batch_submit.py:45,73;
README.md:94-102), where each of k(k+1)/2 VMs re-reads the whole input and
computes one block.  Here the block's pair space is cut into the tiled
kernel's 64x64-sample tiles (thousands of them), every rank takes one
contiguous, equally sized range of the tile enumeration, and the data path has
exactly two exchange steps:

    1. broadcast of the packed bitset from the rank that built it
       (N * words_per_sample * 8 bytes, once)
    2. gather of the thresholded KingResult records on rank 0
       (all_gather of the counts, then a padded gather of count x 24 bytes)

Every tile costs the same (diagonal tiles are evaluated in full and masked at
emit), so equal tile counts are equal work.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from .api import KING_RESULT_DTYPE, ResourceExhaustedError, sort_results


def tile_partition(num_tiles: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous ranges, sizes differing by at most one tile."""
    base, extra = divmod(num_tiles, world_size)
    out, begin = [], 0
    for r in range(world_size):
        end = begin + base + (1 if r < extra else 0)
        out.append((begin, end))
        begin = end
    return out


def broadcast_bitset(bit_sets, src: int = 0, group=None) -> None:
    """Exchange step 1 (in place)."""
    import torch.distributed as dist
    dist.broadcast(bit_sets, src=src, group=group)


def gather_results(local, count: int, overflow: int, dst: int = 0,
                   group=None) -> Optional[np.ndarray]:
    """Exchange step 2.  ``local`` is this rank's [capacity, 6] int32 tensor of
    KingResult records, the first ``count`` valid.  Returns the sorted records
    of all ranks on ``dst`` (None elsewhere); raises on every rank if any rank
    overflowed (cuking.cu:747-751)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    meta = torch.tensor([count, overflow], dtype=torch.int64,
                        device=local.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    if any(int(m[1]) for m in metas):
        raise ResourceExhaustedError(
            "Could not store all results: try increasing the --max_results "
            "parameter.")
    width = max(max(counts), 1)
    send = torch.zeros((width, 6), dtype=torch.int32, device=local.device)
    send[:count] = local[:count]
    recv = ([torch.zeros_like(send) for _ in range(world)]
            if rank == dst else None)
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [recv[r][:counts[r]].cpu().numpy() for r in range(world)]
    flat = np.ascontiguousarray(np.concatenate(parts, axis=0)).view(np.uint32)
    recs = flat.reshape(-1).view(KING_RESULT_DTYPE).copy()
    return sort_results(recs)


def all_pairs_king(compute_tiles: Callable, num_tiles: int, bit_sets,
                   src: int = 0, dst: int = 0, group=None,
                   broadcast: bool = True):
    """One sharded pass.  ``compute_tiles(bit_sets, tile_begin, tile_end)``
    must return (records tensor [cap, 6] int32, count, overflow) for its tile
    range.  Returns (sorted records on dst else None, (tile_begin, tile_end))."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if broadcast:
        broadcast_bitset(bit_sets, src=src, group=group)
    begin, end = tile_partition(num_tiles, world)[rank]
    local, count, overflow = compute_tiles(bit_sets, begin, end)
    return gather_results(local, count, overflow, dst=dst, group=group), (begin, end)
