"""Pair-space sharding over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" for CPU tests).

Replaces the reference's multi-VM fan-out (cloud_batch_submit.py:45,73;
README.md:94-102), where each of k(k+1)/2 VMs re-reads the whole input and
computes one block.  Here the block's pair space is cut into the tiled
kernel's square sample tiles (128 x 128 for the default matrix-core variant,
`cuking_tile_samples`; thousands of them), every rank takes one
contiguous, equally sized range of the tile enumeration, and the data path has
exactly two exchange steps:

    1. broadcast of the packed bitset from the rank that built it
       (N * words_per_sample * 8 bytes, once)
    2. gather of the thresholded KingResult records on rank 0
       (all_gather of the counts, then a padded gather of count x 24 bytes)

A rank whose local work fails (an exception from the kernel call, an
out-of-memory, ...) must not leave the others waiting inside the gather until
the communicator times out: every rank contributes a status word that rides in
the counts exchange of the gather itself (the header row of the device-side and
pipelined gathers, the meta vector of the host-side one; ``agree_on_status``, a
collective of its own, remains for callers that cannot say which form of gather
their ranks take), and a failure anywhere raises on every rank -- the original
exception on the rank that had it, ``RemoteRankError`` on the others.

Every tile costs the same (diagonal tiles are evaluated in full and masked at
emit), so equal tile counts are equal work.

Two forms of the sharded pass:

``all_pairs_king``         broadcast everything, then each rank runs its range
                           of the tile enumeration (simple; the broadcast is
                           exposed).
``all_pairs_king_staged``  the broadcast is cut into sample chunks sent in
                           ascending order; the ranks share the tile rows
                           round-robin (row r belongs to rank r mod W, so every
                           rank has work from the first chunk on) and, as soon
                           as chunk c has landed, a rank converts it and
                           launches the rectangle (its rows that have arrived)
                           x (chunk c).  A pair (i < j) only needs chunk(j) and
                           everything before it, so every rank computes while
                           later chunks are still on the wire: the exchange
                           hides behind the kernel.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

from .api import KING_RESULT_DTYPE, ResourceExhaustedError, sort_results


class RemoteRankError(RuntimeError):
    """Another rank of the job failed; this rank stops with it."""


def agree_on_status(error: Optional[BaseException] = None, group=None, device=None) -> None:
    """Every rank calls this after its local work with the exception it caught
    (or None).  One all-reduce of a world-sized one-hot vector; if any rank
    reports a failure every rank raises: its own exception where there is one,
    RemoteRankError naming the failed ranks elsewhere.  No-op without a process
    group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if error is not None:
            raise error
        return
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = "cpu" if dist.get_backend(group) == "gloo" or device is None else device
    flags = torch.zeros(world, dtype=torch.int32, device=dev)
    if error is not None:
        flags[rank] = 1
    dist.all_reduce(flags, group=group)
    failed = [r for r, f in enumerate(flags.tolist()) if f]
    if error is not None:
        raise error
    if failed:
        raise RemoteRankError(f"rank(s) {failed} failed; rank {rank} stops with them")


# The schedules themselves -- tile partitions, broadcast chunks, the staged row deal -- are
# ONE implementation, cuking_amd/host/schedule.h, shared with the C++ host `cuking
# --num_gpus=N` and reached from here through the C ABI (cuking_schedule_*).
def _lib():
    from . import _lib as binding
    return binding.load()


def tile_partition(num_tiles: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous ranges, sizes differing by at most one tile."""
    import ctypes as C
    out = (C.c_uint64 * (2 * world_size))()
    _lib().cuking_schedule_tile_partition(num_tiles, world_size, out)
    return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(world_size)]


def weighted_tile_partition(num_tiles: int, weights) -> List[Tuple[int, int]]:
    """Contiguous ranges proportional to ``weights`` (a rank's measured speed in
    tiles per millisecond): the GPUs of one node differ by several percent in the
    clock they sustain under this load, and with equal ranges the slowest one
    sets the pace.  Exact cover of [0, num_tiles), monotone, every weight > 0."""
    import ctypes as C
    w = [float(x) for x in weights]
    if not w or not all(x > 0.0 and x == x and x != float("inf") for x in w):
        raise ValueError("weights must be positive")
    arr = (C.c_double * len(w))(*w)
    out = (C.c_uint64 * (2 * len(w)))()
    if _lib().cuking_schedule_weighted_tile_partition(num_tiles, arr, len(w), out) != 0:
        raise ValueError("weights must be positive")
    return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(len(w))]


def broadcast_bitset(bit_sets, src: int = 0, group=None) -> None:
    """Exchange step 1 (in place)."""
    import torch.distributed as dist
    dist.broadcast(bit_sets, src=src, group=group)


def _raise_if_failed(failed_ranks, rank: int, error: Optional[BaseException]) -> None:
    if error is not None:
        raise error
    if failed_ranks:
        raise RemoteRankError(f"rank(s) {failed_ranks} failed; rank {rank} stops with them")


def gather_results(local, count: int, overflow: int, dst: int = 0,
                   group=None, error: Optional[BaseException] = None,
                   device=None) -> Optional[np.ndarray]:
    """Exchange step 2.  ``local`` is this rank's [capacity, 6] int32 tensor of
    KingResult records, the first ``count`` valid.  Returns the sorted records
    of all ranks on ``dst`` (None elsewhere); raises on every rank if any rank
    overflowed (cuking.cu:747-751).  ``error``: the exception this rank's pass
    raised, if any (``local`` may then be None; ``device`` says where its part of
    the exchange lives): the status rides in the counts exchange, and a failure
    anywhere raises on every rank -- no separate agreement collective."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if local is None:
        local = torch.zeros((1, 6), dtype=torch.int32, device=device or "cpu")
        count = overflow = 0
    meta_dev = "cpu" if dist.get_backend(group) == "gloo" else local.device
    meta = torch.tensor([count, overflow, 0 if error is None else 1], dtype=torch.int64,
                        device=meta_dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    _raise_if_failed([r for r, m in enumerate(metas) if int(m[2])], rank, error)
    counts = [int(m[0]) for m in metas]
    if any(int(m[1]) for m in metas):
        raise ResourceExhaustedError(
            "Could not store all results: try increasing the --max_results "
            "parameter.")
    width = max(max(counts), 1)
    send = torch.zeros((width, 6), dtype=torch.int32, device=local.device)
    send[:count] = local[:count]
    if dist.get_backend(group) == "gloo":
        send = send.cpu()   # gloo gathers host tensors only (CPU rehearsals)
    recv = ([torch.zeros_like(send) for _ in range(world)]
            if rank == dst else None)
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [recv[r][:counts[r]].cpu().numpy() for r in range(world)]
    flat = np.ascontiguousarray(np.concatenate(parts, axis=0)).view(np.uint32)
    recs = flat.reshape(-1).view(KING_RESULT_DTYPE).copy()
    return sort_results(recs)


def gather_results_device(local, index_flag, dst: int = 0, group=None,
                          fast_rows: int = 8192, error: Optional[BaseException] = None,
                          capacity: Optional[int] = None, device=None) -> Optional[np.ndarray]:
    """Exchange step 2 without a host round trip before the collective.
    ``index_flag`` is the kernel's own device tensor [count, overflow] (int32).
    Fast path: ONE all-gather of a fixed block per rank -- a header row with
    the two counters, then the first ``fast_rows`` records -- and one copy to
    the host; every rank sees every count, so all of them agree on whether the
    block was enough.  Otherwise (a rank has more records) the counts are
    already known and a padded gather follows, as in gather_results()."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if local is None:
        # this rank's pass failed (``error``): it still takes part, with an empty
        # block of the shape the others use (``capacity`` = their record buffers')
        local = torch.zeros((capacity, 6), dtype=torch.int32, device=device)
        index_flag = torch.zeros(2, dtype=torch.int32, device=device)
    fast_rows = min(fast_rows, local.shape[0])
    block = torch.empty((fast_rows + 1, 6), dtype=torch.int32, device=local.device)
    block[0, :2] = index_flag
    block[0, 2] = 0 if error is None else 1      # status, read with the counts
    block[1:] = local[:fast_rows]
    parts = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(parts, block, group=group)     # (gloo has no all_gather_into_tensor)
    blocks = torch.stack(parts)
    if rank == dst:
        host = blocks.cpu().numpy()              # the one synchronisation point
        heads = host[:, 0, :3]
    else:
        host = None
        heads = blocks[:, 0, :3].cpu().numpy()
    _raise_if_failed([r for r in range(world) if heads[r, 2]], rank, error)
    counts = [int(c) for c in heads[:, 0]]
    if heads[:, 1].any():
        raise ResourceExhaustedError(
            "Could not store all results: try increasing the --max_results "
            "parameter.")
    if max(counts) <= fast_rows:
        if rank != dst:
            return None
        flat = np.concatenate([host[r, 1:1 + counts[r]] for r in range(world)], axis=0)
    else:
        width = max(counts)
        if width > local.shape[0]:
            raise ResourceExhaustedError("record buffer smaller than another rank's count")
        send = local[:width]
        recv = ([torch.empty_like(send) for _ in range(world)] if rank == dst else None)
        dist.gather(send, recv, dst=dst, group=group)
        if rank != dst:
            return None
        flat = torch.cat([recv[r][:counts[r]] for r in range(world)], dim=0).cpu().numpy()
    recs = np.ascontiguousarray(flat).view(np.uint32).reshape(-1).view(KING_RESULT_DTYPE).copy()
    return sort_results(recs)


class PipelinedGather:
    """gather_results_device() split in two so that the exchange of one pass
    overlaps the kernel of the next (nccl): ``begin`` assembles the block on
    the compute stream (i.e. after the pass's kernel) and starts the all-gather
    asynchronously; ``finish`` -- called while the next pass's kernel runs --
    copies the gathered blocks to the host on a side stream and returns the
    sorted records on ``dst``.  Use two record buffers alternately: a buffer
    may be written again once ``finish`` of its pass has returned."""

    def __init__(self, dst: int = 0, group=None, fast_rows: int = 8192):
        import torch
        self.dst, self.group, self.fast_rows = dst, group, fast_rows
        # (host tensors + gloo in the CPU tests: no streams)
        self.side = torch.cuda.Stream() if torch.cuda.is_available() else None

    def begin(self, local, index_flag, error: Optional[BaseException] = None):
        """``error``: the exception this rank's pass raised, if any -- it still
        takes part in the all-gather (header word 2 = 1) so that ``finish``
        raises on every rank instead of the others waiting for a rank that has
        gone."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        rows = min(self.fast_rows, local.shape[0])
        block = torch.empty((rows + 1, 6), dtype=torch.int32, device=local.device)
        block[0, :2] = index_flag
        block[0, 2] = 0 if error is None else 1
        block[1:] = local[:rows]
        parts = [torch.empty_like(block) for _ in range(world)]
        work = dist.all_gather(parts, block, group=self.group, async_op=True)
        return {"local": local, "rows": rows, "parts": parts, "work": work, "block": block,
                "error": error}

    def finish(self, h):
        import torch
        import torch.distributed as dist
        rank = dist.get_rank(self.group)
        world = dist.get_world_size(self.group)
        if self.side is None:
            h["work"].wait()
            blocks = torch.stack(h["parts"])
            host = blocks if rank == self.dst else blocks[:, 0, :3]
        else:
            with torch.cuda.stream(self.side):
                h["work"].wait()                  # side stream waits for the all-gather
                blocks = torch.stack(h["parts"])
                host = (blocks if rank == self.dst else blocks[:, 0, :3]).to(
                    "cpu", non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.side)
            done.synchronize()                    # not the compute stream
        host = host.numpy()
        heads = host[:, 0, :3] if rank == self.dst else host
        if h["error"] is not None:
            raise h["error"]
        if heads[:, 2].any():
            failed = [r for r in range(world) if heads[r, 2]]
            raise RemoteRankError(f"rank(s) {failed} failed; rank {rank} stops with them")
        counts = [int(c) for c in heads[:, 0]]
        if heads[:, 1].any():
            raise ResourceExhaustedError(
                "Could not store all results: try increasing the --max_results "
                "parameter.")
        if max(counts) > h["rows"]:
            # rare: somebody has more records than the block holds
            return gather_results(h["local"], counts[rank], 0, dst=self.dst, group=self.group)
        if rank != self.dst:
            return None
        flat = np.concatenate([host[r, 1:1 + counts[r]] for r in range(world)], axis=0)
        recs = np.ascontiguousarray(flat).view(np.uint32).reshape(-1).view(
            KING_RESULT_DTYPE).copy()
        return sort_results(recs)


def all_pairs_king(compute_tiles: Callable, num_tiles: int, bit_sets,
                   src: int = 0, dst: int = 0, group=None,
                   broadcast: bool = True, record_capacity: Optional[int] = None,
                   device_counts: Optional[bool] = None):
    """One sharded pass.  ``compute_tiles(bit_sets, tile_begin, tile_end)``
    must return (records tensor [cap, 6] int32, count, overflow) for its tile
    range -- or (records tensor, index_flag device tensor [count, overflow]),
    which skips the host round trip before the gather (nccl only).  Returns
    (sorted records on dst else None, (tile_begin, tile_end))."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if broadcast:
        broadcast_bitset(bit_sets, src=src, group=group)
    begin, end = tile_partition(num_tiles, world)[rank]
    out, error = None, None
    try:
        out = compute_tiles(bit_sets, begin, end)
    except Exception as e:  # noqa: BLE001 - re-raised on every rank by the gather
        error = e
    dev = getattr(bit_sets, "device", None)
    # The status of the pass rides in the gather's counts exchange.  A rank whose
    # pass failed has nothing to show which form of gather the others take, nor
    # the shape of their blocks: the caller says (``device_counts``: compute_tiles
    # returns the kernel's device counters; ``record_capacity``: rows of every
    # rank's record buffer).  Without that the ranks agree in a collective of
    # their own first.
    known = device_counts is not None and (not device_counts or record_capacity is not None)
    if not known:
        agree_on_status(error, group=group, device=dev)
        device_counts = len(out) == 2
    if device_counts:
        local, flag = out if out is not None else (None, None)
        return gather_results_device(local, flag, dst=dst, group=group, error=error,
                                     capacity=record_capacity, device=dev), (begin, end)
    local, count, overflow = out if out is not None else (None, 0, 0)
    return gather_results(local, count, overflow, dst=dst, group=group, error=error,
                          device=dev), (begin, end)


# ---------------------------------------------------------------------------
# Staged (overlapped) form
# ---------------------------------------------------------------------------
def chunk_ranges(num_samples: int, tile: int, num_chunks: int) -> List[Tuple[int, int]]:
    """Ascending, tile-aligned sample chunks covering [0, num_samples)."""
    import ctypes as C
    cap = max(1, num_chunks)
    out = (C.c_uint32 * (2 * cap))()
    n = _lib().cuking_schedule_chunk_ranges(num_samples, tile, num_chunks, out)
    return [(int(out[2 * c]), int(out[2 * c + 1])) for c in range(n)]


def rank_tile_share(num_tile_rows: int, world: int, rank: int) -> float:
    """Fraction of the upper-triangle tiles on rank `rank` when tile rows are
    dealt round-robin (row r holds num_tile_rows - r tiles)."""
    t = num_tile_rows
    mine = sum(t - r for r in range(rank, t, world))
    return mine / (t * (t + 1) / 2) if t else 0.0


def staged_schedule(num_samples: int, tile: int, world: int, rank: int,
                    num_chunks: int):
    """What rank ``rank`` does per broadcast chunk: [((chunk_begin, chunk_end),
    rect or None), ...] with rect = ((row_begin, row_end, row_step), (col_begin,
    col_end)) in samples: the rank's tile rows (rank, rank + world, ...) that lie
    below chunk_end x the chunk's columns.  A pair (i < j) is evaluated with the
    chunk that holds j, by the rank that owns i's tile row."""
    import ctypes as C
    cap = max(1, num_chunks)
    out = (C.c_uint32 * (6 * cap))()
    n = _lib().cuking_schedule_staged_steps(num_samples, tile, world, rank, num_chunks, out)
    steps = []
    for k in range(n):
        c0, c1, has_rect, r0, r1, step = (int(out[6 * k + j]) for j in range(6))
        steps.append(((c0, c1), ((r0, r1, step), (c0, c1)) if has_rect else None))
    return steps


def all_pairs_king_staged(ops, num_samples: int, tile: int, bit_sets,
                          num_chunks: int = 8, src: int = 0, dst: int = 0,
                          group=None):
    """One sharded pass with the bitset exchange overlapped with compute.

    ``bit_sets`` is the [num_samples, words_per_sample] tensor (valid on
    ``src``, receive buffer elsewhere).  ``ops`` supplies the per-rank device
    work (see ``GpuStagedOps``): ``begin()``, ``prepare(s0, s1)``,
    ``compute_rect((r0, r1, step), (c0, c1))`` and ``finish() -> (records, count,
    overflow)``.  Returns (sorted records on dst else None, the rank's schedule)."""
    import torch.distributed as dist
    # With a process group (even of one rank) the collectives are issued, so a
    # single-rank nccl run exercises the same calls as an 8-rank one.
    use_dist = dist.is_available() and dist.is_initialized()
    if use_dist:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    steps = staged_schedule(num_samples, tile, world, rank, num_chunks)

    # (begin() reserves the workspace: device allocations that can fail.  A rank that
    #  fails here still takes part in every broadcast and in the gather, where its
    #  status makes every rank raise -- it must not leave the others waiting.)
    error = None
    try:
        ops.begin()
    except Exception as e:  # noqa: BLE001 - raised on every rank below
        error = e
    # All chunk broadcasts are enqueued up front; they complete in order.
    works = [dist.broadcast(bit_sets[c0:c1], src=src, group=group, async_op=True)
             for (c0, c1), _ in steps] if use_dist else []
    for k, ((c0, c1), rect) in enumerate(steps):
        if use_dist and rank != src:
            works[k].wait()        # nccl: the current stream waits, not the host
        if rect is None or error is not None:
            continue               # never reads these samples / already failed:
        try:                       # keep taking part in the broadcasts
            ops.prepare(c0, c1)
            ops.compute_rect(*rect)
        except Exception as e:  # noqa: BLE001 - raised on every rank below
            error = e
    for w in works:                # the source's sends, too, before reuse
        w.wait()
    local = count = overflow = None
    if error is None:
        try:
            local, count, overflow = ops.finish()
        except Exception as e:  # noqa: BLE001
            error = e
    if not use_dist and error is not None:
        raise error
    if not use_dist:
        if overflow:
            raise ResourceExhaustedError(
                "Could not store all results: try increasing the --max_results "
                "parameter.")
        recs = np.ascontiguousarray(local[:count].cpu().numpy()).view(np.uint32)
        return sort_results(recs.reshape(-1).view(KING_RESULT_DTYPE).copy()), steps
    # (the status of the pass rides in the gather's counts exchange)
    return gather_results(local, count, overflow, dst=dst, group=group, error=error,
                          device=getattr(bit_sets, "device", None)), steps


class GpuStagedOps:
    """Device side of ``all_pairs_king_staged`` on one GPU: layout conversion on
    the current stream, rectangle kernels alternating over side streams (so
    the tail of one launch overlaps the head of the next), results appended
    into one buffer."""

    def __init__(self, ctx, submatrix, words_per_sample: int, bit_sets,
                 kin_threshold: float, max_results: int, num_streams: int = 3):
        import torch
        self.ctx, self.sm, self.wps, self.bits = ctx, submatrix, words_per_sample, bit_sets
        self.thr, self.max_results = kin_threshold, max_results
        dev = bit_sets.device
        self.results = torch.zeros((max(max_results, 1), 6), dtype=torch.int32, device=dev)
        self.index_flag = torch.zeros(2, dtype=torch.int32, device=dev)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(num_streams)]
        self.launches = 0

    def begin(self):
        # Workspace (kernel layout, tile prefix, one split slab per stream) sized
        # before anything is enqueued: no allocation, no device-wide wait once a
        # broadcast is in flight (cuking_ctx_reserve).
        if not getattr(self, "_reserved", False):
            import torch
            # (this schedule's rectangles cover every prepared chunk in a union over the
            #  ranks: the chunks may be laid out sorted by missing share -- library option
            #  "filter_sort" 2; a host with rectangles of its own keeps the default)
            self._sort_before = self.ctx.get_option("filter_sort")
            if self._sort_before == 1:
                self.ctx.set_option("filter_sort", 2)
            self.ctx.reserve(self.sm, self.wps, [torch.cuda.current_stream()] + self.streams)
            self._reserved = True
            self._at_reserve = (self.ctx.get_option("workspace_allocations"),
                                self.ctx.get_option("host_syncs"))
        elif getattr(self, "_sort_before", None) == 1 and self.ctx.get_option("filter_sort") == 1:
            self.ctx.set_option("filter_sort", 2)
        self.index_flag.zero_()
        self.launches = 0

    def after_reserve(self):
        """(device allocations, host-side waits) the library has made for this context
        since begin() reserved its workspace -- i.e. while broadcasts may have been in
        flight: both must read 0."""
        a0, s0 = getattr(self, "_at_reserve", (0, 0))
        return (self.ctx.get_option("workspace_allocations") - a0,
                self.ctx.get_option("host_syncs") - s0)

    def prepare(self, s0: int, s1: int):
        self.ctx.prepare_samples(self.sm, self.wps, self.bits, s0, s1)

    def compute_rect(self, rows, cols):
        import torch
        if rows[0] >= rows[1] or cols[0] >= cols[1]:
            return
        ready = torch.cuda.Event()
        ready.record()                       # after zeroing + every prepare so far
        stream = self.streams[self.launches % len(self.streams)]
        stream.wait_event(ready)
        self.ctx.compute_king_rect(self.sm, self.wps, self.bits, rows, cols, self.thr,
                                   self.max_results, self.results,
                                   self.index_flag[0:1], self.index_flag[1:2],
                                   stream=stream)
        self.launches += 1

    def finish(self):
        import torch
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
        count, overflow = (int(x) & 0xFFFFFFFF for x in self.index_flag.tolist())   # (waits)
        if getattr(self, "_sort_before", None) == 1 and self.ctx.get_option("filter_sort") == 2:
            # (the context goes back to its caller's setting; a later begin() sets it again)
            self.ctx.set_option("filter_sort", 1)
            self._reserved_sort_reset = True
        return self.results, min(count, self.max_results), int(overflow)
