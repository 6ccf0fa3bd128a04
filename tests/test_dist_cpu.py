"""world_size-2/3 gloo runs of the multi-GPU orchestration (cuking_amd/dist.py)
on CPU: bitset broadcast from rank 0, contiguous tile ranges per rank, gather
of variable-length record lists on rank 0.  The per-rank kernel call needs a
GPU, so here the checker (oracle) stands in for it, driven by the product's
own tile enumeration (cuking_tile_bounds): what is under test is the
partitioning and the two exchange steps, not the arithmetic."""
import ctypes as C
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleStagedOps:
    """Stand-in for GpuStagedOps: the oracle evaluates each rectangle, and only
    from samples that have been 'prepared' (i.e. have arrived)."""

    def __init__(self, pyoracle, bits, thr, max_results):
        self.o, self.bits, self.thr, self.max_results = pyoracle, bits, thr, max_results

    def begin(self):
        self.recs, self.prepared = [], np.zeros(self.bits.shape[0], dtype=bool)
        self.snapshot = np.zeros_like(self.bits.numpy().view(np.uint64))

    def prepare(self, s0, s1):
        self.snapshot[s0:s1] = self.bits.numpy().view(np.uint64)[s0:s1]
        self.prepared[s0:s1] = True

    def compute_rect(self, rows, cols):
        (r0, r1, step), (c0, c1) = rows, cols
        tile = 64
        assert step % tile == 0 and r0 % tile == 0
        assert self.prepared[c0:c1].all()
        for a in range(r0, r1, step):          # the rank's tile rows
            b = min(a + tile, self.bits.shape[0])
            assert self.prepared[a:b].all()
            lo, hi = min(a, c0), max(b, c1)
            osm = self.o.Submatrix(lo, hi, lo, hi)
            r, _, _ = self.o.compute(osm, np.ascontiguousarray(self.snapshot[lo:hi]), self.thr)
            keep = ((r["sample_i"] >= a) & (r["sample_i"] < b) &
                    (r["sample_j"] >= c0) & (r["sample_j"] < c1))
            self.recs.append(r[keep])

    def finish(self):
        recs = (np.concatenate(self.recs) if self.recs
                else np.zeros(0, dtype=self.o.RESULT_DTYPE))
        count = len(recs)
        overflow = int(count > self.max_results)
        keep = min(count, self.max_results)
        buf = torch.zeros((max(self.max_results, 1), 6), dtype=torch.int32)
        if keep:
            buf[:keep] = torch.from_numpy(
                recs[:keep].view(np.uint32).reshape(-1, 6).view(np.int32).copy())
        return buf, keep, overflow


def _staged_worker(rank, world, port, n, m, thr, chunks, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cuking_amd
    from cuking_amd.dist import all_pairs_king_staged
    from oracle import pyoracle
    from conftest import random_genotypes
    wps = cuking_amd.words_per_sample(m)
    bits = torch.zeros((n, wps), dtype=torch.int64)
    if rank == 0:
        geno = random_genotypes(np.random.default_rng(2), n, m, missing=0.05)
        geno[n - 1] = geno[0]
        geno[n // 3] = geno[n // 2]
        bits.copy_(torch.from_numpy(pyoracle.bitset_from_genotypes(geno).view(np.int64)))
    ops = OracleStagedOps(pyoracle, bits, thr, 100000)
    merged, _ = all_pairs_king_staged(ops, n, 64, bits, num_chunks=chunks)
    if rank == 0:
        host = np.ascontiguousarray(bits.numpy().view(np.uint64))
        exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), host, thr)
        Path(out_path).write_text(f"{int(merged.tobytes() == exp.tobytes())} {len(exp)}")
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, n, m, thr, max_results, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cuking_amd
    from cuking_amd import _lib
    from cuking_amd.dist import all_pairs_king
    from oracle import pyoracle
    from conftest import random_genotypes

    lib = _lib.load()
    sm = cuking_amd.Submatrix(n)
    wps = cuking_amd.words_per_sample(m)
    bits = torch.zeros((n, wps), dtype=torch.int64)
    if rank == 0:  # only the source rank has the packed input
        geno = random_genotypes(np.random.default_rng(1), n, m, missing=0.05)
        geno[n - 1] = geno[0]
        geno[n // 2] = geno[1]
        bits.copy_(torch.from_numpy(pyoracle.bitset_from_genotypes(geno).view(np.int64)))
    num_tiles = lib.cuking_num_tiles(None, C.byref(sm.c))

    def compute_tiles(bit_sets, begin, end):
        host = np.ascontiguousarray(bit_sets.numpy().view(np.uint64))
        recs = []
        rb, re_, cb, ce = (C.c_uint32() for _ in range(4))
        for t in range(begin, end):
            _lib.check(lib.cuking_tile_bounds(None, C.byref(sm.c), t, C.byref(rb),
                                              C.byref(re_), C.byref(cb), C.byref(ce)))
            if (rb.value, re_.value) == (cb.value, ce.value):
                osm = pyoracle.Submatrix(rb.value, re_.value, cb.value, ce.value)
                sub = host[rb.value:re_.value]
            else:
                osm = pyoracle.Submatrix(rb.value, re_.value, cb.value, ce.value)
                sub = np.concatenate([host[rb.value:re_.value], host[cb.value:ce.value]])
            r, _, _ = pyoracle.compute(osm, np.ascontiguousarray(sub), thr)
            recs.append(r)
        recs = np.concatenate(recs) if recs else np.zeros(0, dtype=pyoracle.RESULT_DTYPE)
        count = len(recs)
        overflow = int(count > max_results)
        buf = torch.zeros((max(max_results, 1), 6), dtype=torch.int32)
        keep = min(count, max_results)
        if keep:
            buf[:keep] = torch.from_numpy(
                recs[:keep].view(np.uint32).reshape(-1, 6).view(np.int32).copy())
        return buf, keep if not overflow else max_results, overflow

    try:
        merged, (b, e) = all_pairs_king(compute_tiles, num_tiles, bits)
        status = "ok"
    except cuking_amd.ResourceExhaustedError:
        merged, status = None, "overflow"
    if rank == 0:
        if status == "ok":
            host = np.ascontiguousarray(bits.numpy().view(np.uint64))
            exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), host, thr)
            same = merged.tobytes() == exp.tobytes()
            Path(out_path).write_text(f"{status} {int(same)} {len(exp)} {num_tiles}")
        else:
            Path(out_path).write_text(f"{status} 0 0 {num_tiles}")
    else:
        assert merged is None
        # non-source ranks really received the bitset
        assert status == "overflow" or bool((bits != 0).any())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pass_equals_single_pass(tmp_path, world):
    out = tmp_path / "result.txt"
    # (600 samples: three rows of the default variant's 256-sample tiles)
    mp.spawn(_worker, args=(world, _free_port(), 600, 300, 0.02, 100000, str(out)),
             nprocs=world, join=True)
    status, same, n_exp, tiles = out.read_text().split()
    assert status == "ok" and same == "1"
    assert int(n_exp) > 50 and int(tiles) >= world


def test_overflow_on_any_rank_fails_everywhere(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), 200, 300, -5.0, 50, str(out)),
             nprocs=2, join=True)
    assert out.read_text().split()[0] == "overflow"


@pytest.mark.parametrize("world,chunks", [(2, 3), (3, 5)])
def test_staged_overlapped_pass_equals_single_pass(tmp_path, world, chunks):
    """Chunked broadcast + row bands + rectangles-as-chunks-arrive (gloo)."""
    out = tmp_path / "result.txt"
    mp.spawn(_staged_worker, args=(world, _free_port(), 330, 200, -0.1, chunks, str(out)),
             nprocs=world, join=True)
    same, n_exp = out.read_text().split()
    assert same == "1" and int(n_exp) > 100


@pytest.mark.parametrize("n,tile,world,chunks", [(300, 64, 3, 4), (1000, 64, 8, 8),
                                                 (130, 64, 2, 8), (64, 64, 2, 3),
                                                 (2000, 128, 4, 5), (65, 64, 8, 8)])
def test_staged_schedule_covers_every_pair_once(n, tile, world, chunks):
    from cuking_amd.dist import rank_tile_share, staged_schedule
    cover = np.zeros((n, n), dtype=np.int32)
    t = (n + tile - 1) // tile
    for r in range(world):
        for (c0, c1), rect in staged_schedule(n, tile, world, r, chunks):
            if rect is None:
                continue
            (r0, r1, step), (q0, q1) = rect
            assert (q0, q1) == (c0, c1) and r1 <= c1     # only samples that have arrived
            assert r0 == r * tile and step == world * tile
            for a in range(r0, r1, step):
                b = min(a + tile, n)
                I, J = np.meshgrid(np.arange(a, b), np.arange(q0, q1), indexing="ij")
                cover[I[I < J], J[I < J]] += 1
    assert np.all(cover[np.triu_indices(n, 1)] == 1)
    assert cover.sum() == n * (n - 1) // 2
    shares = [rank_tile_share(t, world, r) for r in range(world)]
    assert abs(sum(shares) - 1.0) < 1e-12
    if t >= 16 * world:
        assert max(shares) <= 1.1 / world


def _device_gather_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuking_amd.api import ResourceExhaustedError
    from cuking_amd.dist import gather_results, gather_results_device
    ok = True
    for fast_rows in (4, 64):                      # fallback gather / one-collective path
        count = 10 + 7 * rank
        rng = np.random.default_rng(100 + rank)
        local = torch.from_numpy(rng.integers(0, 1000, size=(100, 6)).astype(np.int32))
        flag = torch.tensor([count, 0], dtype=torch.int32)
        a = gather_results_device(local, flag, fast_rows=fast_rows)
        b = gather_results(local, count, 0)
        if rank == 0:
            ok &= a.tobytes() == b.tobytes() and len(a) == sum(10 + 7 * r for r in range(world))
        else:
            ok &= a is None and b is None
    # an overflow flag on any rank fails every rank
    flag = torch.tensor([3, 1 if rank == world - 1 else 0], dtype=torch.int32)
    try:
        gather_results_device(local, flag)
        ok = False
    except ResourceExhaustedError:
        pass
    Path(f"{out_path}.{rank}").write_text("ok" if ok else "bad")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_device_side_gather_matches_host_gather(tmp_path, world):
    """gather_results_device (counters taken from the kernel's own device
    tensor, one all-gather on the fast path) == gather_results."""
    out = tmp_path / "res"
    mp.spawn(_device_gather_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    assert all(Path(f"{out}.{r}").read_text() == "ok" for r in range(world))


def _failure_worker(rank, world, port, mode, out_path):
    """Rank `world - 1` fails in its local work; nobody may hang in a collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuking_amd.dist import (PipelinedGather, RemoteRankError, all_pairs_king,
                                 all_pairs_king_staged)
    bad = rank == world - 1
    bits = torch.zeros((200, 4), dtype=torch.int64)
    local = torch.zeros((16, 6), dtype=torch.int32)
    outcome = "returned"
    try:
        if mode == "simple":
            def compute_tiles(b, begin, end):
                if bad:
                    raise RuntimeError("injected kernel failure")
                return local, 0, 0
            all_pairs_king(compute_tiles, 10, bits)
        elif mode == "simple_folded":      # status inside the host-side gather's counts
            def compute_tiles(b, begin, end):
                if bad:
                    raise RuntimeError("injected kernel failure")
                return local, 0, 0
            all_pairs_king(compute_tiles, 10, bits, device_counts=False)
        elif mode == "device_folded":      # ... inside the device-side gather's header row
            def compute_tiles(b, begin, end):
                if bad:
                    raise RuntimeError("injected kernel failure")
                return local, torch.tensor([2, 0], dtype=torch.int32)
            all_pairs_king(compute_tiles, 10, bits, device_counts=True, record_capacity=16)
        elif mode == "staged":
            class Ops:
                def begin(self): pass
                def prepare(self, s0, s1):
                    if bad and s0 > 0:
                        raise RuntimeError("injected kernel failure")
                def compute_rect(self, rows, cols): pass
                def finish(self): return local, 0, 0
            all_pairs_king_staged(Ops(), 200, 64, bits, num_chunks=3)
        elif mode == "staged_begin":       # the workspace reservation itself fails
            class Ops:
                def begin(self):
                    if bad:
                        raise RuntimeError("injected kernel failure")
                def prepare(self, s0, s1): pass
                def compute_rect(self, rows, cols): pass
                def finish(self): return local, 0, 0
            all_pairs_king_staged(Ops(), 200, 64, bits, num_chunks=3)
        else:
            pipe = PipelinedGather(fast_rows=8)
            flag = torch.tensor([2, 0], dtype=torch.int32)
            err = RuntimeError("injected kernel failure") if bad else None
            pipe.finish(pipe.begin(local, flag, error=err))
    except RemoteRankError as e:
        outcome = f"remote:{e}"
    except RuntimeError as e:
        outcome = f"own:{e}"
    Path(f"{out_path}.{rank}").write_text(outcome)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["simple", "simple_folded", "device_folded", "staged",
                                  "staged_begin", "pipelined"])
@pytest.mark.timeout(120)
def test_failure_on_one_rank_raises_on_all(tmp_path, mode):
    """SURVEY section 5 'per-rank error -> abort all ranks': an exception on one
    rank before the gather must surface on every rank, promptly."""
    world = 3
    out = tmp_path / "res"
    mp.spawn(_failure_worker, args=(world, _free_port(), mode, str(out)), nprocs=world,
             join=True)
    got = [Path(f"{out}.{r}").read_text() for r in range(world)]
    assert got[world - 1] == "own:injected kernel failure"
    for r in range(world - 1):
        assert got[r].startswith("remote:") and f"[{world - 1}]" in got[r], got


def _status_ok_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuking_amd.dist import PipelinedGather, agree_on_status
    agree_on_status(None)                                  # nobody failed: returns
    pipe = PipelinedGather(fast_rows=8)
    local = torch.arange(96, dtype=torch.int32).reshape(16, 6) + 1000 * rank
    recs = pipe.finish(pipe.begin(local, torch.tensor([3 + rank, 0], dtype=torch.int32)))
    ok = (recs is None) if rank else (len(recs) == sum(3 + r for r in range(world)))
    Path(f"{out_path}.{rank}").write_text("ok" if ok else "bad")
    dist.destroy_process_group()


def test_pipelined_gather_on_host_tensors(tmp_path):
    out = tmp_path / "res"
    mp.spawn(_status_ok_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert all(Path(f"{out}.{r}").read_text() == "ok" for r in range(2))


def test_weighted_tile_partition():
    from cuking_amd.dist import tile_partition, weighted_tile_partition
    for tiles in (0, 1, 7, 1000, 2_747_896):
        for w in ([1.0], [1, 1, 1], [1.0, 0.9, 1.1, 1.0], [5, 1], [1.0] * 8,
                  [0.97, 1.0, 1.02, 0.95, 1.0, 1.01, 0.99, 1.03]):
            parts = weighted_tile_partition(tiles, w)
            assert len(parts) == len(w) and parts[0][0] == 0 and parts[-1][1] == tiles
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert all(b <= e for b, e in parts)
            for (b, e), x in zip(parts, w):      # proportional to within a tile
                assert abs((e - b) - tiles * x / sum(w)) <= 1.0 + 1e-9
    assert weighted_tile_partition(1000, [1.0] * 8) == tile_partition(1000, 8)
    with pytest.raises(ValueError):
        weighted_tile_partition(10, [1.0, 0.0])
