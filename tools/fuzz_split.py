#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep of the matrix-core variant's remainder
splitting (king_mfma.hip): blocks large enough that pieces of k-steps, scratch
slabs and tickets are really exercised -- whole blocks, tile sub-ranges and the
staged rectangles on several streams at once (one slab per stream).
usage: fuzz_split.py [seed] [cases] [first_case]   (first_case: replay one failure)"""
import sys
import time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
from cuking_amd.dist import GpuStagedOps, staged_schedule, tile_partition
from cuking_amd.synth import cohort_to_device, plan_cohort
from oracle import pyoracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
first_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
ctx = cuking_amd.KingContext(0)
ctx.set_kernel("tiled"); ctx.set_option("variant", 5)
t0 = time.time()
for case in range(cases):
    n = int(rng.integers(130, 3000)); m = int(rng.integers(3000, 40000))
    thr = float(rng.choice([0.03, 0.0884, 0.3]))
    mode = int(rng.choice([-1, -1, 0, 1]))
    wgs = int(rng.choice([3, 16, 64, 256, 256]))
    w = int(rng.integers(2, 6))
    world = int(rng.integers(1, 5)); chunks = int(rng.integers(1, 6))
    streams = [int(rng.integers(1, 4)) for _ in range(world)]
    if case < first_case:
        continue
    ctx.set_option("split_wgs", wgs); ctx.set_option("counts_mode", mode)
    cohort = plan_cohort(n, seed * 1000 + case)
    kind, pa, pb = cohort_to_device(cohort, 0)
    wps = cuking_amd.words_per_sample(m)
    d_bits = torch.zeros((n, wps), dtype=torch.int64, device="cuda:0")
    ctx.synth_bitset(seed * 1000 + case, kind, pa, pb, 0, n, m, out=d_bits)
    torch.cuda.synchronize()
    bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
    exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), bits, thr, threads=16)
    sm = cuking_amd.Submatrix(n)
    tag = (case, n, m, thr, mode, wgs)
    for rep in range(2):
        got = ctx.run(sm, wps, d_bits, thr)
        assert got.tobytes() == exp.tobytes(), ("run", rep, tag)
    tiles = ctx.num_tiles(sm)
    parts = [ctx.run(sm, wps, d_bits, thr, tile_range=r) for r in tile_partition(tiles, w)]
    merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
    assert merged.tobytes() == exp.tobytes(), ("tiles", w, tag)
    parts = []
    for rank in range(world):
        ops = GpuStagedOps(ctx, sm, wps, d_bits, thr, max(len(exp), 1) + 8,
                           num_streams=streams[rank])
        ops.begin()
        for (c0, c1), rect in staged_schedule(n, ctx.tile_samples(), world, rank, chunks):
            if rect is None: continue
            ops.prepare(c0, c1); ops.compute_rect(*rect)
        res, cnt, ovf = ops.finish()
        assert ovf == 0, ("staged overflow", tag)
        parts.append(res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
            cuking_amd.KING_RESULT_DTYPE).copy())
    merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
    if merged.tobytes() != exp.tobytes():
        have = {(int(r["sample_i"]), int(r["sample_j"])) for r in merged}
        want = {(int(r["sample_i"]), int(r["sample_j"])) for r in exp}
        print("staged mismatch", world, chunks, streams, tag, "records", len(merged), len(exp),
              "missing", sorted(want - have)[:8], "extra", sorted(have - want)[:8], flush=True)
        raise SystemExit(1)
    if case % 5 == 0:
        print(f"case {case} ok {tag} records {len(exp)} ({time.time() - t0:.0f}s)", flush=True)
print(f"fuzz_split seed {seed}: {cases} cases OK in {time.time() - t0:.0f}s", flush=True)
