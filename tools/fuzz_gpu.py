#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (not part of pytest): shapes, shards, kernel
variants, lean/full forms, thresholds, tile ranges and staged schedules."""
import sys
import time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
from cuking_amd.dist import GpuStagedOps, staged_schedule, tile_partition
from conftest import random_genotypes
from oracle import pyoracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
ctx = cuking_amd.KingContext(0)
t0 = time.time()
for case in range(cases):
    n = int(rng.integers(2, 700)); m = int(rng.integers(1, 2500))
    if rng.random() < 0.15:   # enough tiles for the XCD-aware order (launches of >= 64 tiles)
        n = int(rng.integers(1400, 2300))
    k = int(rng.integers(1, 4)); shard = int(rng.integers(0, k * (k + 1) // 2))
    thr = float(rng.choice([-1e30, -0.2, 0.0, 0.03, 0.0884, 0.3]))
    variant = int(rng.integers(0, 6)); mode = int(rng.integers(-1, 2))
    kernel = "stream" if rng.random() < 0.15 else "tiled"
    geno = random_genotypes(rng, n, m, missing=float(rng.choice([0.0, 0.02, 0.3])))
    if n > 3:
        geno[n - 1] = geno[0]
        if rng.random() < 0.3: geno[1] = -1
    osm = pyoracle.submatrix(n, k, shard)
    bits = pyoracle.bitset_from_genotypes(geno, osm)
    exp, _, _ = pyoracle.compute(osm, bits, thr, threads=8)
    sm = cuking_amd.Submatrix(n, k, shard)
    ctx.set_kernel(kernel); ctx.set_option("variant", variant); ctx.set_option("counts_mode", mode)
    ctx.set_option("xcd_swizzle", int(rng.integers(0, 3)))
    ctx.set_option("band_rows", int(rng.choice([0, 0, 1, 3, 5, 17])))
    ctx.set_option("split_wgs", int(rng.choice([0, 256, 256])))
    d_bits = (ctx.upload_bitset(bits) if bits.shape[0] else
              torch.zeros(2, dtype=torch.int64, device="cuda:0"))
    wps = cuking_amd.words_per_sample(m)
    tag = (case, n, m, k, shard, thr, kernel, variant, mode)
    got = ctx.run(sm, wps, d_bits, thr)
    assert got.tobytes() == exp.tobytes(), ("run", tag)
    if kernel == "tiled" and bits.shape[0]:
        tiles = ctx.num_tiles(sm)
        if tiles >= 2:
            w = int(rng.integers(2, 5))
            parts = [ctx.run(sm, wps, d_bits, thr, tile_range=r) for r in tile_partition(tiles, w)]
            merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
            assert merged.tobytes() == exp.tobytes(), ("tiles", tag)
        if k == 1 and n >= 2:
            world = int(rng.integers(1, 9)); chunks = int(rng.integers(1, 9))
            parts = []
            for rank in range(world):
                ops = GpuStagedOps(ctx, sm, wps, d_bits, thr, max(len(exp), 1) + 8,
                                   num_streams=int(rng.integers(1, 4)))
                ops.begin()
                for (c0, c1), rect in staged_schedule(n, ctx.tile_samples(), world, rank, chunks):
                    if rect is None: continue
                    ops.prepare(c0, c1); ops.compute_rect(*rect)
                res, cnt, ovf = ops.finish()
                assert ovf == 0, ("staged overflow", tag)
                parts.append(res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
                    cuking_amd.KING_RESULT_DTYPE).copy())
            merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
            assert merged.tobytes() == exp.tobytes(), ("staged", world, chunks, tag)
    if case % 25 == 0:
        print(f"case {case} ok ({time.time() - t0:.0f}s)", flush=True)
print(f"fuzz seed {seed}: {cases} cases OK in {time.time() - t0:.0f}s", flush=True)
