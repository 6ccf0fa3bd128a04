#!/usr/bin/env python3
"""Per-rank compute time of the staged multi-GPU schedule, replayed on ONE GPU
(all data present, no waiting): how much do the chunked, strided rectangle
launches cost against the rank's ideal share of a single whole-block launch?"""
import sys
import time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import cuking_amd
from cuking_amd.dist import GpuStagedOps, rank_tile_share, staged_schedule
from cuking_amd.synth import DEFAULT_SEED, cohort_to_device, plan_cohort

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(round(10000 * world ** 0.5))
m, thr = 100000, 0.05
ctx = cuking_amd.KingContext(0)
cohort = plan_cohort(n, DEFAULT_SEED)
kind, pa, pb = cohort_to_device(cohort)
bits = ctx.synth_bitset(DEFAULT_SEED, kind, pa, pb, 0, n, m)
wps = bits.shape[1]
sm = cuking_amd.Submatrix(n)
tile = ctx.tile_samples()
torch.cuda.synchronize()
t0 = time.perf_counter()
whole = ctx.run(sm, wps, bits, thr, 1 << 20)
t0 = time.perf_counter()
whole = ctx.run(sm, wps, bits, thr, 1 << 20)
t_whole = time.perf_counter() - t0
print(f"world {world}: {n} samples, whole block on one GPU {t_whole * 1e3:.1f} ms "
      f"({sm.NumPairs() / t_whole:.3e} pairs/s)", flush=True)
T = (n + tile - 1) // tile
bands = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [17]
for band in bands:
    ctx.set_option("band_rows", band)
    for chunks, streams in ((8, 2), (6, 2), (8, 3)):
        effs = []
        for rank in range(world):
            ops = GpuStagedOps(ctx, sm, wps, bits, thr, 1 << 20, num_streams=streams)
            def one():
                ops.begin()
                for (c0, c1), rect in staged_schedule(n, tile, world, rank, chunks):
                    if rect is None:
                        continue
                    ops.prepare(c0, c1)
                    ops.compute_rect(*rect)
                return ops.finish()
            one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                one()
            dt = (time.perf_counter() - t0) / 3
            effs.append(t_whole * rank_tile_share(T, world, rank) / dt)
        print(f"band {band:2d} chunks {chunks} streams {streams}: min {min(effs):4.0%} mean {sum(effs) / len(effs):4.0%}  "
              + " ".join(f"{e:3.0%}" for e in effs), flush=True)
