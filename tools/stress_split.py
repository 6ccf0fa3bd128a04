#!/usr/bin/env python3
"""Repeats one (block, form, split, streams) configuration of the staged
schedule many times and counts mismatches against the oracle: to localise rare,
timing-dependent failures.  usage: stress_split.py [reps]"""
import sys
import itertools
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
from cuking_amd.dist import GpuStagedOps, staged_schedule
from cuking_amd.synth import cohort_to_device, plan_cohort
from oracle import pyoracle

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
THR = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
ctx = cuking_amd.KingContext(0)
ctx.set_kernel("tiled"); ctx.set_option("variant", 5)
n, m, thr, chunks = 1015, 33744, THR, 3
cohort = plan_cohort(n, 4242)
kind, pa, pb = cohort_to_device(cohort, 0)
wps = cuking_amd.words_per_sample(m)
d_bits = torch.zeros((n, wps), dtype=torch.int64, device="cuda:0")
ctx.synth_bitset(4242, kind, pa, pb, 0, n, m, out=d_bits)
torch.cuda.synchronize()
bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), bits, thr, threads=16)
sm = cuking_amd.Submatrix(n)
for mode, wgs, streams in itertools.product((1, 0), (16, 0, 256), (1, 3)):
    ctx.set_option("counts_mode", mode); ctx.set_option("split_wgs", wgs)
    bad = 0
    for rep in range(reps):
        ops = GpuStagedOps(ctx, sm, wps, d_bits, thr, len(exp) + 8, num_streams=streams)
        ops.begin()
        for (c0, c1), rect in staged_schedule(n, ctx.tile_samples(), 1, 0, chunks):
            if rect is None:
                continue
            ops.prepare(c0, c1); ops.compute_rect(*rect)
        res, cnt, ovf = ops.finish()
        got = cuking_amd.sort_results(res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
            cuking_amd.KING_RESULT_DTYPE).copy())
        if got.tobytes() != exp.tobytes():
            bad += 1
            if bad <= 2:
                have = {(int(r["sample_i"]), int(r["sample_j"])) for r in got}
                want = {(int(r["sample_i"]), int(r["sample_j"])) for r in exp}
                miss = sorted(want - have)
                print("  mismatch: records", len(got), len(exp), "missing", len(miss), miss[:4],
                      "extra", sorted(have - want)[:4], "tiles of missing",
                      sorted({(i // 128, j // 128) for i, j in miss})[:8],
                      "rows%128", sorted({i % 128 for i, j in miss})[:40],
                      "cols%128", sorted({j % 128 for i, j in miss})[:40], flush=True)
    print(f"form {'full' if mode else 'lean'} split_wgs {wgs} streams {streams}: "
          f"{bad} of {reps} wrong", flush=True)
