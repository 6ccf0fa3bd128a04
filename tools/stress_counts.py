#!/usr/bin/env python3
"""Dense counts (full form) of one block, repeated, against the oracle, with the
split launch arranged so that every workgroup gets exactly one whole tile."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
from cuking_amd.synth import cohort_to_device, plan_cohort
from oracle import pyoracle

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = cuking_amd.KingContext(0)
ctx.set_kernel("tiled"); ctx.set_option("variant", 5)
n, m = 1015, 33744
cohort = plan_cohort(n, 4242)
kind, pa, pb = cohort_to_device(cohort, 0)
wps = cuking_amd.words_per_sample(m)
d_bits = torch.zeros((n, wps), dtype=torch.int64, device="cuda:0")
ctx.synth_bitset(4242, kind, pa, pb, 0, n, m, out=d_bits)
torch.cuda.synchronize()
bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
sm = cuking_amd.Submatrix(n)
oi, oj, oc, _ = pyoracle.all_pairs(pyoracle.submatrix(n), bits)
tiles = ctx.num_tiles(sm)
for wgs in (tiles, 16, 0, 256):
    ctx.set_option("split_wgs", wgs)
    bad = 0
    for rep in range(reps):
        got = ctx.compute_counts(sm, wps, d_bits)[oi, oj]
        wrong = np.zeros(len(oi), dtype=bool)
        for name in oc.dtype.names:
            wrong |= got[name] != oc[name]
        if wrong.any():
            bad += 1
            if bad <= 3:
                idx = np.nonzero(wrong)[0]
                ti, tj = oi[idx] // 128, oj[idx] // 128
                print(f"  wgs {wgs}: {len(idx)} wrong pairs; tiles",
                      sorted(set(zip(ti.tolist(), tj.tolist())))[:6],
                      "first", (int(oi[idx[0]]), int(oj[idx[0]])), got[idx[0]], oc[idx[0]],
                      "rows in tile", sorted(set((oi[idx] % 128).tolist()))[:20],
                      "cols in tile", sorted(set((oj[idx] % 128).tolist()))[:20], flush=True)
    print(f"split_wgs {wgs} ({tiles} tiles): {bad} of {reps} wrong", flush=True)
