import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
from cuking_amd.dist import GpuStagedOps, staged_schedule
from cuking_amd.synth import cohort_to_device, plan_cohort
from oracle import pyoracle
ctx = cuking_amd.KingContext(0)
ctx.set_kernel("tiled"); ctx.set_option("variant", 5)
n, m, thr, seed = 514, 17182, 0.0884, 1003
cohort = plan_cohort(n, seed)
kind, pa, pb = cohort_to_device(cohort, 0)
wps = cuking_amd.words_per_sample(m)
d_bits = torch.zeros((n, wps), dtype=torch.int64, device="cuda:0")
ctx.synth_bitset(seed, kind, pa, pb, 0, n, m, out=d_bits)
torch.cuda.synchronize()
bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), bits, thr, threads=16)
want = {(int(r["sample_i"]), int(r["sample_j"])) for r in exp}
sm = cuking_amd.Submatrix(n)
for mode in (0, 1):
    for wgs in (0, 3, 16, 64, 256):
        for world in (1, 2, 3):
            ctx.set_option("split_wgs", wgs); ctx.set_option("counts_mode", mode)
            parts = []
            for rank in range(world):
                ops = GpuStagedOps(ctx, sm, wps, d_bits, thr, len(exp) + 8, num_streams=1)
                ops.begin()
                for (c0, c1), rect in staged_schedule(n, ctx.tile_samples(), world, rank, 1):
                    if rect is None: continue
                    ops.prepare(c0, c1); ops.compute_rect(*rect)
                res, cnt, ovf = ops.finish()
                parts.append(res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
                    cuking_amd.KING_RESULT_DTYPE).copy())
            merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
            have = {(int(r["sample_i"]), int(r["sample_j"])) for r in merged}
            ok = merged.tobytes() == exp.tobytes()
            print(f"mode {mode} wgs {wgs:3d} world {world}: {'ok' if ok else 'MISMATCH'} "
                  f"{len(merged)}/{len(exp)} missing {sorted(want - have)[:6]} extra {sorted(have - want)[:4]}",
                  flush=True)
print("== whole-block runs")
for mode in (0, 1):
    for wgs in (1, 2, 3, 5, 7, 16):
        ctx.set_option("split_wgs", wgs); ctx.set_option("counts_mode", mode)
        got = ctx.run(sm, wps, d_bits, thr)
        have = {(int(r["sample_i"]), int(r["sample_j"])) for r in got}
        print(f"run mode {mode} wgs {wgs}: {'ok' if got.tobytes() == exp.tobytes() else 'MISMATCH'} {len(got)}/{len(exp)} "
              f"missing {sorted(want - have)[:6]}", flush=True)
        counts = ctx.compute_counts(sm, wps, d_bits) if mode == 1 else None
        if counts is not None:
            oi, oj, oc, _ = pyoracle.all_pairs(pyoracle.submatrix(n), bits)
            sel = counts[oi, oj]
            bad = {f: int((sel[f] != oc[f]).sum()) for f in oc.dtype.names}
            print("    dense counts wrong per field:", bad, flush=True)
