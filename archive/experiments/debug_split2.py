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
n, m, thr, seed = 514, 17182, float(sys.argv[1]) if len(sys.argv) > 1 else 0.0884, 1003
cohort = plan_cohort(n, seed)
kind, pa, pb = cohort_to_device(cohort, 0)
wps = cuking_amd.words_per_sample(m)
d_bits = torch.zeros((n, wps), dtype=torch.int64, device="cuda:0")
ctx.synth_bitset(seed, kind, pa, pb, 0, n, m, out=d_bits)
torch.cuda.synchronize()
bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), bits, thr, threads=16)
sm = cuking_amd.Submatrix(n)
tile = ctx.tile_samples()
for wgs, world in ((16, 1), (16, 2), (3, 3), (64, 1)):
    ctx.set_option("split_wgs", wgs); ctx.set_option("counts_mode", 1)
    parts = []
    for rank in range(world):
        ops = GpuStagedOps(ctx, sm, wps, d_bits, thr, len(exp) + 8, num_streams=1)
        ops.begin()
        for (c0, c1), rect in staged_schedule(n, tile, world, rank, 1):
            if rect is None: continue
            ops.prepare(c0, c1); ops.compute_rect(*rect)
        res, cnt, ovf = ops.finish()
        parts.append(res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
            cuking_amd.KING_RESULT_DTYPE).copy())
    got = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
    print(f"wgs {wgs} world {world}: records {len(got)}/{len(exp)}")
    if len(got) != len(exp):
        have = {(int(r["sample_i"]), int(r["sample_j"])): r for r in got}
        for r in exp:
            key = (int(r["sample_i"]), int(r["sample_j"]))
            if key not in have:
                print("   missing", key, "tile", (key[0] // tile, key[1] // tile), "kin", r["kin"], "ibs", r["ibs0"], r["ibs1"], r["ibs2"])
    if len(got) == len(exp):
        for f in exp.dtype.names:
            bad = np.nonzero(got[f] != exp[f])[0] if f != "kin" else np.nonzero(got[f].view(np.uint32) != exp[f].view(np.uint32))[0]
            if len(bad):
                ti = np.unique(exp["sample_i"][bad] // tile); tj = np.unique(exp["sample_j"][bad] // tile)
                tiles = sorted(set(zip((exp["sample_i"][bad] // tile).tolist(), (exp["sample_j"][bad] // tile).tolist())))
                print(f"   field {f}: {len(bad)} wrong, tiles {tiles[:12]}")
                k = bad[0]
                print("      e.g.", exp["sample_i"][k], exp["sample_j"][k], "got", got[f][k], "want", exp[f][k])
