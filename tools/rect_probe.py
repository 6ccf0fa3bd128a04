#!/usr/bin/env python3
"""Where does the rectangle mode lose time?  Times (HIP events through the ABI):
whole-block band launch, full-square rectangle (half of its workgroups exit
early below the diagonal), and an off-diagonal rectangle without early exits."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import cuking_amd
from cuking_amd.synth import DEFAULT_SEED, cohort_to_device, plan_cohort

n, m, thr = 10048, 100000, 0.05
ctx = cuking_amd.KingContext(0)
ctx.timing_enable(True)
cohort = plan_cohort(n, DEFAULT_SEED)
kind, pa, pb = cohort_to_device(cohort)
bits = ctx.synth_bitset(DEFAULT_SEED, kind, pa, pb, 0, n, m)
wps = bits.shape[1]
sm = cuking_amd.Submatrix(n)
res = torch.zeros((1 << 20, 6), dtype=torch.int32, device="cuda:0")
idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()

def timed(label, fn, tiles, reps=5):
    fn(); torch.cuda.synchronize(); ctx.timing_reset()
    for _ in range(reps):
        idx.zero_(); fn()
    torch.cuda.synchronize()
    t = ctx.timing_collect()
    ms = t.king_ms / reps
    print(f"{label:46s} {ms:7.2f} ms  {tiles:6d} real tiles  {ms / tiles * 1e3:6.3f} us/tile", flush=True)

T = n // 64
timed("band enumeration, whole triangle", lambda: ctx.compute_king(sm, wps, bits, thr, 1 << 20, res, idx[0:1], idx[1:2]), T * (T + 1) // 2)
ctx.prepare_samples(sm, wps, bits, 0, n)
rect = lambda rows, cols: ctx.compute_king_rect(sm, wps, bits, rows, cols, thr, 1 << 20, res, idx[0:1], idx[1:2])
timed("rect full square (half exit early)", lambda: rect((0, n), (0, n)), T * (T + 1) // 2)
h = (T // 2) * 64
timed("rect off-diagonal half x half (no exits)", lambda: rect((0, h), (h, n)), (T // 2) * (T - T // 2))
timed("rect upper-left diagonal half", lambda: rect((0, h), (0, h)), (T // 2) * (T // 2 + 1) // 2)
timed("rect strided rows (every 8th) x all cols", lambda: rect((0, n, 8 * 64), (0, n)), sum(T - r for r in range(0, T, 8)))
timed("rect rows [0,20 tiles) x all cols", lambda: rect((0, 20 * 64), (0, n)), sum(T - r for r in range(0, 20)))
timed("rect rows [137,157) tiles x all cols", lambda: rect((137 * 64, n), (0, n)), sum(T - r for r in range(137, 157)))
timed("rect strided rows (every 8th) x right half", lambda: rect((0, h, 8 * 64), (h, n)), len(range(0, T // 2, 8)) * (T - T // 2))
timed("rect strided rows (every 2nd) x all cols", lambda: rect((0, n, 2 * 64), (0, n)), sum(T - r for r in range(0, T, 2)))
ctx.set_option("band_rows", 4)
timed("  band_rows=4: strided every 8th x all cols", lambda: rect((0, n, 8 * 64), (0, n)), sum(T - r for r in range(0, T, 8)))
ctx.set_option("band_rows", 64)
timed("  band_rows=64: strided every 8th x all cols", lambda: rect((0, n, 8 * 64), (0, n)), sum(T - r for r in range(0, T, 8)))
for g in (15, 17, 13):
    ctx.set_option("band_rows", g)
    timed(f"  band_rows={g}: strided every 8th x all cols", lambda: rect((0, n, 8 * 64), (0, n)), sum(T - r for r in range(0, T, 8)))
    timed(f"  band_rows={g}: rect full square", lambda: rect((0, n), (0, n)), T * (T + 1) // 2)
    timed(f"  band_rows={g}: band enumeration, whole triangle", lambda: ctx.compute_king(sm, wps, bits, thr, 1 << 20, res, idx[0:1], idx[1:2]), T * (T + 1) // 2)
