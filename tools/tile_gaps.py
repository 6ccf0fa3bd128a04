"""GPU box, timing build (-DCUKING_FILTER_TIMING=1, tools/tile_gaps.sh): what a CU does between
the k loops of two tiles of the filter kernel -- the wait for its next workgroup (exit at the
check point -> entry of the next workgroup on the same CU) and the new workgroup's way to its
first request -- beside the k-step time the tiles measure themselves.

usage: python tools/tile_gaps.py [samples] [sites] [threshold]   -> gpurun_out/tile_gaps.txt
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import cuking_amd
from cuking_amd.synth import cohort_to_device, plan_cohort

SEED = 20240229
MAX_RESULTS = 1 << 24


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0884
    ctx = cuking_amd.KingContext(0)
    ctx.timing_enable(True)
    ctx.set_option("reuse_prepared", 1)
    ctx.set_option("variant", 7)
    ctx.set_option("counts_mode", 0)
    wps = cuking_amd.words_per_sample(m)
    sm = cuking_amd.Submatrix(n)
    kind, pa, pb = cohort_to_device(plan_cohort(n, SEED), 0)
    bits = ctx.synth_bitset(SEED, kind, pa, pb, 0, n, m)
    results = torch.zeros((MAX_RESULTS, 6), dtype=torch.int32, device=bits.device)
    index_flag = torch.zeros(2, dtype=torch.int32, device=bits.device)
    out = open("gpurun_out/tile_gaps.txt", "a")
    for rotate in (1, 0, 1, 0):
        ctx.set_option("filter_rotate", rotate)

        def step():
            index_flag.zero_()
            ctx.compute_king(sm, wps, bits, thr, MAX_RESULTS, results, index_flag[0:1], index_flag[1:2])

        step()
        torch.cuda.synchronize()
        before = [ctx.get_option(f"filter_total_{k}") for k in (8, 9, 10, 11)]
        ctx.timing_reset()
        steps = 3
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        t = ctx.timing_collect()
        after = [ctx.get_option(f"filter_total_{k}") for k in (8, 9, 10, 11)]
        d = [a - b for a, b in zip(after, before)]
        tiles = ctx.num_tiles(sm)
        ms = t.king_ms / max(t.king_launches, 1)
        line = (f"{n} x {m} thr {thr} rotate {rotate}: kernel_ms {ms:.3f} tiles {tiles} "
                f"us_per_tile_and_CU {ms * 1e3 * 256 / tiles:.1f} "
                f"gap_exit_to_next_entry_us {d[0] / max(d[1], 1) / 100:.2f} (n {d[1] // steps}) "
                f"entry_to_first_request_us {d[2] / max(d[3], 1) / 100:.2f} (n {d[3] // steps}) "
                f"k_step_us {ctx.get_option('filter_step_ticks16') / 1600:.4f}")
        print(line, flush=True)
        out.write(line + "\n")
    out.close()


if __name__ == "__main__":
    main()
