"""GPU box: the filter variant (7) against the four-product kernel (6) as the MISSING rate
of the cohort rises (the slack of the filter's bound is the missingness).

The synthetic cohort (1 % missing) gets extra missing calls: a random mask of density
2^-k per sample and site OR-ed into both planes (missing = 11, cuking.cu:507-523).  For
every rate: kernel time per pass of both variants (HIP events around the whole call:
filter + refine + dense-quadrant kernels), candidates, dense quadrants -- and the records
of the two variants compared byte for byte.

usage: [MISSING_KS=0,3] [MISSING_HETERO=0.02,2] [MISSING_RELATED=0.25] python tools/missing_curve.py [samples] [sites] [threshold ...]
       -> gpurun_out/missing_curve.txt
"""
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import cuking_amd
from cuking_amd.synth import cohort_to_device, plan_cohort

SEED = 20240229
MAX_RESULTS = 1 << 24


def heterogeneous_missing(bits, fraction, k, gen):
    """`fraction` of the samples (every round(1 / fraction)-th one) get extra missing calls
    of density 2^-k: the few low-call-rate samples of an ordinary cohort."""
    n = bits.shape[0]
    step = max(1, int(round(1.0 / fraction)))
    who = torch.arange(step // 2, n, step, device=bits.device)
    out = bits.clone()
    out[who] = extra_missing(bits[who], k, gen)
    return out, len(who)


def extra_missing(bits, k, gen):
    """bits [n, wps] int64 (het plane | hom plane): OR a mask of bit density 2^-k into
    both planes.  k = 0: nothing."""
    if k == 0:
        return bits
    n, wps = bits.shape
    half = wps // 2
    mask = None
    for _ in range(k):
        r = torch.randint(-(1 << 63), (1 << 63) - 1, (n, half), dtype=torch.int64,
                          device=bits.device, generator=gen)
        mask = r if mask is None else mask & r
    out = bits.clone()
    out[:, :half] |= mask
    out[:, half:2 * half] |= mask
    return out


def run(ctx, variant, sm, wps, bits, thr, steps=5, warmup=1):
    ctx.set_option("variant", variant)
    ctx.set_option("counts_mode", 0)      # lean form forced, as in tools/filter_curve.sh
    results = torch.zeros((MAX_RESULTS, 6), dtype=torch.int32, device=bits.device)
    index_flag = torch.zeros(2, dtype=torch.int32, device=bits.device)
    ctx.invalidate()

    def step():
        index_flag.zero_()
        ctx.compute_king(sm, wps, bits, thr, MAX_RESULTS, results, index_flag[0:1],
                         index_flag[1:2])

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ctx.timing_reset()
    f0 = (ctx.get_option("filter_candidates"), ctx.get_option("filter_dense_quadrants")) \
        if variant == 7 else None
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t = ctx.timing_collect()
    f1 = (ctx.get_option("filter_candidates"), ctx.get_option("filter_dense_quadrants")) \
        if variant == 7 else None
    count, ovf = index_flag.tolist()
    assert not ovf, "result overflow"
    recs = results[:count].cpu().numpy().view(np.uint32).reshape(-1).view(
        cuking_amd.KING_RESULT_DTYPE).copy()
    recs = cuking_amd.sort_results(recs)
    filt = None if f0 is None else ((f1[0] - f0[0]) / steps, (f1[1] - f0[1]) / steps)
    return t.king_ms / max(t.king_launches, 1), recs, filt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    thrs = [float(x) for x in sys.argv[3:]] or [0.0884, 0.0442]
    ctx = cuking_amd.KingContext(0)
    ctx.timing_enable(True)
    ctx.set_option("reuse_prepared", 1)
    wps = cuking_amd.words_per_sample(m)
    sm = cuking_amd.Submatrix(n)
    cohort = plan_cohort(n, SEED)
    kind, pa, pb = cohort_to_device(cohort, 0)
    base = ctx.synth_bitset(SEED, kind, pa, pb, 0, n, m)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(7)
    out = open("gpurun_out/missing_curve.txt", "w")

    def say(s):
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    say(f"# {n} samples x {m} sites, synthetic cohort (1 % missing) + extra missing calls of density 2^-k;")
    say("# lean form forced; kernel_ms = HIP events around the whole call; records of variants 7 and 6 compared")
    ks = [int(x) for x in os.environ.get("MISSING_KS", "0,6,5,4,3,2").split(",")]
    for k in ks:
        bits = extra_missing(base, k, gen)
        half = wps // 2
        # measured missing rate over the stored sites (padding sites of the last word included)
        het, hom = bits[:256, :half], bits[:256, half:2 * half]
        both = (het & hom).cpu().numpy().view(np.uint64)
        rate = float(np.unpackbits(both.view(np.uint8)).mean())
        for thr in thrs:
            ms7, r7, filt = run(ctx, 7, sm, wps, bits, thr)
            ms6, r6, _ = run(ctx, 6, sm, wps, bits, thr)
            same = r7.tobytes() == r6.tobytes()
            say(f"missing {rate:.4f} thr {thr} variant 7 kernel_ms {ms7:.3f} variant 6 kernel_ms {ms6:.3f} "
                f"records {len(r7)} equal {same} candidates {filt[0]:.1f} dense_quadrants {filt[1]:.1f}")
            if not same:
                raise SystemExit("variants disagree")
        del bits
    # The heterogeneous cohort: 2 % of the samples at ~20 % missing calls (density 2^-2 on
    # top of the 1 %), the others as they are.
    hetero = os.environ.get("MISSING_HETERO", "0.02,2")
    if hetero:
        frac, k = hetero.split(",")
        bits, count = heterogeneous_missing(base, float(frac), int(k), gen)
        for thr in thrs:
            for sort in (1, 0):
                ctx.set_option("filter_sort", sort)
                ms7, r7, filt = run(ctx, 7, sm, wps, bits, thr)
                ms6, r6, _ = run(ctx, 6, sm, wps, bits, thr)
                same = r7.tobytes() == r6.tobytes()
                say(f"heterogeneous: {count} of {n} samples with extra missing calls of density 2^-{k}, "
                    f"sorted layout {sort}, thr {thr} variant 7 kernel_ms {ms7:.3f} variant 6 kernel_ms "
                    f"{ms6:.3f} records {len(r7)} equal {same} candidates {filt[0]:.1f} "
                    f"dense_quadrants {filt[1]:.1f}")
                if not same:
                    raise SystemExit("variants disagree")
        ctx.set_option("filter_sort", 1)
        del bits
    # The related cohort: a share of the samples are duplicates of samples elsewhere in the
    # cohort, so that a good part of the TILES holds a record or two -- with and without the
    # hand-over of a tile's few live pairs at the check point (king_filter.hip).
    related = os.environ.get("MISSING_RELATED", "0.25")
    if related:
        pairs = int(n * float(related))
        order = torch.randperm(n, device=base.device, generator=gen)
        bits = base.clone()
        bits[order[pairs:2 * pairs]] = base[order[:pairs]]
        for thr in thrs:
            ms6, r6, _ = run(ctx, 6, sm, wps, bits, thr)
            for emit in (64, 0):
                ctx.set_option("filter_check_emit", emit)
                e0 = ctx.get_option("filter_early_exits")
                ms7, r7, filt = run(ctx, 7, sm, wps, bits, thr)
                exits = (ctx.get_option("filter_early_exits") - e0) / 6   # 1 warm-up + 5 passes
                same = r7.tobytes() == r6.tobytes()
                say(f"related: {pairs} duplicated samples at random places, hand-over cap {emit}, thr {thr} "
                    f"variant 7 kernel_ms {ms7:.3f} variant 6 kernel_ms {ms6:.3f} records {len(r7)} "
                    f"equal {same} candidates {filt[0]:.1f} tiles_left_at_the_check {exits:.0f} "
                    f"of {ctx.num_tiles(sm)}")
                if not same:
                    raise SystemExit("variants disagree")
        ctx.set_option("filter_check_emit", 64)
        del bits
    out.close()


if __name__ == "__main__":
    main()
