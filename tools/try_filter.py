"""Quick parity / timing probe of the filter variant (7) on an MI355X: small cohorts
against the CPU oracle in every mode the variant has, then a few timed passes."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuking_amd  # noqa: E402
from cuking_amd.synth import cohort_to_device, plan_cohort  # noqa: E402
from oracle import pyoracle  # noqa: E402


def cohort_bits(ctx, n, m, seed):
    cohort = plan_cohort(n, seed)
    kind, pa, pb = cohort_to_device(cohort)
    bits = ctx.synth_bitset(seed, kind, pa, pb, 0, n, m)
    torch.cuda.synchronize()
    return bits


def check(ctx, n, m, seed, thr, split=1, shard=0, label="", **opts):
    bits = cohort_bits(ctx, n, m, seed)
    host = bits.cpu().numpy().view(np.uint64)
    osm = pyoracle.submatrix(n, split, shard)
    sm = cuking_amd.Submatrix(n, split, shard)
    exp, _, _ = pyoracle.compute(osm, host if split == 1 else None, thr) if split == 1 else (None, 0, 0)
    ctx.set_option("variant", 7)
    for k, v in opts.items():
        ctx.set_option(k, v)
    got = ctx.run(sm, bits.shape[1], bits, thr)
    ctx.set_option("variant", 6)
    ref = ctx.run(sm, bits.shape[1], bits, thr)
    ok6 = got.tobytes() == ref.tobytes()
    oko = exp is None or got.tobytes() == exp.tobytes()
    print(f"{label or 'case'}: n={n} m={m} thr={thr} opts={opts}: {len(got)} records, "
          f"vs variant 6 {'OK' if ok6 else 'MISMATCH'}, vs oracle {'OK' if oko else 'MISMATCH'}",
          flush=True)
    for k in opts:
        ctx.set_option(k, {"filter_quadrant_cap": 384, "filter_cand_cap": 1 << 25,
                           "counts_mode": -1, "max_launch_blocks": 0}[k])
    return ok6 and oko


def main():
    ctx = cuking_amd.KingContext(0)
    ok = True
    ok &= check(ctx, 333, 5000, 7, 0.05, label="small")
    ok &= check(ctx, 700, 20001, 3, 0.1, label="odd sites")
    ok &= check(ctx, 1500, 30000, 5, 0.04, label="6x6 tiles")
    ok &= check(ctx, 1500, 30000, 5, 0.04, label="dense path", filter_quadrant_cap=0)
    ok &= check(ctx, 1500, 30000, 5, 0.04, label="short list", filter_cand_cap=5)
    ok &= check(ctx, 1500, 30000, 5, 0.04, label="chunks", max_launch_blocks=3)
    ok &= check(ctx, 1500, 30000, 5, 0.04, label="full form", counts_mode=1)
    ok &= check(ctx, 1500, 30000, 5, 0.0, label="thr 0")
    ok &= check(ctx, 1500, 30000, 5, 0.012, label="low thr")
    ok &= check(ctx, 1500, 30000, 5, 0.6, label="thr 0.6")
    print("ALL OK" if ok else "FAILED", flush=True)
    # timing
    for n, m in ((10000, 100000), (40000, 100000)):
        bits = cohort_bits(ctx, n, m, 11)
        sm = cuking_amd.Submatrix(n)
        ctx.set_option("reuse_prepared", 1)
        for variant in (7, 6):
            ctx.set_option("variant", variant)
            got = ctx.run(sm, bits.shape[1], bits, 0.05)
            torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                got = ctx.run(sm, bits.shape[1], bits, 0.05)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print(f"n={n} m={m} variant {variant}: {min(ts) * 1e3:.2f} ms (host clock, whole call), "
                  f"{len(got)} records", flush=True)
        del bits
    ctx.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
