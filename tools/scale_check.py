#!/usr/bin/env python3
"""Scale checks on one MI355X (run through gpurun; not part of pytest because of
their size).  Parity at sizes the oracle cannot cover in full is established
through sub-blocks re-computed by the oracle and size-independent properties.

  c2    BASELINE configs[2]: 100k samples x 100k sites, whole triangle
        (4,999,950,000 pairs), threshold 0.0884 (the default, cuking.cu:43)
  c3    BASELINE configs[3] geometry: 300k samples x 150k sites (11.25 GB bitset),
        whole triangle (4.5e10 pairs) on ONE GPU, threshold 0.0884; same checks as c2
  c4    BASELINE configs[4] geometry: 734k samples x 200k sites on ONE GPU
        (36.7 GB bitset + as much again for the kernel layout): the last tile
        range, which exercises > 2^32-element indexing, sub-blocks vs oracle;
        with --whole also the entire 2.7e11-pair triangle in one call
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def subblock_check(ctx, cuking_amd, pyoracle, bits, res, thr, blocks):
    """res: sorted GPU records of a superset region; blocks: ((a0,a1),(b0,b1))."""
    checked = 0
    for (a0, a1), (b0, b1) in blocks:
        rows = bits[a0:a1].cpu().numpy().view(np.uint64)
        if (a0, a1) == (b0, b1):
            osm = pyoracle.Submatrix(a0, a1, a0, a1)
            host = np.ascontiguousarray(rows)
        else:
            osm = pyoracle.Submatrix(a0, a1, b0, b1)
            host = np.ascontiguousarray(np.concatenate(
                [rows, bits[b0:b1].cpu().numpy().view(np.uint64)]))
        exp, ovf, _ = pyoracle.compute(osm, host, thr, threads=16)
        sel = res[(res["sample_i"] >= a0) & (res["sample_i"] < a1) &
                  (res["sample_j"] >= b0) & (res["sample_j"] < b1)]
        assert sel.tobytes() == exp.tobytes(), ((a0, a1), (b0, b1), len(sel), len(exp))
        checked += (a1 - a0) * (b1 - b0) if (a0, a1) != (b0, b1) else (a1 - a0) * (a1 - a0 - 1) // 2
    return checked


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["c2", "c3", "c4"])
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "scale_check.jsonl"))
    ap.add_argument("--whole", action="store_true", help="c4: also run every pair")
    args = ap.parse_args()
    import torch
    import cuking_amd
    from cuking_amd.synth import DEFAULT_SEED, cohort_to_device, plan_cohort
    from oracle import pyoracle

    ctx = cuking_amd.KingContext(0)
    ctx.timing_enable(True)
    report = {"which": args.which}
    if args.which == "c2":
        n, m, thr = 100_000, 100_000, 0.0884
    elif args.which == "c3":
        n, m, thr = 300_000, 150_000, 0.0884
    else:
        n, m, thr = 734_000, 200_000, 0.05
    wps = cuking_amd.words_per_sample(m)
    cohort = plan_cohort(n, DEFAULT_SEED)
    kind, pa, pb = cohort_to_device(cohort)
    t0 = time.perf_counter()
    bits = ctx.synth_bitset(DEFAULT_SEED, kind, pa, pb, 0, n, m)
    torch.cuda.synchronize()
    report["synth_s"] = time.perf_counter() - t0
    report["bitset_GB"] = bits.numel() * 8 / 1e9
    print(f"[{args.which}] synthesised {n} x {m} ({report['bitset_GB']:.2f} GB) in "
          f"{report['synth_s']:.2f}s", flush=True)
    sm = cuking_amd.Submatrix(n)
    nf = cohort.num_founders

    if args.which in ("c2", "c3"):
        t0 = time.perf_counter()
        res = ctx.run(sm, wps, bits, thr, max_results=4 << 20)
        dt = time.perf_counter() - t0
        tm = ctx.timing_collect()
        pairs = sm.NumPairs()
        report["variant"] = ctx.variant_name()
        report.update(pairs=pairs, wall_s=dt, king_ms=tm.king_ms, prepare_ms=tm.prepare_ms,
                      pairs_per_s=pairs / (tm.king_ms * 1e-3), results=int(len(res)))
        print(f"[{args.which}] {pairs} pairs: kernel {tm.king_ms:.1f} ms, prepare {tm.prepare_ms:.1f} ms, "
              f"{report['pairs_per_s']:.3e} pairs/s, {len(res)} records", flush=True)
        got = {(int(r["sample_i"]), int(r["sample_j"])) for r in res}
        want = {(min(a, b), max(a, b)) for a, b, rel in cohort.planted if rel != "half"}
        assert want <= got, f"{len(want - got)} planted relatives (kin>=0.25) missing"
        assert np.all(res["kin"] > np.float32(thr)) and np.all(res["sample_i"] < res["sample_j"])
        key = res["sample_i"].astype(np.int64) * n + res["sample_j"]
        assert np.all(np.diff(key) > 0)
        blocks = [((0, 128), (0, 128)), ((n - 256, n), (n - 256, n)),
                  ((50_000, 50_128), (n - 192, n)), ((nf - 64, nf + 64), (nf - 64, nf + 64))]
        report["pairs_rechecked_by_oracle"] = subblock_check(
            ctx, cuking_amd, pyoracle, bits, res, thr, blocks)
        if args.which == "c2":
            # idempotence + the other tile shape
            ctx.set_option("variant", 1)
            again = ctx.run(sm, wps, bits, thr, max_results=4 << 20)
            assert again.tobytes() == res.tobytes()
            report["variant1_identical"] = True
    else:
        tiles = ctx.num_tiles(sm)
        tile = ctx.tile_samples()
        report["tiles"] = tiles
        report["variant"] = ctx.variant_name()
        take = 150_000
        # (1) the LAST tiles of the enumeration: largest sample indices,
        #     offsets beyond 2^32 uint4 elements in the kernel layout
        t0 = time.perf_counter()
        res_hi = ctx.run(sm, wps, bits, thr, max_results=4 << 20,
                         tile_range=(tiles - take, tiles))
        dt = time.perf_counter() - t0
        tm = ctx.timing_collect()
        print(f"[c4] last {take} of {tiles} tiles: kernel {tm.king_ms:.1f} ms, prepare "
              f"{tm.prepare_ms:.1f} ms, wall {dt:.1f}s, {len(res_hi)} records", flush=True)
        report.update(last_tiles=take, king_ms=tm.king_ms, prepare_ms=tm.prepare_ms,
                      pairs_per_s=take * tile * tile / (tm.king_ms * 1e-3),
                      results_hi=int(len(res_hi)))
        lib = ctx.lib
        import ctypes as C
        rb, re_, cb, ce = (C.c_uint32() for _ in range(4))
        seen = 0
        for t in (tiles - 1, tiles - 2, tiles - take, tiles - take // 2):
            assert lib.cuking_tile_bounds(ctx.handle, C.byref(sm.c), t, C.byref(rb),
                                          C.byref(re_), C.byref(cb), C.byref(ce)) == 0
            seen += subblock_check(ctx, cuking_amd, pyoracle, bits, res_hi, thr,
                                   [((rb.value, re_.value), (cb.value, ce.value))])
        # derived samples (relatives) sit at the end: their diagonal region is
        # inside the last tiles; every planted pair whose both samples are in the
        # last 64-sample tile row must be present
        report["pairs_rechecked_by_oracle"] = seen
        # (2) the staged rectangle path on the far corner
        import torch as _t
        lo = (n // tile - 20) * tile
        results = _t.zeros((1 << 20, 6), dtype=_t.int32, device="cuda:0")
        idx = _t.zeros(2, dtype=_t.int32, device="cuda:0")
        ctx.prepare_samples(sm, wps, bits, lo, n)
        ctx.compute_king_rect(sm, wps, bits, (lo, n), (lo, n), thr, 1 << 20, results,
                              idx[0:1], idx[1:2])
        _t.cuda.synchronize()
        cnt, ovf = idx.tolist()
        assert ovf == 0
        recs = results[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
            cuking_amd.KING_RESULT_DTYPE).copy()
        recs = cuking_amd.sort_results(recs)
        osm = pyoracle.Submatrix(lo, n, lo, n)
        exp, _, _ = pyoracle.compute(osm, np.ascontiguousarray(
            bits[lo:n].cpu().numpy().view(np.uint64)), thr, threads=16)
        assert recs.tobytes() == exp.tobytes()
        report["corner_rect_pairs"] = (n - lo) * (n - lo - 1) // 2
        report["corner_rect_records"] = int(len(exp))
        if args.whole:
            # (3) every pair of the cohort in one call
            ctx.timing_reset()
            t0 = time.perf_counter()
            res = ctx.run(sm, wps, bits, thr, max_results=8 << 20)
            dt = time.perf_counter() - t0
            tm = ctx.timing_collect()
            pairs = sm.NumPairs()
            print(f"[c4] whole triangle, {pairs} pairs: kernel {tm.king_ms / 1e3:.1f} s, "
                  f"prepare {tm.prepare_ms:.1f} ms, {pairs / (tm.king_ms * 1e-3):.3e} pairs/s, "
                  f"{len(res)} records", flush=True)
            got = {(int(r["sample_i"]), int(r["sample_j"])) for r in res}
            want = {(min(a, b), max(a, b)) for a, b, rel in cohort.planted}
            assert want <= got, f"{len(want - got)} planted relatives missing"
            hi = res[(res["sample_i"] >= lo) & (res["sample_j"] >= lo)]
            assert hi.tobytes() == exp.tobytes()      # far corner again, from the whole run
            report.update(whole_pairs=pairs, whole_king_s=tm.king_ms / 1e3, whole_wall_s=dt,
                          whole_pairs_per_s=pairs / (tm.king_ms * 1e-3),
                          whole_records=int(len(res)))
    report["ok"] = True
    print(json.dumps(report), flush=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(report) + "\n")


if __name__ == "__main__":
    main()
