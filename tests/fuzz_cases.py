"""Seeded random GPU-vs-oracle sweeps, shared by `pytest -m gpu`
(tests/test_gpu_fuzz.py) and the command-line fuzzers (tools/fuzz_gpu.py,
tools/fuzz_split.py, tools/stress_split.py) that run the same cases in bulk.

Every failure is reported as a reproducer: the generator function, its seed and
the index of the case, plus the case's parameters -- `python tools/fuzz_gpu.py
SEED CASES FIRST_CASE` replays it.  The checker is the CPU oracle
(oracle/pyoracle.py); everything checked goes through the C ABI."""
from __future__ import annotations

import itertools
import time

import numpy as np


class FuzzMismatch(AssertionError):
    pass


def _records(res, cnt):
    import cuking_amd
    return res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
        cuking_amd.KING_RESULT_DTYPE).copy()


def _staged(ctx, sm, wps, d_bits, thr, cap, n, world, chunks, streams):
    """Every rank's staged schedule replayed on this GPU; the ranks' records."""
    import cuking_amd
    from cuking_amd.dist import GpuStagedOps, staged_schedule
    parts = []
    for rank in range(world):
        ops = GpuStagedOps(ctx, sm, wps, d_bits, thr, cap, num_streams=streams[rank])
        ops.begin()
        for (c0, c1), rect in staged_schedule(n, ctx.tile_samples(), world, rank, chunks):
            if rect is None:
                continue
            ops.prepare(c0, c1)
            ops.compute_rect(*rect)
        res, cnt, ovf = ops.finish()
        if ovf:
            raise FuzzMismatch("staged overflow")
        parts.append(_records(res, cnt))
    return cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))


def _diff(got, exp):
    have = {(int(r["sample_i"]), int(r["sample_j"])) for r in got}
    want = {(int(r["sample_i"]), int(r["sample_j"])) for r in exp}
    return (f"records {len(got)} vs {len(exp)}, missing {sorted(want - have)[:6]}, "
            f"extra {sorted(have - want)[:6]}")


def run_general(ctx, seed: int, cases: int, first_case: int = 0, log=None) -> int:
    """Shapes, shards, kernel variants, lean/full forms, thresholds, tile ranges
    and staged schedules (tools/fuzz_gpu.py).  Returns the number of cases run."""
    import torch
    import cuking_amd
    from cuking_amd.dist import tile_partition
    from conftest import random_genotypes
    from oracle import pyoracle

    rng = np.random.default_rng(seed)
    num_variants = ctx.lib.cuking_num_variants()
    t0, ran = time.time(), 0
    for case in range(cases):
        n = int(rng.integers(2, 700))
        m = int(rng.integers(1, 2500))
        if rng.random() < 0.15:   # enough tiles for the XCD-aware order (launches of >= 64 tiles)
            n = int(rng.integers(1400, 2300))
        big = False
        if n < 1400 and rng.random() < 0.01:
            # the filter variant's give-up decision: more tiles than CUs (> 23 x 23 tiles of
            # 256 samples) and a threshold inside the noise of so few sites, so that most
            # quadrants of the first round go dense and the later tiles hand theirs over
            n, m, big = int(rng.integers(5900, 6600)), int(rng.integers(100, 200)), True
        k = int(rng.integers(1, 4))
        shard = int(rng.integers(0, k * (k + 1) // 2))
        thr = float(rng.choice([-1e30, -0.2, 0.0, 0.03, 0.0884, 0.3]))
        variant = int(rng.integers(0, num_variants))
        mode = int(rng.integers(-1, 2))
        if big:
            # (the whole triangle; every pair of 6,000 samples would not fit the default
            #  --max_results: a threshold about one sigma out; that variant, lean form, in
            #  three of four cases)
            k, shard, thr = 1, 0, 0.0884
            if rng.random() < 0.75 and num_variants > 7:
                variant, mode = 7, 0
        kernel = "stream" if rng.random() < 0.15 else "tiled"
        missing = float(rng.choice([0.0, 0.02, 0.3]))
        geno = random_genotypes(rng, n, m, missing=missing)
        if n > 3:
            geno[n - 1] = geno[0]
            if rng.random() < 0.3:
                geno[1] = -1
            if rng.random() < 0.3:       # a few low-call-rate samples (the sorted layout's case)
                who = rng.choice(n, size=max(1, n // 16), replace=False)
                geno[who] = np.where(rng.random((len(who), m)) < 0.4, -1, geno[who])
        swizzle = int(rng.integers(0, 3))
        band = int(rng.choice([0, 0, 1, 3, 5, 17]))
        wgs = int(rng.choice([0, 256, 256]))
        reuse = int(rng.integers(0, 2))
        w = int(rng.integers(2, 5))
        world = int(rng.integers(1, 9))
        chunks = int(rng.integers(1, 9))
        streams = [int(rng.integers(1, 4)) for _ in range(world)]
        # filter variant: who computes a pair exactly (candidate list / dense quadrants)
        qcap = int(rng.choice([384, 384, 0, 2]))
        ccap = int(rng.choice([1 << 20, 1 << 20, 0, 5]))
        smin = int(rng.choice([8, 1, 1]))      # remainder pieces of k even for short bitsets
        # ... and its check points inside the k loop (forecast: off / short launches / always;
        # rigorous: off / automatic / an entry of the share menu), for bitsets of >= 4 k-steps
        chk0 = int(rng.choice([1, 0, 2, 2]))
        chk1 = int(rng.choice([1, 0, 3, 5, 7, 9]))
        srt, lazy = int(rng.integers(0, 3)), int(rng.integers(0, 2))   # sorted layout, lazy codes
        emit = int(rng.choice([64, 64, 0, 1, 255]))    # live pairs a tile hands over at the check
        # rotated tiles: as shipped (joins the XCD's position) / off / a phase per tile / one phase
        rot = int(rng.choice([1, 0, 2, 2, 3 + int(rng.integers(0, 64))]))
        pers = int(rng.integers(0, 2))       # one resident workgroup per CU takes tile after tile
        if case < first_case:
            continue
        tag = dict(fuzzer="run_general", seed=seed, case=case, n=n, m=m, split_factor=k,
                   shard=shard, thr=thr, kernel=kernel, variant=variant, counts_mode=mode,
                   xcd_swizzle=swizzle, band_rows=band, split_wgs=wgs, reuse_prepared=reuse,
                   filter_quadrant_cap=qcap, filter_cand_cap=ccap, filter_split_min_steps=smin,
                   filter_check0=chk0, filter_check1=chk1, filter_check_min_steps=4,
                   filter_sort=srt, filter_lazy_codes=lazy, filter_check_emit=emit,
                   filter_rotate=rot, filter_rotate_min_steps=4, filter_persistent=pers,
                   filter_persistent_min_tiles=0)
        osm = pyoracle.submatrix(n, k, shard)
        bits = pyoracle.bitset_from_genotypes(geno, osm)
        exp, _, _ = pyoracle.compute(osm, bits, thr, threads=8)
        sm = cuking_amd.Submatrix(n, k, shard)
        ctx.set_kernel(kernel)
        ctx.set_option("variant", variant)
        ctx.set_option("counts_mode", mode)
        ctx.set_option("xcd_swizzle", swizzle)
        ctx.set_option("band_rows", band)
        ctx.set_option("split_wgs", wgs)
        ctx.set_option("filter_quadrant_cap", qcap)
        ctx.set_option("filter_cand_cap", ccap)
        ctx.set_option("filter_split_min_steps", smin)
        ctx.set_option("filter_check0", chk0)
        ctx.set_option("filter_check1", chk1)
        ctx.set_option("filter_check_emit", emit)
        ctx.set_option("filter_rotate", rot)
        ctx.set_option("filter_rotate_min_steps", 4)
        ctx.set_option("filter_rotate_min_tiles", 0)
        ctx.set_option("filter_persistent", pers)
        ctx.set_option("filter_persistent_min_tiles", 0)
        ctx.set_option("filter_check_min_steps", 4)
        ctx.set_option("filter_sort", srt)
        ctx.set_option("filter_lazy_codes", lazy)
        # (a new bitset may land on a recycled pointer: tell the library)
        ctx.set_option("reuse_prepared", reuse)
        ctx.invalidate()
        d_bits = (ctx.upload_bitset(bits) if bits.shape[0] else
                  torch.zeros(2, dtype=torch.int64, device=f"cuda:{ctx.device}"))
        wps = cuking_amd.words_per_sample(m)
        for rep in range(2 if reuse else 1):     # the second call reuses the layout
            got = ctx.run(sm, wps, d_bits, thr)
            if got.tobytes() != exp.tobytes():
                raise FuzzMismatch(f"run (rep {rep}): {_diff(got, exp)}; reproduce with {tag}")
        if kernel == "tiled" and bits.shape[0]:
            tiles = ctx.num_tiles(sm)
            if tiles >= 2:
                parts = [ctx.run(sm, wps, d_bits, thr, tile_range=r)
                         for r in tile_partition(tiles, w)]
                merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
                if merged.tobytes() != exp.tobytes():
                    raise FuzzMismatch(f"tile ranges ({w}): {_diff(merged, exp)}; reproduce with {tag}")
            if k == 1 and n >= 2:
                ctx.invalidate()
                merged = _staged(ctx, sm, wps, d_bits, thr, max(len(exp), 1) + 8, n, world, chunks,
                                 streams)
                if merged.tobytes() != exp.tobytes():
                    raise FuzzMismatch(f"staged (world {world}, chunks {chunks}, streams {streams}): "
                                       f"{_diff(merged, exp)}; reproduce with {tag}")
        ran += 1
        if log and case % 25 == 0:
            log(f"run_general seed {seed} case {case} ok ({time.time() - t0:.0f}s)")
    ctx.set_option("reuse_prepared", 0)
    ctx.set_option("filter_quadrant_cap", 384)
    ctx.set_option("filter_cand_cap", 1 << 25)
    ctx.set_option("filter_split_min_steps", 8)
    ctx.set_option("filter_check0", 1)
    ctx.set_option("filter_check1", 1)
    ctx.set_option("filter_check_emit", 64)
    ctx.set_option("filter_rotate", 1)
    ctx.set_option("filter_rotate_min_steps", 128)
    ctx.set_option("filter_rotate_min_tiles", 2048)
    ctx.set_option("filter_persistent", 0)
    ctx.set_option("filter_persistent_min_tiles", 2048)
    ctx.set_option("filter_check_min_steps", 64)
    ctx.set_option("filter_sort", 1)
    ctx.set_option("filter_lazy_codes", 1)
    return ran


def run_split(ctx, seed: int, cases: int, first_case: int = 0, log=None) -> int:
    """The matrix-core variants' remainder splitting (king_mfma.hip): blocks large
    enough that pieces of k-steps, scratch slabs and tickets are really exercised
    -- whole blocks, tile sub-ranges and staged rectangles on several streams at
    once (tools/fuzz_split.py)."""
    import torch
    import cuking_amd
    from cuking_amd.dist import tile_partition
    from cuking_amd.synth import cohort_to_device, plan_cohort
    from oracle import pyoracle

    rng = np.random.default_rng(seed)
    matrix_variants = [v for v in range(ctx.lib.cuking_num_variants())
                       if "mfma" in ctx.variant_name(v)]
    ctx.set_kernel("tiled")
    t0, ran = time.time(), 0
    for case in range(cases):
        n = int(rng.integers(130, 3000))
        m = int(rng.integers(3000, 40000))
        thr = float(rng.choice([0.03, 0.0884, 0.3]))
        mode = int(rng.choice([-1, -1, 0, 1]))
        wgs = int(rng.choice([3, 16, 64, 256, 256]))
        w = int(rng.integers(2, 6))
        world = int(rng.integers(1, 5))
        chunks = int(rng.integers(1, 6))
        streams = [int(rng.integers(1, 4)) for _ in range(world)]
        variant = int(rng.choice(matrix_variants))
        if case < first_case:
            continue
        tag = dict(fuzzer="run_split", seed=seed, case=case, n=n, m=m, thr=thr, counts_mode=mode,
                   split_wgs=wgs, variant=variant)
        ctx.set_option("variant", variant)
        ctx.set_option("split_wgs", wgs)
        ctx.set_option("counts_mode", mode)
        ctx.invalidate()
        cohort = plan_cohort(n, seed * 1000 + case)
        kind, pa, pb = cohort_to_device(cohort, ctx.device)
        wps = cuking_amd.words_per_sample(m)
        d_bits = torch.zeros((n, wps), dtype=torch.int64, device=f"cuda:{ctx.device}")
        ctx.synth_bitset(seed * 1000 + case, kind, pa, pb, 0, n, m, out=d_bits)
        torch.cuda.synchronize()
        bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
        exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), bits, thr, threads=16)
        sm = cuking_amd.Submatrix(n)
        for rep in range(2):
            got = ctx.run(sm, wps, d_bits, thr)
            if got.tobytes() != exp.tobytes():
                raise FuzzMismatch(f"run (rep {rep}): {_diff(got, exp)}; reproduce with {tag}")
        tiles = ctx.num_tiles(sm)
        parts = [ctx.run(sm, wps, d_bits, thr, tile_range=r) for r in tile_partition(tiles, w)]
        merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
        if merged.tobytes() != exp.tobytes():
            raise FuzzMismatch(f"tile ranges ({w}): {_diff(merged, exp)}; reproduce with {tag}")
        merged = _staged(ctx, sm, wps, d_bits, thr, max(len(exp), 1) + 8, n, world, chunks, streams)
        if merged.tobytes() != exp.tobytes():
            raise FuzzMismatch(f"staged (world {world}, chunks {chunks}, streams {streams}): "
                               f"{_diff(merged, exp)}; reproduce with {tag}")
        ran += 1
        if log and case % 5 == 0:
            log(f"run_split seed {seed} case {case} ok, {len(exp)} records ({time.time() - t0:.0f}s)")
    return ran


def run_stress(ctx, reps: int, thr: float = 0.03, log=None):
    """One staged configuration repeated `reps` times per (form, split, streams)
    combination of every matrix-core variant: rare, timing-dependent failures
    (tools/stress_split.py).  Returns [(label, wrong, reps), ...]."""
    import torch
    import cuking_amd
    from cuking_amd.synth import cohort_to_device, plan_cohort
    from oracle import pyoracle

    ctx.set_kernel("tiled")
    n, m, chunks = 1015, 33744, 3
    cohort = plan_cohort(n, 4242)
    kind, pa, pb = cohort_to_device(cohort, ctx.device)
    wps = cuking_amd.words_per_sample(m)
    d_bits = torch.zeros((n, wps), dtype=torch.int64, device=f"cuda:{ctx.device}")
    ctx.synth_bitset(4242, kind, pa, pb, 0, n, m, out=d_bits)
    torch.cuda.synchronize()
    bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
    exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), bits, thr, threads=16)
    sm = cuking_amd.Submatrix(n)
    matrix_variants = [v for v in range(ctx.lib.cuking_num_variants())
                       if "mfma" in ctx.variant_name(v)]
    out = []
    for variant, mode, wgs, streams in itertools.product(matrix_variants, (1, 0), (16, 0, 256),
                                                         (1, 3)):
        ctx.set_option("variant", variant)
        ctx.set_option("counts_mode", mode)
        ctx.set_option("split_wgs", wgs)
        ctx.invalidate()
        bad, first = 0, ""
        for _ in range(reps):
            got = _staged(ctx, sm, wps, d_bits, thr, len(exp) + 8, n, 1, chunks, [streams])
            if got.tobytes() != exp.tobytes():
                bad += 1
                first = first or _diff(got, exp)
        label = (f"variant {ctx.variant_name(variant)} form {'full' if mode else 'lean'} "
                 f"split_wgs {wgs} streams {streams}")
        out.append((label, bad, reps, first))
        if log:
            log(f"run_stress {label}: {bad} of {reps} wrong {first}")
    return out
