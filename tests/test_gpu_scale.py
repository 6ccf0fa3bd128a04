"""BASELINE.json configs[2], [3] and [4] on one MI355X, at their full sizes.

The oracle cannot evaluate 5e9 .. 2.7e11 pairs, so parity at these sizes rests
on (a) sub-blocks of the GPU's own output re-computed by the oracle from the
same bitset (first / last / off-diagonal / founder-relative boundary tiles),
(b) size-independent properties: every planted relative reported, strict
threshold, i < j, sorted and unique, idempotence, identical records from an
independent kernel (VALU popcount variant), the union of tile ranges, and
(c) 64-bit indexing: configs[4] holds 734,000 x 6,250 u64 words, so the
samples at the far end sit beyond element offset 2^32 of the bitset
(cuking.cu:205-208, :514) and beyond byte offset 2^32 of the kernel layout
(92 GB in the default kernel's one-code-per-site form with its het-only copy, beside the
36.7 GB bitset).

Times on MI355X (round 1): configs[2] 0.64 s, configs[3] geometry 8.9 s,
configs[4] last tiles 0.6 s, whole configs[4] triangle 72 s
(CUKING_SKIP_WHOLE_C4=1 leaves the last one out of a tuning run; the driver's
run has it).
"""
import ctypes as C
import os
import time

import numpy as np
import pytest

import cuking_amd
from cuking_amd.synth import DEFAULT_SEED, cohort_to_device, plan_cohort

pytestmark = pytest.mark.gpu

MFMA, MFMA4, MFMA5, VALU_T64, VALU_T128 = 7, 6, 5, 1, 2   # default (filter), four / five products, VALU


def synthesise(ctx, n, m):
    import torch
    cohort = plan_cohort(n, DEFAULT_SEED)
    kind, pa, pb = cohort_to_device(cohort)
    bits = ctx.synth_bitset(DEFAULT_SEED, kind, pa, pb, 0, n, m)
    torch.cuda.synchronize()
    return cohort, bits


def release(*tensors):
    import torch
    del tensors
    torch.cuda.empty_cache()


def oracle_block(oracle, bits, thr, rows, cols):
    """Sorted oracle records of the pairs rows x cols (global sample indices)."""
    (a0, a1), (b0, b1) = rows, cols
    host = bits[a0:a1].cpu().numpy().view(np.uint64)
    if rows != cols:
        host = np.concatenate([host, bits[b0:b1].cpu().numpy().view(np.uint64)])
    osm = oracle.Submatrix(a0, a1, b0, b1)
    exp, ovf, _ = oracle.compute(osm, np.ascontiguousarray(host), thr, threads=16)
    assert ovf == 0
    return exp


def check_blocks(oracle, bits, res, thr, blocks):
    pairs = 0
    for rows, cols in blocks:
        (a0, a1), (b0, b1) = rows, cols
        exp = oracle_block(oracle, bits, thr, rows, cols)
        sel = res[(res["sample_i"] >= a0) & (res["sample_i"] < a1) &
                  (res["sample_j"] >= b0) & (res["sample_j"] < b1)]
        assert sel.tobytes() == exp.tobytes(), (rows, cols, len(sel), len(exp))
        pairs += ((a1 - a0) * (a1 - a0 - 1) // 2 if rows == cols
                  else (a1 - a0) * (b1 - b0))
    return pairs


def check_properties(res, cohort, thr, n, relations=("dup", "po", "sib")):
    got = {(int(r["sample_i"]), int(r["sample_j"])) for r in res}
    want = {(min(a, b), max(a, b)) for a, b, rel in cohort.planted if rel in relations}
    assert want and want <= got, f"{len(want - got)} planted relatives missing"
    assert np.all(res["kin"] > np.float32(thr)) and np.all(res["kin"] <= np.float32(0.5))
    assert np.all(res["sample_i"] < res["sample_j"]) and np.all(res["sample_j"] < n)
    key = res["sample_i"].astype(np.int64) * n + res["sample_j"]
    assert np.all(np.diff(key) > 0)                 # sorted (cuking.cu:761-765), unique
    dups = res[np.isin(key, [min(a, b) * n + max(a, b)
                             for a, b, rel in cohort.planted if rel == "dup"])]
    # a duplicate differs from its original only where either is missing:
    # kin exactly 0.5, no opposing homozygotes, no site with exactly one het
    assert len(dups) and np.all(dups["kin"] == np.float32(0.5))
    assert np.all(dups["ibs0"] == 0) and np.all(dups["ibs1"] == 0)


def test_c2_100k_x_100k_whole_triangle(ctx, oracle):
    """configs[2]: 100k samples x 100k sites, 4,999,950,000 pairs, one call."""
    n, m, thr = 100_000, 100_000, 0.0884             # default threshold, cuking.cu:43
    ctx.set_kernel("tiled")
    ctx.set_option("variant", MFMA)
    ctx.set_option("counts_mode", -1)
    cohort, bits = synthesise(ctx, n, m)
    wps = cuking_amd.words_per_sample(m)
    sm = cuking_amd.Submatrix(n)
    assert sm.NumPairs() == 4_999_950_000
    res = ctx.run(sm, wps, bits, thr, max_results=4 << 20)
    check_properties(res, cohort, thr, n)
    nf = cohort.num_founders
    blocks = [((0, 128), (0, 128)), ((n - 256, n), (n - 256, n)),
              ((50_000, 50_128), (n - 192, n)), ((nf - 64, nf + 64), (nf - 64, nf + 64))]
    assert check_blocks(oracle, bits, res, thr, blocks) > 70_000
    # idempotence, an independent kernel (VALU popcount, 64-sample tiles), and the
    # other form (five sums for every pair): identical bytes
    assert ctx.run(sm, wps, bits, thr, max_results=4 << 20).tobytes() == res.tobytes()
    ctx.set_option("variant", VALU_T64)
    assert ctx.run(sm, wps, bits, thr, max_results=4 << 20).tobytes() == res.tobytes()
    ctx.set_option("variant", MFMA5)          # ... and the five- and four-product matrix-core kernels
    assert ctx.run(sm, wps, bits, thr, max_results=4 << 20).tobytes() == res.tobytes()
    ctx.set_option("variant", MFMA4)
    assert ctx.run(sm, wps, bits, thr, max_results=4 << 20).tobytes() == res.tobytes()
    ctx.set_option("variant", MFMA)
    # ... the filter variant with every quadrant that has a candidate handed to the
    # four-product kernel, and with a candidate list of 100 entries
    for key, value, back in (("filter_quadrant_cap", 0, 384), ("filter_cand_cap", 100, 1 << 25)):
        ctx.set_option(key, value)
        assert ctx.run(sm, wps, bits, thr, max_results=4 << 20).tobytes() == res.tobytes(), key
        ctx.set_option(key, back)
    ctx.set_option("counts_mode", 1)
    assert ctx.run(sm, wps, bits, thr, max_results=4 << 20).tobytes() == res.tobytes()
    ctx.set_option("counts_mode", -1)
    # union of three tile ranges == the whole block
    tiles = ctx.num_tiles(sm)
    cuts = [0, tiles // 3, tiles // 3 + 1000, tiles]
    parts = [ctx.run(sm, wps, bits, thr, max_results=4 << 20, tile_range=(a, b), sort=False)
             for a, b in zip(cuts[:-1], cuts[1:])]
    merged = cuking_amd.sort_results(np.concatenate(parts))
    assert merged.tobytes() == res.tobytes()
    release(bits)


def test_c3_300k_x_150k_whole_triangle_on_one_gpu(ctx, oracle):
    """configs[3] geometry (8 GPUs in BASELINE) on ONE GPU: 4.5e10 pairs."""
    n, m, thr = 300_000, 150_000, 0.0884
    ctx.set_kernel("tiled")
    ctx.set_option("variant", MFMA)
    ctx.set_option("counts_mode", -1)
    cohort, bits = synthesise(ctx, n, m)
    wps = cuking_amd.words_per_sample(m)
    assert bits.numel() * 8 == 11_251_200_000          # 11.25 GB
    sm = cuking_amd.Submatrix(n)
    assert sm.NumPairs() == 44_999_850_000
    res = ctx.run(sm, wps, bits, thr, max_results=4 << 20)
    check_properties(res, cohort, thr, n)
    nf = cohort.num_founders
    blocks = [((0, 128), (0, 128)), ((n - 256, n), (n - 256, n)),
              ((150_000, 150_128), (n - 192, n)), ((nf - 64, nf + 64), (nf - 64, nf + 64))]
    assert check_blocks(oracle, bits, res, thr, blocks) > 70_000
    # what 8 ranks would compute (equal contiguous tile ranges, cuking_amd.dist):
    # rank 5's share, from the same call the multi-GPU pass makes
    # (tile bounds name samples only in an unsorted layout: include/cuking_amd.h)
    from cuking_amd.dist import tile_partition
    ctx.set_option("filter_sort", 0)
    b, e = tile_partition(ctx.num_tiles(sm), 8)[5]
    part = ctx.run(sm, wps, bits, thr, max_results=4 << 20, tile_range=(b, e))
    lib, rb, re_, cb, ce = ctx.lib, *(C.c_uint32() for _ in range(4))
    for t in (b, (b + e) // 2, e - 1):
        assert lib.cuking_tile_bounds(ctx.handle, C.byref(sm.c), t, C.byref(rb), C.byref(re_),
                                      C.byref(cb), C.byref(ce)) == 0
        blk = ((rb.value, re_.value), (cb.value, ce.value))
        check_blocks(oracle, bits, part, thr, [blk])
        check_blocks(oracle, bits, res, thr, [blk])
    keys = set(zip(part["sample_i"].tolist(), part["sample_j"].tolist()))
    assert keys <= set(zip(res["sample_i"].tolist(), res["sample_j"].tolist()))
    ctx.set_option("filter_sort", 1)
    release(bits)


@pytest.fixture(scope="module")
def c4(ctx):
    """configs[4] geometry: 734k samples x 200k sites = 36.7 GB bitset."""
    n, m = 734_000, 200_000
    ctx.set_kernel("tiled")
    ctx.set_option("variant", MFMA)
    ctx.set_option("counts_mode", -1)
    cohort, bits = synthesise(ctx, n, m)
    yield n, m, cohort, bits
    release(bits)


def test_c4_734k_x_200k_far_end_beyond_2_32_elements(ctx, oracle, c4):
    """The LAST tiles of the enumeration and the far-corner rectangle: their
    samples start beyond u64 element 2^32 of the bitset, and the kernel layout is
    36.7 GB.  IBS0/1/2 are part of every record compared (configs[4] wording)."""
    import torch
    n, m, cohort, bits = c4
    thr = 0.05
    wps = cuking_amd.words_per_sample(m)
    assert wps == 6250 and bits.numel() * 8 == 36_700_000_000
    first_far = (1 << 32) // wps + 1
    assert first_far < n - 40_000           # > 40k samples live beyond element 2^32
    sm = cuking_amd.Submatrix(n)
    tiles, tile = ctx.num_tiles(sm), ctx.tile_samples()
    take = 150_000
    ctx.set_option("filter_sort", 0)        # (tile bounds name samples only in an unsorted layout)
    res_hi = ctx.run(sm, wps, bits, thr, max_results=4 << 20, tile_range=(tiles - take, tiles))
    assert len(res_hi) > 1000
    rb, re_, cb, ce = (C.c_uint32() for _ in range(4))
    seen, far = 0, 0
    for t in (tiles - 1, tiles - 2, tiles - take, tiles - take // 2, tiles - 777):
        assert ctx.lib.cuking_tile_bounds(ctx.handle, C.byref(sm.c), t, C.byref(rb),
                                          C.byref(re_), C.byref(cb), C.byref(ce)) == 0
        far += rb.value >= first_far and cb.value >= first_far
        seen += check_blocks(oracle, bits, res_hi, thr,
                             [((rb.value, re_.value), (cb.value, ce.value))])
    assert far >= 2 and seen > 40_000
    ctx.set_option("filter_sort", 1)
    # far-corner rectangle through the staged operator
    lo = (n // tile - 20) * tile
    assert lo > first_far
    results = torch.zeros((1 << 20, 6), dtype=torch.int32, device="cuda:0")
    idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    exp = oracle_block(oracle, bits, thr, (lo, n), (lo, n))
    assert len(exp) > 100 and np.all(exp["ibs0"] + exp["ibs1"] + exp["ibs2"] <= m)
    # ... by the default kernel and by the independent VALU kernel (128-sample tiles)
    try:
        for variant in (MFMA, VALU_T128):
            ctx.set_option("variant", variant)
            idx.zero_()
            ctx.prepare_samples(sm, wps, bits, lo, n)
            ctx.compute_king_rect(sm, wps, bits, (lo, n), (lo, n), thr, 1 << 20, results,
                                  idx[0:1], idx[1:2])
            torch.cuda.synchronize()
            cnt, ovf = idx.tolist()
            assert ovf == 0
            recs = cuking_amd.sort_results(results[:cnt].cpu().numpy().view(np.uint32).reshape(-1)
                                           .view(cuking_amd.KING_RESULT_DTYPE).copy())
            assert recs.tobytes() == exp.tobytes(), variant
    finally:
        ctx.set_option("variant", MFMA)


@pytest.mark.skipif(os.environ.get("CUKING_SKIP_WHOLE_C4") == "1",
                    reason="tuning run: the 72 s whole-triangle pass is left out")
def test_c4_734k_x_200k_whole_triangle_on_one_gpu(ctx, oracle, c4):
    """All 269,377,633,000 pairs of configs[4] in one call on one GPU."""
    n, m, cohort, bits = c4
    thr = 0.05
    wps = cuking_amd.words_per_sample(m)
    sm = cuking_amd.Submatrix(n)
    assert sm.NumPairs() == 269_377_633_000
    t0 = time.perf_counter()
    res = ctx.run(sm, wps, bits, thr, max_results=8 << 20)
    print(f"configs[4] whole triangle: {time.perf_counter() - t0:.1f} s, {len(res)} records")
    check_properties(res, cohort, thr, n, relations=("dup", "po", "sib", "half"))
    tile = ctx.tile_samples()
    lo = (n // tile - 20) * tile
    nf = cohort.num_founders
    blocks = [((lo, n), (lo, n)), ((0, 128), (0, 128)), ((0, 128), (n - 128, n)),
              ((nf - 64, nf + 64), (nf - 64, nf + 64)), ((367_000, 367_128), (n - 192, n))]
    assert check_blocks(oracle, bits, res, thr, blocks) > 3_000_000
