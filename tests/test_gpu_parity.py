"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Integer sums and IBS0/1/2 bit-exact; kin bit-exact
in float32 (the reference's two-rounding expression).

PARITY UNPINNED by the reference (it ships no vectors): the
oracle is pinned by the hand-checked KAT and the naive numpy oracle instead
(tests/test_oracle.py)."""
import ctypes as C
import json

import numpy as np
import pytest

import cuking_amd
from cuking_amd import _lib
from cuking_amd.synth import DEFAULT_SEED, cohort_to_device, plan_cohort
from conftest import GOLDEN, random_genotypes

pytestmark = pytest.mark.gpu

NUM_VARIANTS = 8   # 0-4 VALU shapes, 5 / 6 / 7 matrix cores: five / four products, filter (cuking_variant_name)
MFMA_VARIANTS = [5, 6, 7]
KERNELS = [("stream", 0)] + [("tiled", v) for v in range(NUM_VARIANTS)]


def select(ctx, kernel, variant, counts_mode=-1):
    ctx.set_kernel(kernel)
    ctx.set_option("counts_mode", counts_mode)
    if kernel == "tiled":
        ctx.set_option("variant", variant)


def counts_from_oracle(oracle, sm_tuple, bits):
    osm = oracle.Submatrix(*sm_tuple)
    return oracle.all_pairs(osm, bits)


def check_counts(ctx, oracle, sm, bits_host):
    """All six sums of every pair == oracle."""
    d_bits = ctx.upload_bitset(bits_host)
    got = ctx.compute_counts(sm, bits_host.shape[1], d_bits)
    oi, oj, oc, _ = counts_from_oracle(oracle, sm.as_tuple(), bits_host)
    sel = got[oi - sm.i_begin, oj - sm.j_begin]
    for name in oc.dtype.names:
        assert np.array_equal(sel[name], oc[name]), name
    return len(oi)


def test_library_is_the_hip_one(ctx):
    # The context exists => a gfx950 device and the in-tree .so are in use.
    assert _lib.LIB_PATH.exists() and cuking_amd.device_count() >= 1


@pytest.mark.parametrize("kernel,variant", KERNELS)
def test_kat(ctx, oracle, kernel, variant):
    select(ctx, kernel, variant)
    kat = json.loads((GOLDEN / "kat_4x10.json").read_text())
    geno = np.array(kat["genotypes"], dtype=np.int8)
    sm = cuking_amd.Submatrix(4)
    bits = cuking_amd.new_host_bitset(sm, 10)
    col, row = np.nonzero(geno >= 0)
    cuking_amd.pack_host(sm, bits, row, col, geno[col, row])
    d_bits = ctx.upload_bitset(bits)
    counts = ctx.compute_counts(sm, bits.shape[1], d_bits)
    for p in kat["pairs"]:
        c = counts[p["i"], p["j"]]
        for name in counts.dtype.names:
            assert int(c[name]) == p[name], (p, name)
    res = ctx.run(sm, bits.shape[1], d_bits, kat["thresholded"]["kin_threshold"])
    names = kat["samples"]
    got = [[names[r["sample_i"]], names[r["sample_j"]], float(r["kin"]),
            int(r["ibs0"]), int(r["ibs1"]), int(r["ibs2"])] for r in res]
    assert got == kat["thresholded"]["records"]
    assert res["kin"].view(np.uint32)[0] == 0x3F000000


@pytest.mark.parametrize("kernel,variant", KERNELS)
def test_synth_golden_fixture(ctx, kernel, variant):
    """Committed vectors (tests/golden/synth_96x700.json): device generator,
    records unsharded and per shard -- no oracle involved at run time."""
    import torch
    from test_oracle import golden_records, load_synth_golden
    select(ctx, kernel, variant)
    g, bits = load_synth_golden()
    n, m, thr = g["num_samples"], g["num_sites"], g["kin_threshold"]
    as_t = lambda a: torch.tensor(a, dtype=torch.int32, device="cuda:0")
    dev = ctx.synth_bitset(g["seed"], as_t(g["kind"]), as_t(g["pa"]), as_t(g["pb"]), 0, n, m)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), bits)
    got = ctx.run(cuking_amd.Submatrix(n), bits.shape[1], dev, thr)
    assert got.tobytes() == golden_records(g["records"]).tobytes()
    for shard, rows in enumerate(g["shards_split_factor_2"]):
        sm = cuking_amd.Submatrix(n, 2, shard)
        idx = list(range(sm.i_begin, sm.i_end))
        if sm.i_begin != sm.j_begin:
            idx += list(range(sm.j_begin, sm.j_end))
        local = ctx.upload_bitset(np.ascontiguousarray(bits[idx]))
        assert ctx.run(sm, bits.shape[1], local, thr).tobytes() == golden_records(rows).tobytes()


SHAPES = [(2, 1), (3, 31), (5, 32), (7, 33), (16, 64), (33, 65), (63, 255),
          (64, 256), (65, 257), (130, 1000), (257, 3000), (100, 513)]


@pytest.mark.parametrize("kernel,variant", KERNELS)
def test_all_six_counts_every_pair(ctx, oracle, kernel, variant):
    select(ctx, kernel, variant)
    rng = np.random.default_rng(1234)
    total = 0
    for n, m in SHAPES:
        geno = random_genotypes(rng, n, m, missing=0.07)
        if n > 6:
            geno[1] = -1          # a sample with nothing defined
            geno[2] = 0           # a sample without hets
            geno[5] = geno[3]     # duplicates
        sm = cuking_amd.Submatrix(n)
        bits = oracle.bitset_from_genotypes(geno)
        total += check_counts(ctx, oracle, sm, bits)
    assert total > 50000


@pytest.mark.parametrize("kernel,variant", KERNELS)
@pytest.mark.parametrize("thr", [-1e30, -0.3, 0.0, 0.0884, 0.2])
def test_thresholded_records_bit_exact(ctx, oracle, kernel, variant, thr):
    select(ctx, kernel, variant)
    rng = np.random.default_rng(99)
    n, m = 150, 777
    geno = random_genotypes(rng, n, m, missing=0.03)
    geno[40] = geno[10]
    geno[41, :400] = geno[11, :400]
    geno[77] = -1
    geno[78] = 0
    sm = cuking_amd.Submatrix(n)
    bits = oracle.bitset_from_genotypes(geno)
    exp, ovf, cnt = oracle.compute(oracle.submatrix(n), bits, thr)
    assert ovf == 0
    got = ctx.run(sm, bits.shape[1], ctx.upload_bitset(bits), thr)
    assert got.dtype == exp.dtype or got.dtype.descr == exp.dtype.descr
    assert got.tobytes() == exp.tobytes()   # i, j, kin bits, ibs0/1/2


@pytest.mark.parametrize("variant", range(NUM_VARIANTS))
@pytest.mark.parametrize("counts_mode", [0, 1])
@pytest.mark.parametrize("thr", [-1e30, 0.0, 0.1])
def test_lean_and_full_forms_agree_with_oracle(ctx, oracle, variant, counts_mode, thr):
    """counts_mode 0 = four sums + in-wave IBS2 recount for emitted pairs (here
    forced even when every pair is emitted), 1 = five sums for every pair."""
    select(ctx, "tiled", variant, counts_mode)
    rng = np.random.default_rng(123)
    for n, m, k, shard in ((97, 50, 1, 0), (200, 1300, 1, 0), (300, 257, 2, 1)):
        geno = random_genotypes(rng, n, m, missing=0.08)
        geno[n - 1] = geno[0]
        geno[n // 2] = -1
        osm = oracle.submatrix(n, k, shard)
        bits = oracle.bitset_from_genotypes(geno, osm)
        exp, _, _ = oracle.compute(osm, bits, thr)
        got = ctx.run(cuking_amd.Submatrix(n, k, shard), bits.shape[1],
                      ctx.upload_bitset(bits), thr)
        assert got.tobytes() == exp.tobytes(), (n, m, k, shard)
    select(ctx, "tiled", 0)


@pytest.mark.parametrize("kernel,variant", [("stream", 0), ("tiled", 0), ("tiled", 1), ("tiled", 5), ("tiled", 6)])
@pytest.mark.parametrize("k", [2, 3, 4])
def test_split_factor_shards(ctx, oracle, kernel, variant, k):
    """--split_factor / --shard_index (cuking.cu:46-52): every shard equals the
    oracle's shard; their union equals the unsharded run."""
    select(ctx, kernel, variant)
    rng = np.random.default_rng(11)
    n, m = 203, 450
    geno = random_genotypes(rng, n, m, missing=0.02)
    geno[150] = geno[20]
    geno[199] = geno[100]
    thr = -0.1
    full = ctx.run(cuking_amd.Submatrix(n), cuking_amd.words_per_sample(m),
                   ctx.upload_bitset(oracle.bitset_from_genotypes(geno)), thr)
    parts = []
    for shard in range(k * (k + 1) // 2):
        sm = cuking_amd.Submatrix(n, k, shard)
        osm = oracle.submatrix(n, k, shard)
        bits = oracle.bitset_from_genotypes(geno, osm)   # shard-local storage
        assert bits.shape[0] == sm.NumSamples()
        exp, _, _ = oracle.compute(osm, bits, thr)
        got = ctx.run(sm, bits.shape[1], ctx.upload_bitset(bits), thr)
        assert got.tobytes() == exp.tobytes(), shard
        check_counts(ctx, oracle, sm, bits)
        parts.append(got)
    merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
    assert merged.tobytes() == full.tobytes()


@pytest.mark.parametrize("variant", range(NUM_VARIANTS))
def test_tile_ranges_union(ctx, oracle, variant):
    """Pair-space sharding for multi-GPU: disjoint tile ranges == whole block."""
    select(ctx, "tiled", variant)
    rng = np.random.default_rng(5)
    n, m = 300, 200
    geno = random_genotypes(rng, n, m)
    bits = oracle.bitset_from_genotypes(geno)
    sm = cuking_amd.Submatrix(n)
    d_bits = ctx.upload_bitset(bits)
    whole = ctx.run(sm, bits.shape[1], d_bits, -0.05)
    tiles = ctx.num_tiles(sm)
    assert tiles >= 3
    from cuking_amd.dist import tile_partition
    for world in (2, 3):
        parts = [ctx.run(sm, bits.shape[1], d_bits, -0.05, tile_range=r)
                 for r in tile_partition(tiles, world)]
        merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
        assert merged.tobytes() == whole.tobytes()
    with pytest.raises(cuking_amd.CukingError):
        ctx.run(sm, bits.shape[1], d_bits, -0.05, tile_range=(0, tiles + 1))


@pytest.mark.parametrize("variant", [0, 1, 4, 5, 6, 7])
@pytest.mark.parametrize("world,chunks", [(1, 1), (1, 4), (2, 3), (3, 8), (8, 5)])
def test_staged_rectangles_union(ctx, oracle, variant, world, chunks):
    """The overlapped multi-GPU schedule (chunked arrival, row bands, rectangle
    launches on side streams) replayed rank by rank on one GPU == oracle."""
    import torch
    from cuking_amd.dist import GpuStagedOps, staged_schedule
    select(ctx, "tiled", variant)
    rng = np.random.default_rng(31)
    n, m = 700, 333
    geno = random_genotypes(rng, n, m, missing=0.04)
    geno[650] = geno[3]
    geno[699] = geno[320]
    bits = oracle.bitset_from_genotypes(geno)
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, -0.08)
    sm = cuking_amd.Submatrix(n)
    tile = ctx.tile_samples()
    parts = []
    for rank in range(world):
        # the receive buffer starts as garbage and fills chunk by chunk
        d_bits = torch.full((n, bits.shape[1]), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64,
                            device="cuda:0")
        src = torch.from_numpy(bits.view(np.int64))
        ops = GpuStagedOps(ctx, sm, bits.shape[1], d_bits, -0.08, 250000)
        ops.begin()
        for (c0, c1), rect in staged_schedule(n, tile, world, rank, chunks):
            d_bits[c0:c1].copy_(src[c0:c1])          # "chunk arrives"
            if rect is None:
                continue
            ops.prepare(c0, c1)
            ops.compute_rect(*rect)
        res, count, ovf = ops.finish()
        assert ovf == 0
        parts.append(res[:count].cpu().numpy().view(np.uint32).reshape(-1).view(
            cuking_amd.KING_RESULT_DTYPE).copy())
    merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
    assert merged.tobytes() == exp.tobytes()


def test_staged_api_errors(ctx, oracle):
    import torch
    select(ctx, "tiled", 0)
    n = 200
    bits = oracle.bitset_from_genotypes(random_genotypes(np.random.default_rng(1), n, 100))
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    res = torch.zeros((10, 6), dtype=torch.int32, device="cuda:0")
    idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    with pytest.raises(cuking_amd.CukingError):      # not tile aligned
        ctx.prepare_samples(sm, bits.shape[1], d_bits, 10, 200)
    with pytest.raises(cuking_amd.CukingError):      # outside the block
        ctx.prepare_samples(sm, bits.shape[1], d_bits, 0, 264)
    off = cuking_amd.Submatrix(n, 2, 1)              # off-diagonal block
    with pytest.raises(cuking_amd.CukingError):
        ctx.prepare_samples(off, bits.shape[1], d_bits, 0, 100)
    ctx.prepare_samples(sm, bits.shape[1], d_bits, 0, 200)
    with pytest.raises(cuking_amd.CukingError):
        ctx.compute_king_rect(sm, bits.shape[1], d_bits, (0, 100), (0, 200), 0.0, 10,
                              res, idx[0:1], idx[1:2])


@pytest.mark.parametrize("kernel,variant", [("stream", 0), ("tiled", 0), ("tiled", 5), ("tiled", 6)])
def test_result_overflow(ctx, oracle, kernel, variant):
    """cuking.cu:297-313, :747-751: overflow is an error, never truncation."""
    import torch
    select(ctx, kernel, variant)
    rng = np.random.default_rng(2)
    geno = random_genotypes(rng, 80, 300)
    bits = oracle.bitset_from_genotypes(geno)
    sm = cuking_amd.Submatrix(80)
    d_bits = ctx.upload_bitset(bits)
    exp, _, n_all = oracle.compute(oracle.submatrix(80), bits, -5.0)
    assert n_all > 1000
    with pytest.raises(cuking_amd.ResourceExhaustedError) as e:
        ctx.run(sm, bits.shape[1], d_bits, -5.0, max_results=1000)
    assert "--max_results" in str(e.value)
    # raw contract: the counter keeps counting, the flag is set, and the
    # first max_results slots hold valid records
    results = torch.zeros((1000, 6), dtype=torch.int32, device="cuda:0")
    idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    ctx.compute_king(sm, bits.shape[1], d_bits, -5.0, 1000, results, idx[0:1], idx[1:2])
    torch.cuda.synchronize()
    assert idx.tolist() == [n_all, 1]
    recs = results.cpu().numpy().view(np.uint32).reshape(-1).view(cuking_amd.KING_RESULT_DTYPE)
    keys = {(int(r["sample_i"]), int(r["sample_j"])): r for r in exp}
    for r in recs:
        assert keys[(int(r["sample_i"]), int(r["sample_j"]))].tobytes() == r.tobytes()
    # exactly enough room: no overflow, all records
    got = ctx.run(sm, bits.shape[1], d_bits, -5.0, max_results=n_all)
    assert got.tobytes() == exp.tobytes()


@pytest.mark.parametrize("kernel,variant", [("stream", 0), ("tiled", 0), ("tiled", 2), ("tiled", 5), ("tiled", 6)])
def test_long_ranges_are_split_into_several_launches(ctx, oracle, kernel, variant):
    """One launch may not exceed 2^32 - 1 threads (beyond that HIP truncates
    silently): long tile / pair ranges go out as several launches.  The cap is
    lowered here so that a small block needs many launches."""
    select(ctx, kernel, variant)
    rng = np.random.default_rng(77)
    n, m = 500, 300
    geno = random_genotypes(rng, n, m)
    geno[499] = geno[7]
    bits = oracle.bitset_from_genotypes(geno)
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, 0.1)
    d_bits = ctx.upload_bitset(bits)
    try:
        for cap in (1, 7, 1000):
            ctx.set_option("max_launch_blocks", cap)
            got = ctx.run(cuking_amd.Submatrix(n), bits.shape[1], d_bits, 0.1)
            assert got.tobytes() == exp.tobytes(), cap
    finally:
        ctx.set_option("max_launch_blocks", 0)


@pytest.mark.parametrize("mv", MFMA_VARIANTS)
@pytest.mark.parametrize("thr,counts_mode", [(0.1, -1), (-1e30, 0), (-1e30, 1)])
def test_matrix_core_remainder_split(ctx, oracle, thr, counts_mode, mv):
    """The matrix-core variant cuts a remainder of tiles (fewer than one per CU)
    into equal pieces of k-steps over all CUs; partial sums meet in a scratch
    slab.  `split_wgs` stands in for the CU count so that a small block
    exercises pieces that span tile boundaries, whole tiles and no split."""
    select(ctx, "tiled", mv, counts_mode)
    rng = np.random.default_rng(99)
    n, m = 700, 9000
    geno = random_genotypes(rng, n, m, missing=0.03)
    geno[699] = geno[5]
    geno[300] = geno[128]
    bits = oracle.bitset_from_genotypes(geno)
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr)
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    try:
        for wgs in (0, 4, 5, 8, 16, 20, 64):
            ctx.set_option("split_wgs", wgs)
            for _ in range(2):      # the slab must be clean again after a launch
                got = ctx.run(sm, bits.shape[1], d_bits, thr)
                assert got.tobytes() == exp.tobytes(), wgs
        # a shard (off-diagonal block) and a tile sub-range
        ctx.set_option("split_wgs", 6)
        osm = oracle.submatrix(n, 2, 1)
        bits2 = oracle.bitset_from_genotypes(geno, osm)
        exp2, _, _ = oracle.compute(osm, bits2, thr)
        got2 = ctx.run(cuking_amd.Submatrix(n, 2, 1), bits2.shape[1],
                       ctx.upload_bitset(bits2), thr)
        assert got2.tobytes() == exp2.tobytes()
    finally:
        ctx.set_option("split_wgs", 256)
        select(ctx, "tiled", 0)


@pytest.mark.parametrize("mv,sites", [(5, (1 << 24) - 256), (5, (1 << 24) + 512),
                                      (6, (1 << 22) - 256), (6, (1 << 22) + 512),
                                      (6, (1 << 24) + 512),
                                      (7, (1 << 22) - 256), (7, (1 << 22) + 512),
                                      (7, (1 << 24) + 512)])
def test_matrix_core_float_limit(ctx, oracle, mv, sites):
    """The matrix-core variants count in float32: exact while every sum stays
    below 2^24.  Just under the limit with sums as large as they get (every
    site het in every sample) it must still be bit-exact; just over it the
    library switches to the VALU variant with the same tile geometry.  The
    four-product variant decides kinship on an integer that equals the reference's
    float expression below 2^22 sites: from there on it hands over to the
    five-product one; so does the filter variant (its accumulators hold 4 q), on the
    quadrants of its 256-sample tiles."""
    select(ctx, "tiled", mv)
    n = 20
    wps = cuking_amd.words_per_sample(sites)
    rng = np.random.default_rng(3)
    bits = np.zeros((n, wps), dtype=np.uint64)
    bits[:, : wps // 2] = ~np.uint64(0)                 # het plane: all het ...
    bits[:6, wps // 2:] = rng.integers(0, 1 << 63, size=(6, wps // 2), dtype=np.uint64)  # ... some missing
    bits[12] = rng.integers(0, 1 << 63, size=wps, dtype=np.uint64)
    bits[13] = bits[12]
    pad = wps // 2 * 64 - sites                          # padding sites stay missing
    if pad:
        tail = np.uint64((~np.uint64(0)) << np.uint64(64 - pad))
        bits[:, wps - 1] |= tail
        bits[:, wps // 2 - 1] |= tail
    osm = oracle.submatrix(n)
    exp, _, _ = oracle.compute(osm, bits, -1e30, threads=8)
    d_bits = ctx.upload_bitset(bits)
    got = ctx.run(cuking_amd.Submatrix(n), wps, d_bits, -1e30)
    assert got.tobytes() == exp.tobytes()
    assert int(exp["ibs2"].max()) + int(exp["ibs1"].max()) > sites // 2   # sums as large as they get
    # ... and with a threshold the lean forms (and the filter variant's bound) apply to
    ctx.set_option("counts_mode", 0)
    exp, _, _ = oracle.compute(osm, bits, 0.2, threads=8)
    assert len(exp) >= 1
    assert ctx.run(cuking_amd.Submatrix(n), wps, d_bits, 0.2).tobytes() == exp.tobytes()
    select(ctx, "tiled", 0)


@pytest.mark.parametrize("kernel,variant", [("tiled", 5), ("tiled", 6), ("tiled", 7), ("tiled", 0),
                                            ("stream", 0)])
def test_site_position_patterns(ctx, oracle, kernel, variant):
    """Genotype patterns that single out one site position: the matrix-core
    kernel expands sites by their position inside a 4-site nibble (position 3 is
    shifted, the others are masked), inside a 32-site word and inside a
    4-word quad / 256-site k-step -- every position, every genotype, must land
    in the right sum."""
    select(ctx, kernel, variant)
    m = 700                                   # not a multiple of 32, 64, 128 or 256
    rows = []
    for period, phase in [(4, 0), (4, 1), (4, 2), (4, 3), (32, 31), (64, 63), (128, 127),
                          (256, 255), (256, 0), (7, 3)]:
        for g_on, g_off in [(1, 0), (2, 0), (0, 2), (1, -1), (-1, 1), (2, 1)]:
            row = np.full(m, g_off, dtype=np.int8)
            row[phase::period] = g_on
            rows.append(row)
    for g in (0, 1, 2, -1):                   # constant samples
        rows.append(np.full(m, g, dtype=np.int8))
    single = np.zeros(m, dtype=np.int8)       # one het site at the very end
    single[m - 1] = 1
    rows.append(single)
    geno = np.stack(rows)
    bits = oracle.bitset_from_genotypes(geno)
    n = geno.shape[0]
    assert check_counts(ctx, oracle, cuking_amd.Submatrix(n), bits) == n * (n - 1) // 2
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, 0.1)
    got = ctx.run(cuking_amd.Submatrix(n), bits.shape[1], ctx.upload_bitset(bits), 0.1)
    assert got.tobytes() == exp.tobytes()
    select(ctx, "tiled", 0)


def test_options_round_trip(ctx):
    """cuking_ctx_get_option reads back what set_option stored; the default
    kernel variant is the matrix-core one."""
    fresh = cuking_amd.KingContext(0)
    try:
        assert fresh.get_option("variant") == 7 and fresh.variant_name() == "t256_mfma_fp4_filter"
        assert fresh.get_option("split_wgs") > 0 and fresh.get_option("counts_mode") == -1
        assert fresh.get_option("xcd_swizzle") == 2 and fresh.get_option("band_rows") == 0
        for key, value in (("variant", 2), ("band_rows", 9), ("counts_mode", 1), ("split_wgs", 0),
                           ("xcd_swizzle", 0)):
            fresh.set_option(key, value)
            assert fresh.get_option(key) == value
        assert fresh.tile_samples() == 128
        with pytest.raises(cuking_amd.CukingError):
            fresh.get_option("no_such_option")
    finally:
        fresh.close()


def test_appending_calls_share_one_buffer(ctx, oracle):
    """result_index is not reset by the call (cuking.cu:721-722 leaves that to
    the caller), so shards can append into one buffer."""
    import torch
    select(ctx, "tiled", 0)
    rng = np.random.default_rng(8)
    n, m = 120, 300
    geno = random_genotypes(rng, n, m)
    results = torch.zeros((20000, 6), dtype=torch.int32, device="cuda:0")
    idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    for shard in range(3):
        sm = cuking_amd.Submatrix(n, 2, shard)
        bits = oracle.bitset_from_genotypes(geno, oracle.submatrix(n, 2, shard))
        ctx.compute_king(sm, bits.shape[1], ctx.upload_bitset(bits), -0.2, 20000,
                         results, idx[0:1], idx[1:2])
        torch.cuda.synchronize()
    count, ovf = idx.tolist()
    exp, _, _ = oracle.compute(oracle.submatrix(n), oracle.bitset_from_genotypes(geno), -0.2)
    assert ovf == 0 and count == len(exp)
    recs = results[:count].cpu().numpy().view(np.uint32).reshape(-1).view(
        cuking_amd.KING_RESULT_DTYPE).copy()
    assert cuking_amd.sort_results(recs).tobytes() == exp.tobytes()


def test_empty_and_tiny_blocks(ctx, oracle):
    import torch
    select(ctx, "tiled", 0)
    # N=5, k=4: block 3 is empty after clamping
    for shard in range(10):
        sm = cuking_amd.Submatrix(5, 4, shard)
        osm = oracle.submatrix(5, 4, shard)
        geno = random_genotypes(np.random.default_rng(shard), 5, 40)
        bits = oracle.bitset_from_genotypes(geno, osm)
        if bits.shape[0] == 0:
            d_bits = torch.zeros(2, dtype=torch.int64, device="cuda:0")
        else:
            d_bits = ctx.upload_bitset(bits)
        got = ctx.run(sm, cuking_amd.words_per_sample(40), d_bits, -100.0)
        exp, _, _ = oracle.compute(osm, bits, -100.0)
        assert got.tobytes() == exp.tobytes()
    sm1 = cuking_amd.Submatrix(1)
    one = ctx.upload_bitset(oracle.bitset_from_genotypes(np.ones((1, 8), dtype=np.int8)))
    assert len(ctx.run(sm1, 2, one, -100.0)) == 0


def test_argument_errors(ctx, oracle):
    import torch
    bits = torch.zeros(10, dtype=torch.int64, device="cuda:0")
    sm = cuking_amd.Submatrix(100)
    with pytest.raises(ValueError):
        ctx.run(sm, 2, bits)                       # too small for the block
    with pytest.raises(cuking_amd.CukingError) as e:
        ctx.run(cuking_amd.Submatrix(2), 3, bits)  # odd words_per_sample
    assert e.value.status == _lib.ERR_INVALID_ARGUMENT
    with pytest.raises(cuking_amd.CukingError):
        ctx.set_option("variant", 99)


@pytest.mark.parametrize("k,shard", [(1, 0), (3, 1), (3, 3)])
def test_pack_device_matches_host_and_oracle(ctx, oracle, k, shard):
    import torch
    rng = np.random.default_rng(21)
    n, m = 90, 1500
    geno = random_genotypes(rng, n, m, missing=0.1)
    col, row = np.nonzero(geno >= 0)
    alt = geno[col, row].astype(np.int32)
    perm = rng.permutation(len(row))
    row, col, alt = row[perm], col[perm], alt[perm]
    sm = cuking_amd.Submatrix(n, k, shard)
    wps = cuking_amd.words_per_sample(m)
    host = cuking_amd.new_host_bitset(sm, m)
    cuking_amd.pack_host(sm, host, row, col, alt)
    d_bits = torch.full((sm.NumSamples(), wps), -1, dtype=torch.int64, device="cuda:0")
    status = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    ctx.pack_device(sm, wps, d_bits, torch.from_numpy(row).cuda(),
                    torch.from_numpy(col).cuda(), torch.from_numpy(alt).cuda(), status)
    torch.cuda.synchronize()
    assert int(status) == 0
    dev = d_bits.cpu().numpy().view(np.uint64)
    assert np.array_equal(dev, host)
    assert np.array_equal(dev, oracle.bitset_from_genotypes(geno, oracle.submatrix(n, k, shard)))
    # invalid genotype / row are reported, not silently packed
    bad_alt = alt.copy()
    bad_alt[3] = 7
    bad_row = row.copy()
    bad_row[5] = cuking_amd.padded_sites(m) + 64
    status.zero_()
    ctx.pack_device(sm, wps, d_bits, torch.from_numpy(bad_row).cuda(),
                    torch.from_numpy(col).cuda(), torch.from_numpy(bad_alt).cuda(), status)
    torch.cuda.synchronize()
    exp = (1 if sm.Contains(int(col[3])) else 0) | (2 if sm.Contains(int(col[5])) else 0)
    assert int(status) == exp


def test_synth_device_equals_oracle_twin(ctx, oracle):
    import torch
    cohort = plan_cohort(400, seed=77)
    assert cohort.num_founders < 400 and len(cohort.planted) > 10
    kind, pa, pb = cohort_to_device(cohort)
    for m in (1, 63, 64, 1000, 2049):
        dev = ctx.synth_bitset(77, kind, pa, pb, 0, 400, m)
        torch.cuda.synchronize()
        exp = oracle.synth_bitset(77, cohort.kind, cohort.pa, cohort.pb, 0, 400, m)
        assert np.array_equal(dev.cpu().numpy().view(np.uint64), exp), m
    # a slice of rows
    dev = ctx.synth_bitset(77, kind, pa, pb, 380, 400, 500)
    torch.cuda.synchronize()
    exp = oracle.synth_bitset(77, cohort.kind, cohort.pa, cohort.pb, 380, 400, 500)
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), exp)


# ---------------------------------------------------------------------------
# BASELINE configs[1] at full size (10k samples x 100k sites, threshold 0.05):
# the oracle cannot run 5e7 pairs x 50 KB in seconds, so parity is checked
# through sub-blocks and size-independent properties.
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c1(ctx):
    import torch
    n, m = 10000, 100000
    cohort = plan_cohort(n, DEFAULT_SEED)
    kind, pa, pb = cohort_to_device(cohort)
    bits = ctx.synth_bitset(DEFAULT_SEED, kind, pa, pb, 0, n, m)
    torch.cuda.synchronize()
    select(ctx, "tiled", 0)
    res = ctx.run(cuking_amd.Submatrix(n), cuking_amd.words_per_sample(m), bits, 0.05)
    return dict(n=n, m=m, cohort=cohort, bits=bits, res=res)


def test_c1_planted_relatives_found(c1):
    res = c1["res"]
    got = {(int(r["sample_i"]), int(r["sample_j"])): float(r["kin"]) for r in res}
    lo = {"dup": 0.45, "po": 0.2, "sib": 0.15, "half": 0.07}
    hi = {"dup": 0.5, "po": 0.3, "sib": 0.35, "half": 0.19}
    for a, b, rel in c1["cohort"].planted:
        i, j = min(a, b), max(a, b)
        assert (i, j) in got, (i, j, rel)
        assert lo[rel] <= got[(i, j)] <= hi[rel] + 1e-6, (rel, got[(i, j)])
    # unrelated founders stay below the threshold: output is dominated by
    # planted pairs and their close kin
    assert len(res) < 40 * len(c1["cohort"].planted)
    assert np.all(res["sample_i"] < res["sample_j"])
    assert np.all(res["kin"] > np.float32(0.05))
    s = res["ibs0"].astype(np.int64) + res["ibs1"] + res["ibs2"]
    assert np.all(s <= c1["m"]) and np.all(s > 0.9 * c1["m"])
    key = res["sample_i"].astype(np.int64) * c1["n"] + res["sample_j"]
    assert np.all(np.diff(key) > 0)  # sorted, no pair twice


def test_c1_subblocks_match_oracle(ctx, oracle, c1):
    """Diagonal sub-block of founders and the derived tail (relatives), plus
    the rectangle between them, re-computed by the oracle from the same bits."""
    import torch
    n, res, bits = c1["n"], c1["res"], c1["bits"]
    wps = bits.shape[1]
    nf = c1["cohort"].num_founders
    for (a0, a1), (b0, b1) in [((0, 96), (0, 96)), ((n - 160, n), (n - 160, n)),
                               ((64, 160), (n - 96, n))]:
        rows = bits[a0:a1].cpu().numpy().view(np.uint64)
        if (a0, a1) == (b0, b1):
            osm = oracle.Submatrix(a0, a1, a0, a1)
            host = np.ascontiguousarray(rows)
        else:
            osm = oracle.Submatrix(a0, a1, b0, b1)
            host = np.ascontiguousarray(np.concatenate(
                [rows, bits[b0:b1].cpu().numpy().view(np.uint64)]))
        exp, ovf, _ = oracle.compute(osm, host, 0.05)
        sel = res[(res["sample_i"] >= a0) & (res["sample_i"] < a1) &
                  (res["sample_j"] >= b0) & (res["sample_j"] < b1)]
        assert sel.tobytes() == exp.tobytes()
        # and every count of every pair of the sub-block through the ABI
        sm = cuking_amd.Submatrix.from_ranges(*osm.as_tuple())
        select(ctx, "tiled", 0)
        check_counts(ctx, oracle, sm, host)
    assert nf < n - 160 or True


def test_c1_kernels_and_variants_agree(ctx, c1):
    """Idempotence + every kernel shape produces the same records."""
    n, m, bits = c1["n"], c1["m"], c1["bits"]
    sm = cuking_amd.Submatrix(n)
    base = c1["res"].tobytes()
    for variant in range(NUM_VARIANTS):
        select(ctx, "tiled", variant)
        assert ctx.run(sm, bits.shape[1], bits, 0.05).tobytes() == base, variant
    # the streaming kernel on the whole cohort: 10000 x 2500 workgroups of 256
    # threads = 6.4e9 threads, i.e. more than one launch may hold
    select(ctx, "stream", 0)
    assert ctx.run(sm, bits.shape[1], bits, 0.05).tobytes() == base
    # ... and on the last 1500 samples (all relatives live there)
    lo = n - 1500
    select(ctx, "stream", 0)
    sub = cuking_amd.Submatrix.from_ranges(lo, n, lo, n)
    got = ctx.run(sub, bits.shape[1], bits[lo:].contiguous(), 0.05)
    res = c1["res"]
    exp = res[(res["sample_i"] >= lo)]
    assert got.tobytes() == exp.tobytes()
    select(ctx, "tiled", 0)


def test_c1_split_factor_union(ctx, c1):
    """README.md:80-89 style run: --split_factor=4 => 10 shards, union == whole."""
    import torch
    n, bits = c1["n"], c1["bits"]
    wps = bits.shape[1]
    select(ctx, "tiled", 0)
    parts = []
    for shard in range(10):
        sm = cuking_amd.Submatrix(n, 4, shard)
        if sm.i_begin == sm.j_begin:
            local = bits[sm.i_begin:sm.i_end]
        else:
            local = torch.cat([bits[sm.i_begin:sm.i_end], bits[sm.j_begin:sm.j_end]])
        parts.append(ctx.run(sm, wps, local.contiguous(), 0.05))
    merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
    assert merged.tobytes() == c1["res"].tobytes()


def test_out_of_memory_is_reported(ctx):
    """cuking.cu:113-118 exits the process on allocation failure; here it is a
    status the caller can handle."""
    import ctypes as C
    ptr = C.c_void_p()
    st = ctx.lib.cuking_device_alloc(ctx.handle, 1 << 50, C.byref(ptr))
    assert st == _lib.ERR_OUT_OF_MEMORY and not ptr.value
    assert b"hipMalloc" in ctx.lib.cuking_last_error()
    # the context is still usable afterwards
    st = ctx.lib.cuking_device_alloc(ctx.handle, 1 << 20, C.byref(ptr))
    assert st == 0 and ptr.value
    assert ctx.lib.cuking_device_free(ctx.handle, ptr) == 0


def test_streams_from_the_abi(ctx, oracle):
    """Hosts without their own stream objects (the C++ binary) create them
    through the ABI; work on such a stream is ordered and waitable."""
    import ctypes as C
    import torch
    select(ctx, "tiled", 0)
    stream = C.c_void_p()
    assert ctx.lib.cuking_stream_create(ctx.handle, C.byref(stream)) == 0 and stream.value
    geno = random_genotypes(np.random.default_rng(4), 130, 400)
    bits = oracle.bitset_from_genotypes(geno)
    d_bits = ctx.upload_bitset(bits)
    results = torch.zeros((20000, 6), dtype=torch.int32, device="cuda:0")
    idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    sm = cuking_amd.Submatrix(130)
    _lib.check(ctx.lib.cuking_compute_king(ctx.handle, C.byref(sm.c), bits.shape[1],
                                           d_bits.data_ptr(), 0.02, 20000, results.data_ptr(),
                                           idx[0:1].data_ptr(), idx[1:2].data_ptr(), stream))
    host_idx = (C.c_uint32 * 2)()
    _lib.check(ctx.lib.cuking_copy_to_host(ctx.handle, host_idx, idx.data_ptr(), 8, stream))
    _lib.check(ctx.lib.cuking_stream_synchronize(ctx.handle, stream))
    exp, _, _ = oracle.compute(oracle.submatrix(130), bits, 0.02)
    assert host_idx[0] == len(exp) and host_idx[1] == 0
    assert ctx.lib.cuking_stream_destroy(ctx.handle, stream) == 0


def test_c1_sample_order_symmetry(ctx, c1):
    """Size-independent property at full size: KING is symmetric in the pair, so
    reversing the sample order must give the same (kin, ibs0, ibs1, ibs2) for the
    mirrored pair (n-1-j, n-1-i) -- every pair then meets the kernel with rows
    and columns, tiles and lanes exchanged."""
    n, bits = c1["n"], c1["bits"]
    select(ctx, "tiled", 0)
    rev = ctx.run(cuking_amd.Submatrix(n), bits.shape[1], bits.flip(0).contiguous(), 0.05)
    res = c1["res"]
    assert len(rev) == len(res)
    mirrored = rev.copy()
    mirrored["sample_i"] = n - 1 - rev["sample_j"]
    mirrored["sample_j"] = n - 1 - rev["sample_i"]
    mirrored = cuking_amd.sort_results(np.ascontiguousarray(mirrored))
    assert mirrored.tobytes() == res.tobytes()


def test_rect_needs_prepared_samples(ctx, oracle):
    """cuking_compute_king_rect only reads what cuking_prepare_samples has
    converted for THIS block: unprepared samples, or a workspace that another
    call has converted for a different block since, are FAILED_PRECONDITION,
    not silently wrong sums."""
    import torch
    for variant in (0, 5, 6):
        select(ctx, "tiled", variant)
        tile = ctx.tile_samples()
        n = 5 * tile - 17
        geno = random_genotypes(np.random.default_rng(77), n, 260, missing=0.03)
        geno[n - 1] = geno[0]
        bits = oracle.bitset_from_genotypes(geno)
        d_bits = ctx.upload_bitset(bits)
        wps = bits.shape[1]
        sm = cuking_amd.Submatrix(n)
        res = torch.zeros((200000, 6), dtype=torch.int32, device="cuda:0")
        idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")
        other = ctx.upload_bitset(bits[:tile].copy())
        ctx.run(cuking_amd.Submatrix(tile), wps, other, 0.4)     # workspace: another block
        with pytest.raises(cuking_amd.CukingError, match="prepare"):
            ctx.compute_king_rect(sm, wps, d_bits, (0, tile), (0, tile), 0.0, 200000, res,
                                  idx[0:1], idx[1:2])
        ctx.prepare_samples(sm, wps, d_bits, 0, 2 * tile)
        with pytest.raises(cuking_amd.CukingError, match="column samples"):
            ctx.compute_king_rect(sm, wps, d_bits, (0, tile), (tile, 3 * tile), 0.0, 200000,
                                  res, idx[0:1], idx[1:2])
        with pytest.raises(cuking_amd.CukingError, match="row samples"):
            ctx.compute_king_rect(sm, wps, d_bits, (0, 4 * tile, 3 * tile), (0, 2 * tile), 0.0,
                                  200000, res, idx[0:1], idx[1:2])
        ctx.compute_king_rect(sm, wps, d_bits, (0, 2 * tile), (0, 2 * tile), 0.0, 200000, res,
                              idx[0:1], idx[1:2])                # prepared: fine
        ctx.run(cuking_amd.Submatrix(tile), wps, other, 0.4)     # converts another block
        with pytest.raises(cuking_amd.CukingError, match="prepare"):
            ctx.compute_king_rect(sm, wps, d_bits, (0, tile), (0, tile), 0.0, 200000, res,
                                  idx[0:1], idx[1:2])
        # a whole-block call converts everything: rectangles may follow it
        whole = ctx.run(sm, wps, d_bits, -0.05)
        idx.zero_()
        ctx.compute_king_rect(sm, wps, d_bits, (0, n), (0, n), -0.05, 200000, res,
                              idx[0:1], idx[1:2])
        torch.cuda.synchronize()
        cnt, ovf = idx.tolist()
        got = cuking_amd.sort_results(res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
            cuking_amd.KING_RESULT_DTYPE).copy())
        exp, _, _ = oracle.compute(oracle.submatrix(n), bits, -0.05)
        assert ovf == 0 and got.tobytes() == whole.tobytes() == exp.tobytes()


@pytest.mark.parametrize("variant", [0, 5, 6, 7])
def test_calls_on_two_streams_of_one_context(ctx, oracle, variant):
    """Two blocks back to back on two non-blocking streams of ONE context: the
    second call's layout conversion overwrites the workspace the first call's
    pair kernel reads, so the library has to order them (event wait)."""
    import torch
    select(ctx, "tiled", variant)
    rng = np.random.default_rng(5)
    n1, n2, m = 2300, 900, 6000
    g1 = random_genotypes(rng, n1, m, missing=0.02)
    g2 = random_genotypes(rng, n2, m, missing=0.02)
    g1[n1 - 1], g2[n2 - 1] = g1[3], g2[5]
    b1, b2 = oracle.bitset_from_genotypes(g1), oracle.bitset_from_genotypes(g2)
    d1, d2 = ctx.upload_bitset(b1), ctx.upload_bitset(b2)
    thr = 0.03
    e1, _, _ = oracle.compute(oracle.submatrix(n1), b1, thr, threads=16)
    e2, _, _ = oracle.compute(oracle.submatrix(n2), b2, thr, threads=16)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bufs = [(torch.zeros((1 << 16, 6), dtype=torch.int32, device="cuda:0"),
             torch.zeros(2, dtype=torch.int32, device="cuda:0")) for _ in range(2)]
    torch.cuda.synchronize()
    for rounds in range(3):
        for (res, idx) in bufs:
            idx.zero_()
        torch.cuda.synchronize()
        ctx.compute_king(cuking_amd.Submatrix(n1), b1.shape[1], d1, thr, 1 << 16, bufs[0][0],
                         bufs[0][1][0:1], bufs[0][1][1:2], stream=s1)
        ctx.compute_king(cuking_amd.Submatrix(n2), b2.shape[1], d2, thr, 1 << 16, bufs[1][0],
                         bufs[1][1][0:1], bufs[1][1][1:2], stream=s2)
        torch.cuda.synchronize()
        for (res, idx), exp in zip(bufs, (e1, e2)):
            cnt, ovf = idx.tolist()
            got = cuking_amd.sort_results(res[:cnt].cpu().numpy().view(np.uint32).reshape(
                -1).view(cuking_amd.KING_RESULT_DTYPE).copy())
            assert ovf == 0 and got.tobytes() == exp.tobytes(), rounds


@pytest.mark.parametrize("mv", MFMA_VARIANTS)
@pytest.mark.parametrize("split_wgs", [0, 256])
def test_tile_order_options_do_not_change_results(ctx, oracle, split_wgs, mv):
    """XCD-aware workgroup order and band height only permute which workgroup
    evaluates which tile: records, tile-range unions and rectangles stay the
    oracle's for every setting (launches of >= 64 tiles take the XCD order)."""
    select(ctx, "tiled", mv)
    ctx.set_option("split_wgs", split_wgs)
    rng = np.random.default_rng(99)
    n, m = 2700, 700                       # 22 rows of 128-sample tiles: 253 tiles; 11 of 256: 66
    geno = random_genotypes(rng, n, m, missing=0.03)
    geno[n - 1], geno[1500] = geno[7], geno[130]
    bits = oracle.bitset_from_genotypes(geno)
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, 0.06, threads=16)
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    try:
        for swz in (0, 1, 2):
            for rows in (0, 1, 3, 5, 17, 64):
                ctx.set_option("xcd_swizzle", swz)
                ctx.set_option("band_rows", rows)
                got = ctx.run(sm, bits.shape[1], d_bits, 0.06, max_results=1 << 20)
                assert got.tobytes() == exp.tobytes(), (swz, rows)
                tiles = ctx.num_tiles(sm)
                cut = 2 * tiles // 5
                parts = [ctx.run(sm, bits.shape[1], d_bits, 0.06, max_results=1 << 20,
                                 tile_range=r, sort=False)
                         for r in ((0, cut), (cut, cut + 1), (cut + 1, tiles))]
                merged = cuking_amd.sort_results(np.concatenate(parts))
                assert merged.tobytes() == exp.tobytes(), (swz, rows, "ranges")
        # an off-diagonal block as well (no triangle in the enumeration)
        ctx.set_option("xcd_swizzle", 2)
        ctx.set_option("band_rows", 0)
        off = cuking_amd.Submatrix(n, 2, 1)
        idx = list(range(off.i_begin, off.i_end)) + list(range(off.j_begin, off.j_end))
        sub = np.ascontiguousarray(bits[idx])
        e2, _, _ = oracle.compute(oracle.submatrix(n, 2, 1), sub, 0.06, threads=16)
        assert ctx.run(off, bits.shape[1], ctx.upload_bitset(sub), 0.06,
                       max_results=1 << 20).tobytes() == e2.tobytes()
    finally:
        ctx.set_option("xcd_swizzle", 2)
        ctx.set_option("band_rows", 0)
        ctx.set_option("split_wgs", 256)


@pytest.mark.parametrize("mv", MFMA_VARIANTS)
def test_dynamic_tail_of_a_launch(ctx, oracle, mv):
    """Launches of many rounds hand their last tiles out through a counter
    (king_common.h, dyn_tiles): the same records as the static order, launch after
    launch (the counter returns to zero), in both forms, with the launch cut into
    several by the block limit, and for tile sub-ranges."""
    select(ctx, "tiled", mv)
    rng = np.random.default_rng(77)
    n, m = 6000, 500                       # 47 tile rows: 1128 tiles
    geno = random_genotypes(rng, n, m, missing=0.03)
    geno[n - 1], geno[3000], geno[5900] = geno[7], geno[130], geno[5899]
    bits = oracle.bitset_from_genotypes(geno)
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, 0.07, threads=16)
    assert len(exp) >= 3
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    try:
        ctx.set_option("split_wgs", 4)     # 1128 tiles = 282 rounds of 4: no remainder pieces
        ctx.set_option("dyn_tail_tiles", 1)
        assert ctx.get_option("dyn_tail_tiles") == 1
        for mode in (0, 1):
            ctx.set_option("counts_mode", mode)
            for rep in range(3):
                got = ctx.run(sm, bits.shape[1], d_bits, 0.07, max_results=1 << 20)
                assert got.tobytes() == exp.tobytes(), (mode, rep)
        ctx.set_option("counts_mode", -1)
        tiles = ctx.num_tiles(sm)
        cut = 520 * tiles // 1128              # 128-sample tiles: 520 / 1 / 607 tiles
        parts = [ctx.run(sm, bits.shape[1], d_bits, 0.07, max_results=1 << 20, tile_range=r,
                         sort=False) for r in ((0, cut), (cut, cut + 1), (cut + 1, tiles))]
        merged = cuking_amd.sort_results(np.concatenate(parts))
        assert merged.tobytes() == exp.tobytes()
        # the staged multi-GPU rectangles (side streams, one counter each) as well
        import torch
        from cuking_amd.dist import GpuStagedOps, staged_schedule
        src = torch.from_numpy(bits.view(np.int64))
        parts = []
        for rank in range(2):
            d_recv = torch.zeros((n, bits.shape[1]), dtype=torch.int64, device="cuda:0")
            ops = GpuStagedOps(ctx, sm, bits.shape[1], d_recv, 0.07, 1 << 20)
            ops.begin()
            for (c0, c1), rect in staged_schedule(n, ctx.tile_samples(), 2, rank, 2):
                d_recv[c0:c1].copy_(src[c0:c1])
                if rect is None:
                    continue
                ops.prepare(c0, c1)
                ops.compute_rect(*rect)
            res, count, ovf = ops.finish()
            assert ovf == 0
            parts.append(res[:count].cpu().numpy().view(np.uint32).reshape(-1).view(
                cuking_amd.KING_RESULT_DTYPE).copy())
        merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
        assert merged.tobytes() == exp.tobytes()
        ctx.set_option("max_launch_blocks", 700)   # two launches, each with its own tail
        got = ctx.run(sm, bits.shape[1], d_bits, 0.07, max_results=1 << 20)
        assert got.tobytes() == exp.tobytes()
        ctx.set_option("max_launch_blocks", 0)
        ctx.set_option("dyn_tail_tiles", 0)        # and switched off
        got = ctx.run(sm, bits.shape[1], d_bits, 0.07, max_results=1 << 20)
        assert got.tobytes() == exp.tobytes()
    finally:
        ctx.set_option("max_launch_blocks", 0)
        ctx.set_option("counts_mode", -1)
        ctx.set_option("split_wgs", 256)
        ctx.set_option("dyn_tail_tiles", 16384)


@pytest.mark.parametrize("mv", MFMA_VARIANTS)
@pytest.mark.parametrize("counts_mode", [0, 1])
def test_staged_rectangles_with_few_emitting_lanes(ctx, counts_mode, mv):
    """Staged rectangles (invalid tile slots below the diagonal) x remainder split
    with pieces longer than a tile x a threshold only a handful of pairs pass:
    the combination in which an inlined record append once corrupted the sums of
    whole tiles in the full form (tools/fuzz_split.py seed 1, case 3)."""
    import torch
    from cuking_amd.dist import GpuStagedOps, staged_schedule
    from oracle import pyoracle
    select(ctx, "tiled", mv, counts_mode)
    n, m, thr, seed = 514, 17182, 0.0884, 1003
    cohort = plan_cohort(n, seed)
    kind, pa, pb = cohort_to_device(cohort, 0)
    wps = cuking_amd.words_per_sample(m)
    d_bits = ctx.synth_bitset(seed, kind, pa, pb, 0, n, m)
    torch.cuda.synchronize()
    bits = np.ascontiguousarray(d_bits.cpu().numpy().view(np.uint64))
    exp, _, _ = pyoracle.compute(pyoracle.submatrix(n), bits, thr, threads=16)
    assert 20 < len(exp) < 100
    sm = cuking_amd.Submatrix(n)
    try:
        for wgs, world in ((16, 1), (16, 2), (3, 3), (64, 1), (256, 2), (1, 1)):
            ctx.set_option("split_wgs", wgs)
            parts = []
            for rank in range(world):
                ops = GpuStagedOps(ctx, sm, wps, d_bits, thr, len(exp) + 8, num_streams=1)
                ops.begin()
                for (c0, c1), rect in staged_schedule(n, ctx.tile_samples(), world, rank, 1):
                    if rect is not None:
                        ops.prepare(c0, c1)
                        ops.compute_rect(*rect)
                res, cnt, ovf = ops.finish()
                assert ovf == 0
                parts.append(res[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
                    cuking_amd.KING_RESULT_DTYPE).copy())
            merged = cuking_amd.sort_results(np.ascontiguousarray(np.concatenate(parts)))
            assert merged.tobytes() == exp.tobytes(), (wgs, world)
    finally:
        ctx.set_option("split_wgs", 256)
        ctx.set_option("counts_mode", -1)


@pytest.mark.parametrize("kernel,variant", KERNELS)
def test_closed_form_class_counts_on_the_gpu(ctx, kernel, variant):
    """The 16-class known answers of tests/test_numerics.py straight against the
    device (no oracle in between): the six sums are read off the chosen
    multiplicities, kin from exact rational arithmetic rounded twice."""
    from conftest import kin_exact_two_roundings, pair_from_classes
    from test_numerics import CLASS_CASES, NAMES
    select(ctx, kernel, variant)
    sm = cuking_amd.Submatrix(2)
    for name in sorted(CLASS_CASES):
        geno, want = pair_from_classes(CLASS_CASES[name], seed=len(name))
        bits = cuking_amd.new_host_bitset(sm, geno.shape[1])
        col, row = np.nonzero(geno >= 0)
        cuking_amd.pack_host(sm, bits, row, col, geno[col, row])
        d_bits = ctx.upload_bitset(bits)
        got = ctx.compute_counts(sm, bits.shape[1], d_bits)[0, 1]
        assert {n: int(got[n]) for n in NAMES} == want, name
        if min(want["het_i"], want["het_j"]) > 0:
            res = ctx.run(sm, bits.shape[1], d_bits, -1e30)
            exact = kin_exact_two_roundings(want["het_i"], want["het_j"], want["both_het"],
                                            want["opposing_hom"])
            assert len(res) == 1 and res["kin"].view(np.uint32)[0] == exact.view(np.uint32), name
            assert (int(res["ibs0"][0]), int(res["ibs2"][0])) == (
                want["opposing_hom"], want["concordant_hom"] + want["both_het"])
            assert int(res["ibs1"][0]) == want["shared"] - int(res["ibs0"][0]) - int(res["ibs2"][0])


@pytest.mark.parametrize("kernel,variant", [("tiled", 5), ("tiled", 6), ("tiled", 2), ("stream", 0)])
def test_wide_pair_with_4opp_beyond_2_24_on_the_gpu(ctx, kernel, variant):
    """4.6 M sites, 4 x opposing_hom > 2^24: exact sums (float32 accumulation in the
    matrix-core kernel included) and the documented left-to-right float32 kin."""
    from conftest import pair_from_classes
    from test_numerics import NAMES, reference_expression_f32
    select(ctx, kernel, variant)
    opp = (1 << 22) + 3
    mult = [[5, 7, opp // 2, 3], [11, 400_001, 13, 2], [opp - opp // 2, 17, 19, 1],
            [4, 6, 8, 10]]
    geno, want = pair_from_classes(mult, seed=5)
    sm = cuking_amd.Submatrix(2)
    bits = cuking_amd.new_host_bitset(sm, geno.shape[1])
    col, row = np.nonzero(geno >= 0)
    cuking_amd.pack_host(sm, bits, row, col, geno[col, row])
    d_bits = ctx.upload_bitset(bits)
    got = ctx.compute_counts(sm, bits.shape[1], d_bits)[0, 1]
    assert {n: int(got[n]) for n in NAMES} == want
    for mode in (0, 1):
        ctx.set_option("counts_mode", mode)
        res = ctx.run(sm, bits.shape[1], d_bits, -1e30)
        ref = reference_expression_f32(want["het_i"], want["het_j"], want["both_het"],
                                       want["opposing_hom"])
        assert len(res) == 1 and res["kin"].view(np.uint32)[0] == ref.view(np.uint32)
        assert int(res["ibs0"][0]) == want["opposing_hom"]
    ctx.set_option("counts_mode", -1)


def test_reserved_workspace_means_no_allocation_and_no_host_wait(oracle):
    """cuking_ctx_reserve sizes the kernel layout, the tile prefix and the named
    streams' split slabs and filter scratch up front; the compute / prepare calls for the block on
    those streams then allocate nothing and never wait for the device (what a
    host needs once collectives are in flight: host/multi_gpu.cc)."""
    import torch
    from cuking_amd.dist import staged_schedule, tile_partition
    c = cuking_amd.KingContext(0)
    try:
        rng = np.random.default_rng(77)
        n, m, thr = 700, 1500, 0.02
        geno = random_genotypes(rng, n, m)
        geno[n - 1] = geno[3]
        sm = cuking_amd.Submatrix(n)
        bits = oracle.bitset_from_genotypes(geno)
        exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr)
        d_bits = c.upload_bitset(bits)
        wps = bits.shape[1]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        assert c.get_option("workspace_allocations") == 0
        c.reserve(sm, wps, streams)
        a0, s0 = c.get_option("workspace_allocations"), c.get_option("host_syncs")
        # layout, the sample sort's scratch, prefix table, two split slabs, two filter scratch sets
        assert a0 == 7
        c.reserve(sm, wps, streams)         # idempotent
        assert c.get_option("workspace_allocations") == a0
        results = torch.zeros((len(exp) + 8, 6), dtype=torch.int32, device="cuda:0")
        idx = torch.zeros(2, dtype=torch.int32, device="cuda:0")

        def records():
            torch.cuda.synchronize()
            cnt = int(idx[0])
            assert int(idx[1]) == 0
            r = results[:cnt].cpu().numpy().view(np.uint32).reshape(-1).view(
                cuking_amd.KING_RESULT_DTYPE).copy()
            idx.zero_()
            torch.cuda.synchronize()
            return cuking_amd.sort_results(r)

        torch.cuda.synchronize()
        for s in streams:                   # whole block, then tile ranges, on either stream
            c.compute_king(sm, wps, d_bits, thr, len(exp) + 8, results, idx[0:1], idx[1:2], stream=s)
            assert records().tobytes() == exp.tobytes()
        for k, r in enumerate(tile_partition(c.num_tiles(sm), 3)):
            c.compute_king(sm, wps, d_bits, thr, len(exp) + 8, results, idx[0:1], idx[1:2],
                           stream=streams[k % 2], tile_range=r)
        assert records().tobytes() == exp.tobytes()
        c.invalidate()
        for (c0, c1), rect in staged_schedule(n, c.tile_samples(), 1, 0, 3):   # staged form
            c.prepare_samples(sm, wps, d_bits, c0, c1, stream=streams[0])
            c.compute_king_rect(sm, wps, d_bits, rect[0], rect[1], thr, len(exp) + 8, results,
                                idx[0:1], idx[1:2], stream=streams[0])
        assert records().tobytes() == exp.tobytes()
        # a smaller block fits the same workspace (only its prefix table is new data)
        small = cuking_amd.Submatrix.from_ranges(0, 300, 0, 300)
        exp_small, _, _ = oracle.compute(oracle.Submatrix(0, 300, 0, 300),
                                         np.ascontiguousarray(bits[:300]), thr)
        c.compute_king(small, wps, d_bits, thr, len(exp) + 8, results, idx[0:1], idx[1:2],
                       stream=streams[1])
        assert records().tobytes() == exp_small.tobytes()
        assert c.get_option("workspace_allocations") == a0
        # (the small block's new prefix table is one pageable upload that waits for
        #  its own stream -- the only host-side wait since the reservation)
        assert c.get_option("host_syncs") - s0 <= 1
        with pytest.raises(cuking_amd.CukingError):
            c.reserve(sm, wps, [torch.cuda.Stream() for _ in range(9)])
    finally:
        c.close()


@pytest.mark.parametrize("variant", [5, 6, 7, 0])
def test_reuse_prepared_layout_and_invalidate(oracle, variant):
    """Option "reuse_prepared": a repeated call on the same (block, width, shape,
    bitset pointer) launches the pair kernel only; a host that rewrites the bitset
    in place says so with cuking_invalidate().  Off (the default), every call
    converts, like the reference's kernel reading the bitset as it is."""
    import torch
    c = cuking_amd.KingContext(0)
    try:
        c.set_option("variant", variant)
        rng = np.random.default_rng(5)
        n, m, thr = 400, 2000, 0.1
        geno = random_genotypes(rng, n, m)
        sm = cuking_amd.Submatrix(n)
        bits = oracle.bitset_from_genotypes(geno)
        before, _, _ = oracle.compute(oracle.submatrix(n), bits, thr)
        d_bits = c.upload_bitset(bits)
        wps = bits.shape[1]
        geno2 = geno.copy()
        geno2[7] = geno2[3]                      # sample 7 becomes a duplicate of sample 3
        bits2 = oracle.bitset_from_genotypes(geno2)
        after, _, _ = oracle.compute(oracle.submatrix(n), bits2, thr)
        assert len(after) == len(before) + 1
        c.timing_enable(True)
        # default: every call converts and sees the bitset as it is
        assert c.run(sm, wps, d_bits, thr).tobytes() == before.tobytes()
        d_bits.copy_(torch.from_numpy(bits2.view(np.int64)))
        assert c.run(sm, wps, d_bits, thr).tobytes() == after.tobytes()
        assert c.timing_collect().prepare_launches == 2
        # reuse: the second call launches no conversion
        c.set_option("reuse_prepared", 1)
        assert c.get_option("reuse_prepared") == 1
        c.invalidate()
        c.timing_reset()
        d_bits.copy_(torch.from_numpy(bits.view(np.int64)))
        torch.cuda.synchronize()
        assert c.run(sm, wps, d_bits, thr).tobytes() == before.tobytes()
        skipped = c.get_option("conversions_skipped")
        assert c.run(sm, wps, d_bits, thr).tobytes() == before.tobytes()
        assert c.run(sm, wps, d_bits, thr, tile_range=(0, c.num_tiles(sm))).tobytes() == before.tobytes()
        t = c.timing_collect()
        assert t.prepare_launches == 1 and t.king_launches >= 3
        assert c.get_option("conversions_skipped") == skipped + 2
        # in-place rewrite + invalidate: the new contents are converted and seen
        d_bits.copy_(torch.from_numpy(bits2.view(np.int64)))
        torch.cuda.synchronize()
        c.invalidate()
        c.timing_reset()
        assert c.run(sm, wps, d_bits, thr).tobytes() == after.tobytes()
        assert c.timing_collect().prepare_launches == 1
        # another pointer, width or block is never taken for the prepared one
        other = c.upload_bitset(bits)
        assert c.run(sm, wps, other, thr).tobytes() == before.tobytes()
        half = cuking_amd.Submatrix.from_ranges(0, 200, 0, 200)
        exp_half, _, _ = oracle.compute(oracle.Submatrix(0, 200, 0, 200),
                                        np.ascontiguousarray(bits[:200]), thr)
        assert c.run(half, wps, other, thr).tobytes() == exp_half.tobytes()
    finally:
        c.close()


@pytest.mark.parametrize("missing", [0.0, 0.02, 0.35])
def test_filter_variant_every_path_gives_the_oracle_records(ctx, oracle, missing):
    """The filter variant (king_filter.hip) decides with a bound WHO computes a pair
    exactly: candidate list + one wavefront per pair, or the four-product kernel
    over a dense quadrant.  Whatever the missingness (0.35: the bound lets most
    pairs through, whole quadrants go dense), however short the list and however
    small the launch chunks, with the remainder of the launch cut into pieces of k or not,
    the records are the oracle's."""
    select(ctx, "tiled", 7)
    rng = np.random.default_rng(int(missing * 100) + 3)
    n, m = 1100, 2500                         # 5 x 5 tiles of 256, the last one ragged
    geno = random_genotypes(rng, n, m, missing=missing)
    # a family of 30 near-identical samples inside one quadrant, relatives across tiles
    for k in range(30):
        geno[300 + k] = np.where(rng.random(m) < 0.02, geno[5], geno[300])
    geno[n - 1], geno[700], geno[1023] = geno[2], geno[130], geno[256]
    bits = oracle.bitset_from_genotypes(geno)
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    off = cuking_amd.Submatrix(n, 2, 1)
    idx = list(range(off.i_begin, off.i_end)) + list(range(off.j_begin, off.j_end))
    sub = np.ascontiguousarray(bits[idx])
    d_sub = ctx.upload_bitset(sub)
    defaults = {"filter_quadrant_cap": 384, "filter_cand_cap": 1 << 25, "max_launch_blocks": 0,
                "filter_split_min_steps": 8}
    try:
        for thr in (0.03, 0.0884, 0.2):
            exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr, threads=16)
            e2, _, _ = oracle.compute(oracle.submatrix(n, 2, 1), sub, thr, threads=16)
            assert len(exp) >= 30 * 29 // 2
            for opts in ({}, {"filter_quadrant_cap": 0}, {"filter_quadrant_cap": 3},
                         {"filter_cand_cap": 0}, {"filter_cand_cap": 7}, {"max_launch_blocks": 2},
                         {"filter_cand_cap": 40, "max_launch_blocks": 3},
                         # the launch's 15 tiles as 8 pieces of k each (remainder split)
                         {"filter_split_min_steps": 1},
                         {"filter_split_min_steps": 1, "max_launch_blocks": 7, "filter_quadrant_cap": 2}):
                for k, v in {**defaults, **opts}.items():
                    ctx.set_option(k, v)
                got = ctx.run(sm, bits.shape[1], d_bits, thr, max_results=1 << 20)
                assert got.tobytes() == exp.tobytes(), (thr, opts)
                got = ctx.run(off, bits.shape[1], d_sub, thr, max_results=1 << 20)
                assert got.tobytes() == e2.tobytes(), (thr, opts, "off-diagonal block")
    finally:
        for k, v in defaults.items():
            ctx.set_option(k, v)


def test_filter_variant_tile_geometry_and_fallbacks(ctx, oracle):
    """256-sample tiles whatever runs underneath: the lean form with a threshold in
    (0, 1/2) takes the bound; the full form, thresholds outside that range and the
    diagnostic counts take the four-product kernel on the quadrants of the same
    tiles -- same tile count, same tile bounds, same records."""
    select(ctx, "tiled", 7)
    assert ctx.tile_samples() == 256
    rng = np.random.default_rng(77)
    n, m = 700, 900
    geno = random_genotypes(rng, n, m, missing=0.03)
    geno[n - 1], geno[300] = geno[0], geno[290]
    bits = oracle.bitset_from_genotypes(geno)
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    assert ctx.num_tiles(sm) == 6           # 3 tile rows, upper triangle
    for thr, mode in ((0.05, 0), (0.05, 1), (0.0, -1), (-0.1, 0), (0.5, 0), (0.7, -1),
                      (float("nan"), 0)):
        ctx.set_option("counts_mode", mode)
        exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr)
        got = ctx.run(sm, bits.shape[1], d_bits, thr, max_results=1 << 20)
        assert got.tobytes() == exp.tobytes(), (thr, mode)
        parts = [ctx.run(sm, bits.shape[1], d_bits, thr, max_results=1 << 20, tile_range=r,
                         sort=False) for r in ((0, 1), (1, 4), (4, 6))]
        assert cuking_amd.sort_results(np.concatenate(parts)).tobytes() == exp.tobytes(), (thr, mode)
    ctx.set_option("counts_mode", -1)
    check_counts(ctx, oracle, sm, bits)


@pytest.mark.parametrize("missing", [0.0, 0.02, 0.35])
def test_filter_check_points_inside_the_k_loop(ctx, oracle, missing):
    """The filter kernel's check points (king_filter.hip): after a share of the k-steps a
    tile tests its sums against the bound over the sites so far.  The rigorous check lets a
    tile none of whose pairs can still become a candidate leave (every term of the
    numerator's X is non-negative), and one with a FEW such pairs as well, handing them to
    the candidate list as they stand; the forecast at an eighth of the sites sends a tile whose
    quadrants look dense to the exact kernel at once -- computed by the fallback launch
    (four-product kernel, persistent mode, gated on a device word) behind the filter
    kernel.  Every entry of the share menu forced, the forecast forced, both, neither, with
    and without the remainder pieces: the oracle's records every time; and the exits really
    happen."""
    select(ctx, "tiled", 7, counts_mode=0)
    rng = np.random.default_rng(int(missing * 100) + 11)
    n, m = 1100, 6200                         # 5 x 5 tiles of 256; 25 k-steps of 256 sites
    geno = random_genotypes(rng, n, m, missing=missing)
    for k in range(12):                        # a family inside one quadrant, relatives across tiles
        geno[300 + k] = np.where(rng.random(m) < 0.02, geno[5], geno[300])
    geno[n - 1], geno[700], geno[1023] = geno[2], geno[130], geno[256]
    for k in range(9):                         # 81 related pairs in one quadrant of tile (0, 2)
        geno[10 + k] = np.where(rng.random(m) < 0.02, geno[6], geno[10])
        geno[520 + k] = np.where(rng.random(m) < 0.02, geno[7], geno[10])
    bits = oracle.bitset_from_genotypes(geno)
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    off = cuking_amd.Submatrix(n, 2, 1)
    idx = list(range(off.i_begin, off.i_end)) + list(range(off.j_begin, off.j_end))
    sub = np.ascontiguousarray(bits[idx])
    d_sub = ctx.upload_bitset(sub)
    defaults = {"filter_check0": 1, "filter_check1": 1, "filter_check_emit": 64,
                "filter_rotate": 1, "filter_persistent": 0, "filter_persistent_min_tiles": 2048,
                "filter_split_min_steps": 8,
                "max_launch_blocks": 0, "filter_quadrant_cap": 384, "split_wgs": 256}
    ctx.set_option("filter_check_min_steps", 4)
    ctx.set_option("filter_rotate_min_steps", 4)
    try:
        for thr in (0.03, 0.0884, 0.2):
            exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr, threads=16)
            e2, _, _ = oracle.compute(oracle.submatrix(n, 2, 1), sub, thr, threads=16)
            assert len(exp) >= 12 * 11 // 2
            # (15 tiles on 256 CUs: with remainder splitting every tile goes out as pieces of
            #  k, which have no check points -- "split_wgs": 0 sends whole tiles)
            cases = [{}, {"filter_check0": 0, "filter_check1": 0}, {"filter_check0": 2},
                     {"filter_check0": 2, "split_wgs": 0},
                     {"filter_check0": 2, "filter_check1": 0, "split_wgs": 0},
                     {"filter_check0": 2, "filter_quadrant_cap": 0, "split_wgs": 0},
                     {"filter_check0": 2, "filter_check1": 5, "filter_split_min_steps": 1},
                     {"filter_check0": 2, "max_launch_blocks": 4, "split_wgs": 0}]
            cases += [{"filter_check0": 0, "filter_check1": 2 + k, "split_wgs": 0}
                      for k in range(1, 8)]
            cases += [{"filter_check0": 0, "filter_check_emit": e, "split_wgs": 0}
                      for e in (0, 1, 100)]
            # rotated tiles: every tile starts at a phase boundary of its own (2), all at one
            # (3 + phase), with the forecast, a forced entry, the remainder pieces
            cases += [{"filter_rotate": 2, "split_wgs": 0}, {"filter_rotate": 2},
                      {"filter_rotate": 2, "filter_check0": 2, "split_wgs": 0},
                      {"filter_rotate": 2, "filter_check0": 2, "filter_check1": 0, "split_wgs": 0},
                      {"filter_rotate": 2, "filter_check0": 2, "filter_quadrant_cap": 0, "split_wgs": 0}]
            # ... and the persistent launch (one resident workgroup per CU takes tile after
            # tile): every second case above once more that way
            cases += [{**c, "filter_persistent": 1, "filter_persistent_min_tiles": 0}
                      for c in cases[3::2] if c.get("split_wgs") == 0]
            cases += [{"filter_rotate": 3 + ph, "filter_check0": c0, "filter_check1": 2 + k,
                       "split_wgs": 0, "filter_persistent": ph & 1, "filter_persistent_min_tiles": 0}
                      for ph, c0, k in ((1, 0, 3), (9, 2, 1), (13, 0, 7), (31, 2, 4), (40, 0, 2),
                                        (57, 2, 5), (63, 0, 3), (63, 2, 7))]
            for opts in cases:
                for k, v in {**defaults, **opts}.items():
                    ctx.set_option(k, v)
                exits0 = ctx.get_option("filter_early_exits")
                dense0 = ctx.get_option("filter_dense_quadrants")
                got = ctx.run(sm, bits.shape[1], d_bits, thr, max_results=1 << 20)
                assert got.tobytes() == exp.tobytes(), (thr, opts)
                exits = ctx.get_option("filter_early_exits") - exits0
                dense = ctx.get_option("filter_dense_quadrants") - dense0
                whole = opts.get("split_wgs") == 0
                if missing == 0.0 and thr == 0.2 and opts.get("filter_check1", 1) >= 3 and whole:
                    # unrelated samples sit near kinship 0: from ~0.7 of the sites on no pair of
                    # a tile without relatives can reach 0.2.  Of the 15 tiles the 5 on the
                    # diagonal stay (each sample against itself is alive at any threshold) and
                    # so does (0, 2) with 81 related pairs in one quadrant; (0, 4) and (1, 3)
                    # hold one duplicate each: they hand it to the candidate list at the check
                    # and leave with the 7 tiles that hold nothing.
                    # (handing over switched off: those two stay; with a cap of 100, (0, 2) goes)
                    emit = opts.get("filter_check_emit", 64)
                    assert exits == (7 if emit == 0 else 10 if emit == 100 else 9), (thr, opts, exits)
                if missing == 0.35 and opts.get("filter_check0") == 2 and whole and \
                        "max_launch_blocks" not in opts:
                    # the bound thins nothing out: the tiles leave at the forecast (counted as 4
                    # quadrants each; a tile of the ragged last column may stay under "three of
                    # four quadrants dense") and the fallback launch computes all of them
                    assert dense >= 4 * 10, (thr, opts, dense)
                got = ctx.run(off, bits.shape[1], d_sub, thr, max_results=1 << 20)
                assert got.tobytes() == e2.tobytes(), (thr, opts, "off-diagonal block")
    finally:
        ctx.set_option("filter_check_min_steps", 64)
        ctx.set_option("filter_rotate_min_steps", 128)
        for k, v in defaults.items():
            ctx.set_option(k, v)
        ctx.set_option("counts_mode", -1)


def test_filter_rotated_tiles_join_the_position_of_their_xcd(ctx, oracle):
    """Rotated tiles (king_filter.hip): a tile starts its k loop at the phase boundary the
    tiles of its XCD have reached, runs to the end of the sites, wraps around and ends where
    it started -- so that the tiles an XCD holds at a time read the same k-steps at the
    same time and share them through its L2.  300 tiles on 256 CUs at a threshold where
    the tiles leave at the check point: the tiles of the second round start where the first
    left (another phase than the first), the oracle's records either way, whole block and
    off-diagonal block."""
    select(ctx, "tiled", 7, counts_mode=0)
    rng = np.random.default_rng(97)
    n, m = 6000, 6200                          # 24 x 24 tiles of 256; 25 k-steps of 256 sites
    geno = random_genotypes(rng, n, m, missing=0.01)
    geno[n - 1], geno[3000], geno[5900], geno[300] = geno[7], geno[130], geno[5899], geno[299]
    bits = oracle.bitset_from_genotypes(geno)
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    off = cuking_amd.Submatrix(n, 2, 1)
    idx = list(range(off.i_begin, off.i_end)) + list(range(off.j_begin, off.j_end))
    sub = np.ascontiguousarray(bits[idx])
    d_sub = ctx.upload_bitset(sub)
    thr = 0.2
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr, threads=16)
    e2, _, _ = oracle.compute(oracle.submatrix(n, 2, 1), sub, thr, threads=16)
    assert len(exp) >= 4
    ctx.set_option("filter_check_min_steps", 4)
    ctx.set_option("filter_rotate_min_steps", 4)
    ctx.set_option("filter_rotate_min_tiles", 0)
    ctx.set_option("split_wgs", 0)             # whole tiles only (remainder pieces never rotate)
    try:
        for rotate, persistent in ((1, 0), (0, 0), (1, 1), (0, 1), (1, 1)):
            ctx.set_option("filter_rotate", rotate)
            ctx.set_option("filter_persistent", persistent)
            ctx.set_option("filter_persistent_min_tiles", 0)
            r0 = ctx.get_option("filter_rotated_tiles")
            x0 = ctx.get_option("filter_early_exits")
            got = ctx.run(sm, bits.shape[1], d_bits, thr, max_results=1 << 20)
            assert got.tobytes() == exp.tobytes(), rotate
            rotated = ctx.get_option("filter_rotated_tiles") - r0
            exits = ctx.get_option("filter_early_exits") - x0
            # 300 tiles, 24 on the diagonal: the others leave at the check
            assert exits >= 250, (rotate, exits)
            if rotate:
                assert 1 <= rotated <= 300 - 256 + 8, (rotate, rotated)
            else:
                assert rotated == 0
            got = ctx.run(off, bits.shape[1], d_sub, thr, max_results=1 << 20)
            assert got.tobytes() == e2.tobytes(), (rotate, "off-diagonal block")
    finally:
        for k, v in (("filter_check_min_steps", 64), ("filter_rotate_min_steps", 128),
                     ("filter_rotate_min_tiles", 2048), ("filter_rotate", 1), ("split_wgs", 256),
                     ("filter_persistent", 0), ("filter_persistent_min_tiles", 2048),
                     ("counts_mode", -1)):
            ctx.set_option(k, v)


def test_filter_sorted_layout_and_lazy_codes(ctx, oracle):
    """The filter variant's kernel layout holds the samples of a prepared range sorted by
    their share of missing calls (king_sort.hip), and the four-product kernel's codes are
    converted only when a launch hands something to that kernel (lazy codes).  A cohort
    with a few low-call-rate samples at random positions: the oracle's records and counts
    with the sort on and off and the codes lazy and eager -- whole block, off-diagonal
    block, tile ranges, staged rectangles, full form, diagnostic counts, thresholds
    without a bound -- and with the sort the bad samples spoil far fewer quadrants."""
    from fuzz_cases import _staged
    select(ctx, "tiled", 7)
    rng = np.random.default_rng(41)
    n, m = 1100, 6200
    geno = random_genotypes(rng, n, m, missing=0.01)
    bad = rng.choice(n, size=66, replace=False)
    geno[bad] = np.where(rng.random((66, m)) < 0.4, -1, geno[bad])
    geno[n - 1], geno[700], geno[1023], geno[bad[0]] = geno[2], geno[130], geno[256], geno[bad[1]]
    bits = oracle.bitset_from_genotypes(geno)
    d_bits = ctx.upload_bitset(bits)
    wps = bits.shape[1]
    sm = cuking_amd.Submatrix(n)
    off = cuking_amd.Submatrix(n, 2, 1)
    idx = list(range(off.i_begin, off.i_end)) + list(range(off.j_begin, off.j_end))
    sub = np.ascontiguousarray(bits[idx])
    d_sub = ctx.upload_bitset(sub)
    dense = {}
    try:
        for sort, lazy in ((1, 1), (0, 1), (2, 0), (0, 0)):
            ctx.set_option("filter_sort", sort)
            ctx.set_option("filter_lazy_codes", lazy)
            for reuse in (0, 1):
                ctx.set_option("reuse_prepared", reuse)
                ctx.invalidate()
                for thr, mode in ((0.0884, 0), (0.05, -1), (0.0884, 1), (0.0, -1), (0.7, 0),
                                  (0.03, 0), (0.0884, 0)):
                    ctx.set_option("counts_mode", mode)
                    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr, threads=16)
                    d0 = ctx.get_option("filter_dense_quadrants")
                    got = ctx.run(sm, wps, d_bits, thr, max_results=1 << 20)
                    assert got.tobytes() == exp.tobytes(), (sort, lazy, reuse, thr, mode)
                    if (thr, mode, reuse) == (0.0884, 0, 0):
                        dense[min(sort, 1)] = ctx.get_option("filter_dense_quadrants") - d0
                    e2, _, _ = oracle.compute(oracle.submatrix(n, 2, 1), sub, thr, threads=16)
                    got = ctx.run(off, wps, d_sub, thr, max_results=1 << 20)
                    assert got.tobytes() == e2.tobytes(), (sort, lazy, reuse, thr, mode, "off-diagonal")
                    parts = [ctx.run(sm, wps, d_bits, thr, max_results=1 << 20, tile_range=r,
                                     sort=False) for r in ((0, 4), (4, 9), (9, 15))]
                    assert cuking_amd.sort_results(np.concatenate(parts)).tobytes() == exp.tobytes(), \
                        (sort, lazy, reuse, thr, mode, "tile ranges")
            ctx.set_option("reuse_prepared", 0)
            ctx.set_option("counts_mode", 0)
            exp, _, _ = oracle.compute(oracle.submatrix(n), bits, 0.0884, threads=16)
            ctx.invalidate()
            got = _staged(ctx, sm, wps, d_bits, 0.0884, len(exp) + 8, n, 3, 2, [1, 2, 3])
            assert got.tobytes() == exp.tobytes(), (sort, lazy, "staged")
            ctx.set_option("counts_mode", -1)
            check_counts(ctx, oracle, sm, bits)
            check_counts(ctx, oracle, off, sub)
        # in stored order ~8 of every 128 samples are bad: every quadrant of the 45 with a
        # pair i < j goes dense; sorted, the 66 bad samples fill the last quadrant row and
        # column
        assert dense[0] >= 40 and dense[1] <= 20, dense
    finally:
        for k, v in (("filter_sort", 1), ("filter_lazy_codes", 1), ("reuse_prepared", 0),
                     ("counts_mode", -1)):
            ctx.set_option(k, v)


@pytest.mark.parametrize("missing,thr", [(0.35, 0.05), (0.02, 0.004), (0.03, 0.07), (0.03, 0.085)])
def test_filter_variant_gives_up_on_a_cohort_its_bound_cannot_thin_out(ctx, oracle, missing, thr):
    """Heavy missingness, or a threshold inside the noise of unrelated pairs: nearly every
    quadrant goes to the four-product kernel, and once most finished quadrants of a launch
    have, the remaining tiles hand theirs over without computing the product at all
    (1,200 quadrants on 256 CUs: the second round sees the first).  Same records -- also in
    the mixed regime (500 sites: kinship noise of unrelated pairs ~0.03, thresholds 0.07 /
    0.085), where some quadrants list a few hundred candidates, others go dense, and the
    counters behind the decision move while workgroups read them (the decision is one per
    workgroup: a wavefront that left alone would take its share of the stage requests
    with it -- caught by test_dynamic_tail_of_a_launch in round 3)."""
    select(ctx, "tiled", 7, counts_mode=0)
    rng = np.random.default_rng(5)
    n, m = 6000, 500
    geno = random_genotypes(rng, n, m, missing=missing)
    geno[n - 1], geno[3000], geno[5900] = geno[7], geno[130], geno[5899]
    bits = oracle.bitset_from_genotypes(geno)
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, thr, threads=16)
    d_bits = ctx.upload_bitset(bits)
    sm = cuking_amd.Submatrix(n)
    before = ctx.get_option("filter_dense_quadrants")
    try:
        for rep in range(6):
            ctx.set_option("dyn_tail_tiles", 1 if rep % 2 else 16384)
            got = ctx.run(sm, bits.shape[1], d_bits, thr, max_results=8 << 20)
            assert got.tobytes() == exp.tobytes(), rep
    finally:
        ctx.set_option("dyn_tail_tiles", 16384)
        ctx.set_option("counts_mode", -1)
    dense = ctx.get_option("filter_dense_quadrants") - before
    if thr < 0.06:
        # (300 tiles = 1,176 quadrants with a pair i < j: most of them went dense, every time)
        assert dense > 6 * 600
    else:
        assert dense > 0
