"""CPU tests of the product's host side: the C ABI library loads, exports every
symbol include/cuking_amd.h declares, and its host-only helpers agree with the
oracle.  No compute calls (there is no GPU here) -- and no CPU fallback: the
device entry points must fail loudly without a gfx950 device."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import cuking_amd
from cuking_amd import _lib
from conftest import ROOT, random_genotypes


def has_gpu():
    return cuking_amd.device_count() > 0


def declared_symbols():
    text = (ROOT / "include" / "cuking_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cuking_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header, not exported"
    # and the binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == names
    assert lib.cuking_abi_version() == 2


def test_header_is_plain_c(tmp_path):
    """The boundary must be consumable from C (no C++ / torch types)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "cuking_amd.h"\n'
                   "int main(void) { cuking_result r; cuking_submatrix s;"
                   " (void)r; (void)s; return sizeof(cuking_result) == 24 ? 0 : 1; }\n")
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic",
                    f"-I{ROOT / 'include'}", str(src), "-o", str(exe)],
                   check=True)
    assert subprocess.run([str(exe)]).returncode == 0


@pytest.mark.parametrize("n", [1, 2, 5, 10, 37, 100, 734000])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 7])
def test_submatrix_matches_oracle(oracle, n, k):
    olib = oracle.load()
    for shard in range(k * (k + 1) // 2):
        sm = cuking_amd.Submatrix(n, k, shard)
        osm = oracle.submatrix(n, k, shard)
        assert sm.as_tuple() == osm.as_tuple()
        assert sm.NumRows() == olib.orc_num_rows(C.byref(osm))
        assert sm.NumCols() == olib.orc_num_cols(C.byref(osm))
        assert sm.NumSamples() == olib.orc_num_samples(C.byref(osm))
        probe = {0, n - 1, n // 2, sm.i_begin, max(sm.i_end, 1) - 1, sm.j_begin,
                 max(sm.j_end, 1) - 1}
        for s in probe:
            if 0 <= s < n:
                assert sm.Contains(s) == bool(olib.orc_contains(C.byref(osm), s))
                if sm.Contains(s):
                    assert sm.SampleOffset(s) == olib.orc_sample_offset(C.byref(osm), s)
        # pairs the block holds
        if n <= 100:
            exp = sum(1 for i in range(sm.i_begin, sm.i_end)
                      for j in range(max(sm.j_begin, i + 1), sm.j_end))
            assert sm.NumPairs() == exp
    with pytest.raises(cuking_amd.CukingError) as e:
        cuking_amd.Submatrix(n, k, k * (k + 1) // 2)
    assert e.value.status == _lib.ERR_INVALID_ARGUMENT
    assert "Invalid shard index" in e.value.message       # cuking.cu:461
    with pytest.raises(cuking_amd.CukingError) as e:
        cuking_amd.Submatrix(n, 0, 0)
    assert "Invalid split factor" in e.value.message      # cuking.cu:456


def test_sizing_matches_oracle_and_survey(oracle):
    olib = oracle.load()
    for m in list(range(1, 200)) + [10000, 100000, 150000, 200000, 4194304]:
        assert cuking_amd.padded_sites(m) == olib.orc_padded_sites(m)
        assert cuking_amd.words_per_sample(m) == olib.orc_words_per_sample(m)
    # algorithmic bytes per pair
    for m, b in ((10000, 5024), (100000, 50016), (150000, 75008), (200000, 100000)):
        assert cuking_amd.bytes_per_pair(cuking_amd.words_per_sample(m)) == b


@pytest.mark.parametrize("k,shard", [(1, 0), (2, 1), (3, 2), (3, 4)])
def test_pack_host_matches_oracle(oracle, k, shard):
    rng = np.random.default_rng(5)
    n, m = 29, 150
    geno = random_genotypes(rng, n, m, missing=0.2)
    col, row = np.nonzero(geno >= 0)
    alt = geno[col, row].astype(np.int32)
    perm = rng.permutation(len(row))
    sm = cuking_amd.Submatrix(n, k, shard)
    bits = cuking_amd.new_host_bitset(sm, m)
    cuking_amd.pack_host(sm, bits, row[perm], col[perm], alt[perm])
    osm = oracle.submatrix(n, k, shard)
    obits = oracle.new_bitset(osm, m)
    oracle.pack(osm, obits, row, col, alt)
    assert bits.shape == obits.shape and np.array_equal(bits, obits)


def test_pack_host_concurrent_threads(oracle):
    """cuking.cu:550-553: files are packed concurrently into one bitset."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(6)
    n, m = 40, 3000
    geno = random_genotypes(rng, n, m, missing=0.1)
    col, row = np.nonzero(geno >= 0)
    alt = geno[col, row].astype(np.int32)
    perm = rng.permutation(len(row))
    row, col, alt = row[perm], col[perm], alt[perm]
    sm = cuking_amd.Submatrix(n)
    bits = cuking_amd.new_host_bitset(sm, m)
    chunks = np.array_split(np.arange(len(row)), 16)
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda c: cuking_amd.pack_host(sm, bits, row[c], col[c], alt[c]),
                    chunks))
    assert np.array_equal(bits, oracle.bitset_from_genotypes(geno))


@pytest.mark.parametrize("k,shard", [(1, 0), (2, 1), (3, 4)])
def test_pack_host_orders_of_the_input(oracle, k, shard):
    """cuking_pack_host collects the bits of one 64-site word column per sample and clears
    each word once when the table is site-major (the Spark writer's order), falls back to a
    bit at a time when it is not: every order of the same triples gives the oracle's bitset
    -- site-major, sample-major, shuffled, site-major with a shuffled stretch in the
    middle -- and an error in the middle of a column leaves the thread's masks clean."""
    rng = np.random.default_rng(11)
    n, m = 61, 900                       # 15 word columns, the last one partial
    geno = random_genotypes(rng, n, m, missing=0.1)
    site, sample = np.nonzero(geno.T >= 0)            # site-major
    alt = geno[sample, site].astype(np.int32)
    assert len(site) > 40000
    sm = cuking_amd.Submatrix(n, k, shard)
    osm = oracle.submatrix(n, k, shard)
    exp = oracle.new_bitset(osm, m)
    oracle.pack(osm, exp, site, sample, alt)
    orders = {"site-major": np.arange(len(site)),
              "sample-major": np.lexsort((site, sample)),
              "shuffled": rng.permutation(len(site))}
    mixed = np.arange(len(site))
    mixed[20000:30000] = rng.permutation(mixed[20000:30000])
    orders["site-major with a shuffled stretch"] = mixed
    for name, o in orders.items():
        bits = cuking_amd.new_host_bitset(sm, m)
        cuking_amd.pack_host(sm, bits, site[o], sample[o], alt[o])
        assert np.array_equal(bits, exp), name
    # in two calls that cut a word column in two
    bits = cuking_amd.new_host_bitset(sm, m)
    cuking_amd.pack_host(sm, bits, site[:12345], sample[:12345], alt[:12345])
    cuking_amd.pack_host(sm, bits, site[12345:], sample[12345:], alt[12345:])
    assert np.array_equal(bits, exp)
    # an invalid genotype / site in the middle: reported, and what the thread had collected
    # so far does not leak into the next call
    for bad_alt, bad_site, status in ((3, None, _lib.ERR_FAILED_PRECONDITION),
                                      (None, 64 * bits.shape[1] // 2, _lib.ERR_INVALID_ARGUMENT)):
        a2, s2 = alt.copy(), site.copy()
        at = int(np.flatnonzero([osm.i_begin <= c < osm.i_end or osm.j_begin <= c < osm.j_end
                                 for c in sample[5000:5200]])[0]) + 5000
        if bad_alt is not None:
            a2[at] = bad_alt
        else:
            s2[at] = bad_site
        scratch = cuking_amd.new_host_bitset(sm, m)
        with pytest.raises(cuking_amd.CukingError) as e:
            cuking_amd.pack_host(sm, scratch, s2, sample, a2)
        assert e.value.status == status
        bits = cuking_amd.new_host_bitset(sm, m)
        cuking_amd.pack_host(sm, bits, site, sample, alt)
        assert np.array_equal(bits, exp)


def test_pack_host_errors():
    sm = cuking_amd.Submatrix(3)
    bits = cuking_amd.new_host_bitset(sm, 64)
    with pytest.raises(cuking_amd.CukingError) as e:
        cuking_amd.pack_host(sm, bits, [0], [0], [3])
    assert e.value.status == _lib.ERR_FAILED_PRECONDITION
    assert "Invalid value for n_alt_alleles (3)" in e.value.message  # :699-701
    with pytest.raises(cuking_amd.CukingError) as e:
        cuking_amd.pack_host(sm, bits, [64], [0], [0])
    assert e.value.status == _lib.ERR_INVALID_ARGUMENT
    cuking_amd.pack_host(sm, bits, [0], [99], [0])  # foreign sample: skipped


def test_sort_results_matches_reference_order():
    rng = np.random.default_rng(0)
    recs = np.zeros(500, dtype=cuking_amd.KING_RESULT_DTYPE)
    recs["sample_i"] = rng.integers(0, 6, 500)
    recs["sample_j"] = rng.integers(0, 6, 500)
    recs["kin"] = rng.random(500).astype(np.float32)
    exp = recs[np.lexsort((recs["kin"], recs["sample_j"], recs["sample_i"]))]
    assert cuking_amd.sort_results(recs.copy()).tobytes() == exp.tobytes()


@pytest.mark.parametrize("n,k,shard", [(1, 1, 0), (2, 1, 0), (63, 1, 0), (64, 1, 0),
                                       (65, 1, 0), (200, 1, 0), (1000, 1, 0),
                                       (300, 2, 1), (300, 3, 4), (5, 4, 9),
                                       (2500, 1, 0)])
def test_tiles_cover_each_pair_exactly_once(n, k, shard):
    """Host logic of the pair-space tiling (any contiguous split of the tile
    enumeration is a valid multi-GPU sharding)."""
    lib = _lib.load()
    sm = cuking_amd.Submatrix(n, k, shard)
    tiles = lib.cuking_num_tiles(None, C.byref(sm.c))
    edge = lib.cuking_tile_samples(None)
    if sm.NumRows() == 0 or sm.NumCols() == 0:
        assert tiles == 0
        return
    seen = {}
    rb, re_, cb, ce = (C.c_uint32() for _ in range(4))
    cover = np.zeros((sm.NumRows(), sm.NumCols()), dtype=np.int32)
    for t in range(tiles):
        _lib.check(lib.cuking_tile_bounds(None, C.byref(sm.c), t, C.byref(rb),
                                          C.byref(re_), C.byref(cb), C.byref(ce)))
        key = (rb.value, cb.value)
        assert key not in seen
        seen[key] = t
        assert 0 < re_.value - rb.value <= edge and 0 < ce.value - cb.value <= edge
        cover[rb.value - sm.i_begin:re_.value - sm.i_begin,
              cb.value - sm.j_begin:ce.value - sm.j_begin] += 1
    I, J = np.meshgrid(np.arange(sm.i_begin, sm.i_end),
                       np.arange(sm.j_begin, sm.j_end), indexing="ij")
    assert np.all(cover[I < J] == 1)       # every pair in exactly one tile
    assert cover.max() <= 1
    with pytest.raises(cuking_amd.CukingError):
        _lib.check(lib.cuking_tile_bounds(None, C.byref(sm.c), tiles, None, None,
                                          None, None))


def test_tile_partition_is_balanced():
    from cuking_amd.dist import tile_partition
    for tiles in (0, 1, 7, 8, 12403, 65_800_000):
        for world in (1, 2, 3, 8):
            parts = tile_partition(tiles, world)
            assert parts[0][0] == 0 and parts[-1][1] == tiles
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.skipif(has_gpu(), reason="checks behaviour without a GPU")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(cuking_amd.CukingError) as e:
        cuking_amd.KingContext(0)
    assert e.value.status == _lib.ERR_DEVICE
    assert "no CPU path" in e.value.message


def test_narrow_triples_matches_pack_host(oracle):
    """cuking_narrow_triples (host half of the compact device pack): the kept
    entries, replayed bit by bit, give the bitset cuking_pack_host builds; same
    filter (cuking.cu:677-679) and the same errors (:698-702)."""
    import ctypes as C
    from cuking_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n, m = 37, 210
    geno = random_genotypes(rng, n, m, missing=0.1)
    col, row = np.nonzero(geno >= 0)
    alt = geno[col, row].astype(np.int32)
    perm = rng.permutation(len(row))
    row, col, alt = (np.ascontiguousarray(a[perm]) for a in (row.astype(np.int64),
                                                             col.astype(np.int64), alt))
    for k, shard in ((1, 0), (3, 1), (3, 3)):
        sm = cuking_amd.Submatrix(n, k, shard)
        want = cuking_amd.new_host_bitset(sm, m)
        cuking_amd.pack_host(sm, want, row, col, alt)
        site = np.zeros(len(row), dtype=np.uint32)
        sa = np.zeros(len(row), dtype=np.uint32)
        kept = C.c_size_t(0)
        _lib.check(lib.cuking_narrow_triples(C.byref(sm.c), want.shape[1], row.ctypes.data,
                                             col.ctypes.data, alt.ctypes.data, len(row),
                                             site.ctypes.data, sa.ctypes.data, C.byref(kept)))
        inside = np.array([sm.Contains(int(c)) for c in col])
        assert kept.value == int(inside.sum())
        got = cuking_amd.new_host_bitset(sm, m)
        plane = want.shape[1] // 2
        for s_, x in zip(site[:kept.value].tolist(), sa[:kept.value].tolist()):
            sample, g = x & 0x3FFFFFFF, x >> 30
            bit = ~np.uint64(1 << (s_ & 63))
            if g in (0, 2):
                got[sample, s_ >> 6] &= bit
            if g in (0, 1):
                got[sample, plane + (s_ >> 6)] &= bit
        assert np.array_equal(got, want)
    sm = cuking_amd.Submatrix(n)
    wps = cuking_amd.words_per_sample(m)
    kept = C.c_size_t(0)
    buf = np.zeros(4, dtype=np.uint32)
    one = lambda r, c, a: (np.array([r], np.int64), np.array([c], np.int64), np.array([a], np.int32))
    r_, c_, a_ = one(0, 1, 3)
    assert lib.cuking_narrow_triples(C.byref(sm.c), wps, r_.ctypes.data, c_.ctypes.data,
                                     a_.ctypes.data, 1, buf.ctypes.data, buf[2:].ctypes.data,
                                     C.byref(kept)) == _lib.ERR_FAILED_PRECONDITION
    assert b"Invalid value for n_alt_alleles (3)" in lib.cuking_last_error()
    r_, c_, a_ = one(10 ** 6, 1, 1)
    assert lib.cuking_narrow_triples(C.byref(sm.c), wps, r_.ctypes.data, c_.ctypes.data,
                                     a_.ctypes.data, 1, buf.ctypes.data, buf[2:].ctypes.data,
                                     C.byref(kept)) == _lib.ERR_INVALID_ARGUMENT
    r_, c_, a_ = one(5, n + 4, 7)          # outside the block: skipped before validation
    assert lib.cuking_narrow_triples(C.byref(sm.c), wps, r_.ctypes.data, c_.ctypes.data,
                                     a_.ctypes.data, 1, buf.ctypes.data, buf[2:].ctypes.data,
                                     C.byref(kept)) == 0 and kept.value == 0


def test_matrix_core_loops_hold_no_foreign_waits():
    """The LDS-DMA requests of king_mfma.hip are inline asm the compiler's wait-count
    pass cannot see: a spill reload (or any load of its own) still in flight at the
    top of a k-loop makes it insert `s_waitcnt vmcnt(0)` there, which drains the whole
    prefetch pipeline every k-step (-10 % once).  Every library build checks the
    assembly it leaves in build_tmp/; so does this test, on the library under test."""
    from cuking_amd import build
    asm = build.PKG / "build_tmp" / "king_mfma-hip-amdgcn-amd-amdhsa-gfx950.s"
    if not asm.exists() or asm.stat().st_mtime < (build.CSRC / "king_mfma.hip").stat().st_mtime:
        build.build_library(force=True)
    assert build.check_mfma_loops(asm) == []
    # and the checker itself notices what it is there for
    text = asm.read_text()
    bad = text.replace("s_waitcnt vmcnt(24)", "s_waitcnt vmcnt(0)\n\ts_waitcnt vmcnt(24)")
    assert bad != text
    broken = asm.with_suffix(".broken.s")
    broken.write_text(bad)
    try:
        assert build.check_mfma_loops(broken) != []
    finally:
        broken.unlink()


def test_filter_kernel_loop_holds_no_foreign_waits_and_no_scratch():
    """The same for king_filter_kernel (the default pair kernel): its k-loop holds the
    hand-counted `vmcnt(22)` hand-overs only, and no scratch access."""
    from cuking_amd import build
    asm = build.PKG / "build_tmp" / "king_filter-hip-amdgcn-amd-amdhsa-gfx950.s"
    if not asm.exists() or asm.stat().st_mtime < (build.CSRC / "king_filter.hip").stat().st_mtime:
        build.build_library(force=True)
    assert build.check_filter_loop(asm) == []
    text = asm.read_text()
    bad = text.replace("s_waitcnt vmcnt(22)", "s_waitcnt vmcnt(0)\n\ts_waitcnt vmcnt(22)")
    assert bad != text
    broken = asm.with_suffix(".broken.s")
    broken.write_text(bad)
    try:
        assert build.check_filter_loop(broken) != []
    finally:
        broken.unlink()


def test_default_variant_and_its_tile_geometry_without_a_context():
    """Host-side facts a multi-GPU driver relies on before it has a device: eight
    compiled shapes, the default one is the filter variant with 256-sample tiles, and
    the tile enumeration of a block follows from that alone."""
    lib = _lib.load()
    assert lib.cuking_num_variants() == 8
    names = [lib.cuking_variant_name(v).decode() for v in range(8)]
    assert names[7] == "t256_mfma_fp4_filter" and names[6] == "t128_mfma_fp4_n4"
    assert names[5] == "t128_mfma_fp4" and lib.cuking_variant_name(8) == b""
    if "CUKING_AMD_VARIANT" not in __import__("os").environ:
        assert lib.cuking_tile_samples(None) == 256
        sm = cuking_amd.Submatrix(100_000)
        assert lib.cuking_num_tiles(None, C.byref(sm.c)) == 391 * 392 // 2
