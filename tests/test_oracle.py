"""CPU tests of the oracle itself (no GPU, no product code).

The reference ships no tests or vectors, so the oracle's C
restatement (oracle/king_oracle.c, bit planes + popcounts like cuking.cu) is
pinned against (a) the hand-checked KAT and (b) a naive per-genotype numpy
oracle that shares no code with it.  PARITY UNPINNED by the reference itself.
"""
import ctypes as C
import json

import numpy as np
import pytest

from conftest import GOLDEN, random_genotypes


def load_kat():
    return json.loads((GOLDEN / "kat_4x10.json").read_text())


def test_kat_counts_and_kin_bits(oracle, naive):
    kat = load_kat()
    geno = np.array(kat["genotypes"], dtype=np.int8)
    sm = oracle.submatrix(geno.shape[0])
    bits = oracle.bitset_from_genotypes(geno, sm)
    oi, oj, counts, kin = oracle.all_pairs(sm, bits)
    got = {(int(i), int(j)): (c, k) for i, j, c, k in zip(oi, oj, counts, kin)}
    for p in kat["pairs"]:
        c, k = got[(p["i"], p["j"])]
        for name in ("het_i", "het_j", "both_het", "opposing_hom",
                     "concordant_hom", "shared"):
            assert int(c[name]) == p[name], (p, name)
        assert hex(np.float32(k).view(np.uint32)) == p["kin_bits"]
        # the naive loop agrees with the hand-written numbers too
        nc = naive.pair_counts_loop(geno[p["i"]], geno[p["j"]])
        assert nc == tuple(p[n] for n in ("het_i", "het_j", "both_het",
                                          "opposing_hom", "concordant_hom",
                                          "shared"))


def test_kat_thresholded_records(oracle):
    kat = load_kat()
    geno = np.array(kat["genotypes"], dtype=np.int8)
    sm = oracle.submatrix(geno.shape[0])
    bits = oracle.bitset_from_genotypes(geno, sm)
    res, ovf, n = oracle.compute(sm, bits, kat["thresholded"]["kin_threshold"])
    assert ovf == 0 and n == 1
    names = kat["samples"]
    got = [[names[r["sample_i"]], names[r["sample_j"]], float(r["kin"]),
            int(r["ibs0"]), int(r["ibs1"]), int(r["ibs2"])] for r in res]
    assert got == kat["thresholded"]["records"]


@pytest.mark.parametrize("n,m,seed", [
    (2, 1, 0), (3, 31, 1), (5, 32, 2), (7, 33, 3), (9, 63, 4), (11, 64, 5),
    (13, 65, 6), (17, 127, 7), (20, 300, 8), (33, 129, 9), (70, 257, 10)])
def test_c_oracle_equals_naive_loop_and_matmul(oracle, naive, n, m, seed):
    rng = np.random.default_rng(seed)
    geno = random_genotypes(rng, n, m, missing=0.1)
    sm = oracle.submatrix(n)
    bits = oracle.bitset_from_genotypes(geno, sm)
    oi, oj, counts, kin = oracle.all_pairs(sm, bits)
    li, lj, lc = naive.all_pairs_loop(geno)
    mi, mj, mc = naive.all_pairs_matmul(geno)
    assert np.array_equal(oi, li) and np.array_equal(oj, lj)
    assert np.array_equal(oi, mi) and np.array_equal(oj, mj)
    assert np.array_equal(lc, mc)
    for k, name in enumerate(counts.dtype.names):
        assert np.array_equal(counts[name].astype(np.int64), lc[:, k]), name
    nk = naive.kin_f32(lc[:, 0], lc[:, 1], lc[:, 2], lc[:, 3])
    # bit-exact float32, NaN patterns included
    assert np.array_equal(kin.view(np.uint32), nk.view(np.uint32))


def test_thresholded_equals_naive(oracle, naive):
    rng = np.random.default_rng(42)
    geno = random_genotypes(rng, 60, 500, missing=0.02)
    geno[7] = geno[3]            # duplicate
    geno[9, :250] = geno[4, :250]  # half identical
    sm = oracle.submatrix(60)
    bits = oracle.bitset_from_genotypes(geno, sm)
    for thr in (-10.0, -0.25, 0.0, 0.0884, 0.3):
        res, ovf, n = oracle.compute(sm, bits, thr)
        exp = naive.king(geno, thr)
        assert ovf == 0 and n == len(exp)
        assert res.tobytes() == exp.tobytes()
    assert any((r["sample_i"], r["sample_j"]) == (3, 7) for r in
               oracle.compute(sm, bits, 0.4)[0])


def test_edge_cases(oracle, naive):
    m = 70
    geno = np.zeros((6, m), dtype=np.int8)
    geno[0] = -1                          # everything missing
    geno[1] = 0                           # no hets at all
    geno[2] = 1                           # all het
    geno[3] = 1                           # duplicate of 2
    geno[4] = 2
    geno[5, ::2] = 1
    sm = oracle.submatrix(6)
    bits = oracle.bitset_from_genotypes(geno, sm)
    oi, oj, counts, kin = oracle.all_pairs(sm, bits)
    pairs = {(int(i), int(j)): (c, k) for i, j, c, k in zip(oi, oj, counts, kin)}
    c, k = pairs[(0, 1)]
    assert int(c["shared"]) == 0 and np.isnan(k)         # 0/0
    c, k = pairs[(1, 4)]
    assert int(c["opposing_hom"]) == m and k == -np.inf  # -4m/0
    c, k = pairs[(2, 3)]
    assert k == np.float32(0.5) and int(c["both_het"]) == m
    # NaN and -inf never pass any finite threshold (cuking.cu:297)
    res, _, _ = oracle.compute(sm, bits, -1e30)
    emitted = {(int(r["sample_i"]), int(r["sample_j"])) for r in res}
    assert (0, 1) not in emitted and (1, 4) not in emitted
    assert (2, 3) in emitted
    assert res.tobytes() == naive.king(geno, -1e30).tobytes()


def test_strict_threshold(oracle):
    geno = np.ones((2, 40), dtype=np.int8)  # kin exactly 0.5
    sm = oracle.submatrix(2)
    bits = oracle.bitset_from_genotypes(geno, sm)
    assert len(oracle.compute(sm, bits, 0.5)[0]) == 0
    assert len(oracle.compute(sm, bits, np.nextafter(np.float32(0.5),
                                                     np.float32(0)))[0]) == 1


def test_padding_is_missing(oracle):
    lib = oracle.load()
    assert lib.orc_padded_sites(1) == 32 and lib.orc_padded_sites(32) == 32
    assert lib.orc_padded_sites(33) == 64
    assert lib.orc_words_per_sample(1) == 2        # cuking.cu:513
    assert lib.orc_words_per_sample(64) == 2
    assert lib.orc_words_per_sample(65) == 4
    assert lib.orc_words_per_sample(10000) == 314
    assert lib.orc_words_per_sample(100000) == 3126
    assert lib.orc_words_per_sample(150000) == 4688
    assert lib.orc_words_per_sample(200000) == 6250
    geno = np.ones((2, 5), dtype=np.int8)
    bits = oracle.bitset_from_genotypes(geno)
    # plane = 1 word; bits 5..63 stay set in both planes (missing)
    assert bits[0, 0] == np.uint64(0xFFFFFFFFFFFFFFFF)
    assert bits[0, 1] == np.uint64(0xFFFFFFFFFFFFFFE0)


def test_pack_encoding_and_errors(oracle):
    sm = oracle.submatrix(3)
    bits = oracle.new_bitset(sm, 64)
    oracle.pack(sm, bits, [0, 1, 2, 5], [0, 0, 0, 2], [0, 1, 2, 1])
    het, hom = int(bits[0, 0]), int(bits[0, 1])
    assert (het >> 0) & 1 == 0 and (hom >> 0) & 1 == 0  # hom-ref 00
    assert (het >> 1) & 1 == 1 and (hom >> 1) & 1 == 0  # het 10
    assert (het >> 2) & 1 == 0 and (hom >> 2) & 1 == 1  # hom-var 01
    assert (het >> 3) & 1 == 1 and (hom >> 3) & 1 == 1  # untouched = missing
    assert int(bits[1, 0]) == 0xFFFFFFFFFFFFFFFF        # sample 1 untouched
    with pytest.raises(ValueError):
        oracle.pack(sm, bits, [0], [0], [3])            # cuking.cu:698-702
    with pytest.raises(ValueError):
        oracle.pack(sm, bits, [64], [0], [0])           # beyond padded sites
    # a sample outside the shard is skipped silently (cuking.cu:677-679)
    oracle.pack(sm, bits, [0], [17], [0])


def test_conflicting_duplicates_and_order_independence(oracle):
    sm = oracle.submatrix(2)
    a = oracle.new_bitset(sm, 10)
    b = oracle.new_bitset(sm, 10)
    rows, cols, alts = [3, 3, 4], [1, 1, 0], [1, 2, 2]
    oracle.pack(sm, a, rows, cols, alts)
    oracle.pack(sm, b, rows[::-1], cols[::-1], alts[::-1])
    assert np.array_equal(a, b)
    # 1 then 2 on one site clears both bits = hom-ref
    assert (int(a[1, 0]) >> 3) & 1 == 0 and (int(a[1, 1]) >> 3) & 1 == 0


@pytest.mark.parametrize("n", [1, 2, 5, 10, 37, 100])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 7])
def test_submatrix_blocks_partition_the_triangle(oracle, n, k):
    lib = oracle.load()
    seen = np.zeros((n, n), dtype=np.int32)
    shards = k * (k + 1) // 2
    for shard in range(shards):
        sm = oracle.submatrix(n, k, shard)
        ib, ie, jb, je = sm.as_tuple()
        assert ib <= ie <= n and jb <= je <= n and ib <= jb
        rows, cols = lib.orc_num_rows(C.byref(sm)), lib.orc_num_cols(C.byref(sm))
        assert rows == ie - ib and cols == je - jb
        stored = lib.orc_num_samples(C.byref(sm))
        assert stored == (rows if ib == jb else rows + cols)
        offs = set()
        for s in range(n):
            if lib.orc_contains(C.byref(sm), s):
                offs.add(lib.orc_sample_offset(C.byref(sm), s))
        assert offs == set(range(stored))
        for i in range(ib, ie):
            for j in range(max(jb, i + 1), je):
                seen[i, j] += 1
    iu = np.triu_indices(n, 1)
    assert np.all(seen[iu] == 1) and seen.sum() == n * (n - 1) // 2
    with pytest.raises(ValueError):
        oracle.submatrix(n, k, shards)
    with pytest.raises(ValueError):
        oracle.submatrix(n, 0, 0)


def test_reference_block_numbering(oracle):
    # cuking.cu:136-144 with k = 4: shards 0..3 = (0,0..3), 4..6 = (1,1..3), ...
    n, k = 400, 4
    exp = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (1, 3), (2, 2),
           (2, 3), (3, 3)]
    for shard, (bi, bj) in enumerate(exp):
        sm = oracle.submatrix(n, k, shard)
        assert sm.as_tuple() == (bi * 100, bi * 100 + 100, bj * 100,
                                 bj * 100 + 100)


@pytest.mark.parametrize("k", [2, 3, 5])
def test_sharded_union_equals_unsharded(oracle, k):
    rng = np.random.default_rng(7)
    n, m = 41, 200
    geno = random_genotypes(rng, n, m, missing=0.03)
    geno[30] = geno[2]
    full_sm = oracle.submatrix(n)
    full, _, _ = oracle.compute(full_sm, oracle.bitset_from_genotypes(geno, full_sm), -0.2)
    parts = []
    for shard in range(k * (k + 1) // 2):
        sm = oracle.submatrix(n, k, shard)
        bits = oracle.bitset_from_genotypes(geno, sm)  # shard-local storage
        parts.append(oracle.compute(sm, bits, -0.2)[0])
    merged = np.concatenate(parts)
    merged = merged[np.lexsort((merged["kin"], merged["sample_j"],
                                merged["sample_i"]))]
    assert merged.tobytes() == full.tobytes()


def test_overflow_and_multithreaded(oracle):
    rng = np.random.default_rng(3)
    geno = random_genotypes(rng, 50, 300)
    sm = oracle.submatrix(50)
    bits = oracle.bitset_from_genotypes(geno, sm)
    full, ovf, n = oracle.compute(sm, bits, -5.0)
    assert ovf == 0 and n == len(full) > 100
    cut, ovf, n2 = oracle.compute(sm, bits, -5.0, max_results=100)
    assert ovf == 1 and n2 == n and len(cut) == 100
    mt, ovf, n3 = oracle.compute(sm, bits, -5.0, threads=4)
    assert ovf == 0 and n3 == n and mt.tobytes() == full.tobytes()


def test_sort_order(oracle):
    from oracle.pyoracle import RESULT_DTYPE
    rng = np.random.default_rng(0)
    recs = np.zeros(200, dtype=RESULT_DTYPE)
    recs["sample_i"] = rng.integers(0, 5, 200)
    recs["sample_j"] = rng.integers(0, 5, 200)
    recs["kin"] = rng.random(200).astype(np.float32)
    exp = recs[np.lexsort((recs["kin"], recs["sample_j"], recs["sample_i"]))]
    got = recs.copy()
    oracle.load().orc_sort(got.ctypes.data_as(C.c_void_p), got.size)
    assert got.tobytes() == exp.tobytes()


def load_synth_golden():
    g = json.loads((GOLDEN / "synth_96x700.json").read_text())
    bits = np.frombuffer(bytes.fromhex("".join(g["bitset_hex"])), dtype=np.uint64).reshape(
        g["num_samples"], g["bitset_words_per_sample"]).copy()
    return g, bits


def golden_records(rows):
    from oracle.pyoracle import RESULT_DTYPE
    out = np.zeros(len(rows), dtype=RESULT_DTYPE)
    for k, (i, j, kin_bits, a, b, c) in enumerate(rows):
        out[k] = (i, j, np.uint32(kin_bits).view(np.float32), a, b, c)
    return out


def test_synth_golden_fixture(oracle):
    """tests/golden/synth_96x700.json (made by make_golden.py): generator,
    pack layout and records are stable."""
    g, bits = load_synth_golden()
    again = oracle.synth_bitset(g["seed"], g["kind"], g["pa"], g["pb"], 0, g["num_samples"],
                                g["num_sites"])
    assert np.array_equal(again, bits)
    res, ovf, _ = oracle.compute(oracle.submatrix(g["num_samples"]), bits, g["kin_threshold"])
    assert res.tobytes() == golden_records(g["records"]).tobytes()
    pairs = {(r[0], r[1]) for r in g["records"]}
    assert {(3, 90), (10, 91), (11, 91), (91, 92), (10, 93)} <= pairs


def test_formula_is_the_published_king_robust_estimator(oracle):
    """The reference's expression (cuking.cu:291-294) is algebraically the
    between-family KING-robust estimator of Manichaikul et al. 2010 (eq. 11),
    which Hail documents for hl.king (linked at cuking.cu:231):
        phi = (N_AaAa - 2 N_AAaa) / (2 min(N_Aa_i, N_Aa_j))
              + 1/2 - (N_Aa_i + N_Aa_j) / (4 min(N_Aa_i, N_Aa_j))
    Evaluated in float64 it agrees with the float32 two-rounding value to
    within float32 resolution (|delta| <= ~2^-23 at |kin|<=1;
    bit-equality with a float64 implementation is not definable)."""
    rng = np.random.default_rng(2010)
    geno = random_genotypes(rng, 80, 4000, missing=0.02)
    geno[79] = geno[0]
    sm = oracle.submatrix(80)
    _, _, c, kin = oracle.all_pairs(sm, oracle.bitset_from_genotypes(geno, sm))
    hi, hj = c["het_i"].astype(np.float64), c["het_j"].astype(np.float64)
    bh, opp = c["both_het"].astype(np.float64), c["opposing_hom"].astype(np.float64)
    mn = np.minimum(hi, hj)
    ok = mn > 0
    paper = (bh - 2 * opp) / (2 * mn) + 0.5 - (hi + hj) / (4 * mn)
    assert ok.sum() > 3000
    delta = np.abs(paper[ok] - kin[ok].astype(np.float64))
    assert delta.max() <= 2.0 ** -22
    # the float32 value is the two-rounding evaluation fl(0.5 + fl(num/den)), not
    # the rounding of the float64 value: near kin = 0 (quotient near -0.5) most
    # pairs differ in the last bits, which is why parity is defined against the
    # reference's float32 expression and only a tolerance against float64
    direct = (0.5 + (2 * bh - 4 * opp - hi - hj)[ok] / (4 * mn[ok])).astype(np.float32)
    assert np.mean(direct != kin[ok]) > 0.05
