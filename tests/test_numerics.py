"""The numerics contract of the path, on the CPU:

* the six sums are exact integers -- closed-form known answers from chosen
  multiplicities of all 16 genotype-pair classes, including one pair wide
  enough that 4 x opposing_hom exceeds 2^24;
* kin is the reference's float32 expression (cuking.cu:289-294): two IEEE
  roundings whenever numerator and denominator are exact, which every
  association order and every FMA contraction guarantees below 2^22 sites;
* against Hail's float64 estimator (hl.king, the formula linked at
  cuking.cu:231) only |kin32 - kin64| <= 2^-24 (|q| + |kin|) is definable
  (double rounding, App. A.3); <= 2^-24 for every emitted record.

PARITY UNPINNED by the reference (no vectors there); these are the strongest
pins that can be derived without it.
"""
from fractions import Fraction

import numpy as np
import pytest

from conftest import (kin_exact_two_roundings, pair_from_classes, random_genotypes,
                      round_to_f32)

NAMES = ("het_i", "het_j", "both_het", "opposing_hom", "concordant_hom", "shared")

# rows: state of sample i (hom-ref, het, hom-alt, missing); columns: sample j
PRIMES = [[101, 103, 107, 109], [113, 127, 131, 137], [139, 149, 151, 157],
          [163, 167, 173, 179]]
CLASS_CASES = {
    "distinct_primes": PRIMES,
    "no_hets": [[50, 0, 7, 3], [0, 0, 0, 0], [9, 0, 60, 2], [1, 0, 4, 5]],
    "only_missing_overlap": [[0, 0, 0, 9], [0, 0, 0, 8], [0, 0, 0, 7], [6, 5, 4, 3]],
    "identical_hets": [[0, 0, 0, 0], [0, 777, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1]],
    "one_site": [[0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]],
    "k_padding_edges": [[31, 1, 0, 0], [0, 33, 0, 0], [0, 0, 63, 1], [1, 0, 0, 65]],
}


def one_hot_cases():
    for a in range(4):
        for b in range(4):
            m = [[0] * 4 for _ in range(4)]
            m[a][b] = 37 + 4 * a + b
            yield f"only_{a}{b}", m


CLASS_CASES.update(dict(one_hot_cases()))


def oracle_pair(oracle, geno):
    sm = oracle.submatrix(2)
    bits = oracle.bitset_from_genotypes(geno, sm)
    _, _, counts, kin = oracle.all_pairs(sm, bits)
    return counts[0], kin[0]


@pytest.mark.parametrize("name", sorted(CLASS_CASES))
def test_closed_form_class_counts(oracle, naive, name):
    geno, want = pair_from_classes(CLASS_CASES[name], seed=len(name))
    counts, kin = oracle_pair(oracle, geno)
    assert {n: int(counts[n]) for n in NAMES} == want
    assert naive.pair_counts_loop(geno[0], geno[1]) == tuple(want[n] for n in NAMES)
    if min(want["het_i"], want["het_j"]) > 0:
        exact = kin_exact_two_roundings(want["het_i"], want["het_j"], want["both_het"],
                                        want["opposing_hom"])
        assert np.float32(kin).view(np.uint32) == exact.view(np.uint32)
    else:
        assert not (kin > -np.inf)      # -inf or NaN: never emitted (App. A.4)


def reference_expression_f32(het_i, het_j, both_het, opp):
    """cuking.cu:291-294 left to right, one float32 rounding per operation
    (no contraction)."""
    f = np.float32
    num = f(f(f(f(2) * f(both_het)) - f(f(4) * f(opp))) - f(het_i)) - f(het_j)
    den = f(4) * f(min(het_i, het_j))
    with np.errstate(divide="ignore", invalid="ignore"):
        return f(f(0.5) + f(f(num) / den))


def test_wide_pair_with_4opp_beyond_2_24(oracle):
    """4 x opposing_hom = 2^24 + 12 > 2^24 (4.6 M sites): the sums stay exact
    integers; kin is the reference's expression evaluated without contraction
    (above 2^22 sites the expression is no longer association-independent --
    the documented edge of the bit-exactness contract)."""
    opp = (1 << 22) + 3
    mult = [[5, 7, opp // 2, 3], [11, 400_001, 13, 2], [opp - opp // 2, 17, 19, 1],
            [4, 6, 8, 10]]
    geno, want = pair_from_classes(mult, seed=5)
    assert 4 * want["opposing_hom"] > (1 << 24)
    counts, kin = oracle_pair(oracle, geno)
    assert {n: int(counts[n]) for n in NAMES} == want
    ref = reference_expression_f32(want["het_i"], want["het_j"], want["both_het"],
                                   want["opposing_hom"])
    assert np.float32(kin).view(np.uint32) == ref.view(np.uint32)


def test_numerator_is_association_independent_below_2_22_sites():
    """App. A.2: for M < 2^22 every partial sum of 2bh - 4opp - hi - hj is an
    integer below 2^24 in magnitude, so float32 evaluates it exactly in every
    order and with every fused multiply-add; above that it does not."""
    rng = np.random.default_rng(7)
    f = np.float32
    for _ in range(20000):
        m = int(rng.integers(1, 1 << 22))
        # a feasible split of m sites: opp + hi-only + hj-only + bh + rest
        cuts = np.sort(rng.integers(0, m + 1, size=4))
        opp, only_i, only_j, bh = (int(x) for x in np.diff(np.concatenate([[0], cuts])))
        hi, hj = only_i + bh, only_j + bh
        exact = 2 * bh - 4 * opp - hi - hj
        a, b, c, d = f(2) * f(bh), f(4) * f(opp), f(hi), f(hj)
        orders = [((a - b) - c) - d, (a - c) - (b + d), a - ((b + c) + d),
                  ((a - d) - c) - b, (a - b) - (c + d)]
        # fused forms: fma(2, bh, -4 opp) etc. are exact products + one rounding
        fused = [f(float(2 * bh - 4 * opp)) - c - d,
                 f(float(2 * bh - hi)) - b - d]
        for v in orders + fused:
            assert float(v) == exact
    # ... and a witness that the bound matters (9,000,003 sites > 2^22): hi + hj
    # is odd and above 2^24, so the order (a - b) - (c + d) rounds it away
    bh, opp, hi, hj = 9_000_001, 1, 9_000_001, 9_000_002
    a, b, c, d = f(2) * f(bh), f(4) * f(opp), f(hi), f(hj)
    assert float(((a - b) - c) - d) == 2 * bh - 4 * opp - hi - hj == -5
    assert float((a - b) - (c + d)) == -6


def hail_king_f64(het_i, het_j, both_het, opp):
    """Hail's between-family KING estimator in float64 (documentation of
    hl.king, linked at cuking.cu:231)."""
    het_i, het_j = np.asarray(het_i, np.float64), np.asarray(het_j, np.float64)
    num = 2.0 * np.asarray(both_het, np.float64) - 4.0 * np.asarray(opp, np.float64) - het_i - het_j
    with np.errstate(divide="ignore", invalid="ignore"):
        return 0.5 + num / (4.0 * np.minimum(het_i, het_j))


@pytest.mark.parametrize("n,m", [(120, 100_000), (48, 200_000)])
def test_kin_within_double_rounding_of_hail_float64(oracle, n, m):
    """Full-width cohorts (BASELINE configs' 100k / 200k sites) with
    relatives: integer statistics are exactly Hail's; kin differs from the
    float64 estimator by at most the double-rounding bound."""
    rng = np.random.default_rng(m)
    geno = random_genotypes(rng, n, m, missing=0.01)
    geno[1] = geno[0]                                   # duplicate
    geno[3, : m // 2] = geno[2, : m // 2]               # half identical
    child = np.where(rng.random(m) < 0.5, geno[4], geno[5])
    geno[6] = np.where((geno[4] >= 0) & (geno[5] >= 0), child, -1)
    sm = oracle.submatrix(n)
    bits = oracle.bitset_from_genotypes(geno, sm)
    oi, oj, counts, kin = oracle.all_pairs(sm, bits)
    # integer statistics from the definitions on genotypes (indicator products)
    from oracle import naive_oracle
    _, _, nc = naive_oracle.all_pairs_matmul(geno)
    for k, name in enumerate(NAMES):
        assert np.array_equal(counts[name].astype(np.int64), nc[:, k]), name
    k64 = hail_king_f64(nc[:, 0], nc[:, 1], nc[:, 2], nc[:, 3])
    ok = np.isfinite(k64)
    assert ok.sum() > 0.9 * len(k64)
    q = k64[ok] - 0.5
    bound = 2.0 ** -24 * (np.abs(q) + np.abs(k64[ok])) * (1 + 1e-6)
    diff = np.abs(kin[ok].astype(np.float64) - k64[ok])
    assert np.all(diff <= bound), float((diff / bound).max())
    emitted = ok & (k64 > 0.05)
    assert emitted.sum() >= 3
    assert np.all(np.abs(kin[emitted].astype(np.float64) - k64[emitted]) <= 2.0 ** -24)
    # and the double rounding is real: some pairs are NOT the rounded f64 value
    assert np.any(kin[ok] != k64[ok].astype(np.float32))


def test_round_to_f32_helper():
    for x in (Fraction(1, 3), Fraction(-11, 12), Fraction(16777217, 1), Fraction(1, 2),
              Fraction(5, 1 << 30)):
        assert round_to_f32(x) == np.float32(x.numerator / x.denominator)
