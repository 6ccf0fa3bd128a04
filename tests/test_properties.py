"""Property tests (hypothesis) on CPU: the C oracle's bit-plane arithmetic
against the naive per-genotype definitions for arbitrary small cohorts, and
the algebra the lean HIP kernel relies on, checked on the oracle's counts."""
import numpy as np
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as hnp

genotypes = st.integers(2, 12).flatmap(lambda n: st.integers(1, 140).flatmap(
    lambda m: hnp.arrays(np.int8, (n, m), elements=st.integers(-1, 2))))


@settings(max_examples=60, deadline=None)
@given(geno=genotypes, thr=st.sampled_from([-1e30, -0.5, 0.0, 0.0884, 0.25, 0.5]))
def test_oracle_equals_naive_for_any_cohort(oracle, naive, geno, thr):
    sm = oracle.submatrix(geno.shape[0])
    bits = oracle.bitset_from_genotypes(geno, sm)
    res, ovf, n = oracle.compute(sm, bits, thr)
    exp = naive.king(geno, thr, use_loop=True)
    assert ovf == 0 and res.tobytes() == exp.tobytes()


@settings(max_examples=60, deadline=None)
@given(geno=genotypes)
def test_identities_used_by_the_lean_kernel(oracle, geno):
    """ibs1 = het_i + het_j - 2*both_het; shared = het_i + het_j - both_het + hom_hom;
    concordant = hom_hom - opposing (king_kernels.hip) hold for the reference's sums."""
    sm = oracle.submatrix(geno.shape[0])
    bits = oracle.bitset_from_genotypes(geno, sm)
    oi, oj, c, _ = oracle.all_pairs(sm, bits)
    c = {k: c[k].astype(np.int64) for k in c.dtype.names}
    hom = lambda g: (g == 0) | (g == 2)
    hom_hom = np.array([np.sum(hom(geno[i]) & hom(geno[j])) for i, j in zip(oi, oj)],
                       dtype=np.int64)
    ibs0 = c["opposing_hom"]
    ibs2 = c["concordant_hom"] + c["both_het"]
    ibs1 = c["shared"] - ibs0 - ibs2
    assert np.array_equal(ibs1, c["het_i"] + c["het_j"] - 2 * c["both_het"])
    assert np.array_equal(c["shared"], c["het_i"] + c["het_j"] - c["both_het"] + hom_hom)
    assert np.array_equal(c["concordant_hom"], hom_hom - c["opposing_hom"])
    assert np.all(ibs1 >= 0) and np.all(c["both_het"] <= np.minimum(c["het_i"], c["het_j"]))


@settings(max_examples=40, deadline=None)
@given(n=st.integers(1, 60), k=st.integers(1, 7))
def test_shards_partition_pairs(oracle, n, k):
    import ctypes as C
    total = 0
    for shard in range(k * (k + 1) // 2):
        sm = oracle.submatrix(n, k, shard)
        for i in range(sm.i_begin, sm.i_end):
            total += max(0, sm.j_end - max(sm.j_begin, i + 1))
    assert total == n * (n - 1) // 2
