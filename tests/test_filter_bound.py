"""CPU check of the arithmetic behind the filter variant (king_filter.hip): the
upper bound on kinship from ONE plane product per pair plus per-sample counts
never rejects a pair the reference reports, whatever the data (missingness up to
nearly everything, monomorphic sites, identical samples), evaluated in float32
exactly as the kernel's epilogue does.  No GPU involved."""
import numpy as np
import pytest

from conftest import random_genotypes


def sample_stats(bits):
    """What sample_stats_kernel stores: (|Y| - |M|, |H|) from the reference layout
    (first half of a sample's words het, second half hom_var; missing = both)."""
    n, wps = bits.shape
    het, hom = bits[:, :wps // 2], bits[:, wps // 2:]
    pop = lambda x: np.unpackbits(x.view(np.uint8), axis=1).sum(axis=1).astype(np.int64)
    yc, mc, hc = pop(~het), pop(het & hom), pop(het & ~hom)
    return (yc - mc).astype(np.float32), hc.astype(np.float32)


def t_plane(geno):
    """T = R - A per site (+1 hom-ref, -1 hom-alt, 0 het / missing)."""
    return (geno == 0).astype(np.float32) - (geno == 2).astype(np.float32)


def candidates(u, hc, q4, thr):
    """The epilogue of king_filter_kernel, float32 step by step: acc = 4 q."""
    t = np.float32(2.0) - np.float32(4.0) * np.float32(thr)
    hb = (t * hc + np.float32(8.0)).astype(np.float32)     # fmaf: one rounding less, never smaller by more than an ulp
    hb = np.nextafter(hb, np.float32(-np.inf))             # ... so test against the smaller neighbour
    x_lb = (np.float32(-0.5) * q4 + (u[:, None] + u[None, :]).astype(np.float32)).astype(np.float32)
    return x_lb < np.minimum(hb[:, None], hb[None, :])


@pytest.mark.parametrize("n,m,missing,af", [
    (60, 2000, 0.01, (0.05, 0.5)), (60, 2000, 0.0, (0.05, 0.5)), (40, 777, 0.3, (0.05, 0.5)),
    (40, 500, 0.9, (0.05, 0.5)), (50, 1500, 0.02, (0.0, 0.02)), (50, 1500, 0.02, (0.45, 0.5)),
    (30, 63, 0.1, (0.05, 0.5)), (30, 65, 0.5, (0.2, 0.3))])
def test_bound_never_rejects_a_reported_pair(oracle, n, m, missing, af):
    rng = np.random.default_rng(n * m + int(missing * 100))
    geno = random_genotypes(rng, n, m, missing=missing, af_lo=af[0], af_hi=af[1])
    # relatives of every degree, a sample with nothing but missing calls, one all het
    geno[n - 1] = geno[0]
    geno[n - 2] = np.where(rng.random(m) < 0.5, geno[1], geno[2])
    geno[n - 3] = -1
    geno[n - 4] = 1
    geno[n - 5] = np.where(rng.random(m) < 0.03, -1, geno[3])
    sm = oracle.submatrix(n)
    bits = oracle.bitset_from_genotypes(geno, sm)
    oi, oj, counts, kin = oracle.all_pairs(sm, bits)
    u, hc = sample_stats(bits)
    T = t_plane(geno)
    q = T @ T.T
    # the identity behind the bound: X = het_i + het_j - 2 both_het + 4 opposing_hom
    #                                  = Y_i.D_j + D_i.Y_j - 2 q
    Y = (np.abs(T) > 0).astype(np.float32)
    D = (geno >= 0).astype(np.float32)
    X = (counts["het_i"].astype(np.int64) + counts["het_j"] - 2 * counts["both_het"].astype(np.int64)
         + 4 * counts["opposing_hom"].astype(np.int64))
    ident = (Y @ D.T + D @ Y.T - 2 * q)[oi, oj]
    assert np.array_equal(ident.astype(np.int64), X)
    # ... and u_i + u_j - 2 q is a lower bound of it
    assert np.all((u[oi] + u[oj] - 2 * q[oi, oj]) <= X)
    for thr in (0.0005, 0.01, 0.0442, 0.05, 0.0884, 0.177, 0.25, 0.354, 0.49, 0.4999):
        cand = candidates(u, hc, (4 * q).astype(np.float32), thr)[oi, oj]
        reported = kin > np.float32(thr)
        assert not np.any(reported & ~cand), (thr, int(np.sum(reported & ~cand)))


def test_bound_is_tight_on_an_ordinary_cohort(oracle):
    """1 % missing, common variants: at the reference's default threshold the bound
    lets through (next to) nothing but the records."""
    rng = np.random.default_rng(5)
    n, m = 120, 20000
    geno = random_genotypes(rng, n, m, missing=0.01)
    geno[n - 1] = geno[0]
    geno[n - 2] = np.where(rng.random(m) < 0.5, geno[1], geno[2])
    sm = oracle.submatrix(n)
    bits = oracle.bitset_from_genotypes(geno, sm)
    oi, oj, _, kin = oracle.all_pairs(sm, bits)
    u, hc = sample_stats(bits)
    T = t_plane(geno)
    cand = candidates(u, hc, (4 * (T @ T.T)).astype(np.float32), 0.0884)[oi, oj]
    reported = kin > np.float32(0.0884)
    assert reported.sum() >= 1 and cand.sum() <= reported.sum() + 2
