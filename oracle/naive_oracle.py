"""Naive per-genotype KING oracle: no bitsets, no popcounts.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Works on an int8 genotype matrix ``geno[N, M]`` (0/1/2 alt alleles, -1 =
missing) and restates the *definitions* behind cuking.cu:216-240 / :289-307
(and Hail's documented between-family KING estimator linked at cuking.cu:231)
directly on genotypes, so that it shares nothing with the bit-plane code it
checks.  Two forms: an explicit loop over pairs and sites (small inputs) and a
matrix-product form (indicator matrices; exact in int64).
"""
from __future__ import annotations

import numpy as np

from .pyoracle import COUNTS_DTYPE, RESULT_DTYPE


def kin_f32(het_i, het_j, both_het, opp) -> np.ndarray:
    """cuking.cu:289-294 in float32: fl(0.5 + fl(num / den)).

    num and den are integers < 2^24 in magnitude for < 2^22 sites, hence exact
    in float32 in any association order."""
    het_i = np.asarray(het_i, dtype=np.int64)
    het_j = np.asarray(het_j, dtype=np.int64)
    num = (2 * np.asarray(both_het, dtype=np.int64)
           - 4 * np.asarray(opp, dtype=np.int64) - het_i - het_j)
    den = 4 * np.minimum(het_i, het_j)
    assert np.all(np.abs(num) < (1 << 24)) and np.all(den < (1 << 24))
    with np.errstate(divide="ignore", invalid="ignore"):
        q = num.astype(np.float32) / den.astype(np.float32)
        return (np.float32(0.5) + q).astype(np.float32)


def pair_counts_loop(gi: np.ndarray, gj: np.ndarray) -> tuple:
    """Site-by-site definition for one pair."""
    het_i = het_j = both = opp = conc = shared = 0
    for a, b in zip(gi.tolist(), gj.tolist()):
        if a < 0 or b < 0:
            continue  # count only sites defined in both samples
        shared += 1
        het_i += a == 1
        het_j += b == 1
        both += (a == 1 and b == 1)
        opp += (a == 0 and b == 2) or (a == 2 and b == 0)
        conc += (a == 0 and b == 0) or (a == 2 and b == 2)
    return het_i, het_j, both, opp, conc, shared


def all_pairs_loop(geno: np.ndarray):
    n = geno.shape[0]
    ii, jj, rows = [], [], []
    for i in range(n):
        for j in range(i + 1, n):
            ii.append(i)
            jj.append(j)
            rows.append(pair_counts_loop(geno[i], geno[j]))
    counts = np.array(rows, dtype=np.int64).reshape(-1, 6)
    return np.array(ii, dtype=np.uint32), np.array(jj, dtype=np.uint32), counts


def all_pairs_matmul(geno: np.ndarray, i_range=None, j_range=None):
    """Counts for every (i in i_range, j in j_range, i < j) via indicator
    matrix products.  Returns (i, j, counts[.,6] int64) in (i, j) order."""
    n = geno.shape[0]
    i0, i1 = i_range or (0, n)
    j0, j1 = j_range or (0, n)
    def ind(x):  # float64 matmul is exact for these 0/1 sums (< 2^53)
        return x.astype(np.float64)
    gi, gj = geno[i0:i1], geno[j0:j1]
    Di, Dj = ind(gi >= 0), ind(gj >= 0)
    Hi, Hj = ind(gi == 1), ind(gj == 1)
    Ri, Rj = ind(gi == 0), ind(gj == 0)
    Ai, Aj = ind(gi == 2), ind(gj == 2)
    het_i = Hi @ Dj.T
    het_j = Di @ Hj.T
    both = Hi @ Hj.T
    opp = Ri @ Aj.T + Ai @ Rj.T
    conc = Ri @ Rj.T + Ai @ Aj.T
    shared = Di @ Dj.T
    I, J = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
    keep = I < J
    counts = np.stack([m[keep] for m in (het_i, het_j, both, opp, conc, shared)],
                      axis=1).round().astype(np.int64)
    return I[keep].astype(np.uint32), J[keep].astype(np.uint32), counts


def counts_struct(counts: np.ndarray) -> np.ndarray:
    out = np.zeros(counts.shape[0], dtype=COUNTS_DTYPE)
    for k, name in enumerate(COUNTS_DTYPE.names):
        out[name] = counts[:, k]
    return out


def king(geno: np.ndarray, kin_threshold: float, i_range=None, j_range=None,
         use_loop: bool = False) -> np.ndarray:
    """Thresholded, sorted result records (cuking.cu:297-307, :761-765)."""
    if use_loop:
        assert i_range is None and j_range is None
        i, j, c = all_pairs_loop(geno)
    else:
        i, j, c = all_pairs_matmul(geno, i_range, j_range)
    kin = kin_f32(c[:, 0], c[:, 1], c[:, 2], c[:, 3])
    keep = kin > np.float32(kin_threshold)  # strict; NaN / -inf never pass
    res = np.zeros(int(keep.sum()), dtype=RESULT_DTYPE)
    res["sample_i"], res["sample_j"], res["kin"] = i[keep], j[keep], kin[keep]
    res["ibs0"] = c[keep, 3]
    res["ibs2"] = c[keep, 4] + c[keep, 2]
    res["ibs1"] = c[keep, 5] - res["ibs0"] - res["ibs2"]
    order = np.lexsort((res["kin"], res["sample_j"], res["sample_i"]))
    return res[order]
