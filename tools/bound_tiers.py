"""CPU experiment (numpy / torch on the host, no GPU): what a filter tier can see.

For a sample of the synthetic cohort with extra missing calls, every pair's
  * exact kinship (the reference's expression, float64 here: only its spread matters),
  * tier 1: the ONE-product bound of king_filter.hip,
        kin <= 1/2 - (u_i + u_j - 2 q) / (4 min(|H_i|, |H_j|)),   u = |Y| - |M|, q = T_i.T_j,
  * tier 2: exact X from q and the cross term W = Hm_i.M_j + M_i.Hm_j (Hm = the het
        plane as stored: het or missing), i.e. THREE product-equivalents on the same
        two-bit bytes (W is one product of doubled k), denominators still |H|,
  * the prefix form of tier 1 at a fraction f of the sites (every term of X is
        non-negative, so the prefix sum bounds X from below): which pairs a tile could
        already rule out at f, and whether a whole 256 x 256 tile could leave there.
Prints, per missing rate: mean / max of the bound over unrelated pairs, the share of
pairs above each threshold (= candidates), and the share of tiles that hold no
candidate at the prefix.  Model to compare with: tier 1 ~ m (1 + 1.37 m) for unrelated
pairs at missing rate m and het rate 0.365; tier 2 ~ 0.45 m.

usage: python tools/bound_tiers.py [samples=1536] [sites=100000] > profiles/r04_bound_tiers.txt
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from cuking_amd.synth import plan_cohort
from oracle import pyoracle

SEED = 20240229


def planes(bits):
    n, wps = bits.shape
    half = wps // 2
    het = np.unpackbits(bits[:, :half].copy().view(np.uint8), axis=1, bitorder="little")
    hom = np.unpackbits(bits[:, half:2 * half].copy().view(np.uint8), axis=1, bitorder="little")
    return het.astype(bool), hom.astype(bool)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    torch.set_num_threads(8)
    cohort = plan_cohort(n, SEED)
    bits = pyoracle.synth_bitset(SEED, cohort.kind, cohort.pa, cohort.pb, 0, n, m)
    het0, hom0 = planes(bits)
    het0, hom0 = het0[:, :m], hom0[:, :m]
    rng = np.random.default_rng(7)
    founders = np.flatnonzero(cohort.kind == 0)
    iu = np.triu_indices(len(founders), 1)
    print(f"# {n} samples x {m} sites of the synthetic cohort (1 % missing) + extra missing calls;")
    print(f"# statistics over the {len(iu[0])} founder pairs (unrelated); thresholds 0.0884 / 0.05 / 0.0442")
    for extra in (0.0, 0.03, 0.07, 0.12):
        miss = (het0 & hom0) | (rng.random(het0.shape) < extra)
        H = het0 & ~miss
        A = hom0 & ~miss
        R = ~het0 & ~hom0 & ~miss
        f32 = lambda x: torch.from_numpy(np.ascontiguousarray(x[founders]).astype(np.float32))  # noqa: E731
        T = f32(R) - f32(A)
        Hf, Mf, Yf = f32(H), f32(miss), f32(R | A)
        Hm = Hf + Mf
        q = (T @ T.T).numpy()
        W = (Hm @ Mf.T).numpy()
        W = W + W.T
        HD = (Hf @ (1 - Mf).T).numpy()          # het_i over the sites defined in both
        nH, nM, nY = Hf.sum(1).numpy(), Mf.sum(1).numpy(), Yf.sum(1).numpy()
        u = nY - nM
        x_lb = u[:, None] + u[None, :] - 2 * q
        x_true = x_lb + W
        den_b = 4 * np.minimum(nH[:, None], nH[None, :])
        den_t = 4 * np.minimum(HD, HD.T)
        kin = (0.5 - x_true / den_t)[iu]
        t1 = (0.5 - x_lb / den_b)[iu]
        t2 = (0.5 - x_true / den_b)[iu]
        rate = float(miss.mean())
        print(f"missing {rate:.4f}: exact kin mean {kin.mean():+.4f} sd {kin.std():.4f} | "
              f"tier 1 mean {t1.mean():.4f} max {t1.max():.4f} (model {rate * (1 + 1.37 * rate):.4f}) | "
              f"tier 2 mean {t2.mean():.4f} max {t2.max():.4f} (model {0.45 * rate:.4f})")
        for thr in (0.0884, 0.05, 0.0442):
            print(f"    thr {thr}: candidates tier 1 {np.mean(t1 > thr):.5f}  tier 2 {np.mean(t2 > thr):.5f}"
                  f"  (exact kin above: {np.mean(kin > thr):.6f})")
        # prefix form of tier 1: pair ruled out at fraction f iff
        #   u'_i + u'_j - 2 q' >= t min(|H_i|, |H_j|) + 8
        for thr in (0.0884, 0.05):
            t = 2 - 4 * thr
            bound = t * np.minimum(nH[:, None], nH[None, :]) + 8
            line = f"    thr {thr} prefix:"
            for f in (0.80, 0.84, 0.88, 0.92, 0.96):
                k = int(m * f) // 256 * 256
                Tp, Yp, Mp = T[:, :k], Yf[:, :k], Mf[:, :k]
                up = (Yp.sum(1) - Mp.sum(1)).numpy()
                lb = up[:, None] + up[None, :] - 2 * (Tp @ Tp.T).numpy()
                alive = lb < bound
                np.fill_diagonal(alive, False)
                # 256 x 256 tiles of the founder block
                nt = len(founders) // 256
                tiles = [alive[a * 256:(a + 1) * 256, b * 256:(b + 1) * 256].any()
                         for a in range(nt) for b in range(a, nt)]
                line += f"  f={f:.2f} pairs alive {alive[iu].mean():.5f} tiles alive {np.mean(tiles):.2f}"
            print(line)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
