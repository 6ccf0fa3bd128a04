import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import cuking_amd
from conftest import random_genotypes
from oracle import pyoracle as oracle
ctx = cuking_amd.KingContext(0)
ctx.set_option("variant", 5)
rng = np.random.default_rng(99)
n, m = 150, 777
geno = random_genotypes(rng, n, m, missing=0.03)
geno[40] = geno[10]
geno[41, :400] = geno[11, :400]
geno[77] = -1
geno[78] = 0
bits = oracle.bitset_from_genotypes(geno)
sm = cuking_amd.Submatrix(n)
d = ctx.upload_bitset(bits)
for rep in range(3):
    got = ctx.compute_counts(sm, bits.shape[1], d)
    oi, oj, oc, _ = oracle.all_pairs(oracle.submatrix(n), bits)
    sel = got[oi, oj]
    bad = np.nonzero(sel["concordant_hom"] != oc["concordant_hom"])[0]
    print("counts: bad pairs", len(bad))
    exp, _, _ = oracle.compute(oracle.submatrix(n), bits, -1e30)
    res = ctx.run(sm, bits.shape[1], d, -1e30)
    badr = np.nonzero(res["ibs2"] != exp["ibs2"])[0]
    print("run: records", len(res), len(exp), "bad ibs2", len(badr), "other fields equal",
          all(np.array_equal(res[f], exp[f]) for f in ("sample_i", "sample_j", "ibs0", "ibs1")))
    for k in badr[:10]:
        print("   ", res["sample_i"][k], res["sample_j"][k], int(res["ibs2"][k]), int(exp["ibs2"][k]))
    if len(badr):
        print("   rows", np.unique(res["sample_i"][badr] // 32), "cols", np.unique(res["sample_j"][badr] // 32),
              "diffs", np.unique(res["ibs2"][badr].astype(np.int64) - exp["ibs2"][badr]))
