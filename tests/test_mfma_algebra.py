"""CPU check of the arithmetic behind the matrix-core kernel (king_mfma.hip):
the reference's six sums (cuking.cu:219-239) as inner products of bit planes
built with the kernel's truth tables, fp4 values and block scales -- emulated
in numpy against the oracle.  No GPU involved."""
import numpy as np
import pytest

from conftest import random_genotypes

# v_bitop3_b32 truth tables over (het, hom_var, mask), index = 4 het + 2 hom + mask
KINDS = {"A": 0x08, "R": 0x02, "H": 0x20, "D": 0x2A, "Y": 0x0A}
FP4 = {0x0: 0.0, 0x1: 0.5, 0x2: 1.0, 0x4: 2.0}      # E2M1 codes the expansion can produce


def bitop3(a, b, c, table):
    out = np.zeros_like(a)
    for idx in range(8):
        if table >> idx & 1:
            term = np.full_like(a, 0xFFFFFFFF)
            for bit, x in ((4, a), (2, b), (1, c)):
                term &= x if idx & bit else ~x
            out |= term
    return out & np.uint32(0xFFFFFFFF)


def fragments(het, hom, kind):
    """[f][word] -> 8 fp4 values per word (nibble q = site 4q+f), as floats,
    with the scale 2^(1-f) (f = 3: shifted to position 0 first) applied."""
    frags = []
    for f in range(4):
        if f < 3:
            word = bitop3(het, hom, np.uint32(0x11111111 << f), KINDS[kind])
            scale = 2.0 ** (1 - f)
        else:
            word = bitop3(het >> np.uint32(3), hom >> np.uint32(3), np.uint32(0x11111111),
                          KINDS[kind])
            scale = 2.0
        nibbles = np.stack([(word >> np.uint32(4 * q)) & np.uint32(0xF) for q in range(8)], -1)
        assert set(np.unique(nibbles).tolist()) <= set(FP4)
        vals = np.vectorize(FP4.get)(nibbles).astype(np.float32)
        frags.append(vals * np.float32(scale))
    return np.stack(frags)          # [4, samples, words, 8]; every entry 0.0 or 1.0


@pytest.mark.parametrize("n,m,missing", [(40, 333, 0.1), (25, 1000, 0.0), (30, 64, 0.5)])
def test_plane_products_equal_the_six_sums(oracle, n, m, missing):
    rng = np.random.default_rng(n * m)
    geno = random_genotypes(rng, n, m, missing=missing)
    geno[n - 1] = geno[0]
    bits = oracle.bitset_from_genotypes(geno)
    wps = bits.shape[1]
    words32 = bits.view(np.uint32).reshape(n, wps * 2)
    het, hom = words32[:, :wps], words32[:, wps:]
    frag = {k: fragments(het, hom, k) for k in KINDS}
    for k in frag:                   # after scaling every element is exactly 0 or 1
        assert set(np.unique(frag[k]).tolist()) <= {0.0, 1.0}

    def dot(x, y):                   # float32 accumulation, as the MFMA does
        return np.einsum("fiwq,fjwq->ij", frag[x], frag[y], dtype=np.float32)

    opp = dot("A", "R") + dot("R", "A")
    bh, hi, hj, hom_hom = dot("H", "H"), dot("H", "D"), dot("D", "H"), dot("Y", "Y")
    oi, oj, oc, _ = oracle.all_pairs(oracle.submatrix(n), bits)
    got = {
        "het_i": hi, "het_j": hj, "both_het": bh, "opposing_hom": opp,
        "concordant_hom": hom_hom - opp, "shared": hi + hj - bh + hom_hom,
    }
    for name, mat in got.items():
        assert np.array_equal(mat[oi, oj].astype(np.uint32), oc[name]), name
    # the lean form's derived statistics (cuking.cu:301-307)
    ibs0, ibs2 = oc["opposing_hom"], oc["concordant_hom"] + oc["both_het"]
    ibs1 = oc["shared"] - ibs0 - ibs2
    assert np.array_equal((hi + hj - 2 * bh)[oi, oj].astype(np.uint32), ibs1)
    assert np.array_equal((hom_hom - opp + bh)[oi, oj].astype(np.uint32), ibs2)


def test_float32_sums_are_exact_below_2_pow_24():
    """The accumulators hold integers: exact in float32 up to 2^24 in any order."""
    top = np.float32(2 ** 24 - 64)
    assert top + np.float32(64) == np.float32(2 ** 24)
    assert np.float32(2 ** 24) + np.float32(1) == np.float32(2 ** 24)   # the first inexact sum
    rng = np.random.default_rng(0)
    parts = rng.integers(0, 65, size=200000).astype(np.float32)          # 64-site products
    parts = parts[np.cumsum(parts, dtype=np.float64) < 2 ** 24]
    assert np.float32(parts.sum(dtype=np.float64)) == parts.sum(dtype=np.float32)
    assert parts[::-1].sum(dtype=np.float32) == parts.sum(dtype=np.float32)


def test_epilogue_prefilter_is_conservative():
    """king_device.h: kinship_may_pass() may only say "no" for pairs that fail
    the exact test `fl32(0.5 + fl32(num/den)) > thr` (cuking.cu:289-297).  The
    device uses v_rcp_f32 (1 ulp); emulated here with the reciprocal perturbed
    by up to +-4 ulp."""
    rng = np.random.default_rng(11)
    n = 400000
    sites = rng.choice([50, 1000, 100000, 4000000], size=n)
    hi = (rng.random(n) * sites).astype(np.int64)
    hj = np.where(rng.random(n) < 0.5, hi + rng.integers(-3, 4, size=n), (rng.random(n) * sites)).astype(np.int64)
    hj = np.clip(hj, 0, None)
    bh = (rng.random(n) * np.minimum(hi, hj)).astype(np.int64)
    opp = (rng.random(n) * (sites - np.maximum(hi, hj)) * rng.choice([0, 0.01, 0.3], size=n)).astype(np.int64)
    hi[:50] = 0                                   # zero-het samples: -inf / NaN
    f32 = np.float32
    with np.errstate(divide="ignore", invalid="ignore"):
        num = f32(2) * bh.astype(f32) - f32(4) * opp.astype(f32) - hi.astype(f32) - hj.astype(f32)
        den = f32(4) * np.minimum(hi, hj).astype(f32)
        kin = f32(0.5) + (num / den).astype(f32)
        for thr in (f32(-1e30), f32(-0.3), f32(0.0), f32(0.05), f32(0.0884), f32(0.25), f32(0.49)):
            exact = kin > thr
            for ulps in (-4, 0, 4):
                rcp = (f32(1) / den).astype(f32)
                rcp = np.where(np.isfinite(rcp),
                               (rcp.view(np.int32) + ulps).view(np.float32), rcp)
                q = (num * rcp).astype(f32)
                maybe = q >= (thr - f32(0.5)) - f32(1e-5) * (f32(1) + np.abs(q))
                assert not np.any(exact & ~maybe), (thr, ulps)
            # and it does filter: hardly any unrelated-looking pair survives at 0.25
        assert (maybe & ~exact).mean() < 0.05
