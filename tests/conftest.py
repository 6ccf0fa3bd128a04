import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; without a device they fail loudly
    # rather than silently passing on some fallback.
    pass


def random_genotypes(rng, n, m, missing=0.05, af_lo=0.05, af_hi=0.5):
    """int8 [n, m] genotypes, -1 = missing (HWE per site)."""
    af = rng.uniform(af_lo, af_hi, size=m)
    g = (rng.random((n, m)) < af).astype(np.int8) + \
        (rng.random((n, m)) < af).astype(np.int8)
    g[rng.random((n, m)) < missing] = -1
    return g


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.load()
    return pyoracle


@pytest.fixture(scope="session")
def naive():
    from oracle import naive_oracle
    return naive_oracle


@pytest.fixture(scope="session")
def ctx():
    """A KingContext on cuda:0 (GPU tests only)."""
    import cuking_amd
    c = cuking_amd.KingContext(0)
    yield c
    c.close()


# ---------------------------------------------------------------------------
# Closed-form known-answer material: a pair of samples built from chosen
# multiplicities of the 16 genotype-pair classes.
# ---------------------------------------------------------------------------
GENOTYPE_STATES = (0, 1, 2, -1)     # hom-ref, het, hom-alt, missing


def pair_from_classes(mult, seed=0):
    """mult[a][b] = number of sites where sample i has state GENOTYPE_STATES[a]
    and sample j has GENOTYPE_STATES[b].  Returns (geno int8 [2, M] in a
    shuffled site order, expected six sums as a dict) -- the sums follow from
    the multiplicities alone, no bit arithmetic."""
    mult = np.asarray(mult, dtype=np.int64).reshape(4, 4)
    gi = np.concatenate([np.full(int(mult[a, b]), GENOTYPE_STATES[a], dtype=np.int8)
                         for a in range(4) for b in range(4)])
    gj = np.concatenate([np.full(int(mult[a, b]), GENOTYPE_STATES[b], dtype=np.int8)
                         for a in range(4) for b in range(4)])
    perm = np.random.default_rng(seed).permutation(gi.size)
    d = mult[:3, :3]                    # both defined
    expected = {
        "het_i": int(d[1, :].sum()), "het_j": int(d[:, 1].sum()),
        "both_het": int(d[1, 1]),
        "opposing_hom": int(d[0, 2] + d[2, 0]),
        "concordant_hom": int(d[0, 0] + d[2, 2]),
        "shared": int(d.sum()),
    }
    return np.stack([gi[perm], gj[perm]]), expected


def round_to_f32(x):
    """Fraction -> nearest float32 (ties to even), exact rational arithmetic:
    an evaluation of IEEE rounding that shares nothing with any FPU."""
    from fractions import Fraction
    x = Fraction(x)
    if x == 0:
        return np.float32(0.0)
    sign = -1 if x < 0 else 1
    x = abs(x)
    e = 0
    while x >= 2:
        x /= 2
        e += 1
    while x < 1:
        x *= 2
        e -= 1
    assert -126 <= e <= 127, "outside the normal range"
    scaled = x * (1 << 23)              # in [2^23, 2^24)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n & 1):
        n += 1
    return np.float32(sign * float(Fraction(n, 1 << 23) * Fraction(2) ** e))


def kin_exact_two_roundings(het_i, het_j, both_het, opp):
    """cuking.cu:289-294 for counts whose numerator and denominator are exact
    in float32 (< 2^24): q = RN32(num / den), kin = RN32(1/2 + q), in exact
    rational arithmetic."""
    from fractions import Fraction
    num = 2 * both_het - 4 * opp - het_i - het_j
    den = 4 * min(het_i, het_j)
    assert abs(num) < (1 << 24) and 0 < den < (1 << 24)
    q = Fraction(float(round_to_f32(Fraction(num, den))))
    return round_to_f32(Fraction(1, 2) + q)
