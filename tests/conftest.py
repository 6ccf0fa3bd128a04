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
