"""Synthetic cohort description for benchmarks and large parity runs.

The reference ships no synthetic inputs; this is the workload SURVEY.md 8(d)
defines: unrelated founders plus planted relatives so that a thresholded run
has non-trivial output:

    0.5 % duplicates            (kin ~ 0.5)
    1 %   parent-child trios    (each planted child with two founder parents)
    1 %   full siblings         (pairs of children of one founder couple)
    1 %   half siblings         (pairs of children sharing one founder parent)

The genotypes themselves come from the counter-based generator in
csrc/synth.hip (`KingContext.synth_bitset`); this module only builds the
kind / parent arrays it consumes.  Pure integer arithmetic, deterministic.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

DEFAULT_SEED = 20240229

KIND_FOUNDER, KIND_DUP, KIND_CHILD = 0, 1, 2

_M64 = (1 << 64) - 1


def _mix64(x: int) -> int:
    x &= _M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & _M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & _M64
    x ^= x >> 31
    return x


@dataclass
class Cohort:
    num_samples: int
    kind: np.ndarray  # uint32 [N]
    pa: np.ndarray    # uint32 [N]
    pb: np.ndarray    # uint32 [N]
    num_founders: int
    # (i, j, relation) with i < j, for tests: "dup", "po", "sib", "half"
    planted: list


def plan_cohort(num_samples: int, seed: int = DEFAULT_SEED) -> Cohort:
    """Founders first, derived samples at the end (parents are founders)."""
    n = num_samples
    n_dup = n // 200
    n_po = n // 100
    n_sib = (n // 100) // 2 * 2
    n_half = (n // 100) // 2 * 2
    n_derived = n_dup + n_po + n_sib + n_half
    n_f = n - n_derived
    kind = np.zeros(n, dtype=np.uint32)
    pa = np.zeros(n, dtype=np.uint32)
    pb = np.zeros(n, dtype=np.uint32)
    planted = []
    if n_f < 3:
        return Cohort(n, kind, pa, pb, n, planted)

    state = [_mix64(seed ^ 0xC0FFEE)]

    def pick(exclude=()):
        while True:
            state[0] = _mix64(state[0] + 0x9E3779B97F4A7C15)
            f = state[0] % n_f
            if f not in exclude:
                return f

    s = n_f
    for _ in range(n_dup):
        a = pick()
        kind[s], pa[s], pb[s] = KIND_DUP, a, a
        planted.append((a, s, "dup"))
        s += 1
    for _ in range(n_po):
        a = pick()
        b = pick((a,))
        kind[s], pa[s], pb[s] = KIND_CHILD, a, b
        planted.append((a, s, "po"))
        planted.append((b, s, "po"))
        s += 1
    for _ in range(n_sib // 2):
        a = pick()
        b = pick((a,))
        for _k in range(2):
            kind[s], pa[s], pb[s] = KIND_CHILD, a, b
            planted.append((a, s, "po"))
            planted.append((b, s, "po"))
            s += 1
        planted.append((s - 2, s - 1, "sib"))
    for _ in range(n_half // 2):
        a = pick()
        b = pick((a,))
        c = pick((a, b))
        kind[s], pa[s], pb[s] = KIND_CHILD, a, b
        kind[s + 1], pa[s + 1], pb[s + 1] = KIND_CHILD, a, c
        planted += [(a, s, "po"), (b, s, "po"), (a, s + 1, "po"),
                    (c, s + 1, "po"), (s, s + 1, "half")]
        s += 2
    assert s == n
    return Cohort(n, kind, pa, pb, n_f, planted)


def cohort_to_device(cohort: Cohort, device: int = 0):
    import torch
    dev = f"cuda:{device}"
    as_t = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    return as_t(cohort.kind), as_t(cohort.pa), as_t(cohort.pb)
