#!/usr/bin/env python3
"""Regenerates tests/golden/synth_96x700.json: a small synthetic cohort
(cuking_amd.synth plan + oracle/synth_oracle.c genotypes) with the records the
CPU oracle computes for it, unsharded and for split_factor = 2.

NOT reference output (the reference ships no vectors and cannot run here,
SURVEY.md 8c): these vectors pin the oracle and the HIP path against
regressions between rounds.  The KAT kat_4x10.json is the hand-checked pin.

    python tests/golden/make_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from cuking_amd.synth import plan_cohort  # noqa: E402  (pure python, no GPU)
from oracle import pyoracle  # noqa: E402

N, M, SEED, THR = 96, 700, 424242, 0.04


def records(res):
    return [[int(r["sample_i"]), int(r["sample_j"]), int(r["kin"].view(np.uint32)),
             int(r["ibs0"]), int(r["ibs1"]), int(r["ibs2"])] for r in res]


def main():
    cohort = plan_cohort(N, SEED)
    # plan_cohort plants ~3.5 % relatives; N = 96 gives none, so plant a few by hand
    kind, pa, pb = cohort.kind.copy(), cohort.pa.copy(), cohort.pb.copy()
    kind[90], pa[90], pb[90] = 1, 3, 3          # duplicate of 3
    kind[91], pa[91], pb[91] = 2, 10, 11        # child of 10 x 11
    kind[92], pa[92], pb[92] = 2, 10, 11        # full sibling
    kind[93], pa[93], pb[93] = 2, 10, 12        # half sibling
    bits = pyoracle.synth_bitset(SEED, kind, pa, pb, 0, N, M)
    out = {"comment": __doc__.strip().splitlines()[0], "num_samples": N, "num_sites": M,
           "seed": SEED, "kin_threshold": THR,
           "kind": kind.tolist(), "pa": pa.tolist(), "pb": pb.tolist(),
           "bitset_words_per_sample": int(bits.shape[1]),
           "bitset_hex": [row.tobytes().hex() for row in bits],
           "record_fields": ["sample_i", "sample_j", "kin_bits", "ibs0", "ibs1", "ibs2"]}
    res, ovf, _ = pyoracle.compute(pyoracle.submatrix(N), bits, THR)
    assert ovf == 0
    out["records"] = records(res)
    out["shards_split_factor_2"] = []
    for shard in range(3):
        sm = pyoracle.submatrix(N, 2, shard)
        rows = list(range(sm.i_begin, sm.i_end))
        cols = [] if sm.i_begin == sm.j_begin else list(range(sm.j_begin, sm.j_end))
        local = np.ascontiguousarray(bits[rows + cols])
        r, ovf, _ = pyoracle.compute(sm, local, THR)
        out["shards_split_factor_2"].append(records(r))
    path = Path(__file__).resolve().parent / "synth_96x700.json"
    path.write_text(json.dumps(out, indent=None, separators=(",", ":")))
    print(path, len(out["records"]), "records", path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
