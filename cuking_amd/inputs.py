"""Writer for the reference's on-disk input contract, standing in for
mt_to_cuking_inputs.py (which needs Hail + Spark):

    <dir>/metadata.json   {"num_sites": int, "samples": [str, ...]}   (:40-47)
    <dir>/part-*.parquet  row_idx INT64, col_idx INT64, n_alt_alleles INT32,
                          one file per partition, zstd (:26-34); a missing
                          genotype is simply absent (cuking.cu:520-523)

Spark-style details are reproduced on request: OPTIONAL (nullable) columns,
`part-00000-<uuid>.c000.zstd.parquet` names, a `_SUCCESS` marker and a
`_temporary/` junk directory (the reference's non-recursive listing skips it,
cuking.cu:530-540).
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np


def write_input_tables(out_dir, geno: np.ndarray, sample_ids=None,
                       num_files: int = 8, compression: str = "zstd",
                       nullable: bool = True, spark_layout: bool = True,
                       row_group_size: int | None = None,
                       use_dictionary: bool = True, shuffle_seed=None):
    """geno: int8 [num_samples, num_sites], negative = missing."""
    import pyarrow as pa
    import pyarrow.parquet as pq

    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    n, m = geno.shape
    if sample_ids is None:
        sample_ids = [f"S{idx:07d}" for idx in range(n)]
    assert len(sample_ids) == n
    (out / "metadata.json").write_text(
        json.dumps({"num_sites": int(m), "samples": list(sample_ids)}))

    schema = pa.schema([pa.field("row_idx", pa.int64(), nullable=nullable),
                        pa.field("col_idx", pa.int64(), nullable=nullable),
                        pa.field("n_alt_alleles", pa.int32(), nullable=nullable)])
    bounds = np.linspace(0, m, num_files + 1).astype(np.int64)
    rng = np.random.default_rng(shuffle_seed) if shuffle_seed is not None else None
    paths = []
    for f in range(num_files):
        lo, hi = int(bounds[f]), int(bounds[f + 1])
        block = geno[:, lo:hi].T                     # [sites, samples]
        row, col = np.nonzero(block >= 0)            # row-major: site, sample
        alt = block[row, col].astype(np.int32)
        row = row.astype(np.int64) + lo
        col = col.astype(np.int64)
        if rng is not None:
            perm = rng.permutation(len(row))
            row, col, alt = row[perm], col[perm], alt[perm]
        table = pa.table({"row_idx": row, "col_idx": col, "n_alt_alleles": alt},
                         schema=schema)
        name = (f"part-{f:05d}-0f3a9c1e-7b5d-4c2a-9e61-c0ffee000000.c000.zstd.parquet"
                if spark_layout else f"part-{f:05d}.parquet")
        pq.write_table(table, out / name,
                       compression=None if compression in (None, "none") else compression,
                       row_group_size=row_group_size, use_dictionary=use_dictionary)
        paths.append(out / name)
    if spark_layout:
        (out / "_SUCCESS").write_bytes(b"")
        junk = out / "_temporary" / "0"
        junk.mkdir(parents=True, exist_ok=True)
        # A leftover table that must NOT be read (it would corrupt sample 0).
        bad = pa.table({"row_idx": np.arange(min(m, 4), dtype=np.int64),
                        "col_idx": np.zeros(min(m, 4), dtype=np.int64),
                        "n_alt_alleles": np.full(min(m, 4), 2, dtype=np.int32)},
                       schema=schema)
        pq.write_table(bad, junk / "part-99999.parquet")
    return paths


def read_results(out_dir):
    """All `part-*.snappy.parquet` of an output directory as one pyarrow table
    (what cuking_outputs_to_ht.py:12 feeds to Spark)."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    parts = sorted(Path(out_dir).glob("part-*.snappy.parquet"))
    return pa.concat_tables([pq.read_table(p) for p in parts])
