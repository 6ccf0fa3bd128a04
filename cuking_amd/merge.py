"""Downstream helper for sharded runs (`--split-factor k` => k(k+1)/2 part
files): checks that every shard's `part-NNNNN.snappy.parquet` is present,
optionally concatenates them into one table and writes the `_SUCCESS` marker
the reference's fleet script drops after all tasks succeeded
(cloud_batch_submit.py:111-124).  The concatenated table loads exactly like
cuking_outputs_to_ht.py:12-15 expects (columns i, j, kin, ibs0, ibs1, ibs2,
keyed by (i, j)).

    python -m cuking_amd.merge --output-uri out/ --split-factor 4 [--merged all.parquet]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path


def expected_parts(split_factor: int):
    return [f"part-{s:05d}.snappy.parquet"
            for s in range(split_factor * (split_factor + 1) // 2)]


def merge(out_dir, split_factor: int, merged=None, write_success: bool = True):
    import pyarrow as pa
    import pyarrow.parquet as pq
    out = Path(out_dir)
    missing = [p for p in expected_parts(split_factor) if not (out / p).is_file()]
    if missing:
        raise FileNotFoundError(f"{len(missing)} shard outputs missing, first: {missing[0]}")
    tables = [pq.read_table(out / p) for p in expected_parts(split_factor)]
    table = pa.concat_tables(tables)
    keys = set(zip(table.column("i").to_pylist(), table.column("j").to_pylist()))
    if len(keys) != table.num_rows:
        raise ValueError("a sample pair appears in more than one shard")
    if merged:
        pq.write_table(table, merged, compression="snappy")
    if write_success:
        (out / "_SUCCESS").write_bytes(b"")
    return table


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--output-uri", "--output_uri", dest="output_uri", required=True)
    ap.add_argument("--split-factor", "--split_factor", dest="split_factor", type=int, default=1)
    ap.add_argument("--merged", default=None)
    args = ap.parse_args(argv)
    uri = args.output_uri[7:] if args.output_uri.startswith("file://") else args.output_uri
    try:
        t = merge(uri, args.split_factor, args.merged)
    except (FileNotFoundError, ValueError) as e:
        print(f"Error: {e}", file=sys.stderr)
        return 1
    print(f"{t.num_rows} related pairs in {len(expected_parts(args.split_factor))} shards")
    return 0


if __name__ == "__main__":
    sys.exit(main())
