#!/usr/bin/env python3
"""End-to-end timing of the `cuking` binary on real Parquet input, host pack
against the pipelined device pack (phases as printed by the binary, triples/s
from its JSON line).

    cli_timing.py N M FILES [--threads T] [--repeat R] [--keep]

The input is generated file by file in worker processes (a file = a range of
sites, all samples; zstd like mt_to_cuking_inputs.py:31-34), so that
configs[1]-sized inputs (10k x 100k = 1e9 triples) never exist as one array."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def write_part(args):
    out, f, lo, hi, n, seed = args[:6]
    row_group_size = args[6] if len(args) > 6 else None
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng([seed, f])
    af = rng.uniform(0.05, 0.5, size=hi - lo)
    block = ((rng.random((hi - lo, n), dtype=np.float32) < af[:, None]).astype(np.int8) +
             (rng.random((hi - lo, n), dtype=np.float32) < af[:, None]).astype(np.int8))
    block[rng.random((hi - lo, n), dtype=np.float32) < 0.01] = -1
    block[:, n - 1] = block[:, 0]                 # one duplicate pair
    row, col = np.nonzero(block >= 0)             # site-major, like the Spark writer
    table = pa.table({"row_idx": (row + lo).astype(np.int64), "col_idx": col.astype(np.int64),
                      "n_alt_alleles": block[row, col].astype(np.int32)})
    pq.write_table(table, Path(out) / f"part-{f:05d}.zstd.parquet", compression="zstd",
                   compression_level=1, row_group_size=row_group_size)
    return len(row)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("m", type=int)
    ap.add_argument("files", type=int)
    ap.add_argument("--threads", type=int, default=0, help="reader threads (0 = visible CPUs)")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--row-group-size", type=int, default=0,
                    help="rows per Parquet row group (0 = one group per file, pyarrow's default "
                         "below 1 Mi rows...): fewer files than reader threads are then decoded "
                         "one task per row group")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cpus = len(os.sched_getaffinity(0))
    threads = a.threads or cpus
    d = Path(tempfile.mkdtemp(prefix="cuking_cli_"))
    (d / "in").mkdir()
    (d / "in" / "metadata.json").write_text(json.dumps(
        {"num_sites": a.m, "samples": [f"S{k:07d}" for k in range(a.n)]}))
    bounds = np.linspace(0, a.m, a.files + 1).astype(int)
    t0 = time.perf_counter()
    jobs = [(str(d / "in"), f, int(bounds[f]), int(bounds[f + 1]), a.n, 1,
             a.row_group_size or None) for f in range(a.files)]
    triples = 0
    with ProcessPoolExecutor(max(1, min(cpus, 16))) as ex:
        for k, count in enumerate(ex.map(write_part, jobs)):
            triples += count
            if (k + 1) % 8 == 0:
                print(f"  generated {k + 1}/{a.files} files, {time.perf_counter() - t0:.0f}s",
                      flush=True)
    size = sum(p.stat().st_size for p in (d / "in").glob("*.parquet"))
    report = {"samples": a.n, "sites": a.m, "files": a.files, "triples": triples,
              "parquet_MB": size / 1e6, "generate_s": time.perf_counter() - t0,
              "cpus_visible": cpus, "reader_threads": threads, "runs": []}
    print(f"wrote {a.n}x{a.m}: {triples} triples, {size / 1e6:.0f} MB zstd parquet in "
          f"{a.files} files, {report['generate_s']:.1f}s on {cpus} CPUs", flush=True)
    for rep in range(a.repeat):
        for pack in ("host", "device", "auto"):
            t0 = time.perf_counter()
            p = subprocess.run([str(ROOT / "cuking_amd/bin/cuking"), "--input_uri", str(d / "in"),
                                "--output_uri", str(d / f"out_{pack}"), f"--pack={pack}",
                                f"--num_reader_threads={threads}", "--kin_threshold=0.05"],
                               capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if p.returncode:
                print(p.stdout[-800:], p.stderr[-800:])
                raise SystemExit(f"cuking --pack={pack} failed")
            phases = dict((k.strip(), float(v)) for k, v in
                          re.findall(r"^(.*?)\.\.\.\.* ?\(([\d.]+)s\)", p.stdout, flags=re.M))
            summary = json.loads(p.stdout.strip().splitlines()[-1])
            run = {"pack": pack, "pack_chosen": summary["pack"],
                   "decode_tasks": summary.get("decode_tasks"), "wall_s": wall, "phases_s": phases,
                   "read_pack_seconds": summary["read_pack_seconds"],
                   "triples_per_second": summary["triples_per_second"],
                   "decode_thread_seconds": summary["decode_thread_seconds"],
                   "pack_thread_seconds": summary["pack_thread_seconds"],
                   "device_pack_thread_seconds": summary.get("device_pack_thread_seconds"),
                   "kernel_seconds": summary["kernel_seconds"], "results": summary["results"]}
            report["runs"].append(run)
            print(f"rep {rep} pack={pack:6s} wall {wall:6.2f}s  read+pack "
                  f"{run['read_pack_seconds']:.3f}s = {run['triples_per_second']:.3e} triples/s  "
                  f"(decode {run['decode_thread_seconds']:.2f} / pack "
                  f"{run['pack_thread_seconds']:.2f} thread-s)  kernel {run['kernel_seconds']:.3f}s  "
                  f"{run['results']} records  {run['device_pack_thread_seconds'] if pack == 'device' else ''}",
                  flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write(json.dumps(report) + "\n")
    if not a.keep:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
