#!/usr/bin/env python3
"""GPU box: the real-input leg of `cuking` (decode + pack, cuking.cu:547-711) in its four
forms -- host pack / pipelined device pack x streaming decode / whole tables -- on one
real-Parquet input sized to the hardware threads the box shows:

    pack_modes.py [samples] [sites] [files]     (default 2000 x 50000 in 8 files = 1e8
                                                 triples from 8 threads on; 1/8 of the
                                                 sites per visible thread below that)

Prints one line per (pack, decode, repetition): read+pack seconds, triples/s, the
thread-seconds spent decoding and packing, and checks that all four forms write the same
output file.  -> profiles/r04_pack_pipeline.txt
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import cli_timing  # noqa: E402


def main():
    cpus = len(os.sched_getaffinity(0))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else (50_000 if cpus >= 8 else 6_250 * cpus)
    files = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    threads = max(1, min(16, cpus))
    d = Path(tempfile.mkdtemp(prefix="cuking_packmodes_"))
    try:
        (d / "in").mkdir()
        (d / "in" / "metadata.json").write_text(json.dumps(
            {"num_sites": m, "samples": [f"S{k:07d}" for k in range(n)]}))
        bounds = np.linspace(0, m, files + 1).astype(int)
        jobs = [(str(d / "in"), f, int(bounds[f]), int(bounds[f + 1]), n, 1, 2_000_000)
                for f in range(files)]
        t0 = time.perf_counter()
        with ProcessPoolExecutor(max(1, min(8, cpus))) as ex:
            triples = sum(ex.map(cli_timing.write_part, jobs))
        size = sum(p.stat().st_size for p in (d / "in").glob("*.parquet"))
        print(f"# {n} samples x {m} sites: {triples} triples, {size / 1e6:.0f} MB of zstd Parquet in "
              f"{files} files (several row groups each), generated in {time.perf_counter() - t0:.1f} s; "
              f"{cpus} hardware threads visible, --num_reader_threads={threads}", flush=True)
        outputs = {}
        best = {}
        for rep in range(3):
            for pack in ("host", "device"):
                for decode in ("stream", "table"):
                    out = d / f"out_{pack}_{decode}"
                    t0 = time.perf_counter()
                    p = subprocess.run([str(ROOT / "cuking_amd/bin/cuking"), "--input_uri",
                                        str(d / "in"), "--output_uri", str(out), f"--pack={pack}",
                                        f"--decode={decode}", f"--num_reader_threads={threads}",
                                        "--kin_threshold=0.05"], capture_output=True, text=True)
                    wall = time.perf_counter() - t0
                    if p.returncode:
                        raise SystemExit(f"cuking --pack={pack} --decode={decode} failed: {p.stderr[-800:]}")
                    s = json.loads(p.stdout.strip().splitlines()[-1])
                    outputs[(pack, decode)] = (out / "part-00000.snappy.parquet").read_bytes()
                    key = (pack, decode)
                    if key not in best or s["read_pack_seconds"] < best[key]:
                        best[key] = s["read_pack_seconds"]
                    print(f"rep {rep} pack={pack:6s} decode={decode:6s} wall {wall:6.2f} s  read+pack "
                          f"{s['read_pack_seconds']:.3f} s = {s['triples_per_second']:.3e} triples/s  "
                          f"decode {s['decode_thread_seconds']:.2f} / pack {s['pack_thread_seconds']:.2f} "
                          f"thread-s  tasks {s['decode_tasks']}  kernel {s['kernel_seconds']:.3f} s",
                          flush=True)
        same = len(set(outputs.values())) == 1
        print(f"# output files of the four forms identical: {same}")
        for key in sorted(best):
            print(f"# best read+pack, pack={key[0]} decode={key[1]}: {best[key]:.3f} s = "
                  f"{triples / best[key]:.3e} triples/s")
        if not same:
            raise SystemExit("outputs differ")
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
