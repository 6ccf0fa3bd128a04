#!/usr/bin/env python3
"""End-to-end timing of the `cuking` binary on real Parquet input (host vs
device pack), phases as printed by the binary.  usage: cli_timing.py N M files"""
import re
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from cuking_amd.inputs import write_input_tables  # noqa: E402

n, m, files = (int(x) for x in sys.argv[1:4])
rng = np.random.default_rng(1)
af = rng.uniform(0.05, 0.5, size=m)
geno = (rng.random((n, m)) < af).astype(np.int8) + (rng.random((n, m)) < af).astype(np.int8)
geno[rng.random((n, m)) < 0.01] = -1
geno[n - 1] = geno[0]
d = Path(tempfile.mkdtemp(prefix="cuking_cli_"))
t0 = time.perf_counter()
write_input_tables(d / "in", geno, num_files=files)
size = sum(p.stat().st_size for p in (d / "in").glob("*.parquet"))
print(f"wrote {n}x{m}: {int((geno >= 0).sum())} triples, {size / 1e6:.0f} MB parquet, "
      f"{time.perf_counter() - t0:.1f}s", flush=True)
for pack in ("host", "device"):
    for threads in (16,):
        t0 = time.perf_counter()
        p = subprocess.run([str(ROOT / "cuking_amd/bin/cuking"), "--input_uri", str(d / "in"),
                            "--output_uri", str(d / f"out_{pack}"), f"--pack={pack}",
                            f"--num_reader_threads={threads}", "--kin_threshold=0.05"],
                           capture_output=True, text=True)
        wall = time.perf_counter() - t0
        phases = re.findall(r"^(.*?)\.\.\.\.* ?\(([\d.]+)s\)", p.stdout, flags=re.M)
        print(f"pack={pack} threads={threads} rc={p.returncode} wall={wall:.2f}s  " +
              "; ".join(f"{a.strip()[:28]}={b}s" for a, b in phases), flush=True)
        print("   ", p.stdout.strip().splitlines()[-1])
        if p.returncode:
            print(p.stderr[-500:])
