#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep in bulk: shapes, shards, kernel variants,
lean/full forms, thresholds, tile ranges and staged schedules.  The cases are
tests/fuzz_cases.py run_general (a fixed-seed sample of them runs in
`pytest -m gpu`, tests/test_gpu_fuzz.py).
usage: fuzz_gpu.py [seed] [cases] [first_case]   (first_case: replay one failure)"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
import fuzz_cases

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = time.time()
ran = fuzz_cases.run_general(cuking_amd.KingContext(0), seed, cases, first,
                             log=lambda m: print(m, flush=True))
print(f"fuzz seed {seed}: {ran} cases OK in {time.time() - t0:.0f}s", flush=True)
