#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep of the matrix-core variants' remainder
splitting, in bulk (tests/fuzz_cases.py run_split; a fixed-seed sample runs in
`pytest -m gpu`).
usage: fuzz_split.py [seed] [cases] [first_case]   (first_case: replay one failure)"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
import fuzz_cases

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = time.time()
ran = fuzz_cases.run_split(cuking_amd.KingContext(0), seed, cases, first,
                           log=lambda m: print(m, flush=True))
print(f"fuzz_split seed {seed}: {ran} cases OK in {time.time() - t0:.0f}s", flush=True)
