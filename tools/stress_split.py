#!/usr/bin/env python3
"""Repeats one staged configuration many times per (variant, form, split,
streams) combination and counts mismatches against the oracle: rare,
timing-dependent failures (tests/fuzz_cases.py run_stress).
usage: stress_split.py [reps] [threshold]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import cuking_amd
import fuzz_cases

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
rows = fuzz_cases.run_stress(cuking_amd.KingContext(0), reps, thr, log=lambda m: print(m, flush=True))
sys.exit(1 if any(bad for _, bad, _, _ in rows) else 0)
