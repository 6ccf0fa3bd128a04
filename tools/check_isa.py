#!/usr/bin/env python3
"""Lists the matrix-core kernel's LDS-DMA loops in the assembly the last library
build left in cuking_amd/build_tmp/ and what the compiler put into them (the same
check every build runs: cuking_amd/build.py, check_mfma_loops)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from cuking_amd.build import PKG, check_mfma_loops

path = Path(sys.argv[1]) if len(sys.argv) > 1 else \
    PKG / "build_tmp" / "king_mfma-hip-amdgcn-amd-amdhsa-gfx950.s"
problems = check_mfma_loops(path, verbose=True)
for p in problems:
    print("BAD:", p)
sys.exit(1 if problems else 0)
