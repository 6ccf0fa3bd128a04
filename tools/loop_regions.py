#!/usr/bin/env python3
"""Per scheduling region (between sched_barrier(0)s) of the hottest loop of each
king_mfma_kernel instantiation: MFMAs, VALU ANDs / bitops, LDS reads, LDS-DMA
requests, waits.  usage: loop_regions.py [substring of the mangled name]"""
import re
import sys
from pathlib import Path
asm = Path(__file__).resolve().parent.parent / "cuking_amd/build_tmp/king_mfma-hip-amdgcn-amd-amdhsa-gfx950.s"
want = sys.argv[1] if len(sys.argv) > 1 else ""
funcs = re.split(r"\n(?=_ZN6cuking12_GLOBAL__N_116king_mfma_kernel\w+:)", asm.read_text())[1:]
for f in funcs:
    name = f.split(":")[0]
    if want not in name:
        continue
    body = f.split(".Lfunc_end")[0]
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    best = max(blocks, key=lambda b: b.count("v_mfma"))
    print(name, "loop:", best.count("v_mfma"), "MFMAs,", len(re.findall(r"\n\tv_(?!mfma)", best)), "other VALU,",
          best.count("ds_read"), "LDS reads,", best.count("global_load_lds"), "DMA,", best.count("scratch_"), "scratch")
    for i, r in enumerate(best.split("; sched_barrier mask(0x00000000)")):
        print(f"  {i:2d} mfma {r.count('v_mfma')} valu {len(re.findall(chr(10) + chr(9) + 'v_(?!mfma)', r)):3d} "
              f"ds {r.count('ds_read')} dma {r.count('global_load_lds')} salu "
              f"{len(re.findall(chr(10) + chr(9) + 's_(?!waitcnt|nop|barrier)', r)):2d} "
              f"bar {r.count('s_barrier')} waits {[w.strip() for w in re.findall(r's_waitcnt.*', r)]}")
