#!/usr/bin/env python3
"""Summarises hipcc -Rpass-analysis=kernel-resource-usage remarks from stdin."""
import re
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                     ("sgpr", r"TotalSGPRs: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("spill", r"VGPRs Spill: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
print(f"{'kernel':70s} vgpr sgpr spill scratch occ   lds")
for r in rows:
    name = re.sub(r"^_ZN6cuking12_GLOBAL__N_1\d+", "", r["name"])[:70]
    print(f"{name:70s} {r.get('vgpr',0):4d} {r.get('sgpr',0):4d} "
          f"{r.get('spill',0):5d} {r.get('scratch',0):7d} {r.get('occ',0):3d} "
          f"{r.get('lds',0):5d}")
