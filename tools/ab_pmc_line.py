#!/usr/bin/env python3
"""One line for a tools/profile_round.sh --light run: the dominant kernel's median duration
(kernel trace), the matrix pipe's busy share and the chip-wide clock (counter pass).
usage: ab_pmc_line.py <gpurun_out/prof_TAG> [kernel name part]"""
import csv
import glob
import os
import statistics
import sys

src = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "king_filter_kernel"


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


steps, warmup = map(int, open(src + "/trace_steps.txt").read().split())
rows = sorted(csv.DictReader(open(newest(src + "/trace/**/*kernel_trace.csv"))),
              key=lambda r: int(r["Start_Timestamp"]))
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows
      if kernel in r["Kernel_Name"]]
timed = ms[warmup:warmup + steps] if len(ms) >= warmup + steps else ms
med = statistics.median(timed)
acc = {}
for r in csv.DictReader(open(newest(src + "/pmc_mfma/**/*counter_collection.csv"))):
    if kernel in r["Kernel_Name"]:
        a = acc.setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
avg = {k: v[0] / v[1] for k, v in acc.items()}
busy = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (avg["GRBM_GUI_ACTIVE"] / 8)
# (the counter pass's own launches are as long as the traced ones to within a per cent)
clock = avg["GRBM_GUI_ACTIVE"] / 8 / (med * 1e-3) / 1e6
print(f"{kernel} median_ms {med:.3f} launches {len(ms)} mfma_busy {busy:.4f} clock_mhz {clock:.0f} "
      f"busy_x_clock_over_2400 {busy * clock / 2400:.4f}")
