#!/usr/bin/env python3
"""Per-kernel average of rocprofv3 --pmc counters. usage: pmc_summary.py <dir>..."""
import csv
import glob
import re
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(\w+_kernel)\b", name)
            short = m.group(1) if m else name[:40]
            k = (short, r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
        for (kn, cn), (tot, n) in sorted(acc.items()):
            if "elementwise" in kn or "copyBuffer" in kn:
                continue
            print(f"{kn:42s} {cn:24s} avg {tot / n:16.1f}  n={n}")
