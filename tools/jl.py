#!/usr/bin/env python3
"""Prints the key figures of bench.py JSON lines: jl.py <log>..."""
import json
import sys
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
            r = d.get("roofline", {})
            print(path, "value %.4e" % d["value"], "ms/step %.3f" % d["ms_per_step"],
                  "kernel_ms %.3f" % r.get("kernel_ms", 0), "launches", r.get("launches"),
                  "n_gpus", d["n_gpus"], d["config"]["workload"])
