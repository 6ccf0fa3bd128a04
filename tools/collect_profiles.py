#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run from gpurun_out/ into
profiles/ (tracked) and refreshes profiles/hbm_traffic.json.

    collect_profiles.py <round tag, e.g. r02> <prof dir tag> <name suffix> [kernel note]
"""
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
rnd, tag, suffix = sys.argv[1:4]
note = sys.argv[4] if len(sys.argv) > 4 else ""
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
bench = json.loads((src / "bench.json").read_text())
cfg = bench["config"]
key = f"{cfg['samples']}x{cfg['sites']}"
stats = glob.glob(str(src / "trace" / "**" / "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, dst / f"{rnd}_kernel_stats_{suffix}.csv")
shutil.copy(src / "pmc_summary.txt", dst / f"{rnd}_pmc_summary_{suffix}.txt")
(dst / f"{rnd}_bench_{suffix}.json").write_text(json.dumps(bench, indent=1) + "\n")

kernel = bench["roofline"]["kernel"]
avg_ms = calls = None
for row in csv.DictReader(open(stats)):
    if kernel in row["Name"]:
        avg_ms, calls = float(row["AverageNs"]) / 1e6, int(row["Calls"])
        kname = re.sub(r"^void cuking::\(anonymous namespace\)::|\(cuking::TiledArgs\)$", "",
                       row["Name"])
counters = {}
for line in (src / "pmc_summary.txt").read_text().splitlines():
    f = line.split()
    if len(f) >= 4 and f[0] == kernel and f[2] == "avg":
        counters[f[1]] = float(f[3])
traffic = int(counters["FETCH_SIZE"] * 1024 * 2 + counters["WRITE_SIZE"] * 1024)
p = dst / "hbm_traffic.json"
table = json.loads(p.read_text()) if p.exists() else {}
entry = {
    "round": rnd, "kernel": kname + (f" ({note})" if note else ""),
    "FETCH_SIZE_KB_avg": counters["FETCH_SIZE"], "WRITE_SIZE_KB_avg": counters["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE reports 1/2 of wide (16 B/lane) coalesced reads "
                  "(global_load_lds_dwordx4 here), so reads are doubled; WRITE_SIZE exact "
                  "(MI355X_MICROARCH.md, HBM section). Separate --pmc passes "
                  "(tools/profile_round.sh).",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": cfg["pairs"] * bench["roofline"]["hbm"]["algorithmic_bytes_per_pair"]
    if "hbm" in bench["roofline"] else None,
    "source": f"profiles/{rnd}_pmc_summary_{suffix}.txt",
    "rocprof_avg_ms": avg_ms, "rocprof_calls": calls,
    "rocprof_source": f"profiles/{rnd}_kernel_stats_{suffix}.csv (rocprofv3 --kernel-trace --stats "
                      f"of the same bench.py command, {calls} launches incl. warm-up)",
    "hip_event_ms_same_box": bench["roofline"]["kernel_ms"],
    "mfma_busy_fraction": (counters.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024) /
                          (counters["GRBM_GUI_ACTIVE"] / 8) if "GRBM_GUI_ACTIVE" in counters else None,
    "effective_clock_mhz_pmc": counters["GRBM_GUI_ACTIVE"] / 8 / (avg_ms * 1e-3) / 1e6
    if "GRBM_GUI_ACTIVE" in counters else None,
}
if suffix.endswith("full"):
    key += ":full"
table[f"{key}:{kernel}"] = entry
p.write_text(json.dumps(table, indent=1) + "\n")
print(json.dumps(entry, indent=1))
