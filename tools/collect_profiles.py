#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run from gpurun_out/ into
profiles/ (tracked) and refreshes profiles/hbm_traffic.json.

    collect_profiles.py <round tag, e.g. r03> <prof dir tag> <name suffix> [kernel note]

Per kernel of the trace it records rocprofv3's all-launch average AND, from the
per-dispatch durations, the median and the mean of the timed launches (the
warm-up launches of the profiled command, which are cold, left out)."""
import csv
import glob
import json
import re
import shutil
import statistics
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
import os


def newest(pattern):
    # (gpurun merges every run of a tag into the same directory: take the latest)
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


stats = newest(str(src / "trace" / "**" / "*kernel_stats.csv"))
shutil.copy(stats, dst / f"{rnd}_kernel_stats_{suffix}.csv")
shutil.copy(src / "pmc_summary.txt", dst / f"{rnd}_pmc_summary_{suffix}.txt")
(dst / f"{rnd}_bench_{suffix}.json").write_text(json.dumps(bench, indent=1) + "\n")
steps, warmup = 20, 3
if (src / "trace_steps.txt").exists():
    steps, warmup = map(int, (src / "trace_steps.txt").read_text().split())


def short_name(name):
    name = re.sub(r"^(void )?cuking::\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*\)$", "", name)


# per-dispatch durations of every cuking kernel, in launch order
per_kernel = {}
trace = newest(str(src / "trace" / "**" / "*kernel_trace.csv"))
if trace:
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        if "cuking" in r["Kernel_Name"]:
            per_kernel.setdefault(short_name(r["Kernel_Name"]), []).append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
durations = {}
# passes of the profiled command: warm-up + timed steps, and (since round 4, unless
# --reuse-layout) one more untimed + `steps` passes with the layout reused behind them
trace_bench = {}
try:
    trace_bench = json.loads((src / "trace_bench.json").read_text())
except Exception:
    pass
passes = steps + warmup + (steps + 1 if "value_layout_reused" in trace_bench else 0)
for name, ms in per_kernel.items():
    # a pass that exceeds one launch's block limit goes out as several launches
    # (configs[4]: two): durations are per PASS
    per_pass = max(1, round(len(ms) / passes))
    if per_pass > 1 and len(ms) % per_pass == 0:
        ms = [sum(ms[k:k + per_pass]) for k in range(0, len(ms), per_pass)]
    # (the warm-up launches come first, the timed ones behind them)
    timed = ms[warmup:warmup + steps] if len(ms) >= warmup + steps else ms
    durations[name] = {"launches": len(ms), "launches_per_pass": per_pass, "timed_launches": len(timed),
                       "median_ms": statistics.median(timed), "mean_timed_ms": statistics.fmean(timed),
                       "min_ms": min(ms), "max_ms": max(ms), "mean_all_ms": statistics.fmean(ms)}
(dst / f"{rnd}_kernel_durations_{suffix}.json").write_text(json.dumps(durations, indent=1) + "\n")

kernel = bench["roofline"]["kernel"]
avg_ms = calls = kname = None
for row in csv.DictReader(open(stats)):
    if kernel in row["Name"]:
        avg_ms, calls = float(row["AverageNs"]) / 1e6, int(row["Calls"])
        kname = short_name(row["Name"])
pair = durations.get(kname, {})
counters = {}
for line in (src / "pmc_summary.txt").read_text().splitlines():
    f = line.split()
    if len(f) >= 4 and f[0] == kernel and f[2] == "avg":
        counters[f[1]] = float(f[3])
traffic = (int(counters["FETCH_SIZE"] * 1024 * 2 + counters["WRITE_SIZE"] * 1024)
           if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters else None)
p = dst / "hbm_traffic.json"
table = json.loads(p.read_text()) if p.exists() else {}
ref_ms = pair.get("median_ms") or avg_ms
entry = {
    "round": rnd, "kernel": (kname or kernel) + (f" ({note})" if note else ""),
    "FETCH_SIZE_KB_avg": counters.get("FETCH_SIZE"), "WRITE_SIZE_KB_avg": counters.get("WRITE_SIZE"),
    "correction": "gfx950: FETCH_SIZE reports 1/2 of wide (16 B/lane) coalesced reads "
                  "(global_load_lds_dwordx4 here), so reads are doubled; WRITE_SIZE exact "
                  "(MI355X_MICROARCH.md, HBM section). Separate --pmc passes "
                  "(tools/profile_round.sh).",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": cfg["pairs"] * bench["roofline"]["hbm"]["algorithmic_bytes_per_pair"]
    if "hbm" in bench["roofline"] else None,
    "source": f"profiles/{rnd}_pmc_summary_{suffix}.txt",
    "rocprof_median_ms": pair.get("median_ms"), "rocprof_mean_timed_ms": pair.get("mean_timed_ms"),
    "rocprof_avg_ms": avg_ms, "rocprof_calls": calls,
    "rocprof_source": f"profiles/{rnd}_kernel_durations_{suffix}.json, {rnd}_kernel_stats_{suffix}.csv "
                      f"(rocprofv3 --kernel-trace --stats of bench.py --steps {steps} --warmup "
                      f"{warmup}: median / mean over the {steps} timed launches; rocprof_avg_ms "
                      f"is rocprofv3's own average over all {calls} launches incl. the cold ones)",
    "hip_event_ms_same_box": bench["roofline"]["kernel_ms"],
    "frac_same_box": bench["roofline"].get("frac"),
    "mfma_busy_fraction": (counters.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024) /
                          (counters["GRBM_GUI_ACTIVE"] / 8) if "GRBM_GUI_ACTIVE" in counters else None,
    # (counters are per LAUNCH, durations per pass)
    "effective_clock_mhz_pmc": counters["GRBM_GUI_ACTIVE"] * pair.get("launches_per_pass", 1) / 8 /
                               (ref_ms * 1e-3) / 1e6
    if "GRBM_GUI_ACTIVE" in counters and ref_ms else None,
    "effective_clock_note": "GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the kernel's median "
                            "duration: the chip-wide clock under this kernel (MI355X_MICROARCH.md, "
                            "DVFS give-back)",
}
if suffix.endswith("full"):
    key += ":full"
table[f"{key}:{kernel}"] = entry
# the layout-conversion kernel of the same trace against the HBM roofline: it reads
# the block's bitset once and writes the kernel layout once
wps = bench["roofline"].get("hbm", {}).get("algorithmic_bytes_per_pair", 0) // 16
for name, d in durations.items():
    if name.startswith("sample_stats_kernel"):
        read = cfg["samples"] * wps * 8
        gbps = read / (d["median_ms"] * 1e-3) / 1e9
        table[f"{key}:{name}"] = {
            "round": rnd, "kernel": name, "bound": "hbm", "median_ms": d["median_ms"],
            "launches": d["launches"], "bytes_read": read, "bytes_written": 0,
            "achieved_GBps": gbps, "peak_GBps": 8000.0, "frac": gbps / 8000.0,
            "source": f"profiles/{rnd}_kernel_durations_{suffix}.json"}
    if name.startswith("prepare_"):
        read = cfg["samples"] * wps * 8
        # (bits per site and sample: the reference layout 2; nibble layout 4 + the het-only
        #  copy 1, and with <true> the two-bit T2 layout as well; word layout 4; quad layout 2)
        if "nibbles_kernel<true, false>" in name and d["median_ms"] < 0.2:
            continue     # (the gated conversion of the lazy codes: its workgroups left at once)
        write = (read * 7 // 2 if ("nibbles_kernel<true>" in name or "<true, true>" in name) else
                 read if "<false, true>" in name else       # T2 alone: 2 bits in, 2 bits out
                 read * 5 // 2 if "nibbles" in name else
                 read * (2 if "planes" in name else 1))
        gbps = (read + write) / (d["median_ms"] * 1e-3) / 1e9
        table[f"{key}:{name}"] = {
            "round": rnd, "kernel": name, "bound": "hbm", "median_ms": d["median_ms"],
            "launches": d["launches"], "bytes_read": read, "bytes_written": write,
            "achieved_GBps": gbps, "peak_GBps": 8000.0, "frac": gbps / 8000.0,
            "source": f"profiles/{rnd}_kernel_durations_{suffix}.json"}
p.write_text(json.dumps(table, indent=1) + "\n")
print(json.dumps(entry, indent=1))
