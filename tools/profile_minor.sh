#!/bin/bash
# GPU box: rocprofv3 kernel traces of the NON-dominant kernels, each against the
# roofline that bounds it -> gpurun_out/r03_minor_kernels.json
#   king_stream_kernel   (HBM / L2 streaming: 2 x words_per_sample x 8 B per pair)
#   pack_compact_kernel  (device-scope atomics per second; the CLI's --pack=device)
#   prepare_nibbles_kernel, synth_kernel (HBM) ride along in the same traces
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r03_minor
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=${STREAM_SAMPLES:-6000}; M=100000
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stream -- python3 $REPO/bench.py --kernel stream \
  --samples $N --sites $M --kin-threshold 0.05 --extra-configs none --cpu-seconds 0 --no-clock-pass \
  --steps 3 --warmup 1 > $OUT/stream_bench.json 2> $OUT/stream.err
# a 1e8-triple Parquet input (8 files of several row groups), generated on the box
python3 - <<PY
import json, sys
sys.path.insert(0, "$REPO/tools"); sys.path.insert(0, "$REPO")
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path
import numpy as np
import cli_timing
d = Path("$OUT/in"); d.mkdir(parents=True, exist_ok=True)
n, m, files = 2000, 50000, 8
(d / "metadata.json").write_text(json.dumps({"num_sites": m, "samples": [f"S{k:07d}" for k in range(n)]}))
b = np.linspace(0, m, files + 1).astype(int)
with ProcessPoolExecutor(8) as ex:
    t = sum(ex.map(cli_timing.write_part, [(str(d), f, int(b[f]), int(b[f + 1]), n, 1, 2000000) for f in range(files)]))
Path("$OUT/triples.txt").write_text(str(t))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pack -- $REPO/cuking_amd/bin/cuking \
  --input_uri $OUT/in --output_uri $OUT/out --pack=device --num_reader_threads=16 --kin_threshold=0.05 \
  > $OUT/pack_cli.txt 2> $OUT/pack.err
python3 - <<PY
import csv, glob, json, re
out = {}
def stats(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {re.sub(r"\(.*", "", re.sub(r"^(void )?cuking::\(anonymous namespace\)::", "", r["Name"])): r
            for r in csv.DictReader(open(f))}
s = stats("$OUT/stream")
b = json.load(open("$OUT/stream_bench.json"))
k = s["king_stream_kernel"]
pairs, bpp = b["config"]["pairs"], b["roofline"]["algorithmic_bytes_per_pair"]
avg_ms = float(k["AverageNs"]) / 1e6
out["king_stream_kernel"] = {"workload": b["config"]["workload"], "bound": "hbm (L2 / Infinity Cache streaming)",
    "launches": int(k["Calls"]), "rocprof_avg_ms": avg_ms, "hip_event_ms": b["roofline"]["kernel_ms"],
    "algorithmic_bytes_per_launch": pairs * bpp, "achieved_GBps": pairs * bpp / (avg_ms * 1e-3) / 1e9,
    "peak_GBps": 8000.0, "frac": pairs * bpp / (avg_ms * 1e-3) / 1e9 / 8000.0,
    "pairs_per_second": pairs / (avg_ms * 1e-3)}
p = stats("$OUT/pack")
triples = int(open("$OUT/triples.txt").read())
k = p["pack_compact_kernel"]
total_ms = float(k["TotalDurationNs"]) / 1e6
cli = json.loads(open("$OUT/pack_cli.txt").read().strip().splitlines()[-1])
# hom-ref clears two bits (two atomics), het / hom-alt one: ~1.55 per genotype at these frequencies
out["pack_compact_kernel"] = {"workload": "2000 samples x 50000 sites as Parquet (%d triples), cuking --pack=device, 16 reader threads" % triples,
    "bound": "device-scope atomics", "launches": int(k["Calls"]), "total_kernel_ms": total_ms,
    "triples_per_second_of_kernel_time": triples / (total_ms * 1e-3),
    "note": "kernel time summed over the reader threads' streams (they overlap): a lower bound on the rate the atomic units sustain",
    "cli_read_pack_seconds": cli["read_pack_seconds"], "cli_triples_per_second": cli["triples_per_second"]}
for name in ("prepare_nibbles_kernel", "synth_kernel"):
    for tag, st in (("stream", s), ("pack", p)):
        if name in st:
            out.setdefault(name, {})[tag] = {"launches": int(st[name]["Calls"]), "avg_ms": float(st[name]["AverageNs"]) / 1e6}
json.dump(out, open("$REPO/gpurun_out/r03_minor_kernels.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
