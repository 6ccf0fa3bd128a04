#!/bin/bash
# GPU box: configs[2] under bench.py arguments that change how tiles share the XCDs' L2 --
# per set of arguments the kernel's time (HIP events), then one counter pass: L2 hit rate
# (TCC_HIT / TCC_MISS), bytes requested from the fabric, the chip-wide clock.
# usage: tools/l2_probe.sh "<bench args A>" "<bench args B>" ...
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/l2_probe.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
LEAN="--cpu-seconds 0 --extra-configs none --no-clock-pass --no-worst-case --no-h2d-pass --reuse-layout"
i=0
for extra in "$@"; do
  i=$((i + 1))
  D=$REPO/gpurun_out/l2_probe_$i
  rm -rf $D; mkdir -p $D
  python3 $REPO/bench.py $LEAN --steps 6 --warmup 2 $extra > $D/bench.json 2> $D/bench.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/pmc -- \
    python3 $REPO/bench.py $LEAN --steps 2 --warmup 1 $extra > /dev/null 2> $D/pmc.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- \
    python3 $REPO/bench.py $LEAN --steps 2 --warmup 1 $extra > /dev/null 2> $D/fetch.err
  python3 - "$D" "$extra" <<'PY' | tee -a $OUT
import csv, glob, json, sys
d, extra = sys.argv[1], sys.argv[2]
b = json.loads(open(d + "/bench.json").read())
acc = {}
for f in glob.glob(d + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "king_filter_kernel" in r["Kernel_Name"]:
            a = acc.setdefault(r["Counter_Name"], [0.0, set()])
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
# per PASS: the launches of a pass summed (3 passes per counter run)
per_pass = {k: v[0] / 3 for k, v in acc.items()}
ms = b["roofline"]["kernel_ms"]
# the filter kernel's own launches of a pass, summed (kernel trace of the FETCH_SIZE run)
own = 0.0
for f in glob.glob(d + "/fetch/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "king_filter_kernel" in r["Kernel_Name"]:
            own += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
own /= 3
hit, miss = per_pass.get("TCC_HIT_sum", 0), per_pass.get("TCC_MISS_sum", 0)
print(f"[{extra}] kernel_ms {ms:.3f} filter_kernel_ms_under_counters {own:.3f} l2_hit_rate {hit / max(hit + miss, 1):.3f} "
      f"fetch_GB {per_pass.get('FETCH_SIZE', 0) * 1024 * 2 / 1e9:.0f} "
      f"clock_mhz {per_pass.get('GRBM_GUI_ACTIVE', 0) / 8 / (ms * 1e-3) / 1e6:.0f} "
      f"launches_per_pass {len(acc.get('FETCH_SIZE', [0, set()])[1]) / 3:.0f}")
PY
done
