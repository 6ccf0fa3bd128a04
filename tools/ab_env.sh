#!/bin/bash
# GPU box: A/B of run-time settings (environment variables of the library), same
# box, same build.  usage: tools/ab_env.sh "<VAR=VAL ...>" "<VAR=VAL ...>" ...   ("" = defaults)
# CONFIGS="c1 c2" (default) chooses the workloads.
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/ab_env.txt
: > $OUT
for spec in "$@"; do
  for cfg in ${CONFIGS:-c1 c2}; do
    steps=20; [ $cfg = c2 ] && steps=6
    env $spec python bench.py --config $cfg --extra-configs none --cpu-seconds 0 --no-clock-pass \
      --steps $steps --warmup 2 ${BENCH_EXTRA:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$spec]', '$cfg', 'kernel_ms %.3f' % d['roofline']['kernel_ms'])" | tee -a $OUT
  done
done
