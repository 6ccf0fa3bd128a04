#!/bin/bash
# GPU box: A/B of library builds that differ in -D flags, same box, c1 + c2.
# usage: tools/ab_flags.sh "<flags A>" "<flags B>" ...   ("" = the shipped build)
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/ab_flags.txt
: > $OUT
for flags in "$@"; do
  CUKING_EXTRA_HIPFLAGS="$flags" python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
  for cfg in c1 c2; do
    steps=20; [ $cfg = c2 ] && steps=4
    CUKING_EXTRA_HIPFLAGS="$flags" python bench.py --config $cfg --extra-configs none --cpu-seconds 0 --no-clock-pass \
      --steps $steps --warmup 2 ${BENCH_EXTRA:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$flags]', '$cfg', 'kernel_ms %.3f' % d['roofline']['kernel_ms'])" | tee -a $OUT
  done
done
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
