#!/bin/bash
# GPU box: band height / XCD order sweep of a matrix-core variant at configs[2].
# usage: tools/sweep_order.sh [variant] [config]
set -eo pipefail
V=${1:-6}; CFG=${2:-c2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/sweep_order_v${V}_$CFG.txt
: > $OUT
one() {
  python bench.py --variant $V --config $CFG --extra-configs none --cpu-seconds 0 --no-clock-pass \
    --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', 'kernel_ms %.2f' % d['roofline']['kernel_ms'])" | tee -a $OUT
}
one --band-rows 5
for b in 2 3 4 6 8 11 17; do one --band-rows $b; done
one --band-rows 5 --xcd-swizzle 1
one --band-rows 5 --xcd-swizzle 0
one --band-rows 5
