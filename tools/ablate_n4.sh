#!/bin/bash
# GPU box: what the LDS-DMA requests and the stage barrier of the matrix-core
# kernels cost, by leaving them out (tuning build, WRONG results, timing only).
# usage: tools/ablate_n4.sh [variant]      (writes gpurun_out/ablate_v<variant>.txt)
set -eo pipefail
V=${1:-6}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/ablate_v$V.txt
python -c "from cuking_amd import build; build.build_library(force=True, tuning=True)" > /dev/null
run() {  # run <label> <env...>
  local label=$1; shift
  for cfg in c1 c2; do
    steps=20; [ $cfg = c2 ] && steps=3
    env "$@" python bench.py --variant $V --config $cfg --extra-configs none --cpu-seconds 0 \
      --no-clock-pass --no-check --kin-threshold 10 --steps $steps --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label', '$cfg', 'kernel_ms %.3f' % d['roofline']['kernel_ms'])" | tee -a $OUT
  done
}
: > $OUT
run shipped CUKING_NOP=1
run no_dma CUKING_MFMA_ABLATE=1
run no_dma_no_barrier CUKING_MFMA_ABLATE=2
run no_epilogue CUKING_MFMA_ABLATE=3
# back to the shipped configuration
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null
