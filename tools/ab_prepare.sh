#!/bin/bash
# GPU box: conversion kernel (prepare_nibbles_kernel) A/B over -D flags: HIP-event ms
# per conversion at configs[1] and configs[2] (bench.py --convert-every-step).
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
: > gpurun_out/ab_prepare.txt
for flags in "$@"; do
  CUKING_EXTRA_HIPFLAGS="$flags" python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
  for cfg in c1 c2; do
    steps=20; [ $cfg = c2 ] && steps=4
    python bench.py --config $cfg --convert-every-step --extra-configs none --cpu-seconds 0 --no-clock-pass \
      --steps $steps --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
n,m=d['config']['samples'],d['config']['sites']
wps=2*((m+63)//64)
b=n*wps*8*(1+2.25)
ms=d['roofline']['prepare_ms']
print('[$flags]', '$cfg', 'prepare_ms %.4f = %.0f GB/s' % (ms, b/ms/1e6))" | tee -a gpurun_out/ab_prepare.txt
  done
done
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
