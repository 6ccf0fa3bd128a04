#!/bin/bash
# GPU box: per-group cycles of the four-product loop (diagnostic build with
# s_memtime stamps; the stamps perturb: each drains the LDS queue).  Whole-tile
# launches only (the stamps go to the split scratch of a non-split launch).
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS" python -c "from cuking_amd import build; build.build_library(force=True)" 2>&1 | tail -3
for args in "--samples 24000 --sites 100000" "--config c2"; do
  echo "== $args"
  CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS" python bench.py $args --extra-configs none --cpu-seconds 0 \
    --no-clock-pass --steps 2 --warmup 1 2> gpurun_out/stamps.err > gpurun_out/stamps.json || true
  grep "mfma4 stamps" gpurun_out/stamps.err | tail -4 || tail -5 gpurun_out/stamps.err
  python -c "
import json; d=json.load(open('gpurun_out/stamps.json')); print('kernel_ms', d['roofline']['kernel_ms'])" || true
done | tee gpurun_out/r03_stamps_n4.txt
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
