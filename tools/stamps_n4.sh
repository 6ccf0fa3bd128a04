#!/bin/bash
# GPU box: per-group cycles of the four-product loop (diagnostic build with
# s_memtime stamps; the stamps perturb: each drains the LDS queue), c1 and c2.
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS" python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
for cfg in c1 c2; do
  extra=""; [ $cfg = c1 ] && extra="--split-wgs 0"
  echo "== $cfg $extra"
  CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS" python bench.py --config $cfg $extra --extra-configs none --cpu-seconds 0 \
    --no-clock-pass --steps 2 --warmup 1 2>&1 >/dev/null | grep "mfma stamps" | tail -2
done | tee gpurun_out/r03_stamps_n4.txt
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
