#!/bin/bash
# Timing-only ablations of the matrix-core kernel (tuning build; results are
# wrong by construction, hence --no-check): what LDS-DMA, the stage barrier
# and the epilogue cost.  Same box, interleaved.  Rebuild the library after.
set -eo pipefail
python -m cuking_amd.build --lib --tuning > /dev/null 2>&1
for rep in 1 2; do
  for abl in 0 1 2 3; do
    CUKING_MFMA_ABLATE=$abl python bench.py --steps 8 --warmup 3 --cpu-seconds 0 --no-check "$@" > gpurun_out/abl.log 2>&1
    echo "[ablate $abl]"; python tools/jl.py gpurun_out/abl.log
  done
done
