#!/bin/bash
# Tuning-build experiment: persistent workgroups striding over whole tiles
# (CUKING_MFMA_PERSIST=1) against one workgroup per tile, same box, interleaved.
set -eo pipefail
python -m cuking_amd.build --lib --tuning > /dev/null 2>&1
for n in 10000 20000; do
  for rep in 1 2; do
    for mode in "" 1; do
      if [ -n "$mode" ]; then export CUKING_MFMA_PERSIST=1; else unset CUKING_MFMA_PERSIST; fi
      python bench.py --steps 8 --warmup 3 --cpu-seconds 0 --samples $n > gpurun_out/pa.log 2>&1
      echo "[samples $n persist=${mode:-0}]"; python tools/jl.py gpurun_out/pa.log
    done
  done
done
