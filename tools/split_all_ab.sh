#!/bin/bash
# Tuning-build experiment: every launch as one persistent stream-k launch
# (CUKING_MFMA_SPLIT_ALL=1) against the shipped policy, same box, interleaved.
set -eo pipefail
python -m cuking_amd.build --lib --tuning > /dev/null 2>&1
for n in 3000 10000 20000; do
  for rep in 1 2; do
    for mode in "" 1; do
      if [ -n "$mode" ]; then export CUKING_MFMA_SPLIT_ALL=1; else unset CUKING_MFMA_SPLIT_ALL; fi
      python bench.py --steps 8 --warmup 3 --cpu-seconds 0 --samples $n > gpurun_out/sa.log 2>&1
      echo "[samples $n split_all=${mode:-0}]"; python tools/jl.py gpurun_out/sa.log
    done
  done
done
