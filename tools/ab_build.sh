#!/bin/bash
# A/B of compile-time kernel options on the GPU box (same device, interleaved):
#   tools/ab_build.sh "<flags A>" "<flags B>" [bench args...]
set -eo pipefail
A=$1; B=$2; shift 2
for rep in 1 2; do
  for cfg in "$A" "$B"; do
    CUKING_EXTRA_HIPFLAGS="$cfg" python -m cuking_amd.build --lib --force > /dev/null 2>&1
    python bench.py --steps 8 --warmup 3 --cpu-seconds 0 "$@" > gpurun_out/ab.log 2>&1
    echo "[$cfg]"; python tools/jl.py gpurun_out/ab.log
  done
done
