#!/bin/bash
# GPU box: the timing build of the filter kernel (king_filter.hip, CUKING_FILTER_TIMING) under
# tools/tile_gaps.py at configs[2] and configs[1]; the shipped build is restored at the end.
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
: > gpurun_out/tile_gaps.txt
export CUKING_EXTRA_HIPFLAGS="-DCUKING_FILTER_TIMING=1"
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
python tools/tile_gaps.py 100000 100000 0.0884
python tools/tile_gaps.py 10000 100000 0.05
unset CUKING_EXTRA_HIPFLAGS
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
