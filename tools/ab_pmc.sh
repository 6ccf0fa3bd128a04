#!/bin/bash
# GPU box: library builds that differ in -D flags, each with a configs[2] timing, a kernel
# trace and the matrix-pipe counter pass (busy share, chip-wide clock): what a timing-only
# ablation changes -- the pipe's busy share or the clock the chip holds.
# usage: tools/ab_pmc.sh "<flags A>" "<flags B>" ...   ("" = the shipped build)
# BENCH_EXTRA: further bench.py arguments for every run.
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/ab_pmc.txt
: > $OUT
i=0
for flags in "$@"; do
  i=$((i + 1))
  export CUKING_EXTRA_HIPFLAGS="$flags"
  python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
  TRACE_STEPS=6 TRACE_WARMUP=2 bash tools/profile_round.sh ab$i --light ${BENCH_EXTRA:-} > gpurun_out/ab_pmc_$i.log 2>&1
  echo "[$flags] $(python tools/ab_pmc_line.py gpurun_out/prof_ab$i)" | tee -a $OUT
done
unset CUKING_EXTRA_HIPFLAGS
python -c "from cuking_amd import build; build.build_library(force=True)" > /dev/null 2>&1
