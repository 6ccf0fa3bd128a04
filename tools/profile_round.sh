#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + PMC passes.
# Usage: tools/profile_round.sh <tag> [bench args...]
set -eo pipefail
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 5 --warmup 1 --cpu-seconds 0 "$@" > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2> $OUT/pmc_sq2.err
find $OUT -name "*.csv" | head -50
