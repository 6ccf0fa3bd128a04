#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + PMC passes.
# Usage: tools/profile_round.sh <tag> [--light] [bench args...]
# Counters are collected in passes of their own (--pmc with --kernel-trace only),
# FETCH_SIZE / WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes.  The
# profiled invocations run the headline workload only (no other_configs, no
# clock pass, no CPU baseline).  --light: kernel trace + the matrix-pipe counter
# pass only (the long configs[3] / configs[4] passes).
# TRACE_STEPS / TRACE_WARMUP / PMC_STEPS override the step counts (defaults 20 / 3 / 2).
set -eo pipefail
TAG=${1:-r04}; shift || true
LIGHT=0
if [ "$1" = "--light" ]; then LIGHT=1; shift; fi
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
LEAN="--cpu-seconds 0 --extra-configs none --no-clock-pass --no-worst-case --no-h2d-pass"
TS=${TRACE_STEPS:-20}; TW=${TRACE_WARMUP:-3}; PS=${PMC_STEPS:-2}
echo "$TS $TW" > $OUT/trace_steps.txt
if [ $LIGHT = 1 ]; then
  python3 $REPO/bench.py $LEAN --steps $TS --warmup $TW "$@" > $OUT/bench.json 2> $OUT/bench.err
else
  python3 $REPO/bench.py --extra-configs none "$@" > $OUT/bench.json 2> $OUT/bench.err
fi
# per-dispatch durations (kernel_trace.csv) as well as the stats table: the first
# launches of a process are cold, so the summary reports the median and the mean
# of the timed launches beside rocprofv3's all-launch average
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps $TS --warmup $TW $LEAN "$@" > $OUT/trace_bench.json 2> $OUT/trace.err
pmc() {  # pmc <dir> <counters...>
  local d=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -- python3 $REPO/bench.py --steps $PS --warmup ${PMC_WARMUP:-1} $LEAN "${BENCH_ARGS[@]}" > /dev/null 2> $OUT/$d.err
}
BENCH_ARGS=("$@")
DIRS=""
if [ $LIGHT = 0 ]; then
  pmc pmc_fetch FETCH_SIZE
  pmc pmc_write WRITE_SIZE
  pmc pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
  pmc pmc_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE
  DIRS="$OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2"
fi
# matrix-pipe utilisation, measured: busy cycles of the MFMA pipe, MFMA
# instructions and their MOPS by operand class (fp4/fp6 = F6F4)
pmc pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F6F4 SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
python3 $REPO/tools/pmc_summary.py $DIRS $OUT/pmc_mfma > $OUT/pmc_summary.txt 2>&1 || true
find $OUT -name "*kernel_stats.csv" | head -3
tail -12 $OUT/pmc_summary.txt
