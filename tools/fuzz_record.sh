#!/bin/bash
# GPU box: the three fuzzers in bulk against the shipped library; one record file.
# usage: GIT_HEAD=<hash> [FUZZ_SEEDS="1 2 3 4"] [FUZZ_CASES=2500] [SPLIT_SEEDS="1 2"] [STRESS_REPS=300]
#        [FUZZ_OUT=gpurun_out/r03_fuzz.txt] tools/fuzz_record.sh
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=${FUZZ_OUT:-gpurun_out/r04_fuzz.txt}
{
  echo "# Bulk fuzz record.  HEAD ${GIT_HEAD:-unknown}; library built by cuking_amd/build.py"
  echo "# (default flags: $(cat cuking_amd/libcuking_amd.flags 2>/dev/null | tr -d '\n')); $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "# checker: oracle/pyoracle.py; every case through the C ABI; cases = tests/fuzz_cases.py"
  sha256sum cuking_amd/csrc/king_common.h tests/fuzz_cases.py cuking_amd/csrc/king_filter.hip cuking_amd/csrc/king_mfma.hip cuking_amd/csrc/king_kernels.hip cuking_amd/csrc/king_abi.hip cuking_amd/csrc/king_device.h cuking_amd/csrc/king_sort.hip
} > $OUT
# (the exit code of the first failing fuzzer is what the record ends with: under `set -e`
#  a bare `echo rc=$?` would only ever be reached with 0)
rc=0
for seed in ${FUZZ_SEEDS:-1 2 3 4}; do python tools/fuzz_gpu.py $seed ${FUZZ_CASES:-2500} 2>&1 | tail -1 | tee -a $OUT || rc=$?; done
for seed in ${SPLIT_SEEDS:-1 2}; do python tools/fuzz_split.py $seed 400 2>&1 | tail -1 | tee -a $OUT || rc=$?; done
python tools/stress_split.py ${STRESS_REPS:-300} 2>&1 | tee -a $OUT | tail -3 || rc=$?
echo "rc=$rc" >> $OUT
exit $rc
