#!/bin/bash
# GPU box: the three fuzzers in bulk against the shipped library; one record file.
# usage: GIT_HEAD=<hash> tools/fuzz_record.sh  -> gpurun_out/r03_fuzz.txt
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/r03_fuzz.txt
{
  echo "# Bulk fuzz record, round 3.  HEAD ${GIT_HEAD:-unknown}; library built by cuking_amd/build.py"
  echo "# (default flags: $(cat cuking_amd/libcuking_amd.flags 2>/dev/null | tr -d '\n')); $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "# checker: oracle/pyoracle.py; every case through the C ABI; cases = tests/fuzz_cases.py"
  sha256sum cuking_amd/csrc/king_filter.hip cuking_amd/csrc/king_mfma.hip cuking_amd/csrc/king_kernels.hip cuking_amd/csrc/king_abi.hip cuking_amd/csrc/king_device.h
} > $OUT
for seed in 1 2 3 4; do python tools/fuzz_gpu.py $seed 2500 2>&1 | tail -1 | tee -a $OUT; done
for seed in 1 2; do python tools/fuzz_split.py $seed 400 2>&1 | tail -1 | tee -a $OUT; done
python tools/stress_split.py 300 2>&1 | tee -a $OUT | tail -3
echo "rc=$?" >> $OUT
