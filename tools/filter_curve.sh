#!/bin/bash
# GPU box: the filter variant (7) against the four-product kernel (6) as the threshold
# falls towards the noise of unrelated pairs and the bound lets more and more pairs
# through (lean form forced): kernel time per pass, candidates, dense quadrants.
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
OUT=gpurun_out/filter_curve.txt
: > $OUT
for thr in 0.0884 0.05 0.03 0.02 0.015 0.012 0.01 0.008; do
  for v in 7 6; do
    python bench.py --config ${CFG:-c1} --extra-configs none --cpu-seconds 0 --no-clock-pass --no-check \
      --steps 5 --warmup 1 --variant $v --counts-mode 0 --kin-threshold $thr --max-results 16777216 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f=r.get('filter') or {}
print('thr $thr variant $v kernel_ms %.3f records %d candidates %s dense_quadrants %s' % (r['kernel_ms'], d['config']['results_per_step'], f.get('candidates_per_pass'), f.get('dense_quadrants_per_pass')))" | tee -a $OUT
  done
done
