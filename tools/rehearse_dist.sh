#!/bin/bash
# GPU box (one GPU): the N > 1 code path of bench.py as ONE nccl rank (the driver's own
# launch shape) and as THREE gloo ranks sharing the GPU -- records of every form compared
# inside the runs.  usage: tools/rehearse_dist.sh
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
CUKING_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 \
  timeout -k 10 500 python bench.py --gpus 1 --steps 3 --warmup 1 --extra-configs c1 \
  > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err
echo "one nccl rank: rc=$?"
CUKING_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 \
  --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 3 --steps 3 --warmup 2 \
  --samples 30000 --sites 100000 --extra-configs c1 > gpurun_out/bench_reh3.json 2> gpurun_out/bench_reh3.err
echo "three gloo ranks on one GPU: rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/bench_dist1.json", "gpurun_out/bench_reh3.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c = d["config"]
    print(f, "n_gpus", d["n_gpus"], "value %.3e" % d["value"], "ms/step %.2f" % d["ms_per_step"],
          "speedup", c.get("speedup"), "others", list(d.get("other_configs", {})))
PY
