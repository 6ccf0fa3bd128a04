cd $GRAFT_REPO_ROOT
echo "== gloo rehearsal, 3 ranks on one GPU, 30k x 100k (balance path)"
CUKING_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 3 --steps 3 --warmup 2 --samples 30000 --sites 100000 > gpurun_out/bench_reh3.json 2> gpurun_out/bench_reh3.err; echo rc=$?; tail -c 300 gpurun_out/bench_reh3.err
python - <<'PY'
import json
t=open('gpurun_out/bench_reh3.json').read().strip().splitlines()
print(len(t), "stdout lines")
d=json.loads(t[-1])
print(d['value'], d['n_gpus'], d['scaling'], d['ms_per_step'])
print(d['config']['tile_range_balance'])
print(d['config']['rank_kernel_ms_per_step'])
print({k:(v['seconds']) for k,v in d['with_broadcast'].items()})
PY
echo "== single-rank nccl c3 1 step"
CUKING_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 timeout -k 10 600 python bench.py --gpus 1 --steps 1 --warmup 1 --no-broadcast-pass > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo rc=$?
python - <<'PY'
import json
t=open('gpurun_out/bench_dist1.json').read().strip().splitlines()
print(len(t), "stdout lines")
d=json.loads(t[-1]); print(d['value'], d['config']['tile_range_balance'])
PY
