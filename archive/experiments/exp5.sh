cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
echo "== single-rank nccl, default config (c3) 2 steps"
CUKING_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 600 python bench.py --gpus 1 --steps 2 --warmup 1 > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo rc=$?; tail -c 600 gpurun_out/bench_dist1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_dist1.json'))
print(d['value'], d['scaling'], d['config']['workload'], d['ms_per_step'])
print({k:d['config'][k] for k in ('rccl_ranks','backend','rank_kernel_ms_per_step','gather_ms_unpipelined')})
print(d['with_broadcast'])
print(d['roofline']['frac'], d['roofline']['kernel_ms'])
PY
echo "== gloo rehearsal, 3 ranks on one GPU, 30k x 100k"
CUKING_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 3 --steps 2 --warmup 1 --samples 30000 --sites 100000 > gpurun_out/bench_reh3.json 2> gpurun_out/bench_reh3.err; echo rc=$?; tail -c 800 gpurun_out/bench_reh3.err
python - <<'PY'
import json
t=open('gpurun_out/bench_reh3.json').read().strip().splitlines()[-1]
d=json.loads(t)
print(d['value'], d['n_gpus'], d['scaling'], d['config']['workload'], d['ms_per_step'])
print({k:d['config'][k] for k in ('rccl_ranks','backend','rank_kernel_ms_per_step','gather_ms_unpipelined')})
print(d['with_broadcast'])
PY
