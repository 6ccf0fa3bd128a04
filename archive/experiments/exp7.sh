cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/fuzz_gpu.py 21 400 2>&1 | tail -3
timeout -k 10 300 python tools/fuzz_split.py 2>&1 | tail -3
echo "== gloo rehearsal with single-gpu pass"
CUKING_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --samples 20000 --sites 100000 > gpurun_out/bench_reh2.json 2> gpurun_out/bench_reh2.err; echo rc=$?; tail -c 200 gpurun_out/bench_reh2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_reh2.json').read().strip().splitlines()[-1])
print(d['value'], d['single_gpu_same_run'], d['config']['tile_range_balance'])
PY
