cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for s in 1 2 3; do timeout -k 10 400 python tools/fuzz_split.py $s 60 2>&1 | tail -1; done
for s in 31 32; do timeout -k 10 400 python tools/fuzz_gpu.py $s 600 2>&1 | tail -1; done
python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 20 --warmup 3 --counts-mode 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full c1', d['roofline']['kernel_ms'])"
python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lean c1', d['roofline']['kernel_ms'])"
