cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for s in 1 2; do timeout -k 10 300 python tools/fuzz_split.py $s 60 2>&1 | tail -1; done
timeout -k 10 300 python tools/fuzz_gpu.py 77 500 2>&1 | tail -1
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
echo "paired:"; for i in 1 2 3; do run --steps 30 --warmup 5; done; run --config c2 --steps 3 --warmup 1
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_PAIRED=0" python -m cuking_amd.build --lib --force > /dev/null 2>&1
echo "unpaired (6 stages, barrier per k-step):"; for i in 1 2 3; do run --steps 30 --warmup 5; done; run --config c2 --steps 3 --warmup 1
python -m cuking_amd.build --lib --force > /dev/null 2>&1
echo "paired again:"; for i in 1 2; do run --steps 30 --warmup 5; done; run --config c2 --steps 3 --warmup 1
