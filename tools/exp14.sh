cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
echo "new (loads in f2):"; for i in 1 2 3; do run --steps 30 --warmup 5; done; run --config c2 --steps 3 --warmup 1
git stash -q; python -m cuking_amd.build --lib --force > /dev/null 2>&1
echo "old (loads in f3):"; for i in 1 2 3; do run --steps 30 --warmup 5; done; run --config c2 --steps 3 --warmup 1
git stash pop -q; python -m cuking_amd.build --lib --force > /dev/null 2>&1
echo "new again:"; for i in 1 2; do run --steps 30 --warmup 5; done; run --config c2 --steps 3 --warmup 1
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS" python -m cuking_amd.build --lib --force > /dev/null 2>&1
python bench.py --config c2 --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 2 --warmup 1 2>&1 >/dev/null | grep "mfma stamps"
python -m cuking_amd.build --lib --force > /dev/null 2>&1
