# Lean form: 8 vs 10 LDS stages (prefetch distance 3.5 vs 5.5 k-steps at the hand-over). Same box.
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for st in 10 8 10 8; do
  CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_PAIRED_STAGES=$st" python -m cuking_amd.build --lib --force > /dev/null 2>&1
  echo "== $st stages: c1 x2 / c2 / 40000"
  if [ $st = 10 ]; then timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1 || exit 1; fi
  run --steps 30 --warmup 5; run --steps 30 --warmup 5
  run --config c2 --steps 2 --warmup 1
  run --samples 40000 --sites 100000 --steps 3 --warmup 1
done
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS -DCUKING_MFMA_PAIRED_STAGES=10" python -m cuking_amd.build --lib --force > /dev/null 2>&1
python bench.py --config c2 --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 2 --warmup 1 2>&1 >/dev/null | grep "mfma stamps"
python -m cuking_amd.build --lib --force > /dev/null 2>&1
