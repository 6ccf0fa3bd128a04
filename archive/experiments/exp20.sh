# Remainder pieces taken from the counter: piece count sweep at configs[1]-like sizes.
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for w in 0 256 384 512; do
  echo "== split_wgs $w"
  for n in 9856 10000 10240; do run --samples $n --sites 100000 --kin-threshold 0.05 --steps 20 --warmup 3 --split-wgs $w; done
done
