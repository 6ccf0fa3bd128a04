# Timeline of a launch (diagnostic build), whole tiles only and with the
# remainder cut into pieces; then plain timings of both.
cd $GRAFT_REPO_ROOT
tl() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" 2>&1 >/dev/null | grep timeline; }
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_TIMELINE" python -m cuking_amd.build --lib --force > /dev/null 2>&1
for n in 10000; do echo "== whole tiles only, n=$n"; tl --samples $n --sites 100000 --kin-threshold 0.05 --steps 3 --warmup 1; done
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_TIMELINE -DCUKING_SPLIT_ROUNDS=64" python -m cuking_amd.build --lib --force > /dev/null 2>&1
for n in 10000; do echo "== split, n=$n"; tl --samples $n --sites 100000 --kin-threshold 0.05 --steps 3 --warmup 1; done
for flags in "" "-DCUKING_SPLIT_ROUNDS=64"; do
  CUKING_EXTRA_HIPFLAGS="$flags" python -m cuking_amd.build --lib --force > /dev/null 2>&1
  echo "== build flags: '$flags'"
  for n in 9856 10000 10240 10496; do run --samples $n --sites 100000 --kin-threshold 0.05 --steps 20 --warmup 3; done
done
python -m cuking_amd.build --lib --force > /dev/null 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
