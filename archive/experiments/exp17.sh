# Remainder pieces: how many (same box).
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
CUKING_EXTRA_HIPFLAGS="-DCUKING_SPLIT_ROUNDS=64" python -m cuking_amd.build --lib --force > /dev/null 2>&1
for w in 0 256 512 768 1024; do
  echo "== split_wgs $w"
  for n in 10000 10240; do run --samples $n --sites 100000 --kin-threshold 0.05 --steps 20 --warmup 3 --split-wgs $w; done
done
python -m cuking_amd.build --lib --force > /dev/null 2>&1
