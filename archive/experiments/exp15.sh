# Tail of the last round at configs[1]-like sizes: time vs tile count, with and
# without the remainder split beyond 8 rounds.  One box.
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for flags in "" "-DCUKING_SPLIT_ROUNDS=64"; do
  CUKING_EXTRA_HIPFLAGS="$flags" python -m cuking_amd.build --lib --force > /dev/null 2>&1
  echo "== build flags: '$flags'"
  for n in 9856 9984 10000 10112 10240 10368 10496; do
    echo -n "n=$n: "; run --samples $n --sites 100000 --kin-threshold 0.05 --steps 20 --warmup 3
  done
done
python -m cuking_amd.build --lib --force > /dev/null 2>&1
