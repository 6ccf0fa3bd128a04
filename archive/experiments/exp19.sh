# Dynamic tail on/off, same box.
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dynamic_tail or tile_order" 2>&1 | tail -2
for rep in 1 2; do
for d in 0 16384; do
  echo "== dyn_tail_tiles $d"
  CUKING_AMD_DYN_TAIL_TILES=$d run --samples 40000 --sites 100000 --steps 3 --warmup 1
  CUKING_AMD_DYN_TAIL_TILES=$d run --config c2 --steps 2 --warmup 1
done
done
