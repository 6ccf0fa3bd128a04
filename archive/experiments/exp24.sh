# LDS-DMA address arithmetic out of the request gaps: previous build (tools/lib_old.so) vs this one.
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1 || exit 1
cp cuking_amd/libcuking_amd.so /tmp/lib_new.so
for which in new old new old; do
  if [ $which = old ]; then cp tools/lib_old.so cuking_amd/libcuking_amd.so; else cp /tmp/lib_new.so cuking_amd/libcuking_amd.so; fi
  echo "== $which: c1 x2 / c2 / full c1"
  run --steps 30 --warmup 5; run --steps 30 --warmup 5
  run --config c2 --steps 2 --warmup 1
  run --steps 20 --warmup 3 --counts-mode 1
done
cp /tmp/lib_new.so cuking_amd/libcuking_amd.so
