# With the request gaps cleaned up: where the LDS reads go, and paired vs unpaired, again (same box).
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for flags in "-DCUKING_MFMA_PAIRED_STAGES=10" "-DCUKING_MFMA_PAIRED_STAGES=10 -DCUKING_RD_MODE=3" "-DCUKING_MFMA_PAIRED_STAGES=10 -DCUKING_RD_MODE=1" "-DCUKING_MFMA_PAIRED=0" "-DCUKING_MFMA_PAIRED_STAGES=10"; do
  CUKING_EXTRA_HIPFLAGS="$flags" python -m cuking_amd.build --lib --force > /dev/null 2>&1
  echo "== $flags: c1 x2 / c2"
  run --steps 30 --warmup 5; run --steps 30 --warmup 5
  run --config c2 --steps 2 --warmup 1
done
python -m cuking_amd.build --lib --force > /dev/null 2>&1
