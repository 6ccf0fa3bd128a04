# Full form: per-pair out-of-line append (tools/lib_old.so, if present) vs one
# reservation per wavefront and tile (same box).
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
cp cuking_amd/libcuking_amd.so /tmp/lib_new.so
for which in new old new; do
  if [ $which = old ]; then [ -f tools/lib_old.so ] || continue; cp tools/lib_old.so cuking_amd/libcuking_amd.so; else cp /tmp/lib_new.so cuking_amd/libcuking_amd.so; fi
  echo "== $which: full c1 / full c2 / full c1 thr 0 (2.5e7 records) / full c1 thr -1 (5e7 records)"
  run --steps 20 --warmup 3 --counts-mode 1
  run --config c2 --steps 2 --warmup 1 --counts-mode 1
  run --steps 3 --warmup 1 --kin-threshold 0.0 --max-results 60000000 --counts-mode 1
  run --steps 3 --warmup 1 --kin-threshold -1 --max-results 60000000 --counts-mode 1
done
cp /tmp/lib_new.so cuking_amd/libcuking_amd.so
