# Band height at configs[1] again, now that the remainder goes out as pieces (same box).
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for rep in 1 2; do
for b in 0 3 5 9 17 33; do echo -n "band_rows $b: "; run --steps 30 --warmup 5 --band-rows $b; done
done
echo "20000 x 100000 (49 rounds):"
for b in 5 17; do echo -n "band_rows $b: "; run --samples 20000 --sites 100000 --steps 10 --warmup 2 --band-rows $b; done
