# Full form and the VALU kernel with the final build (one box), plus the default line.
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
echo "lean c1:"; run --steps 20 --warmup 3
echo "full c1:"; run --steps 20 --warmup 3 --counts-mode 1
echo "full c2:"; run --config c2 --steps 2 --warmup 1 --counts-mode 1
echo "lean c1, thr -1 (every pair emitted, 5e7 records):"; run --steps 3 --warmup 1 --kin-threshold -1 --max-results 60000000 --counts-mode 0
echo "full c1, thr -1:"; run --steps 3 --warmup 1 --kin-threshold -1 --max-results 60000000 --counts-mode 1
echo "lean c1, thr 0.0 :"; run --steps 3 --warmup 1 --kin-threshold 0.0 --max-results 60000000 --counts-mode 0
echo "full c1, thr 0.0 :"; run --steps 3 --warmup 1 --kin-threshold 0.0 --max-results 60000000 --counts-mode 1
