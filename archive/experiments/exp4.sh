set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gputest_d.log 2>&1 || true
tail -5 gpurun_out/gputest_d.log
O=gpurun_out/exp_full.txt
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3), d['roofline'].get('form'))"; }
for i in 1 2; do
echo -n "c1 lean: " >> $O; run --steps 30 --warmup 5 >> $O
echo -n "c1 full: " >> $O; run --steps 30 --warmup 5 --counts-mode 1 >> $O
done
echo -n "c1 thr 0.004 (auto full): " >> $O; run --steps 10 --warmup 2 --kin-threshold 0.004 --max-results 4000000 --no-check >> $O
echo -n "3000 full thr 0.004: " >> $O; run --samples 3000 --steps 30 --warmup 5 --kin-threshold 0.004 --no-check >> $O
echo -n "c2 full: " >> $O; run --config c2 --steps 2 --warmup 1 --counts-mode 1 >> $O
cat $O
