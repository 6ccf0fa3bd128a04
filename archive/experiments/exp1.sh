set -e
cd $GRAFT_REPO_ROOT
./tools/micro/mfma_k4 > gpurun_out/micro_k4.txt 2>&1 || true
./tools/micro/mfma_fp4 >> gpurun_out/micro_k4.txt 2>&1 || true
cat gpurun_out/micro_k4.txt
L="--cpu-seconds 0 --extra-configs none --no-clock-pass --steps 20 --warmup 3"
for sw in 0 1; do for br in 4 6 8 17; do
  echo "swizzle=$sw band_rows=$br" >> gpurun_out/exp_xcd.txt
  python bench.py $L --xcd-swizzle $sw --band-rows $br | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['value'])" >> gpurun_out/exp_xcd.txt
done; done
for sw in 0 1; do for br in 4 17; do
  echo "c2 swizzle=$sw band_rows=$br" >> gpurun_out/exp_xcd.txt
  python bench.py --config c2 --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 3 --warmup 1 --xcd-swizzle $sw --band-rows $br | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['value'])" >> gpurun_out/exp_xcd.txt
done; done
cat gpurun_out/exp_xcd.txt
