set -e
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for br in 2 3 4 5 6 8 12; do echo -n "c2 sw=1 br=$br: " >> gpurun_out/exp_xcd2.txt; run --config c2 --steps 3 --warmup 1 --xcd-swizzle 1 --band-rows $br >> gpurun_out/exp_xcd2.txt; done
echo -n "c2 sw=0 br=17: " >> gpurun_out/exp_xcd2.txt; run --config c2 --steps 3 --warmup 1 --xcd-swizzle 0 --band-rows 17 >> gpurun_out/exp_xcd2.txt
for br in 3 4 6; do echo -n "100kx150k sw=1 br=$br: " >> gpurun_out/exp_xcd2.txt; run --samples 100000 --sites 150000 --steps 2 --warmup 1 --xcd-swizzle 1 --band-rows $br >> gpurun_out/exp_xcd2.txt; done
echo -n "100kx150k sw=0 br=17: " >> gpurun_out/exp_xcd2.txt; run --samples 100000 --sites 150000 --steps 2 --warmup 1 --xcd-swizzle 0 --band-rows 17 >> gpurun_out/exp_xcd2.txt
for br in 3 4 6; do echo -n "60kx200k sw=1 br=$br: " >> gpurun_out/exp_xcd2.txt; run --samples 60000 --sites 200000 --steps 2 --warmup 1 --xcd-swizzle 1 --band-rows $br >> gpurun_out/exp_xcd2.txt; done
echo -n "60kx200k sw=0 br=17: " >> gpurun_out/exp_xcd2.txt; run --samples 60000 --sites 200000 --steps 2 --warmup 1 --xcd-swizzle 0 --band-rows 17 >> gpurun_out/exp_xcd2.txt
for i in 1 2 3; do for cfg in "0 17" "1 4" "1 17"; do set -- $cfg; echo -n "c1 sw=$1 br=$2: " >> gpurun_out/exp_xcd2.txt; run --steps 30 --warmup 5 --xcd-swizzle $1 --band-rows $2 >> gpurun_out/exp_xcd2.txt; done; done
for cfg in "0 17" "1 4"; do set -- $cfg; echo -n "30kx100k sw=$1 br=$2: " >> gpurun_out/exp_xcd2.txt; run --samples 30000 --steps 5 --warmup 2 --xcd-swizzle $1 --band-rows $2 >> gpurun_out/exp_xcd2.txt; done
cat gpurun_out/exp_xcd2.txt
