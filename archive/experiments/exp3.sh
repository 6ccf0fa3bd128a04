set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp_xcd3.txt
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for i in 1 2; do for br in 5 8 12 17 24 32 40; do echo -n "c1 sw=1 br=$br: " >> $O; run --steps 30 --warmup 5 --xcd-swizzle 1 --band-rows $br >> $O; done; done
for br in 5 8 12 17 24; do echo -n "30k sw=1 br=$br: " >> $O; run --samples 30000 --steps 5 --warmup 2 --xcd-swizzle 1 --band-rows $br >> $O; done
for br in 5 17; do echo -n "20k sw=1 br=$br: " >> $O; run --samples 20000 --steps 8 --warmup 2 --xcd-swizzle 1 --band-rows $br >> $O; done
echo -n "20k sw=0 br=17: " >> $O; run --samples 20000 --steps 8 --warmup 2 --xcd-swizzle 0 --band-rows 17 >> $O
for br in 5 17; do echo -n "5k sw=1 br=$br: " >> $O; run --samples 5000 --steps 30 --warmup 5 --xcd-swizzle 1 --band-rows $br >> $O; done
echo -n "5k sw=0 br=17: " >> $O; run --samples 5000 --steps 30 --warmup 5 --xcd-swizzle 0 --band-rows 17 >> $O
echo -n "c3 sw=1 br=5: " >> $O; run --config c3 --steps 1 --warmup 0 --xcd-swizzle 1 --band-rows 5 >> $O
echo -n "c3 sw=0 br=17: " >> $O; run --config c3 --steps 1 --warmup 0 --xcd-swizzle 0 --band-rows 17 >> $O
cat $O
