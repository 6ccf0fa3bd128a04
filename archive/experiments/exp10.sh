cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 30 --warmup 5 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for i in 1 2; do
for w in 256 512 400 768 1024; do echo -n "c1 split_wgs=$w: "; run --split-wgs $w; done
done
echo -n "c1 split_wgs=512 no swizzle: "; run --split-wgs 512 --xcd-swizzle 0
echo -n "c1 split_wgs=256 no swizzle: "; run --split-wgs 256 --xcd-swizzle 0
