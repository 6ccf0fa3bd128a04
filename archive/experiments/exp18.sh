# Per-XCD progress of a launch (diagnostic build): is the skew the XCD's or the patches'?
cd $GRAFT_REPO_ROOT
tl() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" 2>&1 >/dev/null | grep "timeline: XCD"; }
for x in 0 1 3; do
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_TIMELINE -DCUKING_XCD_XOR=$x" python -m cuking_amd.build --lib --force > /dev/null 2>&1
echo "== 40000 x 100000, XCD x takes the patches of x ^ $x"; tl --samples 40000 --sites 100000 --steps 2 --warmup 1
done
python -m cuking_amd.build --lib --force > /dev/null 2>&1
