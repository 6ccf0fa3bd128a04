# A/B of where (and how spread out) a k-step's LDS reads are issued; one box.
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for round in 1 2; do
for m in 0 1 2 3; do
  CUKING_EXTRA_HIPFLAGS="-DCUKING_RD_MODE=$m" python -m cuking_amd.build --lib --force > /dev/null 2>&1
  echo "RD_MODE=$m:"; for i in 1 2; do run --steps 30 --warmup 5; done; run --config c2 --steps 3 --warmup 1
done
done
for m in 0 2 3; do
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS -DCUKING_RD_MODE=$m" python -m cuking_amd.build --lib --force > /dev/null 2>&1
echo "stamps RD_MODE=$m"; python bench.py --config c2 --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 2 --warmup 1 2>&1 >/dev/null | grep "mfma stamps"
done
python -m cuking_amd.build --lib --force > /dev/null 2>&1
