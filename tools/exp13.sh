cd $GRAFT_REPO_ROOT
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS" python -m cuking_amd.build --lib --force > /dev/null 2>&1
python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 10 --warmup 2 2>&1 >/dev/null | grep "mfma stamps"
python bench.py --config c2 --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 2 --warmup 1 2>&1 >/dev/null | grep "mfma stamps"
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS -DCUKING_MFMA_PAIRED=0" python -m cuking_amd.build --lib --force > /dev/null 2>&1
echo unpaired:; python bench.py --config c2 --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 2 --warmup 1 2>&1 >/dev/null | grep "mfma stamps"
python -m cuking_amd.build --lib --force > /dev/null 2>&1
